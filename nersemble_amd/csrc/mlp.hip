// mlp.hip -- fully fused small MLPs (tcnn FullyFusedMLP equivalents) on gfx950 MFMA.
//
// Replaces tcnn.NetworkWithInputEncoding / tcnn.Network as used by the reference field
// (src/nersemble/nerfstudio/fields/nersemble_nerfacto_field.py:142-153 `mlp_base` 32->64->16, :162-172
// `mlp_head` 18(pad 32)->64->64->3(pad 16) Sigmoid; calls at :285 and :377).
//
// Semantics (tcnn FullyFusedMLP, restated from the published design): fp16 weights without
// biases, row-major [out][in] matrices stored back to back in one flat parameter vector (first layer
// [64][in_pad], hidden [64][64], last [16][64]); input zero-padded to 32, output padded to 16; ReLU hidden
// activations stored in fp16; output activation None / Sigmoid.  Accumulation here is fp32 in the MFMA
// (tcnn accumulates in fp16 on tensor cores) -> ~1e-2 relative agreement is the stated tolerance.
//
// MI355X design
//   * v_mfma_f32_32x32x16_f16, D[neuron][sample] = W[neuron][k] * X[k][sample]: one wave owns a tile of 32
//     samples; lane = (sample n = lane&31, k-half = lane>>5).  The accumulator layout of layer l (lane holds
//     16 neurons of ONE sample) is exactly the B-operand layout of layer l+1 up to a permutation of k, and
//     k is a contraction index -> the weight fragments are stored pre-permuted in LDS and the activations
//     never leave registers between layers (no LDS/HBM round trip, no cross-lane traffic).
//   * backward recomputes the forward in-register (inputs are 64 B/sample; saving [B][64] activations like
//     tcnn would triple the HBM traffic), chains dZ through W^T fragments the same way, and forms weight
//     gradients with samples as the contraction index: dZ and H tiles are transposed through a padded
//     (conflict-free) per-wave LDS tile, multiplied on MFMA, accumulated in registers across all tiles of
//     the wave and reduced block-wide before one fp32 atomic per parameter per block.
//   * these kernels are memory/latency bound (<= 16 MFMAs per 32 samples): MFMA time is noise next to the
//     hash gather, so the design optimises bytes and launches, not matrix-core utilisation.
#include "mlp_device.h"

namespace nsx {

// ------------------------------------------------------------------------------------------------
template <int NH>
__global__ __launch_bounds__(MLP_WAVES * kWave) void mlp_fwd_kernel(const half_t* __restrict__ W, MlpIO io, int64_t B,
                                                                   int n_out, int out_act, half_t* __restrict__ out,
                                                                   int64_t out_stride, int64_t n_tiles,
                                                                   const int64_t* __restrict__ n_dev) {
    NSX_DEVICE_COUNT(B, n_tiles, 32, n_dev);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    f16x8* frags = reinterpret_cast<f16x8*>(smem_raw);
    const FragPlan p = make_plan(NH, false);
    // (the staging copy behind the fragments is only touched for a weight vector that is not 16-byte aligned)
    stage_weights<NH>(W, reinterpret_cast<half_t*>(smem_raw + (size_t)p.total * kWave * sizeof(f16x8)), frags, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, kb = lane >> 5;
    const int fast = input_mode(io);
    // the two 4-neuron runs a lane holds leave as two 8-byte stores when the whole 16-wide row is wanted (mlp_base)
    const bool wide_out = n_out == MLP_OUT && out_act == 0 && (out_stride % 4) == 0 &&
                          (reinterpret_cast<uintptr_t>(out) & 7) == 0;
    const int64_t tile_step = (int64_t)gridDim.x * MLP_WAVES;
    int64_t tile = (int64_t)blockIdx.x * MLP_WAVES + wave;
    f16x8 xn[2];
    if (tile < n_tiles) {
        const int64_t b0 = tile * 32 + n;
        load_input(io, b0 < B ? b0 : B - 1, kb, fast, xn);
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t b_raw = tile * 32 + n;
        const int64_t b = b_raw < B ? b_raw : B - 1;
        f16x8 x[2], h1[4], h2[4];
        x[0] = xn[0]; x[1] = xn[1];
        if (tile + tile_step < n_tiles) {          // the next tile's input is in flight while this tile's MFMAs chain
            const int64_t bn = (tile + tile_step) * 32 + n;
            load_input(io, bn < B ? bn : B - 1, kb, fast, xn);
        }
        f32x16 o = forward_tile<NH>(frags, p, lane, x, h1, h2);
        if (b_raw < B) {
            // rows held by this lane: r = 0..7 -> neurons (r&3) + 8*(r>>2) + 4*kb  (two runs of 4 consecutive)
            if (wide_out) {
                f16x4 lo, hi;
#pragma unroll
                for (int r = 0; r < 4; ++r) { lo[r] = (half_t)o[r]; hi[r] = (half_t)o[4 + r]; }
                half_t* row = out + b * out_stride + 4 * kb;
                *reinterpret_cast<f16x4*>(row) = lo;
                *reinterpret_cast<f16x4*>(row + 8) = hi;
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int m = acc_row(r, kb);
                    if (m < n_out) {
                        float v = o[r];
                        if (out_act == 1) v = sigmoidf_(v);
                        out[b * out_stride + m] = (half_t)v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int TP = 40;   // transposed tile pitch in halfs (32 samples + 8 pad = 80 B: conflict-free b128 row reads)

// write accumulator-layout values (lane = sample, regs = 16 rows of tile mt) transposed into T[neuron][sample]
__device__ __forceinline__ void stage_T(half_t* T, int mt, int n, int kb, const f16x8& lo, const f16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        T[(32 * mt + acc_row(j, kb)) * TP + n] = lo[j];
        T[(32 * mt + acc_row(8 + j, kb)) * TP + n] = hi[j];
    }
}
__device__ __forceinline__ f16x8 read_T(const half_t* T, int row, int t, int kb) {
    return *reinterpret_cast<const f16x8*>(T + row * TP + 16 * t + 8 * kb);
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// accumulators -> two packed fp16 fragments (registers 0..7 | 8..15), no activation
__device__ __forceinline__ void pack16(const f32x16& d, f16x8& lo, f16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lo[j] = (half_t)d[j]; hi[j] = (half_t)d[8 + j]; }
}
// dZ = dA * relu'(a): accumulators masked by the sign of the (packed) activations they belong to
__device__ __forceinline__ void mask_pack16(const f32x16& d, const f16x8& alo, const f16x8& ahi, f16x8& lo, f16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = ((float)alo[j] > 0.f) ? (half_t)d[j] : (half_t)0.f;
        hi[j] = ((float)ahi[j] > 0.f) ? (half_t)d[8 + j] : (half_t)0.f;
    }
}

// Round 6: no LDS transposes.  The weight gradients are products contracted over the SAMPLES of a tile,
//   dW[o][i] += sum_s dZ[o][s] * H[i][s],
// i.e. both MFMA operands are wanted with lane = neuron and the 8 k-elements = samples -- the transpose of what the chained
// forward / backward hold (lane = sample, elements = neurons).  Rounds 1-5 wrote every dZ and H tile to a per-wave LDS tile
// element by element (~190 ds_write_b16 per lane and tile, two wave barriers per product) and read it back transposed: 23-28
// VALU instructions per MFMA at 6-9 % matrix-core duty.  But the transposed tiles are themselves MFMA results of the SAME
// fragments with the operands swapped: the A fragment (lane = row, 8 k) and the B fragment (lane = column, 8 k) of a
// 32x32x16 MFMA have the same register layout, so
//   H^T [sample][neuron] = X^T [sample][k] * W^T [k][neuron]  =  mfma(a = x fragment, b = weight fragment)
// comes out with lane = neuron and registers = the 16 samples acc_row(r, kb) -- which, packed, ARE the two K-step fragments
// (samples acc_row(8 tt + j, kb)) of the weight-gradient product.  Any assignment of samples to k-slots is fine as long as both
// operands use the same one, and every transposed tile here has this one.  The input tile is transposed by a product with
// an identity fragment (exact: fp16 values times one).  ~2 x the MFMAs of a kernel whose matrix cores were idle, no LDS
// traffic besides the weight fragments.
template <int NH>
__global__ __launch_bounds__(MLP_WAVES * kWave) void mlp_bwd_kernel(
    const half_t* __restrict__ W, MlpIO io, int64_t B, int n_out, int out_act, const half_t* __restrict__ dout,
    int64_t dout_stride, float* __restrict__ dW, float* __restrict__ dA, half_t* __restrict__ dBsrc, int64_t n_tiles,
    const int64_t* __restrict__ n_dev, float* __restrict__ dB32) {
    NSX_DEVICE_COUNT(B, n_tiles, 32, n_dev);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    const FragPlan p = make_plan(NH, true);
    f16x8* frags = reinterpret_cast<f16x8*>(smem_raw);
    half_t* tbase = reinterpret_cast<half_t*>(smem_raw + (size_t)p.total * kWave * sizeof(f16x8));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_weights<NH>(W, tbase, frags, true);          // (the region behind the fragments is the staging buffer)
    __syncthreads();
    const int n = lane & 31, kb = lane >> 5;
    const int fast = input_mode(io);

    // weight-gradient accumulators (D[o][i] tiles): Wo: 1x2, Wh: 2x2, W0: 2x1
    f32x16 gWo[2], gWh[2][2], gW0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { gWo[i] = zero16(); gW0[i] = zero16(); gWh[i][0] = zero16(); gWh[i][1] = zero16(); }
    // identity fragments: B[k-slot][n] = (input feature of the slot == n); K-step t holds features 16 t + 8 kb + j
    f16x8 idn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) idn[t][j] = (kmap_natural(t, kb, j) == n) ? (half_t)1.f : (half_t)0.f;

    const int64_t tile_step = (int64_t)gridDim.x * MLP_WAVES;
    int64_t tile = (int64_t)blockIdx.x * MLP_WAVES + wave;
    f16x8 xn[2];
    if (tile < n_tiles) {
        const int64_t b0 = tile * 32 + n;
        load_input(io, b0 < B ? b0 : B - 1, kb, fast, xn);
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t b_raw = tile * 32 + n;
        const bool valid = b_raw < B;
        const int64_t b = valid ? b_raw : B - 1;
        // (the weight fragments are re-read from LDS where they are used: hoisted out of the tile loop they alone are 144
        // registers, and the kernel spilled 104)
        asm volatile("" ::: "memory");
        f16x8 x[2], h1[4], h2[4];
        x[0] = xn[0]; x[1] = xn[1];
        if (tile + tile_step < n_tiles) {          // the next tile's input is in flight while this tile's chains run
            const int64_t bn = (tile + tile_step) * 32 + n;
            load_input(io, bn < B ? bn : B - 1, kb, fast, xn);
        }
        if (!valid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[0][j] = (half_t)0.f; x[1][j] = (half_t)0.f; }
        }
        // ---- forward, both orientations -------------------------------------------------------------------------------
        const f32x16 o = forward_tile<NH>(frags, p, lane, x, h1, h2);          // lane = sample
        f16x8 h1T[2][2], h2T[2][2];                                           // [neuron tile][K-step]: lane = neuron
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x16 zT = zero16();
#pragma unroll
            for (int t = 0; t < 2; ++t) zT = mfma(x[t], frags[(p.w0 + mt * 2 + t) * kWave + lane], zT);
            relu_pack(zT, h1T[mt][0], h1T[mt][1]);
        }
        if constexpr (NH == 1) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 zT = zero16();
#pragma unroll
                for (int t = 0; t < 4; ++t) zT = mfma(h1[t], frags[(p.wh + mt * 4 + t) * kWave + lane], zT);
                relu_pack(zT, h2T[mt][0], h2T[mt][1]);
            }
        }
        const f16x8* hl = (NH == 1) ? h2 : h1;
        const f16x8 (*hlT)[2] = (NH == 1) ? h2T : h1T;
        f16x8 xT[2];                                                          // lane = input feature
        {
            f32x16 xa = zero16();
#pragma unroll
            for (int t = 0; t < 2; ++t) xa = mfma(x[t], idn[t], xa);
            pack16(xa, xT[0], xT[1]);
        }
        // ---- dZ_out, both orientations --------------------------------------------------------------------------------
        f16x8 dzo;                                                            // lane = sample, rows 0..15 in regs 0..7
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = acc_row(r, kb);
            float gv = 0.f;
            if (valid && m < n_out) {
                gv = (float)dout[b * dout_stride + m];
                if (out_act == 1) { const float y = sigmoidf_(o[r]); gv *= y * (1.0f - y); }
            }
            dzo[r] = (half_t)gv;
        }
        f16x8 dzoT[2];                                                        // lane = output neuron n, 16 samples
        {
            f32x16 oT = zero16();
            if (out_act == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) oT = mfma(hl[t], frags[(p.wo + t) * kWave + lane], oT);
            }
            const int64_t s0 = tile * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t bs = s0 + acc_row(r, kb);
                float gv = 0.f;
                if (n < n_out && bs < B) {
                    gv = (float)dout[bs * dout_stride + n];
                    if (out_act == 1) { const float y = sigmoidf_(oT[r]); gv *= y * (1.0f - y); }
                }
                if (r < 8) dzoT[0][r] = (half_t)gv; else dzoT[1][r - 8] = (half_t)gv;
            }
        }
        // ---- dWo += dZo * Hl^T ----------------------------------------------------------------------------------------
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) gWo[nt] = mfma(dzoT[tt], hlT[nt][tt], gWo[nt]);
        // ---- dHl = Wo^T dZo ; dZl = dHl * relu'(Hl), both orientations --------------------------------------------------
        f16x8 dz[4], dzT[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f16x8 wf = frags[(p.woT + mt) * kWave + lane];
            const f32x16 d = mfma(wf, dzo, zero16());
            mask_pack16(d, hl[2 * mt], hl[2 * mt + 1], dz[2 * mt], dz[2 * mt + 1]);
            const f32x16 dT = mfma(dzo, wf, zero16());
            mask_pack16(dT, hlT[mt][0], hlT[mt][1], dzT[mt][0], dzT[mt][1]);
        }
        if constexpr (NH == 1) {
            // ---- dWh += dZ2 * H1^T ----
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) gWh[mt][nt] = mfma(dzT[mt][tt], h1T[nt][tt], gWh[mt][nt]);
            // ---- dH1 = Wh^T dZ2 ; dZ1, both orientations ----
            f16x8 dz1[4], dz1T[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 d = zero16(), dT = zero16();
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f16x8 wf = frags[(p.whT + mt * 4 + t) * kWave + lane];
                    d = mfma(wf, dz[t], d);
                    dT = mfma(dz[t], wf, dT);
                }
                mask_pack16(d, h1[2 * mt], h1[2 * mt + 1], dz1[2 * mt], dz1[2 * mt + 1]);
                mask_pack16(dT, h1T[mt][0], h1T[mt][1], dz1T[mt][0], dz1T[mt][1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[i] = dz1[i];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { dzT[mt][0] = dz1T[mt][0]; dzT[mt][1] = dz1T[mt][1]; }
        }
        // ---- dW0 += dZ1 * X^T ------------------------------------------------------------------------------------------
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) gW0[mt] = mfma(dzT[mt][tt], xT[tt], gW0[mt]);
        // ---- dX = W0^T dZ1 ----
        if (dA || dBsrc || dB32) {
            f32x16 d = zero16();
#pragma unroll
            for (int t = 0; t < 4; ++t) d = mfma(frags[(p.w0T + t) * kWave + lane], dz[t], d);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int k = acc_row(r, kb);                              // input feature index
                    if (k < io.a_dim) {
                        if (dA) dA[b * io.a_dim + k] = d[r] * io.a_mul;
                    } else if (k - io.a_dim < io.b_dim) {
                        if (dBsrc) dBsrc[b * io.b_stride + io.b_off + (k - io.a_dim)] = (half_t)d[r];
                        // the same fp16 value widened (the consumer of this gradient reads fp32: no conversion launch)
                        if (dB32) dB32[b * io.b_dim + (k - io.a_dim)] = (float)(half_t)d[r];
                    }
                }
            }
        }
    }
    // ---- reduce weight gradients over the block's 4 waves, then one atomic per parameter ----
    // Inside ONE wave every accumulator element has its own address, so a wave can store / add its tiles with plain LDS
    // accesses; two buffers, two rounds: waves 0 and 1 store, then waves 2 and 3 add to them.  (LDS float atomics from all
    // four waves at once -- 128 ds_add_f32 per wave -- took 38 us of a 57 us launch on one tile per wave.)
    static_assert(MLP_WAVES == 4, "two-round reduction over four waves");
    __syncthreads();
    constexpr int n_params = MLP_W * MLP_IN + (NH ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W;
    float* red = reinterpret_cast<float*>(smem_raw) + (size_t)(wave & 1) * n_params;      // reuses the LDS: 2 x n_params floats
    auto dump = [&](auto combine) {
        float* r0 = red;
        float* rh = red + MLP_W * MLP_IN;
        float* ro = rh + (NH ? MLP_W * MLP_W : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, kb);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) combine(r0[(32 * mt + row) * MLP_IN + n], gW0[mt][r]);
            if constexpr (NH == 1) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) combine(rh[(32 * mt + row) * MLP_W + 32 * nt + n], gWh[mt][nt][r]);
            }
            if (row < MLP_OUT) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) combine(ro[row * MLP_W + 32 * nt + n], gWo[nt][r]);
            }
        }
    };
    if (wave < 2) dump([](float& dst, float v) { dst = v; });         // every parameter has exactly one owner lane per wave
    __syncthreads();
    if (wave >= 2) dump([](float& dst, float v) { dst += v; });
    __syncthreads();
    const float* ra = reinterpret_cast<const float*>(smem_raw);
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) {
        const float v = ra[i] + ra[n_params + i];
        if (v != 0.f) atomicAdd(&dW[i], v);
    }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (half_t)src[i];
}

static size_t fwd_smem(int NH) {      // fragments + the staging copy of the flat weights
    return (size_t)make_plan(NH, false).total * kWave * sizeof(f16x8) +
           (size_t)(MLP_W * MLP_IN + (NH ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W) * sizeof(half_t);
}
static size_t bwd_smem(int NH) {
    size_t frag = (size_t)make_plan(NH, true).total * kWave * sizeof(f16x8);
    size_t tiles = (size_t)MLP_WAVES * 2 * MLP_W * TP * sizeof(half_t);
    size_t red = 2 * (size_t)(MLP_W * MLP_IN + (NH ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W) * sizeof(float);
    size_t s = frag + tiles;
    return s > red ? s : red;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_mlp_param_count(int n_hidden_mats) {
    return MLP_W * MLP_IN + (n_hidden_mats ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W;
}

static int check_io(const char* who, int n_hidden_mats, int a_dim, int b_dim, int n_out, int out_act,
                    const void* a, const void* b) {
    NSX_REQUIRE(n_hidden_mats == 0 || n_hidden_mats == 1, "%s: n_hidden_mats must be 0 or 1 (got %d)", who, n_hidden_mats);
    NSX_REQUIRE(a_dim >= 0 && b_dim >= 0 && a_dim + b_dim >= 1 && a_dim + b_dim <= MLP_IN,
                "%s: input width %d+%d not in [1,%d]", who, a_dim, b_dim, MLP_IN);
    NSX_REQUIRE(n_out >= 1 && n_out <= MLP_OUT, "%s: n_out=%d not in [1,%d]", who, n_out, MLP_OUT);
    NSX_REQUIRE(out_act == 0 || out_act == 1, "%s: out_act must be 0 (None) or 1 (Sigmoid)", who);
    NSX_REQUIRE((a_dim == 0 || a) && (b_dim == 0 || b), "%s: NULL input segment", who);
    return NSX_OK;
}

int nsx_mlp_fwd(const nsx_half* weights, int n_hidden_mats, int64_t B,
                const float* a, int64_t a_stride, int a_dim, float a_mul, float a_add,
                const nsx_half* b, int64_t b_stride, int b_off, int b_dim,
                int n_out, int out_act, nsx_half* out, int64_t out_stride, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_mlp_fwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(weights && out, "nsx_mlp_fwd: NULL argument");
    if (int rc = check_io("nsx_mlp_fwd", n_hidden_mats, a_dim, b_dim, n_out, out_act, a, b)) return rc;
    MlpIO io{a, a_stride, a_dim, a_mul, a_add, reinterpret_cast<const half_t*>(b), b_stride, b_off, b_dim};
    const int64_t n_tiles = (B + 31) / 32;
    int64_t blocks = (n_tiles + MLP_WAVES - 1) / MLP_WAVES;
    const int64_t cap = (int64_t)num_cus() * 4;
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
    const half_t* W = reinterpret_cast<const half_t*>(weights);
    half_t* o = reinterpret_cast<half_t*>(out);
    if (n_hidden_mats == 0)
        hipLaunchKernelGGL((mlp_fwd_kernel<0>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), fwd_smem(0), st, W, io, B,
                           n_out, out_act, o, out_stride, n_tiles, n_device);
    else
        hipLaunchKernelGGL((mlp_fwd_kernel<1>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), fwd_smem(1), st, W, io, B,
                           n_out, out_act, o, out_stride, n_tiles, n_device);
    NSX_LAUNCH_CHECK("nsx_mlp_fwd launch");
    return NSX_OK;
}

int nsx_mlp_bwd(const nsx_half* weights, int n_hidden_mats, int64_t B,
                const float* a, int64_t a_stride, int a_dim, float a_mul, float a_add,
                const nsx_half* b, int64_t b_stride, int b_off, int b_dim,
                int n_out, int out_act, const nsx_half* dout, int64_t dout_stride,
                float* dweights, float* da, nsx_half* db, float* db_f32, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_mlp_bwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(weights && dout && dweights, "nsx_mlp_bwd: NULL argument");
    if (int rc = check_io("nsx_mlp_bwd", n_hidden_mats, a_dim, b_dim, n_out, out_act, a, b)) return rc;
    MlpIO io{a, a_stride, a_dim, a_mul, a_add, reinterpret_cast<const half_t*>(b), b_stride, b_off, b_dim};
    const int64_t n_tiles = (B + 31) / 32;
    int64_t blocks = (n_tiles + MLP_WAVES - 1) / MLP_WAVES;
    // One block (4 waves) per CU: the backward holds its weight gradients in registers (218 VGPRs + 144 AGPRs with a
    // hidden matrix: one wave per SIMD), so a second block per CU only queues behind the first -- and every block ends with
    // n_params atomics onto the same 7 168 addresses.  Measured in-step (645 k samples, 2 calls per step): 2 blocks per CU
    // 0.191 ms per call, 1: 0.166, 1/2: 0.274 (NSX_OPT_MLP_BWD(0)_HALF_BLOCKS_PER_CU = blocks per CU x 2).
    const int per_cu_x2 = option(NSX_OPT_MLP_BWD_HALF_BLOCKS_PER_CU), per_cu0_x2 = option(NSX_OPT_MLP_BWD0_HALF_BLOCKS_PER_CU);
    const int64_t cap = (int64_t)num_cus() * (n_hidden_mats == 0 ? per_cu0_x2 : per_cu_x2) / 2;
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
    const half_t* W = reinterpret_cast<const half_t*>(weights);
    if (n_hidden_mats == 0)
        hipLaunchKernelGGL((mlp_bwd_kernel<0>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), bwd_smem(0), st, W, io, B,
                           n_out, out_act, reinterpret_cast<const half_t*>(dout), dout_stride, dweights, da,
                           reinterpret_cast<half_t*>(db), n_tiles, n_device, db_f32);
    else
        hipLaunchKernelGGL((mlp_bwd_kernel<1>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), bwd_smem(1), st, W, io, B,
                           n_out, out_act, reinterpret_cast<const half_t*>(dout), dout_stride, dweights, da,
                           reinterpret_cast<half_t*>(db), n_tiles, n_device, db_f32);
    NSX_LAUNCH_CHECK("nsx_mlp_bwd launch");
    return NSX_OK;
}

int nsx_f32_to_f16(const float* src, nsx_half* dst, int64_t n, void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_f32_to_f16: negative size");
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(src && dst, "nsx_f32_to_f16: NULL argument");
    int64_t blocks = (n + 255) / 256;
    if (blocks > num_cus() * 8) blocks = num_cus() * 8;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<half_t*>(dst), n);
    NSX_LAUNCH_CHECK("nsx_f32_to_f16 launch");
    return NSX_OK;
}

}  // extern "C"

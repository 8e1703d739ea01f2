// deform.hip -- fused SE(3) deformation field for gfx950: windowed positional encoding + 6x128 MLP (skip into
// layer 4) + rotation/translation heads + SE(3) exponential map + warp, forward and backward, on MFMA.
//
// Replaces SE3DeformationField.compute_offsets / SE3WarpingField.forward of the reference
// (src/nersemble/nerfstudio/field_components/deformation_field.py:77-116,148-166), which runs
// WindowedNeRFEncoding (windowed_nerf_encoding.py:33-74), torch.cat with the per-sample warp code, eight
// nn.Linear GEMMs under fp16 autocast (nerfstudio MLP), se3_exp_map (util/pytorch3d.py:107-191) and a batched
// 4x4 matmul as ~50 separate launches.  Numerics follow the autocast path: inputs / weights / biases rounded to
// fp16, fp32 accumulation, every layer output rounded to fp16, exponential map and warp in fp32.
//
// MI355X design
//   * v_mfma_f32_32x32x16_f16 with D[neuron][sample]: a wave owns 32 samples; the accumulator layout of one
//     layer is the B-operand layout of the next up to a k-permutation that is folded into pre-packed weight
//     fragments, so the 6 layers + heads chain entirely in registers (same trick as mlp.hip, 4 M-tiles wide).
//   * the warp code is never gathered to [S,128]: each sample carries a slot into the batch's small code table
//     (nersemble_instant_ngp.py:310-316 materialises the gather; its backward was a 46 ms index_put).
//   * weights are packed once per step into MFMA fragment order (fp16); one 8-wave block per CU walks the layers in
//     lock-step while the next layer's fragments are copied L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into the
//     buffer that is not being read -- one barrier per layer, weights cross L2 -> CU once per 256 samples.
//   * backward recomputes the forward (only ReLU bit masks are kept), chains dZ through W^T fragments, and
//     writes dZ / activation tiles in [neuron][sample] order (transposed in registers by an identity-operand MFMA);
//     weight, bias and code-table gradients are then sample-contracted GEMMs (deform_wgrad_kernel: operands staged
//     through a 4-deep LDS ring by LDS-DMA, each scratch byte read once) -- the only form in which K = #samples fits
//     MFMA.
#include "nsx_common.h"

namespace nsx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DFW = 128;        // layer width
constexpr int DF_PE = 45;       // 42 windowed sin/cos + 3 scaled inputs
constexpr int DF_CODE = 128;    // warp code width
constexpr int DF_IN = DF_PE + DF_CODE;      // 173
constexpr int DF_TIN = 11;
constexpr int DF_TW = 8;
constexpr int DF_W4 = DF_IN + DFW;          // 301

// flat fp32 parameter layout (host packs the nn.Linear tensors in this order)
constexpr int P_W0 = 0;
constexpr int P_B0 = P_W0 + DFW * DF_IN;
constexpr int P_W1 = P_B0 + DFW;
constexpr int P_B1 = P_W1 + DFW * DFW;
constexpr int P_W2 = P_B1 + DFW;
constexpr int P_B2 = P_W2 + DFW * DFW;
constexpr int P_W3 = P_B2 + DFW;
constexpr int P_B3 = P_W3 + DFW * DFW;
constexpr int P_W4 = P_B3 + DFW;
constexpr int P_B4 = P_W4 + DFW * DF_W4;
constexpr int P_W5 = P_B4 + DFW;
constexpr int P_B5 = P_W5 + DFW * DFW;
constexpr int P_WR = P_B5 + DFW;            // [3][128]
constexpr int P_BR = P_WR + 3 * DFW;
constexpr int P_WV = P_BR + 3;
constexpr int P_BV = P_WV + 3 * DFW;
constexpr int P_TOTAL = P_BV + 3;           // 127 750

// fragment groups (each fragment = 64 lanes x 8 halfs)
constexpr int F0 = 0;                       // [4][11]
constexpr int F1 = F0 + 44, F2 = F1 + 32, F3 = F2 + 32;
constexpr int F4 = F3 + 32;                 // W4 over the input: [4][11]
constexpr int F4X = F4 + 44;                // W4 over x:         [4][8]
constexpr int F5 = F4X + 32;
constexpr int FH = F5 + 32;                 // [1][8]
constexpr int N_FWD_FRAGS = FH + 8;         // 256
constexpr int BH = N_FWD_FRAGS;             // [4][1]
constexpr int B5 = BH + 4, B4X = B5 + 32, B4C = B4X + 32, B3 = B4C + 32, B2 = B3 + 32, B1 = B2 + 32;
constexpr int B0C = B1 + 32;
constexpr int N_FRAGS = B0C + 32;           // 484
constexpr int N_BIAS = 6 * DFW + 8;         // + heads (6, padded to 8)

__device__ __host__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __host__ __forceinline__ int kchain(int t, int kb, int j) { return 32 * (t >> 1) + acc_row(8 * (t & 1) + j, kb); }
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------------------
// weight packing: fp32 nn.Linear parameters -> fp16 MFMA fragments (+ fp16-rounded biases as fp32)
// ---------------------------------------------------------------------------------------------------------
// The 16 nn.Linear tensors either as ONE flat fp32 vector in the P_* order above or as 16 separate device tensors (what
// nn.Module holds: no torch.cat of the parameters per step).
struct DeformParamSrc {
    const float* flat;          // non-NULL: flat vector
    const float* t[16];         // else: W0 b0 W1 b1 W2 b2 W3 b3 W4 b4 W5 b5 Wr br Wv bv
    __device__ __forceinline__ float at(int flat_index) const {
        if (flat) return flat[flat_index];
        constexpr int starts[17] = {P_W0, P_B0, P_W1, P_B1, P_W2, P_B2, P_W3, P_B3, P_W4, P_B4, P_W5, P_B5,
                                    P_WR, P_BR, P_WV, P_BV, P_TOTAL};
        int k = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) k += (flat_index >= starts[q]) ? 1 : 0;
        return t[k][flat_index - starts[k]];
    }
};

__device__ __forceinline__ float pack_source(const DeformParamSrc& P, int fi, int i, int kb, int j) {
    auto lin = [&](int wbase, int ld, int row, int col) { return P.at(wbase + row * ld + col); };
    if (fi < F1) {                                   // W0 natural k
        const int mt = (fi - F0) / DF_TIN, t = (fi - F0) % DF_TIN, k = 16 * t + 8 * kb + j;
        return k < DF_IN ? lin(P_W0, DF_IN, 32 * mt + i, k) : 0.f;
    }
    if (fi < F4) {                                   // W1..W3
        const int l = (fi - F1) / 32, r = (fi - F1) % 32, mt = r / DF_TW, t = r % DF_TW;
        const int base = l == 0 ? P_W1 : (l == 1 ? P_W2 : P_W3);
        return lin(base, DFW, 32 * mt + i, kchain(t, kb, j));
    }
    if (fi < F4X) {                                  // W4, 11 natural steps over the input
        const int mt = (fi - F4) / DF_TIN, t = (fi - F4) % DF_TIN, k = 16 * t + 8 * kb + j;
        return k < DF_IN ? lin(P_W4, DF_W4, 32 * mt + i, k) : 0.f;
    }
    if (fi < F5) {                                   // W4, 8 chained steps over x
        const int mt = (fi - F4X) / DF_TW, t = (fi - F4X) % DF_TW;
        return lin(P_W4, DF_W4, 32 * mt + i, DF_IN + kchain(t, kb, j));
    }
    if (fi < FH) {                                   // W5
        const int mt = (fi - F5) / DF_TW, t = (fi - F5) % DF_TW;
        return lin(P_W5, DFW, 32 * mt + i, kchain(t, kb, j));
    }
    if (fi < BH) {                                   // heads: rows 0..2 = r, 3..5 = v
        const int t = fi - FH, k = kchain(t, kb, j);
        if (i < 3) return lin(P_WR, DFW, i, k);
        if (i < 6) return lin(P_WV, DFW, i - 3, k);
        return 0.f;
    }
    if (fi < B5) {                                   // heads^T: M = hidden neuron, K-step 0 over head rows
        const int mt = fi - BH, o = kchain(0, kb, j);
        if (o < 3) return lin(P_WR, DFW, o, 32 * mt + i);
        if (o < 6) return lin(P_WV, DFW, o - 3, 32 * mt + i);
        return 0.f;
    }
    const int g = (fi - B5) / 32, r = (fi - B5) % 32, mt = r / DF_TW, t = r % DF_TW, o = kchain(t, kb, j);
    switch (g) {
        case 0: return lin(P_W5, DFW, o, 32 * mt + i);                 // B5
        case 1: return lin(P_W4, DF_W4, o, DF_IN + 32 * mt + i);       // B4x
        case 2: return lin(P_W4, DF_W4, o, DF_PE + 32 * mt + i);       // B4c (code columns)
        case 3: return lin(P_W3, DFW, o, 32 * mt + i);
        case 4: return lin(P_W2, DFW, o, 32 * mt + i);
        case 5: return lin(P_W1, DFW, o, 32 * mt + i);
        default: return lin(P_W0, DF_IN, o, DF_PE + 32 * mt + i);      // B0c
    }
}

__global__ void deform_pack_kernel(DeformParamSrc P, f16x8* __restrict__ frags, float* __restrict__ bias) {
    const int total = N_FRAGS * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int fi = e / 64, ll = e % 64, i = ll & 31, kb = ll >> 5;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)pack_source(P, fi, i, kb, j);
        frags[e] = v;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N_BIAS; e += gridDim.x * blockDim.x) {
        float b = 0.f;
        if (e < 6 * DFW) {
            const int l = e / DFW, n = e % DFW;
            const int base = l == 0 ? P_B0 : l == 1 ? P_B1 : l == 2 ? P_B2 : l == 3 ? P_B3 : l == 4 ? P_B4 : P_B5;
            b = P.at(base + n);
        } else {
            const int o = e - 6 * DFW;
            if (o < 3) b = P.at(P_BR + o);
            else if (o < 6) b = P.at(P_BV + o - 3);
        }
        bias[e] = (float)(half_t)b;            // autocast rounds the bias to fp16
    }
}

// ---------------------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------------------
struct DeformArgs {
    const float* pos;            // [S][3] world positions
    const float* code;           // code rows (fp32)
    int64_t code_stride;
    const int32_t* slot;         // [S] row per sample, or nullptr -> row = sample
    float aabb_min[3], aabb_inv[3], aabb_ext[3];
    float window[7];             // per-frequency window (all ones when windows_param is None)
    const f16x8* frags;
    const float* bias;
    int64_t S;
};

// input fragments of sample b for lane half kb, natural k order (k = 16 t + 8 kb + j):
//   k <  21 : w[f] * sin(2 pi pn_d 2^f)          (k = 7 d + f)        windowed_nerf_encoding.py:47-70
//   k <  42 : w[f] * sin(2 pi pn_d 2^f + pi/2)
//   k <  45 : 2 pi pn_d                           (the 2 pi-SCALED input is appended, :72-73)
//   k < 173 : warp code
// sin(2 pi x) is the native v_sin_f32 (input in revolutions, range-reduced in hardware).
__device__ __forceinline__ void build_input(const DeformArgs& A, int64_t b, int kb, float pn[3], f16x8 x[DF_TIN]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) pn[d] = (A.pos[b * 3 + d] - A.aabb_min[d]) / A.aabb_ext[d];
    const float* crow = A.code + (A.slot ? (int64_t)A.slot[b] : b) * A.code_stride;
    // k depends on the lane half only through + 8 kb: evaluate both compile-time variants and select, so that the
    // frequency / axis / window indices are constants (no dynamically indexed kernel-argument arrays)
    auto pe_value = [&](int k) -> float {
        if (k < 42) {
            const int kk = k < 21 ? k : k - 21;
            const int d = kk / 7, f = kk - 7 * d;
            const float rev = pn[d] * (float)(1 << f) + (k >= 21 ? 0.25f : 0.f);
            return A.window[f] * __builtin_amdgcn_sinf(rev);
        }
        if (k < DF_PE) return 6.283185307179586f * pn[k - 42];
        return 0.f;                                   // code part handled by the caller
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = 16 * t + j, k1 = k0 + 8;
            float v0 = k0 < DF_PE ? pe_value(k0) : 0.f;
            float v1 = k1 < DF_PE ? pe_value(k1) : 0.f;
            float v = kb ? v1 : v0;
            const int k = k0 + 8 * kb;
            if (k1 >= DF_PE) {                         // at least the kb = 1 variant is a code element
                const int kc = k - DF_PE;
                const float cv = crow[kc < 0 ? 0 : kc];
                if (k >= DF_PE) v = cv;
            }
            x[t][j] = (half_t)v;
        }
    }
#pragma unroll
    for (int t = 3; t < DF_TIN; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * t + 8 * kb + j;
            x[t][j] = (k < DF_IN) ? (half_t)crow[k - DF_PE] : (half_t)0.f;
        }
    }
}

// Packed epilogue helpers.  Two accumulator values -> one dword of two halfs (v_cvt_pk_f16_f32, round-to-nearest-even).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    f32x2 v; v[0] = a; v[1] = b;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

// acc (bias already in the accumulator) -> fp16 -> relu -> fragments (chained k order).
// ReLU on the packed halfs is gfx950's v_pk_maximum3_f16(x, 0, 0): the IEEE-754-2019 maximum, which -- unlike
// v_pk_max_f16 -- propagates NaN exactly as torch.relu does (the reference relies on that: a NaN deformation falls back
// to the undeformed point, deformation_field.py:101-102).
// With MASK, also the positions of the non-zero outputs, 1 bit per unit in TWO words per lane: word (D >> 4), bit
// (D & 15) for the low half and 16 + (D & 15) for the high half of fragment dword D = 4 t + j / 2.
template <bool MASK>
__device__ __forceinline__ u32x2 finish_layer(const f32x16 acc[4], f16x8 h[DF_TW]) {
    u32x2 mask; mask[0] = 0u; mask[1] = 0u;
#pragma unroll
    for (int t = 0; t < DF_TW; ++t) {
        u32x4 hv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mt = t >> 1, r = 8 * (t & 1) + 2 * q;
            const uint32_t p = cvt_pk(acc[mt][r], acc[mt][r + 1]);
            uint32_t rl;
            asm("v_pk_maximum3_f16 %0, %1, 0, 0" : "=v"(rl) : "v"(p));
            hv[q] = rl;
            if (MASK) {
                uint32_t nz;                                   // 1 per non-zero half (the compiler would expand a
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(rl), "v"(0x00010001u));   // vector min into cmp + select)
                const int D = 4 * t + q;
                mask[D >> 4] |= nz << (D & 15);
            }
        }
        h[t] = __builtin_bit_cast(f16x8, hv);
    }
    return mask;
}

// ---- weight pipeline: the 8 waves of a block walk the layers in lock-step.  Each layer's fragment group ("stage",
// <= 44 KB) is copied L2 -> LDS by the LDS-DMA path (global_load_lds_dwordx4: one 1-KB fragment per wave-instruction,
// no VGPR round trip) into the buffer that is NOT being read, while the MFMAs of the current stage run; one barrier
// per stage.  A stage is consumed by 8 x 32 samples, so the weights cross L2 -> CU once per 256 samples. ----
constexpr int NW = 8;                                       // waves per block (one block per CU, 2 waves per SIMD)

constexpr int STAGE_FRAGS = 44;                             // largest stage: W0 / W4-input (4 M-tiles x 11 K-steps)

struct DeformLds {
    f16x8 w[2][STAGE_FRAGS * 64];
    float bias[N_BIAS];
};

__device__ __forceinline__ void stage_issue(const f16x8* __restrict__ frags, int first, int count, f16x8* buf) {
    // wave-uniform fragment base (SGPR) + 16 B per lane: the copy needs no per-stage address VGPRs
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t voff = (threadIdx.x & 63u) * 16u;
    const char* base = reinterpret_cast<const char*>(frags) + (size_t)first * 1024;
    for (int f = wave; f < count; f += NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)f * 1024 + voff),
                                         (__attribute__((address_space(3))) void*)(buf + f * 64), 16, 0, 0);
}

// own LDS-DMA copies landed (vmcnt(0)) + everybody finished reading the other buffer
__device__ __forceinline__ void stage_flip(int& cur) {
    __syncthreads();
    cur ^= 1;
}

typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4 lds_cfloat4;

// The bias block never changes after the prologue, so loop-invariant-code motion would hoist every bias read of every
// layer (hundreds of VGPRs) out of the persistent tile loop; laundering the LDS address once per tile keeps them as
// in-loop ds_reads.
__device__ __forceinline__ lds_cfloat* launder_lds(const float* p) {
    lds_cfloat* q = (lds_cfloat*)p;
    asm volatile("" : "+s"(q));
    return q;
}

// accumulators start at the (fp16-rounded) bias: rows acc_row(r, kb) of tile mt are 4 runs of 4 consecutive neurons
__device__ __forceinline__ void acc_init(f32x16 acc[4], lds_cfloat* bias_lds, int kb) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<lds_cfloat4*>(bias_lds + 32 * mt + 8 * q + 4 * kb);
            acc[mt][4 * q + 0] = v.x; acc[mt][4 * q + 1] = v.y; acc[mt][4 * q + 2] = v.z; acc[mt][4 * q + 3] = v.w;
        }
}

// acc[mt] += sum_t W_frag(stage-local index local + mt * KT + t) * in[t], fragments read from LDS one K-step ahead
// (8 fragment registers; two K-steps ahead, pinned with sched_barrier, measured equal in round 4 and was removed)
template <int KT>
__device__ __forceinline__ void gemm_layer_lds(const f16x8* lds, int local, int lane, const f16x8* in, f32x16 acc[4]) {
    constexpr int PF = 1;
    const f16x8* base = lds + (size_t)local * 64 + lane;
    f16x8 a[PF + 1][4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < KT) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[p][mt] = base[(mt * KT + p) * 64];
        }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (t + PF < KT) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[(t + PF) % (PF + 1)][mt] = base[(mt * KT + t + PF) * 64];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma(a[t % (PF + 1)][mt], in[t], acc[mt]);
    }
}

struct Fwd {
    f16x8 x[DF_TIN];
    f16x8 h[DF_TW];              // current activation
    u32x2 m1, m2, m3, m4, m5, m6;   // positions of the positive units of each layer (layout: finish_layer)
    float pn[3];
    float r[3], v[3];            // head outputs (fp16-rounded), valid on both lanes of a sample
};

// ---- transposed tiles for the sample-contracted GEMMs ---------------------------------------------------------
// The weight-gradient GEMMs contract over SAMPLES, so they need operands with lane = neuron and 8 samples per lane --
// the transpose of the chain's fragments (lane = sample, 8 neurons).  The transpose is done on the matrix core:
// D[sample][c] = sum_k X[sample][k] * I[k][c] with the fragment as A operand and a 16x32 selector (identity placed
// in columns 0..15 or 16..31) as B: two K-steps fill one 32-column tile, exactly (products with 1.0, fp32
// accumulation).  The accumulator layout (lane = column/neuron, 16 samples in registers) is stored as is: 32 B per
// lane, fully coalesced -- instead of 64 two-byte stores per activation.
// Tile p of an activation = [64 lanes][16 halfs]; lane (c = lane&31, half), element r <-> sample acc_row(r, half);
// column c of tile p is neuron 32p + c for natural-order inputs and tile_neuron_chain(p, c) for chained fragments.
__device__ __host__ __forceinline__ int tile_neuron_chain(int p, int c) {
    return 32 * p + (c & 3) + 8 * (2 * (c >> 4) + ((c & 7) >> 2)) + 4 * ((c >> 3) & 1);
}

struct TSel { f16x8 lo, hi; };

__device__ __forceinline__ TSel make_tsel(int lane) {
    TSel s;
    const int n = lane & 31, kb = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s.lo[j] = (n == 8 * kb + j) ? (half_t)1.f : (half_t)0.f;
        s.hi[j] = (n == 16 + 8 * kb + j) ? (half_t)1.f : (half_t)0.f;
    }
    return s;
}

// one transposed tile in registers: o0 | o1 = the two K-steps (8 samples each) of lane = column
__device__ __forceinline__ void transpose_pair(const f16x8& f_lo, const f16x8* f_hi, const TSel& sel, u32x4& o0, u32x4& o1) {
    f32x16 d = mfma(f_lo, sel.lo, zero16());
    if (f_hi) d = mfma(*f_hi, sel.hi, d);
#pragma unroll
    for (int r = 0; r < 4; ++r) { o0[r] = cvt_pk(d[2 * r], d[2 * r + 1]); o1[r] = cvt_pk(d[8 + 2 * r], d[9 + 2 * r]); }
}

template <int NF>
__device__ __forceinline__ void store_tile_T(half_t* __restrict__ dst, int lane, const f16x8* frags, const TSel& sel) {
#pragma unroll
    for (int p = 0; p < (NF + 1) / 2; ++p) {
        u32x4 o0, o1;
        transpose_pair(frags[2 * p], (2 * p + 1 < NF) ? &frags[2 * p + 1] : nullptr, sel, o0, o1);
        // tile p = [2 K-steps][64 lanes][8 halfs]: each store instruction writes one contiguous KB, and the weight-gradient
        // kernel can copy the tile into LDS linearly and read it back conflict-free as MFMA operand fragments
        u32x4* out = reinterpret_cast<u32x4*>(dst + (size_t)p * 1024 + (size_t)lane * 8);
        __builtin_nontemporal_store(o0, out);            // written once, read once by the weight-gradient kernel
        __builtin_nontemporal_store(o1, out + 64);
    }
}

// One 32-sample tile through the 6 layers + heads.  On entry stage F0 is resident in L.w[cur]; on exit the stage
// `next_first` (the first stage of whatever follows: F0 of the next tile, or the first backward stage) is resident in
// L.w[cur].  BWD: also keeps the ReLU masks and writes the transposed layer-input tiles for the weight gradients.
// A0F: how many K-steps of the layer input are written as transposed tiles (BWD): all 11 (192 rows: positional encoding +
// warp code), or 4 (64 rows: the positional encoding and the first 19 code columns riding along) when the weight
// gradients of the code columns are formed through the code SLOT (deform_bwd_kernel<true>).
// A6T: write the transposed tile of the heads' input (a6); the SLOTS chain kernel forms the head gradients itself
template <bool BWD, int A0F = DF_TIN, bool A6T = true>
__device__ __forceinline__ void forward_tile(const DeformArgs& A, int64_t b, int lane, Fwd& F, half_t* a_tiles,
                                             DeformLds& L, int& cur, int next_first, int next_count) {
    constexpr int A0_HALFS = ((A0F + 1) / 2) * 32 * 32;         // transposed tiles of the layer input
    const int kb = lane >> 5;
    lds_cfloat* bias = launder_lds(L.bias);
    build_input(A, b, kb, F.pn, F.x);
    TSel tsel;
    if (BWD) {
        tsel = make_tsel(lane);
        store_tile_T<A0F>(a_tiles, lane, F.x, tsel);
    }
    f32x16 acc[4];
    // L0
    stage_issue(A.frags, F1, 32, L.w[cur ^ 1]);
    acc_init(acc, bias + 0 * DFW, kb);
    gemm_layer_lds<DF_TIN>(L.w[cur], 0, lane, F.x, acc);
    F.m1 = finish_layer<BWD>(acc, F.h);
    if (BWD) store_tile_T<DF_TW>(a_tiles + A0_HALFS + 0 * DFW * 32, lane, F.h, tsel);
    stage_flip(cur);
    // L1..L3 (the input fragments are dead once the layer's MFMAs are issued: the output overwrites them)
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        stage_issue(A.frags, l == 1 ? F2 : (l == 2 ? F3 : F4), l == 3 ? 44 : 32, L.w[cur ^ 1]);
        acc_init(acc, bias + l * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, F.h, acc);
        const u32x2 m = finish_layer<BWD>(acc, F.h);
        if (l == 1) F.m2 = m; else if (l == 2) F.m3 = m; else F.m4 = m;
        if (BWD) store_tile_T<DF_TW>(a_tiles + A0_HALFS + l * DFW * 32, lane, F.h, tsel);
        stage_flip(cur);
    }
    // L4: cat[input, x] -- two stages, one accumulator
    stage_issue(A.frags, F4X, 32, L.w[cur ^ 1]);
    acc_init(acc, bias + 4 * DFW, kb);
    gemm_layer_lds<DF_TIN>(L.w[cur], 0, lane, F.x, acc);
    stage_flip(cur);
    stage_issue(A.frags, F5, 40, L.w[cur ^ 1]);
    gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, F.h, acc);
    F.m5 = finish_layer<BWD>(acc, F.h);
    if (BWD) store_tile_T<DF_TW>(a_tiles + A0_HALFS + 4 * DFW * 32, lane, F.h, tsel);
    stage_flip(cur);
    // L5 (+ out_activation ReLU) and the heads share one stage (F5 | FH are contiguous)
    stage_issue(A.frags, next_first, next_count, L.w[cur ^ 1]);
    acc_init(acc, bias + 5 * DFW, kb);
    gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, F.h, acc);
    F.m6 = finish_layer<BWD>(acc, F.h);
    if (BWD && A6T) store_tile_T<DF_TW>(a_tiles + A0_HALFS + 5 * DFW * 32, lane, F.h, tsel);
    // heads (one M-tile, rows 0..5)
    f32x16 o = zero16();
#pragma unroll
    for (int t = 0; t < DF_TW; ++t) o = mfma(L.w[cur][(32 + t) * 64 + lane], F.h[t], o);
    float own[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) own[r] = (float)(half_t)(o[r] + bias[6 * DFW + acc_row(r, kb)]);
    float oth[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(own[r], 32);
    // rows 0..3 live on kb = 0, rows 4..7 on kb = 1
    const float r0 = kb ? oth[0] : own[0], r1 = kb ? oth[1] : own[1], r2 = kb ? oth[2] : own[2];
    const float v0 = kb ? oth[3] : own[3], v1 = kb ? own[0] : oth[0], v2 = kb ? own[1] : oth[1];
    F.r[0] = r0; F.r[1] = r1; F.r[2] = r2;
    F.v[0] = v0; F.v[1] = v1; F.v[2] = v2;
    stage_flip(cur);
}

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

struct Se3Coef { float theta, a, b, c; bool clamped; };

__device__ __forceinline__ Se3Coef se3_coef(const float r[3]) {
    Se3Coef q;
    const float nr = dot3(r, r);
    q.clamped = nr < 1e-4f;
    q.theta = sqrtf(fmaxf(nr, 1e-4f));
    const float inv = 1.0f / q.theta, s = sinf(q.theta), co = cosf(q.theta);
    q.a = inv * s;
    q.b = inv * inv * (1.0f - co);
    q.c = (q.theta - s) / (q.theta * q.theta * q.theta);
    return q;
}

// warped = R p + V v  (Rodrigues with K^2 = r r^T - |r|^2 I)
__device__ __forceinline__ void se3_apply(const float r[3], const float v[3], const float p[3], float out[3]) {
    const Se3Coef q = se3_coef(r);
    float u1[3], u2[3], w1[3], w2[3];
    cross3(r, p, u1); cross3(r, u1, u2);
    cross3(r, v, w1); cross3(r, w1, w2);
#pragma unroll
    for (int d = 0; d < 3; ++d) out[d] = p[d] + q.a * u1[d] + q.b * u2[d] + v[d] + q.b * w1[d] + q.c * w2[d];
}

// ---------------------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_prologue(const DeformArgs& A, DeformLds& L, int first, int count) {
    for (int i = threadIdx.x; i < N_BIAS; i += blockDim.x) L.bias[i] = A.bias[i];
    stage_issue(A.frags, first, count, L.w[0]);
    __syncthreads();
}

__global__ __launch_bounds__(NW * 64, 1) void deform_fwd_kernel(DeformArgs A, float* __restrict__ offsets, int64_t n_tiles,
                                                              const int64_t* __restrict__ n_dev) {
    NSX_DEVICE_COUNT(A.S, n_tiles, 32, n_dev);
    __shared__ __attribute__((aligned(16))) DeformLds L;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_groups = (n_tiles + NW - 1) / NW;
    lds_prologue(A, L, F0, 44);
    int cur = 0;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {        // all waves iterate together
        const int64_t tile = grp * NW + wave;
        const int64_t b_raw = tile * 32 + (lane & 31);
        const int64_t b = b_raw < A.S ? b_raw : A.S - 1;
        Fwd F;
        forward_tile<false>(A, b, lane, F, nullptr, L, cur, F0, 44);
        float w[3];
        se3_apply(F.r, F.v, F.pn, w);
        if (b_raw < A.S && (lane >> 5) == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float wd = (w[d] != w[d]) ? F.pn[d] : w[d];          // NaN deformation -> keep the point
                offsets[b * 3 + d] = wd - F.pn[d];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward kernel with the warp-code columns of the two input layers factored through the code ROW (round 4; the route of every
// forward whose codes are rows of a table -- training, sigma_fn, occupancy update, evaluation -- since round 5).
// Timing probes of round 4 (the probe kernels live in the history, commit 010eae6): building the layer input costs
// 27 % of deform_fwd_kernel, 16 % of it the per-lane loads of the sample's 128-float code row (33 uncoalesced 16-byte loads per
// lane and tile), and the two input GEMMs (W0, W4 over the 176-wide input: 2 x 44 MFMAs) are a third of the tile's 256 MFMAs
// although 128 of their 173 columns multiply a vector that only depends on the sample's code row.  So, as the backward does for
// the gradients: T_l[row][n] = sum_{k >= 48} W_l[n][k] * code16[row][k - 45] (l = 0, 4; fp16 operands, fp32 sums;
// deform_code_terms_kernel, n_rows x 2 x 128 numbers) is added to the bias the accumulators start from, and the GEMMs keep the
// first 3 K-steps (k < 48: the 45 positional-encoding columns + the first 3 code columns, so that the packed fragments stay as
// they are).  Per tile 192 instead of 256 MFMAs, 3 instead of 11 input fragments, 2 code-row loads per lane instead of 33.
// The terms live in LDS when the table has <= TERM_LDS_ROWS rows (a training batch: <= 24 images; an evaluation image: 1), and
// are read from global memory (L2-resident: 1 KB per row) otherwise (the occupancy update draws from all T timesteps) -- the same
// fp32 numbers added in the same order, so the two placements agree bit for bit, and a row's result does not depend on which
// table it sits in.  Same products as deform_fwd_kernel in another summation order: equal up to fp32 rounding of the
// pre-activations, not bit for bit.
// ---------------------------------------------------------------------------------------------------------
constexpr int TERM_KSTEPS = 3;                  // K-steps of the input stages that stay in the GEMM (k < 48)
constexpr int TERM_K0 = 16 * TERM_KSTEPS;       // first input column the terms cover
constexpr int TERM_ROW = 2 * DFW;               // floats per code row: T0 | T4
constexpr int TERM_STRIDE = TERM_ROW + 4;       // LDS row stride (lanes of one instruction read different rows at one offset)
constexpr int TERM_LDS_ROWS = 64;               // = NSX_MAX_SLOTS: every batch table fits

// terms[row] = T0 + b0 | T4 + b4: the value the accumulators of the two input layers start from (the same fp32 add the forward
// kernel made per tile in round 4, made once per row here)
__global__ __launch_bounds__(DFW) void deform_code_terms_kernel(const f16x8* __restrict__ frags, const float* __restrict__ bias,
                                                               const float* __restrict__ code, int64_t code_stride, int n_rows,
                                                               float* __restrict__ terms) {
    const int row = blockIdx.x >> 1, which = blockIdx.x & 1, n = threadIdx.x;
    if (row >= n_rows) return;
    const half_t* f16 = reinterpret_cast<const half_t*>(frags);
    const int group = which ? F4 : F0;                       // both groups: [4 M-tiles][11 K-steps], natural k order
    float acc = 0.f;
    for (int k = TERM_K0; k < DF_IN; ++k) {
        const int fi = group + (n >> 5) * DF_TIN + (k >> 4);
        const float w = (float)f16[((int64_t)fi * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7)];
        const float c = (float)(half_t)code[(int64_t)row * code_stride + (k - DF_PE)];
        acc = __fmaf_rn(w, c, acc);
    }
    terms[(int64_t)row * TERM_ROW + which * DFW + n] = bias[(which ? 4 : 0) * DFW + n] + acc;
}

template <bool LDS_TERMS>
struct DeformLdsT {
    f16x8 w[2][STAGE_FRAGS * 64];
    float bias[N_BIAS];
    float terms[LDS_TERMS ? TERM_LDS_ROWS * TERM_STRIDE : 4];
};
static_assert(sizeof(DeformLdsT<true>) <= 160 * 1024, "the forward's LDS block must fit the CU's 160 KB");

// the first TERM_KSTEPS input fragments of sample b (build_input's first loop); `crow`: the sample's code row.
// Element (t, j) of lane half kb is input column k = 16 t + 8 kb + j: the two candidates k0 = 16 t + j (kb = 0) and k0 + 8 are
// compile-time, so the lane selects the ARGUMENT (axis, frequency, phase, window weight) and evaluates ONE v_sin per element --
// build_input evaluates both candidates and selects the value: twice the quarter-rate transcendentals for the same numbers.
__device__ __forceinline__ void build_input_head(const DeformArgs& A, int64_t b, int kb, const float* crow, float pn[3],
                                                 f16x8 x[TERM_KSTEPS]) {
    // (scalars, not the caller's array: a select between two elements of an array in memory is folded into ONE load with a
    // per-lane index -- scratch memory)
    const float q0 = (A.pos[b * 3 + 0] - A.aabb_min[0]) / A.aabb_ext[0];
    const float q1 = (A.pos[b * 3 + 1] - A.aabb_min[1]) / A.aabb_ext[1];
    const float q2 = (A.pos[b * 3 + 2] - A.aabb_min[2]) / A.aabb_ext[2];
    pn[0] = q0; pn[1] = q1; pn[2] = q2;
    auto P = [&](int d) -> float { return d == 0 ? q0 : (d == 1 ? q1 : q2); };
    // the window weights as opaque scalars: a select between two loads of the by-value argument struct would be folded into ONE
    // load with a selected (per-lane) index, and a dynamically indexed kernel argument is copied to scratch memory
    float win[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) {
        float wf = A.window[f];
        asm volatile("" : "+s"(wf));
        win[f] = wf;
    }
#pragma unroll
    for (int t = 0; t < TERM_KSTEPS; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = 16 * t + j, k1 = k0 + 8;
            float v;
            if (k1 < 42) {                                   // both candidates are windowed sines
                const int q0 = k0 < 21 ? k0 : k0 - 21, q1 = k1 < 21 ? k1 : k1 - 21;
                const int d0 = q0 / 7, f0 = q0 - 7 * d0, d1 = q1 / 7, f1 = q1 - 7 * d1;
                const float p = kb ? P(d1) : P(d0);
                const float sc = kb ? (float)(1 << f1) : (float)(1 << f0);
                const float ph = kb ? (k1 >= 21 ? 0.25f : 0.f) : (k0 >= 21 ? 0.25f : 0.f);
                const float w = kb ? win[f1] : win[f0];
                v = w * __builtin_amdgcn_sinf(p * sc + ph);
            } else {                                         // k0 in [32, 40): sines, the scaled inputs, the first code columns
                auto value = [&](int k) -> float {
                    if (k < 42) {
                        const int q = k - 21, d = q / 7, f = q - 7 * d;
                        return win[f] * __builtin_amdgcn_sinf(P(d) * (float)(1 << f) + 0.25f);
                    }
                    if (k < DF_PE) return 6.283185307179586f * P(k - 42);
                    return crow[k - DF_PE];
                };
                const float v0 = value(k0), v1 = value(k1);
                v = kb ? v1 : v0;
            }
            x[t][j] = (half_t)v;
        }
    }
}

// accumulators start at the row's term (bias included; TP: an LDS or a global pointer)
template <typename TP>
__device__ __forceinline__ void acc_init_terms(f32x16 acc[4], TP term, int kb) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 tv = *reinterpret_cast<const f32x4*>(term + 32 * mt + 8 * q + 4 * kb);
            acc[mt][4 * q + 0] = tv.x; acc[mt][4 * q + 1] = tv.y; acc[mt][4 * q + 2] = tv.z; acc[mt][4 * q + 3] = tv.w;
        }
}

// the TERM_KSTEPS K-steps of an input stage that stay in the GEMM, copied as [4 M-tiles][TERM_KSTEPS] fragments at `local`
__device__ __forceinline__ void gemm_input_head(const f16x8* lds, int local, int lane, const f16x8* in, f32x16 acc[4]) {
    const f16x8* base = lds + (size_t)local * 64 + lane;
#pragma unroll
    for (int t = 0; t < TERM_KSTEPS; ++t) {
        f16x8 a[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = base[(mt * TERM_KSTEPS + t) * 64];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma(a[mt], in[t], acc[mt]);
    }
}

// `n_runs` runs of `run_len` consecutive fragments, `run_stride` fragments apart in the packed buffer, copied back to back
// to buf[dst ...] by LDS-DMA (one 1-KB fragment per wave instruction)
__device__ __forceinline__ void stage_issue_runs(const f16x8* __restrict__ frags, int first, int run_len, int run_stride,
                                                 int n_runs, f16x8* buf, int dst) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t voff = (threadIdx.x & 63u) * 16u;
    const char* base = reinterpret_cast<const char*>(frags);
    const int count = run_len * n_runs;
    for (int f = wave; f < count; f += NW) {
        const int src = first + (f / run_len) * run_stride + (f % run_len);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)src * 1024 + voff),
                                         (__attribute__((address_space(3))) void*)(buf + (size_t)(dst + f) * 64), 16, 0, 0);
    }
}

// Weight stages of the terms kernel (fragments per stage, barriers per tile: 5 instead of the general kernel's 8; 192 instead
// of 256 KB copied L2 -> LDS per 256 samples -- only the 12 fragments of W0 / W4-over-the-input that the GEMM still uses):
//   S0 = W0[:, k < 48] (12) | W1 (32)     S1 = W2 (32)     S2 = W3 (32)     S3 = W4[:, k < 48] (12) | W4 over x (32)
//   S4 = W5 (32) | heads (8)
constexpr int TS_HEAD = 4 * TERM_KSTEPS;        // 12

__device__ __forceinline__ void terms_issue_s0(const f16x8* frags, f16x8* buf) {
    stage_issue_runs(frags, F0, TERM_KSTEPS, DF_TIN, 4, buf, 0);
    stage_issue_runs(frags, F1, 32, 32, 1, buf, TS_HEAD);
}

// A.slot == nullptr: every sample takes row 0 (one code for the whole launch: an evaluation image's timestep)
template <bool LDS_TERMS>
__global__ __launch_bounds__(NW * 64, 1) void deform_fwd_terms_kernel(DeformArgs A, const float* __restrict__ terms, int n_rows,
                                                                    float* __restrict__ offsets, int64_t n_tiles,
                                                                    const int64_t* __restrict__ n_dev) {
    NSX_DEVICE_COUNT(A.S, n_tiles, 32, n_dev);
    __shared__ __attribute__((aligned(16))) DeformLdsT<LDS_TERMS> L;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kb = lane >> 5;
    const int64_t n_groups = (n_tiles + NW - 1) / NW;
    if constexpr (LDS_TERMS) {
        for (int i = threadIdx.x; i < n_rows * TERM_ROW; i += blockDim.x)
            L.terms[(i / TERM_ROW) * TERM_STRIDE + (i % TERM_ROW)] = terms[i];
    }
    for (int i = threadIdx.x; i < N_BIAS; i += blockDim.x) L.bias[i] = A.bias[i];
    terms_issue_s0(A.frags, L.w[0]);
    __syncthreads();
    int cur = 0;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {        // all waves iterate together
        const int64_t tile = grp * NW + wave;
        const int64_t b_raw = tile * 32 + (lane & 31);
        const int64_t b = b_raw < A.S ? b_raw : A.S - 1;
        lds_cfloat* bias = launder_lds(L.bias);
        int row = A.slot ? A.slot[b] : 0;
        row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
        const float* crow = A.code + (int64_t)row * A.code_stride;
        const float* term;
        if constexpr (LDS_TERMS) term = L.terms + row * TERM_STRIDE;
        else term = terms + (int64_t)row * TERM_ROW;
        f16x8 x[TERM_KSTEPS], h[DF_TW];
        f32x16 acc[4];
        float pn[3];
        // S0: L0 (input head, accumulators start at T0 + b0) and L1
        stage_issue(A.frags, F2, 32, L.w[cur ^ 1]);
        build_input_head(A, b, kb, crow, pn, x);
        acc_init_terms(acc, term, kb);
        gemm_input_head(L.w[cur], 0, lane, x, acc);
        finish_layer<false>(acc, h);
        acc_init(acc, bias + 1 * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], TS_HEAD, lane, h, acc);
        finish_layer<false>(acc, h);
        stage_flip(cur);
        // S1: L2
        stage_issue(A.frags, F3, 32, L.w[cur ^ 1]);
        acc_init(acc, bias + 2 * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, h, acc);
        finish_layer<false>(acc, h);
        stage_flip(cur);
        // S2: L3
        stage_issue_runs(A.frags, F4, TERM_KSTEPS, DF_TIN, 4, L.w[cur ^ 1], 0);
        stage_issue_runs(A.frags, F4X, 32, 32, 1, L.w[cur ^ 1], TS_HEAD);
        acc_init(acc, bias + 3 * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, h, acc);
        finish_layer<false>(acc, h);
        stage_flip(cur);
        // S3: L4 = cat[input, x]: the input head from T4 + b4, then the 8 K-steps over x
        stage_issue(A.frags, F5, 40, L.w[cur ^ 1]);
        acc_init_terms(acc, term + DFW, kb);
        gemm_input_head(L.w[cur], 0, lane, x, acc);
        gemm_layer_lds<DF_TW>(L.w[cur], TS_HEAD, lane, h, acc);
        finish_layer<false>(acc, h);
        stage_flip(cur);
        // S4: L5 + heads
        terms_issue_s0(A.frags, L.w[cur ^ 1]);                              // the next tile's first stage
        acc_init(acc, bias + 5 * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, h, acc);
        finish_layer<false>(acc, h);
        f32x16 o = zero16();
#pragma unroll
        for (int t = 0; t < DF_TW; ++t) o = mfma(L.w[cur][(32 + t) * 64 + lane], h[t], o);
        float own[4], oth[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) own[r] = (float)(half_t)(o[r] + bias[6 * DFW + acc_row(r, kb)]);
#pragma unroll
        for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(own[r], 32);
        float rr[3], vv[3];
        rr[0] = kb ? oth[0] : own[0]; rr[1] = kb ? oth[1] : own[1]; rr[2] = kb ? oth[2] : own[2];
        vv[0] = kb ? oth[3] : own[3]; vv[1] = kb ? own[0] : oth[0]; vv[2] = kb ? own[1] : oth[1];
        float w[3];
        se3_apply(rr, vv, pn, w);
        if (b_raw < A.S && kb == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float wd = (w[d] != w[d]) ? pn[d] : w[d];            // NaN deformation -> keep the point
                offsets[b * 3 + d] = wd - pn[d];
            }
        }
        stage_flip(cur);
    }
}

// The terms forward of ONE tile inside the backward chain kernel (SLOTS: codes are rows of the batch's table): the same
// stages, accumulator starts and MFMA order as deform_fwd_terms_kernel -- the recomputed pre-activations are the forward
// kernel's bit for bit, so the ReLU masks are those of the activations the step actually used -- plus what the backward keeps:
// the masks and the transposed layer-input tiles (a0: the 3 input fragments, 48 rows; rows 48..63 of its 64-row tile are
// zeros -- the weight-gradient kernel reads the positional-encoding columns k < 45 of it only).  On entry stage S0 is resident
// in L.w[cur]; on exit the chain's first stage (heads^T | W5^T, 36 fragments).
__device__ __forceinline__ void forward_tile_terms_bwd(const DeformArgs& A, int64_t b, int lane, Fwd& F, half_t* a_tiles,
                                                       DeformLds& L, int& cur, const float* term, const float* crow) {
    constexpr int A0_HALFS = 2 * 32 * 32;                       // the 64-row a0 tile of Lay<true>
    const int kb = lane >> 5;
    lds_cfloat* bias = launder_lds(L.bias);
    const TSel tsel = make_tsel(lane);
    f32x16 acc[4];
    // stage "W0 head" (12 fragments): L0, accumulators start at T0 + b0
    stage_issue(A.frags, F1, 32, L.w[cur ^ 1]);
    build_input_head(A, b, kb, crow, F.pn, F.x);
    store_tile_T<TERM_KSTEPS>(a_tiles, lane, F.x, tsel);
    acc_init_terms(acc, term, kb);
    gemm_input_head(L.w[cur], 0, lane, F.x, acc);
    F.m1 = finish_layer<true>(acc, F.h);
    store_tile_T<DF_TW>(a_tiles + A0_HALFS + 0 * DFW * 32, lane, F.h, tsel);
    stage_flip(cur);
    // L1 .. L3 (one loop body, as in forward_tile: the chain kernel sits at the register limit, and three unrolled layers
    // spilled 95 VGPRs instead of 20)
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        if (l < 3) {
            stage_issue(A.frags, l == 1 ? F2 : F3, 32, L.w[cur ^ 1]);
        } else {
            stage_issue_runs(A.frags, F4, TERM_KSTEPS, DF_TIN, 4, L.w[cur ^ 1], 0);
            stage_issue_runs(A.frags, F4X, 32, 32, 1, L.w[cur ^ 1], TS_HEAD);
        }
        acc_init(acc, bias + l * DFW, kb);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, F.h, acc);
        const u32x2 m = finish_layer<true>(acc, F.h);
        if (l == 1) F.m2 = m; else if (l == 2) F.m3 = m; else F.m4 = m;
        store_tile_T<DF_TW>(a_tiles + A0_HALFS + l * DFW * 32, lane, F.h, tsel);
        stage_flip(cur);
    }
    // stage "W4 head | W4 over x" (44): L4 = cat[input, x], accumulators start at T4 + b4
    stage_issue(A.frags, F5, 40, L.w[cur ^ 1]);
    acc_init_terms(acc, term + DFW, kb);
    gemm_input_head(L.w[cur], 0, lane, F.x, acc);
    gemm_layer_lds<DF_TW>(L.w[cur], TS_HEAD, lane, F.h, acc);
    F.m5 = finish_layer<true>(acc, F.h);
    store_tile_T<DF_TW>(a_tiles + A0_HALFS + 4 * DFW * 32, lane, F.h, tsel);
    stage_flip(cur);
    // L5 (+ out_activation ReLU) and the heads; the chain's first stage arrives meanwhile
    stage_issue(A.frags, BH, 36, L.w[cur ^ 1]);
    acc_init(acc, bias + 5 * DFW, kb);
    gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, F.h, acc);
    F.m6 = finish_layer<true>(acc, F.h);
    f32x16 o = zero16();
#pragma unroll
    for (int t = 0; t < DF_TW; ++t) o = mfma(L.w[cur][(32 + t) * 64 + lane], F.h[t], o);
    float own[4], oth[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) own[r] = (float)(half_t)(o[r] + bias[6 * DFW + acc_row(r, kb)]);
#pragma unroll
    for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(own[r], 32);
    F.r[0] = kb ? oth[0] : own[0]; F.r[1] = kb ? oth[1] : own[1]; F.r[2] = kb ? oth[2] : own[2];
    F.v[0] = kb ? oth[3] : own[3]; F.v[1] = kb ? own[0] : oth[0]; F.v[2] = kb ? own[1] : oth[1];
    stage_flip(cur);
}

// first stage of a tile of the chain kernel's terms forward: the 12 fragments of W0 over the input head
__device__ __forceinline__ void terms_bwd_issue_first(const f16x8* frags, f16x8* buf) {
    stage_issue_runs(frags, F0, TERM_KSTEPS, DF_TIN, 4, buf, 0);
}

constexpr int BWD_TERM_LDS_ROWS = 24;           // code rows whose terms fit beside the chain kernel's 120 KB of LDS

// ---------------------------------------------------------------------------------------------------------
// backward chain kernel: writes per-tile a0..a6, dZ0..dZ5, dZheads, dCode tiles
// ---------------------------------------------------------------------------------------------------------
// per sample-tile scratch layout (halfs), the input tile first:
//   SLOTS = false: a0 [192][32] | a1..a6 [6][128][32] | dZ0..dZ5 [6][128][32] | dZh [32][32] | dC [128][32]     118 KB
//   SLOTS = true : a0 [ 64][32] | a1..a5 [5][128][32] | dZ0..dZ5                                                  92 KB
// SLOTS: every sample's warp code is a row of a small table (the <= 24 time codes of a batch), so everything that touches
// the code columns factors through the slot, exactly as the hash tables' gradient does:
//   dW0[:, code] = sum_s dZ0[:, s] code[slot_s]^T = R0 code,   R0[n][r] = sum_{s: slot_s = r} dZ0[n][s]   (dW4 likewise)
//   dL/dcode[r]  = W0[:, code]^T R0[:, r] + W4[:, code]^T R4[:, r]
// The chain kernel then neither forms the per-sample code gradient (two of its 14 weight stages) nor writes it and the
// code rows of a0 (16 of 118 KB per tile); the weight-gradient kernel accumulates R0 / R4 with a one-hot operand next to
// the positional-encoding columns, and deform_code_expand_kernel finishes the three small products.
// SLOTS (round 4) also forms the HEAD gradients (dWr, dWv, dbr, dbv = dZh a6^T and the row sums of dZh: 774 numbers) inside
// the chain kernel, where a6 and dZh of the tile are in registers: 18 matrix instructions per tile (the transposes of the
// two operands + the product; the chain has ~750) into transient accumulators whose 6 useful rows are added to a 3-KB LDS
// array of the block, written out once per block and summed by deform_code_expand_kernel.  Neither the a6 tile (8 KB) nor
// dZh (2 KB) crosses HBM any more: 102 -> 92 KB written per tile, 111 -> 101 KB read by the weight-gradient kernel.
template <bool SLOTS>
struct Lay {
    static constexpr int A0_FRAGS = SLOTS ? 4 : DF_TIN;
    static constexpr int N_A = SLOTS ? 5 : 6;                                      // transposed activation tiles a1..
    static constexpr int64_t TILE_A = (SLOTS ? 64 : 192) * 32;
    static constexpr int64_t TILE_DZ = TILE_A + N_A * DFW * 32;
    static constexpr int64_t TILE_DZH = TILE_DZ + 6 * DFW * 32;
    static constexpr int64_t TILE_DC = TILE_DZH + (SLOTS ? 0 : 32 * 32);
    static constexpr int64_t TILE_HALFS = SLOTS ? TILE_DC : TILE_DC + DFW * 32;    // 47 104 / 60 416 halfs per 32 samples
    static constexpr int TILE_KB = (int)(TILE_HALFS * 2 / 1024);                   // 92 / 118
    static constexpr int KB_A0 = 0, KB_A = (int)(TILE_A * 2 / 1024), KB_DZ = (int)(TILE_DZ * 2 / 1024);
    static constexpr int KB_DZH = (int)(TILE_DZH * 2 / 1024), KB_DC = (int)(TILE_DC * 2 / 1024);
    static_assert(TILE_HALFS * 2 % 1024 == 0 && TILE_A * 2 % 1024 == 0 && TILE_DZ * 2 % 1024 == 0 &&
                  TILE_DZH * 2 % 1024 == 0 && TILE_DC * 2 % 1024 == 0, "scratch regions must be KB-aligned");
};
// R0 | R4: [2][128 neurons][128 code rows] fp32 at the head of the scratch buffer (SLOTS; zeroed by the chain kernel)
constexpr int64_t SLOT_SUMS_FLOATS = 2 * DFW * 128;
constexpr int64_t SLOT_SUMS_BYTES = SLOT_SUMS_FLOATS * 4;
// head gradients of one chain block, in parameter order: Wr [3][128] | br [3] | Wv [3][128] | bv [3] (= P_WR .. P_TOTAL)
constexpr int HEAD_PARAMS = P_TOTAL - P_WR;
constexpr int HEAD_STRIDE = (HEAD_PARAMS + 63) / 64 * 64;
static_assert(HEAD_PARAMS == 2 * (3 * DFW + 3) && P_BR == P_WR + 3 * DFW && P_WV == P_BR + 3 && P_BV == P_WV + 3 * DFW,
              "the head parameters are one contiguous run");

// dZ = dA * relu'(a): accumulators -> packed halfs, AND-ed with 0xFFFF per positive unit (mask layout: finish_layer)
__device__ __forceinline__ void mask_pack(const f32x16 d[4], u32x2 mask, f16x8 dz[DF_TW]) {
#pragma unroll
    for (int t = 0; t < DF_TW; ++t) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int mt = t >> 1, r = 8 * (t & 1) + 2 * q, D = 4 * t + q;
            const uint32_t p = cvt_pk(d[mt][r], d[mt][r + 1]);
            const uint32_t bits = (mask[D >> 4] >> (D & 15)) & 0x10001u;
            v[q] = p & (bits * 0xFFFFu);
        }
        dz[t] = __builtin_bit_cast(f16x8, v);
    }
}

// TERMS (SLOTS only; round 5): the forward recompute is the terms forward (forward_tile_terms_bwd) -- 1: the rows' terms in LDS
// (<= BWD_TERM_LDS_ROWS rows: a training batch's images), 2: read from global memory (L2).
template <bool SLOTS, int TERMS = 0>
__global__ __launch_bounds__(NW * 64, 1) void deform_bwd_kernel(DeformArgs A, const float* __restrict__ goff,
                                                              half_t* __restrict__ scratch, int64_t n_tiles,
                                                              float* __restrict__ gcode_samples,
                                                              const int64_t* __restrict__ n_dev,
                                                              float* __restrict__ slot_sums,
                                                              float* __restrict__ head_partials,
                                                              const float* __restrict__ terms = nullptr, int n_rows = 0) {
    static_assert(TERMS == 0 || SLOTS, "the terms forward needs codes that are rows of a table");
    using LY = Lay<SLOTS>;
    constexpr int64_t TILE_HALFS = LY::TILE_HALFS, TILE_DZ = LY::TILE_DZ, TILE_DZH = LY::TILE_DZH, TILE_DC = LY::TILE_DC;
    if (SLOTS) {      // the per-slot sums the NEXT kernel adds to: cleared here (also when no sample is left to process)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < SLOT_SUMS_FLOATS;
             i += (int64_t)gridDim.x * blockDim.x) slot_sums[i] = 0.f;
    }
    NSX_DEVICE_COUNT(A.S, n_tiles, 32, n_dev);
    __shared__ __attribute__((aligned(16))) DeformLds L;
    // one table PER WAVE: within one instruction every lane then owns its address, so the sums are plain LDS
    // read-modify-writes (ds_add_f32 costs ~150 cycles per wave instruction; 30 per tile were + 17 % on this kernel)
    __shared__ float head_sums[SLOTS ? NW * HEAD_STRIDE : 1];
    if (SLOTS) {
        for (int i = threadIdx.x; i < NW * HEAD_STRIDE; i += blockDim.x) head_sums[i] = 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 31, kb = lane >> 5;
    const int64_t n_groups = (n_tiles + NW - 1) / NW;
    __shared__ float terms_lds[TERMS == 1 ? BWD_TERM_LDS_ROWS * TERM_STRIDE : 4];
    if constexpr (TERMS == 0) {
        lds_prologue(A, L, F0, 44);
    } else {
        if constexpr (TERMS == 1) {
            for (int i = threadIdx.x; i < n_rows * TERM_ROW; i += blockDim.x)
                terms_lds[(i / TERM_ROW) * TERM_STRIDE + (i % TERM_ROW)] = terms[i];
        }
        for (int i = threadIdx.x; i < N_BIAS; i += blockDim.x) L.bias[i] = A.bias[i];
        terms_bwd_issue_first(A.frags, L.w[0]);
        __syncthreads();
    }
    int cur = 0;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {        // all waves iterate together
        const int64_t tile_raw = grp * NW + wave;
        const bool tile_ok = tile_raw < n_tiles;
        const int64_t tile = tile_ok ? tile_raw : n_tiles - 1;
        const int64_t b_raw = tile * 32 + n;
        const bool valid = tile_ok && b_raw < A.S;
        const int64_t b = (b_raw < A.S) ? b_raw : A.S - 1;
        // waves past the last tile still walk the layers (barriers) but write into the block-private dummy tile
        half_t* T = scratch + (tile_ok ? tile : n_tiles + (int64_t)blockIdx.x * NW + wave) * TILE_HALFS;
        Fwd F;
        if constexpr (TERMS == 0) {
            forward_tile<true, LY::A0_FRAGS, !SLOTS>(A, b, lane, F, T, L, cur, BH, 36);
        } else {
            int row = A.slot[b];
            row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
            const float* term;
            if constexpr (TERMS == 1) term = terms_lds + row * TERM_STRIDE;
            else term = terms + (int64_t)row * TERM_ROW;
            forward_tile_terms_bwd(A, b, lane, F, T, L, cur, term, A.code + (int64_t)row * A.code_stride);
        }
        // ---- SE(3) backward (fp32): g = dL/dwarped ----
        float g[3] = {0.f, 0.f, 0.f};
        if (valid) { g[0] = goff[b * 3]; g[1] = goff[b * 3 + 1]; g[2] = goff[b * 3 + 2]; }
        float wchk[3];
        se3_apply(F.r, F.v, F.pn, wchk);
#pragma unroll
        for (int d = 0; d < 3; ++d) if (wchk[d] != wchk[d]) g[d] = 0.f;       // NaN fallback branch has no gradient
        const Se3Coef q = se3_coef(F.r);
        const float* r = F.r; const float* v = F.v; const float* p = F.pn;
        float u1[3], u2[3], w1[3], w2[3], t1[3], t2[3];
        cross3(r, p, u1); cross3(r, u1, u2); cross3(r, v, w1); cross3(r, w1, w2);
        float dv[3], dr[3];
        // dL/dv = V^T g = g - b (r x g) + c (r x (r x g))
        cross3(r, g, t1); cross3(r, t1, t2);
#pragma unroll
        for (int d = 0; d < 3; ++d) dv[d] = g[d] - q.b * t1[d] + q.c * t2[d];
        const float th = q.theta, s = sinf(th), co = cosf(th);
        const float dLa = dot3(g, u1), dLb = dot3(g, u2) + dot3(g, w1), dLc = dot3(g, w2);
        const float da = (th * co - s) / (th * th);
        const float db = (th * s - 2.0f * (1.0f - co)) / (th * th * th);
        const float dc = (3.0f * s - 2.0f * th - th * co) / (th * th * th * th);
        const float dth = q.clamped ? 0.f : (dLa * da + dLb * db + dLc * dc) / th;
        float pxg[3], vxg[3];
        cross3(p, g, pxg); cross3(v, g, vxg);
        const float rp = dot3(r, p), gr = dot3(g, r), gp = dot3(g, p), rv = dot3(r, v), gv = dot3(g, v);
#pragma unroll
        for (int d = 0; d < 3; ++d)
            dr[d] = dth * r[d] + q.a * pxg[d] + q.b * (g[d] * rp + p[d] * gr - 2.0f * r[d] * gp) + q.b * vxg[d]
                    + q.c * (g[d] * rv + v[d] * gr - 2.0f * r[d] * gv);
        // head gradient fragment (K-step 0, chained rows): kb=0 -> rows 0..3 = dr0,dr1,dr2,dv0 ; kb=1 -> rows 4..7 = dv1,dv2,0,0
        f16x8 dzh;
#pragma unroll
        for (int j = 0; j < 8; ++j) dzh[j] = (half_t)0.f;
        if (kb == 0) { dzh[0] = (half_t)dr[0]; dzh[1] = (half_t)dr[1]; dzh[2] = (half_t)dr[2]; dzh[3] = (half_t)dv[0]; }
        else         { dzh[0] = (half_t)dv[1]; dzh[1] = (half_t)dv[2]; }
        const TSel tsel = make_tsel(lane);
        if constexpr (!SLOTS) store_tile_T<1>(T + TILE_DZH, lane, &dzh, tsel);   // head rows (only columns < 16 are populated)
        else {
            // head gradients of this tile: dWh[rho][k] += sum_s dZh[rho][s] a6[k][s], dbh[rho] += sum_s dZh[rho][s].
            // X = transposed dZh (lane = head column, 8 samples per K-step), Y = transposed a6 tiles (lane = neuron column):
            // accumulator element r of lane (i, half) is head column acc_row(r, half) x neuron tile_neuron_chain(p, i);
            // head column c holds head row tile_neuron_chain(0, c), so rows 0..5 are elements r = 0..5 of the half-0 lanes.
            f16x8 dzv = dzh;
            if (!valid) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dzv[j] = (half_t)0.f;
            }
            u32x4 x0, x1;
            transpose_pair(dzv, nullptr, tsel, x0, x1);
            const f16x8 X0 = __builtin_bit_cast(f16x8, x0), X1 = __builtin_bit_cast(f16x8, x1);
            float* hs = head_sums + wave * HEAD_STRIDE;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                u32x4 y0, y1;
                transpose_pair(F.h[2 * p], &F.h[2 * p + 1], tsel, y0, y1);
                f32x16 hw = mfma(X0, __builtin_bit_cast(f16x8, y0), zero16());
                hw = mfma(X1, __builtin_bit_cast(f16x8, y1), hw);
                if (kb == 0) {
                    float* dst = hs + tile_neuron_chain(p, n);      // (distinct per lane)
                    float old[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) old[r] = dst[r < 3 ? r * DFW : 3 * DFW + 3 + (r - 3) * DFW];
#pragma unroll
                    for (int r = 0; r < 6; ++r) dst[r < 3 ? r * DFW : 3 * DFW + 3 + (r - 3) * DFW] = old[r] + hw[r];
                }
            }
            f16x8 ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;
            f32x16 hb = mfma(X0, ones, zero16());
            hb = mfma(X1, ones, hb);
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 6; ++r) hs[r < 3 ? 3 * DFW + r : 2 * (3 * DFW) + 3 + (r - 3)] += hb[r];
            }
        }
        // ---- chain (stage BH | B5 is resident in L.w[cur]) ----
        f32x16 d[4];
        f16x8 dz[DF_TW];
        // dA6 = heads^T dzh ; dZ5 = dA6 * relu'(a6)
        stage_issue(A.frags, B4X, 32, L.w[cur ^ 1]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) d[mt] = mfma(L.w[cur][mt * 64 + lane], dzh, zero16());
        mask_pack(d, F.m6, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 5 * DFW * 32, lane, dz, tsel);
        // dA5 = W5^T dZ5 ; dZ4
        for (int i = 0; i < 4; ++i) d[i] = zero16();
        gemm_layer_lds<DF_TW>(L.w[cur], 4, lane, dz, d);
        mask_pack(d, F.m5, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 4 * DFW * 32, lane, dz, tsel);
        // keep dZ4 for the code gradient (dC = W4[:, code]^T dZ4 + W0[:, code]^T dZ0), formed at the end
        f16x8 dz4[DF_TW];
        if constexpr (!SLOTS) {
#pragma unroll
            for (int t = 0; t < DF_TW; ++t) dz4[t] = dz[t];
        }
        stage_flip(cur);
        // dA4 = W4[:, x]^T dZ4 ; dZ3
        stage_issue(A.frags, B3, 32, L.w[cur ^ 1]);
        for (int i = 0; i < 4; ++i) d[i] = zero16();
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz, d);
        mask_pack(d, F.m4, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 3 * DFW * 32, lane, dz, tsel);
        stage_flip(cur);
        // L3 -> L2 -> L1
        stage_issue(A.frags, B2, 32, L.w[cur ^ 1]);
        for (int i = 0; i < 4; ++i) d[i] = zero16();
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz, d);
        mask_pack(d, F.m3, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 2 * DFW * 32, lane, dz, tsel);
        stage_flip(cur);
        stage_issue(A.frags, B1, 32, L.w[cur ^ 1]);
        for (int i = 0; i < 4; ++i) d[i] = zero16();
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz, d);
        mask_pack(d, F.m2, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 1 * DFW * 32, lane, dz, tsel);
        stage_flip(cur);
        if constexpr (TERMS != 0) terms_bwd_issue_first(A.frags, L.w[cur ^ 1]);       // first stage of the next tile
        else if constexpr (SLOTS) stage_issue(A.frags, F0, 44, L.w[cur ^ 1]);
        else stage_issue(A.frags, B0C, 32, L.w[cur ^ 1]);
        for (int i = 0; i < 4; ++i) d[i] = zero16();
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz, d);
        mask_pack(d, F.m1, dz);
        store_tile_T<DF_TW>(T + TILE_DZ + 0 * DFW * 32, lane, dz, tsel);
        stage_flip(cur);
        if constexpr (SLOTS) continue;          // the code gradient is formed from the per-slot sums of dZ0 / dZ4
        f32x16 dcode[4];
        for (int i = 0; i < 4; ++i) dcode[i] = zero16();
        stage_issue(A.frags, B4C, 32, L.w[cur ^ 1]);
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz, dcode);
        stage_flip(cur);
        stage_issue(A.frags, F0, 44, L.w[cur ^ 1]);           // first stage of the next tile
        gemm_layer_lds<DF_TW>(L.w[cur], 0, lane, dz4, dcode);
        {
            // code-gradient tile: accumulators -> chained fragments (code index kchain(t, kb, j)) -> transposed tile
            f16x8 dcf[DF_TW];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr)
                    dcf[2 * mt + (rr >> 3)][rr & 7] = valid ? (half_t)dcode[mt][rr] : (half_t)0.f;
            store_tile_T<DF_TW>(T + TILE_DC, lane, dcf, tsel);
            if (gcode_samples && valid) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr)
                        gcode_samples[b * DF_CODE + 32 * mt + acc_row(rr, kb)] = dcode[mt][rr];
            }
        }
        stage_flip(cur);
    }
    if constexpr (SLOTS) {          // every launched block leaves its sums (zeros if it had no tile): the reducer adds gridDim.x rows
        __syncthreads();
        for (int i = threadIdx.x; i < HEAD_STRIDE; i += blockDim.x) {
            float v = head_sums[i];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += head_sums[w * HEAD_STRIDE + i];
            head_partials[(int64_t)blockIdx.x * HEAD_STRIDE + i] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// sample-contracted GEMMs: C[M][N] += sum over samples X[M][s] * Y[N][s]  (fp16 tiles, fp32 accumulation)
// ---------------------------------------------------------------------------------------------------------
// Every weight / bias / code-table gradient of the step in ONE launch that reads each scratch byte (almost) once.
// grid = (block type, sample chunk).  A block type owns a group of gradient GEMMs whose operands it copies, one
// 32-sample tile per pipeline stage, from the scratch into a 4-deep LDS ring with the LDS-DMA path (3 stages in
// flight: the kernel is HBM-bound, ~30 KB per stage and CU); its 8 waves then each own ONE 32-row tile of X and all
// (<= 6) column tiles of Y, so a full 128 x 128..192 gradient lives in the block's accumulators for the whole chunk
// and is added to the gradient buffer with fp32 atomics at the end.
//   type 0: dW0 = dZ0 a0^T, db0 | dW4[:, :173] = dZ4 a0^T, db4       stage = dZ0 dZ4 a0            (28 KB)
//   type 1: dW1 = dZ1 a1^T, db1 | dW2 = dZ2 a2^T, db2                stage = dZ1 a1 dZ2 a2         (32 KB)
//   type 2: dW3 = dZ3 a3^T, db3 | dW4[:, 173:] = dZ4 a4^T            stage = dZ3 a3 dZ4 a4         (32 KB)
//   type 3: dW5 = dZ5 a5^T, db5 | heads: dZh a6^T, dbr, dbv          stage = dZ5 a5 dZh a6         (26 KB; SLOTS: dZ5 a5,
//           16 KB -- the chain kernel forms the head gradients)
//   type 4: code table: onehot(slot) dC^T                            stage = dC + the 32 slots     ( 9 KB)
// SLOTS (Lay<true>): type 0 reads dZ0, dZ4, the 64-row a0 tile and the tile's 32 slots (21 KB) and accumulates, next to the
// positional-encoding columns of dW0 / dW4, the per-slot sums R0 / R4 = dZ onehot(slot)^T; type 4 does not exist.
// Bias gradients are the row sums of dZ: the same X fragments against an all-ones operand.
constexpr int WG_K = 4;                         // LDS-DMA copies per wave and stage
constexpr int WG_PIECES = NW * WG_K;            // 32 one-KB pieces per stage
constexpr int WG_NS = 4;                        // ring depth
constexpr int WG_TYPES = 5;
constexpr int WG_SLOT_KB = 20;                  // SLOTS, type 0: where the tile's slots land inside the stage

// piece q of a stage of block type `type`: source KB offset inside the scratch tile (-1: the slot piece); pieces past
// the end repeat earlier ones (same bytes to the same place) so that every wave issues exactly WG_K copies per stage
template <bool SLOTS>
__device__ __forceinline__ int wg_piece_src(int type, int q, int& dst_kb, bool with_slots) {
    using LY = Lay<SLOTS>;
    constexpr int KB_A0 = LY::KB_A0, KB_A = LY::KB_A, KB_DZ = LY::KB_DZ, KB_DZH = LY::KB_DZH, KB_DC = LY::KB_DC;
    auto a_kb = [](int l) { return l == 0 ? KB_A0 : KB_A + 8 * (l - 1); };   // input of layer l (l = 6: input of the heads)
    auto dz_kb = [](int l) { return KB_DZ + 8 * l; };
    int total, src = 0;
    switch (type) {
        case 0: total = SLOTS ? 21 : 28; break;
        case 1: case 2: total = 32; break;
        case 3: total = SLOTS ? 16 : 26; break;
        default: total = with_slots ? 9 : 8; break;
    }
    q = q % total;
    dst_kb = q;
    if (SLOTS && type == 0) {
        if (q < 8) return dz_kb(0) + q;
        if (q < 16) return dz_kb(4) + q - 8;
        if (q < 20) return a_kb(0) + q - 16;
        return -1;                                                          // q == 20 == WG_SLOT_KB: the slots
    }
    switch (type) {
        case 0: src = q < 8 ? dz_kb(0) + q : (q < 16 ? dz_kb(4) + q - 8 : a_kb(0) + q - 16); break;
        case 1: src = q < 8 ? dz_kb(1) + q : (q < 16 ? a_kb(1) + q - 8 : (q < 24 ? dz_kb(2) + q - 16 : a_kb(2) + q - 24)); break;
        case 2: src = q < 8 ? dz_kb(3) + q : (q < 16 ? a_kb(3) + q - 8 : (q < 24 ? dz_kb(4) + q - 16 : a_kb(4) + q - 24)); break;
        case 3: src = q < 8 ? dz_kb(5) + q : (q < 16 ? a_kb(5) + q - 8 : (q < 18 ? KB_DZH + q - 16 : a_kb(6) + q - 18)); break;
        default: src = q < 8 ? KB_DC + q : -1; break;
    }
    return src;
}

struct WgRole {            // what one wave of a block accumulates
    int x_kb;              // LDS KB offset of its X tile (2 KB); -1: one-hot of the code slot
    int y_kb, n_y;         // first Y tile, number of Y tiles (0: the wave only helps copying)
    int bias;              // also accumulate X * ones
    int x_tile;            // index of the X tile inside its activation (row block)
    int job;               // output mapping, see wg_store
};

template <bool SLOTS>
__device__ __forceinline__ WgRole wg_role(int type, int wave, int n_code_rows) {
    WgRole r{0, 0, 0, 0, wave & 3, -1};
    const int hi = wave >> 2;
    switch (type) {
        case 0: r.x_kb = 8 * hi + 2 * (wave & 3); r.y_kb = 16; r.n_y = SLOTS ? 2 : 6; r.bias = 1; r.job = hi ? 4 : 0; break;
        case 1: r.x_kb = 16 * hi + 2 * (wave & 3); r.y_kb = 8 + 16 * hi; r.n_y = 4; r.bias = 1; r.job = hi ? 2 : 1; break;
        case 2: r.x_kb = 16 * hi + 2 * (wave & 3); r.y_kb = 8 + 16 * hi; r.n_y = 4; r.bias = hi ? 0 : 1; r.job = hi ? 5 : 3; break;
        case 3:
            if (!hi) { r.x_kb = 2 * wave; r.y_kb = 8; r.n_y = 4; r.bias = 1; r.job = 6; }
            else if (wave == 4 && !SLOTS) { r.x_kb = 16; r.y_kb = 18; r.n_y = 4; r.bias = 1; r.x_tile = 0; r.job = 7; }
            break;
        default:
            if (!hi && 32 * wave < n_code_rows) { r.x_kb = -1; r.y_kb = 0; r.n_y = 4; r.job = 8; }
            break;
    }
    return r;
}

// accumulator element r of lane (i, kb) of output tile (x tile mt, y tile nt): X-side row acc_row(r, kb), Y-side column i
// `part` (SLOTS): this chunk's private copy of the parameter-gradient vector -- every element has exactly one owner
// (block type, wave, tile, register, lane), so the block leaves its sums with plain stores and deform_finish_kernel adds the
// chunks; without it the sums go to the gradient buffer with fp32 atomics (one burst of ~1 M sector atomics when all
// blocks finish together: 60 us of a 100 us launch in steady state).
template <bool SLOTS>
__device__ __forceinline__ void wg_store(const f32x16& acc, int job, int mt, int nt, bool is_bias, int lane,
                                         float* __restrict__ gp, float* __restrict__ gcode, int n_code_rows,
                                         float* __restrict__ part = nullptr) {
    auto put = [&](int64_t idx, float v) {
        if (part) part[idx] = v;
        else if (v != 0.f) atomicAdd(&gp[idx], v);
    };
    const int i = lane & 31, kb = lane >> 5;
    if (is_bias && i != 0) return;                          // every column of X * ones holds the row sum
    const int64_t w_off[7] = {P_W0, P_W1, P_W2, P_W3, P_W4, P_W4 + DF_IN, P_W5};
    const int64_t b_off[7] = {P_B0, P_B1, P_B2, P_B3, P_B4, -1, P_B5};
    const int ldc[7] = {DF_IN, DFW, DFW, DFW, DF_W4, DF_W4, DFW};
    const bool y_natural = (job == 0 || job == 4);          // a0 tiles are in natural column order
    // (SLOTS: the a0 tile's columns beyond the positional encoding are code columns -- those come from the slot sums)
    const int n_cols = y_natural ? (SLOTS ? DF_PE : DF_IN) : DFW;
    const int col = y_natural ? 32 * nt + i : tile_neuron_chain(nt, i);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        const int ar = acc_row(r, kb);
        if (job <= 6) {
            const int row = tile_neuron_chain(mt, ar);
            if (is_bias) put(b_off[job] + row, v);
            else if (col < n_cols) put(w_off[job] + (int64_t)row * ldc[job] + col, v);
        } else if (job == 7) {                               // heads: rows 0..2 -> Wr / br, 3..5 -> Wv / bv
            if (ar >= 16) continue;                          // only the first 16 columns of the dZh tile are populated
            const int row = tile_neuron_chain(0, ar);
            if (row >= 6) continue;
            if (is_bias) put(row < 3 ? P_BR + row : P_BV + row - 3, v);
            else put((row < 3 ? (int64_t)P_WR + (int64_t)row * DFW : (int64_t)P_WV + (int64_t)(row - 3) * DFW) + col, v);
        } else {                                             // code table rows (natural), code columns (chained)
            const int row = 32 * mt + ar;
            if (v != 0.f && row < n_code_rows) atomicAdd(&gcode[(int64_t)row * DF_CODE + col], v);
        }
    }
}

// accumulator of (X tile mt of dZ0 | dZ4) x (one-hot tile nt): R[which][neuron][code row] += sum over the tile's samples
__device__ __forceinline__ void wg_store_slot_sums(const f32x16& acc, int which, int mt, int nt, int lane,
                                                   float* __restrict__ slot_sums, int n_code_rows) {
    const int i = lane & 31, kb = lane >> 5;
    const int row = 32 * nt + i;                            // code row (Y side, natural order)
    if (row >= n_code_rows) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r];
        if (v == 0.f) continue;
        const int neuron = tile_neuron_chain(mt, acc_row(r, kb));
        atomicAdd(&slot_sums[((int64_t)which * DFW + neuron) * 128 + row], v);
    }
}

template <bool SLOTS>
__global__ __launch_bounds__(NW * 64, 1) void deform_wgrad_kernel(const half_t* __restrict__ scratch, int64_t n_tiles,
                                                                 const int32_t* __restrict__ slot, int64_t S,
                                                                 float* __restrict__ grad_params,
                                                                 float* __restrict__ grad_code, int n_code_rows,
                                                                 const int64_t* __restrict__ n_dev,
                                                                 float* __restrict__ slot_sums,
                                                                 float* __restrict__ partials) {
    constexpr int TILE_KB = Lay<SLOTS>::TILE_KB;
    NSX_DEVICE_COUNT(S, n_tiles, 32, n_dev);
    __shared__ __attribute__((aligned(16))) char ring[WG_NS * WG_PIECES * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = lane & 31, kb = lane >> 5;
    const int type = blockIdx.x;
    const int64_t per = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int64_t t_begin = (int64_t)blockIdx.y * per, t_end = (t_begin + per < n_tiles) ? t_begin + per : n_tiles;
    if (t_begin >= t_end) return;                            // whole block
    const bool with_slots = SLOTS ? (type == 0) : (type == 4);
    const WgRole role = wg_role<SLOTS>(type, wave, n_code_rows);
    const int n_onehot = (SLOTS && type == 0) ? (n_code_rows + 31) / 32 : 0;      // one-hot Y tiles (<= 4: acc[2..5])
    // this wave's WG_K copies per stage
    int src_kb[WG_K], dst_kb[WG_K];
#pragma unroll
    for (int k = 0; k < WG_K; ++k) src_kb[k] = wg_piece_src<SLOTS>(type, wave + NW * k, dst_kb[k], with_slots);
    const char* sbytes = reinterpret_cast<const char*>(scratch);
    auto issue = [&](int64_t tile, int ring_slot) {
        const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;      // past the end: harmless re-copy (uniform vmcnt)
        char* stage = ring + (size_t)ring_slot * (WG_PIECES * 1024);
#pragma unroll
        for (int k = 0; k < WG_K; ++k) {
            if (src_kb[k] >= 0) {
                const char* g = sbytes + (size_t)tc * (TILE_KB * 1024) + (size_t)src_kb[k] * 1024 + (size_t)lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(stage + dst_kb[k] * 1024), 16, 0, 0);
            } else {                                                  // the tile's 32 code slots (+ 32 of the next tile)
                int64_t si = tc * 32 + lane;
                if (si >= S) si = S - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(slot + si),
                                                 (__attribute__((address_space(3))) void*)(stage + dst_kb[k] * 1024), 4, 0, 0);
            }
        }
    };
    f32x16 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = zero16();
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;
#pragma unroll
    for (int p = 0; p < WG_NS - 1; ++p) issue(t_begin + p, p);
    int cur = 0;
    for (int64_t t = t_begin; t < t_end; ++t) {
        // stage t landed: at most the copies of the WG_NS - 2 younger stages may still be in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_K * (WG_NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(t + WG_NS - 1, (cur + WG_NS - 1) % WG_NS);     // into the slot everybody finished reading before the barrier
        if (role.n_y > 0) {
            const char* stage = ring + (size_t)cur * (WG_PIECES * 1024);
            f16x8 x0, x1;
            if (role.x_kb >= 0) {
                x0 = *reinterpret_cast<const f16x8*>(stage + role.x_kb * 1024 + lane * 16);
                x1 = *reinterpret_cast<const f16x8*>(stage + role.x_kb * 1024 + 1024 + lane * 16);
            } else {
                const int32_t* sl = reinterpret_cast<const int32_t*>(stage + 8 * 1024);
                const int row = 32 * wave + i;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s0 = acc_row(j, kb), s1 = acc_row(8 + j, kb);
                    x0[j] = (t * 32 + s0 < S && sl[s0] == row) ? (half_t)1.f : (half_t)0.f;
                    x1[j] = (t * 32 + s1 < S && sl[s1] == row) ? (half_t)1.f : (half_t)0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (q < role.n_y) {
                    const f16x8 y0 = *reinterpret_cast<const f16x8*>(stage + (role.y_kb + 2 * q) * 1024 + lane * 16);
                    const f16x8 y1 = *reinterpret_cast<const f16x8*>(stage + (role.y_kb + 2 * q + 1) * 1024 + lane * 16);
                    acc[q] = mfma(x0, y0, acc[q]);
                    acc[q] = mfma(x1, y1, acc[q]);
                }
            }
            if (role.bias) {
                acc[6] = mfma(x0, ones, acc[6]);
                acc[6] = mfma(x1, ones, acc[6]);
            }
            if constexpr (SLOTS) {
                if (n_onehot > 0) {
                    // per-slot sums of this wave's dZ rows: Y = onehot(slot) built from the tile's 32 slots in the stage
                    const int32_t* sl = reinterpret_cast<const int32_t*>(stage + WG_SLOT_KB * 1024);
                    int32_t s0v[8], s1v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int s0 = acc_row(j, kb), s1 = acc_row(8 + j, kb);
                        s0v[j] = (t * 32 + s0 < S) ? sl[s0] : -1;
                        s1v[j] = (t * 32 + s1 < S) ? sl[s1] : -1;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q < n_onehot) {
                            const int row = 32 * q + i;
                            f16x8 y0, y1;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                y0[j] = (s0v[j] == row) ? (half_t)1.f : (half_t)0.f;
                                y1[j] = (s1v[j] == row) ? (half_t)1.f : (half_t)0.f;
                            }
                            acc[2 + q] = mfma(x0, y0, acc[2 + q]);
                            acc[2 + q] = mfma(x1, y1, acc[2 + q]);
                        }
                    }
                }
            }
        }
        cur = (cur + 1) % WG_NS;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA write may outlive the block
    if (role.n_y == 0) return;
    float* part = partials ? partials + (int64_t)blockIdx.y * P_TOTAL : nullptr;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if (q < role.n_y) wg_store<SLOTS>(acc[q], role.job, role.x_tile, q, false, lane, grad_params, grad_code, n_code_rows, part);
    if (role.bias) wg_store<SLOTS>(acc[6], role.job, role.x_tile, 0, true, lane, grad_params, grad_code, n_code_rows, part);
    if constexpr (SLOTS) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < n_onehot) wg_store_slot_sums(acc[2 + q], role.job == 4 ? 1 : 0, role.x_tile, q, lane, slot_sums, n_code_rows);
    }
}

// SLOTS, last step: the three small products that finish the code columns and the code-table gradient from the per-slot
// sums R0 / R4 ([128 neurons][128 code rows] fp32 each):
//   blocks [0, 128):        neuron n:  dW0[n][45 + c] += sum_r R0[n][r] code16[r][c],  dW4[n][45 + c] += sum_r R4[n][r] code16[r][c]
//   blocks [128, 128 + R):  code row r: dcode[r][c]   += sum_n W0c[n][c] R0[n][r] + W4c[n][c] R4[n][r]
// code16 = the fp16-rounded code the forward multiplied with (build_input), W0c / W4c = the fp16 code columns of W0 / W4
// as packed for the chain kernel (fragment groups B0C / B4C).  ~6 M multiply-adds: microseconds.
__device__ __forceinline__ float packed_code_weight(const half_t* __restrict__ frags16, int group, int n, int c) {
    // inverse of pack_source for the groups B0C / B4C: element (o = neuron n, code column c = 32 mt + i)
    const int mt = c >> 5, i = c & 31;
    const int n32 = n & 31, kb = (n32 >> 2) & 1, r = (n & 3) + 4 * (n32 >> 3);
    const int t = 2 * (n >> 5) + (r >> 3), j = r & 7;
    const int fi = group + mt * DF_TW + t;
    return (float)frags16[((int64_t)fi * 64 + kb * 32 + i) * 8 + j];
}

// blocks [0, n_reduce): grad_params[i] += sum over the non-empty chunks of partials[chunk][i], for every parameter the
// weight-gradient kernel owns (everything but the code columns of W0 / W4, which the blocks behind them produce).
constexpr int FINISH_REDUCE_BLOCKS = (P_TOTAL + 255) / 256;
// blocks [n_reduce, n_reduce + n_head): the head gradients = the sum of the chain kernel's per-block rows, 16 threads per
// entry (a serial walk over 256 rows per thread took 75 us)
constexpr int FINISH_HEAD_BLOCKS = (HEAD_PARAMS * 16 + 255) / 256;

__global__ __launch_bounds__(256) void deform_code_expand_kernel(const float* __restrict__ slot_sums,
                                                                 const float* __restrict__ code, int64_t code_stride,
                                                                 int n_code_rows, const f16x8* __restrict__ frags,
                                                                 float* __restrict__ grad_params,
                                                                 float* __restrict__ grad_code,
                                                                 const float* __restrict__ partials, int n_chunks,
                                                                 const float* __restrict__ head_partials, int n_chain_blocks,
                                                                 int64_t n_tiles, int64_t S,
                                                                 const int64_t* __restrict__ n_dev) {
    __shared__ float red[256];
    if ((int)blockIdx.x < FINISH_REDUCE_BLOCKS) {
        if (!partials) return;
        // the chunks that had tiles: the same split as deform_wgrad_kernel (device-side sample count included)
        if (n_dev) {
            const int64_t c = *n_dev;
            if (c < S) { S = c < 0 ? 0 : c; n_tiles = (S + 31) / 32; }
        }
        if (S <= 0) return;
        const int64_t per = (n_tiles + n_chunks - 1) / n_chunks;
        const int n_valid = (int)((n_tiles + per - 1) / per);
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx >= P_TOTAL) return;
        int col = -1;                                        // code columns of W0 / W4 are not the reducer's
        if (idx < P_B0) col = idx % DF_IN;
        else if (idx >= P_W4 && idx < P_B4) col = (idx - P_W4) % DF_W4;
        if (col >= DF_PE && col < DF_IN) return;
        if (idx >= P_WR) return;                             // the heads: the blocks behind these
        float acc = 0.f;
        for (int c = 0; c < n_valid; ++c) acc += partials[(int64_t)c * P_TOTAL + idx];
        grad_params[idx] += acc;
        return;
    }
    if ((int)blockIdx.x < FINISH_REDUCE_BLOCKS + FINISH_HEAD_BLOCKS) {
        if (!head_partials) return;
        if (n_dev) {
            const int64_t c = *n_dev;
            if (c < S) S = c;
        }
        if (S <= 0) return;                                  // (the chain kernel wrote nothing)
        const int t = (blockIdx.x - FINISH_REDUCE_BLOCKS) * 256 + threadIdx.x;
        const int e = t >> 4, part = t & 15;
        float acc = 0.f;
        if (e < HEAD_PARAMS)
            for (int c = part; c < n_chain_blocks; c += 16) acc += head_partials[(int64_t)c * HEAD_STRIDE + e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) acc += __shfl_xor(acc, m);
        if (e < HEAD_PARAMS && part == 0) grad_params[P_WR + e] += acc;
        return;
    }
    const int blk = blockIdx.x - FINISH_REDUCE_BLOCKS - FINISH_HEAD_BLOCKS;
    const float* R0 = slot_sums;
    const float* R4 = slot_sums + (int64_t)DFW * 128;
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    if (blk < DFW) {
        const int n = blk;
        const float* R = half ? R4 : R0;
        float acc = 0.f;
        for (int r = 0; r < n_code_rows; ++r) {
            const float v = R[(int64_t)n * 128 + r];
            if (v != 0.f) acc = __fmaf_rn(v, (float)(half_t)code[(int64_t)r * code_stride + c], acc);
        }
        float* dst = grad_params + (half ? (int64_t)P_W4 + (int64_t)n * DF_W4 : (int64_t)P_W0 + (int64_t)n * DF_IN) + DF_PE + c;
        *dst += acc;                                        // (the weight-gradient kernel left these columns alone)
    } else {
        const int r = blk - DFW;
        if (!grad_code) return;
        const half_t* f16 = reinterpret_cast<const half_t*>(frags);
        float acc = 0.f;
        for (int n = 64 * half; n < 64 * half + 64; ++n) {
            acc = __fmaf_rn(packed_code_weight(f16, B0C, n, c), R0[(int64_t)n * 128 + r], acc);
            acc = __fmaf_rn(packed_code_weight(f16, B4C, n, c), R4[(int64_t)n * 128 + r], acc);
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (half == 0) grad_code[(int64_t)r * DF_CODE + c] += red[c] + red[128 + c];
    }
}

static void fill_args(DeformArgs& A, const float* pos, int64_t S, const float* aabb, const float* code,
                      int64_t code_stride, const int32_t* slot, const float* window7, const void* frags,
                      const float* bias) {
    A.pos = pos; A.code = code; A.code_stride = code_stride; A.slot = slot; A.S = S;
    for (int d = 0; d < 3; ++d) {
        A.aabb_min[d] = aabb[d];
        A.aabb_ext[d] = aabb[3 + d] - aabb[d];
        A.aabb_inv[d] = 1.0f / A.aabb_ext[d];
    }
    for (int f = 0; f < 7; ++f) A.window[f] = window7 ? window7[f] : 1.0f;
    A.frags = reinterpret_cast<const f16x8*>(frags);
    A.bias = bias;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_deform_param_count(void) { return P_TOTAL; }
int64_t nsx_deform_pack_bytes(void) { return (int64_t)N_FRAGS * 64 * 16 + (int64_t)N_BIAS * 4; }
// + one private dummy tile per possible wave of the launch (tail waves of the lock-stepped blocks write there)
// (the larger of the two tile layouts + the per-slot sums at the head of the buffer)
// + one private parameter-gradient vector per chunk of the weight-gradient kernel (SLOTS: plain stores + one reduction)
static int64_t tiles_bytes(int64_t S) { return (((S + 31) / 32) + (int64_t)num_cus() * NW) * Lay<false>::TILE_HALFS * 2; }
static int wgrad_chunks_max() { return num_cus() / (WG_TYPES - 1); }
// + the row terms of the backward's forward recompute (<= 128 code rows)
constexpr int64_t BWD_TERMS_BYTES = 128 * TERM_ROW * 4;
int64_t nsx_deform_scratch_bytes(int64_t S) {
    return SLOT_SUMS_BYTES + tiles_bytes(S) + (int64_t)wgrad_chunks_max() * P_TOTAL * 4
           + (int64_t)num_cus() * HEAD_STRIDE * 4 + BWD_TERMS_BYTES;
}

int nsx_deform_pack(const float* params, void* packed, void* stream) {
    NSX_REQUIRE(params && packed, "nsx_deform_pack: NULL argument");
    f16x8* frags = reinterpret_cast<f16x8*>(packed);
    float* bias = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(packed) + (size_t)N_FRAGS * 64 * 16);
    DeformParamSrc src;
    src.flat = params;
    for (int k = 0; k < 16; ++k) src.t[k] = nullptr;
    hipLaunchKernelGGL(deform_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, src, frags, bias);
    NSX_LAUNCH_CHECK("nsx_deform_pack launch");
    return NSX_OK;
}

int nsx_deform_pack_tensors(const void* const* tensors16_host, void* packed, void* stream) {
    NSX_REQUIRE(tensors16_host && packed, "nsx_deform_pack_tensors: NULL argument");
    DeformParamSrc src;
    src.flat = nullptr;
    for (int k = 0; k < 16; ++k) {
        NSX_REQUIRE(tensors16_host[k], "nsx_deform_pack_tensors: tensor %d is NULL", k);
        src.t[k] = reinterpret_cast<const float*>(tensors16_host[k]);
    }
    f16x8* frags = reinterpret_cast<f16x8*>(packed);
    float* bias = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(packed) + (size_t)N_FRAGS * 64 * 16);
    hipLaunchKernelGGL(deform_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, src, frags, bias);
    NSX_LAUNCH_CHECK("nsx_deform_pack_tensors launch");
    return NSX_OK;
}

int nsx_deform_fwd(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code,
                   int64_t code_stride, const int32_t* code_slot, const float* window7_host, float* offsets,
                   const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_deform_fwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(packed && positions && aabb_host && code && offsets, "nsx_deform_fwd: NULL argument");
    DeformArgs A;
    const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(packed) + (size_t)N_FRAGS * 64 * 16);
    fill_args(A, positions, S, aabb_host, code, code_stride, code_slot, window7_host, packed, bias);
    const int64_t n_tiles = (S + 31) / 32;
    int64_t blocks = (n_tiles + NW - 1) / NW;
    if (blocks > num_cus()) blocks = num_cus();
    hipLaunchKernelGGL(deform_fwd_kernel, dim3((unsigned)blocks), dim3(NW * 64), 0, (hipStream_t)stream, A, offsets, n_tiles,
                       n_device);
    NSX_LAUNCH_CHECK("nsx_deform_fwd launch");
    return NSX_OK;
}

int64_t nsx_deform_terms_floats(int n_code_rows) { return (int64_t)(n_code_rows > 0 ? n_code_rows : 0) * TERM_ROW; }

int nsx_deform_fwd_rows(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code_table,
                        int64_t code_stride, const int32_t* code_slot, int n_code_rows, const float* window7_host,
                        float* offsets, float* terms_scratch, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_deform_fwd_rows: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(packed && positions && aabb_host && code_table && offsets && terms_scratch,
                "nsx_deform_fwd_rows: NULL argument");
    NSX_REQUIRE(n_code_rows >= 1, "nsx_deform_fwd_rows: n_code_rows=%d", n_code_rows);
    NSX_REQUIRE(code_slot || n_code_rows == 1, "nsx_deform_fwd_rows: code_slot may only be NULL for a one-row table (got %d rows)",
                n_code_rows);
    DeformArgs A;
    const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(packed) + (size_t)N_FRAGS * 64 * 16);
    fill_args(A, positions, S, aabb_host, code_table, code_stride, code_slot, window7_host, packed, bias);
    hipLaunchKernelGGL(deform_code_terms_kernel, dim3(2 * n_code_rows), dim3(DFW), 0, (hipStream_t)stream, A.frags, A.bias,
                       code_table, code_stride, n_code_rows, terms_scratch);
    NSX_LAUNCH_CHECK("nsx_deform_fwd_rows terms launch");
    const int64_t n_tiles = (S + 31) / 32;
    int64_t blocks = (n_tiles + NW - 1) / NW;
    if (blocks > num_cus()) blocks = num_cus();
    if (n_code_rows <= TERM_LDS_ROWS)
        hipLaunchKernelGGL(deform_fwd_terms_kernel<true>, dim3((unsigned)blocks), dim3(NW * 64), 0, (hipStream_t)stream, A,
                           (const float*)terms_scratch, n_code_rows, offsets, n_tiles, n_device);
    else                                                         // the terms stay in global memory (L2): same numbers, same order
        hipLaunchKernelGGL(deform_fwd_terms_kernel<false>, dim3((unsigned)blocks), dim3(NW * 64), 0, (hipStream_t)stream, A,
                           (const float*)terms_scratch, n_code_rows, offsets, n_tiles, n_device);
    NSX_LAUNCH_CHECK("nsx_deform_fwd_rows launch");
    return NSX_OK;
}

int nsx_deform_bwd(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code,
                   int64_t code_stride, const int32_t* code_slot, int n_code_rows, const float* window7_host,
                   const float* grad_offsets, void* scratch, float* grad_params, float* grad_code_table,
                   float* grad_code_samples, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_deform_bwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(packed && positions && aabb_host && code && grad_offsets && scratch && grad_params,
                "nsx_deform_bwd: NULL argument");
    NSX_REQUIRE(!grad_code_table || (code_slot && n_code_rows >= 1 && n_code_rows <= 128),
                "nsx_deform_bwd: grad_code_table needs code_slot and n_code_rows in [1,128] (got %d)", n_code_rows);
    DeformArgs A;
    const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(packed) + (size_t)N_FRAGS * 64 * 16);
    fill_args(A, positions, S, aabb_host, code, code_stride, code_slot, window7_host, packed, bias);
    const int64_t n_tiles = (S + 31) / 32;
    int64_t blocks = (n_tiles + NW - 1) / NW;
    if (blocks > num_cus()) blocks = num_cus();
    hipStream_t st = (hipStream_t)stream;
    float* slot_sums = reinterpret_cast<float*>(scratch);
    half_t* sc = reinterpret_cast<half_t*>(reinterpret_cast<uint8_t*>(scratch) + SLOT_SUMS_BYTES);
    // the warp codes are rows of a small table and nobody asks for the per-sample code gradient: everything that touches
    // the code columns is formed through the slot (Lay<true>)
    const bool slots = code_slot && grad_code_table && !grad_code_samples;
    float* chunk_partials = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(scratch) + SLOT_SUMS_BYTES + tiles_bytes(S));
    float* head_partials = chunk_partials + (int64_t)wgrad_chunks_max() * P_TOTAL;
    if (slots) {
        // the forward recompute is the terms forward of nsx_deform_fwd_rows (the same pre-activations bit for bit)
        float* terms = head_partials + (int64_t)num_cus() * HEAD_STRIDE;
        hipLaunchKernelGGL(deform_code_terms_kernel, dim3(2 * n_code_rows), dim3(DFW), 0, st, A.frags, A.bias, code, code_stride,
                           n_code_rows, terms);
        NSX_LAUNCH_CHECK("nsx_deform_bwd terms launch");
        if (n_code_rows <= BWD_TERM_LDS_ROWS)
            hipLaunchKernelGGL((deform_bwd_kernel<true, 1>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, A, grad_offsets, sc,
                               n_tiles, grad_code_samples, n_device, slot_sums, head_partials, (const float*)terms, n_code_rows);
        else
            hipLaunchKernelGGL((deform_bwd_kernel<true, 2>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, A, grad_offsets, sc,
                               n_tiles, grad_code_samples, n_device, slot_sums, head_partials, (const float*)terms, n_code_rows);
    } else
        hipLaunchKernelGGL(deform_bwd_kernel<false>, dim3((unsigned)blocks), dim3(NW * 64), 0, st, A, grad_offsets, sc,
                           n_tiles, grad_code_samples, n_device, slot_sums, (float*)nullptr);
    NSX_LAUNCH_CHECK("nsx_deform_bwd chain launch");
    // weight / bias / code-table gradients
    const int n_types = (grad_code_table && !slots) ? WG_TYPES : WG_TYPES - 1;
    int chunks = num_cus() / n_types;                 // one block per CU
    const int64_t max_chunks = (n_tiles + 3) / 4;     // >= 4 sample tiles per block
    if (chunks > max_chunks) chunks = (int)max_chunks;
    if (chunks < 1) chunks = 1;
    if (slots) {
        float* partials = chunk_partials;
        hipLaunchKernelGGL(deform_wgrad_kernel<true>, dim3(n_types, chunks), dim3(NW * 64), 0, st, sc, n_tiles, code_slot, S,
                           grad_params, grad_code_table, n_code_rows, n_device, slot_sums, partials);
        NSX_LAUNCH_CHECK("nsx_deform_bwd wgrad launch");
        hipLaunchKernelGGL(deform_code_expand_kernel, dim3(FINISH_REDUCE_BLOCKS + FINISH_HEAD_BLOCKS + DFW + n_code_rows),
                           dim3(256), 0, st,
                           slot_sums, code, code_stride, n_code_rows, A.frags, grad_params, grad_code_table, partials, chunks,
                           head_partials, (int)blocks, n_tiles, S, n_device);
        NSX_LAUNCH_CHECK("nsx_deform_bwd finish launch");
    } else {
        hipLaunchKernelGGL(deform_wgrad_kernel<false>, dim3(n_types, chunks), dim3(NW * 64), 0, st, sc, n_tiles, code_slot, S,
                           grad_params, grad_code_table, grad_code_table ? n_code_rows : 0, n_device, slot_sums,
                           (float*)nullptr);
        NSX_LAUNCH_CHECK("nsx_deform_bwd wgrad launch");
    }
    return NSX_OK;
}

}  // extern "C"

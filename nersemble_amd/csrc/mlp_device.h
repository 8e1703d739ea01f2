// mlp_device.h -- device-side pieces of the fully fused small MLPs (fragment plan, weight staging, input fragments, the
// chained forward of one 32-sample tile), shared by mlp.hip (nsx_mlp_fwd / nsx_mlp_bwd) and density_fused.hip (the no-grad
// density pass that keeps the hash features in registers).  See mlp.hip for the design notes.
#pragma once
#include "nsx_common.h"


namespace nsx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MLP_W = 64;        // hidden width
constexpr int MLP_IN = 32;       // padded input width
constexpr int MLP_OUT = 16;      // padded output width
constexpr int MLP_WAVES = 4;

struct MlpIO {
    // input vector of sample b = [ a[b][0..a_dim) * a_mul + a_add  (fp32 source),  bsrc[b][b_off .. b_off+b_dim) (fp16 source), 0... ]
    const float* a; int64_t a_stride; int a_dim; float a_mul, a_add;
    const half_t* b; int64_t b_stride; int b_off; int b_dim;
};

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// accumulator register r of lane-half `half` holds row:
__device__ __host__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// k-permutation when the accumulator tile mt of the previous layer is used as B operand:
// K-step t = 2*mt + tt uses registers 8*tt .. 8*tt+7 -> actual neuron index
__device__ __host__ __forceinline__ int kmap_chain(int t, int kb, int j) {
    return 32 * (t >> 1) + acc_row(8 * (t & 1) + j, kb);
}
__device__ __host__ __forceinline__ int kmap_natural(int t, int kb, int j) { return 16 * t + 8 * kb + j; }

// ---- LDS weight fragments ----------------------------------------------------------------------
// Fragment (mt, t) of a matrix product D = Wm * Xm with Wm[M][K]: lane (i = lane&31, kb = lane>>5) element j
// holds Wm[32*mt + i][kmap(t, kb, j)] (zero outside the matrix).
struct FragPlan {
    // forward
    int w0;          // [2 mt][2 t]    W0 [64][32], natural k
    int wh;          // [2 mt][4 t]    Wh [64][64], chained k        (only if NH == 1)
    int wo;          // [1 mt][4 t]    Wo [16->32][64], chained k
    // backward (transposed matrices, K = output neurons in chained order)
    int woT;         // [2 mt][1 t]    Wo^T [64][16->(K-step 0 only)]
    int whT;         // [2 mt][4 t]    Wh^T [64][64]
    int w0T;         // [1 mt][4 t]    W0^T [32][64]
    int total;
};
__device__ __host__ inline FragPlan make_plan(int NH, bool bwd) {
    FragPlan p{};
    int n = 0;
    p.w0 = n; n += 4;
    p.wh = n; if (NH) n += 8;
    p.wo = n; n += 4;
    if (bwd) {
        p.woT = n; n += 2;
        p.whT = n; if (NH) n += 8;
        p.w0T = n; n += 4;
    }
    p.total = n;
    return p;
}

// Staging the fragments.  FORWARD fragments (lane = matrix row): the 8 halfs of a lane are 16 consecutive bytes of its row
// (natural k) or two runs of 8 bytes (chained k: kmap_chain(t, kb, j) = 32 (t >> 1) + 16 (t & 1) + 4 kb + 8 (j >> 2) + (j & 3)),
// so they are read straight from the flat weight vector in global memory (L2-resident: 6-14 KB) with one 16-byte or two
// 8-byte loads and stored to LDS as one 16-byte word -- no staging copy, no block barrier in front of the gather, no LDS
// bank conflicts.  (Rounds 1-4 copied the vector to LDS and gathered every fragment from there with 2-byte reads whose lanes
// sit one matrix row = 64 or 128 bytes apart: 16- to 32-way bank conflicts, 84-89 % of the forward kernel's LDS-active
// cycles -- profiles/pmc/r04_sq_mfma_kernels.json.)  TRANSPOSED fragments of the backward (lane = matrix COLUMN: the lanes
// of one read are consecutive halfs of a row) keep the staged copy: their 2-byte LDS reads are conflict-free.
// A weight vector that is not 16-byte aligned takes the element-wise route for everything.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 chain_row_frag(const half_t* row, int t, int kb) {
    const int base = 32 * (t >> 1) + 16 * (t & 1) + 4 * kb;
    const f16x4 lo = *reinterpret_cast<const f16x4*>(row + base), hi = *reinterpret_cast<const f16x4*>(row + base + 8);
    f16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
}

template <int NH>
__device__ void stage_weights(const half_t* __restrict__ Wg, half_t* __restrict__ tmp, f16x8* frags, bool bwd) {
    constexpr int NPARAM = MLP_W * MLP_IN + (NH ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W;     // halfs, a multiple of 8
    const bool aligned = (reinterpret_cast<uintptr_t>(Wg) & 15) == 0;
    const FragPlan p = make_plan(NH, bwd);
    const int n_fwd = bwd ? p.woT : p.total;                    // fragments [0, n_fwd) are forward fragments
    if (aligned) {
        const half_t* W0g = Wg;
        const half_t* Whg = Wg + MLP_W * MLP_IN;
        const half_t* Wog = Whg + (NH ? MLP_W * MLP_W : 0);
        for (int e = threadIdx.x; e < n_fwd * kWave; e += blockDim.x) {
            const int fi = e / kWave, ll = e % kWave;
            const int i = ll & 31, kb = ll >> 5;
            f16x8 v;
            if (fi < p.wh) {                       // W0: natural k
                const int mt = (fi - p.w0) >> 1, t = (fi - p.w0) & 1;
                v = *reinterpret_cast<const f16x8*>(W0g + (32 * mt + i) * MLP_IN + 16 * t + 8 * kb);
            } else if (fi < p.wo) {                // Wh: chained k
                const int mt = (fi - p.wh) >> 2, t = (fi - p.wh) & 3;
                v = chain_row_frag(Whg + (32 * mt + i) * MLP_W, t, kb);
            } else {                               // Wo (rows >= 16 are zero padding)
                v = chain_row_frag(Wog + (i < MLP_OUT ? i : 0) * MLP_W, fi - p.wo, kb);
                if (i >= MLP_OUT) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
                }
            }
            frags[e] = v;
        }
        if (!bwd) return;
    }
    const half_t* W = Wg;
    if (aligned) {
        for (int i = threadIdx.x; i < NPARAM / 8; i += blockDim.x)
            reinterpret_cast<f16x8*>(tmp)[i] = reinterpret_cast<const f16x8*>(Wg)[i];
        __syncthreads();
        W = tmp;
    }
    // flat parameter layout: W0 [64][32], (Wh [64][64]), Wo [16][64]
    const half_t* W0 = W;
    const half_t* Wh = W + MLP_W * MLP_IN;
    const half_t* Wo = Wh + (NH ? MLP_W * MLP_W : 0);
    for (int e = (aligned ? n_fwd * kWave : 0) + threadIdx.x; e < p.total * kWave; e += blockDim.x) {
        const int fi = e / kWave, ll = e % kWave;
        const int i = ll & 31, kb = ll >> 5;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t x = (half_t)0.f;
            if (fi < p.wh) {                       // W0: mt = (fi-p.w0)/2, t = %2, natural k
                const int mt = (fi - p.w0) >> 1, t = (fi - p.w0) & 1;
                x = W0[(32 * mt + i) * MLP_IN + kmap_natural(t, kb, j)];
            } else if (fi < p.wo) {                // Wh
                const int mt = (fi - p.wh) >> 2, t = (fi - p.wh) & 3;
                x = Wh[(32 * mt + i) * MLP_W + kmap_chain(t, kb, j)];
            } else if (!bwd || fi < p.woT) {       // Wo (rows >= 16 are zero padding)
                const int t = fi - p.wo;
                if (i < MLP_OUT) x = Wo[i * MLP_W + kmap_chain(t, kb, j)];
            } else if (fi < p.whT) {               // Wo^T: M = hidden neuron (2 tiles), K-step 0 of chained out neurons
                const int mt = fi - p.woT;
                const int o = kmap_chain(0, kb, j);          // rows 0..3,8..11 / 4..7,12..15
                if (o < MLP_OUT) x = Wo[o * MLP_W + 32 * mt + i];
            } else if (fi < p.w0T) {               // Wh^T
                const int mt = (fi - p.whT) >> 2, t = (fi - p.whT) & 3;
                x = Wh[kmap_chain(t, kb, j) * MLP_W + 32 * mt + i];
            } else {                               // W0^T: M = input feature (1 tile), K = 64 hidden chained
                const int t = fi - p.w0T;
                x = W0[kmap_chain(t, kb, j) * MLP_IN + i];
            }
            v[j] = x;
        }
        frags[e] = v;
    }
}

// ---- input fragments (orientation: lane = sample) ------------------------------------------------
__device__ __forceinline__ half_t input_elem(const MlpIO& io, int64_t b, int k) {
    if (k < io.a_dim) return (half_t)__fmaf_rn(io.a[b * io.a_stride + k], io.a_mul, io.a_add);
    k -= io.a_dim;
    if (k < io.b_dim) return io.b[b * io.b_stride + io.b_off + k];
    return (half_t)0.f;
}

// how the input fragments are read: 1 = the whole 32-wide fp16 row (mlp_base: hash features), 2 = mlp_head's
// [(d + 1) / 2 (3 fp32), mlp_base's row without its first column (15 fp16)] from two 16-byte loads of the 16-wide row
// + three floats, 0 = element by element (any other layout)
__device__ __forceinline__ int input_mode(const MlpIO& io) {
    if (io.a_dim == 0 && io.b_off == 0 && io.b_dim == MLP_IN && (io.b_stride % 8) == 0 &&
        (reinterpret_cast<uintptr_t>(io.b) & 15) == 0) return 1;
    if (io.a_dim == 3 && io.b_off == 1 && io.b_dim == 15 && io.b_stride == 16 && (reinterpret_cast<uintptr_t>(io.b) & 15) == 0)
        return 2;
    return 0;
}

__device__ __forceinline__ void load_input(const MlpIO& io, int64_t b, int kb, int mode, f16x8 x[2]) {
    if (mode == 1) {
        const f16x8* row = reinterpret_cast<const f16x8*>(io.b + b * io.b_stride);
        x[0] = row[kb];
        x[1] = row[2 + kb];
    } else if (mode == 2) {
        // input k: 0..2 = a * a_mul + a_add, 3..17 = row[1..15], 18..31 = 0;  lane kb holds k = 16 t + 8 kb + j
        const f16x8* row = reinterpret_cast<const f16x8*>(io.b + b * 16);
        const f16x8 r0 = row[0], r1 = row[1];
        f16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f;
        if (kb == 0) {
            const float* a = io.a + b * io.a_stride;
            f16x8 v = z;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[j] = (half_t)__fmaf_rn(a[j], io.a_mul, io.a_add);
#pragma unroll
            for (int j = 3; j < 8; ++j) v[j] = r0[j - 2];
            x[0] = v;
            f16x8 w = z;
            w[0] = r1[6]; w[1] = r1[7];
            x[1] = w;
        } else {
            f16x8 v;
            v[0] = r0[6]; v[1] = r0[7];
#pragma unroll
            for (int j = 2; j < 8; ++j) v[j] = r1[j - 2];
            x[0] = v;
            x[1] = z;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[t][j] = input_elem(io, b, kmap_natural(t, kb, j));
    }
}

__device__ __forceinline__ void relu_pack(const f32x16& d, f16x8& lo, f16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (half_t)fmaxf(d[j], 0.f);
        hi[j] = (half_t)fmaxf(d[8 + j], 0.f);
    }
}

// forward chain for one tile; keeps hidden activations (fp16 fragments, chained-k order)
template <int NH>
__device__ __forceinline__ f32x16 forward_tile(const f16x8* frags, const FragPlan& p, int lane, const f16x8 x[2],
                                               f16x8 h1[4], f16x8 h2[4]) {
    f32x16 z[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        z[mt] = zero16();
#pragma unroll
        for (int t = 0; t < 2; ++t) z[mt] = mfma(frags[(p.w0 + mt * 2 + t) * kWave + lane], x[t], z[mt]);
        relu_pack(z[mt], h1[2 * mt], h1[2 * mt + 1]);
    }
    const f16x8* hl = h1;
    if constexpr (NH == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            z[mt] = zero16();
#pragma unroll
            for (int t = 0; t < 4; ++t) z[mt] = mfma(frags[(p.wh + mt * 4 + t) * kWave + lane], h1[t], z[mt]);
            relu_pack(z[mt], h2[2 * mt], h2[2 * mt + 1]);
        }
        hl = h2;
    }
    f32x16 o = zero16();
#pragma unroll
    for (int t = 0; t < 4; ++t) o = mfma(frags[(p.wo + t) * kWave + lane], hl[t], o);
    return o;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

}  // namespace nsx

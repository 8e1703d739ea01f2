// adam.hip -- fused Adam for the hash tables (403 M fp32 master parameters at H = 32) on gfx950.
//
// Replaces, for the `fields` parameter group, what the reference does per step through torch
// (scripts/train/train_nersemble.py:243-246 Adam(lr 5e-3, eps 1e-15); nersemble_trainer.py:185-186
// GradScaler unscale + inf check + optimizer step) plus tcnn's per-call fp32->fp16 parameter cast:
//   torch:  unscale pass (r+w grad) + ~8 multi-tensor passes over param/grad/m/v  + cast pass   ~ 60 B/param
//   here :  ONE pass: grad is formed on the fly from the factored gradient G (nsx_hash_ensemble_bwd_factored)
//           and the <= 64 code rows (the 1.6 GB table gradient is never materialised), unscaled, the moments and
//           the master weights updated and the fp16 working copy written:  ~ 29 B/param.
// Semantics are torch.optim.Adam's (no amsgrad / weight decay): m.lerp_(g, 1-b1); v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps); skipped entirely when *found_inf != 0 (GradScaler).
#include <cstdlib>
#include "nsx_common.h"

namespace nsx {

struct AdamHyper {
    float lr, beta1, beta2, eps, bc1, bc2_sqrt;     // bc1 = 1 - b1^t ; bc2_sqrt = sqrt(1 - b2^t)
};

// (no contraction inside, and none with the caller's unscale multiply: every kernel that updates a parameter -- fused,
// dense, packed -- then computes the same bits from the same inputs, which is what lets the data-parallel exchange change
// its format with the window without changing a run)
__device__ __forceinline__ void adam_update(float g, float& p, float& m, float& v, const AdamHyper& h) {
#pragma clang fp contract(off)
    m = m + (g - m) * (1.0f - h.beta1);
    v = v * h.beta2 + (1.0f - h.beta2) * g * g;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p = p - (h.lr / h.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ found_inf) {
    bool bad = false;
    const int64_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= !isfinite(x[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;
}

// Block = 256 threads = a tile of 2048/(2*HP) entries (32 at H = 32).  The tile's slice of G ([slot][entry][f], i.e.
// one contiguous run per slot) is staged through LDS with coalesced loads; every thread then owns 4 consecutive
// parameters (float4 streams of master / m / v, half4 store of the working copy) and forms their gradient from the
// LDS-resident G values and code rows.
// CLEAR: the kernel CONSUMES G -- every 16-byte piece it reads that holds a non-zero value is written back as zeros, so
// the buffer is ready for the next backward's scatter without a 1.6 GB fill in front of it (only the pieces the last
// scatter touched are written: a few percent of G once the occupancy grid has pruned the scene).  A skipped step
// (found_inf) still clears.
template <int HP, bool CLEAR>
__global__ __launch_bounds__(256) void adam_hash_factored_kernel(
    float* __restrict__ G, int n_slots, const float* __restrict__ code, int64_t code_stride,
    const float* __restrict__ window, int Hreal, uint64_t total, float* __restrict__ master, float* __restrict__ m,
    float* __restrict__ v, half_t* __restrict__ f16, AdamHyper hy, const float* __restrict__ inv_scale,
    const float* __restrict__ found_inf) {
    const bool skip = found_inf && found_inf[0] != 0.f;
    if (skip && !CLEAR) return;
    constexpr int HV = HP >= 4 ? 4 : HP;                 // parameters per thread (vector width)
    constexpr int TPE = 2 * HP / HV;                     // threads per entry
    constexpr int EPB = 256 / TPE;                       // entries per block tile
    extern __shared__ float smem[];
    float* cs = smem;                                    // [n_slots][HP] fp16-rounded windowed codes
    float* gs = smem + n_slots * HP;                     // [n_slots][EPB * 2]
    for (int i = threadIdx.x; i < n_slots * HP; i += blockDim.x) {
        const int sl = i / HP, h = i % HP;
        float c = 0.f;
        if (h < Hreal) c = code[sl * code_stride + h] * (window ? window[h] : 1.0f);
        cs[i] = (float)(half_t)c;
    }
    const float is = inv_scale ? inv_scale[0] : 1.0f;
    const uint64_t n_tiles = (total + EPB - 1) / EPB;
    const int le = threadIdx.x / TPE;                    // entry within the tile
    const int part = threadIdx.x % TPE;
    const int f = part / (TPE / 2);
    const int hq = part % (TPE / 2);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t e0 = tile * EPB;
        const uint64_t e = e0 + le;
        const bool live = e < total;
        const uint64_t at = (e * 2ull + f) * HP + hq * HV;
        // the three parameter streams are requested first: their HBM latency overlaps the staging of G below
        float pp[HV], mm[HV], vv[HV];
        if (live && !skip) {
#pragma unroll
            for (int k = 0; k < HV; ++k) {          // streamed once per step: keep them out of the way of G in L2
                pp[k] = __builtin_nontemporal_load(master + at + k);
                mm[k] = __builtin_nontemporal_load(m + at + k);
                vv[k] = __builtin_nontemporal_load(v + at + k);
            }
        }
        __syncthreads();
        {
            // one slot's run of the tile is EPB * 2 contiguous floats (a multiple of 4 for every HP): float4 pieces
            static_assert((EPB * 2) % 4 == 0, "tile run must be float4-divisible");
            constexpr int Q = EPB * 2 / 4;
            for (int i = threadIdx.x; i < n_slots * Q; i += blockDim.x) {
                const int sl = i / Q, j = (i % Q) * 4;
                const uint64_t ge = e0 * 2ull + j;
                float* src = G + (uint64_t)sl * total * 2ull + ge;
                float4 g4;
                if (ge + 3 < total * 2ull) {
                    g4 = *reinterpret_cast<const float4*>(src);
                    if (CLEAR && (g4.x != 0.f || g4.y != 0.f || g4.z != 0.f || g4.w != 0.f))
                        *reinterpret_cast<float4*>(src) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    g4 = make_float4(ge < total * 2ull ? src[0] : 0.f, ge + 1 < total * 2ull ? src[1] : 0.f,
                                     ge + 2 < total * 2ull ? src[2] : 0.f, 0.f);
                    if (CLEAR) {
                        for (int t = 0; t < 3; ++t)
                            if (ge + t < total * 2ull) src[t] = 0.f;
                    }
                }
                *reinterpret_cast<float4*>(gs + sl * (EPB * 2) + j) = g4;
            }
        }
        __syncthreads();
        if (!live || skip) continue;
        float g[HV];
#pragma unroll
        for (int k = 0; k < HV; ++k) g[k] = 0.f;
        for (int sl = 0; sl < n_slots; ++sl) {
            const float gv = gs[sl * (EPB * 2) + le * 2 + f];
            if (gv != 0.f) {
#pragma unroll
                for (int k = 0; k < HV; ++k) g[k] = __fmaf_rn(gv, cs[sl * HP + hq * HV + k], g[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            if (hq * HV + k < Hreal) adam_update(g[k] * is, pp[k], mm[k], vv[k], hy);
        }
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            __builtin_nontemporal_store(pp[k], master + at + k);
            __builtin_nontemporal_store(mm[k], m + at + k);
            __builtin_nontemporal_store(vv[k], v + at + k);
            f16[at + k] = (half_t)pp[k];                  // read again by the next forward pass
        }
    }
}

// ---- the same pass with the gradient formed on the matrix cores: many gradient planes --------------------------------------
// The level-parallel exchange (engine/level_parallel.py) hands this pass one gradient plane per (source rank, code row): up to
// 8 x 24 = 192 planes over 1 / N of the entries.  The gradient of an entry is then a [64 (entry, feature)] x [planes] x [32 grids]
// product per 32 entries -- 6 144 multiply-adds per parameter pair against 3.2 KB of HBM traffic: the kernel above, which walks
// the planes on the VALU out of 74 KB of LDS per block, takes 1.56 ms for the 2^20 entries of the finest two levels where their
// bytes take 0.42 ms.  Here a wave owns 32 consecutive (entry, feature) rows at a time and forms their gradient with
// v_mfma_f32_32x32x16_bf16: rows = (entry, feature), columns = grids -- the D layout of the instruction (lane = column, 16 rows per
// lane) is then exactly the table layout [(entry, feature)][grid]: every load / store of master / moments / working copy is a
// pair of full 128-byte lines per wave instruction.  G is fp32 and must stay so (loss-scaled sums: outside the fp16 range), the
// codes are fp16 values: G is split into three truncated bf16 pieces (8 + 8 + 8 bits: exact up to 2^-24 of the value), a code
// into two (exact), and five of the six products are accumulated (the sixth is below 2^-24) -- the sum carries fp32 accuracy;
// its rounding differs from the VALU chain's in the last bits, which is why only plane counts the VALU kernel was never given
// (> 64: no single-process run has them) come here.  A operand straight from global memory (lane = row, 8 planes per lane: eight
// dword loads whose half-waves read one 128-byte line each), B operand (codes) from LDS in fragment order.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float acc16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class P>
__device__ __forceinline__ P* at_bytes(P* base, unsigned int bytes) {
    return reinterpret_cast<P*>(reinterpret_cast<char*>(base) + bytes);
}

__device__ __forceinline__ unsigned int hi_halves(unsigned int a, unsigned int b) {      // (a >> 16) | (b & 0xffff0000)
    return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

#define MFMA_ADAM_WAVES(T) ((T) >= 8 ? 2 : 3)              /* waves per SIMD the register budget is set for */
// T: K steps of 16 planes, a template parameter -- the column of G a row block needs is then a fixed set of registers requested
// in one go (with T a run-time bound the loads end up behind per-step branches, each with its own wait).
template <int T, bool CLEAR>
__global__ __launch_bounds__(256, MFMA_ADAM_WAVES(T)) void adam_hash_factored_mfma_kernel(
    float* __restrict__ G, int n_slots, const float* __restrict__ code, int64_t code_stride,
    const float* __restrict__ window, int Hreal, uint64_t total, float* __restrict__ master, float* __restrict__ m,
    float* __restrict__ v, half_t* __restrict__ f16, AdamHyper hy, const float* __restrict__ inv_scale,
    const float* __restrict__ found_inf) {
    constexpr int HP = 32;
    const bool skip = found_inf && found_inf[0] != 0.f;
    if (skip && !CLEAR) return;
    extern __shared__ float smem[];
    // fragment order: [K step][half-wave][grid][8 planes] bf16, the high pieces then the low pieces
    unsigned short* ch = reinterpret_cast<unsigned short*>(smem);
    unsigned short* cl = ch + T * 16 * HP;
    for (int i = threadIdx.x; i < T * 16 * HP; i += blockDim.x) {
        const int sl = i / HP, h = i % HP;
        float c = 0.f;
        if (sl < n_slots && h < Hreal) c = code[sl * code_stride + h] * (window ? window[h] : 1.0f);
        c = (float)(half_t)c;
        const unsigned int u = __float_as_uint(c) & 0xffff0000u;
        const float lo = c - __uint_as_float(u);             // <= 3 significant bits: exact in bf16
        const int at = ((sl >> 3) * HP + h) * 8 + (sl & 7);
        ch[at] = (unsigned short)(u >> 16);
        cl[at] = (unsigned short)(__float_as_uint(lo) >> 16);
    }
    __syncthreads();
    const float is = inv_scale ? inv_scale[0] : 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    const uint64_t n_rows = total * 2ull;                    // (entry, feature) rows
    const uint64_t n_blocks = (n_rows + 31) / 32;
    const u32x4* bh = reinterpret_cast<const u32x4*>(ch);
    const u32x4* bl = reinterpret_cast<const u32x4*>(cl);
    const bool col_live = j < Hreal;
    // Addresses: a wave-uniform 64-bit base (SGPRs, scalar arithmetic) + one 32-bit lane offset that does not change over the
    // loop -- with per-lane 64-bit addresses the 96 loads of a column cost 192 registers before the first one is issued.
    // (BYTE offsets, so that base + zero-extended offset is the instruction's own addressing mode; the launcher checks that
    // 9 planes of the range stay below 4 GB)
    const unsigned int lane_g = (unsigned int)(kg * 8) * (unsigned int)n_rows * 4u;  // + the lane's row within the block
    const unsigned int lane_t = (unsigned int)(4 * kg * HP + j) * 4u;
    const uint64_t wave0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    for (uint64_t rb = wave0; rb < n_blocks; rb += (uint64_t)gridDim.x * 4) {
        const uint64_t row0 = rb * 32;
        const bool full = row0 + 32 <= n_rows;               // (all but the last row block of the range)
        // Every load of G is unconditional: a load under a per-lane condition is compiled into a branch with its own wait -- one
        // HBM round trip per element.  Rows past the end read the last row (their results are never stored), planes past
        // n_slots read a valid plane against code rows that are zero.
        const bool a_live = row0 + j < n_rows;
        const unsigned int jc = (a_live ? (unsigned int)j : (unsigned int)(n_rows - 1 - row0)) * 4u;
        const unsigned int off_g = lane_g + jc;
        float* gbase = G + row0;
        // the row block's whole column of G is requested at once (<= 96 registers): one HBM round trip per row block instead
        // of one per K step
        float gq[T][8];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int sa = t * 16 + i;                   // the plane of the lower half-wave; the upper one reads sa + 8
                if (t < T - 1) {
                    gq[t][i] = *at_bytes(gbase + (uint64_t)sa * n_rows, off_g);
                } else {
                    float* src = gbase + (uint64_t)(sa < n_slots ? sa : 0) * n_rows;
                    gq[t][i] = *at_bytes(src, (sa + 8 < n_slots) ? off_g : jc);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (or the scheduler sinks each K step's loads to its products)
        acc16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (CLEAR) {
                // (the plane stride through an opaque scalar: the stores' bases are then formed again here, 8 scalar adds,
                // instead of 96 address pairs kept from the loads)
                uint64_t stride = n_rows;
                asm volatile("" : "+s"(stride));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (gq[t][i] != 0.f && a_live && t * 16 + kg * 8 + i < n_slots)
                        *at_bytes(gbase + (uint64_t)(t * 16 + i) * stride, off_g) = 0.f;
            }
            if (skip) continue;
            unsigned int u1[8], u2[8], u3[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u1[i] = __float_as_uint(gq[t][i]);
                const float r = gq[t][i] - __uint_as_float(u1[i] & 0xffff0000u);    // exact
                u2[i] = __float_as_uint(r);
                const float r2 = r - __uint_as_float(u2[i] & 0xffff0000u);          // exact
                u3[i] = __float_as_uint(r2);
            }
            u32x4 a1, a2, a3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a1[i] = hi_halves(u1[2 * i], u1[2 * i + 1]);
                a2[i] = hi_halves(u2[2 * i], u2[2 * i + 1]);
                a3[i] = hi_halves(u3[2 * i], u3[2 * i + 1]);
            }
            const u32x4 h8 = bh[(t * 2 + kg) * HP + j];
            const u32x4 l8 = bl[(t * 2 + kg) * HP + j];
            const bf16x8 A1 = __builtin_bit_cast(bf16x8, a1), A2 = __builtin_bit_cast(bf16x8, a2),
                         A3 = __builtin_bit_cast(bf16x8, a3);
            const bf16x8 BH = __builtin_bit_cast(bf16x8, h8), BL = __builtin_bit_cast(bf16x8, l8);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, BH, acc, 0, 0, 0);      // the small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, BL, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, BL, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, BH, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, BH, acc, 0, 0, 0);
        }
        if (skip) continue;
        // the three parameter streams after the products (their registers are the column's until here; the other waves of
        // the CU cover the round trip)
        float pp[16], mm[16], vv[16];
        if (full) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint64_t at = (row0 + 8 * (q >> 2) + (q & 3)) * HP;
                pp[q] = __builtin_nontemporal_load(at_bytes(master + at, lane_t));
                mm[q] = __builtin_nontemporal_load(at_bytes(m + at, lane_t));
                vv[q] = __builtin_nontemporal_load(at_bytes(v + at, lane_t));
            }
            if (!col_live) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint64_t at = (row0 + 8 * (q >> 2) + (q & 3)) * HP;
                adam_update(acc[q] * is, pp[q], mm[q], vv[q], hy);
                __builtin_nontemporal_store(pp[q], at_bytes(master + at, lane_t));
                __builtin_nontemporal_store(mm[q], at_bytes(m + at, lane_t));
                __builtin_nontemporal_store(vv[q], at_bytes(v + at, lane_t));
                *at_bytes(f16 + at, lane_t / 2) = (half_t)pp[q];
            }
        } else {
            if (!col_live) continue;
#pragma unroll 1
            for (int q = 0; q < 16; ++q) {
                const uint64_t row = row0 + 8 * (q >> 2) + 4 * kg + (q & 3);
                if (row >= n_rows) continue;
                float p1 = master[row * HP + j], m1 = m[row * HP + j], v1 = v[row * HP + j];
                float gsel = 0.f;
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) gsel = qq == q ? acc[qq] : gsel;
                adam_update(gsel * is, p1, m1, v1, hy);
                master[row * HP + j] = p1;
                m[row * HP + j] = m1;
                v[row * HP + j] = v1;
                f16[row * HP + j] = (half_t)p1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void adam_dense_kernel(const float* __restrict__ grad, int64_t n,
                                                         float* __restrict__ master, float* __restrict__ m,
                                                         float* __restrict__ v, half_t* __restrict__ f16, AdamHyper hy,
                                                         const float* __restrict__ inv_scale,
                                                         const float* __restrict__ found_inf) {
    if (found_inf && found_inf[0] != 0.f) return;
    const float is = inv_scale ? inv_scale[0] : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = master[i], mm = m[i], vv = v[i];
        adam_update(grad[i] * is, p, mm, vv, hy);
        master[i] = p; m[i] = mm; v[i] = vv;
        if (f16) f16[i] = (half_t)p;
    }
}

// ---- data-parallel pieces (engine/sharded_adam.py): fp16 dense gradient for the reduce-scatter, inf check and Adam
// on the rank's shard ----
__global__ __launch_bounds__(256) void check_finite_f16_kernel(const half_t* __restrict__ x, int64_t n,
                                                               float* __restrict__ found_inf) {
    bool bad = false;
    // fp16 inf / NaN <=> exponent field all ones
    const int64_t n8 = n / 8;
    const uint4* x8 = reinterpret_cast<const uint4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = x8[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) bad |= ((w[k] & 0x7C00u) == 0x7C00u) || ((w[k] & 0x7C000000u) == 0x7C000000u);
    }
    const uint16_t* xs = reinterpret_cast<const uint16_t*>(x);
    for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= (xs[i] & 0x7C00u) == 0x7C00u;
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;
}

__global__ __launch_bounds__(256) void adam_dense_f16grad_kernel(const half_t* __restrict__ grad, int64_t n,
                                                                 float* __restrict__ master, float* __restrict__ m,
                                                                 float* __restrict__ v, half_t* __restrict__ f16,
                                                                 AdamHyper hy, const float* __restrict__ inv_scale,
                                                                 const float* __restrict__ found_inf) {
    if (found_inf && found_inf[0] != 0.f) {
        // GradScaler: nothing moves -- but the fp16 output still has to hold this rank's CURRENT values: in the narrow
        // exchange it is this rank's piece of the all-gather, a buffer that is re-allocated (as zeros) whenever the width
        // changes; a skip on such a step used to gather zeros into every rank's working tables (advisor, round 5)
        if (f16)
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
                f16[i] = (half_t)master[i];
        return;
    }
    const float is = inv_scale ? inv_scale[0] : 1.0f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 p4 = reinterpret_cast<const float4*>(master)[i];
        const float4 m4 = reinterpret_cast<const float4*>(m)[i];
        const float4 v4 = reinterpret_cast<const float4*>(v)[i];
        const uint2 g2 = reinterpret_cast<const uint2*>(grad)[i];
        float p[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const half2_t ga = __builtin_bit_cast(half2_t, g2.x), gb = __builtin_bit_cast(half2_t, g2.y);
        const float g[4] = {(float)ga.x, (float)ga.y, (float)gb.x, (float)gb.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(g[k] * is, p[k], mm[k], vv[k], hy);
        reinterpret_cast<float4*>(master)[i] = make_float4(p[0], p[1], p[2], p[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (f16) {
            half2_t oa, ob;
            oa.x = (half_t)p[0]; oa.y = (half_t)p[1]; ob.x = (half_t)p[2]; ob.y = (half_t)p[3];
            reinterpret_cast<uint2*>(f16)[i] = make_uint2(__builtin_bit_cast(uint32_t, oa), __builtin_bit_cast(uint32_t, ob));
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = master[i], mm = m[i], vv = v[i];
        adam_update((float)grad[i] * is, p, mm, vv, hy);
        master[i] = p; m[i] = mm; v[i] = vv;
        if (f16) f16[i] = (half_t)p;
    }
}

// out[e][f][h] (+)= scale * sum_slot G[slot][e][f] * code'[slot][h], written as fp16 (same tile staging as the fused
// Adam kernel: G crosses HBM once).
// Bucket mode (bucket_entries > 0; nsx_hash_grad_expand_f16_bucket): `out` is ONE bucket of a bucketed reduce-scatter --
// [rank][bucket_entries] entries, rank r's piece holding the table entries r * rank_entries + entry_base ... -- so the
// kernel walks VIRTUAL entries v = r * bucket_entries + w and reads the table entry they stand for; entries beyond the
// table (the padding of the last shard) are written as zeros.  bucket_entries is a multiple of the tile (EPB).
template <int HP>
__global__ __launch_bounds__(256) void expand_f16_kernel(
    const float* __restrict__ G, int n_slots, const float* __restrict__ code, int64_t code_stride,
    const float* __restrict__ window, int Hreal, uint64_t total, half_t* __restrict__ out, float scale, int accumulate,
    uint64_t bucket_entries, uint64_t rank_entries, uint64_t entry_base, uint64_t virtual_total, int out_width,
    float* __restrict__ beyond_width) {
    // out_width < HP (nsx_hash_grad_expand_f16_bucket_width): only the grids [0, out_width) are written, packed as
    // [entry][f][out_width] -- the caller knows the others' gradient to be zero (the coarse-to-fine window has not reached
    // them); a conditioned code that is NOT zero there raises *beyond_width instead of being dropped silently.
    constexpr int HV = HP >= 4 ? 4 : HP;
    constexpr int TPE = 2 * HP / HV;
    constexpr int EPB = 256 / TPE;
    extern __shared__ float smem[];
    float* cs = smem;
    float* gs = smem + n_slots * HP;
    for (int i = threadIdx.x; i < n_slots * HP; i += blockDim.x) {
        const int sl = i / HP, h = i % HP;
        float c = 0.f;
        if (h < Hreal) c = code[sl * code_stride + h] * (window ? window[h] : 1.0f);
        cs[i] = (float)(half_t)c;
        if (h >= out_width && cs[i] != 0.f && beyond_width) beyond_width[0] = 1.0f;
    }
    const uint64_t n_walk = bucket_entries ? virtual_total : total;
    const uint64_t n_tiles = (n_walk + EPB - 1) / EPB;
    const int le = threadIdx.x / TPE, part = threadIdx.x % TPE, f = part / (TPE / 2), hq = part % (TPE / 2);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t v0 = tile * EPB;                       // first (virtual) entry of the tile = its place in `out`
        const uint64_t e0 = bucket_entries ? (v0 / bucket_entries) * rank_entries + entry_base + (v0 % bucket_entries) : v0;
        __syncthreads();
        constexpr int Q = EPB * 2 / 4;
        for (int i = threadIdx.x; i < n_slots * Q; i += blockDim.x) {
            const int sl = i / Q, j = (i % Q) * 4;
            const uint64_t ge = e0 * 2ull + j;
            const float* src = G + (uint64_t)sl * total * 2ull + ge;
            float4 g4;
            if (ge + 3 < total * 2ull) g4 = *reinterpret_cast<const float4*>(src);
            else g4 = make_float4(ge < total * 2ull ? src[0] : 0.f, ge + 1 < total * 2ull ? src[1] : 0.f,
                                  ge + 2 < total * 2ull ? src[2] : 0.f, 0.f);
            *reinterpret_cast<float4*>(gs + sl * (EPB * 2) + j) = g4;
        }
        __syncthreads();
        const uint64_t e = e0 + le;
        if (v0 + le >= n_walk) continue;
        if (e >= total && !bucket_entries) continue;
        if (hq * HV >= out_width) continue;                   // grids the packed output does not carry
        float g[HV];
#pragma unroll
        for (int k = 0; k < HV; ++k) g[k] = 0.f;
        if (e < total) {
            for (int sl = 0; sl < n_slots; ++sl) {
                const float gv = gs[sl * (EPB * 2) + le * 2 + f];
                if (gv != 0.f) {
#pragma unroll
                    for (int k = 0; k < HV; ++k) g[k] = __fmaf_rn(gv, cs[sl * HP + hq * HV + k], g[k]);
                }
            }
        }
        const uint64_t at = ((v0 + le) * 2ull + f) * (uint64_t)out_width + hq * HV;
#pragma unroll
        for (int k = 0; k < HV; ++k) {
            if (hq * HV + k >= out_width) continue;
            const float prev = accumulate ? (float)out[at + k] : 0.f;
            out[at + k] = (half_t)(prev + scale * g[k]);
        }
    }
}

// Adam on the grids [0, width) of a shard's entries: grad is PACKED [entry][f][width] (what the narrow reduce-scatter
// delivered), master / moments / working tables keep the full [entry][f][HP] layout; the new fp16 values are also written
// packed, for the narrow all-gather.  The arithmetic per element is adam_dense_f16grad_kernel's.
__global__ __launch_bounds__(256) void adam_f16grad_width_kernel(const half_t* __restrict__ grad, int64_t n_packed,
                                                                 int width, int HP, float* __restrict__ master,
                                                                 float* __restrict__ m, float* __restrict__ v,
                                                                 half_t* __restrict__ f16, half_t* __restrict__ packed_out,
                                                                 AdamHyper hy, const float* __restrict__ inv_scale,
                                                                 const float* __restrict__ found_inf) {
    const bool skip = found_inf && found_inf[0] != 0.f;      // GradScaler: nothing moves -- but the all-gather still needs
    const float is = inv_scale ? inv_scale[0] : 1.0f;        // this rank's CURRENT values in the packed buffer
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_packed; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / width;                       // (entry, f)
        const int h = (int)(i - row * width);
        const int64_t at = row * HP + h;
        if (skip) { packed_out[i] = f16[at]; continue; }
        float p = master[at], mm = m[at], vv = v[at];
        adam_update((float)grad[i] * is, p, mm, vv, hy);
        master[at] = p; m[at] = mm; v[at] = vv;
        const half_t ph = (half_t)p;
        f16[at] = ph;
        packed_out[i] = ph;
    }
}

// working tables [entry][f][HP] <- the gathered packed values [entry][f][width] of ALL ranks (rank-major = entry order)
__global__ __launch_bounds__(256) void tables_unpack_width_kernel(const half_t* __restrict__ packed, int64_t n_packed,
                                                                  int width, int HP, half_t* __restrict__ f16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_packed; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / width;
        f16[row * HP + (i - row * width)] = packed[i];
    }
}

// expand_f16_kernel's expansion for the first W < HP grids of a padded width HP >= 8 (the narrow exchange of the schedule's
// first grid and of its ramp).  In expand_f16_kernel only the threads of the first grid quad work then --
// 1 / 8 of a block at HP = 32 -- behind an LDS staging of 32-entry tiles with two barriers each: 0.70 ms for the 1.2 GB of G
// where the bytes take 0.15 ms.  Here a thread owns one entry (both features): its values of every plane are one 8-byte load each,
// coalesced across the wave, nothing is staged, and the 2 W results leave as one vector store.  The arithmetic per element
// is the kernel's above (the planes in ascending order, fmaf, zeros skipped): the same bits.
template <int W, bool CONSUME>
__global__ __launch_bounds__(256) void expand_f16_narrow_kernel(
    float* __restrict__ G, int n_slots, const float* __restrict__ code, int64_t code_stride,
    const float* __restrict__ window, int Hreal, int HP, uint64_t total, half_t* __restrict__ out, float scale,
    int accumulate, uint64_t bucket_entries, uint64_t rank_entries, uint64_t entry_base, uint64_t virtual_total,
    float* __restrict__ beyond_width) {
    extern __shared__ float smem[];
    float* cs = smem;                                         // [n_slots][W]
    for (int i = threadIdx.x; i < n_slots * HP; i += blockDim.x) {
        const int sl = i / HP, h = i % HP;
        float c = 0.f;
        if (h < Hreal) c = code[sl * code_stride + h] * (window ? window[h] : 1.0f);
        c = (float)(half_t)c;
        if (h < W) cs[sl * W + h] = c;
        else if (c != 0.f && beyond_width) beyond_width[0] = 1.0f;
    }
    __syncthreads();
    const uint64_t n_walk = bucket_entries ? virtual_total : total;
    float2* G2 = reinterpret_cast<float2*>(G);
    // CONSUME (nsx_hash_grad_expand_f16_bucket_width_consume): the pairs found non-zero are cleared as they are read -- G is
    // all zeros again once every piece of the exchange has been expanded, and the next backward's scatter needs no 1.2 GB
    // fill in front of it (what nsx_adam_hash_factored_consume does for the single-GPU optimizer pass)
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_walk; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = bucket_entries ? (v / bucket_entries) * rank_entries + entry_base + (v % bucket_entries) : v;
        float g0[W], g1[W];
#pragma unroll
        for (int k = 0; k < W; ++k) { g0[k] = 0.f; g1[k] = 0.f; }
        if (e < total) {
            int sl = 0;
            for (; sl + 4 <= n_slots; sl += 4) {              // four planes in flight
                float2 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = G2[(uint64_t)(sl + u) * total + e];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (CONSUME && (q[u].x != 0.f || q[u].y != 0.f)) G2[(uint64_t)(sl + u) * total + e] = float2{0.f, 0.f};
                    if (q[u].x != 0.f) {
#pragma unroll
                        for (int k = 0; k < W; ++k) g0[k] = __fmaf_rn(q[u].x, cs[(sl + u) * W + k], g0[k]);
                    }
                    if (q[u].y != 0.f) {
#pragma unroll
                        for (int k = 0; k < W; ++k) g1[k] = __fmaf_rn(q[u].y, cs[(sl + u) * W + k], g1[k]);
                    }
                }
            }
            for (; sl < n_slots; ++sl) {
                const float2 q = G2[(uint64_t)sl * total + e];
                if (CONSUME && (q.x != 0.f || q.y != 0.f)) G2[(uint64_t)sl * total + e] = float2{0.f, 0.f};
                if (q.x != 0.f) {
#pragma unroll
                    for (int k = 0; k < W; ++k) g0[k] = __fmaf_rn(q.x, cs[sl * W + k], g0[k]);
                }
                if (q.y != 0.f) {
#pragma unroll
                    for (int k = 0; k < W; ++k) g1[k] = __fmaf_rn(q.y, cs[sl * W + k], g1[k]);
                }
            }
        }
        half_t* dst = out + v * 2ull * W;                     // [entry][f][W]
        __attribute__((aligned(16))) half_t res[2 * W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float p0 = accumulate ? (float)dst[k] : 0.f, p1 = accumulate ? (float)dst[W + k] : 0.f;
            res[k] = (half_t)(p0 + scale * g0[k]);
            res[W + k] = (half_t)(p1 + scale * g1[k]);
        }
        if (W == 1) {
            *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(res);
        } else if (W == 2) {
            *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(res);
        } else {
#pragma unroll
            for (int q = 0; q < W / 4; ++q)
                reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(res)[q];
        }
    }
}

template <int HP>
static int launch_expand_f16(const float* G, int n_slots, const float* code, int64_t code_stride, const float* window,
                             int H, uint64_t total, nsx_half* out, float scale, int accumulate, hipStream_t st,
                             uint64_t bucket_entries = 0, uint64_t rank_entries = 0, uint64_t entry_base = 0,
                             uint64_t virtual_total = 0, int out_width = HP, float* beyond_width = nullptr,
                             bool consume = false) {
    constexpr int HV = HP >= 4 ? 4 : HP;
    constexpr int EPB = 256 / (2 * HP / HV);
    NSX_REQUIRE(bucket_entries % EPB == 0, "nsx_hash_grad_expand_f16_bucket: bucket of %llu entries is not a multiple of "
                "the kernel's tile (%d entries)", (unsigned long long)bucket_entries, EPB);
    if (HP >= 8 && out_width < HP && out_width <= 16 && (reinterpret_cast<uintptr_t>(G) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const size_t smem_n = (size_t)n_slots * out_width * sizeof(float);
        const dim3 grid(num_cus() * 8), block(256);
        float* Gw = const_cast<float*>(G);
#define NSX_EXPN_CASE(WW) case WW: \
        if (consume) hipLaunchKernelGGL((expand_f16_narrow_kernel<WW, true>), grid, block, smem_n, st, Gw, n_slots, code, \
            code_stride, window, H, HP, total, reinterpret_cast<half_t*>(out), scale, accumulate, bucket_entries, rank_entries, \
            entry_base, virtual_total, beyond_width); \
        else hipLaunchKernelGGL((expand_f16_narrow_kernel<WW, false>), grid, block, smem_n, st, Gw, n_slots, code, \
            code_stride, window, H, HP, total, reinterpret_cast<half_t*>(out), scale, accumulate, bucket_entries, rank_entries, \
            entry_base, virtual_total, beyond_width); \
        break;
        switch (out_width) { NSX_EXPN_CASE(1) NSX_EXPN_CASE(2) NSX_EXPN_CASE(4) NSX_EXPN_CASE(8) NSX_EXPN_CASE(16) }
#undef NSX_EXPN_CASE
        NSX_LAUNCH_CHECK("nsx_hash_grad_expand_f16 (narrow) launch");
        return NSX_OK;
    }
    NSX_REQUIRE(!consume, "nsx_hash_grad_expand_f16_bucket_width_consume: only the narrow expansion (padded H >= 8, width <= "
                "16, aligned buffers) clears the planes it reads");
    const size_t smem = ((size_t)n_slots * HP + (size_t)n_slots * EPB * 2) * sizeof(float);
    hipLaunchKernelGGL((expand_f16_kernel<HP>), dim3(num_cus() * 8), dim3(256), smem, st, G, n_slots, code, code_stride,
                       window, H, total, reinterpret_cast<half_t*>(out), scale, accumulate, bucket_entries, rank_entries,
                       entry_base, virtual_total, out_width, beyond_width);
    NSX_LAUNCH_CHECK("nsx_hash_grad_expand_f16 launch");
    return NSX_OK;
}

// ---- the small parameter groups (two fused MLPs, two embeddings, 16 deformation tensors: ~0.27 M parameters) -----------
// torch.optim.Adam(fused=True) per group costs ~0.15 ms of host time per group and step (optimizer hooks, _foreach_add_
// of the step tensors, profiler ranges) plus one _amp_foreach_non_finite_check_and_unscale_ each -- in steady state the
// step is host-bound, so the three groups' unscale + inf check and their Adam updates are ONE launch each over a
// by-value table of tensor references.  blockIdx.y = tensor, blockIdx.x strides over its elements.
struct TensorTable {
    nsx_tensor_ref t[NSX_MAX_TENSORS];
    int n;
};
struct GroupHyper {
    AdamHyper h[NSX_MAX_GROUPS];
};

// GradScaler.unscale_ (torch._amp_foreach_non_finite_check_and_unscale_): found_inf[group] = 1 if any gradient element
// is non-finite (checked on the scaled value), every element multiplied by *inv_scale in place.
__global__ __launch_bounds__(256) void multi_unscale_check_kernel(TensorTable T, const float* __restrict__ inv_scale,
                                                                  float* __restrict__ found_inf) {
    const nsx_tensor_ref r = T.t[blockIdx.y];
    float* g = reinterpret_cast<float*>(r.grad);
    if (!g) return;
    const float is = inv_scale ? inv_scale[0] : 1.0f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !isfinite(v);
        g[i] = is == 1.0f ? v : v * is;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[r.group] = 1.0f;
}

// torch.amp.GradScaler.update (torch._amp_update_scale_ on the sum of the groups' flags) + everything the next step
// derives from the scale, in one single-thread launch: the flags are copied out for the host's deferred read and cleared
// for the next step, 1 / scale is refreshed, and the new scale is written to every place that caches it (the one-hot
// loss gradient the backward starts from).
struct ScaleMirrors {
    float* p[NSX_MAX_SCALE_MIRRORS];
    int n;
};
__global__ void grad_scaler_update_kernel(const float* found_inf, int n_groups, float* __restrict__ scale,
                                          int32_t* __restrict__ growth_tracker, float* inv_scale, float* found_copy,
                                          float* clear_flags, ScaleMirrors M, float growth_factor, float backoff_factor,
                                          int growth_interval, int enabled) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int i = 0; i < n_groups; ++i) {
        const float f = found_inf[i];
        total += f;
        if (found_copy) found_copy[i] = f;
    }
    if (clear_flags) {
        for (int i = 0; i < n_groups; ++i) clear_flags[i] = 0.f;
    }
    float s = scale[0];
    if (enabled) {
        if (total != 0.f) {
            s = s * backoff_factor;
            growth_tracker[0] = 0;
        } else {
            const int successful = growth_tracker[0] + 1;
            if (successful == growth_interval) {
                const float grown = s * growth_factor;
                if (isfinite(grown)) s = grown;
                growth_tracker[0] = 0;
            } else {
                growth_tracker[0] = successful;
            }
        }
        scale[0] = s;
    }
    if (inv_scale) inv_scale[0] = (float)(1.0 / (double)s);
    for (int k = 0; k < M.n; ++k) M.p[k][0] = s;
}

struct PresentIndex {
    int16_t at[NSX_MAX_TENSORS];      // tensor i's element of `present`, -1: always present
};

__global__ __launch_bounds__(256) void multi_adam_kernel(TensorTable T, GroupHyper H, const float* __restrict__ found_inf,
                                                         const float* __restrict__ present, PresentIndex P) {
    const nsx_tensor_ref r = T.t[blockIdx.y];
    const float* g = reinterpret_cast<const float*>(r.grad);
    if (!g) return;                                          // no gradient this step: torch skips the parameter
    if (found_inf && found_inf[r.group] != 0.f) return;      // GradScaler: the whole group's step is skipped
    // data-parallel runs: NO rank had a gradient for this tensor (the all-reduced `gradient` is the zeros the ranks put in to
    // join the collective): a single process leaves such a parameter alone, so does every rank (nsx_multi_adam_present)
    if (present && P.at[blockIdx.y] >= 0 && present[P.at[blockIdx.y]] == 0.f) return;
    AdamHyper hy = H.h[r.group];
    if (r.step > 0) {                                        // the tensor's own step count (torch: state[p]["step"])
        hy.bc1 = (float)(1.0 - pow((double)hy.beta1, (double)r.step));
        hy.bc2_sqrt = (float)sqrt(1.0 - pow((double)hy.beta2, (double)r.step));
    }
    float* p = reinterpret_cast<float*>(r.param);
    float* m = reinterpret_cast<float*>(r.exp_avg);
    float* v = reinterpret_cast<float*>(r.exp_avg_sq);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += (int64_t)gridDim.x * blockDim.x) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_update(g[i], pp, mm, vv, hy);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

static AdamHyper make_hyper(float lr, float beta1, float beta2, float eps, int64_t step) {
    AdamHyper h;
    h.lr = lr; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
    h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    return h;
}

template <int HP, bool CLEAR>
static int launch_adam_factored(float* G, int n_slots, const float* code, int64_t code_stride, const float* window,
                                int H, uint64_t total, float* master, float* m, float* v, nsx_half* f16, AdamHyper hy,
                                const float* inv_scale, const float* found_inf, hipStream_t st) {
    constexpr int HV = HP >= 4 ? 4 : HP;
    constexpr int EPB = 256 / (2 * HP / HV);
    const size_t smem = ((size_t)n_slots * HP + (size_t)n_slots * EPB * 2) * sizeof(float);
    // Blocks per CU.  Alone, 4 (16 waves) is the optimum: 2.1-2.2 ms against 2.4-2.55 ms with 8 on the 12 GB pass -- the seven
    // interleaved streams keep more DRAM pages open with fewer concurrent tiles (tools/adam_bench.py).  Inside the training
    // step the pass runs beside the next step's marching and deformation forward, and what counts is the pair: 5 gives the
    // shortest step (7.92-7.97 / 3.61-3.66 ms per step early / steady against 8.02-8.07 / 3.78-3.79 with 4; 6: 7.87-7.93 /
    // 3.63-3.64; 8 fills every wave slot, the other stream's kernels then simply wait for the pass: 7.93 / 3.73).
    const int per_cu = option(NSX_OPT_ADAM_BLOCKS_PER_CU);
    hipLaunchKernelGGL((adam_hash_factored_kernel<HP, CLEAR>), dim3(num_cus() * per_cu), dim3(256), smem, st, G, n_slots, code,
                       code_stride, window, H, total, master, m, v, reinterpret_cast<half_t*>(f16), hy, inv_scale,
                       found_inf);
    NSX_LAUNCH_CHECK("nsx_adam_hash_factored launch");
    return NSX_OK;
}

template <bool CLEAR>
static int adam_hash_factored_entry(float* G, int n_slots, const float* code_table, int64_t code_stride,
                                    const float* window, int H, const nsx_grid_geom* g, float* master, float* exp_avg,
                                    float* exp_avg_sq, nsx_half* tables_f16, float lr, float beta1, float beta2, float eps,
                                    int64_t step, const float* inv_scale, const float* found_inf, void* stream) {
    NSX_REQUIRE(G && code_table && g && master && exp_avg && exp_avg_sq && tables_f16, "nsx_adam_hash_factored: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_adam_hash_factored: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_ADAM_SLOTS, "nsx_adam_hash_factored: n_slots=%d not in [1,%d]", n_slots,
                NSX_MAX_ADAM_SLOTS);
    NSX_REQUIRE(step >= 1, "nsx_adam_hash_factored: step must be >= 1");
    NSX_REQUIRE((reinterpret_cast<uintptr_t>(G) & 15) == 0, "nsx_adam_hash_factored: G must be 16-byte aligned");
    const uint64_t total = g->offset[g->n_levels];
    const AdamHyper hy = make_hyper(lr, beta1, beta2, eps, step);
    hipStream_t st = (hipStream_t)stream;
    if (n_slots > NSX_MAX_SLOTS && nsx_padded_grids(H) == 32 && total * 2ull * 9ull * 4ull < (1ull << 32)) {
        const int T = (n_slots + 15) / 16;
        const size_t smem = (size_t)T * 16 * 32 * 2 * sizeof(unsigned short);
#define NSX_ADAM_MFMA_CASE(TT) case TT: hipLaunchKernelGGL((adam_hash_factored_mfma_kernel<TT, CLEAR>), \
        dim3(num_cus() * MFMA_ADAM_WAVES(TT)), \
        dim3(256), smem, st, G, n_slots, code_table, code_stride, window, H, total, master, exp_avg, exp_avg_sq, \
        reinterpret_cast<half_t*>(tables_f16), hy, inv_scale, found_inf); break;
        switch (T) {
            NSX_ADAM_MFMA_CASE(5) NSX_ADAM_MFMA_CASE(6) NSX_ADAM_MFMA_CASE(7) NSX_ADAM_MFMA_CASE(8) NSX_ADAM_MFMA_CASE(9)
            NSX_ADAM_MFMA_CASE(10) NSX_ADAM_MFMA_CASE(11) NSX_ADAM_MFMA_CASE(12)
        }
#undef NSX_ADAM_MFMA_CASE
        static_assert(NSX_MAX_ADAM_SLOTS == 12 * 16 && NSX_MAX_SLOTS == 4 * 16, "the K steps instantiated above");
        NSX_LAUNCH_CHECK("nsx_adam_hash_factored (matrix-core expansion) launch");
        return NSX_OK;
    }
#define NSX_ADAM_CASE(HP) case HP: return launch_adam_factored<HP, CLEAR>(G, n_slots, code_table, code_stride, window, H, \
        total, master, exp_avg, exp_avg_sq, tables_f16, hy, inv_scale, found_inf, st);
    switch (nsx_padded_grids(H)) {
        NSX_ADAM_CASE(1) NSX_ADAM_CASE(2) NSX_ADAM_CASE(4) NSX_ADAM_CASE(8) NSX_ADAM_CASE(16) NSX_ADAM_CASE(32)
    }
#undef NSX_ADAM_CASE
    set_error("nsx_adam_hash_factored: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

// The tail of a data-parallel gradient bucket (engine/parallel.py): a few small fp32 arrays -- the gradients that do not live
// in the step's gradient buffer (the two embeddings'), the presence counts, the flags -- copied behind the buffer's fixed part
// in ONE launch, and back (averaged) after the all-reduce.  blockIdx.y = piece.
struct BucketPieces {
    float* p[NSX_MAX_BUCKET_PIECES];
    int64_t n[NSX_MAX_BUCKET_PIECES];
    int64_t at[NSX_MAX_BUCKET_PIECES];     // element offset of the piece behind `flat_at`
};
__global__ __launch_bounds__(256) void bucket_pack_kernel(float* __restrict__ flat_at, BucketPieces P) {
    const int k = blockIdx.y;
    const float* src = P.p[k];
    float* dst = flat_at + P.at[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n[k]; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
// blockIdx.y < n_pieces: piece k = flat_at[at_k ..] * scale; blockIdx.y == n_pieces: flat[0 .. n_scale) *= scale
__global__ __launch_bounds__(256) void bucket_unpack_kernel(float* __restrict__ flat, int64_t n_scale, float scale,
                                                            const float* __restrict__ flat_at, BucketPieces P, int n_pieces) {
    const int k = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    if (k == n_pieces) {
        if (scale != 1.0f)
            for (int64_t i = i0; i < n_scale; i += step) flat[i] *= scale;
        return;
    }
    const float* src = flat_at + P.at[k];
    float* dst = P.p[k];
    for (int64_t i = i0; i < P.n[k]; i += step) dst[i] = src[i] * scale;
}

static int fill_pieces(float* const* pieces_host, const int64_t* sizes_host, int n_pieces, BucketPieces& P, int64_t& most,
                       int64_t& total, const char* who) {
    NSX_REQUIRE(n_pieces >= 0 && n_pieces <= NSX_MAX_BUCKET_PIECES && (n_pieces == 0 || (pieces_host && sizes_host)),
                "%s: %d pieces (limit %d)", who, n_pieces, NSX_MAX_BUCKET_PIECES);
    most = 0; total = 0;
    for (int k = 0; k < NSX_MAX_BUCKET_PIECES; ++k) { P.p[k] = nullptr; P.n[k] = 0; P.at[k] = 0; }
    for (int k = 0; k < n_pieces; ++k) {
        NSX_REQUIRE(sizes_host[k] >= 0 && (sizes_host[k] == 0 || pieces_host[k]), "%s: piece %d: bad size or NULL", who, k);
        P.p[k] = pieces_host[k]; P.n[k] = sizes_host[k]; P.at[k] = total;
        total += sizes_host[k];
        if (sizes_host[k] > most) most = sizes_host[k];
    }
    return NSX_OK;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_check_finite(const float* x, int64_t n, float* found_inf, void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_check_finite: negative size");
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(x && found_inf, "nsx_check_finite: NULL argument");
    hipLaunchKernelGGL(check_finite_kernel, dim3(num_cus() * 8), dim3(256), 0, (hipStream_t)stream, x, n, found_inf);
    NSX_LAUNCH_CHECK("nsx_check_finite launch");
    return NSX_OK;
}

int nsx_adam_hash_factored(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                           const float* window, int H, const nsx_grid_geom* g, float* master, float* exp_avg,
                           float* exp_avg_sq, nsx_half* tables_f16, float lr, float beta1, float beta2, float eps,
                           int64_t step, const float* inv_scale, const float* found_inf, void* stream) {
    return adam_hash_factored_entry<false>(const_cast<float*>(G), n_slots, code_table, code_stride, window, H, g, master,
                                           exp_avg, exp_avg_sq, tables_f16, lr, beta1, beta2, eps, step, inv_scale,
                                           found_inf, stream);
}

int nsx_adam_hash_factored_consume(float* G, int n_slots, const float* code_table, int64_t code_stride,
                                   const float* window, int H, const nsx_grid_geom* g, float* master, float* exp_avg,
                                   float* exp_avg_sq, nsx_half* tables_f16, float lr, float beta1, float beta2, float eps,
                                   int64_t step, const float* inv_scale, const float* found_inf, void* stream) {
    return adam_hash_factored_entry<true>(G, n_slots, code_table, code_stride, window, H, g, master, exp_avg, exp_avg_sq,
                                          tables_f16, lr, beta1, beta2, eps, step, inv_scale, found_inf, stream);
}

int nsx_adam_dense(const float* grad, int64_t n, float* master, float* exp_avg, float* exp_avg_sq,
                   nsx_half* params_f16, float lr, float beta1, float beta2, float eps, int64_t step,
                   const float* inv_scale, const float* found_inf, void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_adam_dense: negative size");
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(grad && master && exp_avg && exp_avg_sq, "nsx_adam_dense: NULL argument");
    NSX_REQUIRE(step >= 1, "nsx_adam_dense: step must be >= 1");
    const AdamHyper hy = make_hyper(lr, beta1, beta2, eps, step);
    hipLaunchKernelGGL(adam_dense_kernel, dim3(num_cus() * 4), dim3(256), 0, (hipStream_t)stream, grad, n, master,
                       exp_avg, exp_avg_sq, reinterpret_cast<half_t*>(params_f16), hy, inv_scale, found_inf);
    NSX_LAUNCH_CHECK("nsx_adam_dense launch");
    return NSX_OK;
}

int nsx_check_finite_f16(const nsx_half* x, int64_t n, float* found_inf, void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_check_finite_f16: negative size");
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(x && found_inf, "nsx_check_finite_f16: NULL argument");
    NSX_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "nsx_check_finite_f16: x must be 16-byte aligned");
    hipLaunchKernelGGL(check_finite_f16_kernel, dim3(num_cus() * 8), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(x), n, found_inf);
    NSX_LAUNCH_CHECK("nsx_check_finite_f16 launch");
    return NSX_OK;
}

int nsx_adam_dense_f16grad(const nsx_half* grad, int64_t n, float* master, float* exp_avg, float* exp_avg_sq,
                           nsx_half* params_f16, float lr, float beta1, float beta2, float eps, int64_t step,
                           const float* inv_scale, const float* found_inf, void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_adam_dense_f16grad: negative size");
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(grad && master && exp_avg && exp_avg_sq, "nsx_adam_dense_f16grad: NULL argument");
    NSX_REQUIRE(step >= 1, "nsx_adam_dense_f16grad: step must be >= 1");
    NSX_REQUIRE(((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(params_f16)) & 7) == 0 &&
                ((reinterpret_cast<uintptr_t>(master) | reinterpret_cast<uintptr_t>(exp_avg) |
                  reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
                "nsx_adam_dense_f16grad: buffers must be 16-byte (fp32) / 8-byte (fp16) aligned");
    const AdamHyper hy = make_hyper(lr, beta1, beta2, eps, step);
    hipLaunchKernelGGL(adam_dense_f16grad_kernel, dim3(num_cus() * 4), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(grad), n, master, exp_avg, exp_avg_sq,
                       reinterpret_cast<half_t*>(params_f16), hy, inv_scale, found_inf);
    NSX_LAUNCH_CHECK("nsx_adam_dense_f16grad launch");
    return NSX_OK;
}

int nsx_hash_grad_expand_f16(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                             const float* window, int H, const nsx_grid_geom* g, nsx_half* dtables_f16, float scale,
                             int accumulate, void* stream) {
    NSX_REQUIRE(G && code_table && dtables_f16 && g, "nsx_hash_grad_expand_f16: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_grad_expand_f16: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_grad_expand_f16: n_slots=%d not in [1,%d]", n_slots,
                NSX_MAX_SLOTS);
    const uint64_t total = g->offset[g->n_levels];
    hipStream_t st = (hipStream_t)stream;
#define NSX_EXP_CASE(HP) case HP: return launch_expand_f16<HP>(G, n_slots, code_table, code_stride, window, H, total, \
        dtables_f16, scale, accumulate, st);
    switch (nsx_padded_grids(H)) {
        NSX_EXP_CASE(1) NSX_EXP_CASE(2) NSX_EXP_CASE(4) NSX_EXP_CASE(8) NSX_EXP_CASE(16) NSX_EXP_CASE(32)
    }
#undef NSX_EXP_CASE
    set_error("nsx_hash_grad_expand_f16: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

static int expand_bucket(const float* G, int n_slots, const float* code_table, int64_t code_stride, const float* window,
                         int H, const nsx_grid_geom* g, nsx_half* bucket_f16, float scale, int accumulate,
                         int64_t shard_elements, int64_t bucket_elements, int64_t bucket_index, int world_size, int width,
                         float* beyond_width, void* stream, bool consume = false);

int nsx_hash_grad_expand_f16_bucket(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                                    const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16, float scale,
                                    int accumulate, int64_t shard_elements, int64_t bucket_elements, int64_t bucket_index,
                                    int world_size, void* stream) {
    return expand_bucket(G, n_slots, code_table, code_stride, window, H, g, bucket_f16, scale, accumulate, shard_elements,
                         bucket_elements, bucket_index, world_size, nsx_padded_grids(H), nullptr, stream);
}

int nsx_hash_grad_expand_f16_bucket_width(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                                          const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16,
                                          float scale, int accumulate, int64_t shard_elements, int64_t bucket_elements,
                                          int64_t bucket_index, int world_size, int width, float* beyond_width,
                                          void* stream) {
    const int Hp = nsx_padded_grids(H);
    NSX_REQUIRE(width >= 1 && width <= Hp && (width & (width - 1)) == 0,
                "nsx_hash_grad_expand_f16_bucket_width: width %d is not a power of two in [1, %d]", width, Hp);
    return expand_bucket(G, n_slots, code_table, code_stride, window, H, g, bucket_f16, scale, accumulate, shard_elements,
                         bucket_elements, bucket_index, world_size, width, beyond_width, stream);
}

int nsx_hash_grad_expand_f16_bucket_width_consume(float* G, int n_slots, const float* code_table, int64_t code_stride,
                                                  const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16,
                                                  float scale, int accumulate, int64_t shard_elements,
                                                  int64_t bucket_elements, int64_t bucket_index, int world_size, int width,
                                                  float* beyond_width, void* stream) {
    const int Hp = nsx_padded_grids(H);
    NSX_REQUIRE(width >= 1 && width <= Hp && (width & (width - 1)) == 0,
                "nsx_hash_grad_expand_f16_bucket_width_consume: width %d is not a power of two in [1, %d]", width, Hp);
    return expand_bucket(G, n_slots, code_table, code_stride, window, H, g, bucket_f16, scale, accumulate, shard_elements,
                         bucket_elements, bucket_index, world_size, width, beyond_width, stream, true);
}

int nsx_adam_dense_f16grad_width(const nsx_half* grad_packed, int64_t n_entries, int width, int H_padded, float* master,
                                 float* exp_avg, float* exp_avg_sq, nsx_half* params_f16, nsx_half* packed_out, float lr,
                                 float beta1, float beta2, float eps, int64_t step, const float* inv_scale,
                                 const float* found_inf, void* stream) {
    NSX_REQUIRE(n_entries >= 0, "nsx_adam_dense_f16grad_width: negative size");
    if (n_entries == 0) return NSX_OK;
    NSX_REQUIRE(grad_packed && master && exp_avg && exp_avg_sq && params_f16 && packed_out,
                "nsx_adam_dense_f16grad_width: NULL argument");
    NSX_REQUIRE(width >= 1 && width <= H_padded && H_padded <= 32, "nsx_adam_dense_f16grad_width: width %d of %d grids",
                width, H_padded);
    NSX_REQUIRE(step >= 1, "nsx_adam_dense_f16grad_width: step must be >= 1");
    const AdamHyper hy = make_hyper(lr, beta1, beta2, eps, step);
    hipLaunchKernelGGL(adam_f16grad_width_kernel, dim3(num_cus() * 4), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(grad_packed), n_entries * 2 * width, width, H_padded, master, exp_avg,
                       exp_avg_sq, reinterpret_cast<half_t*>(params_f16), reinterpret_cast<half_t*>(packed_out), hy,
                       inv_scale, found_inf);
    NSX_LAUNCH_CHECK("nsx_adam_dense_f16grad_width launch");
    return NSX_OK;
}

int nsx_tables_unpack_width(const nsx_half* packed, int64_t n_entries, int width, int H_padded, nsx_half* tables_f16,
                            void* stream) {
    NSX_REQUIRE(n_entries >= 0, "nsx_tables_unpack_width: negative size");
    if (n_entries == 0) return NSX_OK;
    NSX_REQUIRE(packed && tables_f16, "nsx_tables_unpack_width: NULL argument");
    NSX_REQUIRE(width >= 1 && width <= H_padded && H_padded <= 32, "nsx_tables_unpack_width: width %d of %d grids", width,
                H_padded);
    hipLaunchKernelGGL(tables_unpack_width_kernel, dim3(num_cus() * 4), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(packed), n_entries * 2 * width, width, H_padded,
                       reinterpret_cast<half_t*>(tables_f16));
    NSX_LAUNCH_CHECK("nsx_tables_unpack_width launch");
    return NSX_OK;
}

static int expand_bucket(const float* G, int n_slots, const float* code_table, int64_t code_stride, const float* window,
                         int H, const nsx_grid_geom* g, nsx_half* bucket_f16, float scale, int accumulate,
                         int64_t shard_elements, int64_t bucket_elements, int64_t bucket_index, int world_size, int width,
                         float* beyond_width, void* stream, bool consume) {
    NSX_REQUIRE(G && code_table && bucket_f16 && g, "nsx_hash_grad_expand_f16_bucket: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_grad_expand_f16_bucket: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_grad_expand_f16_bucket: n_slots=%d not in [1,%d]",
                n_slots, NSX_MAX_SLOTS);
    const int Hp = nsx_padded_grids(H);
    const int64_t per_entry = 2ll * Hp;
    NSX_REQUIRE(world_size >= 1 && bucket_elements >= per_entry && shard_elements >= bucket_elements &&
                shard_elements % bucket_elements == 0 && bucket_elements % per_entry == 0 && bucket_index >= 0 &&
                (bucket_index + 1) * bucket_elements <= shard_elements,
                "nsx_hash_grad_expand_f16_bucket: shard %lld / bucket %lld / index %lld / world %d",
                (long long)shard_elements, (long long)bucket_elements, (long long)bucket_index, world_size);
    const uint64_t total = g->offset[g->n_levels];
    const uint64_t be = (uint64_t)(bucket_elements / per_entry), re = (uint64_t)(shard_elements / per_entry);
    hipStream_t st = (hipStream_t)stream;
#define NSX_EXPB_CASE(HP) case HP: return launch_expand_f16<HP>(G, n_slots, code_table, code_stride, window, H, total, \
        bucket_f16, scale, accumulate, st, be, re, (uint64_t)bucket_index * be, (uint64_t)world_size * be, width, beyond_width, \
        consume);
    switch (Hp) {
        NSX_EXPB_CASE(1) NSX_EXPB_CASE(2) NSX_EXPB_CASE(4) NSX_EXPB_CASE(8) NSX_EXPB_CASE(16) NSX_EXPB_CASE(32)
    }
#undef NSX_EXPB_CASE
    set_error("nsx_hash_grad_expand_f16_bucket: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

static int fill_table(const nsx_tensor_ref* tensors, int n_tensors, int n_groups, TensorTable& T, int64_t& n_max,
                      const char* who) {
    NSX_REQUIRE(tensors && n_tensors >= 1 && n_tensors <= NSX_MAX_TENSORS, "%s: n_tensors=%d not in [1,%d]", who, n_tensors,
                NSX_MAX_TENSORS);
    NSX_REQUIRE(n_groups >= 1 && n_groups <= NSX_MAX_GROUPS, "%s: n_groups=%d not in [1,%d]", who, n_groups, NSX_MAX_GROUPS);
    T.n = n_tensors;
    n_max = 0;
    for (int i = 0; i < n_tensors; ++i) {
        NSX_REQUIRE(tensors[i].n >= 0 && tensors[i].group >= 0 && tensors[i].group < n_groups,
                    "%s: tensor %d: bad size or group", who, i);
        NSX_REQUIRE(!tensors[i].grad || tensors[i].n == 0 || tensors[i].param, "%s: tensor %d: NULL parameter", who, i);
        T.t[i] = tensors[i];
        if (tensors[i].grad && tensors[i].n > n_max) n_max = tensors[i].n;
    }
    return NSX_OK;
}

int nsx_multi_unscale_check(const nsx_tensor_ref* tensors_host, int n_tensors, int n_groups, const float* inv_scale,
                            float* found_inf, void* stream) {
    NSX_REQUIRE(found_inf, "nsx_multi_unscale_check: NULL found_inf");
    TensorTable T;
    int64_t n_max = 0;
    if (int rc = fill_table(tensors_host, n_tensors, n_groups, T, n_max, "nsx_multi_unscale_check")) return rc;
    if (n_max == 0) return NSX_OK;
    int64_t bx = (n_max + 1023) / 1024;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(multi_unscale_check_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream,
                       T, inv_scale, found_inf);
    NSX_LAUNCH_CHECK("nsx_multi_unscale_check launch");
    return NSX_OK;
}

int nsx_multi_adam(const nsx_tensor_ref* tensors_host, int n_tensors, const nsx_adam_group* groups_host, int n_groups,
                   const float* found_inf, void* stream) {
    return nsx_multi_adam_present(tensors_host, n_tensors, groups_host, n_groups, found_inf, nullptr, nullptr, 0, stream);
}

int nsx_multi_adam_present(const nsx_tensor_ref* tensors_host, int n_tensors, const nsx_adam_group* groups_host, int n_groups,
                           const float* found_inf, const float* present, const int32_t* present_index_host, int n_present,
                           void* stream) {
    NSX_REQUIRE(groups_host, "nsx_multi_adam: NULL groups");
    NSX_REQUIRE(!present || (present_index_host && n_present >= 1 && n_present <= 32767),
                "nsx_multi_adam_present: a presence vector needs the tensors' indices into it (n_present=%d)", n_present);
    PresentIndex P;
    for (int i = 0; i < NSX_MAX_TENSORS; ++i) P.at[i] = -1;
    if (present) {
        for (int i = 0; i < n_tensors && i < NSX_MAX_TENSORS; ++i) {
            NSX_REQUIRE(present_index_host[i] >= -1 && present_index_host[i] < n_present,
                        "nsx_multi_adam_present: tensor %d: index %d outside the presence vector [0,%d)", i,
                        present_index_host[i], n_present);
            P.at[i] = (int16_t)present_index_host[i];
        }
    }
    TensorTable T;
    int64_t n_max = 0;
    if (int rc = fill_table(tensors_host, n_tensors, n_groups, T, n_max, "nsx_multi_adam")) return rc;
    if (n_max == 0) return NSX_OK;
    GroupHyper H;
    for (int k = 0; k < n_groups; ++k) {
        NSX_REQUIRE(groups_host[k].step >= 1, "nsx_multi_adam: group %d: step must be >= 1", k);
        H.h[k] = make_hyper(groups_host[k].lr, groups_host[k].beta1, groups_host[k].beta2, groups_host[k].eps,
                            groups_host[k].step);
    }
    for (int i = 0; i < n_tensors; ++i)
        NSX_REQUIRE(!tensors_host[i].grad || tensors_host[i].n == 0 || (tensors_host[i].exp_avg && tensors_host[i].exp_avg_sq),
                    "nsx_multi_adam: tensor %d: NULL moments", i);
    int64_t bx = (n_max + 1023) / 1024;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(multi_adam_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, T, H,
                       found_inf, present, P);
    NSX_LAUNCH_CHECK("nsx_multi_adam launch");
    return NSX_OK;
}

int nsx_bucket_pack(float* flat_at, const float* const* pieces_host, const int64_t* sizes_host, int n_pieces, void* stream) {
    BucketPieces P;
    int64_t most, total;
    if (int rc = fill_pieces(const_cast<float* const*>(reinterpret_cast<const float* const*>(pieces_host)), sizes_host, n_pieces, P,
                             most, total, "nsx_bucket_pack"))
        return rc;
    if (total == 0) return NSX_OK;
    NSX_REQUIRE(flat_at, "nsx_bucket_pack: NULL destination");
    int64_t bx = (most + 1023) / 1024;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(bucket_pack_kernel, dim3((unsigned)bx, (unsigned)n_pieces), dim3(256), 0, (hipStream_t)stream, flat_at, P);
    NSX_LAUNCH_CHECK("nsx_bucket_pack launch");
    return NSX_OK;
}

int nsx_bucket_unpack(float* flat, int64_t n_scale, float scale, const float* flat_at, float* const* pieces_host,
                      const int64_t* sizes_host, int n_pieces, void* stream) {
    BucketPieces P;
    int64_t most, total;
    if (int rc = fill_pieces(pieces_host, sizes_host, n_pieces, P, most, total, "nsx_bucket_unpack")) return rc;
    NSX_REQUIRE(n_scale >= 0 && (n_scale == 0 || flat) && (total == 0 || flat_at), "nsx_bucket_unpack: NULL buffer");
    if (n_scale > most) most = n_scale;
    if (most == 0) return NSX_OK;
    int64_t bx = (most + 1023) / 1024;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(bucket_unpack_kernel, dim3((unsigned)bx, (unsigned)(n_pieces + 1)), dim3(256), 0, (hipStream_t)stream, flat,
                       n_scale, scale, flat_at, P, n_pieces);
    NSX_LAUNCH_CHECK("nsx_bucket_unpack launch");
    return NSX_OK;
}

int nsx_grad_scaler_update(const float* found_inf, int n_groups, float* scale, int32_t* growth_tracker, float* inv_scale,
                           float* found_copy, float* clear_flags, float* const* scale_mirrors_host, int n_mirrors,
                           float growth_factor, float backoff_factor, int growth_interval, int enabled, void* stream) {
    NSX_REQUIRE(found_inf && scale && growth_tracker, "nsx_grad_scaler_update: NULL argument");
    NSX_REQUIRE(n_groups >= 1 && n_groups <= NSX_MAX_GROUPS, "nsx_grad_scaler_update: n_groups=%d not in [1,%d]", n_groups,
                NSX_MAX_GROUPS);
    NSX_REQUIRE(n_mirrors >= 0 && n_mirrors <= NSX_MAX_SCALE_MIRRORS && (n_mirrors == 0 || scale_mirrors_host),
                "nsx_grad_scaler_update: n_mirrors=%d not in [0,%d]", n_mirrors, NSX_MAX_SCALE_MIRRORS);
    NSX_REQUIRE(growth_interval >= 1, "nsx_grad_scaler_update: growth_interval must be >= 1");
    ScaleMirrors M;
    M.n = n_mirrors;
    for (int k = 0; k < NSX_MAX_SCALE_MIRRORS; ++k) M.p[k] = k < n_mirrors ? scale_mirrors_host[k] : nullptr;
    for (int k = 0; k < n_mirrors; ++k) NSX_REQUIRE(M.p[k], "nsx_grad_scaler_update: mirror %d is NULL", k);
    hipLaunchKernelGGL(grad_scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, found_inf, n_groups, scale,
                       growth_tracker, inv_scale, found_copy, clear_flags, M, growth_factor, backoff_factor, growth_interval,
                       enabled);
    NSX_LAUNCH_CHECK("nsx_grad_scaler_update launch");
    return NSX_OK;
}

}  // extern "C"

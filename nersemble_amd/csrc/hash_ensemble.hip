// hash_ensemble.hip -- fused HashEnsemble forward / backward for gfx950 (CDNA4).
//
// Replaces, as ONE kernel each way, what the reference does in HashEnsemble.forward
// (src/nersemble/nerfstudio/field_components/hash_ensemble.py:93-158): C separate tcnn HashGrid
// launches (:102-104), torch.stack (:106), einops rearrange (:112), grid window (:133-138) and the
// blend einsum 'bdh,bh->bd' (:155-156) -- without ever materialising the [B, 32, H] product.
//
// MI355X design
//   * native table layout [entry][f][h] fp16: all H grids of one entry are contiguous, 4*H bytes
//     (H=32: exactly one 128-B line per trilinear corner).  All C tcnn encodings share geometry
//     (hash_ensemble.py:84-85), so the corner index is computed once and shared by every grid.
//   * lane mapping: LPE = H/4 lanes cover one entry with one 16-B load each, so a wave-instruction
//     fetches 64/LPE whole entries (H=32: 8 full 128-B lines, fully coalesced per line).
//     lane = (sample s, chunk q); the wave walks the 8 corners with 8 independent loads in flight,
//     accumulates w_k * <features, code> with v_dot2c_f32_f16 (fp32 accumulate), then reduces the
//     LPE/2 lanes of each feature with cross-lane xor shuffles.
//   * the per-sample code row is read once per sample (optionally through an index into the [T][H]
//     time embedding -- no [B][H] gather is materialised), rounded to fp16 like the reference does
//     (hash_ensemble.py:155) and kept packed in registers.
//   * per-level results are staged through a tiny LDS tile so the [B][32] fp16 output is written
//     with coalesced dword stores.
// Roofline: HBM-bound; algorithmic bytes per sample = 512*H (8 corners x 16 levels x 4H B) + 80.
#include "nsx_common.h"

namespace nsx {

template <int H>
struct EnsCfg {
    static_assert(H == 1 || H == 2 || H == 4 || H == 8 || H == 16 || H == 32, "padded grid count");
    static constexpr int ROW_BYTES = 4 * H;                          // 2 features x H grids x fp16
    static constexpr int LANE_BYTES = ROW_BYTES < 16 ? ROW_BYTES : 16;
    static constexpr int NDW = LANE_BYTES / 4;                       // dwords per lane load
    static constexpr int LPE = ROW_BYTES / LANE_BYTES;               // lanes per entry
    static constexpr int SPW = kWave / LPE;                          // samples per wave step
    static constexpr int DPF = H >= 2 ? H / 2 : 1;                   // dwords per feature plane
};

template <int NDW> struct LaneVec;
template <> struct LaneVec<1> { uint32_t d[1]; };
template <> struct LaneVec<2> { uint32_t d[2]; };
template <> struct LaneVec<4> { uint32_t d[4]; };

template <int NDW>
__device__ __forceinline__ LaneVec<NDW> load_lane(const uint8_t* p) {
    LaneVec<NDW> r;
    if constexpr (NDW == 4) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w;
    } else if constexpr (NDW == 2) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        r.d[0] = v.x; r.d[1] = v.y;
    } else {
        r.d[0] = *reinterpret_cast<const uint32_t*>(p);
    }
    return r;
}

struct Cell {
    uint32_t cx, cy, cz;
    float wx, wy, wz;
};

__device__ __forceinline__ Cell cell_of(float scale, float px, float py, float pz) {
    Cell c;
    float p = __fmaf_rn(scale, px, 0.5f); float f = floorf(p); c.cx = (uint32_t)(int32_t)f; c.wx = p - f;
    p = __fmaf_rn(scale, py, 0.5f); f = floorf(p); c.cy = (uint32_t)(int32_t)f; c.wy = p - f;
    p = __fmaf_rn(scale, pz, 0.5f); f = floorf(p); c.cz = (uint32_t)(int32_t)f; c.wz = p - f;
    return c;
}

// Level-local entry indices of the 8 corners (k bit0 = x, bit1 = y, bit2 = z).
__device__ __forceinline__ void corner_indices(const Cell& c, uint32_t res, uint32_t size, bool hashed,
                                               uint32_t idx[8]) {
    if (hashed) {
        const uint32_t mask = size - 1u;
        const uint32_t y0 = c.cy * 2654435761u, y1 = (c.cy + 1u) * 2654435761u;
        const uint32_t z0 = c.cz * 805459861u, z1 = (c.cz + 1u) * 805459861u;
        const uint32_t x0 = c.cx, x1 = c.cx + 1u;
        idx[0] = (x0 ^ y0 ^ z0) & mask; idx[1] = (x1 ^ y0 ^ z0) & mask;
        idx[2] = (x0 ^ y1 ^ z0) & mask; idx[3] = (x1 ^ y1 ^ z0) & mask;
        idx[4] = (x0 ^ y0 ^ z1) & mask; idx[5] = (x1 ^ y0 ^ z1) & mask;
        idx[6] = (x0 ^ y1 ^ z1) & mask; idx[7] = (x1 ^ y1 ^ z1) & mask;
    } else if (__builtin_amdgcn_ballot_w64((c.cx >= res) | (c.cy >= res) | (c.cz >= res)) == 0) {
        // Every lane of the wave is IN CONTRACT (positions in [0, 1): cell coordinates <= res - 1).  Then each corner
        // index is at most res + res^2 + res^3 < 2 res^3 <= 2 size, so `% size` is one conditional subtraction --
        // min(i, i - size) on unsigned values -- and the products fit 24 bits (res <= 81 on a dense level of 2^19
        // entries, <= 645 at the 2^28 cap: nsx_grid_geometry makes a level dense only if res^3 <= size).  The generic
        // path below costs ~180 VALU instructions per level and lane, this one ~40; the field zeroes samples outside
        // the scene box before they get here (nersemble_nerfacto_field.py:268-269), so it is the path that runs.
        const uint32_t r2 = res * res;
        const uint32_t y0 = __umul24(c.cy, res), y1 = y0 + res;
        const uint32_t z0 = __umul24(c.cz, r2), z1 = z0 + r2;
        const uint32_t x0 = c.cx, x1 = c.cx + 1u;
        auto wrap = [size](uint32_t i) { const uint32_t j = i - size; return j < i ? j : i; };
        idx[0] = wrap(x0 + y0 + z0); idx[1] = wrap(x1 + y0 + z0);
        idx[2] = wrap(x0 + y1 + z0); idx[3] = wrap(x1 + y1 + z0);
        idx[4] = wrap(x0 + y0 + z1); idx[5] = wrap(x1 + y0 + z1);
        idx[6] = wrap(x0 + y1 + z1); idx[7] = wrap(x1 + y1 + z1);
    } else {
        // out-of-contract coordinates (negative / beyond the box: uint32 wrap-around): exact idx % size for any idx
        const float inv = 1.0f / (float)size;
        const uint32_t r2 = res * res;
        const uint32_t y0 = c.cy * res, y1 = y0 + res;
        const uint32_t z0 = c.cz * r2, z1 = z0 + r2;
        const uint32_t x0 = c.cx, x1 = c.cx + 1u;
        idx[0] = umod(x0 + y0 + z0, size, inv); idx[1] = umod(x1 + y0 + z0, size, inv);
        idx[2] = umod(x0 + y1 + z0, size, inv); idx[3] = umod(x1 + y1 + z0, size, inv);
        idx[4] = umod(x0 + y0 + z1, size, inv); idx[5] = umod(x1 + y0 + z1, size, inv);
        idx[6] = umod(x0 + y1 + z1, size, inv); idx[7] = umod(x1 + y1 + z1, size, inv);
    }
}

__device__ __forceinline__ void corner_weights(const Cell& c, float w[8]) {
    const float ax = 1.0f - c.wx, ay = 1.0f - c.wy, az = 1.0f - c.wz;
    const float yz00 = ay * az, yz10 = c.wy * az, yz01 = ay * c.wz, yz11 = c.wy * c.wz;
    w[0] = ax * yz00; w[1] = c.wx * yz00; w[2] = ax * yz10; w[3] = c.wx * yz10;
    w[4] = ax * yz01; w[5] = c.wx * yz01; w[6] = ax * yz11; w[7] = c.wx * yz11;
}

// Packed fp16 code for this lane's dwords (code * window, rounded to fp16 once).
template <int H>
__device__ __forceinline__ void load_code(const float* __restrict__ row, const float* __restrict__ window,
                                          int Hreal, int q, half2_t cw[EnsCfg<H>::NDW]) {
    using C = EnsCfg<H>;
    if constexpr (H == 1) {
        float c0 = row[0] * (window ? window[0] : 1.0f);
        cw[0] = half2_t{(half_t)c0, (half_t)c0};
    } else {
#pragma unroll
        for (int j = 0; j < C::NDW; ++j) {
            const int pair = (q * C::NDW + j) % C::DPF;
            const int h0 = 2 * pair, h1 = h0 + 1;
            float c0 = 0.f, c1 = 0.f;
            if (h0 < Hreal) c0 = row[h0] * (window ? window[h0] : 1.0f);
            if (h1 < Hreal) c1 = row[h1] * (window ? window[h1] : 1.0f);
            cw[j] = half2_t{(half_t)c0, (half_t)c1};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int H, int WAVES>
__device__ __forceinline__ void ens_fwd_body(
    const float* __restrict__ x, int64_t B, const uint8_t* __restrict__ tab, const nsx_grid_geom& g,
    const float* __restrict__ code, int64_t code_stride, const int32_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, uint32_t* __restrict__ out, int64_t n_tiles,
    const int64_t* __restrict__ n_dev) {
    using C = EnsCfg<H>;
    NSX_DEVICE_COUNT(B, n_tiles, C::SPW, n_dev);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // [WAVES][SPW][L + 4] dwords
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int L = g.n_levels;
    const int row_dw = L + 4;                       // +16 B pad: conflict-free 8-row column writes
    uint32_t* stage = smem + (size_t)wave * C::SPW * row_dw;
    half_t* stage_h = reinterpret_cast<half_t*>(stage);
    const int s = lane / C::LPE, q = lane % C::LPE;
    const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t wave_count = (int64_t)gridDim.x * WAVES;

    for (int64_t tile = wave_global; tile < n_tiles; tile += wave_count) {
        const int64_t b_raw = tile * C::SPW + s;
        const int64_t b = b_raw < B ? b_raw : B - 1;
        const float px = x[b * 3 + 0], py = x[b * 3 + 1], pz = x[b * 3 + 2];
        half2_t cw[C::NDW];
        {
            const int64_t r = code_index ? (int64_t)code_index[b] : b;
            load_code<H>(code + r * code_stride, window, Hreal, q, cw);
        }
        for (int l = 0; l < L; ++l) {
            const float scale = g.scale[l];
            const uint32_t res = g.res[l], size = g.size[l], off = g.offset[l];
            const bool hashed = g.hashed[l] != 0;
            const Cell c = cell_of(scale, px, py, pz);
            uint32_t idx[8];
            float w[8];
            corner_indices(c, res, size, hashed, idx);
            corner_weights(c, w);
            LaneVec<C::NDW> v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint8_t* p = tab + (size_t)(off + idx[k]) * C::ROW_BYTES + q * C::LANE_BYTES;
                v[k] = load_lane<C::NDW>(p);
            }
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (H == 1) {
                    const half2_t t = as_half2(v[k].d[0]);
                    acc0 = __fmaf_rn(w[k], (float)t.x * (float)cw[0].x, acc0);
                    acc1 = __fmaf_rn(w[k], (float)t.y * (float)cw[0].x, acc1);
                } else if constexpr (C::NDW > C::DPF) {     // H = 2, 4: both features in one lane
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int j = 0; j < C::DPF; ++j) {
                        t0 = dot2(v[k].d[j], cw[j], t0);
                        t1 = dot2(v[k].d[C::DPF + j], cw[C::DPF + j], t1);
                    }
                    acc0 = __fmaf_rn(w[k], t0, acc0);
                    acc1 = __fmaf_rn(w[k], t1, acc1);
                } else {                                    // H >= 8: one feature per lane
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < C::NDW; ++j) t = dot2(v[k].d[j], cw[j], t);
                    acc0 = __fmaf_rn(w[k], t, acc0);
                }
            }
            if constexpr (C::LPE >= 2) {
                // lanes [0, LPE/2) of a sample hold feature 0 partials, [LPE/2, LPE) feature 1
#pragma unroll
                for (int m = 1; m < C::LPE / 2; m <<= 1) acc0 += __shfl_xor(acc0, m);
                if ((q & (C::LPE / 2 - 1)) == 0) {
                    const int f = q / (C::LPE / 2);
                    stage_h[(s * row_dw + l) * 2 + f] = (half_t)acc0;
                }
            } else {
                stage[s * row_dw + l] = as_u32(half2_t{(half_t)acc0, (half_t)acc1});
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // coalesced dword stores of the [SPW][L] tile
        const int64_t base_row = tile * C::SPW;
        for (int i = lane; i < C::SPW * L; i += kWave) {
            const int r = i / L, cdw = i - r * L;
            if (base_row + r < B) out[(base_row + r) * L + cdw] = stage[r * row_dw + cdw];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int H, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void ens_fwd_kernel(
    const float* __restrict__ x, int64_t B, const uint8_t* __restrict__ tab, const nsx_grid_geom g,
    const float* __restrict__ code, int64_t code_stride, const int32_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, uint32_t* __restrict__ out, int64_t n_tiles,
    const int64_t* __restrict__ n_dev) {
    ens_fwd_body<H, WAVES>(x, B, tab, g, code, code_stride, code_index, window, Hreal, out, n_tiles, n_dev);
}

// Several sample sets ("sources": the ranks of a level-parallel job, level_parallel.hip) in ONE launch: blockIdx.y is the
// source, whose arrays sit at fixed byte strides from source 0's; a block works on one source only, so nothing about
// the tile, the code slots or the duplicate-merging keys changes.  One source's ~90 k samples on two levels are one or
// two tiles per wave -- a launch of its own is mostly its own latency.
template <int H, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void ens_fwd_sources_kernel(
    const uint8_t* __restrict__ x, const uint8_t* __restrict__ tab, const nsx_grid_geom g,
    const uint8_t* __restrict__ code, int64_t code_stride, const uint8_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, uint8_t* __restrict__ out, const uint8_t* __restrict__ n_dev,
    const EnsSources src) {
    using C = EnsCfg<H>;
    const int j = blockIdx.y;
    const int64_t B = src.B[j];
    ens_fwd_body<H, WAVES>(reinterpret_cast<const float*>(x + j * src.x_stride), B, tab, g,
                           reinterpret_cast<const float*>(code + j * src.code_stride), code_stride,
                           reinterpret_cast<const int32_t*>(code_index + j * src.slot_stride), window, Hreal,
                           reinterpret_cast<uint32_t*>(out + j * src.out_stride), (B + C::SPW - 1) / C::SPW,
                           reinterpret_cast<const int64_t*>(n_dev + j * src.count_stride));
}

// DPP row_shl:J -- lane i receives the value of lane i+J of its 16-lane row (0 when out of the row)
template <int J>
__device__ __forceinline__ int dpp_row_shl(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x100 + J, 0xf, 0xf, true); }

// Sum of `val` over the run of equal keys that starts at this lane (runs live inside aligned 8-lane groups), in three
// doubling steps: equal keys are CONTIGUOUS inside a run, so after the step with
// distance d every lane holds the sum of its run over [lane, lane + 2d) -- a segmented suffix scan (lane i adds lane
// i + d's partial sum iff that lane is in the same 8-group and carries the same key).  Same terms, tree order.  The
// three `same` predicates depend on the keys only: a caller that merges several values under one key computes them once.
struct RunLinks { bool same1, same2, same4; };
__device__ __forceinline__ RunLinks run_links(uint32_t key, int s7) {
    // same1: the next lane continues this lane's run; same2 / same4: the run is unbroken over the next 2 / 4 lanes.
    // Derived from ADJACENT equality only -- keys A B A must not link lanes 0 and 2 -- by the same doubling:
    // unbroken over [i, i+2] = same1[i] & same1[i+1]; over [i, i+4] = same2[i] & same2[i+2].  same1 is false on the last
    // lane of an 8-group, which also stops every longer link at the group boundary.
    // Every DPP read is a statement of its own, executed by ALL lanes: inside the right-hand side of a `&&` it would
    // run under the narrowed EXEC mask of the left-hand side (a disabled source lane reads as 0) -- or be folded with it.
    const uint32_t next_key = (uint32_t)dpp_row_shl<1>((int)key);
    const int s1 = (int)((s7 + 1 < 8) & (next_key == key));
    const int s1_next = dpp_row_shl<1>(s1);
    const int s2 = s1 & s1_next;
    const int s2_next2 = dpp_row_shl<2>(s2);
    const int s4 = s2 & s2_next2;
    RunLinks r;
    r.same1 = s1 != 0; r.same2 = s2 != 0; r.same4 = s4 != 0;
    return r;
}
__device__ __forceinline__ float run_sum(float val, const RunLinks& r) {
    float sum = val;
    float nv = __builtin_bit_cast(float, dpp_row_shl<1>(__builtin_bit_cast(int, sum)));
    sum += r.same1 ? nv : 0.f;
    nv = __builtin_bit_cast(float, dpp_row_shl<2>(__builtin_bit_cast(int, sum)));
    sum += r.same2 ? nv : 0.f;
    nv = __builtin_bit_cast(float, dpp_row_shl<4>(__builtin_bit_cast(int, sum)));
    sum += r.same4 ? nv : 0.f;
    return sum;
}

// ------------------------------------------------------------------------------------------------
// backward: table gradient (fp32 atomics into the native layout), code gradient, position gradient
// ------------------------------------------------------------------------------------------------
// FACTORED: the per-sample code is row code_index[b] of a SMALL table (n_slots rows), so
//   dL/dtable[e][f][h] = sum_slot G[e][slot][f] * code'[slot][h],  G[e][slot][f] = sum_{b in slot} w * dout_f.
// The kernel then scatters only 2 scalars per (sample, level, corner) into G (one corner per lane) instead of
// 2*H values; nsx_hash_grad_expand turns G into the table gradient with a tiny dense contraction.
// DCODE = false: the caller wants no code gradient (e.g. while the coarse-to-fine window pins the code to ones,
// hash_ensemble.py:114-115): the per-grid accumulation -- a fifth of the kernel's VALU work -- is compiled out.
// MODE: BWD_DENSE (2H atomics per corner into the table gradient), BWD_FACTORED (gather + G scatter in one kernel),
// BWD_GATHER (dL/dcode and dL/dx only: the table gradient is scattered by ens_scatter_kernel on another stream).
constexpr int BWD_DENSE = 0, BWD_FACTORED = 1, BWD_GATHER = 2;
template <int H, int WAVES, int MODE, bool DCODE>
__device__ __forceinline__ void ens_bwd_body(
    const float* __restrict__ x, int64_t B, const uint8_t* __restrict__ tab, const nsx_grid_geom& g,
    const float* __restrict__ code, int64_t code_stride, const int32_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, const float* __restrict__ dout,
    float* __restrict__ dtab, float* __restrict__ dcode, float* __restrict__ dx, int64_t n_tiles,
    int n_slots, float* __restrict__ nonfinite, const int64_t* __restrict__ n_dev,
    float* __restrict__ csum_part) {
    using C = EnsCfg<H>;
    // csum_part (DCODE only): the code gradient leaves the kernel summed per code row -- [gridDim.x][n_slots][H] block
    // partials, accumulated in LDS (code_sums_reduce_kernel adds the blocks and applies the window) -- instead of as a
    // [B][H] tensor that torch then index_add_s into n_slots rows.  Every block writes its partial, also an idle one.
    // LDS: [n_slots][H], only when csum_part.  H = 32 (SPW = 8): one table PER WAVE -- the head lanes of one wave
    // instruction then all have distinct (row, grid) addresses (one 8-sample group per grid quarter), so they add with
    // plain LDS accesses instead of ds_add_f32, which costs ~150 cycles per wave instruction (csrc/mlp.hip's tail).
    extern __shared__ float csum[];
    constexpr bool kWaveTables = (C::SPW == 8);
    const bool code_sums = DCODE && csum_part != nullptr;
    const int n_tables = kWaveTables ? WAVES : 1;
    if (n_dev) {
        const int64_t n__ = *n_dev;
        if (n__ < B) { B = n__ < 0 ? 0 : n__; n_tiles = (B + C::SPW - 1) / C::SPW; }
    }
    if (B <= 0) {
        if (!code_sums) return;
        n_tiles = 0;
    }
    if (code_sums) {
        for (int i = threadIdx.x; i < n_tables * n_slots * H; i += WAVES * kWave) csum[i] = 0.f;
        __syncthreads();
    }
    bool bad = false;                 // a non-finite value was added to the factored gradient
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int L = g.n_levels;
    // backward lane mapping: lane = q * SPW + s (chunk-major) so the samples of a tile sit in ADJACENT lanes --
    // consecutive samples of a ray share coarse cells, and duplicates are merged with DPP before the atomics
    const int s = lane % C::SPW, q = lane / C::SPW;
    const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t wave_count = (int64_t)gridDim.x * WAVES;
    constexpr int NH = 2 * C::NDW;   // fp16 values per lane load
    const size_t g_total = g.offset[L];

    for (int64_t tile = wave_global; tile < n_tiles; tile += wave_count) {
        const int64_t b_raw = tile * C::SPW + s;
        const bool valid = b_raw < B;
        const int64_t b = valid ? b_raw : B - 1;
        const float px = x[b * 3 + 0], py = x[b * 3 + 1], pz = x[b * 3 + 2];
        half2_t cw[C::NDW];
        const int64_t crow = code_index ? (int64_t)code_index[b] : b;
        load_code<H>(code + crow * code_stride, window, Hreal, q, cw);
        float dc[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) dc[i] = 0.f;
        float dxa = 0.f, dya = 0.f, dza = 0.f;
        const float* drow = dout + b * (2 * L);

        for (int l = 0; l < L; ++l) {
            const float scale = g.scale[l];
            const uint32_t res = g.res[l], size = g.size[l], off = g.offset[l];
            const bool hashed = g.hashed[l] != 0;
            float g0 = drow[2 * l], g1 = drow[2 * l + 1];
            if (!valid) { g0 = 0.f; g1 = 0.f; }
            const Cell c = cell_of(scale, px, py, pz);
            uint32_t idx[8];
            float w[8];
            corner_indices(c, res, size, hashed, idx);
            corner_weights(c, w);
            LaneVec<C::NDW> v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint8_t* p = tab + (size_t)(off + idx[k]) * C::ROW_BYTES + q * C::LANE_BYTES;
                v[k] = load_lane<C::NDW>(p);
            }
            // per-lane feature selector for H >= 8 (one feature plane per lane)
            const float gl = (C::LPE >= 2) ? ((q / (C::LPE / 2 > 0 ? C::LPE / 2 : 1)) ? g1 : g0) : 0.f;
            const float ax = 1.0f - c.wx, ay = 1.0f - c.wy, az = 1.0f - c.wz;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // blended_k = sum_f g_f * sum_h table[f][h] * code[h]  (this lane's share)
                float blended = 0.f;
                float* gdst = (MODE == BWD_DENSE && dtab)
                                  ? dtab + ((size_t)(off + idx[k]) * C::ROW_BYTES + q * C::LANE_BYTES) / 2 : nullptr;
                if constexpr (H == 1) {
                    const half2_t t = as_half2(v[k].d[0]);
                    const float cf = (float)cw[0].x;
                    blended = (g0 * (float)t.x + g1 * (float)t.y) * cf;
                    if constexpr (DCODE) dc[0] = __fmaf_rn(w[k], g0 * (float)t.x + g1 * (float)t.y, dc[0]);
                    if (gdst) {
                        atomicAdd(gdst + 0, w[k] * g0 * cf);
                        atomicAdd(gdst + 1, w[k] * g1 * cf);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < C::NDW; ++j) {
                        float gf;
                        if constexpr (C::NDW > C::DPF) gf = (j >= C::DPF) ? g1 : g0;
                        else gf = gl;
                        const half2_t t = as_half2(v[k].d[j]);
                        const float wg = w[k] * gf;
                        blended = __fmaf_rn(gf, dot2(v[k].d[j], cw[j], 0.f), blended);
                        if constexpr (DCODE) {
                            dc[2 * j + 0] = __fmaf_rn(wg, (float)t.x, dc[2 * j + 0]);
                            dc[2 * j + 1] = __fmaf_rn(wg, (float)t.y, dc[2 * j + 1]);
                        }
                        if (gdst) {
                            atomicAdd(gdst + 2 * j + 0, wg * (float)cw[j].x);
                            atomicAdd(gdst + 2 * j + 1, wg * (float)cw[j].y);
                        }
                    }
                }
                // d w_k / d pos_d = sign * product of the other two weights
                const float ox = (k & 2 ? c.wy : ay) * (k & 4 ? c.wz : az);
                const float oy = (k & 1 ? c.wx : ax) * (k & 4 ? c.wz : az);
                const float oz = (k & 1 ? c.wx : ax) * (k & 2 ? c.wy : ay);
                const float sb = scale * blended;
                dxa = __fmaf_rn((k & 1) ? sb : -sb, ox, dxa);
                dya = __fmaf_rn((k & 2) ? sb : -sb, oy, dya);
                dza = __fmaf_rn((k & 4) ? sb : -sb, oz, dza);
            }
            if constexpr (MODE == BWD_FACTORED) {
                if (dtab) {
                    // 16 (corner, feature) items per sample and level, LPE lanes per sample -> 16/LPE instructions.
                    // item = i*LPE + q: bit0 = feature, bit1 = x, bit2 = y, bit3 = z.  With G stored [slot][entry][f]
                    // the 4 items (f0,f1) x (x0,x1) of one (y,z) are 16 contiguous bytes whenever entry(x1) =
                    // entry(x0)+1 (always on dense levels, for even x on hashed levels): the memory system charges
                    // per distinct 32-B sector per instruction (measured ~20 G sectors/s), so they cost ONE request.
                    constexpr int NI = 16 / C::LPE;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int item = i * C::LPE + q;
                        const int f = item & 1;
                        uint32_t ik;
                        float wk;
                        if constexpr (C::LPE == 8) {
                            // item = 8 i + q: corner = 4 i + (q >> 1) -- a 4-way select per instruction, not an 8-way one
                            const int c4 = q >> 1;
                            ik = idx[4 * i]; wk = w[4 * i];
#pragma unroll
                            for (int cc = 1; cc < 4; ++cc) {
                                if (c4 == cc) { ik = idx[4 * i + cc]; wk = w[4 * i + cc]; }
                            }
                        } else {
                            const int c = (item >> 1) & 7;
                            ik = idx[0]; wk = w[0];
#pragma unroll
                            for (int cc = 1; cc < 8; ++cc) {
                                if (c == cc) { ik = idx[cc]; wk = w[cc]; }
                            }
                        }
                        float val = wk * (f ? g1 : g0);
                        const uint32_t key = ((off + ik) << 6) | (uint32_t)crow;      // entry < 2^26 (checked at launch), slot < 64
                        // merge runs of equal keys over the adjacent sample lanes (only the run head issues)
                        const float sum = run_sum(val, run_links(key, s & 7));
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true);
                        const bool head = ((s & 7) == 0) || (pk != key);
                        if (head && sum != 0.f) {
                            float* gp = dtab + (((size_t)crow * g_total + (size_t)(off + ik)) * 2 + f);
                            atomicAdd(gp, sum);
                            bad |= !isfinite(sum);
                        }
                    }
                }
            }
        }
        // position gradient: reduce over the LPE lanes of the sample
        if (dx) {
#pragma unroll
            for (int m = 1; m < C::LPE; m <<= 1) {
                dxa += __shfl_xor(dxa, m * C::SPW); dya += __shfl_xor(dya, m * C::SPW); dza += __shfl_xor(dza, m * C::SPW);
            }
            if (q == 0 && valid) { dx[b * 3 + 0] = dxa; dx[b * 3 + 1] = dya; dx[b * 3 + 2] = dza; }
        }
        // code gradient: grid h receives contributions of both feature planes
        if (DCODE && code_sums) {
            // summed per code row in LDS.  Samples of a tile are adjacent lanes and, being consecutive samples of a
            // ray, nearly always share their row: runs of equal rows are merged with DPP first, the run head adds.
            // (Rows of invalid lanes hold zeros: their upstream gradient was forced to zero above.)
            const uint32_t rk = (uint32_t)crow;
            const uint32_t pk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rk, 0x111, 0xf, 0xf, true);
            const bool head = ((s & 7) == 0) || (pk != rk);
            float* crow_sums = csum + (kWaveTables ? (size_t)wave * n_slots * H : 0) + (size_t)rk * H;
            const RunLinks links = run_links(rk, s & 7);
            // plain add only where it is race-free: this lane is the ONLY run head of its 8-sample group (all 8 samples
            // share the code row -- consecutive samples of a ray); groups of one instruction differ in the grid they own.
            // Several heads in a group (rows A B A ...) could meet on one address: those lanes use the atomic.
            const unsigned long long head_bits = __ballot(head);
            const bool lone_head = __popcll((head_bits >> (lane & ~7)) & 0xFFull) == 1;
            auto emit = [&](int h, float val) {
                const float sum = run_sum(val, links);
                if (head && h < Hreal && sum != 0.f) {
                    if (kWaveTables && lone_head) crow_sums[h] += sum;    // this wave's table, this lane's own address
                    else atomicAdd(crow_sums + h, sum);
                }
            };
            if constexpr (H == 1) {
                emit(0, dc[0]);
            } else if constexpr (C::NDW > C::DPF) {   // H = 2, 4
#pragma unroll
                for (int j = 0; j < C::DPF; ++j) {
                    emit(2 * j, dc[2 * j] + dc[2 * (C::DPF + j)]);
                    emit(2 * j + 1, dc[2 * j + 1] + dc[2 * (C::DPF + j) + 1]);
                }
            } else {                                   // H >= 8: both partner lanes hold the sum; each emits half
#pragma unroll
                for (int i = 0; i < NH; ++i) dc[i] += __shfl_xor(dc[i], (C::LPE / 2) * C::SPW);
                const bool upper = q >= C::LPE / 2;
#pragma unroll
                for (int jj = 0; jj < C::NDW / 2; ++jj) {
                    const int j0 = jj, j1 = jj + C::NDW / 2;
                    const int pair = (q * C::NDW + (upper ? j1 : j0)) % C::DPF;
                    const float a = upper ? dc[2 * j1] : dc[2 * j0];
                    const float bq = upper ? dc[2 * j1 + 1] : dc[2 * j0 + 1];
                    emit(2 * pair, a);
                    emit(2 * pair + 1, bq);
                }
            }
        } else if (DCODE && dcode) {
            if constexpr (H == 1) {
                if (valid) dcode[b] = dc[0];
            } else if constexpr (C::NDW > C::DPF) {   // H = 2, 4: features are dwords [0,DPF) and [DPF,2DPF)
#pragma unroll
                for (int j = 0; j < C::DPF; ++j) {
                    const int h0 = 2 * j;
                    const float a = dc[2 * j] + dc[2 * (C::DPF + j)];
                    const float bq = dc[2 * j + 1] + dc[2 * (C::DPF + j) + 1];
                    if (valid && h0 < Hreal) dcode[b * Hreal + h0] = a;
                    if (valid && h0 + 1 < Hreal) dcode[b * Hreal + h0 + 1] = bq;
                }
            } else {                                   // H >= 8: partner lane q ^ (LPE/2) holds the other plane
#pragma unroll
                for (int i = 0; i < NH; ++i) dc[i] += __shfl_xor(dc[i], (C::LPE / 2) * C::SPW);
                if (valid && q < C::LPE / 2) {
#pragma unroll
                    for (int j = 0; j < C::NDW; ++j) {
                        const int pair = (q * C::NDW + j) % C::DPF;
                        const int h0 = 2 * pair;
                        if (h0 < Hreal) dcode[b * Hreal + h0] = dc[2 * j];
                        if (h0 + 1 < Hreal) dcode[b * Hreal + h0 + 1] = dc[2 * j + 1];
                    }
                }
            }
        }
    }
    if (nonfinite && __any(bad) && lane == 0) nonfinite[0] = 1.0f;
    if (code_sums) {
        __syncthreads();
        float* part = csum_part + (size_t)blockIdx.x * n_slots * H;
        for (int i = threadIdx.x; i < n_slots * H; i += WAVES * kWave) {
            float v = csum[i];
            if constexpr (kWaveTables) {
#pragma unroll
                for (int w = 1; w < WAVES; ++w) v += csum[(size_t)w * n_slots * H + i];
            }
            part[i] = v;
        }
    }
}

template <int H, int WAVES, int MODE, bool DCODE>
__global__ __launch_bounds__(WAVES * kWave) void ens_bwd_kernel(
    const float* __restrict__ x, int64_t B, const uint8_t* __restrict__ tab, const nsx_grid_geom g,
    const float* __restrict__ code, int64_t code_stride, const int32_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, const float* __restrict__ dout,
    float* __restrict__ dtab, float* __restrict__ dcode, float* __restrict__ dx, int64_t n_tiles,
    int n_slots, float* __restrict__ nonfinite, const int64_t* __restrict__ n_dev,
    float* __restrict__ csum_part) {
    ens_bwd_body<H, WAVES, MODE, DCODE>(x, B, tab, g, code, code_stride, code_index, window, Hreal, dout, dtab, dcode, dx,
                                        n_tiles, n_slots, nonfinite, n_dev, csum_part);
}

// The factored backward with in-kernel code-row sums over several sources in one launch (see ens_fwd_sources_kernel):
// source j adds to ITS gradient planes (src.plane_base[j] ...), has its own code rows (src.rows[j] of them), its own
// block partials of the code sums and its own dL/dx rows.
template <int H, int WAVES, bool TABLE_GRAD>
__global__ __launch_bounds__(WAVES * kWave) void ens_bwd_sources_kernel(
    const uint8_t* __restrict__ x, const uint8_t* __restrict__ tab, const nsx_grid_geom g,
    const uint8_t* __restrict__ code, int64_t code_stride, const uint8_t* __restrict__ code_index,
    const float* __restrict__ window, int Hreal, const uint8_t* __restrict__ dout, float* __restrict__ G,
    uint8_t* __restrict__ dx, float* __restrict__ nonfinite, const uint8_t* __restrict__ n_dev,
    float* __restrict__ csum_part, const EnsSources src) {
    using C = EnsCfg<H>;
    const int j = blockIdx.y;
    const int64_t B = src.B[j];
    const int rows = src.plane_base[j + 1] - src.plane_base[j];
    float* planes = TABLE_GRAD ? G + (size_t)src.plane_base[j] * (size_t)g.offset[g.n_levels] * 2 : nullptr;
    ens_bwd_body<H, WAVES, TABLE_GRAD ? BWD_FACTORED : BWD_GATHER, true>(
        reinterpret_cast<const float*>(x + j * src.x_stride), B, tab, g,
        reinterpret_cast<const float*>(code + j * src.code_stride), code_stride,
        reinterpret_cast<const int32_t*>(code_index + j * src.slot_stride), window, Hreal,
        reinterpret_cast<const float*>(dout + j * src.dout_stride), planes, nullptr,
        reinterpret_cast<float*>(dx + j * src.dx_stride), (B + C::SPW - 1) / C::SPW, rows,
        TABLE_GRAD ? nonfinite : nullptr, reinterpret_cast<const int64_t*>(n_dev + j * src.count_stride),
        csum_part + (size_t)j * src.csum_floats);
}

// dcode_rows[row][h] = window[h] * sum_blocks part[block][row][h]: the second stage of the in-kernel code-gradient sums
// (the chain rule through code' = code * window, hash_ensemble.py:133-138, folded in).  One block per code row; 1024 / H
// groups of H threads walk the ~2000 block partials with four loads in flight each (the walk is latency-bound: 256
// dependent rounds of 256 threads took 0.1 ms per step), then a tree over the groups.  Fixed order: deterministic sums.
constexpr int kCodeSumThreads = 1024;
template <int H>
__device__ __forceinline__ void code_sums_reduce_body(const float* __restrict__ part, int n_blocks, int n_slots,
                                                      const float* __restrict__ window, int Hreal,
                                                      float* __restrict__ dcode_rows) {
    __shared__ float red[kCodeSumThreads];
    const int row = blockIdx.x;
    constexpr int G = kCodeSumThreads / H;             // block-partials walked in parallel
    const int h = threadIdx.x % H, grp = threadIdx.x / H;
    const size_t step = (size_t)n_slots * H;
    const float* p = part + (size_t)row * H + h;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = grp;
    for (; b + 3 * G < n_blocks; b += 4 * G) {
        a0 += p[(size_t)b * step];
        a1 += p[(size_t)(b + G) * step];
        a2 += p[(size_t)(b + 2 * G) * step];
        a3 += p[(size_t)(b + 3 * G) * step];
    }
    for (; b < n_blocks; b += G) a0 += p[(size_t)b * step];
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
#pragma unroll
    for (int half = G / 2; half >= 1; half >>= 1) {
        if (grp < half) red[threadIdx.x] += red[threadIdx.x + half * H];
        __syncthreads();
    }
    if (grp == 0 && h < Hreal) dcode_rows[(size_t)row * Hreal + h] = red[h] * (window ? window[h] : 1.0f);
}

template <int H>
__global__ __launch_bounds__(kCodeSumThreads) void code_sums_reduce_kernel(const float* __restrict__ part, int n_blocks,
                                                                           int n_slots, const float* __restrict__ window,
                                                                           int Hreal, float* __restrict__ dcode_rows) {
    code_sums_reduce_body<H>(part, n_blocks, n_slots, window, Hreal, dcode_rows);
}

// grid (most rows of a source, sources): the second stage of ens_bwd_sources_kernel's code sums
template <int H>
__global__ __launch_bounds__(kCodeSumThreads) void code_sums_reduce_sources_kernel(
    const float* __restrict__ part, int n_blocks, const float* __restrict__ window, int Hreal,
    uint8_t* __restrict__ dcode_rows, const EnsSources src) {
    const int j = blockIdx.y;
    const int rows = src.plane_base[j + 1] - src.plane_base[j];
    if ((int)blockIdx.x >= rows) return;
    code_sums_reduce_body<H>(part + (size_t)j * src.csum_floats, n_blocks, rows, window, Hreal,
                             reinterpret_cast<float*>(dcode_rows + j * src.rows_stride));
}

// ------------------------------------------------------------------------------------------------
// factored table gradient alone: G[slot][entry][f] += w_corner * dout_f
// ------------------------------------------------------------------------------------------------
// The scatter half of the factored backward as its own kernel.  It touches neither the tables nor the codes (the
// gradient factors through the code slot), so it is independent of H, needs ~1/3 of the fused kernel's registers and
// -- being bound by the rate of the memory-side fp32 atomics, not by bandwidth -- runs on its own stream BESIDE the
// bandwidth-bound rest of the backward (the gather half, then the deformation field's backward, which only needs the
// gather's dL/dx).  Same lane mapping, same merging of equal cells over 8 adjacent samples (DPP run-length sums) and
// the same sector grouping of the (f, x) neighbours as ens_bwd_kernel<BWD_FACTORED>: the two produce the same sums up
// to the order of the atomics.
template <int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void ens_scatter_kernel(
    const float* __restrict__ x, int64_t B, const nsx_grid_geom g, const int32_t* __restrict__ code_index,
    const float* __restrict__ dout, float* __restrict__ G, int64_t n_tiles, float* __restrict__ nonfinite,
    const int64_t* __restrict__ n_dev) {
    constexpr int SPW = 8, LPE = 8;                   // 8 samples per wave step, 8 lanes (2 items each) per sample
    NSX_DEVICE_COUNT(B, n_tiles, SPW, n_dev);
    bool bad = false;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int L = g.n_levels;
    const int s = lane % SPW, q = lane / SPW;
    const int64_t wave_global = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t wave_count = (int64_t)gridDim.x * WAVES;
    const size_t g_total = g.offset[L];
    // this lane's two items: item = i * LPE + q, bit0 = feature, bits 1-3 = corner (x, y, z)
    const int f = q & 1, c0 = (q >> 1) & 3;           // corner of item 0 (z = 0); item 1 is the same (x, y) at z = 1
    for (int64_t tile = wave_global; tile < n_tiles; tile += wave_count) {
        const int64_t b_raw = tile * SPW + s;
        const bool valid = b_raw < B;
        const int64_t b = valid ? b_raw : B - 1;
        const float px = x[b * 3 + 0], py = x[b * 3 + 1], pz = x[b * 3 + 2];
        const uint32_t crow = (uint32_t)code_index[b];
        const float* drow = dout + b * (2 * L);
        float* gslot = G + (size_t)crow * g_total * 2 + f;
        for (int l = 0; l < L; ++l) {
            const float scale = g.scale[l];
            const uint32_t res = g.res[l], size = g.size[l], off = g.offset[l];
            const bool hashed = g.hashed[l] != 0;
            const float gf = valid ? drow[2 * l + f] : 0.f;
            const Cell c = cell_of(scale, px, py, pz);
            // the two corners of this lane: (x, y) from c0, z = 0 / 1
            const uint32_t cx = c.cx + (uint32_t)(c0 & 1), cy = c.cy + (uint32_t)(c0 >> 1);
            uint32_t ik[2];
            if (hashed) {
                const uint32_t mask = size - 1u, hxy = cx ^ (cy * 2654435761u);
                ik[0] = (hxy ^ (c.cz * 805459861u)) & mask;
                ik[1] = (hxy ^ ((c.cz + 1u) * 805459861u)) & mask;
            } else {
                const float inv = 1.0f / (float)size;
                const uint32_t r2 = res * res, bxy = cx + cy * res;
                ik[0] = umod(bxy + c.cz * r2, size, inv);
                ik[1] = umod(bxy + (c.cz + 1u) * r2, size, inv);
            }
            // corner weight in ens_bwd_kernel's association: w = wx * (wy * wz)
            const float wyz0 = ((c0 >> 1) ? c.wy : 1.0f - c.wy) * (1.0f - c.wz);
            const float wyz1 = ((c0 >> 1) ? c.wy : 1.0f - c.wy) * c.wz;
            const float wx = (c0 & 1) ? c.wx : 1.0f - c.wx;
            const float val2[2] = {(wx * wyz0) * gf, (wx * wyz1) * gf};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float val = val2[i];
                const uint32_t key = ((off + ik[i]) << 6) | crow;                      // entry < 2^26, slot < 64
                const float sum = run_sum(val, run_links(key, s));
                const uint32_t pk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true);
                const bool head = (s == 0) || (pk != key);
                if (head && sum != 0.f) {
                    atomicAdd(gslot + (size_t)(off + ik[i]) * 2, sum);
                    bad |= !isfinite(sum);
                }
            }
        }
    }
    if (nonfinite && __any(bad) && lane == 0) nonfinite[0] = 1.0f;
}

// ------------------------------------------------------------------------------------------------
// layout conversion + index dump
// ------------------------------------------------------------------------------------------------
__global__ void tables_from_tcnn_kernel(const float* __restrict__ src, int H, int Hp, int F_enc, int P,
                                        uint64_t total, half_t* __restrict__ dst16, float* __restrict__ dst32) {
    // one thread per native element (e, f, h)
    const uint64_t n = total * 2ull * (uint64_t)Hp;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % Hp);
        const int f = (int)((i / Hp) & 1);
        const uint64_t e = i / (2ull * Hp);
        float v = 0.f;
        if (h < H) {
            const int c = h / P, p = h % P;
            v = src[((uint64_t)c * total + e) * F_enc + p * 2 + f];
        }
        dst16[i] = (half_t)v;
        if (dst32) dst32[i] = v;
    }
}

__global__ void tables_to_tcnn_kernel(const float* __restrict__ src, int H, int Hp, int F_enc, int P,
                                      uint64_t total, float* __restrict__ dst) {
    const uint64_t n = total * 2ull * (uint64_t)H;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const int f = (int)((i / H) & 1);
        const uint64_t e = i / (2ull * H);
        const int c = h / P, p = h % P;
        dst[((uint64_t)c * total + e) * F_enc + p * 2 + f] = src[(e * 2ull + f) * Hp + h];
    }
}

// dtables[e][f][h] (+)= sum_slot G[e][slot][f] * code'[slot][h]   (code' = fp16(code*window), as in the forward)
template <int HP>
__global__ __launch_bounds__(256) void grad_expand_kernel(const float* __restrict__ G, int n_slots,
                                                          const float* __restrict__ code, int64_t code_stride,
                                                          const float* __restrict__ window, int Hreal,
                                                          uint64_t total, float* __restrict__ dtab, int accumulate) {
    extern __shared__ float cs[];     // [n_slots][HP]
    for (int i = threadIdx.x; i < n_slots * HP; i += blockDim.x) {
        const int sl = i / HP, h = i % HP;
        float c = 0.f;
        if (h < Hreal) c = code[sl * code_stride + h] * (window ? window[h] : 1.0f);
        cs[i] = (float)(half_t)c;
    }
    __syncthreads();
    // one thread per (entry, f, h-quad): 2*HP/4 threads per entry (HP>=4) -> coalesced float4 stores
    constexpr int HQ = HP >= 4 ? HP / 4 : 1;
    constexpr int HV = HP >= 4 ? 4 : HP;
    const uint64_t n = total * 2ull * HQ;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int hq = (int)(i % HQ);
        const int f = (int)((i / HQ) & 1);
        const uint64_t e = i / (2ull * HQ);
        const float* gr = G + e * 2ull + f;                 // G is [slot][entry][f]
        float acc[HV];
#pragma unroll
        for (int v = 0; v < HV; ++v) acc[v] = 0.f;
        for (int sl = 0; sl < n_slots; ++sl) {
            const float gv = gr[(uint64_t)sl * total * 2ull];
            if (gv != 0.f) {
#pragma unroll
                for (int v = 0; v < HV; ++v) acc[v] = __fmaf_rn(gv, cs[sl * HP + hq * HV + v], acc[v]);
            }
        }
        float* dst = dtab + (e * 2ull + f) * HP + hq * HV;
#pragma unroll
        for (int v = 0; v < HV; ++v) dst[v] = accumulate ? dst[v] + acc[v] : acc[v];
    }
}

// Eval-time pre-blend: every ray of an evaluation image carries the same time code, so the blend over the H grids can
// be applied to the TABLES once -- out[e][f] = fp16( sum_h fp16(code_h * window_h) * table[e][f][h] ) -- and the image
// rendered from one plain 2-feature hash grid (hashgrid_compat.hip): 4 B instead of 128 B per corner.
template <int HP>
__global__ __launch_bounds__(256) void preblend_kernel(const half_t* __restrict__ tab, uint64_t total,
                                                       const float* __restrict__ code, const float* __restrict__ window,
                                                       int Hreal, half_t* __restrict__ out) {
    __shared__ float cs[HP];
    if (threadIdx.x < HP) {
        const int h = threadIdx.x;
        float c = 0.f;
        if (h < Hreal) c = code[h] * (window ? window[h] : 1.0f);
        cs[h] = (float)(half_t)c;
    }
    __syncthreads();
    const uint64_t n = total * 2ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const half_t* row = tab + i * HP;
        float acc = 0.f;
        if constexpr (HP >= 8) {
#pragma unroll
            for (int v = 0; v < HP / 8; ++v) {
                const uint4 q = reinterpret_cast<const uint4*>(row)[v];
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const half2_t t = as_half2(w[k]);
                    acc = __fmaf_rn((float)t.x, cs[8 * v + 2 * k], acc);
                    acc = __fmaf_rn((float)t.y, cs[8 * v + 2 * k + 1], acc);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < HP; ++h) acc = __fmaf_rn((float)row[h], cs[h], acc);
        }
        out[i] = (half_t)acc;
    }
}

__global__ void hash_indices_kernel(const float* __restrict__ x, int64_t B, const nsx_grid_geom g,
                                    uint32_t* __restrict__ out) {
    const int L = g.n_levels;
    const int64_t n = B * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / L;
        const int l = (int)(i - b * L);
        const Cell c = cell_of(g.scale[l], x[b * 3], x[b * 3 + 1], x[b * 3 + 2]);
        uint32_t idx[8];
        corner_indices(c, g.res[l], g.size[l], g.hashed[l] != 0, idx);
#pragma unroll
        for (int k = 0; k < 8; ++k) out[i * 8 + k] = idx[k];
    }
}

static int ens_layout(int H, int* F_enc, int* P) {
    const int total = 2 * H;
    *F_enc = total >= 8 ? 8 : total;
    *P = total >= 8 ? 4 : H;
    return (total + 7) / 8;
}

template <int H>
static int launch_fwd(const float* x, int64_t B, const nsx_half* tables, int Hreal, const nsx_grid_geom* g,
                      const float* code, int64_t code_stride, const int32_t* code_index, const float* window,
                      nsx_half* out, const int64_t* n_dev, hipStream_t st) {
    using C = EnsCfg<H>;
    constexpr int WAVES = 4;
    const int64_t n_tiles = (B + C::SPW - 1) / C::SPW;
    int64_t blocks = (n_tiles + WAVES - 1) / WAVES;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    const size_t smem = (size_t)WAVES * C::SPW * (g->n_levels + 4) * sizeof(uint32_t);
    hipLaunchKernelGGL((ens_fwd_kernel<H, WAVES>), dim3((unsigned)blocks), dim3(WAVES * kWave), smem, st, x, B,
                       reinterpret_cast<const uint8_t*>(tables), *g, code, code_stride, code_index, window, Hreal,
                       reinterpret_cast<uint32_t*>(out), n_tiles, n_dev);
    NSX_LAUNCH_CHECK("nsx_hash_ensemble_fwd launch");
    return NSX_OK;
}

template <int H>
static int launch_bwd(const float* x, int64_t B, const nsx_half* tables, int Hreal, const nsx_grid_geom* g,
                      const float* code, int64_t code_stride, const int32_t* code_index, const float* window,
                      const float* dout, float* dtables, float* dcode, float* dx, const int64_t* n_dev, hipStream_t st,
                      int n_slots = 0, float* nonfinite = nullptr, float* dcode_rows = nullptr,
                      float* csum_part = nullptr) {
    using C = EnsCfg<H>;
    constexpr int WAVES = 4;
    const int64_t n_tiles = (B + C::SPW - 1) / C::SPW;
    int64_t blocks = (n_tiles + WAVES - 1) / WAVES;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    const bool dc = dcode || dcode_rows;
    const size_t smem = dcode_rows ? (size_t)(C::SPW == 8 ? WAVES : 1) * n_slots * H * sizeof(float) : 0;
#define NSX_BWD_LAUNCH(MODE, DC, NS, NF)                                                                               \
    hipLaunchKernelGGL((ens_bwd_kernel<H, WAVES, MODE, DC>), dim3((unsigned)blocks), dim3(WAVES * kWave), smem, st, x, B, \
                       reinterpret_cast<const uint8_t*>(tables), *g, code, code_stride, code_index, window, Hreal,       \
                       dout, dtables, dcode, dx, n_tiles, NS, NF, n_dev, dcode_rows ? csum_part : nullptr)
    if (n_slots > 0 && dtables) {
        if (dc) NSX_BWD_LAUNCH(BWD_FACTORED, true, n_slots, nonfinite);
        else NSX_BWD_LAUNCH(BWD_FACTORED, false, n_slots, nonfinite);
    } else if (n_slots > 0 || !dtables) {              // no table gradient asked of THIS kernel: the gather half alone
        if (dc) NSX_BWD_LAUNCH(BWD_GATHER, true, n_slots, nullptr);
        else NSX_BWD_LAUNCH(BWD_GATHER, false, n_slots, nullptr);
    } else {
        if (dc) NSX_BWD_LAUNCH(BWD_DENSE, true, 0, nullptr);
        else NSX_BWD_LAUNCH(BWD_DENSE, false, 0, nullptr);
    }
#undef NSX_BWD_LAUNCH
    NSX_LAUNCH_CHECK("nsx_hash_ensemble_bwd launch");
    if (dcode_rows) {
        hipLaunchKernelGGL((code_sums_reduce_kernel<H>), dim3((unsigned)n_slots), dim3(kCodeSumThreads), 0, st,
                           csum_part, (int)blocks, n_slots, window, Hreal, dcode_rows);
        NSX_LAUNCH_CHECK("nsx_hash_ensemble_bwd_codesum reduce launch");
    }
    return NSX_OK;
}

template <int H>
static int launch_fwd_sources(int W, const EnsSources& src, const void* x, const nsx_half* tables, int Hreal,
                              const nsx_grid_geom* g, const void* code, int64_t code_row_stride, const void* code_slot,
                              const float* window, void* out, const void* n_dev, hipStream_t st) {
    using C = EnsCfg<H>;
    constexpr int WAVES = 4;
    int64_t most = 0;
    for (int j = 0; j < W; ++j) most = src.B[j] > most ? src.B[j] : most;
    if (most == 0) return NSX_OK;
    int64_t blocks = ((most + C::SPW - 1) / C::SPW + WAVES - 1) / WAVES;
    int64_t cap = (int64_t)num_cus() * 8 / W;
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    const size_t smem = (size_t)WAVES * C::SPW * (g->n_levels + 4) * sizeof(uint32_t);
    hipLaunchKernelGGL((ens_fwd_sources_kernel<H, WAVES>), dim3((unsigned)blocks, (unsigned)W), dim3(WAVES * kWave), smem, st,
                       reinterpret_cast<const uint8_t*>(x), reinterpret_cast<const uint8_t*>(tables), *g,
                       reinterpret_cast<const uint8_t*>(code), code_row_stride, reinterpret_cast<const uint8_t*>(code_slot),
                       window, Hreal, reinterpret_cast<uint8_t*>(out), reinterpret_cast<const uint8_t*>(n_dev), src);
    NSX_LAUNCH_CHECK("ens_fwd_sources launch");
    return NSX_OK;
}

template <int H>
static int launch_bwd_sources(int W, EnsSources& src, const void* x, const nsx_half* tables, int Hreal,
                              const nsx_grid_geom* g, const void* code, int64_t code_row_stride, const void* code_slot,
                              const float* window, const void* dout, float* G, void* dx, float* nonfinite, const void* n_dev,
                              float* csum_part, int64_t csum_capacity, void* dcode_rows, hipStream_t st) {
    using C = EnsCfg<H>;
    constexpr int WAVES = 4;
    int64_t most = 0;
    int most_rows = 1;
    for (int j = 0; j < W; ++j) {
        most = src.B[j] > most ? src.B[j] : most;
        const int rows = src.plane_base[j + 1] - src.plane_base[j];
        most_rows = rows > most_rows ? rows : most_rows;
    }
    int64_t blocks = ((most + C::SPW - 1) / C::SPW + WAVES - 1) / WAVES;
    int64_t cap = (int64_t)num_cus() * 8 / W;
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;                     // sources without samples still get their (zero) code sums written
    src.csum_floats = blocks * most_rows * H;
    NSX_REQUIRE(src.csum_floats * W <= csum_capacity, "ens_bwd_sources: the code-sum scratch holds %lld floats, %lld needed",
                (long long)csum_capacity, (long long)(src.csum_floats * W));
    const size_t smem = (size_t)(C::SPW == 8 ? WAVES : 1) * most_rows * H * sizeof(float);
#define NSX_BWD_SOURCES(TG)                                                                                                \
    hipLaunchKernelGGL((ens_bwd_sources_kernel<H, WAVES, TG>), dim3((unsigned)blocks, (unsigned)W), dim3(WAVES * kWave), smem, \
                       st, reinterpret_cast<const uint8_t*>(x), reinterpret_cast<const uint8_t*>(tables), *g,              \
                       reinterpret_cast<const uint8_t*>(code), code_row_stride, reinterpret_cast<const uint8_t*>(code_slot), \
                       window, Hreal, reinterpret_cast<const uint8_t*>(dout), G, reinterpret_cast<uint8_t*>(dx), nonfinite, \
                       reinterpret_cast<const uint8_t*>(n_dev), csum_part, src)
    if (G) NSX_BWD_SOURCES(true);
    else NSX_BWD_SOURCES(false);
#undef NSX_BWD_SOURCES
    NSX_LAUNCH_CHECK("ens_bwd_sources launch");
    hipLaunchKernelGGL((code_sums_reduce_sources_kernel<H>), dim3((unsigned)most_rows, (unsigned)W), dim3(kCodeSumThreads),
                       0, st, csum_part, (int)blocks, window, Hreal, reinterpret_cast<uint8_t*>(dcode_rows), src);
    NSX_LAUNCH_CHECK("ens_bwd_sources reduce launch");
    return NSX_OK;
}

static int check_geom(const nsx_grid_geom* g, int Hp, const char* who) {
    NSX_REQUIRE(g != nullptr, "%s: geometry is NULL", who);
    NSX_REQUIRE(g->n_levels >= 1 && g->n_levels <= NSX_MAX_LEVELS, "%s: bad n_levels %d", who, g->n_levels);
    const uint64_t bytes = (uint64_t)g->offset[g->n_levels] * 4ull * (uint64_t)Hp;
    NSX_REQUIRE(bytes > 0, "%s: empty geometry", who);
    return NSX_OK;
}

template <int HP>
static int launch_expand(const float* G, int n_slots, const float* code, int64_t code_stride, const float* window,
                         int H, uint64_t total, float* dtables, int accumulate, hipStream_t st) {
    const size_t smem = (size_t)n_slots * HP * sizeof(float);
    hipLaunchKernelGGL((grad_expand_kernel<HP>), dim3(num_cus() * 8), dim3(256), smem, st, G, n_slots, code,
                       code_stride, window, H, total, dtables, accumulate);
    NSX_LAUNCH_CHECK("nsx_hash_grad_expand launch");
    return NSX_OK;
}

static int check_sources(int W, const EnsSources& src, const nsx_grid_geom* g, int H, const char* who) {
    NSX_REQUIRE(W >= 1 && W <= NSX_MAX_LEVELS, "%s: %d sources not in [1,%d]", who, W, NSX_MAX_LEVELS);
    NSX_REQUIRE(H >= 1 && H <= 32, "%s: H=%d not in [1,32]", who, H);
    if (int rc = check_geom(g, nsx_padded_grids(H), who)) return rc;
    for (int j = 0; j < W; ++j) {
        const int rows = src.plane_base[j + 1] - src.plane_base[j];
        NSX_REQUIRE(src.B[j] >= 0 && rows >= 1 && rows <= NSX_MAX_SLOTS, "%s: source %d brings %lld samples, %d code rows",
                    who, j, (long long)src.B[j], rows);
    }
    return NSX_OK;
}

int ens_fwd_sources(int W, const EnsSources& src, const void* x, const nsx_half* tables, int H, const nsx_grid_geom* g,
                    const void* code, int64_t code_row_stride, const void* code_slot, const float* window, void* out,
                    const void* n_dev, hipStream_t st) {
    if (int rc = check_sources(W, src, g, H, "ens_fwd_sources")) return rc;
    NSX_REQUIRE(x && tables && code && code_slot && out && n_dev, "ens_fwd_sources: NULL argument");
    switch (nsx_padded_grids(H)) {
#define NSX_FS_CASE(HP) case HP: return launch_fwd_sources<HP>(W, src, x, tables, H, g, code, code_row_stride, code_slot, \
                                                               window, out, n_dev, st);
        NSX_FS_CASE(1) NSX_FS_CASE(2) NSX_FS_CASE(4) NSX_FS_CASE(8) NSX_FS_CASE(16) NSX_FS_CASE(32)
#undef NSX_FS_CASE
    }
    set_error("ens_fwd_sources: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int ens_bwd_sources(int W, EnsSources& src, const void* x, const nsx_half* tables, int H, const nsx_grid_geom* g,
                    const void* code, int64_t code_row_stride, const void* code_slot, const float* window, const void* dout,
                    float* G, void* dx, float* nonfinite, const void* n_dev, float* csum_part, int64_t csum_capacity,
                    void* dcode_rows, hipStream_t st) {
    if (int rc = check_sources(W, src, g, H, "ens_bwd_sources")) return rc;
    NSX_REQUIRE(x && tables && code && code_slot && dout && dx && n_dev && csum_part && dcode_rows,
                "ens_bwd_sources: NULL argument");
    // the duplicate-merging key packs (entry, slot) into 32 bits: ((offset + index) << 6) | slot
    NSX_REQUIRE(!G || g->offset[g->n_levels] < (1u << 26), "ens_bwd_sources: more than 2^26 entries");
    switch (nsx_padded_grids(H)) {
#define NSX_BS_CASE(HP) case HP: return launch_bwd_sources<HP>(W, src, x, tables, H, g, code, code_row_stride, code_slot, \
                                                               window, dout, G, dx, nonfinite, n_dev, csum_part,         \
                                                               csum_capacity, dcode_rows, st);
        NSX_BS_CASE(1) NSX_BS_CASE(2) NSX_BS_CASE(4) NSX_BS_CASE(8) NSX_BS_CASE(16) NSX_BS_CASE(32)
#undef NSX_BS_CASE
    }
    set_error("ens_bwd_sources: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_tables_from_tcnn(const float* tcnn_params, int H, const nsx_grid_geom* g, nsx_half* native_f16,
                         float* native_master_f32, void* stream) {
    NSX_REQUIRE(tcnn_params && native_f16 && g, "nsx_tables_from_tcnn: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_tables_from_tcnn: H=%d not in [1,32]", H);
    NSX_REQUIRE(2 * H <= 8 || (2 * H) % 8 == 0, "nsx_tables_from_tcnn: 2H must be <= 8 or a multiple of 8 "
                "(hash_ensemble.py:80-82)");
    int F_enc, P;
    ens_layout(H, &F_enc, &P);
    const int Hp = nsx_padded_grids(H);
    const uint64_t total = g->offset[g->n_levels];
    hipLaunchKernelGGL(tables_from_tcnn_kernel, dim3(num_cus() * 8), dim3(256), 0, (hipStream_t)stream, tcnn_params, H,
                       Hp, F_enc, P, total, reinterpret_cast<half_t*>(native_f16), native_master_f32);
    NSX_LAUNCH_CHECK("nsx_tables_from_tcnn launch");
    return NSX_OK;
}

int nsx_tables_to_tcnn(const float* native_master_f32, int H, const nsx_grid_geom* g, float* tcnn_params,
                       void* stream) {
    NSX_REQUIRE(tcnn_params && native_master_f32 && g, "nsx_tables_to_tcnn: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_tables_to_tcnn: H=%d not in [1,32]", H);
    NSX_REQUIRE(2 * H <= 8 || (2 * H) % 8 == 0, "nsx_tables_to_tcnn: 2H must be <= 8 or a multiple of 8");
    int F_enc, P;
    ens_layout(H, &F_enc, &P);
    const int Hp = nsx_padded_grids(H);
    const uint64_t total = g->offset[g->n_levels];
    hipLaunchKernelGGL(tables_to_tcnn_kernel, dim3(num_cus() * 8), dim3(256), 0, (hipStream_t)stream,
                       native_master_f32, H, Hp, F_enc, P, total, tcnn_params);
    NSX_LAUNCH_CHECK("nsx_tables_to_tcnn launch");
    return NSX_OK;
}

int nsx_tables_preblend(const nsx_half* tables, int H, const nsx_grid_geom* g, const float* code_row,
                        const float* window, nsx_half* blended, void* stream) {
    NSX_REQUIRE(tables && g && code_row && blended, "nsx_tables_preblend: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_tables_preblend: H=%d not in [1,32]", H);
    const uint64_t total = g->offset[g->n_levels];
    hipStream_t st = (hipStream_t)stream;
    const half_t* tab = reinterpret_cast<const half_t*>(tables);
    half_t* out = reinterpret_cast<half_t*>(blended);
#define NSX_PB_CASE(HP) case HP: hipLaunchKernelGGL((preblend_kernel<HP>), dim3(num_cus() * 8), dim3(256), 0, st, tab, \
        total, code_row, window, H, out); break;
    switch (nsx_padded_grids(H)) {
        NSX_PB_CASE(1) NSX_PB_CASE(2) NSX_PB_CASE(4) NSX_PB_CASE(8) NSX_PB_CASE(16) NSX_PB_CASE(32)
        default: set_error("nsx_tables_preblend: unsupported H=%d", H); return NSX_ERR_UNSUPPORTED;
    }
#undef NSX_PB_CASE
    NSX_LAUNCH_CHECK("nsx_tables_preblend launch");
    return NSX_OK;
}

int nsx_hash_ensemble_fwd(const float* x, int64_t B, const nsx_half* tables, int H, const nsx_grid_geom* g,
                          const float* code, int64_t code_stride, const int32_t* code_index,
                          const float* window, nsx_half* out, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_ensemble_fwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && tables && code && out, "nsx_hash_ensemble_fwd: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_ensemble_fwd: H=%d not in [1,32]", H);
    const int Hp = nsx_padded_grids(H);
    if (int rc = check_geom(g, Hp, "nsx_hash_ensemble_fwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    switch (Hp) {
        case 1: return launch_fwd<1>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
        case 2: return launch_fwd<2>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
        case 4: return launch_fwd<4>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
        case 8: return launch_fwd<8>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
        case 16: return launch_fwd<16>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
        case 32: return launch_fwd<32>(x, B, tables, H, g, code, code_stride, code_index, window, out, n_device, st);
    }
    set_error("nsx_hash_ensemble_fwd: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int nsx_hash_ensemble_bwd(const float* x, int64_t B, const nsx_half* tables, int H, const nsx_grid_geom* g,
                          const float* code, int64_t code_stride, const int32_t* code_index,
                          const float* window, const float* dout, float* dtables, float* dcode, float* dx,
                          const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_ensemble_bwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && tables && code && dout, "nsx_hash_ensemble_bwd: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_ensemble_bwd: H=%d not in [1,32]", H);
    const int Hp = nsx_padded_grids(H);
    if (int rc = check_geom(g, Hp, "nsx_hash_ensemble_bwd")) return rc;
    hipStream_t st = (hipStream_t)stream;
    switch (Hp) {
        case 1: return launch_bwd<1>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
        case 2: return launch_bwd<2>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
        case 4: return launch_bwd<4>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
        case 8: return launch_bwd<8>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
        case 16: return launch_bwd<16>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
        case 32: return launch_bwd<32>(x, B, tables, H, g, code, code_stride, code_index, window, dout, dtables, dcode, dx, n_device, st);
    }
    set_error("nsx_hash_ensemble_bwd: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int nsx_hash_ensemble_bwd_factored(const float* x, int64_t B, const nsx_half* tables, int H,
                                   const nsx_grid_geom* g, const float* code_table, int64_t code_stride,
                                   int n_slots, const int32_t* code_slot, const float* window,
                                   const float* dout, float* G, float* dcode, float* dx, float* nonfinite,
                                   const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_ensemble_bwd_factored: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && tables && code_table && code_slot && dout, "nsx_hash_ensemble_bwd_factored: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_ensemble_bwd_factored: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_ensemble_bwd_factored: n_slots=%d not in [1,%d]",
                n_slots, NSX_MAX_SLOTS);
    const int Hp = nsx_padded_grids(H);
    if (int rc = check_geom(g, Hp, "nsx_hash_ensemble_bwd_factored")) return rc;
    // the duplicate-merging key packs (entry, slot) into 32 bits: ((offset + index) << 6) | slot
    NSX_REQUIRE(!G || g->offset[g->n_levels] < (1u << 26), "nsx_hash_ensemble_bwd_factored: %u table entries exceed the "
                "2^26 the run-merge key can address", g->offset[g->n_levels]);
    hipStream_t st = (hipStream_t)stream;
    switch (Hp) {
        case 1: return launch_bwd<1>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
        case 2: return launch_bwd<2>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
        case 4: return launch_bwd<4>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
        case 8: return launch_bwd<8>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
        case 16: return launch_bwd<16>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
        case 32: return launch_bwd<32>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, G, dcode, dx, n_device, st, n_slots, nonfinite);
    }
    set_error("nsx_hash_ensemble_bwd_factored: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int64_t nsx_hash_codesum_scratch_floats(int n_slots, int H) {
    if (n_slots < 1 || n_slots > NSX_MAX_SLOTS || H < 1 || H > 32) return 0;
    return (int64_t)num_cus() * 8 * n_slots * nsx_padded_grids(H);
}

int nsx_hash_ensemble_bwd_codesum(const float* x, int64_t B, const nsx_half* tables, int H,
                                  const nsx_grid_geom* g, const float* code_table, int64_t code_stride,
                                  int n_slots, const int32_t* code_slot, const float* window,
                                  const float* dout, float* G, float* dcode_rows, float* scratch, float* dx,
                                  float* nonfinite, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_ensemble_bwd_codesum: negative batch");
    NSX_REQUIRE(dcode_rows && scratch, "nsx_hash_ensemble_bwd_codesum: dcode_rows / scratch is NULL");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_ensemble_bwd_codesum: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_ensemble_bwd_codesum: n_slots=%d not in [1,%d]",
                n_slots, NSX_MAX_SLOTS);
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {                                       // no sample: the sums are zero
        if (hipMemsetAsync(dcode_rows, 0, (size_t)n_slots * H * sizeof(float), st) != hipSuccess) {
            set_error("nsx_hash_ensemble_bwd_codesum: memset failed");
            return NSX_ERR_INVALID;
        }
        return NSX_OK;
    }
    NSX_REQUIRE(x && tables && code_table && code_slot && dout, "nsx_hash_ensemble_bwd_codesum: NULL argument");
    const int Hp = nsx_padded_grids(H);
    if (int rc = check_geom(g, Hp, "nsx_hash_ensemble_bwd_codesum")) return rc;
    NSX_REQUIRE(!G || g->offset[g->n_levels] < (1u << 26), "nsx_hash_ensemble_bwd_codesum: %u table entries exceed the "
                "2^26 the run-merge key can address", g->offset[g->n_levels]);
    switch (Hp) {
#define NSX_CS_CASE(HP) case HP: return launch_bwd<HP>(x, B, tables, H, g, code_table, code_stride, code_slot, window, dout, \
                                                       G, nullptr, dx, n_device, st, n_slots, nonfinite, dcode_rows, scratch);
        NSX_CS_CASE(1) NSX_CS_CASE(2) NSX_CS_CASE(4) NSX_CS_CASE(8) NSX_CS_CASE(16) NSX_CS_CASE(32)
#undef NSX_CS_CASE
    }
    set_error("nsx_hash_ensemble_bwd_codesum: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int nsx_hash_ensemble_bwd_scatter(const float* x, int64_t B, const nsx_grid_geom* g, int n_slots,
                                  const int32_t* code_slot, const float* dout, float* G, float* nonfinite,
                                  int blocks_per_cu, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_ensemble_bwd_scatter: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && code_slot && dout && G, "nsx_hash_ensemble_bwd_scatter: NULL argument");
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_ensemble_bwd_scatter: n_slots=%d not in [1,%d]",
                n_slots, NSX_MAX_SLOTS);
    if (int rc = check_geom(g, 1, "nsx_hash_ensemble_bwd_scatter")) return rc;
    // the duplicate-merging key packs (entry, slot) into 32 bits
    NSX_REQUIRE(g->offset[g->n_levels] < (1u << 26), "nsx_hash_ensemble_bwd_scatter: %u table entries exceed the 2^26 "
                "the run-merge key can address", g->offset[g->n_levels]);
    constexpr int WAVES = 4;
    const int64_t n_tiles = (B + 7) / 8;
    int64_t blocks = (n_tiles + WAVES - 1) / WAVES;
    if (blocks_per_cu < 1) blocks_per_cu = 8;
    const int64_t cap = (int64_t)num_cus() * blocks_per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((ens_scatter_kernel<WAVES>), dim3((unsigned)blocks), dim3(WAVES * kWave), 0, (hipStream_t)stream, x,
                       B, *g, code_slot, dout, G, n_tiles, nonfinite, n_device);
    NSX_LAUNCH_CHECK("nsx_hash_ensemble_bwd_scatter launch");
    return NSX_OK;
}

int nsx_hash_grad_expand(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                         const float* window, int H, const nsx_grid_geom* g, float* dtables, int accumulate,
                         void* stream) {
    NSX_REQUIRE(G && code_table && dtables && g, "nsx_hash_grad_expand: NULL argument");
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_hash_grad_expand: H=%d not in [1,32]", H);
    NSX_REQUIRE(n_slots >= 1 && n_slots <= NSX_MAX_SLOTS, "nsx_hash_grad_expand: n_slots=%d not in [1,%d]", n_slots,
                NSX_MAX_SLOTS);
    const uint64_t total = g->offset[g->n_levels];
    hipStream_t st = (hipStream_t)stream;
    switch (nsx_padded_grids(H)) {
        case 1: return launch_expand<1>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
        case 2: return launch_expand<2>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
        case 4: return launch_expand<4>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
        case 8: return launch_expand<8>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
        case 16: return launch_expand<16>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
        case 32: return launch_expand<32>(G, n_slots, code_table, code_stride, window, H, total, dtables, accumulate, st);
    }
    set_error("nsx_hash_grad_expand: unsupported H=%d", H);
    return NSX_ERR_UNSUPPORTED;
}

int nsx_hash_indices(const float* x, int64_t B, const nsx_grid_geom* g, uint32_t* idx, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hash_indices: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && g && idx, "nsx_hash_indices: NULL argument");
    hipLaunchKernelGGL(hash_indices_kernel, dim3(num_cus() * 8), dim3(256), 0, (hipStream_t)stream, x, B, *g, idx);
    NSX_LAUNCH_CHECK("nsx_hash_indices launch");
    return NSX_OK;
}

}  // extern "C"

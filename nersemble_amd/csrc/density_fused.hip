// density_fused.hip -- the no-grad density pass on a PRE-BLENDED grid as one kernel: (world position + offset) -> scene-box
// normalisation + in-box selector -> trilinear lookup in one 2-feature table (16 levels) -> mlp_base on MFMA -> mlp_base's
// output row and density = exp(h0) * selector.
//
// Replaces, for the evaluation image / the sampler's sigma_fn on a single-timestep bundle, the four launches
//   nsx_sample_positions (normalise)  ->  nsx_hashgrid_fwd (F = 2)  ->  nsx_mlp_fwd (mlp_base)  ->  nsx_density_fwd
// behind NeRSembleNeRFactoField.density_fn / get_density (nersemble_nerfacto_field.py:228-301; callers
// nersemble_instant_ngp.py:235-266 sigma_fn and :184-196 occupancy update) with the H tables blended by the image's one time
// code (HashEnsemble.preblend, hash_ensemble.py:155-156 is linear in the tables).  An evaluation image marches 25.9 M samples:
// the [S, 32] fp16 features (64 B written + 64 B read per sample), the normalised positions (12 + 12 B) and the selector
// never reach HBM, and the 8 MFMAs of mlp_base run under the gather latency of the next tile's lookups.
//
// Arithmetic is the four kernels' own, operation by operation (normalisation: field_glue.hip sample_positions_kernel; lookup:
// hashgrid_compat.hip hashgrid_fwd_pair_kernel -- the x = 0 / x = 1 corners of a cell are gathered by the two lanes of a
// sample, whose loads leave in ONE instruction and mostly hit one line, fp32 partial sums added across the pair; features
// rounded to fp16; mlp_base: mlp_device.h forward_tile, the same fragments in the same order; density: expf of the fp16 h0):
// outputs are bit-identical to the four-launch route (tests/test_field_gpu.py::test_fused_density_equals_the_four_launches).
#include "mlp_device.h"
#pragma clang fp contract(off)

namespace nsx {

struct FusedBox { float lo[3], ext[3]; };

__device__ __forceinline__ uint32_t fused_entry_of(const uint32_t c[3], uint32_t res, uint32_t size, bool hashed) {
    if (hashed) return (c[0] ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u)) & (size - 1u);
    return umod(c[0] + c[1] * res + c[2] * res * res, size, 1.0f / (float)size);
}

constexpr int FUSED_LEVELS = 16;        // 16 levels x 2 features = the 32 inputs of mlp_base

template <int NH>
__global__ __launch_bounds__(MLP_WAVES * kWave) void density_fused_kernel(
    const float* __restrict__ pos_world, const float* __restrict__ off, int64_t S, FusedBox box,
    const half_t* __restrict__ table, const nsx_grid_geom g, const half_t* __restrict__ W, half_t* __restrict__ base_out,
    int64_t base_stride, float* __restrict__ density, int64_t n_tiles, const int64_t* __restrict__ n_dev) {
    NSX_DEVICE_COUNT(S, n_tiles, 32, n_dev);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    f16x8* frags = reinterpret_cast<f16x8*>(smem_raw);
    const FragPlan p = make_plan(NH, false);
    stage_weights<NH>(W, reinterpret_cast<half_t*>(smem_raw + (size_t)p.total * kWave * sizeof(f16x8)), frags, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, kb = lane >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * MLP_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * MLP_WAVES) {
        const int64_t b_raw = tile * 32 + n;
        const int64_t b = b_raw < S ? b_raw : S - 1;
        // ---- scene-box normalisation of (position + offset), in-box selector (field_glue.hip)
        float xs[3];
        bool inside = true;
        {
            float q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = pos_world[b * 3 + a];
                if (off) v = v + off[b * 3 + a];
                q[a] = (v - box.lo[a]) / box.ext[a];
                inside = inside && (q[a] > 0.0f) && (q[a] < 1.0f);
            }
            const float m = inside ? 1.0f : 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a) xs[a] = q[a] * m;
        }
        // ---- lookup: lane kb gathers the 4 (y, z) corners of x-corner kb of every level; the pair adds its partial sums
        float f0[FUSED_LEVELS], f1[FUSED_LEVELS];
#pragma unroll
        for (int l = 0; l < FUSED_LEVELS; ++l) {
            const float scale = g.scale[l];
            uint32_t c0[3]; float w[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float pp = __fmaf_rn(scale, xs[d], 0.5f), fl = floorf(pp);
                c0[d] = (uint32_t)(int32_t)fl; w[d] = pp - fl;
            }
            const float wx = kb ? w[0] : 1.f - w[0];
            const uint32_t res = g.res[l], size = g.size[l];
            const bool hashed = g.hashed[l] != 0;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t c[3] = {c0[0] + (uint32_t)kb, c0[1] + (k & 1), c0[2] + ((k >> 1) & 1)};
                const uint32_t e = fused_entry_of(c, res, size, hashed);
                const float wk = wx * ((k & 1) ? w[1] : 1.f - w[1]) * ((k & 2) ? w[2] : 1.f - w[2]);
                const uint32_t rv = *reinterpret_cast<const uint32_t*>(table + ((size_t)g.offset[l] + e) * 2);
                const half2_t h = as_half2(rv);
                a0 = __fmaf_rn(wk, (float)h.x, a0);
                a1 = __fmaf_rn(wk, (float)h.y, a1);
            }
            f0[l] = a0 + __shfl_xor(a0, 32);
            f1[l] = a1 + __shfl_xor(a1, 32);
        }
        // ---- input fragments of mlp_base: element (t, j) of lane (n, kb) is feature k = 16 t + 8 kb + j = level k / 2, f = k & 1
        f16x8 x[2], h1[4], h2[4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lo = 8 * t + (j >> 1), hi = lo + 4;
                const float v = (j & 1) ? (kb ? f1[hi] : f1[lo]) : (kb ? f0[hi] : f0[lo]);
                x[t][j] = (half_t)v;
            }
        const f32x16 o = forward_tile<NH>(frags, p, lane, x, h1, h2);
        if (b_raw < S) {
            // rows held by this lane: r = 0..7 -> neurons (r & 3) + 8 (r >> 2) + 4 kb: two runs of 4 consecutive halfs
            f16x4 lo4, hi4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo4[r] = (half_t)o[r]; hi4[r] = (half_t)o[4 + r]; }
            if (base_out) {
                half_t* row = base_out + b * base_stride + 4 * kb;
                *reinterpret_cast<f16x4*>(row) = lo4;
                *reinterpret_cast<f16x4*>(row + 8) = hi4;
            }
            if (kb == 0) density[b] = expf((float)lo4[0]) * (inside ? 1.0f : 0.0f);       // field_glue.hip density_fwd_kernel
        }
    }
}

static size_t fused_smem(int NH) {      // fragments + the staging copy of the flat weights (unaligned vectors only)
    return (size_t)make_plan(NH, false).total * kWave * sizeof(f16x8) +
           (size_t)(MLP_W * MLP_IN + (NH ? MLP_W * MLP_W : 0) + MLP_OUT * MLP_W) * sizeof(half_t);
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_density_fused_fwd(const float* positions_world, const float* offsets, int64_t S, const float* field_aabb_host,
                          const nsx_half* table, const nsx_grid_geom* g, const nsx_half* base_weights, int base_hidden_mats,
                          nsx_half* base_out, int64_t base_out_stride, float* density, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_density_fused_fwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(positions_world && field_aabb_host && table && g && base_weights && density,
                "nsx_density_fused_fwd: NULL argument");
    NSX_REQUIRE(g->n_levels == FUSED_LEVELS, "nsx_density_fused_fwd: built for %d levels x 2 features (got %d levels)",
                FUSED_LEVELS, g->n_levels);
    NSX_REQUIRE(base_hidden_mats == 0 || base_hidden_mats == 1, "nsx_density_fused_fwd: base_hidden_mats must be 0 or 1 (got %d)",
                base_hidden_mats);
    NSX_REQUIRE(!base_out || (base_out_stride >= MLP_OUT && base_out_stride % 4 == 0 &&
                              (reinterpret_cast<uintptr_t>(base_out) & 7) == 0),
                "nsx_density_fused_fwd: base_out rows must be >= 16 halfs, a multiple of 4 halfs apart and 8-byte aligned");
    FusedBox box;
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = field_aabb_host[d];
        box.ext[d] = field_aabb_host[3 + d] - field_aabb_host[d];
    }
    hipStream_t st = (hipStream_t)stream;
    const half_t* t = reinterpret_cast<const half_t*>(table);
    const half_t* W = reinterpret_cast<const half_t*>(base_weights);
    half_t* bo = reinterpret_cast<half_t*>(base_out);
    // A pass of many million samples (an evaluation image: 6.5 M per 32 768-ray bundle) is enqueued in slices of 2^20: the
    // blocks of one launch stride through its samples together, and the ray-ordered samples they hold at any moment share
    // their coarse cells in L2 -- over 25 strides of ONE launch the blocks drift apart and the kernel ran 0.40 ms per 2^20
    // samples instead of 0.33 (profiles/r06_eval_image.txt).  Slicing needs host-known row counts: a call under a device-side
    // count is one launch.
    const int64_t slice = n_device ? S : ((int64_t)1 << 20);
    for (int64_t s0 = 0; s0 < S; s0 += slice) {
        const int64_t Sn = S - s0 < slice ? S - s0 : slice;
        const int64_t n_tiles = (Sn + 31) / 32;
        int64_t blocks = (n_tiles + MLP_WAVES - 1) / MLP_WAVES;
        const int64_t cap = (int64_t)num_cus() * 8;
        if (blocks > cap) blocks = cap;
        const float* pw = positions_world + s0 * 3;
        const float* of = offsets ? offsets + s0 * 3 : nullptr;
        half_t* bos = bo ? bo + s0 * base_out_stride : nullptr;
        if (base_hidden_mats == 0)
            hipLaunchKernelGGL((density_fused_kernel<0>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), fused_smem(0), st,
                               pw, of, Sn, box, t, *g, W, bos, base_out_stride, density + s0, n_tiles, n_device);
        else
            hipLaunchKernelGGL((density_fused_kernel<1>), dim3((unsigned)blocks), dim3(MLP_WAVES * kWave), fused_smem(1), st,
                               pw, of, Sn, box, t, *g, W, bos, base_out_stride, density + s0, n_tiles, n_device);
        NSX_LAUNCH_CHECK("nsx_density_fused_fwd launch");
    }
    return NSX_OK;
}

}  // extern "C"

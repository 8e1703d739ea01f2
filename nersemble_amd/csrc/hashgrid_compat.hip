// hashgrid_compat.hip -- a plain tcnn-shaped HashGrid encoding (one table, F features per level, tcnn AoS layout).
//
// Operator-level compatibility path: lets the reference's own HashEnsemble module run unmodified on this library
// (tcnn.Encoding(n_input_dims=3, {"otype": "HashGrid", ...}) at hash_ensemble.py:42-50 / :102-104).  The fused
// HashEnsemble kernels (hash_ensemble.hip) are the performance path; this one is the straightforward per-(sample,
// level) formulation of the same algorithm: fp16 table [total][F] in, fp16 features [B][L*F] out, fp32 atomics for
// the parameter gradient, analytic dL/dx.
#include "nsx_common.h"

namespace nsx {

struct CellC { uint32_t c[3]; float w[3]; };

__device__ __forceinline__ uint32_t entry_of(const uint32_t c[3], uint32_t res, uint32_t size, bool hashed) {
    if (hashed) return (c[0] ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u)) & (size - 1u);
    return umod(c[0] + c[1] * res + c[2] * res * res, size, 1.0f / (float)size);
}

// Forward lookup with TWO lanes per (sample, level) (round 4; the one-lane-per-item form it replaced -- one lane gathers all 8
// corners -- was 10 % slower on the evaluation image and is gone since round 5): lane pair = the two x-corners of the cell, each lane gathers the 4
// (y, z) corners of its x.  Entries of x and x + 1 are neighbours in the table (dense levels: consecutive indices; hashed
// levels: x enters the hash un-multiplied, so x ^ h and (x + 1) ^ h differ in their low bits only and share a 128-byte line
// unless x is the last of its 32-block), and lanes of ONE instruction that hit one line cost one request: 32 lines per
// wave instruction instead of 64.  The 4-byte gathers of the one-lane kernel run at 254 G/s = 32.5 TB/s of 128-byte lines,
// the L2's limit (MI355X_MICROARCH: ~34.5 TB/s).
template <int F>
__global__ __launch_bounds__(256) void hashgrid_fwd_pair_kernel(const float* __restrict__ x, int64_t B,
                                                                const half_t* __restrict__ table, const nsx_grid_geom g,
                                                                half_t* __restrict__ out) {
    const int L = g.n_levels;
    const int shift = (L & (L - 1)) == 0 ? __builtin_ctz((unsigned)L) : -1;
    const int64_t n_items = B * L;
    const int64_t n_iter = (n_items * 2 + (int64_t)gridDim.x * blockDim.x - 1) / ((int64_t)gridDim.x * blockDim.x);
    for (int64_t it = 0; it < n_iter; ++it) {           // (all lanes iterate together: the pair exchange below is a shuffle)
        const int64_t t = it * (int64_t)gridDim.x * blockDim.x + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int64_t i_raw = t >> 1;
        const bool live = i_raw < n_items;
        const int64_t i = live ? i_raw : n_items - 1;
        const int xc = (int)(t & 1);
        const int64_t b = shift >= 0 ? (i >> shift) : i / L;
        const int l = (int)(i - b * L);
        const float scale = g.scale[l];
        uint32_t c0[3]; float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float p = __fmaf_rn(scale, x[b * 3 + d], 0.5f), f = floorf(p);
            c0[d] = (uint32_t)(int32_t)f; w[d] = p - f;
        }
        const float wx = xc ? w[0] : 1.f - w[0];
        const uint32_t res = g.res[l], size = g.size[l];
        const bool hashed = g.hashed[l] != 0;
        float acc[F];
#pragma unroll
        for (int j = 0; j < F; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c[3] = {c0[0] + (uint32_t)xc, c0[1] + (k & 1), c0[2] + ((k >> 1) & 1)};
            const uint32_t e = entry_of(c, res, size, hashed);
            const float wk = wx * ((k & 1) ? w[1] : 1.f - w[1]) * ((k & 2) ? w[2] : 1.f - w[2]);
            typedef uint32_t row_t __attribute__((ext_vector_type(F / 2)));
            const row_t rv = *reinterpret_cast<const row_t*>(table + ((size_t)g.offset[l] + e) * F);
#pragma unroll
            for (int j = 0; j < F / 2; ++j) {
                const half2_t h = as_half2(rv[j]);
                acc[2 * j] = __fmaf_rn(wk, (float)h.x, acc[2 * j]);
                acc[2 * j + 1] = __fmaf_rn(wk, (float)h.y, acc[2 * j + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < F; ++j) acc[j] += __shfl_xor(acc[j], 1);
        if (live && xc == 0) {
            typedef uint32_t row_t __attribute__((ext_vector_type(F / 2)));
            row_t ov;
#pragma unroll
            for (int j = 0; j < F / 2; ++j) {
                half2_t h; h.x = (half_t)acc[2 * j]; h.y = (half_t)acc[2 * j + 1];
                ov[j] = as_u32(h);
            }
            *reinterpret_cast<row_t*>(out + b * (int64_t)(L * F) + l * F) = ov;
        }
    }
}

template <int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(const float* __restrict__ x, int64_t B,
                                                           const half_t* __restrict__ table, const nsx_grid_geom g,
                                                           const half_t* __restrict__ dout, float* __restrict__ dtable,
                                                           float* __restrict__ dx) {
    const int L = g.n_levels;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B * L; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / L;
        const int l = (int)(i - b * L);
        const float scale = g.scale[l];
        uint32_t c0[3]; float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float p = __fmaf_rn(scale, x[b * 3 + d], 0.5f), f = floorf(p);
            c0[d] = (uint32_t)(int32_t)f; w[d] = p - f;
        }
        float go[F];
#pragma unroll
        for (int j = 0; j < F; ++j) go[j] = (float)dout[b * (int64_t)(L * F) + l * F + j];
        float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t c[3] = {c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1)};
            const uint32_t e = entry_of(c, g.res[l], g.size[l], g.hashed[l] != 0);
            const float wd[3] = {(k & 1) ? w[0] : 1.f - w[0], (k & 2) ? w[1] : 1.f - w[1], (k & 4) ? w[2] : 1.f - w[2]};
            const float wk = wd[0] * wd[1] * wd[2];
            const size_t at = ((size_t)g.offset[l] + e) * F;
            float dotv = 0.f;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                if (dtable) atomicAdd(dtable + at + j, wk * go[j]);
                dotv = __fmaf_rn(go[j], (float)table[at + j], dotv);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float other = wd[(d + 1) % 3] * wd[(d + 2) % 3];
                gx[d] += scale * (((k >> d) & 1) ? other : -other) * dotv;
            }
        }
        if (dx) {
#pragma unroll
            for (int d = 0; d < 3; ++d) atomicAdd(dx + b * 3 + d, gx[d]);
        }
    }
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_hashgrid_fwd(const float* x, int64_t B, const nsx_half* table, int F, const nsx_grid_geom* g, nsx_half* out,
                     void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hashgrid_fwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && table && g && out, "nsx_hashgrid_fwd: NULL argument");
    NSX_REQUIRE(F == 2 || F == 4 || F == 8, "nsx_hashgrid_fwd: n_features_per_level must be 2, 4 or 8 (got %d)", F);
    const dim3 grid(num_cus() * 8), block(256);
    hipStream_t st = (hipStream_t)stream;
    const half_t* t = reinterpret_cast<const half_t*>(table);
    half_t* o = reinterpret_cast<half_t*>(out);
    if (F == 2) hipLaunchKernelGGL((hashgrid_fwd_pair_kernel<2>), grid, block, 0, st, x, B, t, *g, o);
    else if (F == 4) hipLaunchKernelGGL((hashgrid_fwd_pair_kernel<4>), grid, block, 0, st, x, B, t, *g, o);
    else hipLaunchKernelGGL((hashgrid_fwd_pair_kernel<8>), grid, block, 0, st, x, B, t, *g, o);
    NSX_LAUNCH_CHECK("nsx_hashgrid_fwd launch");
    return NSX_OK;
}

int nsx_hashgrid_bwd(const float* x, int64_t B, const nsx_half* table, int F, const nsx_grid_geom* g,
                     const nsx_half* dout, float* dtable, float* dx_zeroed, void* stream) {
    NSX_REQUIRE(B >= 0, "nsx_hashgrid_bwd: negative batch");
    if (B == 0) return NSX_OK;
    NSX_REQUIRE(x && table && g && dout, "nsx_hashgrid_bwd: NULL argument");
    NSX_REQUIRE(F == 2 || F == 4 || F == 8, "nsx_hashgrid_bwd: n_features_per_level must be 2, 4 or 8 (got %d)", F);
    const dim3 grid(num_cus() * 8), block(256);
    hipStream_t st = (hipStream_t)stream;
    const half_t* t = reinterpret_cast<const half_t*>(table);
    const half_t* d = reinterpret_cast<const half_t*>(dout);
    if (F == 2) hipLaunchKernelGGL((hashgrid_bwd_kernel<2>), grid, block, 0, st, x, B, t, *g, d, dtable, dx_zeroed);
    else if (F == 4) hipLaunchKernelGGL((hashgrid_bwd_kernel<4>), grid, block, 0, st, x, B, t, *g, d, dtable, dx_zeroed);
    else hipLaunchKernelGGL((hashgrid_bwd_kernel<8>), grid, block, 0, st, x, B, t, *g, d, dtable, dx_zeroed);
    NSX_LAUNCH_CHECK("nsx_hashgrid_bwd launch");
    return NSX_OK;
}

}  // extern "C"

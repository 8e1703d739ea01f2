// nsx_common.h -- shared helpers of libnsx.so (error reporting, launch checks, device utils).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nsx.h"

namespace nsx {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int option(int which);      // nsx_get_option without the range check (nsx_core.hip)

#define NSX_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            ::nsx::set_error(__VA_ARGS__);             \
            return NSX_ERR_INVALID;                    \
        }                                              \
    } while (0)

#define NSX_LAUNCH_CHECK(what)                                         \
    do {                                                               \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) return ::nsx::hip_fail(e__, what);      \
    } while (0)

constexpr int kWave = 64;   // CDNA wavefront

// Device-side element counts (the `n_device` argument of the per-sample entry points, include/nsx.h).
// First statement of a per-sample kernel: shrink the element count (and the tile count derived from it) to the value in
// device memory, if one is attached; nothing to do -> the whole grid returns.
#define NSX_DEVICE_COUNT(B, n_tiles, per_tile, n_dev)                                              \
    do {                                                                                           \
        if (n_dev) {                                                                               \
            const int64_t n__ = *(n_dev);                                                          \
            if (n__ < (B)) { (B) = n__ < 0 ? 0 : n__; (n_tiles) = ((B) + (per_tile) - 1) / (per_tile); } \
        }                                                                                          \
        if ((B) <= 0) return;                                                                      \
    } while (0)

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half2_t as_half2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ float dot2(uint32_t a, half2_t b, float c) {
    // v_dot2c_f32_f16: c + a.lo*b.lo + a.hi*b.hi, fp32 accumulate
    return __builtin_amdgcn_fdot2(as_half2(a), b, c, false);
}

// exact idx % size for any uint32 idx, size >= 2 (dense hash-grid levels; size is level-uniform)
__device__ __forceinline__ uint32_t umod(uint32_t idx, uint32_t size, float inv_size) {
    uint32_t q = (uint32_t)(__uint2float_rz(idx) * inv_size);
    int32_t r = (int32_t)(idx - q * size);
    if (r < 0) r += (int32_t)size;
    if ((uint32_t)r >= size) r -= (int32_t)size;
    return (uint32_t)r;
}

// Several sample sets in one launch of the HashEnsemble kernels (hash_ensemble.hip: ens_*_sources_kernel; the caller is
// level_parallel.hip).  Source j's arrays start j * stride BYTES behind source 0's; B[j] is its capacity (the valid count
// is read from its device counter), plane_base[j] its first gradient plane, plane_base[j + 1] - plane_base[j] its code rows.
struct EnsSources {
    int64_t x_stride, slot_stride, count_stride, code_stride, out_stride, dout_stride, dx_stride, rows_stride;
    int64_t csum_floats;                        // block partials of the code sums per source (set by the launcher)
    int64_t B[NSX_MAX_LEVELS];
    int32_t plane_base[NSX_MAX_LEVELS + 1];
};
int ens_fwd_sources(int n_sources, const EnsSources& src, const void* x, const nsx_half* tables, int H, const nsx_grid_geom* g,
                    const void* code, int64_t code_row_stride, const void* code_slot, const float* window, void* out,
                    const void* n_dev, hipStream_t st);
// G may be NULL (no table gradient: the gather half alone); csum_part holds csum_capacity floats
int ens_bwd_sources(int n_sources, EnsSources& src, const void* x, const nsx_half* tables, int H, const nsx_grid_geom* g,
                    const void* code, int64_t code_row_stride, const void* code_slot, const float* window, const void* dout,
                    float* G, void* dx, float* nonfinite, const void* n_dev, float* csum_part, int64_t csum_capacity,
                    void* dcode_rows, hipStream_t st);

inline int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace nsx

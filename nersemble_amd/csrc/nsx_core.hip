// nsx_core.hip -- version, error state, host-side grid geometry.
#include "nsx_common.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstring>

namespace nsx {

// launch-shape options (include/nsx.h): value, lowest and highest admissible value
static std::atomic<int> g_opt[NSX_OPT_COUNT] = {{5}, {2}, {2}, {1}};
static const int g_opt_lo[NSX_OPT_COUNT] = {1, 1, 1, 0}, g_opt_hi[NSX_OPT_COUNT] = {8, 8, 8, 1};

int option(int which) { return g_opt[which].load(std::memory_order_relaxed); }

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NSX_ERR_HIP;
}

}  // namespace nsx

extern "C" {

int nsx_version(void) { return NSX_VERSION; }

const char* nsx_last_error(void) { return nsx::g_err; }

int nsx_set_option(int option, int value) {
    NSX_REQUIRE(option >= 0 && option < NSX_OPT_COUNT, "nsx_set_option: unknown option %d", option);
    NSX_REQUIRE(value >= nsx::g_opt_lo[option] && value <= nsx::g_opt_hi[option], "nsx_set_option: option %d takes %d..%d (got %d)",
                option, nsx::g_opt_lo[option], nsx::g_opt_hi[option], value);
    nsx::g_opt[option].store(value, std::memory_order_relaxed);
    return NSX_OK;
}

int nsx_get_option(int option) {
    NSX_REQUIRE(option >= 0 && option < NSX_OPT_COUNT, "nsx_get_option: unknown option %d", option);
    return nsx::option(option);
}

int nsx_padded_grids(int H) {
    int p = 1;
    while (p < H) p <<= 1;
    return p;
}

// Geometry of the multi-resolution hash grid.  Same arithmetic (fp32 exp2f/log2f/ceilf, uint32
// sizes, 8-entry alignment, hashmap cap) as the tcnn HashGrid constructor the reference reaches
// through hash_ensemble.py:41-50.
int nsx_grid_geometry(int n_levels, float per_level_scale, int base_resolution,
                      int log2_hashmap_size, nsx_grid_geom* out) {
    NSX_REQUIRE(out != nullptr, "nsx_grid_geometry: out is NULL");
    NSX_REQUIRE(n_levels >= 1 && n_levels <= NSX_MAX_LEVELS, "nsx_grid_geometry: n_levels %d not in [1,%d]",
                n_levels, NSX_MAX_LEVELS);
    NSX_REQUIRE(log2_hashmap_size >= 3 && log2_hashmap_size <= 28, "nsx_grid_geometry: log2_hashmap_size %d",
                log2_hashmap_size);
    NSX_REQUIRE(base_resolution >= 2, "nsx_grid_geometry: base_resolution %d", base_resolution);
    memset(out, 0, sizeof(*out));
    out->n_levels = n_levels;
    out->log2_hashmap_size = log2_hashmap_size;
    out->base_resolution = base_resolution;
    out->per_level_scale = per_level_scale;
    const float l2s = log2f(per_level_scale);
    const uint32_t cap = 1u << log2_hashmap_size;
    uint64_t offset = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float scale = exp2f((float)l * l2s) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xffffffffu / 2u;
        uint32_t n = (powf((float)res, 3.0f) > (float)max_params) ? max_params : res * res * res;
        n = ((n + 7u) / 8u) * 8u;
        if (n > cap) n = cap;
        // dense iff the 3-D stride walk completes with stride <= size (tcnn grid_index)
        uint64_t stride = 1;
        for (int d = 0; d < 3 && stride <= n; ++d) stride *= res;
        out->scale[l] = scale;
        out->res[l] = res;
        out->size[l] = n;
        out->hashed[l] = (n < stride) ? 1u : 0u;
        out->offset[l] = (uint32_t)offset;
        offset += n;
        NSX_REQUIRE(offset < (1ull << 32), "nsx_grid_geometry: table too large");
        if (out->hashed[l]) NSX_REQUIRE((n & (n - 1)) == 0, "nsx_grid_geometry: hashed level %d size not pow2", l);
    }
    out->offset[n_levels] = (uint32_t)offset;
    return NSX_OK;
}

}  // extern "C"

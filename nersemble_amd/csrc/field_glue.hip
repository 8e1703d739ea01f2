// field_glue.hip -- the elementwise glue around the field kernels, fused: sample positions, scene-box
// normalisation + in-box selector, and the trunc_exp density epilogue.
//
// Replaces the ~30 tiny torch launches per pass of
//   Frustums.get_positions                      origins + directions * (starts + ends) / 2 (+ offsets)      (nerfstudio)
//   VolumetricSampler.get_sigma_fn              origins[ray_idx] + dirs[ray_idx] * (t0 + t1)[:, None] / 2    (nerfstudio)
//   SceneBox.get_normalized_positions + selector + mask   nersemble_nerfacto_field.py:257, :268-269
//   split / trunc_exp / selector mask           nersemble_nerfacto_field.py:286-293
// Arithmetic follows torch's op order with contraction off, so results are bit-identical to the unfused form.
#include "nsx_common.h"
#pragma clang fp contract(off)

namespace nsx {

struct Box { float lo[3], ext[3]; };

// pos = o + (d * (t0 + t1)) / 2 (+ off);  o/d either per sample [S][3] or gathered through ray_idx from [R][3]
__global__ __launch_bounds__(256) void sample_positions_kernel(
    const float* __restrict__ o, const float* __restrict__ d, const int64_t* __restrict__ ray_idx,
    const float* __restrict__ t0, const float* __restrict__ t1, const float* __restrict__ off, int64_t S, Box box,
    float* __restrict__ pos_world, float* __restrict__ pos_n, uint8_t* __restrict__ sel,
    const int64_t* __restrict__ n_dev) {
    int64_t unused_tiles = 0;
    NSX_DEVICE_COUNT(S, unused_tiles, 1, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = ray_idx ? ray_idx[i] : i;
        const float tt = t0 ? (t0[i] + t1[i]) : 0.f;
        float p[3];
        bool inside = true;
        // pos_world: the sample position itself; the offsets (a deformation, in normalised units added to the world
        // position exactly as nersemble_instant_ngp.py:257-259 does) only enter the normalised output -- so one launch
        // serves a caller that needs both (the deformation field's backward wants the undeformed position)
        float raw[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = t0 ? o[r * 3 + a] + (d[r * 3 + a] * tt) / 2.0f : o[r * 3 + a];
            raw[a] = v;
            if (off) v = v + off[i * 3 + a];
            p[a] = v;
        }
        if (pos_world) { pos_world[i * 3] = raw[0]; pos_world[i * 3 + 1] = raw[1]; pos_world[i * 3 + 2] = raw[2]; }
        if (pos_n) {
            float q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                q[a] = (p[a] - box.lo[a]) / box.ext[a];
                inside = inside && (q[a] > 0.0f) && (q[a] < 1.0f);
            }
            const float m = inside ? 1.0f : 0.0f;
            pos_n[i * 3] = q[0] * m; pos_n[i * 3 + 1] = q[1] * m; pos_n[i * 3 + 2] = q[2] * m;
            if (sel) sel[i] = inside ? 1 : 0;
        }
    }
}

// d pos = g * sel / ext
__global__ __launch_bounds__(256) void normalise_bwd_kernel(const float* __restrict__ g, const uint8_t* __restrict__ sel,
                                                            int64_t S, Box box, float* __restrict__ dpos,
                                                            const int64_t* __restrict__ n_dev) {
    int64_t unused_tiles = 0;
    NSX_DEVICE_COUNT(S, unused_tiles, 1, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = sel[i] ? 1.0f : 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) dpos[i * 3 + a] = (g[i * 3 + a] * m) / box.ext[a];
    }
}

// density = exp(float(h0)) * sel
__global__ __launch_bounds__(256) void density_fwd_kernel(const half_t* __restrict__ base, int64_t stride,
                                                          const uint8_t* __restrict__ sel, int64_t S,
                                                          float* __restrict__ density, const int64_t* __restrict__ n_dev) {
    int64_t unused_tiles = 0;
    NSX_DEVICE_COUNT(S, unused_tiles, 1, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x)
        density[i] = expf((float)base[i * stride]) * (sel[i] ? 1.0f : 0.0f);
}

// trunc_exp backward: d h0 = g * sel * exp(clamp(h0, -15, 15)), written as fp16 into column 0 of a zeroed [S][stride]
__global__ __launch_bounds__(256) void density_bwd_kernel(const half_t* __restrict__ base, int64_t stride,
                                                          const uint8_t* __restrict__ sel, const float* __restrict__ g,
                                                          int64_t S, half_t* __restrict__ dbase,
                                                          const int64_t* __restrict__ n_dev) {
    int64_t unused_tiles = 0;
    NSX_DEVICE_COUNT(S, unused_tiles, 1, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        const float h = (float)base[i * stride];
        const float gd = g[i] * (sel[i] ? 1.0f : 0.0f);
        dbase[i * stride] = (half_t)(gd * expf(fminf(fmaxf(h, -15.0f), 15.0f)));
    }
}

// dst_a[i][:] = src_a[index[i]][:] for up to NSX_MAX_GATHER arrays in ONE launch (rows of 4-byte words; 16-byte pieces
// when every row is a multiple of 16 B)
struct GatherArgs {
    const uint8_t* src[NSX_MAX_GATHER];
    uint8_t* dst[NSX_MAX_GATHER];
    int words[NSX_MAX_GATHER];       // row length in pieces (uint32 or uint4)
    int vec[NSX_MAX_GATHER];         // 1: uint4 pieces, 0: uint32 pieces
    int via[NSX_MAX_GATHER];         // 1: the row is via[index[i]] (an array indexed by what the index points at)
    int n_arrays;
};

// Under a device-side count (n_device) rows [*n_dev, n) of every destination are written as ZEROS by the
// same launch: the caller's arrays keep the capacity, and nothing downstream ever sees uninitialised rows (this used to be
// a torch fill in front of every call).
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs A, const int64_t* __restrict__ index, int64_t n,
                                                          const int64_t* __restrict__ n_dev,
                                                          const int64_t* __restrict__ via) {
    int64_t n_valid = n;
    if (n_dev) {
        const int64_t c = *n_dev;
        if (c < n_valid) n_valid = c < 0 ? 0 : c;
    }
    const int a = blockIdx.y;
    const int64_t words = A.words[a];
    const int64_t total = n * words, valid = n_valid * words;
    const bool hop = A.via[a] != 0;
    if (A.vec[a]) {
        const uint4* src = reinterpret_cast<const uint4*>(A.src[a]);
        uint4* dst = reinterpret_cast<uint4*>(A.dst[a]);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t row = i / words, w = i - row * words;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (i < valid) {
                int64_t at = index[row];
                if (hop) at = via[at];
                val = src[at * words + w];
            }
            dst[i] = val;
        }
    } else {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(A.src[a]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(A.dst[a]);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t row = i / words, w = i - row * words;
            uint32_t val = 0u;
            if (i < valid) {
                int64_t at = index[row];
                if (hop) at = via[at];
                val = src[at * words + w];
            }
            dst[i] = val;
        }
    }
}

static Box make_box(const float* aabb) {
    Box b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = aabb ? aabb[a] : 0.f; b.ext[a] = aabb ? aabb[3 + a] - aabb[a] : 1.f; }
    return b;
}
static unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ray r: rint(times[r] * (T - 1)) must be the timestep of the code row the ray was given (nsx_check_code_rows)
__global__ __launch_bounds__(256) void check_code_rows_kernel(const float* __restrict__ times, const int32_t* __restrict__ slots,
                                                             int64_t R, const int32_t* __restrict__ row_timesteps, int n_rows,
                                                             int n_timesteps, int32_t* __restrict__ flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int32_t s = slots[r];
    bool bad = s < 0 || s >= n_rows;
    if (!bad) bad = row_timesteps[s] != (int32_t)rintf(times[r] * (float)(n_timesteps - 1));   // torch: (t * (T - 1)).round().int()
    if (bad) atomicOr(flag, 1);
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_sample_positions(const float* origins, const float* directions, const int64_t* ray_indices,
                         const float* t_starts, const float* t_ends, const float* offsets, int64_t S,
                         const float* aabb_host, float* pos_world, float* pos_normalised, uint8_t* selector,
                         const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_sample_positions: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(origins, "nsx_sample_positions: NULL origins");
    NSX_REQUIRE((t_starts == nullptr) == (t_ends == nullptr), "nsx_sample_positions: t_starts/t_ends must come together");
    NSX_REQUIRE(!t_starts || directions, "nsx_sample_positions: directions required with t_starts");
    NSX_REQUIRE(pos_world || pos_normalised, "nsx_sample_positions: no output requested");
    NSX_REQUIRE(!pos_normalised || aabb_host, "nsx_sample_positions: aabb required for normalised positions");
    hipLaunchKernelGGL(sample_positions_kernel, dim3(grid_for(S)), dim3(256), 0, (hipStream_t)stream, origins, directions,
                       ray_indices, t_starts, t_ends, offsets, S, make_box(aabb_host), pos_world, pos_normalised, selector,
                       n_device);
    NSX_LAUNCH_CHECK("nsx_sample_positions launch");
    return NSX_OK;
}

static int gather_rows(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                       const int64_t* index, const int64_t* via, const uint8_t* use_via, int64_t n, const int64_t* n_device,
                       void* stream);

int nsx_gather_rows(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                    const int64_t* index, int64_t n, const int64_t* n_device, void* stream) {
    return gather_rows(n_arrays, srcs, row_bytes, dsts, index, nullptr, nullptr, n, n_device, stream);
}

int nsx_gather_rows_via(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                        const int64_t* index, const int64_t* via, const uint8_t* use_via_host, int64_t n,
                        const int64_t* n_device, void* stream) {
    NSX_REQUIRE(via && use_via_host, "nsx_gather_rows_via: NULL argument");
    return gather_rows(n_arrays, srcs, row_bytes, dsts, index, via, use_via_host, n, n_device, stream);
}

static int gather_rows(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                       const int64_t* index, const int64_t* via, const uint8_t* use_via, int64_t n, const int64_t* n_device,
                       void* stream) {
    NSX_REQUIRE(n >= 0, "nsx_gather_rows: negative row count");
    NSX_REQUIRE(n_arrays >= 1 && n_arrays <= NSX_MAX_GATHER, "nsx_gather_rows: n_arrays=%d not in [1,%d]", n_arrays,
                NSX_MAX_GATHER);
    if (n == 0) return NSX_OK;
    NSX_REQUIRE(srcs && row_bytes && dsts && index, "nsx_gather_rows: NULL argument");
    GatherArgs A;
    A.n_arrays = n_arrays;
    int64_t max_pieces = 1;
    for (int a = 0; a < n_arrays; ++a) {
        NSX_REQUIRE(srcs[a] && dsts[a] && row_bytes[a] > 0 && row_bytes[a] % 4 == 0,
                    "nsx_gather_rows: array %d: NULL pointer or row size %lld not a positive multiple of 4 bytes", a,
                    (long long)row_bytes[a]);
        A.src[a] = reinterpret_cast<const uint8_t*>(srcs[a]);
        A.dst[a] = reinterpret_cast<uint8_t*>(dsts[a]);
        const bool vec = row_bytes[a] % 16 == 0 && (reinterpret_cast<uintptr_t>(srcs[a]) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(dsts[a]) & 15) == 0;
        A.vec[a] = vec ? 1 : 0;
        A.via[a] = (use_via && use_via[a]) ? 1 : 0;
        A.words[a] = (int)(row_bytes[a] / (vec ? 16 : 4));
        if (n * A.words[a] > max_pieces) max_pieces = n * A.words[a];
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(max_pieces), n_arrays), dim3(256), 0, (hipStream_t)stream, A, index, n,
                       n_device, via);
    NSX_LAUNCH_CHECK("nsx_gather_rows launch");
    return NSX_OK;
}

int nsx_check_code_rows(const float* ray_times, const int32_t* ray_slots, int64_t R, const int32_t* row_timesteps,
                        int n_code_rows, int n_timesteps, int32_t* flag, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_check_code_rows: negative ray count");
    if (R == 0) return NSX_OK;
    NSX_REQUIRE(ray_times && ray_slots && row_timesteps && flag, "nsx_check_code_rows: NULL argument");
    NSX_REQUIRE(n_code_rows >= 1 && n_timesteps >= 1, "nsx_check_code_rows: n_code_rows=%d n_timesteps=%d", n_code_rows,
                n_timesteps);
    hipLaunchKernelGGL(check_code_rows_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ray_times,
                       ray_slots, R, row_timesteps, n_code_rows, n_timesteps, flag);
    NSX_LAUNCH_CHECK("nsx_check_code_rows launch");
    return NSX_OK;
}

int nsx_normalise_bwd(const float* grad_pos_normalised, const uint8_t* selector, int64_t S, const float* aabb_host,
                      float* grad_pos_world, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_normalise_bwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(grad_pos_normalised && selector && aabb_host && grad_pos_world, "nsx_normalise_bwd: NULL argument");
    hipLaunchKernelGGL(normalise_bwd_kernel, dim3(grid_for(S)), dim3(256), 0, (hipStream_t)stream, grad_pos_normalised,
                       selector, S, make_box(aabb_host), grad_pos_world, n_device);
    NSX_LAUNCH_CHECK("nsx_normalise_bwd launch");
    return NSX_OK;
}

int nsx_density_fwd(const nsx_half* base_out, int64_t stride, const uint8_t* selector, int64_t S, float* density,
                    const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_density_fwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(base_out && selector && density, "nsx_density_fwd: NULL argument");
    hipLaunchKernelGGL(density_fwd_kernel, dim3(grid_for(S)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(base_out), stride, selector, S, density, n_device);
    NSX_LAUNCH_CHECK("nsx_density_fwd launch");
    return NSX_OK;
}

int nsx_density_bwd(const nsx_half* base_out, int64_t stride, const uint8_t* selector, const float* grad_density,
                    int64_t S, nsx_half* grad_base_out_zeroed, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0, "nsx_density_bwd: negative sample count");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(base_out && selector && grad_density && grad_base_out_zeroed, "nsx_density_bwd: NULL argument");
    hipLaunchKernelGGL(density_bwd_kernel, dim3(grid_for(S)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(base_out), stride, selector, grad_density, S,
                       reinterpret_cast<half_t*>(grad_base_out_zeroed), n_device);
    NSX_LAUNCH_CHECK("nsx_density_bwd launch");
    return NSX_OK;
}

}  // extern "C"

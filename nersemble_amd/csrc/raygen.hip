// raygen.hip -- pinhole ray generation on the device: the step in FRONT of the hot path.
//
// Replaces what the reference's datamanager reaches through nerfstudio's RayGenerator -> Cameras.generate_rays
// (NeRSembleVanillaDataManager.next_train, datamanager/nersemble_datamanager.py:76-81; cameras are perspective with all
// distortion parameters zero, dataparser/nersemble_dataparser.py:237-244): for R (camera, y, x) pixel coordinates
//   d_cam  = ((x - cx) / fx, -(y - cy) / fy, -1)            OpenGL camera frame, pixel centre already in (y, x)
//   d      = normalise(R_c2w d_cam)                          unit world direction
//   origin = t_c2w
//   pixel_area = |d - d(y, x + 1)| * |d - d(y + 1, x)|       nerfstudio's neighbouring-pixel differences
// ~35 torch launches (3 x (stack, einsum, norm, div), 2 x (sub, norm), gathers of 5 intrinsics + the pose) -> 1.
// One thread per ray, 40 B in / 28 B out per ray: launch-bound, not bandwidth-bound.  fp32, contraction off: the
// operation order below is the oracle's (oracle/cameras.py).
#include "nsx_common.h"
#pragma clang fp contract(off)

namespace nsx {

__device__ __forceinline__ void unit_dir(const float* __restrict__ m, float fx, float fy, float cx, float cy, float y,
                                         float x, float out[3]) {
    const float d0 = (x - cx) / fx, d1 = -((y - cy) / fy), d2 = -1.0f;
    float w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = (m[i * 4 + 0] * d0 + m[i * 4 + 1] * d1) + m[i * 4 + 2] * d2;
    const float n = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = w[i] / n;
}

__global__ __launch_bounds__(256) void generate_rays_kernel(
    const float* __restrict__ c2w, const float* __restrict__ fx, const float* __restrict__ fy,
    const float* __restrict__ cx, const float* __restrict__ cy, int64_t n_cameras, const int64_t* __restrict__ cams,
    const float* __restrict__ ys, const float* __restrict__ xs, int64_t R, float* __restrict__ origins,
    float* __restrict__ directions, float* __restrict__ pixel_area) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = cams[i];
        c = c < 0 ? 0 : (c >= n_cameras ? n_cameras - 1 : c);      // (validated on the host where the indices live there)
        const float* m = c2w + c * 12;
        const float f0 = fx[c], f1 = fy[c], c0 = cx[c], c1 = cy[c];
        const float y = ys[i], x = xs[i];
        float d[3], dxn[3], dyn[3];
        unit_dir(m, f0, f1, c0, c1, y, x, d);
        unit_dir(m, f0, f1, c0, c1, y, x + 1.0f, dxn);
        unit_dir(m, f0, f1, c0, c1, y + 1.0f, x, dyn);
        float ax = 0.f, ay = 0.f;
        {
            const float a0 = d[0] - dxn[0], a1 = d[1] - dxn[1], a2 = d[2] - dxn[2];
            ax = sqrtf((a0 * a0 + a1 * a1) + a2 * a2);
            const float b0 = d[0] - dyn[0], b1 = d[1] - dyn[1], b2 = d[2] - dyn[2];
            ay = sqrtf((b0 * b0 + b1 * b1) + b2 * b2);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            origins[i * 3 + a] = m[a * 4 + 3];
            directions[i * 3 + a] = d[a];
        }
        if (pixel_area) pixel_area[i] = ax * ay;
    }
}

}  // namespace nsx

using namespace nsx;

extern "C" int nsx_generate_rays(const float* camera_to_worlds, const float* fx, const float* fy, const float* cx,
                                 const float* cy, int64_t n_cameras, const int64_t* camera_indices, const float* ys,
                                 const float* xs, int64_t R, float* origins, float* directions, float* pixel_area,
                                 void* stream) {
    NSX_REQUIRE(R >= 0 && n_cameras >= 1, "nsx_generate_rays: bad sizes (R=%lld, cameras=%lld)", (long long)R,
                (long long)n_cameras);
    if (R == 0) return NSX_OK;
    NSX_REQUIRE(camera_to_worlds && fx && fy && cx && cy && camera_indices && ys && xs && origins && directions,
                "nsx_generate_rays: NULL argument");
    int64_t blocks = (R + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, camera_to_worlds,
                       fx, fy, cx, cy, n_cameras, camera_indices, ys, xs, R, origins, directions, pixel_area);
    NSX_LAUNCH_CHECK("nsx_generate_rays launch");
    return NSX_OK;
}

// losses.hip -- the per-ray loss terms, their reduction and the training metrics of one step in two small kernels.
//
// Replaces the ~110 element-wise / reduction launches (and their autograd mirror) that the reference issues on
// [4096]-ray tensors every iteration:
//   get_masked_rgb_loss, get_alpha_loss, get_depth_loss        models/base.py:90-133, 204-215
//   the reductions of the distortion / empty / near losses     models/base.py:136-202, 224-249
//   get_metrics_dict (psnr, psnr_masked, num_samples_per_batch) nersemble_instant_ngp.py:409-422
//   functools.reduce(torch.add, loss_dict.values())            nersemble_trainer.py:184
// All arithmetic is fp32 (the reference runs these under autocast, where MSE / sums are fp32 as well).
// A step is launch-bound here, not bandwidth-bound: the point is 2 launches instead of ~110.
#include "nsx_common.h"

namespace nsx {

constexpr int NL_UNROLL = 8;       // rays per thread and round of ray_losses_fwd_kernel
constexpr int NL_THREADS = 256;    // one block; 4 waves find a slot beside co-running kernels (16 waited: 20-33 us for 5 us of work)

struct LossCfg {
    int use_masked_rgb;
    float alpha_thr, l_alpha, l_depth, l_dist, l_empty, l_near;
};

__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// partial sums: 0 sq_all | 1 sq_masked 2 n_masked | 3 sq_psnr_mask 4 n_psnr_mask | 5 abs_alpha 6 n_bg |
//               7 sq_depth 8 n_depth | 9..13 sample-loss columns | 14 n_samples | 15 (max) last ray with samples + 1
constexpr int NP = 16;

__global__ __launch_bounds__(NL_THREADS) void ray_losses_fwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ acc, const float* __restrict__ depth,
    const float* __restrict__ image, const uint8_t* __restrict__ alpha_map, const float* __restrict__ depth_t,
    const float* __restrict__ per_ray, const int64_t* __restrict__ packed, int64_t R, LossCfg cfg,
    float* __restrict__ out) {
    __shared__ float red[NP][NL_THREADS / 64];
    float p[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) p[k] = 0.f;
    // NL_UNROLL rays per thread and round: every load of the round is unconditional (clamped row, uniform NULL tests) and
    // issued before the first use -- one memory latency per round instead of four dependent ones per ray.  The sums are
    // taken in the same per-thread order as a plain loop over r (bit-identical results).
    for (int64_t r0 = threadIdx.x; r0 < R; r0 += (int64_t)NL_UNROLL * NL_THREADS) {
        float im[NL_UNROLL][3], pr[NL_UNROLL][3], pray[NL_UNROLL][5], accv[NL_UNROLL], dt[NL_UNROLL], dp[NL_UNROLL];
        int am[NL_UNROLL];
        int64_t ns[NL_UNROLL];
#pragma unroll
        for (int u = 0; u < NL_UNROLL; ++u) {
            const int64_t r = r0 + (int64_t)u * NL_THREADS;
            const int64_t rr = r < R ? r : R - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) { im[u][c] = image[rr * 3 + c]; pr[u][c] = rgb[rr * 3 + c]; }
            am[u] = alpha_map ? (int)alpha_map[rr] : 0;
            accv[u] = alpha_map ? acc[rr] : 0.f;
            dt[u] = depth_t ? depth_t[rr] : 0.f;
            dp[u] = depth_t ? depth[rr] : 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) pray[u][k] = per_ray ? per_ray[rr * 5 + k] : 0.f;
            ns[u] = packed ? packed[2 * rr + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < NL_UNROLL; ++u) {
            const int64_t r = r0 + (int64_t)u * NL_THREADS;
            if (r >= R) break;
            float sq = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = im[u][c] - pr[u][c];
                sq += d * d;
            }
            p[0] += sq;
            if (alpha_map) {
                const float a = (float)am[u] / 255.0f;
                if (a > cfg.alpha_thr) { p[1] += sq * (1.0f / 3.0f); p[2] += 1.f; }
                if (am[u] > 127) { p[3] += sq * (1.0f / 3.0f); p[4] += 1.f; }
                if (a < 1.0f) { p[5] += fabsf(accv[u] - a); p[6] += 1.f; }
            }
            if (depth_t) {
                const float t = dt[u];
                if (t > 0.f) { const float d = t - dp[u]; p[7] += d * d; p[8] += 1.f; }
            }
            if (per_ray) {
#pragma unroll
                for (int k = 0; k < 5; ++k) p[9 + k] += pray[u][k];
            }
            if (packed) {
                const int64_t n = ns[u];
                p[14] += (float)n;
                if (n > 0) p[15] = fmaxf(p[15], (float)(r + 1));
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const float v = (k == 15) ? wave_max(p[k]) : wave_add(p[k]);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float s[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        float v = red[k][0];
        for (int w = 1; w < NL_THREADS / 64; ++w) v = (k == 15) ? fmaxf(v, red[k][w]) : v + red[k][w];
        s[k] = v;
    }
    const float mse = s[0] / (3.0f * (float)R);
    const bool masked = cfg.use_masked_rgb && alpha_map;
    const float rgb_loss = masked ? s[1] / fmaxf(s[2], 1.0f) : mse;
    const float alpha_loss = (alpha_map && cfg.l_alpha > 0.f) ? cfg.l_alpha * (s[5] / fmaxf(s[6], 1.0f)) : 0.f;
    const float depth_loss = (depth_t && cfg.l_depth > 0.f) ? cfg.l_depth * (s[7] / fmaxf(s[8], 1.0f)) : 0.f;
    const float n_eff = fmaxf(s[15], 1.0f);                 // torch_efficient_distloss: ray_id.max() + 1
    float dist = 0.f, empty = 0.f, near = 0.f;
    if (per_ray) {
        dist = cfg.l_dist * (s[9] / n_eff);
        empty = cfg.l_empty * (s[10] / fmaxf(s[11], 1.0f));
        near = cfg.l_near * (s[12] / fmaxf(s[13], 1.0f));
    }
    out[NSX_LOSS_RGB] = rgb_loss;
    out[NSX_LOSS_ALPHA] = alpha_loss;
    out[NSX_LOSS_DEPTH] = depth_loss;
    out[NSX_LOSS_DIST] = dist;
    out[NSX_LOSS_EMPTY] = empty;
    out[NSX_LOSS_NEAR] = near;
    // same association as reduce(torch.add, [rgb, alpha, dist, empty, near, depth]) (absent terms are exact zeros)
    out[NSX_LOSS_TOTAL] = ((((rgb_loss + alpha_loss) + dist) + empty) + near) + depth_loss;
    out[NSX_LOSS_PSNR] = 10.0f * log10f(1.0f / mse);
    out[NSX_LOSS_PSNR_MASKED] = alpha_map ? 10.0f * log10f(1.0f / (s[3] / fmaxf(s[4], 1.0f))) : 0.f;
    out[NSX_LOSS_NUM_SAMPLES] = s[14];
#pragma unroll
    for (int k = 0; k < 5; ++k) out[NSX_LOSS_SAMPLE_SUMS + k] = s[9 + k];
    out[NSX_LOSS_SAMPLE_SUMS + 5] = s[2];                   // denominators kept for the backward pass
    out[NSX_LOSS_SAMPLE_SUMS + 6] = s[6];
    out[NSX_LOSS_SAMPLE_SUMS + 7] = s[8];
    out[NSX_LOSS_SAMPLE_SUMS + 8] = n_eff;
}

__global__ __launch_bounds__(256) void ray_losses_bwd_kernel(
    const float* __restrict__ rgb, const float* __restrict__ acc, const float* __restrict__ depth,
    const float* __restrict__ image, const uint8_t* __restrict__ alpha_map, const float* __restrict__ depth_t,
    int64_t R, LossCfg cfg, float n_rays, const float* __restrict__ out, const float* __restrict__ g,
    float* __restrict__ g_rgb, float* __restrict__ g_acc, float* __restrict__ g_depth, float* __restrict__ g3) {
    const float gt = g[NSX_LOSS_TOTAL];
    const float G_rgb = g[NSX_LOSS_RGB] + gt, G_alpha = g[NSX_LOSS_ALPHA] + gt, G_depth = g[NSX_LOSS_DEPTH] + gt;
    const float n_mask = out[NSX_LOSS_SAMPLE_SUMS + 5], n_bg = out[NSX_LOSS_SAMPLE_SUMS + 6];
    const float n_d = out[NSX_LOSS_SAMPLE_SUMS + 7], n_eff = out[NSX_LOSS_SAMPLE_SUMS + 8];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0 && g3) {
        // coefficients for nsx_sample_losses_bwd, which applies 1/n_rays to the distortion term itself
        g3[0] = (g[NSX_LOSS_DIST] + gt) * cfg.l_dist * (n_rays / n_eff);
        g3[1] = (g[NSX_LOSS_EMPTY] + gt) * cfg.l_empty;
        g3[2] = (g[NSX_LOSS_NEAR] + gt) * cfg.l_near;
    }
    if (r >= R) return;
    const bool masked = cfg.use_masked_rgb && alpha_map;
    const float a = alpha_map ? (float)alpha_map[r] / 255.0f : 1.0f;
    float c_rgb;
    if (masked) c_rgb = (a > cfg.alpha_thr) ? G_rgb * (2.0f / 3.0f) / fmaxf(n_mask, 1.0f) : 0.f;
    else c_rgb = G_rgb * 2.0f / (3.0f * (float)R);
#pragma unroll
    for (int c = 0; c < 3; ++c) g_rgb[r * 3 + c] = c_rgb * (rgb[r * 3 + c] - image[r * 3 + c]);
    float ga = 0.f;
    if (alpha_map && cfg.l_alpha > 0.f && a < 1.0f) {
        const float d = acc[r] - a;
        ga = G_alpha * cfg.l_alpha * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / fmaxf(n_bg, 1.0f);
    }
    g_acc[r] = ga;
    float gd = 0.f;
    if (depth_t && cfg.l_depth > 0.f) {
        const float t = depth_t[r];
        if (t > 0.f) gd = G_depth * cfg.l_depth * 2.0f * (depth[r] - t) / fmaxf(n_d, 1.0f);
    }
    g_depth[r] = gd;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_ray_losses_fwd(const float* rgb, const float* accumulation, const float* depth, const float* image,
                       const uint8_t* alpha_map, const float* depth_targets, const float* per_ray_sample,
                       const int64_t* packed_info, int64_t R, int use_masked_rgb, float alpha_mask_threshold,
                       float lambda_alpha, float lambda_depth, float lambda_dist, float lambda_empty, float lambda_near,
                       float* out, void* stream) {
    NSX_REQUIRE(R >= 1, "nsx_ray_losses_fwd: needs at least one ray (got %lld)", (long long)R);
    NSX_REQUIRE(rgb && accumulation && depth && image && out, "nsx_ray_losses_fwd: NULL argument");
    LossCfg cfg{use_masked_rgb, alpha_mask_threshold, lambda_alpha, lambda_depth, lambda_dist, lambda_empty, lambda_near};
    hipLaunchKernelGGL(ray_losses_fwd_kernel, dim3(1), dim3(NL_THREADS), 0, (hipStream_t)stream, rgb, accumulation,
                       depth, image, alpha_map, depth_targets, per_ray_sample, packed_info, R, cfg, out);
    NSX_LAUNCH_CHECK("nsx_ray_losses_fwd launch");
    return NSX_OK;
}

int nsx_ray_losses_bwd(const float* rgb, const float* accumulation, const float* depth, const float* image,
                       const uint8_t* alpha_map, const float* depth_targets, int64_t R, int use_masked_rgb,
                       float alpha_mask_threshold, float lambda_alpha, float lambda_depth, float lambda_dist,
                       float lambda_empty, float lambda_near, int64_t n_rays, const float* out, const float* grad_out,
                       float* grad_rgb, float* grad_accumulation, float* grad_depth, float* sample_grads,
                       void* stream) {
    NSX_REQUIRE(R >= 1 && n_rays >= 1, "nsx_ray_losses_bwd: needs at least one ray");
    NSX_REQUIRE(rgb && accumulation && depth && image && out && grad_out && grad_rgb && grad_accumulation && grad_depth,
                "nsx_ray_losses_bwd: NULL argument");
    LossCfg cfg{use_masked_rgb, alpha_mask_threshold, lambda_alpha, lambda_depth, lambda_dist, lambda_empty, lambda_near};
    hipLaunchKernelGGL(ray_losses_bwd_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb,
                       accumulation, depth, image, alpha_map, depth_targets, R, cfg, (float)n_rays, out, grad_out,
                       grad_rgb, grad_accumulation, grad_depth, sample_grads);
    NSX_LAUNCH_CHECK("nsx_ray_losses_bwd launch");
    return NSX_OK;
}

}  // extern "C"

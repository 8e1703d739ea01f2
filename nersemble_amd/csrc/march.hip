// march.hip -- occupancy-grid ray marching (nerfacc traverse_grids equivalent) for gfx950.
//
// Replaces OccGridEstimator.sampling's native part as called from the reference sampler
// (src/nersemble/nerfstudio/model_components/nersemble_volumetric_sampler.py:95-108): ray/AABB slab test,
// 3-D DDA through the res^3 boolean grid, fixed-step lattice anchored at the (jittered) near plane; a sample
// [t, t+step] is emitted iff its midpoint lies in an occupied voxel.  Two passes (count -> pack -> fill), as
// nerfacc does, with the single host read-back of the total between them.
//
// Integer outputs (per-ray counts, ray indices, cell ids) AND the t values are bit-exact against the CPU
// oracle: everything is plain fp32 with contraction disabled (no fused multiply-add), IEEE division.
//
// Cost model: <= ~700 lattice steps + <= 3*res voxel steps per ray.  One lane per ray (the lattice and DDA
// recurrences are serial fp32 additions that must keep their order to stay bit-exact), so the walk is bound by
// the LATENCY of the occupancy loads, not by bandwidth: the voxel walk runs 8 voxels ahead so that 8 loads are in
// flight per lane, and rays are spread 16 to a wave (256 waves instead of 64) to cut the divergence tail.
#include "nsx_common.h"
#pragma clang fp contract(off)

namespace nsx {

struct Aabb { float v[6]; };

__device__ __forceinline__ bool ray_aabb(const float o[3], const float d[3], const Aabb& bb, float& tmin_o,
                                         float& tmax_o) {
    const float ix = 1.0f / d[0], iy = 1.0f / d[1], iz = 1.0f / d[2];
    float tmin, tmax, tmin_t, tmax_t;
    if (ix >= 0) { tmin = (bb.v[0] - o[0]) * ix; tmax = (bb.v[3] - o[0]) * ix; }
    else         { tmin = (bb.v[3] - o[0]) * ix; tmax = (bb.v[0] - o[0]) * ix; }
    if (iy >= 0) { tmin_t = (bb.v[1] - o[1]) * iy; tmax_t = (bb.v[4] - o[1]) * iy; }
    else         { tmin_t = (bb.v[4] - o[1]) * iy; tmax_t = (bb.v[1] - o[1]) * iy; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (iz >= 0) { tmin_t = (bb.v[2] - o[2]) * iz; tmax_t = (bb.v[5] - o[2]) * iz; }
    else         { tmin_t = (bb.v[5] - o[2]) * iz; tmax_t = (bb.v[2] - o[2]) * iz; }
    if (tmin > tmax_t || tmin_t > tmax) return false;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return false;
    tmin_o = tmin; tmax_o = tmax;
    return true;
}

#ifndef NSX_MARCH_RPB
#define NSX_MARCH_RPB 16
#endif
#ifndef NSX_MARCH_NB
#define NSX_MARCH_NB 8
#endif
constexpr int kRaysPerBlock = NSX_MARCH_RPB;   // one partially filled wave per block: 4096 rays -> 256 waves, one per CU

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// FILL: 0 = count only; 1 = write every sample's start to t0[n] (and its cell id); 2 = the counting pass that also keeps the
// starts of its first `cap` samples (nsx_march_count_stash: t0 = the ray's row of the stash)
template <int FILL>
__device__ __forceinline__ int64_t march_ray(const float o[3], const float d[3], const Aabb& bb,
                                             const uint8_t* __restrict__ binary, int res, float near_plane,
                                             float far_plane, float step, float* __restrict__ t0,
                                             float* __restrict__ t1, int32_t* __restrict__ cells, int cap = 0) {
    const float eps = 1e-6f;
    float tmin, tmax;
    if (!ray_aabb(o, d, bb, tmin, tmax)) return 0;
    const float this_tmin = fmaxf(tmin, near_plane);
    const float this_tmax = fminf(tmax, far_plane);
    if (this_tmin >= this_tmax) return 0;
    float t_last = near_plane;
    int64_t n = 0;
    for (;;) {
        if (t_last + step * 0.5f >= this_tmin) break;
        t_last += step;
    }
    float inv[3], voxel[3], tdist[3], delta[3];
    int cur[3], fin[3], stepi[3], over[3];
    const float ts = this_tmin + eps, te = this_tmax - eps;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        inv[a] = 1.0f / d[a];
        voxel[a] = (bb.v[3 + a] - bb.v[a]) / (float)res;
        const float rs = o[a] + d[a] * ts;
        const float re = o[a] + d[a] * te;
        cur[a] = clampi((int)(((rs - bb.v[a]) / (bb.v[3 + a] - bb.v[a])) * (float)res), 0, res - 1);
        fin[a] = clampi((int)(((re - bb.v[a]) / (bb.v[3 + a] - bb.v[a])) * (float)res), 0, res - 1);
        const int idelta = d[a] > 0 ? 1 : 0;
        const float start = (float)(cur[a] + idelta);
        const float tmx = ((bb.v[a] + ((start * voxel[a]) - rs)) * inv[a]) + this_tmin;
        tdist[a] = (d[a] == 0.0f) ? this_tmax : tmx;
        const float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
        stepi[a] = (int)sf;
        const float dtmp = voxel[a] * inv[a] * sf;
        delta[a] = (d[a] == 0.0f) ? this_tmax : dtmp;
        over[a] = fin[a] + stepi[a];
    }
    // The DDA recurrence (cur / tdist) never depends on the grid's contents, so the voxel walk runs NB voxels ahead
    // of the lattice walk: NB cell ids + exit times are produced first, their occupancy bytes are fetched with NB
    // independent loads (one L2 round trip instead of NB dependent ones -- the walk is latency-bound: one lane per
    // ray), then the lattice steps of those voxels are emitted in order.  Every floating-point recurrence keeps its
    // operation order, so the outputs stay bit-identical to the one-voxel-at-a-time formulation.
    constexpr int NB = NSX_MARCH_NB;
    const float half_step = step * 0.5f;
    bool walking = true;
    while (walking) {
        float t_exit[NB];
        int32_t cell_id[NB];
        int nb = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            t_exit[i] = 0.f;
            cell_id[i] = 0;
            if (walking) {
                const float t_trav = fminf(tdist[0], fminf(tdist[1], tdist[2]));
                t_exit[i] = fminf(t_trav, this_tmax);
                cell_id[i] = (cur[0] * res + cur[1]) * res + cur[2];
                nb = i + 1;
                // branch-free axis step (the three-way branch diverges on every voxel): the chosen axis moves by
                // its step / delta, the others add 0 / +0.0f, which leaves their values bit-identical
                const bool ax0 = tdist[0] < tdist[1] && tdist[0] < tdist[2];
                const bool ax1 = !ax0 && tdist[1] < tdist[2];
                const bool ax2 = !ax0 && !ax1;
                cur[0] += ax0 ? stepi[0] : 0;
                cur[1] += ax1 ? stepi[1] : 0;
                cur[2] += ax2 ? stepi[2] : 0;
                tdist[0] += ax0 ? delta[0] : 0.0f;
                tdist[1] += ax1 ? delta[1] : 0.0f;
                tdist[2] += ax2 ? delta[2] : 0.0f;
                walking = !((ax0 && cur[0] == over[0]) || (ax1 && cur[1] == over[1]) || (ax2 && cur[2] == over[2]));
            }
        }
        // All NB occupancy bytes are folded into one mask BEFORE any sample of this batch is stored: vmcnt retires
        // loads and stores in issue order, so a load consumed after the emit loop of an earlier voxel would also wait
        // for that voxel's stores to be acknowledged (a full L2 round trip per voxel once HBM is busy).
        uint32_t occupied = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) occupied |= (uint32_t)(binary[cell_id[i]] != 0) << i;
        asm volatile("" : "+v"(occupied));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i < nb) {
                // one loop for empty and occupied voxels: both advance the same lattice recurrence, an occupied
                // voxel also emits.  (Leaving as soon as t_next >= t_exit would only skip a test that the next
                // iteration's exit check repeats: fl(t_next + step/2) >= t_next >= t_exit.)
                const float t_trav = t_exit[i];
                const bool emit = (occupied >> i) & 1u;
                for (;;) {
                    if (t_last + half_step >= t_trav) break;
                    const float t_next = t_last + step;
                    if (emit) {
                        if (FILL == 1) {
                            t0[n] = t_last;          // t1[n] = t0[n] + step: written by the caller, coalesced
                            if (cells) cells[n] = cell_id[i];
                        } else if (FILL == 2) {
                            if (n < cap) t0[n] = t_last;
                        }
                        n++;
                    }
                    t_last = t_next;
                }
            }
        }
    }
    return n;
}

__global__ __launch_bounds__(64) void march_count_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, int64_t R, Aabb bb,
                                                         const uint8_t* __restrict__ binary, int res,
                                                         const float* __restrict__ near, float far_plane, float step,
                                                         int64_t* __restrict__ counts) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    counts[r] = march_ray<0>(o, d, bb, binary, res, near[r], far_plane, step, nullptr, nullptr, nullptr);
}

__global__ __launch_bounds__(64) void march_fill_kernel(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d, int64_t R, Aabb bb,
                                                        const uint8_t* __restrict__ binary, int res,
                                                        const float* __restrict__ near, float far_plane, float step,
                                                        const int64_t* __restrict__ packed, float* __restrict__ t0,
                                                        float* __restrict__ t1, int64_t* __restrict__ ray_idx,
                                                        int32_t* __restrict__ cells) {
    // One wave per block: lanes 0..kRaysPerBlock-1 each walk a ray, all 64 lanes share the coalesced passes.
    const int lane = threadIdx.x;
    const int64_t r_first = (int64_t)blockIdx.x * kRaysPerBlock;
    const int64_t r = r_first + lane;
    const bool has_ray = lane < kRaysPerBlock && r < R;
    long long s = 0, cnt = 0;
    if (has_ray) { s = packed[2 * r]; cnt = packed[2 * r + 1]; }
    const int n_rays = (int)((R - r_first) < kRaysPerBlock ? (R - r_first) : kRaysPerBlock);
    // ray indices first: the block's rays own one contiguous run of the output and their counts are already known,
    // so all 64 lanes write it with coalesced stores (one lane per ray would scatter 8-byte stores from a serial loop)
    for (int k = 0; k < n_rays; ++k) {
        const long long sk = __shfl(s, k), nk = __shfl(cnt, k);
        for (long long i = lane; i < nk; i += kWave) ray_idx[sk + i] = r_first + k;
    }
    if (has_ray && cnt != 0) {
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        march_ray<1>(o, d, bb, binary, res, near[r], far_plane, step, t0 + s, t1 + s, cells ? cells + s : nullptr);
    }
    // interval ends: every emitted sample is [t, fl(t + step)] by construction, so the ends are one coalesced
    // read-add-write over the block's run of starts instead of a second scattered store per lattice step
    // (kPost independent loads in flight per lane: a dependent load -> store per iteration would pay the HBM
    // latency once per 64 samples)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const long long run_begin = __shfl(s, 0), run_end = __shfl(s + cnt, n_rays - 1);
    constexpr int kPost = 8;
    for (long long base = run_begin + lane; base < run_end; base += (long long)kWave * kPost) {
        float v[kPost];
#pragma unroll
        for (int u = 0; u < kPost; ++u) {
            const long long i = base + (long long)u * kWave;
            v[u] = i < run_end ? __hip_atomic_load(t0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kPost; ++u) {
            const long long i = base + (long long)u * kWave;
            if (i < run_end) t1[i] = v[u] + step;
        }
    }
}

// The counting pass that keeps what it walks past (nsx_march_count_stash): stash[r][0 .. min(count, cap)) = the starts of ray
// r's samples, *over = 1 if some ray has more than `cap` (its tail is then missing: the caller marches that batch again).
__global__ __launch_bounds__(64) void march_count_stash_kernel(const float* __restrict__ rays_o,
                                                               const float* __restrict__ rays_d, int64_t R, Aabb bb,
                                                               const uint8_t* __restrict__ binary, int res,
                                                               const float* __restrict__ near, float far_plane, float step,
                                                               int64_t* __restrict__ counts, float* __restrict__ stash, int cap,
                                                               int64_t* __restrict__ over) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const int64_t n = march_ray<2>(o, d, bb, binary, res, near[r], far_plane, step, stash + r * (int64_t)cap, nullptr, nullptr,
                                   cap);
    counts[r] = n;
    if (n > cap) *over = 1;
}

// Pass 2 from the stash: one wave per ray copies its starts to their place in the packed arrays; ends and ray indices as
// march_fill_kernel writes them (t1 = fl(t0 + step)).
__global__ __launch_bounds__(256) void march_fill_from_stash_kernel(const float* __restrict__ stash, int cap, int64_t R, float step,
                                                                    const int64_t* __restrict__ packed, float* __restrict__ t0,
                                                                    float* __restrict__ t1, int64_t* __restrict__ ray_idx) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x / kWave);
    if (r >= R) return;
    const long long s = packed[2 * r];
    long long cnt = packed[2 * r + 1];
    if (cnt > cap) cnt = cap;                        // (never, when the caller honoured *over)
    const float* src = stash + r * (int64_t)cap;
    for (long long i = lane; i < cnt; i += kWave) {
        const float v = src[i];
        t0[s + i] = v;
        t1[s + i] = v + step;
        ray_idx[s + i] = r;
    }
}

// packed_info[r] = (exclusive prefix of counts, counts[r]); total written to total[0].  Single block of 4 waves: every
// thread owns kPackItems CONSECUTIVE rays (its loads are independent and issued together, one memory latency per tile of
// 4096 rays instead of one per 64), scans them serially, and the thread totals go through one wave scan + 4 LDS words.
constexpr int kPackThreads = 256, kPackItems = 16;
__global__ __launch_bounds__(kPackThreads) void pack_info_kernel(const int64_t* __restrict__ counts, int64_t R,
                                                                 int64_t* __restrict__ packed, int64_t* __restrict__ total) {
    __shared__ int64_t wsum[kPackThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t carry = 0;                                         // the same value in every thread
    for (int64_t base = 0; base < R; base += (int64_t)kPackThreads * kPackItems) {
        const int64_t first = base + (int64_t)threadIdx.x * kPackItems;
        int64_t c[kPackItems];
#pragma unroll
        for (int k = 0; k < kPackItems; ++k) c[k] = (first + k < R) ? counts[first + k] : 0;
        int64_t sum = 0;
#pragma unroll
        for (int k = 0; k < kPackItems; ++k) sum += c[k];
        int64_t v = sum;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int64_t t = __shfl_up(v, dlt);
            if (lane >= dlt) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int64_t woff = 0, tile_total = 0;
#pragma unroll
        for (int w = 0; w < kPackThreads / 64; ++w) {
            const int64_t s = wsum[w];
            if (w < wave) woff += s;
            tile_total += s;
        }
        int64_t run = carry + woff + (v - sum);
#pragma unroll
        for (int k = 0; k < kPackItems; ++k) {
            if (first + k < R) {
                packed[2 * (first + k)] = run;
                packed[2 * (first + k) + 1] = c[k];
            }
            run += c[k];
        }
        carry += tile_total;
        __syncthreads();                                       // wsum is rewritten by the next tile
    }
    if (threadIdx.x == 0) total[0] = carry;
}

// counts per ray (nerfacc.pack_info).  Ray indices arrive sorted, ~200 samples per ray: each wave merges its runs of
// equal indices and issues ONE atomic per run (correct for unsorted input too, which merely merges less).
__global__ __launch_bounds__(256) void ray_hist_kernel(const int64_t* __restrict__ ray_idx, int64_t S, int64_t R,
                                                       unsigned long long* __restrict__ counts,
                                                       const int64_t* __restrict__ n_dev) {
    int64_t unused_tiles = 0;
    NSX_DEVICE_COUNT(S, unused_tiles, 1, n_dev);
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < S; base += stride) {
        const int64_t i = base + lane;
        const int64_t r = i < S ? ray_idx[i] : -1;
        const int64_t prev = __shfl_up(r, 1);
        const bool head = lane == 0 || prev != r;
        const unsigned long long heads = __ballot(head);
        if (head && r >= 0 && r < R) {
            const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
            const int len = rest ? __ffsll((long long)rest) : 64 - lane;
            atomicAdd(&counts[r], (unsigned long long)len);
        }
    }
}

// Stream compaction of a per-sample mask whose samples are packed per ray, with the kept samples' packed_info already
// known (the visibility kernel counted per ray, nsx_pack_info scanned the counts): one wave per ray writes the ascending
// indices of its visible samples at the ray's offset -- the output of the three-kernel scan compaction (nsx_compact_mask)
// without its passes over the mask, and no histogram of the kept ray indices afterwards.
constexpr int kCompactWaves = 4;
__global__ __launch_bounds__(kCompactWaves * 64) void compact_rays_kernel(
    const uint8_t* __restrict__ mask, const int64_t* __restrict__ packed_all, const int64_t* __restrict__ packed_kept,
    int64_t R, int64_t* __restrict__ kept, const int64_t* __restrict__ total, int64_t* __restrict__ n_kept) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kCompactWaves + (threadIdx.x >> 6);
    if (r == 0 && lane == 0 && n_kept) n_kept[0] = total[0];
    if (r >= R) return;
    const int64_t s = packed_all[2 * r], n = packed_all[2 * r + 1];
    int64_t out = packed_kept[2 * r];
    for (int64_t base = 0; base < n; base += 64) {
        const int64_t i = base + lane;
        const bool v = i < n && mask[s + i] != 0;
        const unsigned long long m = __ballot(v);
        if (v) kept[out + __popcll(m & ((1ull << lane) - 1ull))] = s + i;
        out += __popcll(m);
    }
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_compact_rays(const uint8_t* mask, const int64_t* packed_info_all, const int64_t* packed_info_kept, int64_t R,
                     int64_t* kept, const int64_t* total_kept, int64_t* n_kept, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_compact_rays: negative ray count");
    if (R == 0) return NSX_OK;
    NSX_REQUIRE(mask && packed_info_all && packed_info_kept && kept, "nsx_compact_rays: NULL argument");
    NSX_REQUIRE(!n_kept || total_kept, "nsx_compact_rays: n_kept requires total_kept");
    hipLaunchKernelGGL(compact_rays_kernel, dim3((unsigned)((R + kCompactWaves - 1) / kCompactWaves)),
                       dim3(kCompactWaves * 64), 0, (hipStream_t)stream, mask, packed_info_all, packed_info_kept, R, kept,
                       total_kept, n_kept);
    NSX_LAUNCH_CHECK("nsx_compact_rays launch");
    return NSX_OK;
}

static int check_march(const char* who, const float* rays_o, const float* rays_d, const float* aabb,
                       const uint8_t* binary, int res, const float* near, float step) {
    NSX_REQUIRE(rays_o && rays_d && aabb && binary && near, "%s: NULL argument", who);
    NSX_REQUIRE(res >= 1 && res <= 1024, "%s: resolution %d out of range", who, res);
    NSX_REQUIRE(step > 0.0f, "%s: render_step_size must be > 0", who);
    return NSX_OK;
}

int nsx_march_count(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                    const uint8_t* binary, int res, const float* near, float far_plane, float step,
                    int64_t* counts, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_march_count: negative ray count");
    if (R == 0) return NSX_OK;
    if (int rc = check_march("nsx_march_count", rays_o, rays_d, aabb_host, binary, res, near, step)) return rc;
    NSX_REQUIRE(counts, "nsx_march_count: NULL counts");
    Aabb bb;
    for (int i = 0; i < 6; ++i) bb.v[i] = aabb_host[i];
    hipLaunchKernelGGL(march_count_kernel, dim3((unsigned)((R + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(kRaysPerBlock), 0, (hipStream_t)stream, rays_o,
                       rays_d, R, bb, binary, res, near, far_plane, step, counts);
    NSX_LAUNCH_CHECK("nsx_march_count launch");
    return NSX_OK;
}

int nsx_march_count_stash(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                          const uint8_t* binary, int res, const float* near, float far_plane, float step,
                          int64_t* counts, float* stash, int stash_cap, int64_t* over, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_march_count_stash: negative ray count");
    if (R == 0) return NSX_OK;
    if (int rc = check_march("nsx_march_count_stash", rays_o, rays_d, aabb_host, binary, res, near, step)) return rc;
    NSX_REQUIRE(counts && stash && over, "nsx_march_count_stash: NULL argument");
    NSX_REQUIRE(stash_cap >= 1, "nsx_march_count_stash: stash_cap %d", stash_cap);
    Aabb bb;
    for (int i = 0; i < 6; ++i) bb.v[i] = aabb_host[i];
    hipLaunchKernelGGL(march_count_stash_kernel, dim3((unsigned)((R + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(kRaysPerBlock), 0,
                       (hipStream_t)stream, rays_o, rays_d, R, bb, binary, res, near, far_plane, step, counts, stash, stash_cap,
                       over);
    NSX_LAUNCH_CHECK("nsx_march_count_stash launch");
    return NSX_OK;
}

int nsx_march_fill_from_stash(const float* stash, int stash_cap, int64_t R, float step, const int64_t* packed_info,
                              float* t_starts, float* t_ends, int64_t* ray_indices, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_march_fill_from_stash: negative ray count");
    if (R == 0) return NSX_OK;
    NSX_REQUIRE(stash && packed_info && t_starts && t_ends && ray_indices, "nsx_march_fill_from_stash: NULL argument");
    NSX_REQUIRE(stash_cap >= 1 && step > 0.f, "nsx_march_fill_from_stash: stash_cap %d, step %g", stash_cap, (double)step);
    hipLaunchKernelGGL(march_fill_from_stash_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, stash,
                       stash_cap, R, step, packed_info, t_starts, t_ends, ray_indices);
    NSX_LAUNCH_CHECK("nsx_march_fill_from_stash launch");
    return NSX_OK;
}

int nsx_copy_to_host_async(void* dst_pinned_host, const void* src_device, int64_t bytes, void* stream) {
    NSX_REQUIRE(dst_pinned_host && src_device && bytes >= 0, "nsx_copy_to_host_async: NULL argument / negative size");
    if (bytes == 0) return NSX_OK;
    hipError_t e = hipMemcpyAsync(dst_pinned_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "nsx_copy_to_host_async");
    return NSX_OK;
}

int nsx_pack_info(const int64_t* counts, int64_t R, int64_t* packed_info, int64_t* total, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_pack_info: negative ray count");
    NSX_REQUIRE(total, "nsx_pack_info: NULL total");
    NSX_REQUIRE(R == 0 || (counts && packed_info), "nsx_pack_info: NULL argument");
    // 4 waves, not 16: the counting pass of the next step runs beside the table optimizer's pass, and a 1024-thread block
    // waited for 16 free wave slots on one CU (0.6 ms on average, up to 1.4 ms, for 10 us of work; profiles/r02, r03)
    hipLaunchKernelGGL(pack_info_kernel, dim3(1), dim3(kPackThreads), 0, (hipStream_t)stream, counts, R, packed_info, total);
    NSX_LAUNCH_CHECK("nsx_pack_info launch");
    return NSX_OK;
}

int nsx_march_fill(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                   const uint8_t* binary, int res, const float* near, float far_plane, float step,
                   const int64_t* packed_info, float* t_starts, float* t_ends, int64_t* ray_indices,
                   int32_t* cells, void* stream) {
    NSX_REQUIRE(R >= 0, "nsx_march_fill: negative ray count");
    if (R == 0) return NSX_OK;
    if (int rc = check_march("nsx_march_fill", rays_o, rays_d, aabb_host, binary, res, near, step)) return rc;
    NSX_REQUIRE(packed_info && t_starts && t_ends && ray_indices, "nsx_march_fill: NULL argument");
    Aabb bb;
    for (int i = 0; i < 6; ++i) bb.v[i] = aabb_host[i];
    hipLaunchKernelGGL(march_fill_kernel, dim3((unsigned)((R + kRaysPerBlock - 1) / kRaysPerBlock)), dim3(kWave), 0, (hipStream_t)stream, rays_o,
                       rays_d, R, bb, binary, res, near, far_plane, step, packed_info, t_starts, t_ends, ray_indices,
                       cells);
    NSX_LAUNCH_CHECK("nsx_march_fill launch");
    return NSX_OK;
}

int nsx_ray_histogram(const int64_t* ray_indices, int64_t S, int64_t R, int64_t* counts_zeroed, const int64_t* n_device, void* stream) {
    NSX_REQUIRE(S >= 0 && R >= 0, "nsx_ray_histogram: negative size");
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(ray_indices && counts_zeroed, "nsx_ray_histogram: NULL argument");
    int64_t blocks = (S + 255) / 256;
    if (blocks > num_cus() * 8) blocks = num_cus() * 8;
    hipLaunchKernelGGL(ray_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ray_indices, S, R,
                       reinterpret_cast<unsigned long long*>(counts_zeroed), n_device);
    NSX_LAUNCH_CHECK("nsx_ray_histogram launch");
    return NSX_OK;
}

}  // extern "C"

// occ_grid.hip -- the occupancy-grid update of the path, everything except the density query itself.
//
// Replaces nerfacc 0.5.2 OccGridEstimator._update / _sample_uniform_and_occupied_cells as reached from
// nersemble_instant_ngp.py:184-196:  nonzero() of the binaries, two randint draws, an index gather of the grid
// coordinates, rand() jitter, the aabb mapping, the indexed EMA-max assignment, the masked mean and the threshold --
// ~25 torch launches with three boolean-index host syncs -- by five small kernels and one 4-byte read-back:
//
//   occ_block_count / occ_block_scan / occ_compact   ascending list of occupied cells + their number (device)
//   occ_sample_cells                                 slot -> (cell, jittered world position, random timestep)
//   occ_scatter_max                                  per-cell max of the slot values (order-independent atomic max)
//   occ_ema                                          occs = max(occs * decay, max) on the touched cells, partial sums
//   occ_threshold                                    mean (double), clamp, binaries
//
// Random numbers: Philox4x32-10 with key = seed, counter = (slot, purpose, step) -- the same stream on every
// data-parallel rank (identical grids without communication) and in oracle/occgrid.c (bit-exact cell ids, positions
// and timesteps in tests/test_occ_grid_gpu.py).  Arithmetic: fp32, contraction off, torch's operation order.
#include "nsx_common.h"
#pragma clang fp contract(off)

namespace nsx {

constexpr int kCellsPerBlock = 1024;

struct Philox4 { uint32_t w[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }

// ---- compaction of the occupied cells (ascending) -----------------------------------------------------------
__global__ __launch_bounds__(256) void occ_block_count_kernel(const uint8_t* __restrict__ binaries, int64_t n_cells,
                                                              int32_t* __restrict__ block_counts) {
    __shared__ int32_t part[4];
    const int64_t base = (int64_t)blockIdx.x * kCellsPerBlock;
    int32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kCellsPerBlock / 256; ++k) {
        const int64_t c = base + k * 256 + threadIdx.x;
        cnt += (c < n_cells && binaries[c]) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the block counts in place (one block; n_blocks is a few thousand), total -> *n_occ.
// 256 threads, not 1024: the scan of the visibility mask sits on the step's critical path, and a 16-wave block has to wait
// for a CU with 16 free wave slots -- 164 us behind a co-running deformation forward (profiles/r03_timeline_first_steps.txt)
constexpr int kScanThreads = 256;
template <typename CountT>
__global__ __launch_bounds__(1024) void occ_block_scan_kernel(int32_t* __restrict__ block_counts, int n_blocks,
                                                              CountT* __restrict__ n_occ) {
    __shared__ int32_t wave_sum[16];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += (int)blockDim.x) {
        const int i = base + threadIdx.x;
        const int32_t v = i < n_blocks ? block_counts[i] : 0;
        int32_t s = v;                                    // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t t = __shfl_up(s, o);
            if ((threadIdx.x & 63) >= o) s += t;
        }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = s;
        __syncthreads();
        int32_t before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wave_sum[w];
        if (i < n_blocks) block_counts[i] = before + s - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = before + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_occ = (CountT)carry;
}

template <typename IndexT>
__global__ __launch_bounds__(256) void occ_compact_kernel(const uint8_t* __restrict__ binaries, int64_t n_cells,
                                                          const int32_t* __restrict__ block_offsets,
                                                          IndexT* __restrict__ occupied) {
    __shared__ int32_t wave_cnt[4];
    const int64_t base = (int64_t)blockIdx.x * kCellsPerBlock;
    int32_t running = block_offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < kCellsPerBlock / 256; ++k) {
        const int64_t c = base + k * 256 + threadIdx.x;
        const bool set = c < n_cells && binaries[c];
        const uint64_t mask = __ballot(set);
        if (lane == 0) wave_cnt[wave] = __popcll(mask);
        __syncthreads();
        int32_t before = running;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        if (set) occupied[before + __popcll(mask & ((1ull << lane) - 1ull))] = (IndexT)c;
        running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// ---- slot -> cell, position, timestep -------------------------------------------------------------------------
struct Box6 { float lo[3], ext[3]; };

__global__ __launch_bounds__(256) void occ_sample_cells_kernel(
    int64_t n_cells, int res, Box6 box, int warmup, int64_t n_uniform, const int32_t* __restrict__ occupied,
    int64_t n_occ, uint32_t k0, uint32_t k1, uint32_t step_lo, uint32_t step_hi, int n_timesteps, int64_t M,
    int32_t* __restrict__ cell_ids, float* __restrict__ positions, int32_t* __restrict__ timesteps,
    float* __restrict__ times) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < M; s += (int64_t)gridDim.x * blockDim.x) {
        const Philox4 a = philox4x32_10((uint32_t)s, 0u, step_lo, step_hi, k0, k1);
        const Philox4 b = philox4x32_10((uint32_t)s, 1u, step_lo, step_hi, k0, k1);
        int64_t c;
        if (warmup) c = s;
        else if (s < n_uniform) c = (int64_t)(a.w[0] % (uint32_t)n_cells);
        else if (n_uniform < n_occ) c = occupied[a.w[0] % (uint32_t)n_occ];
        else c = occupied[s - n_uniform];
        cell_ids[s] = (int32_t)c;
        const int ijk[3] = {(int)(c / ((int64_t)res * res)), (int)((c / res) % res), (int)(c % res)};
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float x = ((float)ijk[ax] + u01(a.w[1 + ax])) / (float)res;
            positions[s * 3 + ax] = box.lo[ax] + x * box.ext[ax];
        }
        const int32_t t = (int32_t)(b.w[0] % (uint32_t)n_timesteps);
        timesteps[s] = t;
        times[s] = n_timesteps > 1 ? (float)t / (float)(n_timesteps - 1) : 0.0f;
    }
}

// ---- EMA-max + threshold -----------------------------------------------------------------------------------------
// newmax[c] = 0 while no slot of this update named cell c, else 1 + the bit pattern of the largest slot value
// (non-negative floats order like their bit patterns, so an unsigned atomic max is the float max).
constexpr uint32_t kUntouched = 0u;

__global__ __launch_bounds__(256) void occ_scatter_max_kernel(const int32_t* __restrict__ cell_ids,
                                                              const float* __restrict__ values, int64_t M,
                                                              int64_t n_cells, uint32_t* __restrict__ newmax) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < M; s += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = cell_ids[s];
        float v = values[s];
        if (c < 0 || c >= n_cells) continue;
        // nerfacc: occs[idx] = maximum(occs[idx] * decay, occ) -- every queried cell is decayed.  A negative value loses
        // against the decayed one exactly like 0; a NaN is taken as 0 too (the cell decays instead of being poisoned: the
        // one deviation from torch.maximum, stated in oracle/occgrid.c)
        if (!(v >= 0.0f)) v = 0.0f;
        // non-negative floats order like their bit patterns; +1 keeps 0 free as the "untouched" mark (+inf + 1 is
        // still below the sign bit)
        atomicMax(&newmax[c], __float_as_uint(v) + 1u);
    }
}

__global__ __launch_bounds__(256) void occ_ema_kernel(float* __restrict__ occs, uint32_t* __restrict__ newmax,
                                                      int64_t n_cells, float ema_decay,
                                                      double* __restrict__ block_sum, int64_t* __restrict__ block_cnt) {
    __shared__ double ssum[4];
    __shared__ int64_t scnt[4];
    const int64_t base = (int64_t)blockIdx.x * kCellsPerBlock;
    double sum = 0.0;
    int64_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kCellsPerBlock / 256; ++k) {
        const int64_t c = base + k * 256 + threadIdx.x;
        if (c >= n_cells) continue;
        float o = occs[c];
        const uint32_t m = newmax[c];
        if (m != kUntouched) {
            const float nv = __uint_as_float(m - 1u);
            const float decayed = o * ema_decay;
            o = decayed > nv ? decayed : nv;
            occs[c] = o;
            newmax[c] = kUntouched;                                      // ready for the next update
        }
        if (o >= 0.0f) { sum += (double)o; ++cnt; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off);
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = sum; scnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sum[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        block_cnt[blockIdx.x] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
    }
}

// every block re-derives the threshold from the (few thousand) block partials in the same fixed order
__global__ __launch_bounds__(256) void occ_threshold_kernel(const float* __restrict__ occs, int64_t n_cells,
                                                            const double* __restrict__ block_sum,
                                                            const int64_t* __restrict__ block_cnt, int n_blocks,
                                                            float occ_thre, uint8_t* __restrict__ binaries,
                                                            float* __restrict__ thre_out) {
    __shared__ double ssum[256];
    __shared__ int64_t scnt[256];
    double sum = 0.0;
    int64_t cnt = 0;
    for (int i = threadIdx.x; i < n_blocks; i += 256) { sum += block_sum[i]; cnt += block_cnt[i]; }
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    float thre = scnt[0] > 0 ? (float)(ssum[0] / (double)scnt[0]) : __uint_as_float(0x7FC00000u);
    if (thre > occ_thre) thre = occ_thre;                                // torch.clamp(mean, max=occ_thre)
    if (blockIdx.x == 0 && threadIdx.x == 0 && thre_out) *thre_out = thre;
    const int64_t base = (int64_t)blockIdx.x * kCellsPerBlock;
#pragma unroll
    for (int k = 0; k < kCellsPerBlock / 256; ++k) {
        const int64_t c = base + k * 256 + threadIdx.x;
        if (c < n_cells) binaries[c] = occs[c] > thre ? 1 : 0;
    }
}

static inline int n_cell_blocks(int64_t n_cells) { return (int)((n_cells + kCellsPerBlock - 1) / kCellsPerBlock); }

}  // namespace nsx

using namespace nsx;

extern "C" {

int64_t nsx_occ_scratch_bytes(int64_t n_cells) {
    if (n_cells <= 0) return 0;
    const int64_t nb = n_cell_blocks(n_cells);
    // [newmax u32 n_cells | block counts/offsets i32 nb | block sums f64 nb | block counts i64 nb], 16-B aligned parts
    auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
    return up(n_cells * 4) + up(nb * 4) + up(nb * 8) + up(nb * 8);
}

int nsx_occ_compact(const uint8_t* binaries, int64_t n_cells, int32_t* occupied, int32_t* n_occ, void* scratch,
                    void* stream) {
    NSX_REQUIRE(binaries && occupied && n_occ && scratch, "nsx_occ_compact: NULL argument");
    NSX_REQUIRE(n_cells > 0 && n_cells < (1ll << 31), "nsx_occ_compact: n_cells=%lld out of range", (long long)n_cells);
    const int nb = n_cell_blocks(n_cells);
    auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
    int32_t* counts = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(scratch) + up(n_cells * 4));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(occ_block_count_kernel, dim3(nb), dim3(256), 0, st, binaries, n_cells, counts);
    hipLaunchKernelGGL(occ_block_scan_kernel<int32_t>, dim3(1), dim3(kScanThreads), 0, st, counts, nb, n_occ);
    hipLaunchKernelGGL(occ_compact_kernel<int32_t>, dim3(nb), dim3(256), 0, st, binaries, n_cells, counts, occupied);
    NSX_LAUNCH_CHECK("nsx_occ_compact launch");
    return NSX_OK;
}

int nsx_compact_mask(const uint8_t* mask, int64_t n, int64_t* kept, int64_t* n_kept, void* scratch, void* stream) {
    NSX_REQUIRE(mask && kept && n_kept && scratch, "nsx_compact_mask: NULL argument");
    NSX_REQUIRE(n > 0 && n < (1ll << 31), "nsx_compact_mask: n=%lld out of range", (long long)n);
    const int nb = n_cell_blocks(n);
    auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
    int32_t* counts = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(scratch) + up(n * 4));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(occ_block_count_kernel, dim3(nb), dim3(256), 0, st, mask, n, counts);
    hipLaunchKernelGGL(occ_block_scan_kernel<int64_t>, dim3(1), dim3(kScanThreads), 0, st, counts, nb, n_kept);
    hipLaunchKernelGGL(occ_compact_kernel<int64_t>, dim3(nb), dim3(256), 0, st, mask, n, counts, kept);
    NSX_LAUNCH_CHECK("nsx_compact_mask launch");
    return NSX_OK;
}

int nsx_occ_sample_cells(int res, const float* aabb_host, int warmup, const int32_t* occupied, int64_t n_occ,
                         uint64_t seed, int64_t step, int n_timesteps, int64_t M, int32_t* cell_ids, float* positions,
                         int32_t* timesteps, float* times, void* stream) {
    NSX_REQUIRE(res > 0 && res <= 1024 && aabb_host, "nsx_occ_sample_cells: bad grid");
    NSX_REQUIRE(n_timesteps >= 1, "nsx_occ_sample_cells: n_timesteps=%d", n_timesteps);
    const int64_t n_cells = (int64_t)res * res * res;
    const int64_t n_uniform = n_cells / 4;
    const int64_t want = warmup ? n_cells : n_uniform + (n_uniform < n_occ ? n_uniform : n_occ);
    NSX_REQUIRE(M == want, "nsx_occ_sample_cells: M=%lld but the update at this state has %lld slots", (long long)M,
                (long long)want);
    NSX_REQUIRE(warmup || n_occ == 0 || occupied, "nsx_occ_sample_cells: NULL occupied list");
    if (M == 0) return NSX_OK;
    NSX_REQUIRE(cell_ids && positions && timesteps && times, "nsx_occ_sample_cells: NULL output");
    Box6 box;
    for (int a = 0; a < 3; ++a) { box.lo[a] = aabb_host[a]; box.ext[a] = aabb_host[3 + a] - aabb_host[a]; }
    int64_t blocks = (M + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(occ_sample_cells_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_cells, res,
                       box, warmup, n_uniform, occupied, n_occ, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                       (uint32_t)((uint64_t)step >> 32), n_timesteps, M, cell_ids, positions, timesteps, times);
    NSX_LAUNCH_CHECK("nsx_occ_sample_cells launch");
    return NSX_OK;
}

int nsx_occ_update(float* occs, uint8_t* binaries, int64_t n_cells, const int32_t* cell_ids, const float* occ_values,
                   int64_t M, float ema_decay, float occ_thre, void* scratch_zeroed_once, float* threshold_out,
                   void* stream) {
    NSX_REQUIRE(occs && binaries && scratch_zeroed_once, "nsx_occ_update: NULL argument");
    NSX_REQUIRE(n_cells > 0 && n_cells < (1ll << 31), "nsx_occ_update: n_cells=%lld out of range", (long long)n_cells);
    NSX_REQUIRE(M >= 0 && (M == 0 || (cell_ids && occ_values)), "nsx_occ_update: bad slot arrays");
    const int nb = n_cell_blocks(n_cells);
    auto up = [](int64_t v) { return (v + 15) / 16 * 16; };
    uint8_t* base = reinterpret_cast<uint8_t*>(scratch_zeroed_once);
    uint32_t* newmax = reinterpret_cast<uint32_t*>(base);
    double* bsum = reinterpret_cast<double*>(base + up(n_cells * 4) + up((int64_t)nb * 4));
    int64_t* bcnt = reinterpret_cast<int64_t*>(base + up(n_cells * 4) + up((int64_t)nb * 4) + up((int64_t)nb * 8));
    hipStream_t st = (hipStream_t)stream;
    if (M > 0) {
        int64_t blocks = (M + 255) / 256;
        const int64_t cap = (int64_t)num_cus() * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(occ_scatter_max_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cell_ids, occ_values, M,
                           n_cells, newmax);
    }
    hipLaunchKernelGGL(occ_ema_kernel, dim3(nb), dim3(256), 0, st, occs, newmax, n_cells, ema_decay, bsum, bcnt);
    hipLaunchKernelGGL(occ_threshold_kernel, dim3(nb), dim3(256), 0, st, occs, n_cells, bsum, bcnt, nb, occ_thre,
                       binaries, threshold_out);
    NSX_LAUNCH_CHECK("nsx_occ_update launch");
    return NSX_OK;
}

}  // extern "C"

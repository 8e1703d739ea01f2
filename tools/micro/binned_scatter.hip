// Experiment, second route for VERDICT r05 item 7: the factored hash-gradient scatter without memory-side atomics by BINNING.
//
//   The owner-computes prototype (owner_scatter.hip) paid for its redundancy: every tile block of a (slot, level) walks all
//   samples of the slot (2.9 - 3.4 ms at 2^20 samples).  Here every (sample, level, (y, z) corner pair) ITEM is produced once,
//   routed to the bin of its (slot, level, 16 K-entry tile) and reduced there:
//     K1 count   block-private LDS histogram of the items of a contiguous chunk of samples, flushed with one atomic per
//                (block, non-empty bin)                                             -> bin sizes, exclusive scan (one block)
//     K2 fill    the same walk: the block reserves a run in every bin it feeds (one atomic per non-empty bin) and writes its
//                16-byte items {local index x0 | x1, wyz g0, wyz g1, wx} there: runs of ~100 items, coalesced
//     K3 reduce  one block per bin: items streamed with 16-byte loads, four ds_add_f32 each into a 128 KB LDS tile, the tile
//                stored with plain coalesced writes (G needs no clearing: every tile of every plane is written)
//   Traffic at 2^20 samples: 67 M items x 16 B written + read = 2.1 GB, + 1.2 GB of G: ~3.4 GB instead of ~41 M sector atomics.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -munsafe-fp-atomics tools/micro/binned_scatter.hip \
//                 -L nersemble_amd/csrc -lnsx -o tools/micro/binned_scatter
//   run:    tools/micro/binned_scatter [log2_samples=20] [n_slots=24] [coherent=0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "nsx.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int TILE_LOG2 = 14, TILE = 1 << TILE_LOG2;
constexpr int MAX_BINS = 24 * 400;                         // 24 slots x 387 tiles
constexpr int CHUNK = 4096;                                // samples per block of K1 / K2

struct Tiles { int level[400]; int first[400]; int first_tile[17]; int n; };

__device__ __forceinline__ void cell1(float scale, float p, uint32_t& c, float& w) {
    const float q = __fmaf_rn(scale, p, 0.5f);
    const float f = floorf(q);
    c = (uint32_t)(int32_t)f;
    w = q - f;
}

// the items of (sample b, level l): f(tile, i0, i1, a0, a1, wx) -- hashed level: 4 (y, z) pairs whose x neighbours share a
// tile; dense level: 8 single corners (i1 = i0, wx = 0)
template <typename F>
__device__ __forceinline__ void items_of(const float* __restrict__ x, const float* __restrict__ dout, int64_t b, int l,
                                         const nsx_grid_geom& g, F&& f) {
    const float px = x[b * 3], py = x[b * 3 + 1], pz = x[b * 3 + 2];
    uint32_t cx, cy, cz; float wx, wy, wz;
    const float scale = g.scale[l];
    cell1(scale, px, cx, wx); cell1(scale, py, cy, wy); cell1(scale, pz, cz, wz);
    const float g0 = dout[b * 2 * g.n_levels + 2 * l], g1 = dout[b * 2 * g.n_levels + 2 * l + 1];
    const uint32_t size = g.size[l];
    if (g.hashed[l]) {
        const uint32_t mask = size - 1u;
        const uint32_t yh[2] = {cy * 2654435761u, (cy + 1u) * 2654435761u};
        const uint32_t zh[2] = {cz * 805459861u, (cz + 1u) * 805459861u};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t h = yh[k & 1] ^ zh[k >> 1];
            const float wyz = ((k & 1) ? wy : 1.0f - wy) * ((k >> 1) ? wz : 1.0f - wz);
            const uint32_t e0 = (cx ^ h) & mask, e1 = ((cx + 1u) ^ h) & mask;
            f((int)(e0 >> TILE_LOG2), e0 & (TILE - 1), e1 & (TILE - 1), wyz * g0, wyz * g1, wx);
        }
    } else {
        const uint32_t res = g.res[l];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t idx = (cx + (k & 1)) + (cy + ((k >> 1) & 1)) * res + (cz + (k >> 2)) * res * res;
            if (idx >= size) idx -= size;
            const float w = ((k & 1) ? wx : 1.0f - wx) * (((k & 2) ? wy : 1.0f - wy) * ((k & 4) ? wz : 1.0f - wz));
            f((int)(idx >> TILE_LOG2), idx & (TILE - 1), idx & (TILE - 1), w * g0, w * g1, 0.0f);
        }
    }
}

__global__ __launch_bounds__(256) void count_kernel(const float* __restrict__ x, const int32_t* __restrict__ slot,
                                                    const float* __restrict__ dout, int64_t S, const nsx_grid_geom g,
                                                    const Tiles tiles, int n_bins, uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t hist[];
    for (int i = threadIdx.x; i < n_bins; i += 256) hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * CHUNK, hi = lo + CHUNK < S ? lo + CHUNK : S;
    for (int64_t b = lo + threadIdx.x; b < hi; b += 256) {
        const int base = slot[b] * tiles.n;
        for (int l = 0; l < g.n_levels; ++l)
            items_of(x, dout, b, l, g, [&](int t, uint32_t, uint32_t, float, float, float) {
                atomicAdd(&hist[base + tiles.first_tile[l] + t], 1u);
            });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += 256)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t* __restrict__ counts, int n_bins, uint32_t* __restrict__ offs,
                                                    uint32_t* __restrict__ cursor) {
    __shared__ uint32_t part[1024];
    const int per = (n_bins + 1023) / 1024;
    uint32_t s = 0;
    for (int i = 0; i < per; ++i) { const int k = threadIdx.x * per + i; if (k < n_bins) s += counts[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; } }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int i = 0; i < per; ++i) {
        const int k = threadIdx.x * per + i;
        if (k < n_bins) { offs[k] = run; cursor[k] = run; run += counts[k]; }
    }
    if (threadIdx.x == 1023) offs[n_bins] = run;
}

__global__ __launch_bounds__(256) void fill_kernel(const float* __restrict__ x, const int32_t* __restrict__ slot,
                                                   const float* __restrict__ dout, int64_t S, const nsx_grid_geom g,
                                                   const Tiles tiles, int n_bins, uint32_t* __restrict__ cursor,
                                                   uint4* __restrict__ items) {
    extern __shared__ uint32_t hist[];
    for (int i = threadIdx.x; i < n_bins; i += 256) hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * CHUNK, hi = lo + CHUNK < S ? lo + CHUNK : S;
    for (int64_t b = lo + threadIdx.x; b < hi; b += 256) {
        const int base = slot[b] * tiles.n;
        for (int l = 0; l < g.n_levels; ++l)
            items_of(x, dout, b, l, g, [&](int t, uint32_t, uint32_t, float, float, float) {
                atomicAdd(&hist[base + tiles.first_tile[l] + t], 1u);
            });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins; i += 256) {
        const uint32_t c = hist[i];
        hist[i] = c ? atomicAdd(&cursor[i], c) : 0u;             // this block's run in bin i starts here
    }
    __syncthreads();
    for (int64_t b = lo + threadIdx.x; b < hi; b += 256) {
        const int base = slot[b] * tiles.n;
        for (int l = 0; l < g.n_levels; ++l)
            items_of(x, dout, b, l, g, [&](int t, uint32_t i0, uint32_t i1, float a0, float a1, float wx) {
                const uint32_t pos = atomicAdd(&hist[base + tiles.first_tile[l] + t], 1u);
                items[pos] = make_uint4(i0 | (i1 << 14), __float_as_uint(a0), __float_as_uint(a1), __float_as_uint(wx));
            });
    }
}

__global__ __launch_bounds__(1024) void reduce_kernel(const uint4* __restrict__ items, const uint32_t* __restrict__ offs,
                                                      const nsx_grid_geom g, const Tiles tiles, float2* __restrict__ G2,
                                                      uint64_t g_total) {
    extern __shared__ float acc[];
    const int t = blockIdx.x, s = blockIdx.y;
    const int l = tiles.level[t];
    const uint32_t first = (uint32_t)tiles.first[t];
    const uint32_t n_in = g.size[l] - first < (uint32_t)TILE ? g.size[l] - first : (uint32_t)TILE;
    for (int i = threadIdx.x; i < 2 * (int)n_in; i += 1024) acc[i] = 0.f;
    __syncthreads();
    const int bin = s * tiles.n + t;
    const uint32_t end = offs[bin + 1];
    for (uint32_t i = offs[bin] + threadIdx.x; i < end; i += 1024) {
        const uint4 it = items[i];
        const uint32_t i0 = it.x & (TILE - 1), i1 = (it.x >> 14) & (TILE - 1);
        const float a0 = __uint_as_float(it.y), a1 = __uint_as_float(it.z), wx = __uint_as_float(it.w);
        const float u = 1.0f - wx;
        atomicAdd(&acc[2 * i0], u * a0); atomicAdd(&acc[2 * i0 + 1], u * a1);
        if (wx != 0.f) { atomicAdd(&acc[2 * i1], wx * a0); atomicAdd(&acc[2 * i1 + 1], wx * a1); }
    }
    __syncthreads();
    float2* dst = G2 + (uint64_t)s * g_total + g.offset[l] + first;
    const float2* a2 = reinterpret_cast<const float2*>(acc);
    for (uint32_t i = threadIdx.x; i < n_in; i += 1024) dst[i] = a2[i];
}

int main(int argc, char** argv) {
    const int log2s = argc > 1 ? atoi(argv[1]) : 20, T = argc > 2 ? atoi(argv[2]) : 24, coherent = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t S = 1ll << log2s;
    nsx_grid_geom g;
    if (nsx_grid_geometry(16, 1.4472692012786865f, 16, 19, &g)) { printf("geometry: %s\n", nsx_last_error()); return 1; }
    const uint64_t total = g.offset[16];
    Tiles tiles; tiles.n = 0;
    for (int l = 0; l < 16; ++l) {
        tiles.first_tile[l] = tiles.n;
        for (uint32_t f = 0; f < g.size[l]; f += TILE) { tiles.level[tiles.n] = l; tiles.first[tiles.n] = (int)f; ++tiles.n; }
    }
    tiles.first_tile[16] = tiles.n;
    const int n_bins = T * tiles.n;
    if (n_bins > MAX_BINS) { printf("too many bins\n"); return 1; }
    printf("S = 2^%d, %d slots, %d tiles per slot (%d bins), %s samples\n", log2s, T, tiles.n, n_bins, coherent ? "ray-coherent" : "uniform");
    std::vector<float> hx(S * 3), hd(S * 32);
    std::vector<int32_t> hslot(S);
    srand(1);
    auto rnd = []() { return (float)rand() / ((float)RAND_MAX + 1.0f); };
    if (coherent) {
        const int64_t per = S / 4096;
        for (int r = 0; r < 4096; ++r) {
            float o[3] = {rnd(), rnd(), rnd()}, e[3] = {rnd(), rnd(), rnd()};
            const int sl = rand() % T;
            for (int64_t k = 0; k < per; ++k) {
                const float a = (float)k / (float)per * 0.3f;
                for (int d = 0; d < 3; ++d) hx[(r * per + k) * 3 + d] = fminf(0.999f, fmaxf(0.0f, o[d] + a * (e[d] - o[d])));
                hslot[r * per + k] = sl;
            }
        }
    } else {
        for (int64_t i = 0; i < S * 3; ++i) hx[i] = rnd() * 0.999f;
        for (int64_t i = 0; i < S; ++i) hslot[i] = rand() % T;
    }
    for (int64_t i = 0; i < S * 32; ++i) hd[i] = rnd() - 0.5f;
    float *x, *d, *Ga, *Gb; int32_t* slot; uint32_t *counts, *offs, *cursor; uint4* items;
    const int64_t max_items = S * (11 * 4 + 5 * 8);
    CK(hipMalloc(&x, S * 12)); CK(hipMalloc(&d, S * 128)); CK(hipMalloc(&slot, S * 4));
    CK(hipMalloc(&counts, (n_bins + 1) * 4)); CK(hipMalloc(&offs, (n_bins + 1) * 4)); CK(hipMalloc(&cursor, (n_bins + 1) * 4));
    CK(hipMalloc(&items, max_items * 16));
    CK(hipMalloc(&Ga, (size_t)T * total * 8)); CK(hipMalloc(&Gb, (size_t)T * total * 8));
    CK(hipMemcpy(x, hx.data(), S * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(d, hd.data(), S * 128, hipMemcpyHostToDevice));
    CK(hipMemcpy(slot, hslot.data(), S * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 8));
    const int blocks = (int)((S + CHUNK - 1) / CHUNK);
    const size_t hsm = (size_t)n_bins * 4;
    hipEvent_t e[5]; for (auto& ev : e) CK(hipEventCreate(&ev));
    auto binned = [&](bool timed) {
        CK(hipMemsetAsync(counts, 0, (n_bins + 1) * 4, 0));
        if (timed) CK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), hsm, 0, x, slot, d, S, g, tiles, n_bins, counts);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, 0, counts, n_bins, offs, cursor);
        if (timed) CK(hipEventRecord(e[1]));
        hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), hsm, 0, x, slot, d, S, g, tiles, n_bins, cursor, items);
        if (timed) CK(hipEventRecord(e[2]));
        hipLaunchKernelGGL(reduce_kernel, dim3(tiles.n, T), dim3(1024), TILE * 8, 0, items, offs, g, tiles,
                           reinterpret_cast<float2*>(Gb), total);
        if (timed) CK(hipEventRecord(e[3]));
    };
    auto atomics = [&]() {
        CK(hipMemsetAsync(Ga, 0, (size_t)T * total * 8, 0));
        if (nsx_hash_ensemble_bwd_scatter(x, S, &g, T, slot, d, Ga, nullptr, 8, nullptr, nullptr)) { printf("%s\n", nsx_last_error()); exit(1); }
    };
    float ms, a, b, c;
    for (int rep = 0; rep < 3; ++rep) {
        binned(true); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e[0], e[3])); CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2]));
        CK(hipEventElapsedTime(&c, e[2], e[3]));
        printf("binned: %.3f ms  (count + scan %.3f, fill %.3f, reduce %.3f)\n", ms, a, b, c);
        atomics(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e[0])); for (int i = 0; i < 3; ++i) atomics(); CK(hipEventRecord(e[1])); CK(hipEventSynchronize(e[1]));
        CK(hipEventElapsedTime(&ms, e[0], e[1])); printf("atomics (fill + nsx_hash_ensemble_bwd_scatter): %.3f ms\n", ms / 3);
    }
    std::vector<float> va((size_t)T * total * 2), vb((size_t)T * total * 2);
    CK(hipMemcpy(va.data(), Ga, va.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(vb.data(), Gb, vb.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, err = 0;
    for (size_t i = 0; i < va.size(); ++i) { mx = std::max(mx, (double)fabsf(va[i])); err = std::max(err, (double)fabsf(va[i] - vb[i])); }
    printf("max |G| %.4g, max |binned - atomics| %.4g (%.2g of the maximum)\n", mx, err, err / mx);
    return 0;
}

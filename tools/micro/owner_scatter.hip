// Experiment (VERDICT r05 "next round" item 7): the factored hash-gradient scatter WITHOUT memory-side atomics.
//
//   owner-computes: one block owns a 16 K-entry tile of one (code slot, level) plane of G, accumulates in LDS (ds_add_f32)
//   and stores its tile with plain coalesced writes -- G needs no clearing and sees no atomics.  The price is redundancy:
//   every block of a (slot, level) walks ALL samples of that slot.  Two things keep the walk cheap:
//     * with the reference's primes the x prime is 1, so on a hashed level idx = x ^ (y P1 ^ z P2) and -- x < 2^14 -- the
//       tile (idx >> 14) of a corner depends on (y, z) only: a sample has 4 (y, z) corner pairs, each pair's two x
//       neighbours fall in the SAME tile;
//     * a pre-pass writes those 4 tile numbers (5 bits each) per (level, sample) once; the tile blocks then read ONE dword
//       per sample and compare -- the hash arithmetic is not repeated 32 times.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -munsafe-fp-atomics tools/micro/owner_scatter.hip \
//                 -L nersemble_amd/csrc -lnsx -Wl,-rpath,$PWD/nersemble_amd/csrc -o tools/micro/owner_scatter
//   run:    tools/micro/owner_scatter [log2_samples=20] [n_slots=24] [coherent=0|1]
// Prints the time of the pre-pass + tile kernel against nsx_hash_ensemble_bwd_scatter (the atomics kernel) and the largest
// difference of the two gradients.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "nsx.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef TILE_LOG2_
#define TILE_LOG2_ 14
#endif
constexpr int TILE_LOG2 = TILE_LOG2_, TILE = 1 << TILE_LOG2;   // entries per tile: 128 KB of float2 in LDS at 14, 64 KB at 13
constexpr int SB = 19 - TILE_LOG2;                        // bits of a tile number on a hashed level (2^19 entries)
constexpr int THREADS = 1024;

struct Tiles { int level[512]; int first[512]; int n; };

__device__ __forceinline__ void cell1(float scale, float p, uint32_t& c, float& w) {
    const float q = __fmaf_rn(scale, p, 0.5f);
    const float f = floorf(q);
    c = (uint32_t)(int32_t)f;
    w = q - f;
}

// signature of a (level, sample): hashed level -- the tile of the 4 (y, z) corner pairs, 5 bits each (bit 31 clear);
// dense level -- lowest and highest tile any corner can fall in (8 bits each, bit 31 set)
// (second version: everything the tile blocks read is written in SLOT-SORTED order -- position i = sample perm[i] -- so that a
// block's walk over its slot is a contiguous stream: the first version gathered sig[perm[i]], one 64-byte line per dword)
__global__ __launch_bounds__(256) void signature_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm,
                                                        const float* __restrict__ dout, int64_t S, const nsx_grid_geom g,
                                                        uint32_t* __restrict__ sig, float* __restrict__ xs,
                                                        float2* __restrict__ ds) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const int64_t b = perm[i];
    const float py = x[b * 3 + 1], pz = x[b * 3 + 2], px = x[b * 3];
    xs[i * 3] = px; xs[i * 3 + 1] = py; xs[i * 3 + 2] = pz;
    for (int l = 0; l < g.n_levels; ++l)
        ds[(int64_t)l * S + i] = float2{dout[b * 2 * g.n_levels + 2 * l], dout[b * 2 * g.n_levels + 2 * l + 1]};
    for (int l = 0; l < g.n_levels; ++l) {
        uint32_t cy, cz, cx; float wy, wz, wx;
        cell1(g.scale[l], py, cy, wy); cell1(g.scale[l], pz, cz, wz);
        uint32_t s;
        if (g.hashed[l]) {
            const uint32_t mask = g.size[l] - 1u;
            const uint32_t y0 = cy * 2654435761u, y1 = (cy + 1u) * 2654435761u;
            const uint32_t z0 = cz * 805459861u, z1 = (cz + 1u) * 805459861u;
            s = (((y0 ^ z0) & mask) >> TILE_LOG2) | ((((y1 ^ z0) & mask) >> TILE_LOG2) << SB) |
                ((((y0 ^ z1) & mask) >> TILE_LOG2) << (2 * SB)) | ((((y1 ^ z1) & mask) >> TILE_LOG2) << (3 * SB));
        } else {
            cell1(g.scale[l], px, cx, wx);
            const uint32_t res = g.res[l], size = g.size[l];
            const uint32_t lo = cx + cy * res + cz * res * res, hi = lo + 1u + res + res * res;
            // (in-contract positions: lo < size; hi may wrap once -- then every tile is a candidate)
            s = 0x80000000u | (hi >= size ? 0x00FF00u : (((hi >> TILE_LOG2) & 0xFFu) << 8)) | ((lo >> TILE_LOG2) & 0xFFu);
        }
        sig[(int64_t)l * S + i] = s;
    }
}

__global__ __launch_bounds__(THREADS) void owner_scatter_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ slot_off,
    const uint32_t* __restrict__ sig, int64_t S, const float2* __restrict__ ds, const nsx_grid_geom g, const Tiles tiles,
    float2* __restrict__ G2, uint64_t g_total) {
    extern __shared__ float acc[];                         // [TILE][2]
    const int t = blockIdx.x, s = blockIdx.y;
    const int l = tiles.level[t];
    const uint32_t first = (uint32_t)tiles.first[t];
    const uint32_t size = g.size[l], res = g.res[l], off = g.offset[l];
    const uint32_t n_in = size - first < (uint32_t)TILE ? size - first : (uint32_t)TILE;
    const uint32_t r = first >> TILE_LOG2;
    for (int i = threadIdx.x; i < 2 * TILE; i += THREADS) acc[i] = 0.f;
    __syncthreads();
    const float scale = g.scale[l];
    const bool hashed = g.hashed[l] != 0;
    const uint32_t* sl = sig + (int64_t)l * S;
    const float2* dl = ds + (int64_t)l * S;
    const int end = slot_off[s + 1];
    for (int i0_ = slot_off[s] + threadIdx.x; i0_ < end; i0_ += 4 * THREADS) {
      uint32_t sg4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sg4[u] = (i0_ + u * THREADS < end) ? sl[i0_ + u * THREADS] : 0xFFFFFFFFu;   // four in flight
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0_ + u * THREADS;
        if (i >= end) break;
        const int b = i;
        const uint32_t sg = sg4[u];
        if (hashed) {
            constexpr uint32_t SM = (1u << SB) - 1u;
            const uint32_t m0 = (sg & SM) == r, m1 = ((sg >> SB) & SM) == r, m2 = ((sg >> (2 * SB)) & SM) == r,
                           m3 = ((sg >> (3 * SB)) & SM) == r;
            if (!(m0 | m1 | m2 | m3)) continue;
            const float px = x[b * 3], py = x[b * 3 + 1], pz = x[b * 3 + 2];
            uint32_t cx, cy, cz; float wx, wy, wz;
            cell1(scale, px, cx, wx); cell1(scale, py, cy, wy); cell1(scale, pz, cz, wz);
            const float2 gg = dl[b];
            const float g0 = gg.x, g1 = gg.y;
            const uint32_t mask = size - 1u;
            const uint32_t yh[2] = {cy * 2654435761u, (cy + 1u) * 2654435761u};
            const uint32_t zh[2] = {cz * 805459861u, (cz + 1u) * 805459861u};
            const uint32_t hit[4] = {m0, m1, m2, m3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!hit[k]) continue;
                const uint32_t h = yh[k & 1] ^ zh[k >> 1];
                const float wyz = ((k & 1) ? wy : 1.0f - wy) * ((k >> 1) ? wz : 1.0f - wz);
                const uint32_t i0 = ((cx ^ h) & mask) - first, i1 = (((cx + 1u) ^ h) & mask) - first;
                const float w0 = (1.0f - wx) * wyz, w1 = wx * wyz;
                atomicAdd(&acc[2 * i0], w0 * g0); atomicAdd(&acc[2 * i0 + 1], w0 * g1);
                atomicAdd(&acc[2 * i1], w1 * g0); atomicAdd(&acc[2 * i1 + 1], w1 * g1);
            }
        } else {
            const uint32_t lo = sg & 0xFFu, hi = (sg >> 8) & 0xFFu;
            if (r < lo || r > hi) continue;
            const float px = x[b * 3], py = x[b * 3 + 1], pz = x[b * 3 + 2];
            uint32_t cx, cy, cz; float wx, wy, wz;
            cell1(scale, px, cx, wx); cell1(scale, py, cy, wy); cell1(scale, pz, cz, wz);
            const float2 gg = dl[b];
            const float g0 = gg.x, g1 = gg.y;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t idx = (cx + (k & 1)) + (cy + ((k >> 1) & 1)) * res + (cz + (k >> 2)) * res * res;
                if (idx >= size) idx -= size;
                const uint32_t it = idx - first;
                if (it < n_in) {
                    const float w = ((k & 1) ? wx : 1.0f - wx) * (((k & 2) ? wy : 1.0f - wy) * ((k & 4) ? wz : 1.0f - wz));
                    atomicAdd(&acc[2 * it], w * g0); atomicAdd(&acc[2 * it + 1], w * g1);
                }
            }
        }
      }
    }
    __syncthreads();
    float2* dst = G2 + (uint64_t)s * g_total + off + first;
    const float2* a2 = reinterpret_cast<const float2*>(acc);
    for (uint32_t i = threadIdx.x; i < n_in; i += THREADS) dst[i] = a2[i];
}

int main(int argc, char** argv) {
    const int log2s = argc > 1 ? atoi(argv[1]) : 20, T = argc > 2 ? atoi(argv[2]) : 24, coherent = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t S = 1ll << log2s;
    nsx_grid_geom g;
    if (nsx_grid_geometry(16, 1.4472692012786865f, 16, 19, &g)) { printf("geometry: %s\n", nsx_last_error()); return 1; }
    const uint64_t total = g.offset[16];
    Tiles tiles; tiles.n = 0;
    for (int l = 0; l < 16; ++l)
        for (uint32_t f = 0; f < g.size[l]; f += TILE) { tiles.level[tiles.n] = l; tiles.first[tiles.n] = (int)f; ++tiles.n; }
    printf("S = 2^%d, %d slots, %d tiles per slot, entries %llu, %s samples\n", log2s, T, tiles.n, (unsigned long long)total,
           coherent ? "ray-coherent" : "uniform");
    std::vector<float> hx(S * 3), hd(S * 32);
    std::vector<int32_t> hslot(S), hperm(S), hoff(T + 1, 0);
    srand(1);
    auto rnd = []() { return (float)rand() / ((float)RAND_MAX + 1.0f); };
    if (coherent) {
        // 4096 rays of S / 4096 samples each along a straight segment, rays sorted by slot (as the pixel sampler need not)
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
    for (int64_t i = 0; i < S; ++i) hoff[hslot[i] + 1]++;
    for (int s = 0; s < T; ++s) hoff[s + 1] += hoff[s];
    { std::vector<int32_t> cur(hoff.begin(), hoff.end() - 1); for (int64_t i = 0; i < S; ++i) hperm[cur[hslot[i]]++] = (int32_t)i; }
    float *x, *d, *Ga, *Gb, *xs; float2* ds; int32_t *slot, *perm, *off; uint32_t* sig;
    CK(hipMalloc(&x, S * 12)); CK(hipMalloc(&d, S * 128)); CK(hipMalloc(&slot, S * 4)); CK(hipMalloc(&perm, S * 4));
    CK(hipMalloc(&off, (T + 1) * 4)); CK(hipMalloc(&sig, S * 16 * 4)); CK(hipMalloc(&xs, S * 12)); CK(hipMalloc(&ds, S * 16 * 8));
    CK(hipMalloc(&Ga, (size_t)T * total * 8)); CK(hipMalloc(&Gb, (size_t)T * total * 8));
    CK(hipMemcpy(x, hx.data(), S * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(d, hd.data(), S * 128, hipMemcpyHostToDevice));
    CK(hipMemcpy(slot, hslot.data(), S * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(perm, hperm.data(), S * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(off, hoff.data(), (T + 1) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)owner_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto owner = [&]() {
        hipLaunchKernelGGL(signature_kernel, dim3((S + 255) / 256), dim3(256), 0, 0, x, perm, d, S, g, sig, xs, ds);
        hipLaunchKernelGGL(owner_scatter_kernel, dim3(tiles.n, T), dim3(THREADS), TILE * 8, 0, xs, off, sig, S, ds, g, tiles,
                           reinterpret_cast<float2*>(Gb), total);
    };
    auto atomics = [&]() {
        CK(hipMemsetAsync(Ga, 0, (size_t)T * total * 8, 0));
        if (nsx_hash_ensemble_bwd_scatter(x, S, &g, T, slot, d, Ga, nullptr, 8, nullptr, nullptr)) { printf("%s\n", nsx_last_error()); exit(1); }
    };
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        owner(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) owner(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("owner-computes (signature pre-pass + tile kernel): %.3f ms\n", ms / 5);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(signature_kernel, dim3((S + 255) / 256), dim3(256), 0, 0, x, perm, d, S, g, sig, xs, ds);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("   of which the signature pre-pass:               %.3f ms\n", ms / 5);
        atomics(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) atomics(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("atomics (fill + nsx_hash_ensemble_bwd_scatter):    %.3f ms\n", ms / 5);
    }
    std::vector<float> a((size_t)T * total * 2), b((size_t)T * total * 2);
    CK(hipMemcpy(a.data(), Ga, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Gb, b.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, err = 0;
    for (size_t i = 0; i < a.size(); ++i) { mx = std::max(mx, (double)fabsf(a[i])); err = std::max(err, (double)fabsf(a[i] - b[i])); }
    printf("max |G| %.4g, max |owner - atomics| %.4g (%.2g of the maximum)\n", mx, err, err / mx);
    return 0;
}

/*
 * oracle/occgrid.c -- CPU restatement of the occupancy-grid update.
 * TEST INFRASTRUCTURE ONLY (see nsx_oracle.h).
 *
 * Reference call site: nersemble_instant_ngp.py:184-196 (update_occupancy_grid callback) ->
 * nerfacc 0.5.2 OccGridEstimator.update_every_n_steps / _update (UPSTREAM, un-vendored: restated from the published
 * algorithm, SURVEY.md A.4 -- "parity unpinned"):
 *     cells  = all cells                                  while step < warmup_steps
 *            = n uniform draws  U  (all occupied cells, or n draws from them when there are more than n),  n = N / 4
 *     x      = (ijk + U[0,1)^3) / res, mapped into the aabb
 *     occ    = occ_eval_fn(x)          (density at a random timestep x render step, nersemble_instant_ngp.py:187-191)
 *     occs[c]  = max(occs[c] * ema_decay, occ)
 *     binaries = occs > min(mean(occs[occs >= 0]), occ_thre)
 *
 * Two things nerfacc leaves to torch are fixed here, identically in csrc/occ_grid.hip:
 *   * random numbers: torch's device generator cannot be restated on another device, so both sides draw from
 *     Philox4x32-10 (Salmon et al., SC'11; known-answer vectors checked in tests/test_occ_grid_cpu.py) with
 *     key = seed and counter = (slot, purpose, step, 0).  purpose 0: word 0 picks the cell (slot < n: uniform,
 *     word % N; else the (word % n_occ)-th occupied cell), words 1-3 are the jitter ((w >> 8) * 2^-24, the
 *     24-bit construction of torch's uniform); purpose 1: word 0 % T is the timestep of the density query.
 *   * duplicate cells in one update: torch's indexed assignment keeps an arbitrary one of the duplicates' values;
 *     here the cell takes max(occs * decay, max over its duplicates) -- one of the outcomes torch can produce,
 *     and the deterministic one.
 * The mean is accumulated in double and rounded once (torch reduces in fp32 in an unspecified order; the double sum
 * is order-independent at fp32 resolution).
 */
#include "nsx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void nsxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; /* 2^-24 */ }

/* Number of density queries of the update at `step`. */
int64_t nsxo_occ_num_slots(int64_t n_cells, int64_t n_occ, int warmup) {
    if (warmup) return n_cells;
    const int64_t n = n_cells / 4;
    return n + (n < n_occ ? n : n_occ);
}

/* Cell pick + jitter + random timestep for every slot.  binaries [res^3] (0/1); outputs: cell_ids [M] int32,
 * positions [M][3] fp32 (world), timesteps [M] int32, times [M] fp32 = timestep / (T - 1) (0 when T == 1).
 * Returns M (= nsxo_occ_num_slots for the grid's current number of occupied cells). */
int64_t nsxo_occ_sample_cells(const uint8_t* binaries, int res, const float* aabb, int warmup, uint64_t seed,
                              int64_t step, int n_timesteps, int32_t* cell_ids, float* positions,
                              int32_t* timesteps, float* times) {
    const int64_t N = (int64_t)res * res * res;
    int32_t* occupied = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int64_t n_occ = 0;
    for (int64_t c = 0; c < N; ++c)
        if (binaries[c]) occupied[n_occ++] = (int32_t)c;
    const int64_t n = N / 4;
    const int64_t M = nsxo_occ_num_slots(N, n_occ, warmup);
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int64_t s = 0; s < M; ++s) {
        uint32_t ctr[4] = {(uint32_t)s, 0u, (uint32_t)step, (uint32_t)((uint64_t)step >> 32)}, w[4], w2[4];
        nsxo_philox4x32_10(ctr, key, w);
        ctr[1] = 1u;
        nsxo_philox4x32_10(ctr, key, w2);
        int64_t c;
        if (warmup) c = s;
        else if (s < n) c = (int64_t)(w[0] % (uint32_t)N);
        else if (n < n_occ) c = occupied[w[0] % (uint32_t)n_occ];
        else c = occupied[s - n];
        cell_ids[s] = (int32_t)c;
        const int ijk[3] = {(int)(c / ((int64_t)res * res)), (int)((c / res) % res), (int)(c % res)};
        for (int a = 0; a < 3; ++a) {
            const float x = ((float)ijk[a] + u01(w[1 + a])) / (float)res;
            positions[s * 3 + a] = aabb[a] + x * (aabb[3 + a] - aabb[a]);
        }
        const int32_t t = (int32_t)(w2[0] % (uint32_t)n_timesteps);
        timesteps[s] = t;
        times[s] = n_timesteps > 1 ? (float)t / (float)(n_timesteps - 1) : 0.0f;
    }
    free(occupied);
    return M;
}

/* EMA-max update + threshold.  occs [N] fp32 and binaries [N] (0/1) are updated in place; returns the threshold. */
float nsxo_occ_update(float* occs, uint8_t* binaries, int64_t n_cells, const int32_t* cell_ids,
                      const float* occ_values, int64_t M, float ema_decay, float occ_thre) {
    float* newmax = (float*)malloc(sizeof(float) * (size_t)n_cells);
    for (int64_t c = 0; c < n_cells; ++c) newmax[c] = -1.0f;
    for (int64_t s = 0; s < M; ++s) {
        const int32_t c = cell_ids[s];
        if (c < 0) continue;
        const float v = occ_values[s];
        /* nerfacc: occs[idx] = maximum(occs[idx] * decay, occ) -- a queried cell is decayed whatever its value.
           Densities are >= 0 (trunc_exp); a negative value loses against the decayed one exactly like 0 does, so it
           counts as 0.  A NaN (the kernel's unsigned atomic max on the bit pattern would rank it highest) is taken
           as 0 as well: the cell decays instead of being poisoned -- the ONE deviation from torch.maximum, which
           would propagate the NaN into occs, the mean and every later threshold. */
        const float vv = (v >= 0.0f) ? v : 0.0f;
        if (vv > newmax[c]) newmax[c] = vv;
    }
    double sum = 0.0;
    int64_t cnt = 0;
    for (int64_t c = 0; c < n_cells; ++c) {
        if (newmax[c] >= 0.0f) {
            const float decayed = occs[c] * ema_decay;
            occs[c] = decayed > newmax[c] ? decayed : newmax[c];
        }
        if (occs[c] >= 0.0f) { sum += (double)occs[c]; ++cnt; }
    }
    float thre = cnt > 0 ? (float)(sum / (double)cnt) : NAN;
    if (thre > occ_thre) thre = occ_thre;                  /* torch.clamp(mean, max=occ_thre) */
    for (int64_t c = 0; c < n_cells; ++c) binaries[c] = occs[c] > thre ? 1 : 0;
    free(newmax);
    return thre;
}

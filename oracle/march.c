/*
 * oracle/march.c -- CPU restatement of the occupancy-grid ray marcher and the per-ray scans.
 * TEST INFRASTRUCTURE ONLY (see nsx_oracle.h).
 *
 * Follows the reference call sites
 *   nersemble_volumetric_sampler.py:95-108  (OccGridEstimator.sampling)
 *   nersemble_instant_ngp.py:325-343        (pack_info, render_weight_from_density, accumulate)
 *   models/base.py:224-249                  (flatten_eff_distloss)
 * and restates nerfacc 0.5.2 (ray_aabb_intersect, traverse_grids with the fixed-step lattice, exclusive-sum
 * transmittance) and torch_efficient_distloss from their published algorithms -- both are
 * un-vendored third-party packages: "parity unpinned".
 *
 * Bit-exactness contract: the traversal is pure fp32 with NO fused multiply-add (compiled -ffp-contract=off;
 * the HIP kernel uses `#pragma clang fp contract(off)`), so sample counts, ray indices, cell ids and the t
 * values themselves are bit-identical between this file and the GPU.
 */
#include "nsx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* nerfacc ray_aabb_intersect (slab test).  Returns hit; tmin/tmax written on hit. */
static int ray_aabb(const float* o, const float* d, const float* aabb, float* tmin_o, float* tmax_o) {
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    float tmin, tmax, tmin_t, tmax_t;
    if (inv[0] >= 0) { tmin = (aabb[0] - o[0]) * inv[0]; tmax = (aabb[3] - o[0]) * inv[0]; }
    else             { tmin = (aabb[3] - o[0]) * inv[0]; tmax = (aabb[0] - o[0]) * inv[0]; }
    if (inv[1] >= 0) { tmin_t = (aabb[1] - o[1]) * inv[1]; tmax_t = (aabb[4] - o[1]) * inv[1]; }
    else             { tmin_t = (aabb[4] - o[1]) * inv[1]; tmax_t = (aabb[1] - o[1]) * inv[1]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (inv[2] >= 0) { tmin_t = (aabb[2] - o[2]) * inv[2]; tmax_t = (aabb[5] - o[2]) * inv[2]; }
    else             { tmin_t = (aabb[5] - o[2]) * inv[2]; tmax_t = (aabb[2] - o[2]) * inv[2]; }
    if (tmin > tmax_t || tmin_t > tmax) return 0;
    if (tmin_t > tmin) tmin = tmin_t;
    if (tmax_t < tmax) tmax = tmax_t;
    if (tmax <= 0) return 0;
    *tmin_o = tmin; *tmax_o = tmax;
    return 1;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* One ray through a single-level res^3 boolean grid.  If t0/t1 are NULL only counts.
 * cells (optional) receives the flat cell id of every emitted sample. Returns the sample count. */
static int64_t march_ray(const float* o, const float* d, const float* aabb, const uint8_t* binary, int res,
                         float near_plane, float far_plane, float step, float* t0, float* t1, int32_t* cells) {
    const float eps = 1e-6f;
    float tmin, tmax;
    if (!ray_aabb(o, d, aabb, &tmin, &tmax)) return 0;
    const float this_tmin = fmaxf(tmin, near_plane);
    const float this_tmax = fminf(tmax, far_plane);
    if (this_tmin >= this_tmax) return 0;
    float t_last = near_plane;
    int64_t n = 0;
    /* not continuous: march the lattice (anchored at the near plane) until t_mid is right after this_tmin */
    for (;;) {
        if (t_last + step * 0.5f >= this_tmin) break;
        t_last += step;
    }
    /* setup_traversal */
    float inv[3], voxel[3], tdist[3], delta[3];
    int cur[3], fin[3], stepi[3], over[3];
    const float ts = this_tmin + eps, te = this_tmax - eps;
    for (int a = 0; a < 3; ++a) {
        inv[a] = 1.0f / d[a];
        voxel[a] = (aabb[3 + a] - aabb[a]) / (float)res;
        const float rs = o[a] + d[a] * ts;
        const float re = o[a] + d[a] * te;
        cur[a] = clampi((int)(((rs - aabb[a]) / (aabb[3 + a] - aabb[a])) * (float)res), 0, res - 1);
        fin[a] = clampi((int)(((re - aabb[a]) / (aabb[3 + a] - aabb[a])) * (float)res), 0, res - 1);
        const int idelta = d[a] > 0 ? 1 : 0;
        const float start = (float)(cur[a] + idelta);
        const float tmx = ((aabb[a] + ((start * voxel[a]) - rs)) * inv[a]) + this_tmin;
        tdist[a] = (d[a] == 0.0f) ? this_tmax : tmx;
        const float sf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
        stepi[a] = (int)sf;
        const float dtmp = voxel[a] * inv[a] * sf;
        delta[a] = (d[a] == 0.0f) ? this_tmax : dtmp;
        over[a] = fin[a] + stepi[a];
    }
    for (;;) {
        float t_trav = fminf(tdist[0], fminf(tdist[1], tdist[2]));
        t_trav = fminf(t_trav, this_tmax);
        const int32_t cell = (cur[0] * res + cur[1]) * res + cur[2];
        if (!binary[cell]) {
            for (;;) {
                if (t_last + step * 0.5f >= t_trav) break;
                t_last += step;
            }
        } else {
            for (;;) {
                if (t_last + step * 0.5f >= t_trav) break;
                const float t_next = t_last + step;
                if (t0) { t0[n] = t_last; t1[n] = t_next; }
                if (cells) cells[n] = cell;
                n++;
                t_last = t_next;
                if (t_next >= t_trav) break;
            }
        }
        /* single_traversal */
        if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) {
            cur[0] += stepi[0]; tdist[0] += delta[0];
            if (cur[0] == over[0]) break;
        } else if (tdist[1] < tdist[2]) {
            cur[1] += stepi[1]; tdist[1] += delta[1];
            if (cur[1] == over[1]) break;
        } else {
            cur[2] += stepi[2]; tdist[2] += delta[2];
            if (cur[2] == over[2]) break;
        }
    }
    return n;
}

/* near[R]: per-ray near plane (already jittered by the caller when stratified). */
void nsxo_march_count(const float* rays_o, const float* rays_d, int64_t R, const float* aabb,
                      const uint8_t* binary, int res, const float* near, float far_plane, float step,
                      int64_t* counts) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < R; ++r)
        counts[r] = march_ray(rays_o + 3 * r, rays_d + 3 * r, aabb, binary, res, near[r], far_plane, step,
                              NULL, NULL, NULL);
}

void nsxo_march_fill(const float* rays_o, const float* rays_d, int64_t R, const float* aabb,
                     const uint8_t* binary, int res, const float* near, float far_plane, float step,
                     const int64_t* starts, float* t0, float* t1, int64_t* ray_idx, int32_t* cells) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < R; ++r) {
        int64_t n = march_ray(rays_o + 3 * r, rays_d + 3 * r, aabb, binary, res, near[r], far_plane, step,
                              t0 + starts[r], t1 + starts[r], cells ? cells + starts[r] : NULL);
        for (int64_t i = 0; i < n; ++i) ray_idx[starts[r] + i] = r;
    }
}

/* nerfacc render_transmittance_from_density + weights, per packed ray, double accumulation. */
void nsxo_render_weights(const float* t0, const float* t1, const float* sigma, const int64_t* packed /*[R][2]*/,
                         int64_t R, float* weights, float* trans, float* alphas) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < R; ++r) {
        double cum = 0.0;
        for (int64_t i = packed[2 * r]; i < packed[2 * r] + packed[2 * r + 1]; ++i) {
            const double sdt = (double)sigma[i] * (double)(t1[i] - t0[i]);
            const double T = exp(-cum), a = 1.0 - exp(-sdt);
            if (trans) trans[i] = (float)T;
            if (alphas) alphas[i] = (float)a;
            weights[i] = (float)(T * a);
            cum += sdt;
        }
    }
}

/* backward of weights w.r.t. sigma:  ds_i = gw_i * T_{i+1} - sum_{j>i} gw_j w_j ; dsigma = ds * dt */
void nsxo_render_weights_bwd(const float* t0, const float* t1, const float* sigma, const int64_t* packed, int64_t R,
                             const float* gw, float* dsigma) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < R; ++r) {
        const int64_t s = packed[2 * r], n = packed[2 * r + 1];
        double cum = 0.0;
        double* T = (double*)malloc(sizeof(double) * (size_t)(n + 1));
        double* w = (double*)malloc(sizeof(double) * (size_t)(n + 1));
        for (int64_t i = 0; i < n; ++i) {
            const double sdt = (double)sigma[s + i] * (double)(t1[s + i] - t0[s + i]);
            T[i] = exp(-cum);
            w[i] = T[i] * (1.0 - exp(-sdt));
            cum += sdt;
        }
        T[n] = exp(-cum);
        double suffix = 0.0;
        for (int64_t i = n - 1; i >= 0; --i) {
            const double ds = (double)gw[s + i] * T[i + 1] - suffix;
            dsigma[s + i] = (float)(ds * (double)(t1[s + i] - t0[s + i]));
            suffix += (double)gw[s + i] * w[i];
        }
        free(T); free(w);
    }
}

/* nerfacc accumulate_along_rays: out[r][c] = sum_i w_i * v[i][c]  (v NULL -> C must be 1, out = sum w) */
void nsxo_accumulate(const float* w, const float* v, int C, const int64_t* packed, int64_t R, float* out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < R; ++r) {
        double acc[16] = {0};
        for (int64_t i = packed[2 * r]; i < packed[2 * r] + packed[2 * r + 1]; ++i)
            for (int c = 0; c < C; ++c) acc[c] += (double)w[i] * (v ? (double)v[i * C + c] : 1.0);
        for (int c = 0; c < C; ++c) out[r * C + c] = (float)acc[c];
    }
}

/* torch_efficient_distloss.flatten_eff_distloss forward + backward wrt w (A.5). ray segments given packed.
 * loss = (sum_i 1/3 * interval_i * w_i^2 + 2 w_i (m_i W_pre_i - WM_pre_i)) / n_rays */
double nsxo_distloss(const float* w, const float* m, const float* interval, const int64_t* packed, int64_t R,
                     int64_t n_rays, float* grad_w /* may be NULL */) {
    double total = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total)
    for (int64_t r = 0; r < R; ++r) {
        const int64_t s = packed[2 * r], n = packed[2 * r + 1];
        double W = 0.0, WM = 0.0;
        for (int64_t i = 0; i < n; ++i) { W += w[s + i]; WM += (double)w[s + i] * m[s + i]; }
        double wpre = 0.0, wmpre = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double wi = w[s + i], mi = m[s + i], iv = interval[s + i];
            total += (1.0 / 3.0) * iv * wi * wi + 2.0 * wi * (mi * wpre - wmpre);
            if (grad_w) {
                const double wsuf = W - wpre - wi, wmsuf = WM - wmpre - wi * mi;
                grad_w[s + i] = (float)(((2.0 / 3.0) * iv * wi + 2.0 * (mi * (wpre - wsuf) + (wmsuf - wmpre)))
                                        / (double)n_rays);
            }
            wpre += wi; wmpre += wi * mi;
        }
    }
    return total / (double)n_rays;
}

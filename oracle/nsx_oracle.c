/*
 * oracle/nsx_oracle.c -- CPU restatement of the hash-grid / HashEnsemble part of
 * the hot path.  TEST INFRASTRUCTURE ONLY (see nsx_oracle.h header comment).
 *
 * Follows:
 *   - reference call sites  src/nersemble/nerfstudio/field_components/hash_ensemble.py:31-50
 *     (encoding config), :93-158 (HashEnsemble.forward: per-encoding lookup, rearrange
 *     'b c (l p f) -> b (l f) (c p)', blend einsum 'bdh,bh->bd').
 *   - tiny-cuda-nn HashGrid (git HEAD, un-vendored, "parity unpinned"): geometry
 *     grid_scale/grid_resolution, pos_fract (fmaf(scale,x,0.5)), grid_index (dense
 *     stride walk, coherent-prime hash {1, 2654435761, 805459861}, % level size),
 *     trilinear weights, AoS fp16 parameter layout, backward scatter + dy_dx.
 *     Restated from the published algorithm.
 *
 * Numerics of this oracle ("the spec" the HIP kernels are held to):
 *   integer outputs (cell coordinates, entry indices)  -> bit exact
 *   fp outputs: exact fp32 fmaf for positions; interpolation/blend sums carried in
 *   double and rounded once -> HIP (fp32 accumulate) must agree to fp16/fp32 rounding.
 */
#include "nsx_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* ---------- fp16 <-> fp32, round-to-nearest-even, software (portable) ---------- */
uint16_t nsxo_f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t  exp  = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0));
    exp = exp - 127 + 15;
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        int shift = 14 - exp;                       /* 14..24 */
        uint32_t hm = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t hm = mant >> 13;
    uint32_t rem = mant & 0x1fffu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)exp << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) h++;   /* carry may roll into exp: correct */
    return h;
}

float nsxo_h2f(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) x = sign;
        else {
            int e = -1;
            do { e++; mant <<= 1; } while (!(mant & 0x400u));
            mant &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7f800000u | (mant << 13);
    } else {
        x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    }
    float f; memcpy(&f, &x, 4);
    return f;
}

/* ---------- geometry (tcnn GridEncodingTemplated ctor; A.1) ---------- */
static uint32_t next_multiple_u32(uint32_t v, uint32_t m) { return ((v + m - 1) / m) * m; }

void nsxo_grid_geometry(int n_levels, float per_level_scale, int base_resolution,
                        int log2_hashmap_size, nsxo_grid_geom* g) {
    memset(g, 0, sizeof(*g));
    g->n_levels = n_levels;
    g->log2_hashmap_size = log2_hashmap_size;
    g->base_resolution = base_resolution;
    g->per_level_scale = per_level_scale;
    const float log2_scale = log2f(per_level_scale);
    uint32_t offset = 0;
    for (int l = 0; l < n_levels; ++l) {
        float scale = exp2f((float)l * log2_scale) * (float)base_resolution - 1.0f;
        uint32_t res = (uint32_t)ceilf(scale) + 1u;
        uint32_t max_params = 0xffffffffu / 2u;
        uint32_t n;
        if (powf((float)res, 3.0f) > (float)max_params) n = max_params;
        else n = res * res * res;
        n = next_multiple_u32(n, 8u);
        uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        g->scale[l] = scale; g->res[l] = res; g->size[l] = n; g->offset[l] = offset;
        offset += n;
    }
    g->offset[n_levels] = offset;
}

/* ---------- per (sample, level) cell coordinates, weights, indices ---------- */
static inline void cell_of(float scale, const float* x, uint32_t* c0, float* w) {
    for (int d = 0; d < 3; ++d) {
        float p = fmaf(scale, x[d], 0.5f);
        float fl = floorf(p);
        c0[d] = (uint32_t)(int32_t)fl;
        w[d] = p - fl;
    }
}

static inline uint32_t entry_index(const uint32_t c[3], uint32_t res, uint32_t size) {
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= size; ++d) { index += c[d] * stride; stride *= res; }
    if (size < stride)
        index = (c[0] * 1u) ^ (c[1] * 2654435761u) ^ (c[2] * 805459861u);
    return index % size;
}

void nsxo_hashgrid_indices(const float* x, int64_t B, const nsxo_grid_geom* g,
                           uint32_t* idx, float* w) {
    const int L = g->n_levels;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int l = 0; l < L; ++l) {
            uint32_t c0[3]; float wl[3];
            cell_of(g->scale[l], x + 3 * b, c0, wl);
            for (int d = 0; d < 3; ++d) w[(b * L + l) * 3 + d] = wl[d];
            for (int k = 0; k < 8; ++k) {
                uint32_t c[3] = { c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1) };
                idx[(b * L + l) * 8 + k] = entry_index(c, g->res[l], g->size[l]);
            }
        }
    }
}

void nsxo_hashgrid_fwd(const float* x, int64_t B, const uint16_t* table, int F_enc,
                       const nsxo_grid_geom* g, uint16_t* out) {
    const int L = g->n_levels;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int l = 0; l < L; ++l) {
            uint32_t c0[3]; float wl[3];
            cell_of(g->scale[l], x + 3 * b, c0, wl);
            double acc[8] = {0};
            for (int k = 0; k < 8; ++k) {
                uint32_t c[3] = { c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1) };
                uint32_t e = entry_index(c, g->res[l], g->size[l]);
                double wk = 1.0;
                for (int d = 0; d < 3; ++d) wk *= ((k >> d) & 1) ? (double)wl[d] : 1.0 - (double)wl[d];
                const uint16_t* row = table + ((size_t)g->offset[l] + e) * (size_t)F_enc;
                for (int j = 0; j < F_enc; ++j) acc[j] += wk * (double)nsxo_h2f(row[j]);
            }
            for (int j = 0; j < F_enc; ++j)
                out[(size_t)b * (size_t)(L * F_enc) + (size_t)(l * F_enc + j)] = nsxo_f2h((float)acc[j]);
        }
    }
}

/* layout helpers for the ensemble (hash_ensemble.py:84-86, :107-112) */
static inline void ens_layout(int H, int* F_enc, int* P, int* C) {
    const int total = 2 * H;
    *F_enc = total >= 8 ? 8 : total;
    *P = total >= 8 ? 4 : H;
    *C = (total + 7) / 8;
}

void nsxo_ensemble_fwd(const float* x, int64_t B, const uint16_t* tables, int H,
                       const nsxo_grid_geom* g, const float* codew, uint16_t* out) {
    const int L = g->n_levels;
    int F_enc, P, C; ens_layout(H, &F_enc, &P, &C);
    const size_t enc_elems = (size_t)g->offset[L] * (size_t)F_enc;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float code16[64];
        for (int h = 0; h < H; ++h) code16[h] = nsxo_h2f(nsxo_f2h(codew[b * H + h]));
        for (int l = 0; l < L; ++l) {
            uint32_t c0[3]; float wl[3];
            cell_of(g->scale[l], x + 3 * b, c0, wl);
            double acc[2] = {0, 0};
            for (int k = 0; k < 8; ++k) {
                uint32_t c[3] = { c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1) };
                uint32_t e = entry_index(c, g->res[l], g->size[l]);
                double wk = 1.0;
                for (int d = 0; d < 3; ++d) wk *= ((k >> d) & 1) ? (double)wl[d] : 1.0 - (double)wl[d];
                for (int h = 0; h < H; ++h) {
                    const int c_enc = h / P, p = h % P;
                    const uint16_t* row = tables + (size_t)c_enc * enc_elems
                                        + ((size_t)g->offset[l] + e) * (size_t)F_enc + (size_t)(p * 2);
                    acc[0] += wk * (double)nsxo_h2f(row[0]) * (double)code16[h];
                    acc[1] += wk * (double)nsxo_h2f(row[1]) * (double)code16[h];
                }
            }
            out[(size_t)b * (size_t)(L * 2) + (size_t)(l * 2 + 0)] = nsxo_f2h((float)acc[0]);
            out[(size_t)b * (size_t)(L * 2) + (size_t)(l * 2 + 1)] = nsxo_f2h((float)acc[1]);
        }
    }
}

/* ---- CPU BASELINE port of the same forward (bench.py cpu_baseline only; NOT the checker) -------------------------
 * What a CPU implementation tuned like one would be looks like: fp32 accumulation, fp16 decode through a 64 K-entry
 * table, one 16-byte row read per (encoding, corner), the whole batch spread over the cores.  Held to the checker above
 * in tests/test_oracle_hash.py (fp32 vs double accumulation: within one fp16 ulp). */
static float g_h2f_lut[65536];
static int g_h2f_lut_ready = 0;

void nsxo_ensemble_fwd_fast(const float* x, int64_t B, const uint16_t* tables, int H,
                            const nsxo_grid_geom* g, const float* codew, uint16_t* out) {
    const int L = g->n_levels;
    int F_enc, P, C; ens_layout(H, &F_enc, &P, &C);
    const size_t enc_elems = (size_t)g->offset[L] * (size_t)F_enc;
    if (!g_h2f_lut_ready) {
        for (uint32_t i = 0; i < 65536u; ++i) g_h2f_lut[i] = nsxo_h2f((uint16_t)i);
        g_h2f_lut_ready = 1;
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t b = 0; b < B; ++b) {
        float code16[64];
        for (int h = 0; h < H; ++h) code16[h] = g_h2f_lut[nsxo_f2h(codew[b * H + h])];
        for (int l = 0; l < L; ++l) {
            uint32_t c0[3]; float wl[3];
            cell_of(g->scale[l], x + 3 * b, c0, wl);
            float acc0 = 0.f, acc1 = 0.f;
            for (int k = 0; k < 8; ++k) {
                uint32_t c[3] = { c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1) };
                const uint32_t e = entry_index(c, g->res[l], g->size[l]);
                const float wk = ((k & 1) ? wl[0] : 1.f - wl[0]) * (((k >> 1) & 1) ? wl[1] : 1.f - wl[1])
                               * (((k >> 2) & 1) ? wl[2] : 1.f - wl[2]);
                float t0 = 0.f, t1 = 0.f;
                for (int ce = 0; ce < C; ++ce) {
                    const uint16_t* row = tables + (size_t)ce * enc_elems + ((size_t)g->offset[l] + e) * (size_t)F_enc;
                    const float* cd = code16 + ce * P;
                    for (int p = 0; p < P && ce * P + p < H; ++p) {
                        t0 += g_h2f_lut[row[2 * p]] * cd[p];
                        t1 += g_h2f_lut[row[2 * p + 1]] * cd[p];
                    }
                }
                acc0 += wk * t0;
                acc1 += wk * t1;
            }
            out[(size_t)b * (size_t)(L * 2) + (size_t)(l * 2 + 0)] = nsxo_f2h(acc0);
            out[(size_t)b * (size_t)(L * 2) + (size_t)(l * 2 + 1)] = nsxo_f2h(acc1);
        }
    }
}

void nsxo_ensemble_bwd(const float* x, int64_t B, const uint16_t* tables, int H,
                       const nsxo_grid_geom* g, const float* codew, const float* dout,
                       float* dtable, float* dcodew, float* dx) {
    const int L = g->n_levels;
    int F_enc, P, C; ens_layout(H, &F_enc, &P, &C);
    const size_t enc_elems = (size_t)g->offset[L] * (size_t)F_enc;
    /* table gradient is a scatter with collisions: accumulate in double, serial over b
     * (deterministic), then add into the caller's fp32 buffer. */
    double* acc_tab = NULL;
    if (dtable) acc_tab = (double*)calloc((size_t)C * enc_elems, sizeof(double));
    for (int64_t b = 0; b < B; ++b) {
        float code16[64];
        double dcode[64];
        for (int h = 0; h < H; ++h) { code16[h] = nsxo_h2f(nsxo_f2h(codew[b * H + h])); dcode[h] = 0.0; }
        double dxa[3] = {0, 0, 0};
        for (int l = 0; l < L; ++l) {
            uint32_t c0[3]; float wl[3];
            cell_of(g->scale[l], x + 3 * b, c0, wl);
            const double g0 = (double)dout[b * (L * 2) + l * 2 + 0];
            const double g1 = (double)dout[b * (L * 2) + l * 2 + 1];
            for (int k = 0; k < 8; ++k) {
                uint32_t c[3] = { c0[0] + (k & 1), c0[1] + ((k >> 1) & 1), c0[2] + ((k >> 2) & 1) };
                uint32_t e = entry_index(c, g->res[l], g->size[l]);
                double wd[3];
                for (int d = 0; d < 3; ++d) wd[d] = ((k >> d) & 1) ? (double)wl[d] : 1.0 - (double)wl[d];
                const double wk = wd[0] * wd[1] * wd[2];
                double blended = 0.0;   /* sum_h sum_f g_f * table * code */
                for (int h = 0; h < H; ++h) {
                    const int c_enc = h / P, p = h % P;
                    const size_t at = (size_t)c_enc * enc_elems
                                    + ((size_t)g->offset[l] + e) * (size_t)F_enc + (size_t)(p * 2);
                    const double t0 = (double)nsxo_h2f(tables[at]), t1 = (double)nsxo_h2f(tables[at + 1]);
                    if (acc_tab) {
                        acc_tab[at]     += wk * g0 * (double)code16[h];
                        acc_tab[at + 1] += wk * g1 * (double)code16[h];
                    }
                    const double gt = g0 * t0 + g1 * t1;
                    dcode[h] += wk * gt;
                    blended += gt * (double)code16[h];
                }
                for (int d = 0; d < 3; ++d) {
                    const double sgn = ((k >> d) & 1) ? 1.0 : -1.0;
                    const double other = wd[(d + 1) % 3] * wd[(d + 2) % 3];
                    dxa[d] += (double)g->scale[l] * sgn * other * blended;
                }
            }
        }
        if (dcodew) for (int h = 0; h < H; ++h) dcodew[b * H + h] = (float)dcode[h];
        if (dx) for (int d = 0; d < 3; ++d) dx[b * 3 + d] = (float)dxa[d];
    }
    if (dtable) {
        const size_t n = (size_t)C * enc_elems;
        for (size_t i = 0; i < n; ++i) dtable[i] += (float)acc_tab[i];
        free(acc_tab);
    }
}

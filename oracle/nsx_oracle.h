/*
 * oracle/nsx_oracle.h -- CPU restatement (plain C) of the per-sample hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nersemble_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it (as the checker / the reported CPU baseline).
 *
 * Parity status: the algorithms restated here live in un-vendored third-party
 * packages of the reference (tiny-cuda-nn @ git HEAD, nerfacc 0.5.2,
 * torch_efficient_distloss), whose source is NOT under /root/reference and for
 * which the reference holds no tests -> those parts are "parity unpinned"
 * (restated from the published algorithms).  The
 * reference-OWNED glue around them (HashEnsemble rearrange/window/blend,
 * windowed PE, se3_exp_map, schedulers, chunker, dist-loss selection) is pinned
 * by tests/golden/ fixtures generated from the reference's own Python
 * (tests/golden/make_golden.py).
 */
#ifndef NSX_ORACLE_H
#define NSX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NSXO_MAX_LEVELS 32

typedef struct {
    int32_t  n_levels;
    int32_t  log2_hashmap_size;
    int32_t  base_resolution;
    float    per_level_scale;
    float    scale[NSXO_MAX_LEVELS];      /* grid_scale(l)                       */
    uint32_t res[NSXO_MAX_LEVELS];        /* grid_resolution(scale)              */
    uint32_t size[NSXO_MAX_LEVELS];       /* entries in level (params_in_level)  */
    uint32_t offset[NSXO_MAX_LEVELS + 1]; /* entry offset of each level          */
} nsxo_grid_geom;

uint16_t nsxo_f2h(float f);
float    nsxo_h2f(uint16_t h);

/* tcnn GridEncoding constructor geometry (hash_ensemble.py:31-50 config). */
void nsxo_grid_geometry(int n_levels, float per_level_scale, int base_resolution,
                        int log2_hashmap_size, nsxo_grid_geom* g);

/* Per (sample, level): the 8 corner entry indices (uint32, level-local) and the
 * 3 fractional weights.  idx[B][L][8], w[B][L][3]. */
void nsxo_hashgrid_indices(const float* x, int64_t B, const nsxo_grid_geom* g,
                           uint32_t* idx, float* w);

/* One tcnn HashGrid encoding forward: table fp16 [total_entries][F_enc] (AoS),
 * out fp16 [B][L*F_enc].  fp32 accumulate, one rounding to fp16. */
void nsxo_hashgrid_fwd(const float* x, int64_t B, const uint16_t* table, int F_enc,
                       const nsxo_grid_geom* g, uint16_t* out);

/* Fused HashEnsemble forward (hash_ensemble.py:93-158 in one pass):
 * tables: C = ceil(2H/8) encodings in tcnn layout, each [total][F_enc] fp16,
 * concatenated; logical grid h = c*P + p, feature j = p*F + f.
 * codew[B][H] fp32 = conditioning code already multiplied by the grid window
 * (host does window/soft-transition, exactly as hash_ensemble.py:119-138).
 * out fp16 [B][L*2]. code is rounded to fp16 before use (hash_ensemble.py:155). */
void nsxo_ensemble_fwd(const float* x, int64_t B, const uint16_t* tables, int H,
                       const nsxo_grid_geom* g, const float* codew, uint16_t* out);
/* CPU-baseline port of the same forward (fp32 accumulate, table-driven fp16 decode): bench.py cpu_baseline only. */
void nsxo_ensemble_fwd_fast(const float* x, int64_t B, const uint16_t* tables, int H,
                       const nsxo_grid_geom* g, const float* codew, uint16_t* out);

/* Backward of the fused ensemble: given dout fp32 [B][L*2], produces
 * dtable fp32 (tcnn layout, same shape as tables; ACCUMULATED into),
 * dcodew fp32 [B][H] (gradient w.r.t. the windowed code), dx fp32 [B][3]. */
void nsxo_ensemble_bwd(const float* x, int64_t B, const uint16_t* tables, int H,
                       const nsxo_grid_geom* g, const float* codew, const float* dout,
                       float* dtable, float* dcodew, float* dx);

/* ---- ray marching / per-ray scans (oracle/march.c) ---- */
void nsxo_march_count(const float* rays_o, const float* rays_d, int64_t R, const float* aabb,
                      const uint8_t* binary, int res, const float* near, float far_plane, float step,
                      int64_t* counts);
void nsxo_march_fill(const float* rays_o, const float* rays_d, int64_t R, const float* aabb,
                     const uint8_t* binary, int res, const float* near, float far_plane, float step,
                     const int64_t* starts, float* t0, float* t1, int64_t* ray_idx, int32_t* cells);
void nsxo_render_weights(const float* t0, const float* t1, const float* sigma, const int64_t* packed,
                         int64_t R, float* weights, float* trans, float* alphas);
void nsxo_render_weights_bwd(const float* t0, const float* t1, const float* sigma, const int64_t* packed,
                             int64_t R, const float* gw, float* dsigma);
void nsxo_accumulate(const float* w, const float* v, int C, const int64_t* packed, int64_t R, float* out);
double nsxo_distloss(const float* w, const float* m, const float* interval, const int64_t* packed, int64_t R,
                     int64_t n_rays, float* grad_w);

/* ---- occupancy-grid update (oracle/occgrid.c) ---- */
void nsxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int64_t nsxo_occ_num_slots(int64_t n_cells, int64_t n_occ, int warmup);
int64_t nsxo_occ_sample_cells(const uint8_t* binaries, int res, const float* aabb, int warmup, uint64_t seed,
                              int64_t step, int n_timesteps, int32_t* cell_ids, float* positions,
                              int32_t* timesteps, float* times);
float nsxo_occ_update(float* occs, uint8_t* binaries, int64_t n_cells, const int32_t* cell_ids,
                      const float* occ_values, int64_t M, float ema_decay, float occ_thre);

#ifdef __cplusplus
}
#endif
#endif

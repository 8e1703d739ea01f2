/*
 * include/nsx.h -- C ABI of libnsx.so, the MI355X (gfx950) native library for the
 * NeRSemble per-sample hot path.
 *
 * The reference (tobias-kirschstein/nersemble) is pure Python and reaches its native
 * code through three third-party Python packages; this library supplies those native
 * entry points.  Every function documents the reference interface it replaces as
 * file:line relative to the reference repository root.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch / C++ types.
 *   - All data pointers are DEVICE pointers owned by the caller (borrowed for the call,
 *     never retained or freed); `stream` is a hipStream_t passed as void*.
 *   - Return value: 0 = NSX_OK, negative = error; nsx_last_error() returns a
 *     thread-local human readable message.  No exceptions cross the ABI.
 *   - Kernels are enqueued on `stream`; no hidden synchronisation unless stated.
 *   - fp16 buffers are IEEE binary16 bit patterns (uint16_t storage).
 */
#ifndef NSX_H
#define NSX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NSX_OK 0
#define NSX_ERR_INVALID (-1)
#define NSX_ERR_HIP (-2)
#define NSX_ERR_UNSUPPORTED (-3)

#define NSX_MAX_LEVELS 32
#define NSX_MAX_SLOTS 64
#define NSX_MAX_ADAM_SLOTS 192   /* gradient planes nsx_adam_hash_factored(_consume) reads (level-parallel runs: one per
                                   (source rank, code row), engine/level_parallel.py) */
#define NSX_VERSION 128

typedef uint16_t nsx_half;

/* Multi-resolution hash-grid geometry.  Replaces the tcnn HashGrid constructor reached
 * from hash_ensemble.py:41-50 (TCNNHashEncodingConfig.setup). */
typedef struct nsx_grid_geom {
    int32_t  n_levels;
    int32_t  log2_hashmap_size;
    int32_t  base_resolution;
    float    per_level_scale;
    float    scale[NSX_MAX_LEVELS];      /* exp2f(l*log2f(s))*base - 1                       */
    uint32_t res[NSX_MAX_LEVELS];        /* ceilf(scale)+1                                   */
    uint32_t size[NSX_MAX_LEVELS];       /* entries of the level                             */
    uint32_t offset[NSX_MAX_LEVELS + 1]; /* first entry of each level; [n_levels] = total    */
    uint32_t hashed[NSX_MAX_LEVELS];     /* 1: coherent-prime hash & (size-1); 0: dense walk */
} nsx_grid_geom;

int         nsx_version(void);
const char* nsx_last_error(void);

/* Launch-shape options: how many workgroups a few bandwidth- or register-bound kernels launch per CU.  They never change a
 * result, only how a kernel shares the device with its neighbours (the table optimizer beside the next step's
 * marching).  Process-wide atomics with the defaults the measurements chose; a value outside the option's range is an
 * error.  These replace the NSX_* environment variables of rounds 2-4: no entry point of this library reads the environment
 * (tests/test_boundary.py checks the binary for getenv). */
#define NSX_OPT_ADAM_BLOCKS_PER_CU 0           /* nsx_adam_hash_factored(_consume): 1..8, default 5 */
#define NSX_OPT_MLP_BWD_HALF_BLOCKS_PER_CU 1   /* nsx_mlp_bwd with a hidden matrix: blocks per CU x 2, 1..8, default 2 */
#define NSX_OPT_MLP_BWD0_HALF_BLOCKS_PER_CU 2  /* nsx_mlp_bwd without one: blocks per CU x 2, 1..8, default 2 */
#define NSX_OPT_LP_ONE_LAUNCH 3                /* nsx_lp_fwd_run / nsx_lp_bwd_run: 1 (default) = all source ranks in ONE launch
                                                  (grid.y = source rank), 0 = one launch per source rank */
#define NSX_OPT_COUNT 4
int nsx_set_option(int option, int value);
int nsx_get_option(int option);                /* the current value, or NSX_ERR_INVALID */

/* Host-only.  hash_ensemble.py:31-50. */
int nsx_grid_geometry(int n_levels, float per_level_scale, int base_resolution,
                      int log2_hashmap_size, nsx_grid_geom* out);

/* Device-side element counts.  The number of samples a step keeps after visibility pruning is known on the device long
 * before the host could read it back.  Every per-sample entry point of this library therefore takes, right in front of
 * `stream`, an optional `const int64_t* n_device` (device pointer to ONE int64, or NULL): buffers are allocated for the row
 * count passed by value (the capacity), the launch is sized for it, and the kernels read *n_device WHEN THEY RUN and
 * process only the first min(*n_device, capacity) rows -- the call can be enqueued without a host synchronisation.
 * NULL = every row.  Rows beyond *n_device are neither read nor written (nsx_gather_rows zero-fills them, see there).
 * Taken by: nsx_sample_positions, nsx_normalise_bwd, nsx_density_fwd/bwd, nsx_gather_rows,
 * nsx_hash_ensemble_fwd/bwd(_factored/_codesum/_scatter), nsx_mlp_fwd/bwd, nsx_deform_fwd/bwd, nsx_ray_histogram; per-ray
 * entry points follow through packed_info.  (Rounds 2-3 attached the pointer through a thread-local
 * nsx_device_count_begin/_end scope matched by row count; removed: the ABI holds no hidden state.) */

/* Padded number of grids used by the interleaved layout (next power of two >= H; the reference's H is any value with
 * 2H <= 8 or 2H a multiple of 8, hash_ensemble.py:80-82). */
int nsx_padded_grids(int H);

/* ---- parameter layout conversion (checkpoint compatibility) --------------------------------
 * tcnn layout  : C = ceil(2H/8) encodings, each [total_entries][F_enc] (F_enc = 8, or 2H if
 *                2H < 8), feature j = p*2+f of encoding c belongs to logical grid h = c*P+p
 *                (hash_ensemble.py:84-112; state-dict keys
 *                field.hash_ensemble.hash_encodings.{c}.params).
 * native layout: [total_entries][2][Hp] fp16 -- all grids of one entry are contiguous
 *                (Hp = nsx_padded_grids(H); one entry = 4*Hp bytes = one 128-B line at H=32).
 * src/dst tcnn buffers are fp32 (the reference keeps fp32 master params). */
int nsx_tables_from_tcnn(const float* tcnn_params, int H, const nsx_grid_geom* g,
                         nsx_half* native_f16, float* native_master_f32 /* may be NULL */,
                         void* stream);
int nsx_tables_to_tcnn(const float* native_master_f32, int H, const nsx_grid_geom* g,
                       float* tcnn_params, void* stream);
/* fp32 native gradient/master -> tcnn layout is the same permutation (use nsx_tables_to_tcnn). */

/* ---- HashEnsemble ---------------------------------------------------------------------------
 * Replaces HashEnsemble.forward (hash_ensemble.py:93-158): the C tcnn HashGrid launches
 * (:102-104), torch.stack (:106), einops rearrange (:112), grid window (:133-138) and the
 * blend einsum (:155-156) as ONE kernel that never materialises [B, 32, H].
 *   x          [B][3] fp32 in [0,1)  (the field zeroes out-of-box samples first,
 *              nersemble_nerfacto_field.py:268-269)
 *   tables     native layout fp16
 *   code       fp32 rows of H values; row of sample b = code[(code_index ? code_index[b] : b) * code_stride]
 *              (code_index lets the caller pass the [T][H] time embedding + per-sample timestep instead
 *              of the gathered [B][H] copy, nersemble_instant_ngp.py:310-312)
 *   window     [H] fp32 or NULL: per-grid window multiplied onto the code (hash_ensemble.py:133-138)
 *   out        [B][2*n_levels] fp16
 */
int nsx_hash_ensemble_fwd(const float* x, int64_t B, const nsx_half* tables, int H,
                          const nsx_grid_geom* g, const float* code, int64_t code_stride,
                          const int32_t* code_index, const float* window, nsx_half* out,
                          const int64_t* n_device, void* stream);

/* Backward of the above (tcnn kernel_grid_backward + kernel_grid_backward_input + einsum/rearrange
 * backward, reached through autograd from hash_ensemble.py:102-156).
 *   dout       [B][2*n_levels] fp32 (upstream gradient, already loss-scaled by the caller if desired)
 *   dtables    native layout fp32, ACCUMULATED into with atomics (caller zeroes); may be NULL
 *   dcode      [B][H] fp32: gradient w.r.t. the (windowed) per-sample code row; may be NULL
 *   dx         [B][3] fp32; may be NULL
 */
int nsx_hash_ensemble_bwd(const float* x, int64_t B, const nsx_half* tables, int H,
                          const nsx_grid_geom* g, const float* code, int64_t code_stride,
                          const int32_t* code_index, const float* window, const float* dout,
                          float* dtables, float* dcode, float* dx, const int64_t* n_device, void* stream);

/* Factored table gradient -- the MI355X-native backward used when the per-sample code is a row of a SMALL
 * table (the <= 24 distinct time codes of a training batch, nersemble_instant_ngp.py:310-312).  Since
 *   dL/dtable[e][f][h] = sum_slot G[e][slot][f] * code'[slot][h],   G[e][slot][f] = sum_{b in slot} w_b * dout_b[f],
 * the kernel scatters 2 scalars per (sample, level, corner) into G instead of 2H table values (32x fewer
 * atomics at H=32); nsx_hash_grad_expand then produces the native-layout fp32 table gradient.
 *   code_table [n_slots][code_stride] fp32, code_slot [B] int32 in [0, n_slots), n_slots <= NSX_MAX_SLOTS
 *   G          [n_slots][total_entries][2] fp32, ACCUMULATED into with atomics (caller zeroes); may be NULL
 *   dcode      [B][H] fp32 per-sample gradient w.r.t. the windowed code row; may be NULL.  dx [B][3]; may be NULL
 *   nonfinite  device float, set to 1 when a value added to G was inf/NaN (never cleared here); may be NULL.
 *              This is GradScaler's inf check on the table gradient (nersemble_trainer.py:186) without a pass over G:
 *              G holds a non-finite value iff one was added to it (sums of finite fp32 terms of this size cannot
 *              overflow: |dout| <= 65504 * w, w <= 1, <= 2^23 terms). */
int nsx_hash_ensemble_bwd_factored(const float* x, int64_t B, const nsx_half* tables, int H,
                                   const nsx_grid_geom* g, const float* code_table, int64_t code_stride,
                                   int n_slots, const int32_t* code_slot, const float* window,
                                   const float* dout, float* G, float* dcode, float* dx, float* nonfinite,
                                   const int64_t* n_device, void* stream);
/* The factored backward with the CODE gradient reduced inside the kernel.  Once the coarse-to-fine window is open
 * (window_hash_encodings > 1: steps 40 000 ... 300 000 of the reference's schedule, train_nersemble.py:77-78) the time
 * codes are trained, and autograd through hash_ensemble.py:125-138,155-156 + the nn.Embedding lookup
 * (nersemble_instant_ngp.py:300-318) sums the per-sample code gradients into the <= 24 rows a batch uses.  Here the kernel
 * accumulates those sums per block in LDS (runs of samples with equal rows merged with DPP first), writes one partial
 * per block, and a one-block-per-row second stage adds the partials and applies the window's chain rule:
 *   dcode_rows [n_slots][H] fp32 = window[h] * sum_{b: code_slot[b] = row} dL/dcode'_b[h]     (OVERWRITTEN; required)
 *   scratch    float[nsx_hash_codesum_scratch_floats(n_slots, H)]                             (block partials)
 * Everything else as nsx_hash_ensemble_bwd_factored (G may be NULL: gather half only).  No [B][H] tensor exists. */
int64_t nsx_hash_codesum_scratch_floats(int n_slots, int H);
int nsx_hash_ensemble_bwd_codesum(const float* x, int64_t B, const nsx_half* tables, int H,
                                  const nsx_grid_geom* g, const float* code_table, int64_t code_stride,
                                  int n_slots, const int32_t* code_slot, const float* window,
                                  const float* dout, float* G, float* dcode_rows, float* scratch, float* dx,
                                  float* nonfinite, const int64_t* n_device, void* stream);
/* The two halves of the factored backward as separate launches, so that they can run on separate streams:
 *   nsx_hash_ensemble_bwd_factored(..., G = NULL, ...)   the GATHER half: dcode and dx only (bandwidth-bound table reads)
 *   nsx_hash_ensemble_bwd_scatter                        the SCATTER half: G only.  It needs neither the tables nor the
 *                                                        codes (the gradient factors through the code slot), is bound by
 *                                                        the rate of memory-side fp32 atomics and uses ~1/3 of the fused
 *                                                        kernel's registers, so it overlaps the gather half and the
 *                                                        deformation field's backward (which needs only the gather's dx).
 * Same sums as the fused kernel up to the order of the atomics.  blocks_per_cu caps the persistent grid (<= 0: 8) --
 * a small value leaves the CU's registers to the kernels it runs beside. */
int nsx_hash_ensemble_bwd_scatter(const float* x, int64_t B, const nsx_grid_geom* g, int n_slots,
                                  const int32_t* code_slot, const float* dout, float* G, float* nonfinite,
                                  int blocks_per_cu, const int64_t* n_device, void* stream);
/* dtables (native fp32) = (accumulate ? dtables : 0) + expand(G, code_table*window): the dense table gradient that
 * autograd would have produced through hash_ensemble.py:155-156 (einsum) and the tcnn encodings' backward; only for
 * callers that want a materialised .grad (torch optimizers, the dense all-reduce path). */
int nsx_hash_grad_expand(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                         const float* window, int H, const nsx_grid_geom* g, float* dtables, int accumulate,
                         void* stream);

/* Eval-time fast path (no reference counterpart -- hash_ensemble.py:155-158 blends per sample):
 * all rays of an evaluation image share one time code, so  blended[e][f] = fp16(sum_h fp16(code_h window_h) *
 * tables[e][f][h])  is formed once per image (reads the 0.8 GB tables once) and the image is rendered with
 * nsx_hashgrid_fwd(F = 2) on the 25 MB result: 32x less gather traffic.  Linear in the tables, so it equals the
 * per-sample blend up to fp16 rounding order (blend-then-interpolate instead of interpolate-then-blend). */
int nsx_tables_preblend(const nsx_half* tables, int H, const nsx_grid_geom* g, const float* code_row /* [H] */,
                        const float* window /* [H] or NULL */, nsx_half* blended /* [total_entries][2] */, void* stream);

/* ---- plain tcnn-shaped HashGrid encoding (compatibility path) -------------------------------------------------
 * tcnn.Encoding(3, {"otype": "HashGrid", n_levels, n_features_per_level F in {2,4,8}, log2_hashmap_size,
 * base_resolution, per_level_scale, "Linear"}) as instantiated at hash_ensemble.py:42-50 and called at :102-104:
 * table fp16 [total_entries][F] (tcnn AoS), out fp16 [B][n_levels*F].  Backward: dtable fp32 (same shape,
 * ACCUMULATED, may be NULL), dx fp32 [B][3] ACCUMULATED into a caller-zeroed buffer (may be NULL).  Lets the
 * reference's own HashEnsemble module run on this library; the fused kernels above are the fast path. */
int nsx_hashgrid_fwd(const float* x, int64_t B, const nsx_half* table, int F, const nsx_grid_geom* g, nsx_half* out,
                     void* stream);
int nsx_hashgrid_bwd(const float* x, int64_t B, const nsx_half* table, int F, const nsx_grid_geom* g,
                     const nsx_half* dout, float* dtable, float* dx_zeroed, void* stream);

/* ---- fully fused MLPs (tcnn FullyFusedMLP equivalents) ---------------------------------------------------
 * Replaces tcnn.NetworkWithInputEncoding (Identity encoding) / tcnn.Network as built at
 * nersemble_nerfacto_field.py:142-153 (mlp_base: 32 -> 64 -> 16, no output activation) and :162-172
 * (mlp_head: 18 -> 64 -> 64 -> 3, Sigmoid), called at :285 and :377.
 *   weights   fp16, tcnn flat layout: W0 [64][32] | (Wh [64][64] if n_hidden_mats == 1) | Wo [16][64],
 *             row-major [out][in], no biases, input zero-padded to 32, output padded to 16.
 *   input of sample b = [ a[b][0..a_dim) * a_mul + a_add  (fp32 source),
 *                         b[b][b_off .. b_off + b_dim)      (fp16 source), zero padding ]
 *             -- the two segments let mlp_head read (dir+1)/2 (nersemble_nerfacto_field.py:313) and the 15
 *             geometry features straight from mlp_base's output without materialising the torch.cat (:371-375).
 *   out       [B][out_stride] fp16, columns [0, n_out) written; out_act 0 = None, 1 = Sigmoid.
 * nsx_mlp_bwd recomputes the forward (no saved activations) and produces
 *   dweights  fp32 [nsx_mlp_param_count], ACCUMULATED into (caller zeroes)
 *   da        fp32 [B][a_dim] (may be NULL), db fp16 written at b's layout [B][b_stride] cols b_off.. (may be NULL)
 *   db_f32    fp32 [B][b_dim] contiguous (may be NULL): the values of db (rounded to fp16 first) widened to fp32 -- what
 *             the HashEnsemble backward reads as `dout`, without a conversion launch in between
 * from dout fp16 [B][dout_stride] (AMP semantics: gradients of fp16 activations are fp16, loss-scaled by the caller). */
int nsx_mlp_param_count(int n_hidden_mats);
int nsx_mlp_fwd(const nsx_half* weights, int n_hidden_mats, int64_t B,
                const float* a, int64_t a_stride, int a_dim, float a_mul, float a_add,
                const nsx_half* b, int64_t b_stride, int b_off, int b_dim,
                int n_out, int out_act, nsx_half* out, int64_t out_stride, const int64_t* n_device, void* stream);
int nsx_mlp_bwd(const nsx_half* weights, int n_hidden_mats, int64_t B,
                const float* a, int64_t a_stride, int a_dim, float a_mul, float a_add,
                const nsx_half* b, int64_t b_stride, int b_off, int b_dim,
                int n_out, int out_act, const nsx_half* dout, int64_t dout_stride,
                float* dweights, float* da, nsx_half* db, float* db_f32, const int64_t* n_device, void* stream);
int nsx_f32_to_f16(const float* src, nsx_half* dst, int64_t n, void* stream);

/* ---- field glue (elementwise, fused) ------------------------------------------------------------------------
 * nsx_sample_positions: pos = o + (d * (t0 + t1)) / 2 (+ offsets) -- Frustums.get_positions (nerfstudio) and the
 *   sampler's sigma_fn positions; o/d are per sample [S][3], or [R][3] gathered through ray_indices.  With
 *   t_starts == NULL pos = o (+ offsets).  Optionally also the scene-box normalisation, in-box selector and
 *   masking of nersemble_nerfacto_field.py:257,268-269 (pos_normalised = normalised * selector, selector u8).
 * nsx_normalise_bwd: dL/dpos_world = dL/dpos_normalised * selector / extent.
 * nsx_density_fwd/bwd: density = trunc_exp(float(h0)) * selector (nersemble_nerfacto_field.py:286-293); backward
 *   g * selector * exp(clamp(h0, -15, 15)) written as fp16 into column 0 of a caller-zeroed [S][stride] buffer. */
/* (pos_world receives the position WITHOUT `offsets`; they enter pos_normalised only.) */
int nsx_sample_positions(const float* origins, const float* directions, const int64_t* ray_indices,
                         const float* t_starts, const float* t_ends, const float* offsets, int64_t S,
                         const float* aabb_host, float* pos_world, float* pos_normalised, uint8_t* selector,
                         const int64_t* n_device, void* stream);
/* dsts[a][i][:] = srcs[a][index[i]][:] for n_arrays <= NSX_MAX_GATHER device arrays in one launch (row_bytes[a] a
 * multiple of 4; srcs / row_bytes / dsts are HOST arrays of device pointers / sizes).  Replaces the index_select /
 * advanced-indexing launches after the visibility test: nerfacc's ray_indices[keep], t_starts[keep], t_ends[keep]
 * (inside OccGridEstimator.sampling, called at nersemble_volumetric_sampler.py:95-108), the per-field gathers that build
 * the packed RaySamples (nersemble_volumetric_sampler.py:117-134) and the compaction of the sigma-pass values that the
 * main pass reuses.  With n_device given, rows [*n_device, n) of every destination are written as zeros. */
/* Pinhole ray generation (csrc/raygen.hip): what the reference's datamanager obtains from nerfstudio's RayGenerator ->
 * Cameras.generate_rays for the pixel sampler's (camera, y, x) triples (datamanager/nersemble_datamanager.py:76-81;
 * perspective cameras without distortion, dataparser/nersemble_dataparser.py:237-244).
 *   camera_to_worlds [n_cameras][3][4] fp32 (OpenGL convention), fx / fy / cx / cy [n_cameras] fp32
 *   camera_indices [R] int64, ys / xs [R] fp32 pixel coordinates WITH the pixel-centre offset (+0.5) already added
 *   origins / directions [R][3] fp32 (unit directions), pixel_area [R] fp32 (may be NULL) */
int nsx_generate_rays(const float* camera_to_worlds, const float* fx, const float* fy, const float* cx, const float* cy,
                      int64_t n_cameras, const int64_t* camera_indices, const float* ys, const float* xs, int64_t R,
                      float* origins, float* directions, float* pixel_area, void* stream);
/* The reference derives a sample's time-code rows from its ray's normalised time -- round(times * (T - 1)),
 * nersemble_instant_ngp.py:249 (sigma_fn) and :300-318 (main pass) --; this path indexes the batch's compacted code tables
 * with a per-ray slot that the datamanager attaches (image of the cached batch -> row).  The two agree iff
 * row_timesteps[ray_slots[r]] == rint(ray_times[r] * (n_timesteps - 1)) for every ray; a ray that disagrees (or whose slot is
 * outside [0, n_code_rows)) ORs 1 into *flag (device int32, sticky: never cleared here).  One launch, no host read. */
int nsx_check_code_rows(const float* ray_times, const int32_t* ray_slots, int64_t R, const int32_t* row_timesteps,
                        int n_code_rows, int n_timesteps, int32_t* flag, void* stream);
#define NSX_MAX_GATHER 12
int nsx_gather_rows(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                    const int64_t* index, int64_t n, const int64_t* n_device, void* stream);
/* The same launch with arrays that are indexed by what the index points at: array a with use_via_host[a] != 0 takes
 * dsts[a][i] = srcs[a][via[index[i]]] (via: device int64) -- the rays' origins / directions of the kept samples
 * (index = kept sample ids, via = the marched samples' ray indices) beside the per-sample arrays, one launch instead of
 * two dependent ones. */
int nsx_gather_rows_via(int n_arrays, const void* const* srcs, const int64_t* row_bytes, void* const* dsts,
                        const int64_t* index, const int64_t* via, const uint8_t* use_via_host, int64_t n,
                        const int64_t* n_device, void* stream);
int nsx_normalise_bwd(const float* grad_pos_normalised, const uint8_t* selector, int64_t S, const float* aabb_host,
                      float* grad_pos_world, const int64_t* n_device, void* stream);
int nsx_density_fwd(const nsx_half* base_out, int64_t stride, const uint8_t* selector, int64_t S, float* density,
                    const int64_t* n_device, void* stream);
int nsx_density_bwd(const nsx_half* base_out, int64_t stride, const uint8_t* selector, const float* grad_density,
                    int64_t S, nsx_half* grad_base_out_zeroed, const int64_t* n_device, void* stream);

/* ---- fused SE(3) deformation field -------------------------------------------------------------------------
 * Replaces SE3DeformationField.compute_offsets (deformation_field.py:148-166): WindowedNeRFEncoding
 * (windowed_nerf_encoding.py:33-74) + torch.cat with the warp code + 6x128 MLP with a skip into layer 4 + the
 * two 128->3 heads (8 nn.Linear GEMMs under fp16 autocast, deformation_field.py:50-69,85-88) + se3_exp_map
 * (util/pytorch3d.py:107-191) + homogeneous warp with NaN fallback (:96-102), as one MFMA kernel.
 *   params     flat fp32 [nsx_deform_param_count()]:  W0[128][173] b0 | W1 b1 | W2 b2 | W3 b3 | W4[128][301] b4 |
 *              W5 b5 | Wr[3][128] br | Wv[3][128] bv   (the reference's mlp_stem.layers.{0..5}, mlp_r, mlp_v)
 *   packed     device scratch of nsx_deform_pack_bytes(): fp16 MFMA weight fragments + fp16-rounded biases,
 *              refreshed with nsx_deform_pack whenever the parameters change
 *   positions  [S][3] fp32 WORLD positions; aabb_host = 6 host floats; offsets are in NORMALISED space
 *              (warped - normalised, deformation_field.py:162)
 *   code / code_slot: warp code rows (fp32, stride code_stride); row of sample s = code_slot ? code_slot[s] : s
 *   window7_host: 7 per-frequency window weights (host), NULL = no window (windows_param None)
 * Backward (recomputes the forward; scratch of nsx_deform_scratch_bytes(S)): grad_params fp32 [param_count] and
 * grad_code_table fp32 [n_code_rows][128] (needs code_slot, n_code_rows <= 128) are ACCUMULATED into (caller
 * zeroes); grad_code_samples fp32 [S][128] (per-sample code gradient, written) -- either may be NULL.
 * With code_slot and grad_code_table and WITHOUT grad_code_samples (the training step: <= 24 time codes per batch) every
 * quantity that touches the code columns is formed through the slot -- dW0[:, code] = R0 code, dW4[:, code] = R4 code,
 * dL/dcode[r] = W0c^T R0[:, r] + W4c^T R4[:, r] with R_l[n][r] = sum over the samples of slot r of dZ_l[n] -- from fp32
 * per-slot sums (the per-sample code gradient is never formed nor rounded to fp16): three launches (chain, sample-contracted
 * weight gradients with per-chunk partial vectors, finish), 102 instead of 118 KB of scratch traffic per 32 samples. */
int     nsx_deform_param_count(void);
int64_t nsx_deform_pack_bytes(void);
int64_t nsx_deform_scratch_bytes(int64_t S);
int nsx_deform_pack(const float* params, void* packed, void* stream);
/* The same from the 16 nn.Linear tensors where they live (host array of 16 device pointers, fp32 contiguous, in the
 * order W0 b0 W1 b1 W2 b2 W3 b3 W4 b4 W5 b5 Wr br Wv bv): no concatenation of the parameters per optimizer step. */
int nsx_deform_pack_tensors(const void* const* tensors16_host, void* packed, void* stream);
int nsx_deform_fwd(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code,
                   int64_t code_stride, const int32_t* code_slot, const float* window7_host, float* offsets,
                   const int64_t* n_device, void* stream);
/* nsx_deform_fwd when every sample's code is row code_slot[s] of a table of n_code_rows rows (always, on this path: the
 * time codes of the batch, of the image, of the dataset): the 125 code columns k >= 48 of the two input layers are factored
 * through the row -- T_l[row][n] = sum_k W_l[n][k] code16[row][k - 45] once per launch (terms_scratch:
 * nsx_deform_terms_floats(n_code_rows) floats of device memory, required), added to the bias; the input GEMMs keep 3 of their
 * 11 K-steps.  Same products in another summation order (fp32): equal to nsx_deform_fwd up to the rounding of the
 * pre-activations; a sample's result depends on its row's VALUES only (not on the table the row sits in, nor on the table's
 * size: <= 64 rows keep the terms in LDS, larger tables read them from L2 -- the same numbers in the same order).
 * code_slot == NULL is allowed for a one-row table: every sample takes row 0 (an evaluation image's single timestep).
 * Round 5: the route of every table-indexed forward of the model (sampler sigma_fn, occupancy update, evaluation, the
 * fused pass); nsx_deform_fwd stays the per-sample-code operator of SE3DeformationField.compute_offsets(positions, codes). */
int64_t nsx_deform_terms_floats(int n_code_rows);
int nsx_deform_fwd_rows(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code_table,
                        int64_t code_stride, const int32_t* code_slot, int n_code_rows, const float* window7_host,
                        float* offsets, float* terms_scratch, const int64_t* n_device, void* stream);
int nsx_deform_bwd(const void* packed, const float* positions, int64_t S, const float* aabb_host, const float* code,
                   int64_t code_stride, const int32_t* code_slot, int n_code_rows, const float* window7_host,
                   const float* grad_offsets, void* scratch, float* grad_params, float* grad_code_table,
                   float* grad_code_samples, const int64_t* n_device, void* stream);

/* ---- fused no-grad density pass on a pre-blended grid (csrc/density_fused.hip) --------------------------------------
 * Replaces NeRSembleNeRFactoField.density_fn / get_density (nersemble_nerfacto_field.py:228-301) for bundles whose rays share
 * ONE timestep -- an evaluation image (evaluate_nersemble.py:141 -> get_outputs_for_camera_ray_bundle), whose sampler marches
 * 25.9 M samples through sigma_fn (nersemble_instant_ngp.py:235-266) -- with the H tables blended once per image into one
 * 2-feature grid (nsx_tables_preblend: the blend of hash_ensemble.py:155-156 is linear in the tables).  One launch instead
 * of nsx_sample_positions (normalise) + nsx_hashgrid_fwd (F = 2) + nsx_mlp_fwd (mlp_base) + nsx_density_fwd; the [S][32]
 * features, the normalised positions and the selector stay in registers.  Outputs are bit-identical to the four launches.
 *   positions_world [S][3] fp32, offsets [S][3] fp32 or NULL (deformation, added to the world position as :257-259 does)
 *   field_aabb_host: 6 floats (host); table fp16 [total_entries][2]; g: 16 levels
 *   base_weights: mlp_base's flat fp16 vector (nsx_mlp_fwd's layout), base_hidden_mats 0 or 1
 *   base_out fp16 [S][base_out_stride >= 16] or NULL (the whole 16-wide row: h0 + 15 geometry features), density fp32 [S] */
int nsx_density_fused_fwd(const float* positions_world, const float* offsets, int64_t S, const float* field_aabb_host,
                          const nsx_half* table, const nsx_grid_geom* g, const nsx_half* base_weights, int base_hidden_mats,
                          nsx_half* base_out, int64_t base_out_stride, float* density, const int64_t* n_device, void* stream);

/* ---- occupancy-grid ray marching (nerfacc 0.5.2 traverse_grids equivalent) -----------------------------
 * Replaces the native part of OccGridEstimator.sampling called at nersemble_volumetric_sampler.py:95-108:
 * ray/AABB slab test + DDA through ONE res^3 boolean grid level (grid_levels=1, train_nersemble.py:100) +
 * fixed-step lattice anchored at the per-ray near plane; a sample [t, t+step] is emitted iff its midpoint
 * lies in an occupied voxel.  aabb is a HOST pointer to 6 floats (min xyz, max xyz); binary is a device
 * uint8/bool [res][res][res]; near is a device [R] (near plane, already jittered when stratified).
 * Two passes like nerfacc: nsx_march_count -> nsx_pack_info (device scan; caller reads *total back) ->
 * nsx_march_fill.  Counts, ray indices, cell ids and t values are bit-exact against the oracle. */
int nsx_march_count(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                    const uint8_t* binary, int res, const float* near, float far_plane, float step,
                    int64_t* counts, void* stream);
int nsx_pack_info(const int64_t* counts, int64_t R, int64_t* packed_info /* [R][2] start,count */,
                  int64_t* total /* device scalar */, void* stream);
/* The read-back of the marched total (nerfacc's one host synchronisation per sampling call) as an asynchronous copy into
 * PINNED host memory on `stream`: a counting pass issued a step ahead on its own stream (OccGridEstimator.prefetch_march)
 * leaves the number on the host before the step that needs it starts. */
int nsx_copy_to_host_async(void* dst_pinned_host, const void* src_device, int64_t bytes, void* stream);
int nsx_march_fill(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                   const uint8_t* binary, int res, const float* near, float far_plane, float step,
                   const int64_t* packed_info, float* t_starts, float* t_ends, int64_t* ray_indices,
                   int32_t* cells /* may be NULL */, void* stream);
/* The counting pass that also KEEPS what it walks past (round 6): stash [R][stash_cap] fp32 receives the starts of each ray's
 * first stash_cap samples; *over (device int64, zeroed by the caller) is set when some ray has more.  A counting pass runs a
 * step ahead on its own stream (OccGridEstimator.prefetch_march): with the stash the step itself no longer walks the grid a
 * second time -- nsx_march_fill_from_stash copies the starts to their packed places (one wave per ray) and writes the same
 * t_ends = fl(t_starts + step) and ray indices as nsx_march_fill, bit for bit (tests/test_march_gpu.py).  Replaces the second
 * traversal of nersemble_volumetric_sampler.py:95-108 / nerfacc's traverse_grids on the step's critical path only. */
int nsx_march_count_stash(const float* rays_o, const float* rays_d, int64_t R, const float* aabb_host,
                          const uint8_t* binary, int res, const float* near, float far_plane, float step,
                          int64_t* counts, float* stash, int stash_cap, int64_t* over, void* stream);
int nsx_march_fill_from_stash(const float* stash, int stash_cap, int64_t R, float step, const int64_t* packed_info,
                              float* t_starts, float* t_ends, int64_t* ray_indices, void* stream);
/* counts[r] += #samples with ray index r (nerfacc.pack_info, nersemble_instant_ngp.py:325); caller zeroes counts. */
int nsx_ray_histogram(const int64_t* ray_indices, int64_t S, int64_t R, int64_t* counts_zeroed, const int64_t* n_device, void* stream);

/* ---- per-ray scans on packed samples ----------------------------------------------------------------------
 * nerfacc.render_weight_from_density (nersemble_instant_ngp.py:326-331) and render_visibility_from_density
 * (inside sampling): T_i = exp(-sum_{j<i} sigma_j dt_j), alpha_i = 1-exp(-sigma_i dt_i), w_i = T_i alpha_i,
 * visibility = T >= early_stop_eps && (alpha_thre <= 0 || alpha >= alpha_thre).  Any output may be NULL. */
int nsx_render_weights_fwd(const float* t_starts, const float* t_ends, const float* sigmas,
                           const int64_t* packed_info, int64_t R, float* weights, float* trans, float* alphas,
                           uint8_t* visibility, float early_stop_eps, float alpha_thre,
                           const float* alpha_thre_dev /* device scalar overriding alpha_thre; may be NULL */,
                           void* stream);
/* The visibility test alone, with the number of visible samples of every ray (visible_per_ray [R] int64, every entry
 * written): what OccGridEstimator.sampling does with render_visibility_from_density before it compacts -- the counts
 * are the kept samples' nerfacc.pack_info input. */
int nsx_render_visibility(const float* t_starts, const float* t_ends, const float* sigmas, const int64_t* packed_info,
                          int64_t R, uint8_t* visibility, int64_t* visible_per_ray, float early_stop_eps, float alpha_thre,
                          const float* alpha_thre_dev /* may be NULL */, void* stream);
int nsx_render_weights_bwd(const float* t_starts, const float* t_ends, const float* sigmas,
                           const int64_t* packed_info, int64_t R, const float* grad_weights, float* grad_sigmas,
                           void* stream);
/* nerfacc.accumulate_along_rays (renderers at nersemble_instant_ngp.py:334-343, nersemble_deformation_renderer.py:22-25):
 * out[r][c] = sum_i w_i * values[i][c]; values NULL => C = 1, out = sum_i w_i.  C in {1, 3}. */
int nsx_accumulate_fwd(const float* weights, const float* values, int C, const int64_t* packed_info, int64_t R,
                       float* out, void* stream);
int nsx_accumulate_bwd(const float* weights, const float* values, int C, const int64_t* ray_indices, int64_t S,
                       const float* grad_out, float* grad_weights /* may be NULL */,
                       float* grad_values /* may be NULL */, void* stream);
/* Fused compositing of nersemble_instant_ngp.py:325-343,359-362 (render_weight_from_density + RGBRenderer(white|black
 * background) + AccumulationRenderer + DepthRenderer("expected", clipped to [min, max] sample midpoint) + the
 * DeformationRenderer accumulation of `aux`) in one pass; nsx_composite_bwd returns dL/dsigma and dL/drgb from the
 * gradients of all four outputs (any gradient pointer may be NULL). */
int nsx_composite_fwd(const float* t_starts, const float* t_ends, const float* sigmas, const float* rgb,
                      const float* aux /* [S][3] or NULL */, const int64_t* packed_info, int64_t R, float background,
                      float* clip_workspace /* device float[2]: receives min/max sample midpoint */, float* weights,
                      float* rgb_ray, float* acc_ray, float* depth_ray, float* aux_ray, void* stream);
int nsx_composite_bwd(const float* t_starts, const float* t_ends, const float* sigmas, const float* rgb,
                      const int64_t* packed_info, int64_t R, float background, const float* clip_workspace,
                      const float* acc_ray, const float* depth_ray, const float* grad_weights, const float* grad_rgb_ray,
                      const float* grad_acc_ray, const float* grad_depth_ray, float* grad_sigmas, float* grad_rgb,
                      void* stream);
/* The same two passes with the per-sample colours (and, in the backward, their gradient) in fp16 -- what the fused
 * mlp_head writes and reads (nersemble_nerfacto_field.py:377 casts to fp32 with `.to(directions)`, autograd casts the
 * gradient back; both are value-preserving resp. one round-to-nearest, done here instead of in two conversion launches). */
int nsx_composite_fwd_h(const float* t_starts, const float* t_ends, const float* sigmas, const nsx_half* rgb,
                        const float* aux /* [S][3] or NULL */, const int64_t* packed_info, int64_t R, float background,
                        float* clip_workspace, float* weights, float* rgb_ray, float* acc_ray, float* depth_ray,
                        float* aux_ray, void* stream);
int nsx_composite_bwd_h(const float* t_starts, const float* t_ends, const float* sigmas, const nsx_half* rgb,
                        const int64_t* packed_info, int64_t R, float background, const float* clip_workspace,
                        const float* acc_ray, const float* depth_ray, const float* grad_weights, const float* grad_rgb_ray,
                        const float* grad_acc_ray, const float* grad_depth_ray, float* grad_sigmas, nsx_half* grad_rgb,
                        void* stream);
/* Fused per-sample losses of one step: distortion (models/base.py:224-249, rays < max_ray), empty and near losses
 * (models/base.py:136-202; Normal CDF with sigma = (eps/3)^2, accumulated weights per ray) in one segmented-scan pass.
 *   per_ray   [R][5] = { dist term, sum w^2 over "very near" samples, their count, sum (A - cdf)^2 over "near"
 *             samples, their count }; the caller sums over rays:  dist = S0 / n_rays, empty = S1 / max(S2,1),
 *             near = S3 / max(S4,1).
 *   bwd: sums = the 5 column sums (device), grads = dL/d{dist, empty, near} (device float[3]); writes dL/dw. */
int nsx_sample_losses_fwd(const float* weights, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                          int64_t R, const float* depth_targets /* [R] or NULL */, float eps, int64_t max_ray,
                          float* per_ray, void* stream);
int nsx_sample_losses_bwd(const float* weights, const float* t_starts, const float* t_ends, const int64_t* packed_info,
                          int64_t R, const float* depth_targets, float eps, int64_t max_ray, int64_t n_rays,
                          const float* sums, const float* grads, float* grad_weights, void* stream);
/* The per-ray loss terms of one training step, their reduction over rays, the summed loss and the training metrics
 * (models/base.py:90-133,204-215 get_masked_rgb_loss / get_alpha_loss / get_depth_loss; the reductions of the
 * distortion / empty / near losses, base.py:136-202,224-249; nersemble_instant_ngp.py:409-422 get_metrics_dict;
 * nersemble_trainer.py:184 reduce(add, loss_dict.values())) in ONE single-block kernel instead of ~110 element-wise
 * launches on [R] tensors.  All pointers are device pointers; out is float[NSX_LOSS_OUT]:
 *   out[NSX_LOSS_RGB]   = masked ? sum_{alpha>thr} mean_c (img-rgb)^2 / max(#,1) : MSE(img, rgb)
 *   out[NSX_LOSS_ALPHA] = lambda_alpha * sum_{alpha<1} |acc - alpha| / max(#,1)        (alpha = alpha_map / 255; 0 if NULL)
 *   out[NSX_LOSS_DEPTH] = lambda_depth * sum_{t>0} (t - depth)^2 / max(#,1)           (0 if depth_targets NULL)
 *   out[NSX_LOSS_DIST / EMPTY / NEAR] from per_ray_sample = nsx_sample_losses_fwd's [R][5] (0 if NULL):
 *       lambda_dist * S0 / n_eff (n_eff = last ray with samples + 1, as torch_efficient_distloss divides by
 *       ray_id.max()+1), lambda_empty * S1 / max(S2,1), lambda_near * S3 / max(S4,1)
 *   out[NSX_LOSS_TOTAL] = ((((rgb + alpha) + dist) + empty) + near) + depth
 *   out[NSX_LOSS_PSNR], out[NSX_LOSS_PSNR_MASKED] (alpha_map > 127), out[NSX_LOSS_NUM_SAMPLES] = sum packed_info[:,1]
 *   out[NSX_LOSS_SAMPLE_SUMS ..+4] = S0..S4 (the `sums` argument of nsx_sample_losses_bwd), ..+5..+8 = denominators.
 * bwd: grad_out = dL/d out (float[NSX_LOSS_OUT], device; entries of TOTAL and of the individual terms add up);
 * writes dL/d rgb [R][3], dL/d accumulation [R], dL/d depth [R] and sample_grads float[3] = the `grads` argument of
 * nsx_sample_losses_bwd (launch it afterwards on the same stream with n_rays and out + NSX_LOSS_SAMPLE_SUMS). */
#define NSX_LOSS_OUT 24
#define NSX_LOSS_RGB 0
#define NSX_LOSS_ALPHA 1
#define NSX_LOSS_DEPTH 2
#define NSX_LOSS_DIST 3
#define NSX_LOSS_EMPTY 4
#define NSX_LOSS_NEAR 5
#define NSX_LOSS_TOTAL 6
#define NSX_LOSS_PSNR 7
#define NSX_LOSS_PSNR_MASKED 8
#define NSX_LOSS_NUM_SAMPLES 9
#define NSX_LOSS_SAMPLE_SUMS 10
int nsx_ray_losses_fwd(const float* rgb, const float* accumulation, const float* depth, const float* image,
                       const uint8_t* alpha_map /* [R] or NULL */, const float* depth_targets /* [R] or NULL */,
                       const float* per_ray_sample /* [R][5] or NULL */, const int64_t* packed_info /* or NULL */,
                       int64_t R, int use_masked_rgb, float alpha_mask_threshold, float lambda_alpha,
                       float lambda_depth, float lambda_dist, float lambda_empty, float lambda_near, float* out,
                       void* stream);
int nsx_ray_losses_bwd(const float* rgb, const float* accumulation, const float* depth, const float* image,
                       const uint8_t* alpha_map, const float* depth_targets, int64_t R, int use_masked_rgb,
                       float alpha_mask_threshold, float lambda_alpha, float lambda_depth, float lambda_dist,
                       float lambda_empty, float lambda_near, int64_t n_rays, const float* out, const float* grad_out,
                       float* grad_rgb, float* grad_accumulation, float* grad_depth, float* sample_grads /* or NULL */,
                       void* stream);
/* torch_efficient_distloss.flatten_eff_distloss (models/base.py:245-247): per-ray loss terms
 * ray_loss[r] = (sum_i 1/3 interval_i w_i^2 + 2 w_i (m_i Wpre_i - WMpre_i)) / n_rays for rays r < max_ray (0 otherwise,
 * base.py:235) and grad_weights = grad_scale * dloss/dw.  ray_loss / grad_weights may be NULL. */
int nsx_distloss(const float* weights, const float* midpoints, const float* intervals, const int64_t* packed_info,
                 int64_t R, int64_t max_ray, int64_t n_rays, float grad_scale, float* ray_loss, float* grad_weights,
                 void* stream);

/* ---- optimizer step for the hash tables --------------------------------------------------------------------
 * Replaces, for the 403 M-parameter `fields` group, torch.optim.Adam(lr 5e-3, eps 1e-15)
 * (train_nersemble.py:243-246) + GradScaler unscale / inf check / conditional step (nersemble_trainer.py:185-186,
 * 199-203) + tcnn's per-call fp32 -> fp16 parameter cast, in ONE pass over the parameters.
 * torch.optim.Adam semantics (no amsgrad, no weight decay); `step` is the 1-based step count; inv_scale /
 * found_inf are DEVICE scalars (may be NULL): gradients are multiplied by *inv_scale and the whole update is
 * skipped when *found_inf != 0.  nsx_adam_hash_factored forms the gradient on the fly from the factored
 * gradient G (see nsx_hash_ensemble_bwd_factored) -- the dense table gradient is never materialised.
 * n_slots <= NSX_MAX_SLOTS: the planes are walked in slot order by fp32 FMAs (the bits every single-process run and the
 * golden vectors have).  NSX_MAX_SLOTS < n_slots <= NSX_MAX_ADAM_SLOTS with 17..32 grids (level-parallel runs only): the
 * product G x code runs on the matrix cores on a three-way bf16 split of G -- fp32 accuracy (<= 1e-6 of sum |G c|), other
 * rounding than the FMA chain. */
int nsx_check_finite(const float* x, int64_t n, float* found_inf /* set to 1 if any non-finite */, void* stream);
int nsx_adam_hash_factored(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                           const float* window, int H, const nsx_grid_geom* g, float* master, float* exp_avg,
                           float* exp_avg_sq, nsx_half* tables_f16, float lr, float beta1, float beta2, float eps,
                           int64_t step, const float* inv_scale, const float* found_inf, void* stream);
/* The same step, CONSUMING G: every 16-byte piece of G that holds a non-zero value is written back as zeros while it is
 * read (also when the step is skipped because *found_inf != 0), so the buffer can take the next backward's scatter
 * without a fill of its own.  G must be 16-byte aligned. */
int nsx_adam_hash_factored_consume(float* G, int n_slots, const float* code_table, int64_t code_stride,
                           const float* window, int H, const nsx_grid_geom* g, float* master, float* exp_avg,
                           float* exp_avg_sq, nsx_half* tables_f16, float lr, float beta1, float beta2, float eps,
                           int64_t step, const float* inv_scale, const float* found_inf, void* stream);
int nsx_adam_dense(const float* grad, int64_t n, float* master, float* exp_avg, float* exp_avg_sq,
                   nsx_half* params_f16 /* may be NULL */, float lr, float beta1, float beta2, float eps,
                   int64_t step, const float* inv_scale, const float* found_inf, void* stream);

/* ---- data-parallel optimizer step for the hash tables (no reference counterpart: train_nersemble.py:272-274 is
 * single-GPU).  Each rank expands its factored gradient to a dense fp16 gradient scaled by 1/world
 * (tcnn's own table gradients are fp16), RCCL reduce-scatters it, the rank that owns a 1/world shard of the master
 * weights / Adam moments checks the shard for inf/NaN, runs Adam on it and writes the shard of the fp16 working
 * tables, which RCCL all-gathers.  (engine/sharded_adam.py) */
int nsx_hash_grad_expand_f16(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                             const float* window, int H, const nsx_grid_geom* g, nsx_half* dtables_f16, float scale,
                             int accumulate, void* stream);
/* One BUCKET of the same dense gradient for a bucketed reduce-scatter: the ranks' shards are shard_elements long (rank r
 * owns table elements [r * shard, (r + 1) * shard)), cut into shard / bucket pieces; bucket k gathers piece k of every
 * rank -- bucket_f16 [world][bucket_elements], rank r's part = table elements r * shard + k * bucket ... (zeros beyond the
 * table's end) -- so that reduce_scatter(bucket k) hands rank r piece k of ITS shard, and the expansion of bucket k + 1
 * runs while bucket k is on the links.  bucket_elements: a multiple of 1024. */
int nsx_hash_grad_expand_f16_bucket(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                                    const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16, float scale,
                                    int accumulate, int64_t shard_elements, int64_t bucket_elements, int64_t bucket_index,
                                    int world_size, void* stream);
/* The exchange restricted to the grids the coarse-to-fine window has reached (round 4).  While ceil(window) <= width the
 * grids [width, H) have zero gradient and zero Adam moments (hash_ensemble.py:133-138: their window weight is 0), so a
 * data-parallel step needs to exchange only [entry][f][width] of every entry: 1 / 32 of the bytes while one grid is on
 * (steps 0 ... 40 000 of the reference's schedule), 1/16 ... 1/2 along the ramp.  State keeps the full layout.
 *   _bucket_width : bucket k of the packed gradient, [world][bucket_entries][2][width] fp16; bucket_elements /
 *                   shard_elements still count FULL-layout elements (entries * 2 * padded H).  *beyond_width (may be
 *                   NULL) is set to 1 if a conditioned code is non-zero at a grid >= width -- the caller's premise fails.
 *   _adam ..width : torch.optim.Adam on the grids [0, width) of n_entries entries: grad_packed [n_entries][2][width];
 *                   master / moments / params_f16 in the full layout (pointers to the shard's first entry); the new fp16
 *                   values also go to packed_out [n_entries][2][width] (a skipped step copies the current ones there).
 *   _unpack_width : params_f16 [n_entries][2][padded H] <- packed [n_entries][2][width] (after the all-gather).
 *   _bucket_width_consume (round 6): _bucket_width that CLEARS the pairs of G it finds non-zero while it reads them (padded
 *                   H >= 8, width <= 16): once every bucket of the step has been expanded G is all zeros again, and the next
 *                   backward needs no fill in front of its scatter (nsx_adam_hash_factored_consume's device for the
 *                   data-parallel exchange).  One code table per step only (a second expansion would read zeros). */
int nsx_hash_grad_expand_f16_bucket_width(const float* G, int n_slots, const float* code_table, int64_t code_stride,
                                          const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16,
                                          float scale, int accumulate, int64_t shard_elements, int64_t bucket_elements,
                                          int64_t bucket_index, int world_size, int width, float* beyond_width,
                                          void* stream);
int nsx_hash_grad_expand_f16_bucket_width_consume(float* G, int n_slots, const float* code_table, int64_t code_stride,
                                                  const float* window, int H, const nsx_grid_geom* g, nsx_half* bucket_f16,
                                                  float scale, int accumulate, int64_t shard_elements,
                                                  int64_t bucket_elements, int64_t bucket_index, int world_size, int width,
                                                  float* beyond_width, void* stream);
int nsx_adam_dense_f16grad_width(const nsx_half* grad_packed, int64_t n_entries, int width, int H_padded, float* master,
                                 float* exp_avg, float* exp_avg_sq, nsx_half* params_f16, nsx_half* packed_out, float lr,
                                 float beta1, float beta2, float eps, int64_t step, const float* inv_scale,
                                 const float* found_inf, void* stream);
int nsx_tables_unpack_width(const nsx_half* packed, int64_t n_entries, int width, int H_padded, nsx_half* tables_f16,
                            void* stream);
int nsx_check_finite_f16(const nsx_half* x, int64_t n, float* found_inf /* set to 1 if any inf/NaN */, void* stream);
int nsx_adam_dense_f16grad(const nsx_half* grad, int64_t n, float* master, float* exp_avg, float* exp_avg_sq,
                           nsx_half* params_f16 /* may be NULL */, float lr, float beta1, float beta2, float eps,
                           int64_t step, const float* inv_scale, const float* found_inf, void* stream);

/* The small parameter groups (mlp_base / mlp_head, the two time embeddings, the 16 deformation tensors) in ONE launch
 * each: GradScaler.unscale_ + inf check (nersemble_trainer.py:185-186: found_inf[group] = 1 if any element of a
 * group's gradients is non-finite; gradients multiplied by *inv_scale in place) and torch.optim.Adam's update per group
 * (train_nersemble.py:247-256; skipped for a group whose found_inf is set, and for tensors without a gradient).
 * Tensor references are HOST arrays of device pointers (fp32, contiguous); at most NSX_MAX_TENSORS / NSX_MAX_GROUPS. */
#define NSX_MAX_TENSORS 64
#define NSX_MAX_GROUPS 8
typedef struct nsx_tensor_ref {
    void*   param;
    void*   grad;         /* NULL: no gradient this step */
    void*   exp_avg;
    void*   exp_avg_sq;
    int64_t n;
    int32_t group;
    int32_t step;         /* nsx_multi_adam: 1-based step count of THIS tensor -- torch.optim.Adam keeps `step` per
                             parameter and starts it when the parameter first receives a gradient (time_embedding: at
                             step 40 000, when the window opens); 0: use the group's step */
} nsx_tensor_ref;
typedef struct nsx_adam_group {
    float   lr, beta1, beta2, eps;
    int64_t step;         /* 1-based step count of the group */
} nsx_adam_group;
int nsx_multi_unscale_check(const nsx_tensor_ref* tensors_host, int n_tensors, int n_groups, const float* inv_scale,
                            float* found_inf /* [n_groups] */, void* stream);
int nsx_multi_adam(const nsx_tensor_ref* tensors_host, int n_tensors, const nsx_adam_group* groups_host, int n_groups,
                   const float* found_inf /* [n_groups], may be NULL */, void* stream);
/* nsx_multi_adam for a data-parallel rank (the optimizer step of nersemble_trainer.py:185-203 when several processes share
 * it).  torch.optim.Adam leaves a parameter without a gradient alone (no moment decay, no
 * `step`): with several ranks every rank joins the gradient all-reduce with zeros for such a parameter, and WHETHER any rank
 * had a gradient is known on the device only (the counts travel in the same bucket).  present: fp32 device vector [n_present]
 * of those counts; present_index_host [n_tensors]: tensor i's element of it (-1: not subject to the rule).  A tensor whose count
 * is 0 is skipped exactly like one whose grad is NULL; the caller takes its host-side step count back when the counts have
 * reached the host (as it does for found_inf).  present NULL: nsx_multi_adam. */
int nsx_multi_adam_present(const nsx_tensor_ref* tensors_host, int n_tensors, const nsx_adam_group* groups_host, int n_groups,
                           const float* found_inf, const float* present, const int32_t* present_index_host, int n_present,
                           void* stream);

/* The tail of a data-parallel gradient bucket (the reference is single-process, train_nersemble.py:272-274; the gradients are
 * those of nersemble_trainer.py:183-184's backward): n_pieces <= NSX_MAX_BUCKET_PIECES small fp32 device arrays -- the
 * gradients that do not live in the step's gradient buffer, the presence counts, the flags -- copied one after the other to
 * flat_at in ONE launch (nsx_bucket_pack), and after the all-reduce (nsx_bucket_unpack) piece k = scale x its section of
 * flat_at, flat[0 .. n_scale) *= scale (the average over the ranks; scale = 1: untouched).  HOST arrays of device pointers /
 * element counts. */
#define NSX_MAX_BUCKET_PIECES 16
int nsx_bucket_pack(float* flat_at, const float* const* pieces_host, const int64_t* sizes_host, int n_pieces, void* stream);
int nsx_bucket_unpack(float* flat, int64_t n_scale, float scale, const float* flat_at, float* const* pieces_host,
                      const int64_t* sizes_host, int n_pieces, void* stream);

/* torch.amp.GradScaler.update() for the step's found_inf flags (nersemble_trainer.py:186-203: one scale update from all
 * optimizer groups) in ONE launch: total = sum(found_inf[0..n_groups)); the scale backs off when total != 0, grows after
 * growth_interval clean steps (torch._amp_update_scale_'s rules, the growth only if the result is finite); found_inf is
 * copied to found_copy (may be NULL; what the host reads one step late); clear_flags [n_groups] (may be NULL, may be
 * found_inf itself) is zeroed -- the NEXT step's flag buffer; *inv_scale (may be NULL) = 1 / new scale in double precision,
 * rounded -- the next step's; the new scale is also written to the n_mirrors device addresses of scale_mirrors_host (places
 * that cache it, e.g. the one-hot loss gradient a backward starts from).  A table optimizer that runs on another stream
 * still reads THIS step's flags and 1 / scale while the update executes: give the next step its own buffers (two slots in
 * turn).  enabled = 0: the scale stays (a disabled scaler), flags and derived values are still handled. */
#define NSX_MAX_SCALE_MIRRORS 8
int nsx_grad_scaler_update(const float* found_inf, int n_groups, float* scale, int32_t* growth_tracker, float* inv_scale,
                           float* found_copy, float* clear_flags, float* const* scale_mirrors_host, int n_mirrors,
                           float growth_factor, float backoff_factor, int growth_interval, int enabled, void* stream);

/* Debug/parity helper: the 8 level-local entry indices per (sample, level), uint32 [B][L][8] -- what tcnn's HashGrid
 * (instantiated at hash_ensemble.py:42-50; algorithm: SURVEY.md A.1) computes internally.  Integer outputs are held
 * bit-exact to the oracle. */
int nsx_hash_indices(const float* x, int64_t B, const nsx_grid_geom* g, uint32_t* idx, void* stream);

/* ---- occupancy-grid update (csrc/occ_grid.hip) ------------------------------------------------------------------
 * Replaces nerfacc 0.5.2 OccGridEstimator._update / _sample_uniform_and_occupied_cells as the reference reaches them
 * through the update_occupancy_grid callback, nersemble_instant_ngp.py:184-196 (every 16 steps: all cells while
 * step < 256, then N/4 uniform draws + the occupied cells (N/4 draws from them when there are more); jittered cell
 * positions; density at a random timestep x render step; occs = max(occs * 0.95, occ); binaries = occs > min(mean,
 * occ_thre)).  Random numbers come from Philox4x32-10 keyed by `seed` with counter (slot, purpose, step): identical
 * on every data-parallel rank and in the CPU oracle (oracle/occgrid.c), which states the conventions.
 * The density query between nsx_occ_sample_cells and nsx_occ_update is the caller's (deformation + HashEnsemble +
 * mlp_base kernels of this library on `positions` / `timesteps`). */

/* Bytes of the scratch buffer shared by the three calls below; the caller zeroes it ONCE after allocation (the calls
 * leave it zeroed again). */
int64_t nsx_occ_scratch_bytes(int64_t n_cells);

/* Ascending list of the occupied cells (`occupied`, capacity n_cells, int32) and their number (*n_occ, device). */
int nsx_occ_compact(const uint8_t* binaries, int64_t n_cells, int32_t* occupied, int32_t* n_occ, void* scratch,
                    void* stream);

/* The same stream compaction for a per-SAMPLE mask: ascending indices of the non-zero bytes of `mask` [n] as int64 in
 * kept[0 .. *n_kept) (the index type torch and nsx_gather_rows use; rows beyond *n_kept are left unwritten) and their
 * number as a device int64 -- what the visibility test of OccGridEstimator.sampling needs (nerfacc: masks.nonzero(), a
 * host synchronisation; here the count stays on the device, to be passed on as n_device).  Neither output needs clearing.
 * scratch: nsx_occ_scratch_bytes(n) bytes. */
int nsx_compact_mask(const uint8_t* mask, int64_t n, int64_t* kept, int64_t* n_kept, void* scratch, void* stream);
/* The same compaction when the mask's samples are packed per ray (packed_info_all [R][2]) and the packed_info of the
 * KEPT samples is already known (packed_info_kept [R][2]: nsx_render_visibility counted per ray, nsx_pack_info scanned):
 * one wave per ray writes kept[offset_r ...] -- the same ascending list as nsx_compact_mask's, in one launch, and the
 * kept samples' per-ray counts need no histogram afterwards.  n_kept (may be NULL) receives *total_kept. */
int nsx_compact_rays(const uint8_t* mask, const int64_t* packed_info_all, const int64_t* packed_info_kept, int64_t R,
                     int64_t* kept, const int64_t* total_kept, int64_t* n_kept, void* stream);

/* Slot s of the update -> cell id, jittered world position, random timestep and its normalised time t / (T - 1).
 * warmup != 0: slot s is cell s (M = res^3).  Otherwise M = N/4 + min(N/4, n_occ) with n_occ read back by the caller
 * from nsx_occ_compact (the one 4-byte read-back of the update; nerfacc's boolean indexing syncs three times). */
int nsx_occ_sample_cells(int res, const float* aabb_host, int warmup, const int32_t* occupied, int64_t n_occ,
                         uint64_t seed, int64_t step, int n_timesteps, int64_t M, int32_t* cell_ids, float* positions,
                         int32_t* timesteps, float* times, void* stream);

/* occs[c] = max(occs[c] * ema_decay, max over the slots of cell c of occ_values) for every cell named in cell_ids,
 * then binaries = occs > min(mean(occs[occs >= 0]), occ_thre) (mean accumulated in double).  threshold_out (device,
 * may be NULL) receives the threshold. */
int nsx_occ_update(float* occs, uint8_t* binaries, int64_t n_cells, const int32_t* cell_ids, const float* occ_values,
                   int64_t M, float ema_decay, float occ_thre, void* scratch, float* threshold_out, void* stream);

/* ---- training-step drivers (csrc/step.hip) ------------------------------------------------------------------
 * One optimisation step of the reference (engine/nersemble_trainer.py:169-206 -> nersemble_instant_ngp.py:280-422)
 * is ~45 launches of the entry points above.  Issued one by one from the host language they cost ~10 us each of
 * marshalling -- more than the kernels themselves once the occupancy grid has pruned the scene (the step is then paced
 * by the HOST).  The drivers below enqueue whole phases of the step from C: same kernels, same arguments, same order
 * as the per-kernel path (engine/fused_pass.py, nerfacc.py, the sampler), held to it bit for bit by
 * tests/test_native_step_gpu.py.  The caller owns every byte: it asks nsx_step_plan for the sizes and the offsets of
 * the sub-buffers, allocates the workspaces and passes their base pointers.
 *
 *   nsx_step_sample     the sampler behind NeRSembleVolumetricSampler.forward (nersemble_volumetric_sampler.py:95-134)
 *                       after the traversal's counting pass: nsx_march_fill, the sigma_fn density pass
 *                       (nersemble_instant_ngp.py:235-266: midpoints, per-ray code-row gather, deformation
 *                       [nsx_deform_fwd_rows: the batch's <= 64 code rows, terms in LDS], scene-box
 *                       normalisation, HashEnsemble, mlp_base, trunc_exp), the visibility test, its stream compaction, the
 *                       gathers of the kept samples (intervals, rays, code slots, and the sigma pass's forward values the
 *                       main pass reuses) and pack_info of the kept samples.  The kept count stays on the device.
 *                       `tables_ready_event` (optional): an event behind the writer of `tables` on ANOTHER stream (the
 *                       table optimizer's pass of the previous step); the wait is enqueued in front of the HashEnsemble
 *                       kernel only, so the traversal and the deformation field run beside that writer.
 *   nsx_step_main_fwd   NeRSembleNGPModel.get_outputs + get_loss_dict + get_metrics_dict on the kept samples (:300-422)
 *   nsx_step_main_bwd   its backward in three stages (0: losses ... mlp_base, 1: HashEnsemble, 2: normalisation +
 *                       deformation field) so that the caller can resolve the factored-gradient buffer and start
 *                       collectives in between.
 * Struct fields are grouped pointers / int64 / int32 / float so that no padding depends on the compiler; the Python
 * binding (nersemble_amd/_lib.py) builds its mirrors by parsing this header, tests check sizeof and a field echo. */
typedef struct nsx_step_plan {
    int64_t S;                  /* marched samples = capacity of every per-sample array */
    int64_t R;
    int64_t sample_bytes;       /* workspace of nsx_step_sample */
    int64_t m_ri;               /* marched: ray index int64 [S] */
    int64_t m_t0;
    int64_t m_t1;
    int64_t m_pos;              /* world midpoints [S][3] */
    int64_t m_slot;             /* code row of the sample's ray int32 [S] */
    int64_t m_off;              /* deformation offsets [S][3] */
    int64_t m_pn;               /* normalised positions [S][3] */
    int64_t m_sel;              /* in-box selector u8 [S] */
    int64_t m_feat;             /* hash features fp16 [S][32] */
    int64_t m_base;             /* mlp_base output fp16 [S][16] */
    int64_t m_dens;             /* density [S] */
    int64_t m_vis;              /* visibility mask u8 [S] */
    int64_t m_keep;             /* ascending indices of the visible samples int64 [S] */
    int64_t m_scratch;          /* scan scratch of nsx_compact_mask */
    int64_t m_terms;            /* nsx_deform_terms_floats(n_code_rows) floats of nsx_deform_fwd_rows */
    int64_t n_kept;             /* device int64: number of kept samples (the n_device of everything downstream) */
    int64_t k_ri;               /* kept (first *n_kept rows valid, capacity S): ray index int64 */
    int64_t k_t0;
    int64_t k_t1;
    int64_t k_org;              /* ray origin per sample [S][3] */
    int64_t k_dir;
    int64_t k_slot;             /* code slot int32 [S] */
    int64_t k_off;              /* sigma-pass forward values of the kept samples: offsets, hash features, mlp_base output */
    int64_t k_feat;
    int64_t k_base;
    int64_t k_counts;           /* samples per ray int64 [R] */
    int64_t k_packed;           /* packed_info int64 [R][2] */
    int64_t k_total;            /* device int64 (unused by the drivers) */
    int64_t fwd_bytes;          /* workspace of nsx_step_main_fwd (kept until the backward has run) */
    int64_t f_pos;
    int64_t f_pn;
    int64_t f_sel;
    int64_t f_dens;
    int64_t f_rgb16;            /* per-sample colours fp16 [S][3] */
    int64_t f_w;                /* rendering weights [S] */
    int64_t f_rgb;              /* per ray [R][3] */
    int64_t f_acc;
    int64_t f_depth;
    int64_t f_aux;              /* rendered deformation [R][3] */
    int64_t f_clip;
    int64_t f_per_ray;          /* [R][5] of nsx_sample_losses_fwd */
    int64_t grad_bytes;         /* parameter gradients (handed to the caller's autograd) */
    int64_t g_head;             /* mlp_head fp32 [nsx_mlp_param_count(head_hidden)] */
    int64_t g_base;
    int64_t g_deform;           /* fp32 [nsx_deform_param_count()] */
    int64_t g_code_deform;      /* fp32 [n_code_rows][128] */
    int64_t g_zero_end;         /* [g_head, g_zero_end) is cleared by stage 0 */
    int64_t g_code_hash;        /* fp32 [n_code_rows][H] (written whole by the code-sum kernel) */
    int64_t bwd_bytes;          /* scratch of the backward (free again when stage 2 has been enqueued) */
    int64_t b_grgb;
    int64_t b_gacc;
    int64_t b_gdep;
    int64_t b_g3;
    int64_t b_gw;
    int64_t b_ds;               /* [b_ds, b_zero_end) is cleared by stage 0: dL/dsigma, dL/drgb fp16, dL/dbase_out fp16 */
    int64_t b_dc16;
    int64_t b_dbase;
    int64_t b_zero_end;
    int64_t b_dout;             /* dL/dfeatures fp32 [S][32] */
    int64_t b_dx;
    int64_t b_goff;
    int64_t b_csum;             /* block partials of the code sums */
    int64_t b_plane;            /* int32 [S]: slot % hash_planes (nsx_step_main.hash_planes > 0) */
    int64_t b_deform;           /* nsx_deform_scratch_bytes(S) */
} nsx_step_plan;
int nsx_step_plan_make(int64_t S, int64_t R, int n_code_rows, int H, int base_hidden, int head_hidden,
                       nsx_step_plan* out);

typedef struct nsx_step_sample {
    const float* origins;            /* [R][3] */
    const float* directions;
    const float* near_planes;        /* [R] (jittered), the ones the counting pass ran with */
    const int64_t* packed_march;     /* [R][2] of the counting pass */
    const uint8_t* binaries;         /* occupancy grid [res]^3 */
    const int32_t* ray_slots;        /* [R]: row of the ray's samples in deform_codes / hash_codes = its code slot in the main
                                        pass's compacted code tables (the batch's images: the sigma_fn pass and the main pass
                                        read the SAME rows, round 5; rounds 3-4 indexed the dataset's [T] tables here) */
    const float* ray_times;          /* [R] or NULL: the rays' normalised times.  With row_timesteps the driver checks on the
                                        device that rint(times * (n_timesteps - 1)) == row_timesteps[ray_slots] for every ray
                                        (the reference derives its code rows from the times, nersemble_instant_ngp.py:249,
                                        300-318) and raises *rows_flag (sticky, never cleared here) when a ray disagrees */
    const int32_t* row_timesteps;    /* [n_code_rows] timestep of every code row, or NULL */
    int32_t* rows_flag;              /* device int32 or NULL */
    const void* deform_packed;
    const float* deform_codes;       /* [n_code_rows][128] */
    const nsx_half* tables;
    const nsx_grid_geom* geom;
    const float* hash_codes;         /* [n_code_rows][H] (conditioned time codes; ones [n_code_rows][1] in the compact
                                        first-grid phase) */
    const float* hash_window;        /* [H] or NULL */
    const nsx_half* base_w16;
    const float* alpha_thre_dev;     /* device scalar: min(alpha_thre, occs.mean()) */
    const float* window7_host;       /* 7 floats or NULL */
    uint8_t* ws;                     /* plan->sample_bytes */
    const nsx_step_plan* plan;
    const void* tables_ready_event;  /* hipEvent_t or NULL: `stream` waits for it right in front of the HashEnsemble */
    const float* march_stash;        /* [R][march_stash_cap] from nsx_march_count_stash (no ray beyond the cap), or NULL: pass 2
                                        of the traversal walks the grid again (nsx_march_fill) */
    int64_t R;
    int64_t S;
    int64_t deform_code_stride;
    int64_t hash_code_stride;
    int32_t grid_res;
    int32_t H;
    int32_t base_hidden;
    int32_t base_out_dim;
    int32_t base_act;
    int32_t n_code_rows;             /* rows of deform_codes / hash_codes (= the plan's) */
    int32_t n_timesteps;             /* T of the model (the check above) */
    int32_t phase;                   /* 0: the whole sampler; level-parallel runs split it at the HashEnsemble, whose features
                                        then arrive through the sample exchange (nsx_lp_*): 1 = everything in front of it
                                        (traversal ... normalised positions m_pn, code slots m_slot), 2 = everything behind
                                        it (mlp_base on m_feat ... the kept samples) */
    int32_t march_stash_cap;
    float far_plane;
    float step;
    float early_stop_eps;
    float reserved_f;
    float occ_aabb[6];
    float deform_aabb[6];
    float field_aabb[6];
} nsx_step_sample;
int nsx_step_sample_run(const nsx_step_sample* a, void* stream);

typedef struct nsx_step_main {
    uint8_t* ws_sample;              /* the workspace nsx_step_sample_run filled (kept arrays, n_kept) */
    uint8_t* ws_fwd;                 /* plan->fwd_bytes */
    float* out;                      /* float[NSX_LOSS_OUT] */
    const float* image;              /* [R][3] */
    const uint8_t* alpha_map;        /* [R] or NULL */
    const float* depth_targets;      /* [R] */
    const nsx_half* tables;
    const nsx_grid_geom* geom;
    const float* code_hash;          /* [n_code_rows][H] conditioned code rows of the batch */
    const float* hash_window;
    const void* deform_packed;
    const float* code_deform;        /* [n_code_rows][128] */
    const nsx_half* base_w16;
    const nsx_half* head_w16;
    const float* window7_host;
    const float* grad_out;           /* backward: dL/d out, float[NSX_LOSS_OUT] */
    uint8_t* ws_bwd;                 /* plan->bwd_bytes */
    uint8_t* grads;                  /* plan->grad_bytes */
    float* G;                        /* factored table gradient [n_code_rows][entries][2] or NULL */
    float* nonfinite;                /* device flag or NULL */
    const nsx_step_plan* plan;
    int64_t R;
    int64_t S;
    int64_t code_hash_stride;
    int64_t code_deform_stride;
    int64_t max_ray;
    int32_t H;                       /* grids the hash kernels run with (1 in the compact first-grid phase) */
    int32_t n_code_rows;
    int32_t base_hidden;
    int32_t base_out_dim;
    int32_t base_act;
    int32_t head_hidden;
    int32_t head_act;
    int32_t geo_dim;
    int32_t use_masked;
    int32_t need_code_grad;          /* the code gradient (window open); 0: nsx_hash_ensemble_bwd_factored without it */
    int32_t scatter_separately;      /* H == 1: the scatter as its own kernel (nsx_hash_ensemble_bwd_scatter) */
    int32_t hash_planes;             /* 0: G has one plane per code row.  P > 0 (only where every code row is the same, the
                                        compact first-grid phase): G is [P][entries][2], a sample adds to plane slot % P */
    float background;
    float thr;
    float l_alpha;
    float l_depth;
    float l_dist;
    float l_empty;
    float l_near;
    float eps;
    float field_aabb[6];
    float deform_aabb[6];
} nsx_step_main;
int nsx_step_main_fwd(const nsx_step_main* a, void* stream);
int nsx_step_main_bwd(const nsx_step_main* a, int stage, void* stream);
/* Optional HIP-event timing of the kernel calls the drivers make (the per-call events a host-language binding records
 * around its own calls do not see them): nsx_step_profile(1, tag) switches it on for the calls that follow (`tag` is
 * stored with every record, e.g. the index of the step), (0, .) off.  After the stream has drained: _count records, _get
 * reads one (name of the entry point, milliseconds between the two events, rows the call was sized for, info4 = {H or
 * n_hidden_mats, n_code_rows, 1 if the call ran under the kept-sample count, tag}); _reset recycles the events.
 * Process-wide host-side state behind one mutex (records of concurrent callers interleave; every call is safe);
 * off by default. */
int nsx_step_profile(int enable, int tag);
int nsx_step_profile_count(void);
int nsx_step_profile_get(int i, char* name_out, int name_capacity, float* ms, int64_t* rows, int32_t* info4);
int nsx_step_profile_reset(void);
/* sizeof / field echo for the binding's layout checks: nsx_step_echo writes the fields of the struct `kind`
 * (0 sample, 1 main) as doubles in declaration order (pointers as addresses, arrays element by element). */
int64_t nsx_step_sizeof(int kind /* 0 sample, 1 main, 2 plan */);
int nsx_step_echo(int kind, const void* s, double* out, int capacity);

/* ---- level-parallel exchange (csrc/level_parallel.hip; host side: nersemble_amd/engine/level_parallel.py) ---------------
 * The reference trains on ONE GPU (scripts/train/train_nersemble.py:272-274 hard-codes world_size = 1); this is the
 * package's data-parallel contract for the HashEnsemble once the coarse-to-fine window is open (train_nersemble.py:77-78):
 * rank r of W owns levels [r L / W, (r + 1) L / W) of all H grids -- a contiguous entry range of the [entry][f][h] tables --
 * and evaluates / differentiates them for EVERY rank's samples.  The columns out[:, 2 l + f] of HashEnsemble.forward
 * (hash_ensemble.py:93-158) depend on level l's entries only, so samples travel and parameters do not.  The binding issues
 * four collectives per step on byte buffers whose layout nsx_lp_layout_make fixes (or lets the library issue them:
 * nsx_lp_forward / nsx_lp_backward below); everything in between is enqueued by the entry points below (the per-source-rank
 * work is the UNCHANGED nsx_hash_ensemble_fwd / _bwd_codesum kernel body on a sub-geometry).
 *
 *   1. nsx_lp_fwd_pack    this rank's [count | positions | code slots | conditioned code rows]   -> ALL-GATHER (fwd_bytes)
 *   2. nsx_lp_fwd_run     ONE launch over all W source ranks (grid.y = source rank; NSX_OPT_LP_ONE_LAUNCH = 0: W launches):
 *                         the owned levels for the samples of source rank j -> block j of `send`
 *                                                                                               -> ALL-TO-ALL (feat_bytes)
 *   3. nsx_lp_fwd_unpack  column blocks -> features [S][2 L] fp16
 *   4. nsx_lp_bwd_pack    [count | dL/dfeatures column block fp16 | positions | slots] for every owner
 *                                                                                               -> ALL-TO-ALL (bwd_bytes)
 *   5. nsx_lp_bwd_run     one launch (as 2.): factored table gradient of the owned entries into the planes of source rank j
 *                         (plane = sum of the code rows of ranks < j, + code row), partial dL/dx and partial code-row
 *                         gradient into block j of `ret`                                        -> ALL-TO-ALL (ret_bytes)
 *   6. nsx_lp_bwd_unpack  partials summed over the owners (fixed order) -> dL/dx [S][3], dL/dcode [rows][H]
 *
 * Capacities (S_cap = max over the ranks of the marched samples, R_cap = max code rows) are host-known; the number of VALID
 * rows of every rank travels inside the payloads and is read by the kernels when they run (n_device semantics). */
typedef struct nsx_lp_layout {
    int64_t S_cap;
    int64_t fwd_bytes;          /* the all-gather payload of ONE rank */
    int64_t f_count;            /* int64: valid rows */
    int64_t f_pn;               /* float [S_cap][3] normalised positions */
    int64_t f_slot;             /* int32 [S_cap] code row of the sample within ITS rank's code rows */
    int64_t f_codes;            /* float [R_cap][H] conditioned code rows */
    int64_t feat_bytes;         /* one (owner -> source) block of the feature all-to-all: fp16 [S_cap][n2] */
    int64_t bwd_bytes;          /* one (source -> owner) block of the gradient all-to-all */
    int64_t b_count;
    int64_t b_dz;               /* fp16 [S_cap][n2] */
    int64_t b_pn;
    int64_t b_slot;
    int64_t ret_bytes;          /* one (owner -> source) block of the return all-to-all */
    int64_t r_dx;               /* float [S_cap][3] */
    int64_t r_dcode;            /* float [R_cap][H] */
    int32_t W;
    int32_t R_cap;
    int32_t H;
    int32_t n2;                 /* feature columns per rank = 2 x owned levels */
    int32_t level_of[32];   /* (NSX_MAX_LEVELS) global level of (owner j, its i-th level) at [j * n2 / 2 + i]: where block j's columns
                                           sit in the [S][2 L] feature row.  nsx_lp_layout_make fills the contiguous assignment
                                           (owner j holds levels [j n2/2, (j+1) n2/2)); a binding that balances the owners'
                                           work -- the finest levels cost twice the coarsest -- overwrites it with its
                                           permutation of 0 .. W n2/2 - 1 */
} nsx_lp_layout;
int nsx_lp_layout_make(int W, int64_t S_cap, int R_cap, int H, int n2, nsx_lp_layout* out);
int64_t nsx_lp_sizeof(void);
int nsx_lp_fwd_pack(const nsx_lp_layout* lay, const float* pn, const int32_t* slot, int64_t S, const int64_t* n_device,
                    const float* codes, int64_t code_stride, int rows, uint8_t* payload, void* stream);
/* gathered: [W][fwd_bytes]; sizes_host / rows_host: HOST arrays [W] (row capacity and code rows of every rank);
 * tables / sub_geom: this rank's entry range and its geometry (entry offsets re-based to 0); send: [W][feat_bytes];
 * codes_packed (optional): float [sum rows][H], the job's code rows in gradient-plane order (what the optimizer pass reads) */
int nsx_lp_fwd_run(const nsx_lp_layout* lay, const uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host,
                   const nsx_half* tables, const nsx_grid_geom* sub_geom, const float* window, uint8_t* send,
                   float* codes_packed, void* stream);
int nsx_lp_fwd_unpack(const nsx_lp_layout* lay, const uint8_t* recv, int64_t S, const int64_t* n_device, nsx_half* feats,
                      void* stream);
/* dout: fp32 [S][W * n2] (fp16-representable values: nsx_mlp_bwd emits them); send: [W][bwd_bytes] */
int nsx_lp_bwd_pack(const nsx_lp_layout* lay, const float* dout, const float* pn, const int32_t* slot, int64_t S,
                    const int64_t* n_device, uint8_t* send, void* stream);
/* recv: [W][bwd_bytes]; gathered: the forward's all-gather (its code rows); G: [sum rows][owned entries][2] fp32 or NULL;
 * dz_scratch: float [W][S_cap][n2]; csum_scratch: nsx_hash_codesum_scratch_floats(R_cap, H) floats; ret: [W][ret_bytes] */
int nsx_lp_bwd_run(const nsx_lp_layout* lay, const uint8_t* recv, const uint8_t* gathered, const int64_t* sizes_host,
                   const int32_t* rows_host, const nsx_half* tables, const nsx_grid_geom* sub_geom, const float* window,
                   float* G, float* dz_scratch, float* csum_scratch, uint8_t* ret, float* nonfinite, void* stream);
int nsx_lp_bwd_unpack(const nsx_lp_layout* lay, const uint8_t* ret_recv, int64_t S, const int64_t* n_device, int rows,
                      float* dx, float* dcode, void* stream);

/* ---- collectives issued by the library (csrc/comm.hip) --------------------------------------------------------------------
 * (No reference counterpart: scripts/train/train_nersemble.py:272-274 hard-codes one process; the training step whose
 * HashEnsemble calls these surround is nersemble_trainer.py:169-206.)
 * The level-parallel exchange above with its collectives enqueued from C: one call per direction puts
 * pack -> collective -> kernels -> collective -> unpack on the caller's stream (the binding otherwise pays five
 * torch.distributed calls and the Python between them per step -- on a rank whose step takes ~2 ms that was what it waited
 * for).  An nsx_comm is an RCCL communicator owned by the library: rank 0 of the job draws nsx_comm_unique_id, the binding
 * carries the NSX_COMM_ID_BYTES bytes to every process (any host-side channel), every process calls nsx_comm_create with the
 * HIP device current that it computes on (collective: returns when all ranks have joined).  RCCL is resolved at run time from
 * the librccl.so.1 the process already holds (PyTorch's); nsx_comm_library names another file, before first use.
 * nsx_lp_forward / nsx_lp_backward take the arguments of the six nsx_lp_* calls they chain (same buffers, same layouts) plus
 * the receive buffers; emulate_rank >= 0: the communicator has ONE rank and this process plays rank emulate_rank of lay->W
 * whose other ranks are replicas of it (what a rank of a W-GPU job computes and issues, on one GPU); -1: a real job. */
#define NSX_COMM_ID_BYTES 128
typedef struct nsx_comm nsx_comm;
int nsx_comm_library(const char* path);
int nsx_comm_unique_id(uint8_t* id128);
int nsx_comm_create(const uint8_t* id128, int world_size, int rank, nsx_comm** out);
int nsx_comm_destroy(nsx_comm* comm);
int nsx_comm_world_size(const nsx_comm* comm);
int nsx_comm_rank(const nsx_comm* comm);
/* in-place sum over the ranks of `count` floats (the bucket of the small gradients, engine/parallel.py) */
int nsx_comm_all_reduce_sum(nsx_comm* comm, float* data, int64_t count, void* stream);
/* payload: [fwd_bytes]; gathered: [W][fwd_bytes] (kept until the backward); send, recv: [W][feat_bytes]; feats: fp16 [S][2 L] */
int nsx_lp_forward(const nsx_lp_layout* lay, nsx_comm* comm, int emulate_rank, const float* pn, const int32_t* slot, int64_t S,
                   const int64_t* n_device, const float* codes, int64_t code_stride, int rows, uint8_t* payload,
                   uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host, const nsx_half* tables,
                   const nsx_grid_geom* sub_geom, const float* window, uint8_t* send, float* codes_packed, uint8_t* recv,
                   nsx_half* feats, void* stream);
/* send, recv: [W][bwd_bytes]; ret, ret_recv: [W][ret_bytes]; the rest as nsx_lp_bwd_pack / _run / _unpack */
int nsx_lp_backward(const nsx_lp_layout* lay, nsx_comm* comm, int emulate_rank, const float* dout, const float* pn,
                    const int32_t* slot, int64_t S, const int64_t* n_device, uint8_t* send, uint8_t* recv,
                    const uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host, const nsx_half* tables,
                    const nsx_grid_geom* sub_geom, const float* window, float* G, float* dz_scratch, float* csum_scratch,
                    uint8_t* ret, uint8_t* ret_recv, float* nonfinite, int rows, float* dx, float* dcode, void* stream);

#ifdef __cplusplus
}
#endif
#endif

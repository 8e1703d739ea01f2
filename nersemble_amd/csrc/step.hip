// step.hip -- training-step drivers (include/nsx.h, "training-step drivers"): whole phases of one optimisation step
// enqueued from C.  Every launch here is a call of an entry point of this library with the arguments the per-kernel path
// (engine/fused_pass.py, nerfacc.py, model_components/nersemble_volumetric_sampler.py) passes; nothing is computed
// differently.  Reference: engine/nersemble_trainer.py:169-206 (one iteration), nersemble_instant_ngp.py:235-266
// (field_density_fn), :280-364 (get_outputs), :366-422 (losses / metrics), nersemble_volumetric_sampler.py:95-134.
#include "nsx_common.h"
#include <cstring>
#include <mutex>
#include <vector>

namespace nsx {

static inline int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }

struct Carver {
    int64_t off = 0;
    int64_t take(int64_t bytes) {
        const int64_t at = off;
        off += up256(bytes < 0 ? 0 : bytes);
        return at;
    }
};

template <typename T>
static inline T* at(uint8_t* base, int64_t off) { return reinterpret_cast<T*>(base + off); }
template <typename T>
static inline const T* cat(const uint8_t* base, int64_t off) { return reinterpret_cast<const T*>(base + off); }

// plane[i] = slot[i] % P for the kept samples (nsx_step_main.hash_planes)
__global__ __launch_bounds__(256) void plane_of_slot_kernel(const int32_t* __restrict__ slot, int64_t S, int P,
                                                            int32_t* __restrict__ plane, const int64_t* __restrict__ n_dev) {
    if (n_dev) {
        const int64_t n = *n_dev;
        if (n < S) S = n < 0 ? 0 : n;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += (int64_t)gridDim.x * 256)
        plane[i] = (int32_t)((uint32_t)slot[i] % (uint32_t)P);
}

#define NSX_TRY(call)                    \
    do {                                 \
        const int rc__ = (call);         \
        if (rc__ != NSX_OK) return rc__; \
    } while (0)

// Optional HIP-event timing of the drivers' kernel calls (nsx_step_profile*, include/nsx.h): what bench.py's per-call
// events were when every kernel was launched from Python.  Events are recorded on the stream the call launches on.
struct ProfRec {
    const char* name;
    hipEvent_t a, b;
    int64_t rows;
    int32_t H, n_slots, counted, tag;
};
// process-wide, behind ONE mutex (include/nsx.h): the records of concurrent callers interleave, no call races
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static int g_prof_tag = -1;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {          // (g_prof_mu held)
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    hipStream_t st;
    bool on;
    ProfRec r;
    ProfScope(const char* name, void* stream, int64_t rows, int H, int n_slots, int counted)
        : st((hipStream_t)stream), on(false) {
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            on = g_prof_on;
            if (!on) return;
            r = ProfRec{name, prof_event(), prof_event(), rows, H, n_slots, counted, g_prof_tag};
        }
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};

// a kernel call of a driver: timed when profiling is on
#define NSX_CALL(name, rows, H, n_slots, counted, call)                \
    do {                                                               \
        ::nsx::ProfScope ps__(name, stream, rows, H, n_slots, counted); \
        const int rc__ = (call);                                       \
        if (rc__ != NSX_OK) return rc__;                               \
    } while (0)

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_step_plan_make(int64_t S, int64_t R, int n_code_rows, int H, int base_hidden, int head_hidden,
                       nsx_step_plan* p) {
    NSX_REQUIRE(p != nullptr, "nsx_step_plan_make: out is NULL");
    NSX_REQUIRE(S >= 1 && R >= 1, "nsx_step_plan_make: S=%lld R=%lld must be >= 1", (long long)S, (long long)R);
    NSX_REQUIRE(n_code_rows >= 1 && n_code_rows <= NSX_MAX_SLOTS, "nsx_step_plan_make: n_code_rows=%d not in [1,%d]",
                n_code_rows, NSX_MAX_SLOTS);
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_step_plan_make: H=%d not in [1,32]", H);
    memset(p, 0, sizeof(*p));
    p->S = S;
    p->R = R;
    {
        Carver c;
        p->m_ri = c.take(S * 8);
        p->m_t0 = c.take(S * 4);
        p->m_t1 = c.take(S * 4);
        p->m_pos = c.take(S * 12);
        p->m_slot = c.take(S * 4);
        p->m_off = c.take(S * 12);
        p->m_pn = c.take(S * 12);
        p->m_sel = c.take(S);
        p->m_feat = c.take(S * 64);
        p->m_base = c.take(S * 32);
        p->m_dens = c.take(S * 4);
        p->m_vis = c.take(S);
        p->m_keep = c.take(S * 8);
        p->m_scratch = c.take(nsx_occ_scratch_bytes(S));
        p->m_terms = c.take(nsx_deform_terms_floats(n_code_rows) * 4);
        p->n_kept = c.take(8);
        p->k_ri = c.take(S * 8);
        p->k_t0 = c.take(S * 4);
        p->k_t1 = c.take(S * 4);
        p->k_org = c.take(S * 12);
        p->k_dir = c.take(S * 12);
        p->k_slot = c.take(S * 4);
        p->k_off = c.take(S * 12);
        p->k_feat = c.take(S * 64);
        p->k_base = c.take(S * 32);
        p->k_counts = c.take(R * 8);
        p->k_packed = c.take(R * 16);
        p->k_total = c.take(8);
        p->sample_bytes = c.off;
    }
    {
        Carver c;
        p->f_pos = c.take(S * 12);
        p->f_pn = c.take(S * 12);
        p->f_sel = c.take(S);
        p->f_dens = c.take(S * 4);
        p->f_rgb16 = c.take(S * 6);
        p->f_w = c.take(S * 4);
        p->f_rgb = c.take(R * 12);
        p->f_acc = c.take(R * 4);
        p->f_depth = c.take(R * 4);
        p->f_aux = c.take(R * 12);
        p->f_clip = c.take(8);
        p->f_per_ray = c.take(R * 20);
        p->fwd_bytes = c.off;
    }
    {
        Carver c;
        p->g_head = c.take((int64_t)nsx_mlp_param_count(head_hidden) * 4);
        p->g_base = c.take((int64_t)nsx_mlp_param_count(base_hidden) * 4);
        p->g_deform = c.take((int64_t)nsx_deform_param_count() * 4);
        p->g_code_deform = c.take((int64_t)n_code_rows * 128 * 4);
        p->g_zero_end = c.off;
        p->g_code_hash = c.take((int64_t)n_code_rows * H * 4);
        p->grad_bytes = c.off;
    }
    {
        Carver c;
        p->b_grgb = c.take(R * 12);
        p->b_gacc = c.take(R * 4);
        p->b_gdep = c.take(R * 4);
        p->b_g3 = c.take(12);
        p->b_gw = c.take(S * 4);
        p->b_ds = c.take(S * 4);
        p->b_dc16 = c.take(S * 6);
        p->b_dbase = c.take(S * 32);
        p->b_zero_end = c.off;
        p->b_dout = c.take(S * 128);
        p->b_dx = c.take(S * 12);
        p->b_goff = c.take(S * 12);
        p->b_csum = c.take(nsx_hash_codesum_scratch_floats(n_code_rows, H) * 4);
        p->b_plane = c.take(S * 4);
        p->b_deform = c.take(nsx_deform_scratch_bytes(S));
        p->bwd_bytes = c.off;
    }
    return NSX_OK;
}

int nsx_step_sample_run(const nsx_step_sample* a, void* stream) {
    NSX_REQUIRE(a && a->plan && a->ws, "nsx_step_sample_run: NULL argument");
    const nsx_step_plan& p = *a->plan;
    const int64_t S = a->S, R = a->R;
    NSX_REQUIRE(S >= 1 && S == p.S && R == p.R, "nsx_step_sample_run: S=%lld R=%lld do not match the plan (%lld, %lld)",
                (long long)S, (long long)R, (long long)p.S, (long long)p.R);
    NSX_REQUIRE(a->origins && a->directions && a->near_planes && a->packed_march && a->binaries &&
                a->ray_slots && a->deform_packed && a->deform_codes && a->tables && a->geom && a->hash_codes &&
                a->base_w16 && a->alpha_thre_dev, "nsx_step_sample_run: NULL argument");
    NSX_REQUIRE(a->base_out_dim == 16 && a->geom->n_levels * 2 == 32, "nsx_step_sample_run: the drivers are built for 16 "
                "levels x 2 features and a 16-wide mlp_base output (got %d levels, %d)", a->geom->n_levels, a->base_out_dim);
    NSX_REQUIRE(a->n_code_rows >= 1 && a->n_code_rows <= NSX_MAX_SLOTS &&
                p.m_terms + nsx_deform_terms_floats(a->n_code_rows) * 4 <= p.sample_bytes,
                "nsx_step_sample_run: n_code_rows=%d does not match the plan", a->n_code_rows);
    uint8_t* w = a->ws;
    int64_t* m_ri = at<int64_t>(w, p.m_ri);
    float* m_t0 = at<float>(w, p.m_t0);
    float* m_t1 = at<float>(w, p.m_t1);
    float* m_pos = at<float>(w, p.m_pos);
    int32_t* m_slot = at<int32_t>(w, p.m_slot);
    float* m_off = at<float>(w, p.m_off);
    float* m_pn = at<float>(w, p.m_pn);
    uint8_t* m_sel = at<uint8_t>(w, p.m_sel);
    nsx_half* m_feat = at<nsx_half>(w, p.m_feat);
    nsx_half* m_base = at<nsx_half>(w, p.m_base);
    float* m_dens = at<float>(w, p.m_dens);
    uint8_t* m_vis = at<uint8_t>(w, p.m_vis);
    int64_t* m_keep = at<int64_t>(w, p.m_keep);
    int64_t* n_kept = at<int64_t>(w, p.n_kept);
    NSX_REQUIRE(a->phase >= 0 && a->phase <= 2, "nsx_step_sample_run: phase %d not in [0,2]", a->phase);
    const bool front = a->phase != 2, back = a->phase != 1;      // level-parallel runs split at the HashEnsemble
    if (front) {
    // -- pass 2 of the traversal (OccGridEstimator.sampling -> traverse)
    if (a->march_stash) {
        // (the counting pass kept the samples' starts: nothing walks the grid on the step's critical path)
        NSX_CALL("nsx_march_fill_from_stash", R, 0, 0, 0,
                 nsx_march_fill_from_stash(a->march_stash, a->march_stash_cap, R, a->step, a->packed_march, m_t0, m_t1, m_ri,
                                           stream));
    } else {
        NSX_CALL("nsx_march_fill", R, 0, 0, 0,
                 nsx_march_fill(a->origins, a->directions, R, a->occ_aabb, a->binaries, a->grid_res, a->near_planes,
                                a->far_plane, a->step, a->packed_march, m_t0, m_t1, m_ri, nullptr, stream));
    }
    // -- sigma_fn: density at the marched midpoints (get_sigma_fn -> field_density_fn)
    NSX_TRY(nsx_sample_positions(a->origins, a->directions, m_ri, m_t0, m_t1, nullptr, S, nullptr, m_pos, nullptr, nullptr,
                                 nullptr, stream));
    if (a->ray_times && a->row_timesteps && a->rows_flag)
        NSX_TRY(nsx_check_code_rows(a->ray_times, a->ray_slots, R, a->row_timesteps, a->n_code_rows, a->n_timesteps,
                                    a->rows_flag, stream));
    {
        const void* srcs[1] = {a->ray_slots};
        void* dsts[1] = {m_slot};
        const int64_t rb[1] = {4};
        NSX_TRY(nsx_gather_rows(1, srcs, rb, dsts, m_ri, S, nullptr, stream));
    }
    NSX_CALL("nsx_deform_fwd_rows", S, 0, a->n_code_rows, 0,
             nsx_deform_fwd_rows(a->deform_packed, m_pos, S, a->deform_aabb, a->deform_codes, a->deform_code_stride, m_slot,
                                 a->n_code_rows, a->window7_host, m_off, at<float>(w, p.m_terms), nullptr, stream));
    NSX_TRY(nsx_sample_positions(m_pos, nullptr, nullptr, nullptr, nullptr, m_off, S, a->field_aabb, nullptr, m_pn, m_sel,
                                 nullptr, stream));
    }
    if (a->phase == 0) {
    if (a->tables_ready_event &&
        hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)const_cast<void*>(a->tables_ready_event), 0) != hipSuccess)
        return hip_fail(hipGetLastError(), "nsx_step_sample_run: waiting for the tables");
    NSX_CALL("nsx_hash_ensemble_fwd", S, a->H, 0, 0,
             nsx_hash_ensemble_fwd(m_pn, S, a->tables, a->H, a->geom, a->hash_codes, a->hash_code_stride, m_slot,
                                   a->hash_window, m_feat, nullptr, stream));
    }
    if (!back) return NSX_OK;
    NSX_CALL("nsx_mlp_fwd", S, a->base_hidden, 0, 0,
             nsx_mlp_fwd(a->base_w16, a->base_hidden, S, nullptr, 0, 0, 1.0f, 0.0f, m_feat, 32, 0, 32, a->base_out_dim,
                         a->base_act, m_base, a->base_out_dim, nullptr, stream));
    NSX_TRY(nsx_density_fwd(m_base, a->base_out_dim, m_sel, S, m_dens, nullptr, stream));
    // -- visibility test (nerfacc: T >= early_stop_eps && alpha >= min(alpha_thre, occs.mean())) with the visible samples
    //    of every ray counted in the same launch; nerfacc.pack_info of the KEPT samples (nersemble_instant_ngp.py:325) is the
    //    scan of those counts, and with it the compaction is one wave per ray (round 4: mask -> three-kernel scan
    //    compaction -> gathers -> memset -> histogram of the kept ray indices -> scan: 10 launches, now 4)
    int64_t* k_counts = at<int64_t>(w, p.k_counts);
    int64_t* k_packed = at<int64_t>(w, p.k_packed);
    int64_t* k_total = at<int64_t>(w, p.k_total);
    NSX_TRY(nsx_render_visibility(m_t0, m_t1, m_dens, a->packed_march, R, m_vis, k_counts, a->early_stop_eps, 0.0f,
                                  a->alpha_thre_dev, stream));
    NSX_TRY(nsx_pack_info(k_counts, R, k_packed, k_total, stream));
    NSX_TRY(nsx_compact_rays(m_vis, a->packed_march, k_packed, R, m_keep, k_total, n_kept, stream));
    // -- the kept samples: ray index and interval, the rays' fields (through the marched samples' ray indices), the sigma
    //    pass's forward values -- one launch, rows [*n_kept, S) of every destination zeroed
    {
        const void* srcs[9] = {m_ri, m_t0, m_t1, a->origins, a->directions, m_off, m_feat, m_base, m_slot};
        void* dsts[9] = {w + p.k_ri, w + p.k_t0, w + p.k_t1, w + p.k_org, w + p.k_dir, w + p.k_off, w + p.k_feat,
                         w + p.k_base, w + p.k_slot};
        const int64_t rb[9] = {8, 4, 4, 12, 12, 12, 64, 32, 4};
        const uint8_t hop[9] = {0, 0, 0, 1, 1, 0, 0, 0, 0};
        NSX_TRY(nsx_gather_rows_via(9, srcs, rb, dsts, m_keep, m_ri, hop, S, n_kept, stream));
    }
    return NSX_OK;
}

static int check_main(const nsx_step_main* a, const char* who) {
    NSX_REQUIRE(a && a->plan && a->ws_sample && a->ws_fwd && a->out, "%s: NULL argument", who);
    NSX_REQUIRE(a->S >= 1 && a->S == a->plan->S && a->R == a->plan->R, "%s: S / R do not match the plan", who);
    NSX_REQUIRE(a->image && a->depth_targets && a->tables && a->geom && a->code_hash && a->deform_packed && a->code_deform &&
                a->base_w16 && a->head_w16, "%s: NULL argument", who);
    NSX_REQUIRE(a->base_out_dim == 16 && a->geom->n_levels * 2 == 32, "%s: 16 levels x 2 features, 16-wide mlp_base output",
                who);
    return NSX_OK;
}

int nsx_step_main_fwd(const nsx_step_main* a, void* stream) {
    if (int rc = check_main(a, "nsx_step_main_fwd")) return rc;
    const nsx_step_plan& p = *a->plan;
    const int64_t S = a->S, R = a->R;
    const uint8_t* ws = a->ws_sample;
    uint8_t* wf = a->ws_fwd;
    const int64_t* n_dev = cat<int64_t>(ws, p.n_kept);
    const float* org = cat<float>(ws, p.k_org);
    const float* dir = cat<float>(ws, p.k_dir);
    const float* t0 = cat<float>(ws, p.k_t0);
    const float* t1 = cat<float>(ws, p.k_t1);
    const float* off = cat<float>(ws, p.k_off);
    const nsx_half* base_out = cat<nsx_half>(ws, p.k_base);
    const int64_t* packed = cat<int64_t>(ws, p.k_packed);
    float* pos = at<float>(wf, p.f_pos);
    float* pn = at<float>(wf, p.f_pn);
    uint8_t* sel = at<uint8_t>(wf, p.f_sel);
    float* dens = at<float>(wf, p.f_dens);
    nsx_half* rgb16 = at<nsx_half>(wf, p.f_rgb16);
    float* wgt = at<float>(wf, p.f_w);
    float* rgb = at<float>(wf, p.f_rgb);
    float* acc = at<float>(wf, p.f_acc);
    float* depth = at<float>(wf, p.f_depth);
    float* aux = at<float>(wf, p.f_aux);
    float* clip = at<float>(wf, p.f_clip);
    float* per_ray = at<float>(wf, p.f_per_ray);
    // positions, scene-box normalisation of (position + offset) and the in-box selector (offsets: the sigma pass's)
    NSX_TRY(nsx_sample_positions(org, dir, nullptr, t0, t1, off, S, a->field_aabb, pos, pn, sel, n_dev, stream));
    // hash features and mlp_base output are the sigma pass's (same samples, same parameters): density, colour
    NSX_TRY(nsx_density_fwd(base_out, a->base_out_dim, sel, S, dens, n_dev, stream));
    NSX_CALL("nsx_mlp_fwd", S, a->head_hidden, 0, 1,
             nsx_mlp_fwd(a->head_w16, a->head_hidden, S, dir, 3, 3, 0.5f, 0.5f, base_out, a->base_out_dim, 1, a->geo_dim, 3,
                         a->head_act, rgb16, 3, n_dev, stream));
    NSX_TRY(nsx_composite_fwd_h(t0, t1, dens, rgb16, off, packed, R, a->background, clip, wgt, rgb, acc, depth, aux, stream));
    NSX_TRY(nsx_sample_losses_fwd(wgt, t0, t1, packed, R, a->depth_targets, a->eps, a->max_ray, per_ray, stream));
    NSX_TRY(nsx_ray_losses_fwd(rgb, acc, depth, a->image, a->alpha_map, a->depth_targets, per_ray, packed, R, a->use_masked,
                               a->thr, a->l_alpha, a->l_depth, a->l_dist, a->l_empty, a->l_near, a->out, stream));
    return NSX_OK;
}

int nsx_step_main_bwd(const nsx_step_main* a, int stage, void* stream) {
    if (int rc = check_main(a, "nsx_step_main_bwd")) return rc;
    NSX_REQUIRE(a->grad_out && a->ws_bwd && a->grads, "nsx_step_main_bwd: NULL argument");
    NSX_REQUIRE(stage >= 0 && stage <= 2, "nsx_step_main_bwd: stage %d not in [0,2]", stage);
    const nsx_step_plan& p = *a->plan;
    const int64_t S = a->S, R = a->R;
    const uint8_t* ws = a->ws_sample;
    const uint8_t* wf = a->ws_fwd;
    uint8_t* wb = a->ws_bwd;
    uint8_t* wg = a->grads;
    const int64_t* n_dev = cat<int64_t>(ws, p.n_kept);
    const float* dir = cat<float>(ws, p.k_dir);
    const float* t0 = cat<float>(ws, p.k_t0);
    const float* t1 = cat<float>(ws, p.k_t1);
    const int32_t* slot = cat<int32_t>(ws, p.k_slot);
    const nsx_half* feats = cat<nsx_half>(ws, p.k_feat);
    const nsx_half* base_out = cat<nsx_half>(ws, p.k_base);
    const int64_t* packed = cat<int64_t>(ws, p.k_packed);
    const float* pos = cat<float>(wf, p.f_pos);
    const float* pn = cat<float>(wf, p.f_pn);
    const uint8_t* sel = cat<uint8_t>(wf, p.f_sel);
    const float* dens = cat<float>(wf, p.f_dens);
    const nsx_half* rgb16 = cat<nsx_half>(wf, p.f_rgb16);
    const float* wgt = cat<float>(wf, p.f_w);
    const float* rgb = cat<float>(wf, p.f_rgb);
    const float* acc = cat<float>(wf, p.f_acc);
    const float* depth = cat<float>(wf, p.f_depth);
    const float* clip = cat<float>(wf, p.f_clip);
    float* dout = at<float>(wb, p.b_dout);
    float* dx = at<float>(wb, p.b_dx);
    if (stage == 0) {
        float* g_rgb = at<float>(wb, p.b_grgb);
        float* g_acc = at<float>(wb, p.b_gacc);
        float* g_dep = at<float>(wb, p.b_gdep);
        float* g3 = at<float>(wb, p.b_g3);
        float* gw = at<float>(wb, p.b_gw);
        float* ds = at<float>(wb, p.b_ds);
        nsx_half* dc16 = at<nsx_half>(wb, p.b_dc16);
        nsx_half* dbase = at<nsx_half>(wb, p.b_dbase);
        // -- losses
        NSX_TRY(nsx_ray_losses_bwd(rgb, acc, depth, a->image, a->alpha_map, a->depth_targets, R, a->use_masked, a->thr,
                                   a->l_alpha, a->l_depth, a->l_dist, a->l_empty, a->l_near, R, a->out, a->grad_out, g_rgb,
                                   g_acc, g_dep, g3, stream));
        NSX_TRY(nsx_sample_losses_bwd(wgt, t0, t1, packed, R, a->depth_targets, a->eps, a->max_ray, R,
                                      a->out + NSX_LOSS_SAMPLE_SUMS, g3, gw, stream));
        // -- every zero-initialised buffer of the backward (two fills: the scratch part and the parameter gradients)
        if (hipMemsetAsync(wb + p.b_ds, 0, (size_t)(p.b_zero_end - p.b_ds), (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(wg + p.g_head, 0, (size_t)(p.g_zero_end - p.g_head), (hipStream_t)stream) != hipSuccess)
            return hip_fail(hipGetLastError(), "nsx_step_main_bwd: clearing the gradient buffers");
        // -- compositing, mlp_head, trunc_exp density, mlp_base
        NSX_TRY(nsx_composite_bwd_h(t0, t1, dens, rgb16, packed, R, a->background, clip, acc, depth, gw, g_rgb, g_acc, g_dep,
                                    ds, dc16, stream));
        NSX_CALL("nsx_mlp_bwd", S, a->head_hidden, 0, 1,
                 nsx_mlp_bwd(a->head_w16, a->head_hidden, S, dir, 3, 3, 0.5f, 0.5f, base_out, a->base_out_dim, 1, a->geo_dim,
                             3, a->head_act, dc16, 3, at<float>(wg, p.g_head), nullptr, dbase, nullptr, n_dev, stream));
        NSX_TRY(nsx_density_bwd(base_out, a->base_out_dim, sel, ds, S, dbase, n_dev, stream));
        NSX_CALL("nsx_mlp_bwd", S, a->base_hidden, 0, 1,
                 nsx_mlp_bwd(a->base_w16, a->base_hidden, S, nullptr, 0, 0, 1.0f, 0.0f, feats, 32, 0, 32, a->base_out_dim,
                             a->base_act, dbase, a->base_out_dim, at<float>(wg, p.g_base), nullptr, nullptr, dout, n_dev,
                             stream));
        return NSX_OK;
    }
    if (stage == 1) {
        // -- HashEnsemble: factored table gradient into G, code gradient summed per code row, position gradient
        float* G_fused = a->G;
        int hrows = a->n_code_rows;
        if (a->hash_planes > 0) {
            // every code row is the same one (compact first-grid phase): the planes only spread the atomics -- P of them
            // instead of one per row, P / rows of the bytes for the scatter's footprint and for whoever reads G afterwards
            NSX_REQUIRE(a->hash_planes <= a->n_code_rows && !a->need_code_grad,
                        "nsx_step_main_bwd: hash_planes %d with %d code rows (code gradient %d)", a->hash_planes,
                        a->n_code_rows, a->need_code_grad);
            int32_t* plane = at<int32_t>(wb, p.b_plane);
            const unsigned blocks = (unsigned)std::min<int64_t>((S + 255) / 256, 4096);
            hipLaunchKernelGGL(plane_of_slot_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, slot, S,
                               a->hash_planes, plane, n_dev);
            NSX_LAUNCH_CHECK("plane_of_slot_kernel launch");
            slot = plane;
            hrows = a->hash_planes;
        }
        if (a->scatter_separately && a->G) {
            NSX_CALL("nsx_hash_ensemble_bwd_scatter", S, a->H, hrows, 1,
                     nsx_hash_ensemble_bwd_scatter(pn, S, a->geom, hrows, slot, dout, a->G, a->nonfinite, 8, n_dev,
                                                   stream));
            G_fused = nullptr;
        }
        float* nonfinite = G_fused ? a->nonfinite : nullptr;
        if (a->need_code_grad) {
            NSX_CALL("nsx_hash_ensemble_bwd_codesum", S, a->H, a->n_code_rows, 1,
                     nsx_hash_ensemble_bwd_codesum(pn, S, a->tables, a->H, a->geom, a->code_hash, a->code_hash_stride,
                                                   a->n_code_rows, slot, a->hash_window, dout, G_fused,
                                                   at<float>(wg, p.g_code_hash), at<float>(wb, p.b_csum), dx, nonfinite,
                                                   n_dev, stream));
        } else {
            NSX_CALL("nsx_hash_ensemble_bwd_factored", S, a->H, hrows, 1,
                     nsx_hash_ensemble_bwd_factored(pn, S, a->tables, a->H, a->geom, a->code_hash, a->code_hash_stride,
                                                    hrows, slot, a->hash_window, dout, G_fused, nullptr, dx,
                                                    nonfinite, n_dev, stream));
        }
        return NSX_OK;
    }
    // -- stage 2: normalisation (gradient of the offsets), deformation field
    float* goff = at<float>(wb, p.b_goff);
    NSX_TRY(nsx_normalise_bwd(dx, sel, S, a->field_aabb, goff, n_dev, stream));
    NSX_CALL("nsx_deform_bwd", S, 0, a->n_code_rows, 1,
             nsx_deform_bwd(a->deform_packed, pos, S, a->deform_aabb, a->code_deform, a->code_deform_stride, slot,
                            a->n_code_rows, a->window7_host, goff, wb + p.b_deform, at<float>(wg, p.g_deform),
                            at<float>(wg, p.g_code_deform), nullptr, n_dev, stream));
    return NSX_OK;
}

int nsx_step_profile(int enable, int tag) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = enable != 0;
    g_prof_tag = tag;
    return NSX_OK;
}

int nsx_step_profile_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof.size();
}

int nsx_step_profile_get(int i, char* name_out, int name_capacity, float* ms, int64_t* rows, int32_t* info4) {
    NSX_REQUIRE(name_out && name_capacity > 1 && ms && rows && info4, "nsx_step_profile_get: NULL argument");
    ProfRec r;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        NSX_REQUIRE(i >= 0 && i < (int)g_prof.size(), "nsx_step_profile_get: record %d of %d", i, (int)g_prof.size());
        r = g_prof[i];
    }
    strncpy(name_out, r.name, (size_t)name_capacity - 1);
    name_out[name_capacity - 1] = 0;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(ms, r.a, r.b) != hipSuccess)
        return hip_fail(hipGetLastError(), "nsx_step_profile_get");
    *rows = r.rows;
    info4[0] = r.H; info4[1] = r.n_slots; info4[2] = r.counted; info4[3] = r.tag;
    return NSX_OK;
}

int nsx_step_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (const ProfRec& r : g_prof) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    return NSX_OK;
}

int64_t nsx_step_sizeof(int kind) {
    switch (kind) {
        case 0: return (int64_t)sizeof(nsx_step_sample);
        case 1: return (int64_t)sizeof(nsx_step_main);
        case 2: return (int64_t)sizeof(nsx_step_plan);
    }
    return -1;
}

int nsx_step_echo(int kind, const void* s, double* out, int capacity) {
    NSX_REQUIRE(s && out && capacity > 0, "nsx_step_echo: NULL argument");
    int n = 0;
#define PUT(v)                                                                       \
    do {                                                                             \
        NSX_REQUIRE(n < capacity, "nsx_step_echo: capacity %d too small", capacity); \
        out[n++] = (double)(v);                                                      \
    } while (0)
#define PUTP(v) PUT((uintptr_t)(v))
    if (kind == 0) {
        const nsx_step_sample* a = static_cast<const nsx_step_sample*>(s);
        PUTP(a->origins); PUTP(a->directions); PUTP(a->near_planes); PUTP(a->packed_march); PUTP(a->binaries);
        PUTP(a->ray_slots); PUTP(a->ray_times); PUTP(a->row_timesteps); PUTP(a->rows_flag); PUTP(a->deform_packed); PUTP(a->deform_codes); PUTP(a->tables);
        PUTP(a->geom); PUTP(a->hash_codes); PUTP(a->hash_window); PUTP(a->base_w16); PUTP(a->alpha_thre_dev);
        PUTP(a->window7_host); PUTP(a->ws); PUTP(a->plan); PUTP(a->tables_ready_event);
        PUTP(a->march_stash);
        PUT(a->R); PUT(a->S); PUT(a->deform_code_stride); PUT(a->hash_code_stride);
        PUT(a->grid_res); PUT(a->H); PUT(a->base_hidden); PUT(a->base_out_dim); PUT(a->base_act); PUT(a->n_code_rows);
        PUT(a->n_timesteps); PUT(a->phase); PUT(a->march_stash_cap);
        PUT(a->far_plane); PUT(a->step); PUT(a->early_stop_eps); PUT(a->reserved_f);
        for (int i = 0; i < 6; ++i) PUT(a->occ_aabb[i]);
        for (int i = 0; i < 6; ++i) PUT(a->deform_aabb[i]);
        for (int i = 0; i < 6; ++i) PUT(a->field_aabb[i]);
    } else if (kind == 1) {
        const nsx_step_main* a = static_cast<const nsx_step_main*>(s);
        PUTP(a->ws_sample); PUTP(a->ws_fwd); PUTP(a->out); PUTP(a->image); PUTP(a->alpha_map); PUTP(a->depth_targets);
        PUTP(a->tables); PUTP(a->geom); PUTP(a->code_hash); PUTP(a->hash_window); PUTP(a->deform_packed);
        PUTP(a->code_deform); PUTP(a->base_w16); PUTP(a->head_w16); PUTP(a->window7_host); PUTP(a->grad_out);
        PUTP(a->ws_bwd); PUTP(a->grads); PUTP(a->G); PUTP(a->nonfinite); PUTP(a->plan);
        PUT(a->R); PUT(a->S); PUT(a->code_hash_stride); PUT(a->code_deform_stride); PUT(a->max_ray);
        PUT(a->H); PUT(a->n_code_rows); PUT(a->base_hidden); PUT(a->base_out_dim); PUT(a->base_act); PUT(a->head_hidden);
        PUT(a->head_act); PUT(a->geo_dim); PUT(a->use_masked); PUT(a->need_code_grad); PUT(a->scatter_separately);
        PUT(a->hash_planes);
        PUT(a->background); PUT(a->thr); PUT(a->l_alpha); PUT(a->l_depth); PUT(a->l_dist); PUT(a->l_empty); PUT(a->l_near);
        PUT(a->eps);
        for (int i = 0; i < 6; ++i) PUT(a->field_aabb[i]);
        for (int i = 0; i < 6; ++i) PUT(a->deform_aabb[i]);
    } else {
        set_error("nsx_step_echo: kind %d", kind);
        return NSX_ERR_INVALID;
    }
#undef PUT
#undef PUTP
    return n;
}

}  // extern "C"

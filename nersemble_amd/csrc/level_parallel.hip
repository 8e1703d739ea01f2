// level_parallel.hip -- the device side of the level-parallel HashEnsemble exchange (include/nsx.h, "level-parallel
// exchange"; host side: nersemble_amd/engine/level_parallel.py).
//
// The reference trains on one GPU (scripts/train/train_nersemble.py:272-274).  In a data-parallel run with the coarse-to-fine
// window open, rank r owns levels [r L / W, (r + 1) L / W) of all H hash grids: the columns out[:, 2 l + f] of
// HashEnsemble.forward (hash_ensemble.py:93-158) depend on level l's entries only, so samples travel to the levels' owners
// instead of parameters to the samples.  One training step is FOUR collectives on buffers whose layout this file defines:
//
//   forward   all-gather   every rank's [count | positions | code slots | conditioned code rows]      (nsx_lp_fwd_pack)
//             owners run the UNCHANGED nsx_hash_ensemble_fwd on their sub-geometry per source rank     (nsx_lp_fwd_run)
//             all-to-all   fp16 column blocks [S_cap][2 L / W] back to the samples' owners            (nsx_lp_fwd_unpack)
//   backward  all-to-all   [count | dL/dfeatures column block fp16 | positions | slots] per owner     (nsx_lp_bwd_pack)
//             owners run the UNCHANGED nsx_hash_ensemble_bwd_codesum per source rank: the factored table gradient of the
//             owned entries is complete where its optimizer state lives                                (nsx_lp_bwd_run)
//             all-to-all   [partial dL/dx | partial code-row gradient] back, summed over the owners   (nsx_lp_bwd_unpack)
//
// Counts of valid rows travel inside the payloads (device memory end to end: no host synchronisation, no collective of
// their own); capacities are host-known (the marcher's count, exchanged once per step by the host side).
#include "nsx_common.h"
#include <cstring>

namespace nsx {

static inline int64_t lp_up256(int64_t v) { return (v + 255) / 256 * 256; }

__global__ __launch_bounds__(256) void lp_fwd_pack_kernel(const float* __restrict__ pn, const int32_t* __restrict__ slot,
                                                          int64_t S, const int64_t* __restrict__ n_dev,
                                                          const float* __restrict__ codes, int64_t code_stride, int rows,
                                                          int H, uint8_t* __restrict__ payload, const nsx_lp_layout lay) {
    int64_t count = S;
    if (n_dev) {
        const int64_t n = *n_dev;
        count = n < 0 ? 0 : (n < S ? n : S);
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (tid == 0) *reinterpret_cast<int64_t*>(payload + lay.f_count) = count;
    float* dpn = reinterpret_cast<float*>(payload + lay.f_pn);
    int32_t* dsl = reinterpret_cast<int32_t*>(payload + lay.f_slot);
    float* dco = reinterpret_cast<float*>(payload + lay.f_codes);
    for (int64_t i = tid; i < count * 3; i += nthreads) dpn[i] = pn[i];
    for (int64_t i = tid; i < count; i += nthreads) dsl[i] = slot[i];
    for (int64_t i = tid; i < (int64_t)rows * H; i += nthreads) dco[i] = codes[(i / H) * code_stride + (i % H)];
}

struct LpPrefix { int32_t base[NSX_MAX_LEVELS + 1]; };   // plane base of every source rank (W <= levels <= 32)

// codes_packed[plane_base[j] + r][h] = gathered_j.codes[r][h]: the conditioned code rows of the job in gradient-plane order
__global__ __launch_bounds__(256) void lp_codes_pack_kernel(const uint8_t* __restrict__ gathered, const nsx_lp_layout lay,
                                                            const LpPrefix pre, float* __restrict__ out) {
    const int j = blockIdx.x;
    const int rows = pre.base[j + 1] - pre.base[j];
    const float* src = reinterpret_cast<const float*>(gathered + (int64_t)j * lay.fwd_bytes + lay.f_codes);
    float* dst = out + (int64_t)pre.base[j] * lay.H;
    for (int i = threadIdx.x; i < rows * lay.H; i += blockDim.x) dst[i] = src[i];
}

// feats[s][j * n2 + c] = recv_j[s][c]  (dword = one level's two features)
__global__ __launch_bounds__(256) void lp_fwd_unpack_kernel(const uint8_t* __restrict__ recv, int64_t S,
                                                            const int64_t* __restrict__ n_dev, uint32_t* __restrict__ feats,
                                                            const nsx_lp_layout lay) {
    int64_t count = S;
    if (n_dev) {
        const int64_t n = *n_dev;
        count = n < 0 ? 0 : (n < S ? n : S);
    }
    const int n_own = lay.n2 / 2, P = lay.W * n_own;
    const int64_t total = count * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / P;
        const int rem = (int)(i - s * P);
        const int j = rem / n_own, c = rem - j * n_own;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(recv + (int64_t)j * lay.feat_bytes);
        feats[s * P + lay.level_of[rem]] = src[s * n_own + c];
    }
}

__global__ __launch_bounds__(256) void lp_bwd_pack_kernel(const float* __restrict__ dout, const float* __restrict__ pn,
                                                          const int32_t* __restrict__ slot, int64_t S,
                                                          const int64_t* __restrict__ n_dev, uint8_t* __restrict__ send,
                                                          const nsx_lp_layout lay) {
    int64_t count = S;
    if (n_dev) {
        const int64_t n = *n_dev;
        count = n < 0 ? 0 : (n < S ? n : S);
    }
    const int W = lay.W, n_own = lay.n2 / 2, P = W * n_own;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (tid < W) *reinterpret_cast<int64_t*>(send + tid * lay.bwd_bytes + lay.b_count) = count;
    const float2* d2 = reinterpret_cast<const float2*>(dout);
    for (int64_t i = tid; i < count * P; i += nthreads) {
        const int64_t s = i / P;
        const int rem = (int)(i - s * P);
        const int j = rem / n_own, c = rem - j * n_own;
        const float2 v = d2[s * P + lay.level_of[rem]];
        uint32_t* dz = reinterpret_cast<uint32_t*>(send + (int64_t)j * lay.bwd_bytes + lay.b_dz);
        dz[s * n_own + c] = as_u32(half2_t{(half_t)v.x, (half_t)v.y});
    }
    for (int64_t i = tid; i < count * 3; i += nthreads) {
        const float v = pn[i];
        for (int j = 0; j < W; ++j) reinterpret_cast<float*>(send + (int64_t)j * lay.bwd_bytes + lay.b_pn)[i] = v;
    }
    for (int64_t i = tid; i < count; i += nthreads) {
        const int32_t v = slot[i];
        for (int j = 0; j < W; ++j) reinterpret_cast<int32_t*>(send + (int64_t)j * lay.bwd_bytes + lay.b_slot)[i] = v;
    }
}

// dz32[j][s][c] = float(recv_j.dz[s][c]) for s < recv_j.count; the code-gradient sections of `ret` cleared (a source rank
// without samples gets no kernel of its own)
__global__ __launch_bounds__(256) void lp_bwd_arrive_kernel(const uint8_t* __restrict__ recv, float* __restrict__ dz32,
                                                            uint8_t* __restrict__ ret, const nsx_lp_layout lay) {
    const int j = blockIdx.y;
    const uint8_t* blk = recv + (int64_t)j * lay.bwd_bytes;
    int64_t count = *reinterpret_cast<const int64_t*>(blk + lay.b_count);
    count = count < 0 ? 0 : (count < lay.S_cap ? count : lay.S_cap);
    const int n_own = lay.n2 / 2;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(blk + lay.b_dz);
    float2* dst = reinterpret_cast<float2*>(dz32 + (int64_t)j * lay.S_cap * lay.n2);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < count * n_own; i += nthreads) {
        const half2_t h = as_half2(src[i]);
        dst[i] = float2{(float)h.x, (float)h.y};
    }
    float* dc = reinterpret_cast<float*>(ret + (int64_t)j * lay.ret_bytes + lay.r_dcode);
    for (int64_t i = tid; i < (int64_t)lay.R_cap * lay.H; i += nthreads) dc[i] = 0.f;
}

// dx[s][d] = sum_j ret_j.dx[s][d] (s < count), dcode[r][h] = sum_j ret_j.dcode[r][h] (r < rows): fixed order, deterministic
__global__ __launch_bounds__(256) void lp_bwd_unpack_kernel(const uint8_t* __restrict__ ret, int64_t S,
                                                            const int64_t* __restrict__ n_dev, int rows,
                                                            float* __restrict__ dx, float* __restrict__ dcode,
                                                            const nsx_lp_layout lay) {
    int64_t count = S;
    if (n_dev) {
        const int64_t n = *n_dev;
        count = n < 0 ? 0 : (n < S ? n : S);
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (dx) {
        for (int64_t i = tid; i < count * 3; i += nthreads) {
            float acc = 0.f;
            for (int j = 0; j < lay.W; ++j)
                acc += reinterpret_cast<const float*>(ret + (int64_t)j * lay.ret_bytes + lay.r_dx)[i];
            dx[i] = acc;
        }
    }
    if (dcode) {
        for (int64_t i = tid; i < (int64_t)rows * lay.H; i += nthreads) {
            float acc = 0.f;
            for (int j = 0; j < lay.W; ++j)
                acc += reinterpret_cast<const float*>(ret + (int64_t)j * lay.ret_bytes + lay.r_dcode)[i];
            dcode[i] = acc;
        }
    }
}

static inline int lp_blocks(int64_t work) {
    int64_t b = (work + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

static int lp_check_levels(const nsx_lp_layout* lay, const char* who) {
    const int n = lay->W * (lay->n2 / 2);
    uint32_t seen = 0;
    for (int i = 0; i < n; ++i) {
        NSX_REQUIRE(lay->level_of[i] >= 0 && lay->level_of[i] < n && !((seen >> lay->level_of[i]) & 1u),
                    "%s: level_of is not a permutation of 0 .. %d", who, n - 1);
        seen |= 1u << lay->level_of[i];
    }
    return NSX_OK;
}

static int lp_check(const nsx_lp_layout* lay, const char* who) {
    NSX_REQUIRE(lay != nullptr, "%s: layout is NULL", who);
    NSX_REQUIRE(lay->W >= 1 && lay->W <= NSX_MAX_LEVELS && lay->n2 >= 2 && lay->n2 % 2 == 0 && lay->H >= 1 && lay->H <= 32 &&
                lay->R_cap >= 1 && lay->R_cap <= NSX_MAX_SLOTS && lay->S_cap >= 1 && lay->fwd_bytes > 0 &&
                lay->W * (lay->n2 / 2) <= NSX_MAX_LEVELS,
                "%s: not a layout of nsx_lp_layout_make (W=%d n2=%d H=%d R_cap=%d S_cap=%lld)", who, lay->W, lay->n2, lay->H,
                lay->R_cap, (long long)lay->S_cap);
    return NSX_OK;
}

}  // namespace nsx

using namespace nsx;

extern "C" {

int nsx_lp_layout_make(int W, int64_t S_cap, int R_cap, int H, int n2, nsx_lp_layout* out) {
    NSX_REQUIRE(out != nullptr, "nsx_lp_layout_make: out is NULL");
    NSX_REQUIRE(W >= 1 && W <= NSX_MAX_LEVELS, "nsx_lp_layout_make: W=%d not in [1,%d]", W, NSX_MAX_LEVELS);
    NSX_REQUIRE(S_cap >= 1, "nsx_lp_layout_make: S_cap=%lld must be >= 1", (long long)S_cap);
    NSX_REQUIRE(R_cap >= 1 && R_cap <= NSX_MAX_SLOTS, "nsx_lp_layout_make: R_cap=%d not in [1,%d]", R_cap, NSX_MAX_SLOTS);
    NSX_REQUIRE(H >= 1 && H <= 32, "nsx_lp_layout_make: H=%d not in [1,32]", H);
    NSX_REQUIRE(n2 >= 2 && n2 % 2 == 0 && n2 <= 2 * NSX_MAX_LEVELS, "nsx_lp_layout_make: n2=%d (2 x owned levels)", n2);
    memset(out, 0, sizeof(*out));
    NSX_REQUIRE(W * (n2 / 2) <= NSX_MAX_LEVELS, "nsx_lp_layout_make: %d ranks x %d levels exceed %d levels", W, n2 / 2,
                NSX_MAX_LEVELS);
    out->W = W; out->R_cap = R_cap; out->H = H; out->n2 = n2; out->S_cap = S_cap;
    for (int i = 0; i < W * (n2 / 2); ++i) out->level_of[i] = i;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t at = off; off += lp_up256(bytes); return at; };
    out->f_count = take(8);
    out->f_pn = take(S_cap * 12);
    out->f_slot = take(S_cap * 4);
    out->f_codes = take((int64_t)R_cap * H * 4);
    out->fwd_bytes = off;
    out->feat_bytes = lp_up256(S_cap * n2 * 2);
    off = 0;
    out->b_count = take(8);
    out->b_dz = take(S_cap * n2 * 2);
    out->b_pn = take(S_cap * 12);
    out->b_slot = take(S_cap * 4);
    out->bwd_bytes = off;
    off = 0;
    out->r_dx = take(S_cap * 12);
    out->r_dcode = take((int64_t)R_cap * H * 4);
    out->ret_bytes = off;
    return NSX_OK;
}

int nsx_lp_fwd_pack(const nsx_lp_layout* lay, const float* pn, const int32_t* slot, int64_t S, const int64_t* n_device,
                    const float* codes, int64_t code_stride, int rows, uint8_t* payload, void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_fwd_pack")) return rc;
    NSX_REQUIRE(payload && codes && (S == 0 || (pn && slot)), "nsx_lp_fwd_pack: NULL argument");
    NSX_REQUIRE(S >= 0 && S <= lay->S_cap && rows >= 1 && rows <= lay->R_cap,
                "nsx_lp_fwd_pack: S=%lld rows=%d beyond the layout's capacities (%lld, %d)", (long long)S, rows,
                (long long)lay->S_cap, lay->R_cap);
    hipLaunchKernelGGL(lp_fwd_pack_kernel, dim3(lp_blocks(S * 3 + 1)), dim3(256), 0, (hipStream_t)stream, pn, slot, S,
                       n_device, codes, code_stride, rows, lay->H, payload, *lay);
    NSX_LAUNCH_CHECK("nsx_lp_fwd_pack");
    return NSX_OK;
}

int nsx_lp_fwd_run(const nsx_lp_layout* lay, const uint8_t* gathered, const int64_t* sizes_host, const int32_t* rows_host,
                   const nsx_half* tables, const nsx_grid_geom* sub_geom, const float* window, uint8_t* send,
                   float* codes_packed, void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_fwd_run")) return rc;
    NSX_REQUIRE(gathered && sizes_host && rows_host && tables && sub_geom && send, "nsx_lp_fwd_run: NULL argument");
    NSX_REQUIRE(sub_geom->n_levels * 2 == lay->n2, "nsx_lp_fwd_run: the sub-geometry has %d levels, the layout %d columns",
                sub_geom->n_levels, lay->n2);
    LpPrefix pre;
    pre.base[0] = 0;
    for (int j = 0; j < lay->W; ++j) {
        NSX_REQUIRE(sizes_host[j] >= 0 && sizes_host[j] <= lay->S_cap && rows_host[j] >= 1 && rows_host[j] <= lay->R_cap,
                    "nsx_lp_fwd_run: rank %d brings S=%lld rows=%d beyond the layout's capacities", j,
                    (long long)sizes_host[j], rows_host[j]);
        pre.base[j + 1] = pre.base[j] + rows_host[j];
    }
    NSX_REQUIRE(pre.base[lay->W] <= NSX_MAX_ADAM_SLOTS, "nsx_lp_fwd_run: %d code rows in the job's batch (limit %d)",
                pre.base[lay->W], NSX_MAX_ADAM_SLOTS);
    if (codes_packed) {
        hipLaunchKernelGGL(lp_codes_pack_kernel, dim3(lay->W), dim3(256), 0, (hipStream_t)stream, gathered, *lay, pre,
                           codes_packed);
        NSX_LAUNCH_CHECK("nsx_lp_fwd_run (codes)");
    }
    if (option(NSX_OPT_LP_ONE_LAUNCH)) {
        // every source rank's samples in ONE launch (grid.y = source rank): a source's samples on the few owned levels are
        // one or two tiles per wave, W launches of their own are W launch latencies back to back
        EnsSources src{};
        src.x_stride = src.slot_stride = src.count_stride = src.code_stride = lay->fwd_bytes;
        src.out_stride = lay->feat_bytes;
        for (int j = 0; j < lay->W; ++j) { src.B[j] = sizes_host[j]; src.plane_base[j] = pre.base[j]; }
        src.plane_base[lay->W] = pre.base[lay->W];
        return ens_fwd_sources(lay->W, src, gathered + lay->f_pn, tables, lay->H, sub_geom, gathered + lay->f_codes, lay->H,
                               gathered + lay->f_slot, window, send, gathered + lay->f_count, (hipStream_t)stream);
    }
    for (int j = 0; j < lay->W; ++j) {
        if (sizes_host[j] == 0) continue;
        const uint8_t* blk = gathered + (int64_t)j * lay->fwd_bytes;
        const int rc = nsx_hash_ensemble_fwd(reinterpret_cast<const float*>(blk + lay->f_pn), sizes_host[j], tables, lay->H,
                                             sub_geom, reinterpret_cast<const float*>(blk + lay->f_codes), lay->H,
                                             reinterpret_cast<const int32_t*>(blk + lay->f_slot), window,
                                             reinterpret_cast<nsx_half*>(send + (int64_t)j * lay->feat_bytes),
                                             reinterpret_cast<const int64_t*>(blk + lay->f_count), stream);
        if (rc != NSX_OK) return rc;
    }
    return NSX_OK;
}

int nsx_lp_fwd_unpack(const nsx_lp_layout* lay, const uint8_t* recv, int64_t S, const int64_t* n_device, nsx_half* feats,
                      void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_fwd_unpack")) return rc;
    NSX_REQUIRE(S >= 0 && S <= lay->S_cap, "nsx_lp_fwd_unpack: S=%lld beyond the capacity %lld", (long long)S,
                (long long)lay->S_cap);
    if (S == 0) return NSX_OK;
    NSX_REQUIRE(recv && feats, "nsx_lp_fwd_unpack: NULL argument");
    if (int rc = lp_check_levels(lay, "nsx_lp_fwd_unpack")) return rc;
    hipLaunchKernelGGL(lp_fwd_unpack_kernel, dim3(lp_blocks(S * lay->W * (lay->n2 / 2))), dim3(256), 0, (hipStream_t)stream,
                       recv, S, n_device, reinterpret_cast<uint32_t*>(feats), *lay);
    NSX_LAUNCH_CHECK("nsx_lp_fwd_unpack");
    return NSX_OK;
}

int nsx_lp_bwd_pack(const nsx_lp_layout* lay, const float* dout, const float* pn, const int32_t* slot, int64_t S,
                    const int64_t* n_device, uint8_t* send, void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_bwd_pack")) return rc;
    NSX_REQUIRE(send && (S == 0 || (dout && pn && slot)), "nsx_lp_bwd_pack: NULL argument");
    NSX_REQUIRE(S >= 0 && S <= lay->S_cap, "nsx_lp_bwd_pack: S=%lld beyond the capacity %lld", (long long)S,
                (long long)lay->S_cap);
    if (int rc = lp_check_levels(lay, "nsx_lp_bwd_pack")) return rc;
    hipLaunchKernelGGL(lp_bwd_pack_kernel, dim3(lp_blocks(S * lay->W * (lay->n2 / 2) + lay->W)), dim3(256), 0,
                       (hipStream_t)stream, dout, pn, slot, S, n_device, send, *lay);
    NSX_LAUNCH_CHECK("nsx_lp_bwd_pack");
    return NSX_OK;
}

int nsx_lp_bwd_run(const nsx_lp_layout* lay, const uint8_t* recv, const uint8_t* gathered, const int64_t* sizes_host,
                   const int32_t* rows_host, const nsx_half* tables, const nsx_grid_geom* sub_geom, const float* window,
                   float* G, float* dz_scratch, float* csum_scratch, uint8_t* ret, float* nonfinite, void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_bwd_run")) return rc;
    NSX_REQUIRE(recv && gathered && sizes_host && rows_host && tables && sub_geom && dz_scratch && csum_scratch && ret,
                "nsx_lp_bwd_run: NULL argument");
    NSX_REQUIRE(sub_geom->n_levels * 2 == lay->n2, "nsx_lp_bwd_run: the sub-geometry has %d levels, the layout %d columns",
                sub_geom->n_levels, lay->n2);
    hipLaunchKernelGGL(lp_bwd_arrive_kernel, dim3(lp_blocks(lay->S_cap * (lay->n2 / 2)), lay->W), dim3(256), 0,
                       (hipStream_t)stream, recv, dz_scratch, ret, *lay);
    NSX_LAUNCH_CHECK("nsx_lp_bwd_run (arrive)");
    const int64_t entries = sub_geom->offset[sub_geom->n_levels];
    if (option(NSX_OPT_LP_ONE_LAUNCH)) {
        EnsSources src{};
        src.x_stride = src.slot_stride = src.count_stride = lay->bwd_bytes;
        src.code_stride = lay->fwd_bytes;
        src.dout_stride = lay->S_cap * lay->n2 * (int64_t)sizeof(float);
        src.dx_stride = src.rows_stride = lay->ret_bytes;
        src.plane_base[0] = 0;
        for (int j = 0; j < lay->W; ++j) {
            NSX_REQUIRE(sizes_host[j] >= 0 && sizes_host[j] <= lay->S_cap && rows_host[j] >= 1 && rows_host[j] <= lay->R_cap,
                        "nsx_lp_bwd_run: rank %d brings S=%lld rows=%d beyond the layout's capacities", j,
                        (long long)sizes_host[j], rows_host[j]);
            src.B[j] = sizes_host[j];
            src.plane_base[j + 1] = src.plane_base[j] + rows_host[j];
        }
        NSX_REQUIRE(src.plane_base[lay->W] <= NSX_MAX_ADAM_SLOTS, "nsx_lp_bwd_run: %d code rows in the job's batch (limit %d)",
                    src.plane_base[lay->W], NSX_MAX_ADAM_SLOTS);
        return ens_bwd_sources(lay->W, src, recv + lay->b_pn, tables, lay->H, sub_geom, gathered + lay->f_codes, lay->H,
                               recv + lay->b_slot, window, dz_scratch, G, ret + lay->r_dx, G ? nonfinite : nullptr,
                               recv + lay->b_count, csum_scratch, nsx_hash_codesum_scratch_floats(lay->R_cap, lay->H),
                               ret + lay->r_dcode, (hipStream_t)stream);
    }
    int plane = 0;
    for (int j = 0; j < lay->W; ++j) {
        const int rows = rows_host[j];
        NSX_REQUIRE(sizes_host[j] >= 0 && sizes_host[j] <= lay->S_cap && rows >= 1 && rows <= lay->R_cap,
                    "nsx_lp_bwd_run: rank %d brings S=%lld rows=%d beyond the layout's capacities", j,
                    (long long)sizes_host[j], rows);
        if (sizes_host[j] > 0) {
            const uint8_t* blk = recv + (int64_t)j * lay->bwd_bytes;
            uint8_t* rj = ret + (int64_t)j * lay->ret_bytes;
            const float* codes = reinterpret_cast<const float*>(gathered + (int64_t)j * lay->fwd_bytes + lay->f_codes);
            const int rc = nsx_hash_ensemble_bwd_codesum(
                reinterpret_cast<const float*>(blk + lay->b_pn), sizes_host[j], tables, lay->H, sub_geom, codes, lay->H, rows,
                reinterpret_cast<const int32_t*>(blk + lay->b_slot), window, dz_scratch + (int64_t)j * lay->S_cap * lay->n2,
                G ? G + (int64_t)plane * entries * 2 : nullptr, reinterpret_cast<float*>(rj + lay->r_dcode), csum_scratch,
                reinterpret_cast<float*>(rj + lay->r_dx), G ? nonfinite : nullptr,
                reinterpret_cast<const int64_t*>(blk + lay->b_count), stream);
            if (rc != NSX_OK) return rc;
        }
        plane += rows;
    }
    NSX_REQUIRE(plane <= NSX_MAX_ADAM_SLOTS, "nsx_lp_bwd_run: %d code rows in the job's batch (limit %d)", plane,
                NSX_MAX_ADAM_SLOTS);
    return NSX_OK;
}

int nsx_lp_bwd_unpack(const nsx_lp_layout* lay, const uint8_t* ret_recv, int64_t S, const int64_t* n_device, int rows,
                      float* dx, float* dcode, void* stream) {
    if (int rc = lp_check(lay, "nsx_lp_bwd_unpack")) return rc;
    NSX_REQUIRE(ret_recv && S >= 0 && S <= lay->S_cap && rows >= 0 && rows <= lay->R_cap, "nsx_lp_bwd_unpack: bad argument");
    hipLaunchKernelGGL(lp_bwd_unpack_kernel, dim3(lp_blocks(S * 3 + (int64_t)rows * lay->H)), dim3(256), 0,
                       (hipStream_t)stream, ret_recv, S, n_device, rows, dx, dcode, *lay);
    NSX_LAUNCH_CHECK("nsx_lp_bwd_unpack");
    return NSX_OK;
}

int64_t nsx_lp_sizeof(void) { return (int64_t)sizeof(nsx_lp_layout); }

}  // extern "C"

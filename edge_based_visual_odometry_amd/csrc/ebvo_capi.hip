// ebvo_capi.hip -- the C ABI of include/ebvo_hip.h: context, host-buffer entry points,
// device-resident stereo pipeline, profiling.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "ebvo_internal.h"

const char *const g_kernel_names[K_NUM] = {
    "toed_conv",   "toed_nms",   "toed_rowscan", "toed_compact", "toed_finalize", "cand_boxes", "epi_lines",
    "cand_count",  "scan",       "cand_fill",    "edge_patches", "ncc_pairs",     "ncc_stored", "misc"};

// ------------------------------------------------------------------------------------------
int ebvo_fail_hip(ebvo_ctx *ctx, hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx)
        ctx->last_error = buf;
    else
        fprintf(stderr, "[ebvo] %s\n", buf);
    (void)hipGetLastError();
    return EBVO_ERR_HIP;
}

int ebvo_grow(ebvo_ctx *ctx, GrowBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.p)
        return EBVO_OK;
    // the stream may still be using the old allocation
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (b.p)
        hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        ctx->last_error = "hipMalloc failed (out of device memory)";
        return EBVO_ERR_NOMEM;
    }
    b.bytes = want;
    return EBVO_OK;
}

void ebvo_prof_begin(ebvo_ctx *ctx, int kid)
{
    if (!ctx->prof)
        return;
    ProfEvent pe;
    if (!ctx->prof_free.empty())
    {
        pe = ctx->prof_free.back();
        ctx->prof_free.pop_back();
    }
    else
    {
        if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess)
            return;
    }
    pe.kid = kid;
    hipEventRecord(pe.a, ctx->stream);
    ctx->prof_pending.push_back(pe);
}

void ebvo_prof_end(ebvo_ctx *ctx)
{
    if (!ctx->prof || ctx->prof_pending.empty())
        return;
    hipEventRecord(ctx->prof_pending.back().b, ctx->stream);
}

static int prof_drain(ebvo_ctx *ctx)
{
    if (ctx->prof_pending.empty())
        return EBVO_OK;
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (ProfEvent &pe : ctx->prof_pending)
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess)
        {
            ctx->prof_ms[pe.kid] += ms;
            ctx->prof_launches[pe.kid] += 1;
        }
        ctx->prof_free.push_back(pe);
    }
    ctx->prof_pending.clear();
    return EBVO_OK;
}

static void free_buf(GrowBuf &b)
{
    if (b.p)
        hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// ------------------------------------------------------------------------------------------
extern "C" const char *ebvo_strerror(int status)
{
    switch (status)
    {
    case EBVO_OK: return "ok";
    case EBVO_ERR_ARG: return "invalid argument";
    case EBVO_ERR_CAPACITY: return "output capacity too small";
    case EBVO_ERR_HIP: return "HIP runtime error";
    case EBVO_ERR_NOMEM: return "out of memory";
    case EBVO_ERR_STATE: return "call made in the wrong state";
    default: return "unknown status";
    }
}

extern "C" const char *ebvo_last_error(const ebvo_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int ebvo_abi_version(void) { return EBVO_ABI_VERSION; }

extern "C" void ebvo_ctx_destroy(ebvo_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    if (ctx->stream)
        hipStreamSynchronize(ctx->stream);
    for (ProfEvent &pe : ctx->prof_pending)
    {
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    for (ProfEvent &pe : ctx->prof_free)
    {
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    for (int k = 0; k < 2; ++k)
    {
        ImageWS &ws = ctx->im[k];
        hipFree(ws.img);
        hipFree(ws.maps);
        hipFree(ws.flag);
        hipFree(ws.row_cnt);
        hipFree(ws.row_off);
        hipFree(ws.counts);
        hipFree(ws.src);
        hipFree(ws.edges);
        hipFree(ws.all4);
    }
    GrowBuf *bufs[] = {&ctx->lines,      &ctx->boxes_chunk,  &ctx->boxes_group,  &ctx->cand_cnt,  &ctx->row_ptr,
                       &ctx->scan_tmp,   &ctx->col_idx,      &ctx->rc_edges,     &ctx->sims,      &ctx->best,
                       &ctx->keep,       &ctx->patches_raw,  &ctx->patches_norm, &ctx->patches_flag, &ctx->patches_norm_r, &ctx->patches_flag_r,
                       &ctx->match_cnt,  &ctx->scratch_a,    &ctx->scratch_b,    &ctx->scratch_c, &ctx->scratch_d};
    for (GrowBuf *b : bufs)
        free_buf(*b);
    hipFree(ctx->d_params);
    if (ctx->h_small)
        hipHostFree(ctx->h_small);
    if (ctx->stream)
        hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int ebvo_ctx_create(int device, int max_h, int max_w, ebvo_ctx **out)
{
    if (!out || max_h < 32 || max_w < 32 || (int64_t)max_h * max_w > (1ll << 28))
        return EBVO_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return ebvo_fail_hip(nullptr, e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount (no HIP device)",
                             __FILE__, __LINE__);
    if (device < 0 || device >= ndev)
        return EBVO_ERR_ARG;
    ebvo_ctx *ctx = new (std::nothrow) ebvo_ctx();
    if (!ctx)
        return EBVO_ERR_NOMEM;
    ctx->device = device;
    ctx->max_h = max_h;
    ctx->max_w = max_w;
    ctx->cap_edges = max_h * max_w;
    int rc = EBVO_OK;
    auto fail = [&](int code) {
        fprintf(stderr, "[ebvo] ebvo_ctx_create: %s (%s)\n", ebvo_strerror(code), ctx->last_error.c_str());
        ebvo_ctx_destroy(ctx);
        return code;
    };
#define CK(call)                                                             \
    do                                                                       \
    {                                                                        \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess)                                                \
        {                                                                    \
            ebvo_fail_hip(ctx, e_, #call, __FILE__, __LINE__);               \
            return fail(e_ == hipErrorOutOfMemory ? EBVO_ERR_NOMEM : EBVO_ERR_HIP); \
        }                                                                    \
    } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    const size_t H2 = 2 * (size_t)max_h, W2 = 2 * (size_t)max_w, np2 = H2 * W2;
    for (int k = 0; k < 2; ++k)
    {
        ImageWS &ws = ctx->im[k];
        CK(hipMalloc(&ws.img, (size_t)max_h * max_w));
        CK(hipMalloc(&ws.maps, sizeof(double) * np2 * PL_NUM));
        CK(hipMalloc(&ws.flag, np2));
        CK(hipMalloc(&ws.row_cnt, sizeof(int32_t) * 2 * H2));
        CK(hipMalloc(&ws.row_off, sizeof(int32_t) * 2 * (H2 + 1)));
        CK(hipMalloc(&ws.counts, sizeof(int32_t) * 2));
        CK(hipMalloc(&ws.src, sizeof(int32_t) * 2 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.edges, sizeof(ebvo_edge) * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.all4, sizeof(double) * 4 * (size_t)ctx->cap_edges));
    }
    CK(hipMalloc(&ctx->d_params, sizeof(double) * 16));
    CK(hipHostMalloc(&ctx->h_small, sizeof(int32_t) * 64));
#undef CK
    if ((rc = toed_init_constants(ctx)))
        return fail(rc);
    *out = ctx;
    return EBVO_OK;
}

static int check_size(ebvo_ctx *ctx, int h, int w)
{
    if (h < 32 || w < 32 || h > ctx->max_h || w > ctx->max_w)
    {
        ctx->last_error = "image size outside [32, ctx maximum]";
        return EBVO_ERR_ARG;
    }
    return EBVO_OK;
}

static int upload_image(ebvo_ctx *ctx, int slot, const uint8_t *img, int h, int w, ptrdiff_t stride)
{
    if (stride < w)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipMemcpy2DAsync(ctx->im[slot].img, (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h,
                                   hipMemcpyHostToDevice, ctx->stream));
    return EBVO_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int ebvo_toed(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, ebvo_edge *out, int cap,
                         int *n_kept, int *n_total, double *all4, int cap_all, double *t_conv, double *t_nms)
{
    if (!ctx || !img || !n_kept || !n_total || cap < 0 || cap_all < 0 || (cap > 0 && !out))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    if ((rc = upload_image(ctx, 0, img, h, w, stride)))
        return rc;
    ctx->have_pair = ctx->have_run = false;
    float ms_c = 0, ms_n = 0;
    if ((rc = toed_run_device(ctx, 1, h, w, (t_conv || t_nms) ? &ms_c : nullptr, (t_conv || t_nms) ? &ms_n : nullptr)))
        return rc;
    const ImageWS &ws = ctx->im[0];
    *n_kept = ws.n_kept;
    *n_total = ws.n_total;
    if (t_conv) *t_conv = ms_c * 1e-3;
    if (t_nms) *t_nms = ms_n * 1e-3;
    if (ws.n_total > ctx->cap_edges)
    {
        ctx->last_error = "internal edge capacity exceeded";
        return EBVO_ERR_CAPACITY;
    }
    if ((out && ws.n_kept > cap) || (all4 && ws.n_total > cap_all))
        return EBVO_ERR_CAPACITY;
    if (out && ws.n_kept)
        EBVO_HIP(ctx, hipMemcpyAsync(out, ws.edges, sizeof(ebvo_edge) * (size_t)ws.n_kept, hipMemcpyDeviceToHost,
                                     ctx->stream));
    if (all4 && ws.n_total)
        EBVO_HIP(ctx, hipMemcpyAsync(all4, ws.all4, sizeof(double) * 4 * (size_t)ws.n_total, hipMemcpyDeviceToHost,
                                     ctx->stream));
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EBVO_OK;
}

extern "C" int ebvo_toed_pair(ebvo_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                              ptrdiff_t stride_left, ptrdiff_t stride_right, ebvo_edge *out_left,
                              ebvo_edge *out_right, int cap, int n_kept[2], int n_total[2])
{
    if (!ctx || !img_left || !img_right || !n_kept || !n_total || cap < 0)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    if ((rc = upload_image(ctx, 0, img_left, h, w, stride_left)))
        return rc;
    if ((rc = upload_image(ctx, 1, img_right, h, w, stride_right)))
        return rc;
    ctx->have_pair = ctx->have_run = false;
    if ((rc = toed_run_device(ctx, 2, h, w, nullptr, nullptr)))
        return rc;
    ebvo_edge *outs[2] = {out_left, out_right};
    bool too_small = false;
    for (int k = 0; k < 2; ++k)
    {
        n_kept[k] = ctx->im[k].n_kept;
        n_total[k] = ctx->im[k].n_total;
        if (ctx->im[k].n_total > ctx->cap_edges)
            return EBVO_ERR_CAPACITY;
        if (outs[k] && ctx->im[k].n_kept > cap)
            too_small = true;
    }
    if (too_small)
        return EBVO_ERR_CAPACITY;
    for (int k = 0; k < 2; ++k)
        if (outs[k] && ctx->im[k].n_kept)
            EBVO_HIP(ctx, hipMemcpyAsync(outs[k], ctx->im[k].edges, sizeof(ebvo_edge) * (size_t)ctx->im[k].n_kept,
                                         hipMemcpyDeviceToHost, ctx->stream));
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EBVO_OK;
}

extern "C" int ebvo_epipolar_lines(const double F[9], const ebvo_edge *edges, int n, double *lines)
{
    if (!F || (n > 0 && (!edges || !lines)) || n < 0)
        return EBVO_ERR_ARG;
    for (int k = 0; k < n; ++k)
        for (int r = 0; r < 3; ++r)
            lines[(size_t)k * 3 + r] = (F[r * 3 + 0] * edges[k].x + F[r * 3 + 1] * edges[k].y) + F[r * 3 + 2];
    return EBVO_OK;
}

extern "C" int ebvo_epi_candidates(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR,
                                   const double *lines, double epi_thr, double max_disp, double orient_thr_deg,
                                   int stage_mask, int32_t *row_ptr, int32_t *col_idx, int64_t cap,
                                   int64_t *n_pairs)
{
    if (!ctx || nL < 0 || nR < 0 || !row_ptr || !n_pairs || cap < 0 || (cap > 0 && !col_idx) ||
        (nL > 0 && (!L || !lines)) || (nR > 0 && !R) || (stage_mask & ~EBVO_STAGE_ALL) || stage_mask == 0)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_b, sizeof(ebvo_edge) * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_c, sizeof(ebvo_edge) * ((size_t)nR + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->lines, sizeof(double) * 3 * ((size_t)nL + 1))))
        return rc;
    if (nL)
    {
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, L, sizeof(ebvo_edge) * (size_t)nL, hipMemcpyHostToDevice,
                                     ctx->stream));
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->lines.p, lines, sizeof(double) * 3 * (size_t)nL, hipMemcpyHostToDevice,
                                     ctx->stream));
    }
    if (nR)
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_c.p, R, sizeof(ebvo_edge) * (size_t)nR, hipMemcpyHostToDevice,
                                     ctx->stream));
    ctx->have_run = false;
    int64_t np = 0;
    if ((rc = match_candidates_device(ctx, (const ebvo_edge *)ctx->scratch_b.p, nL, (const ebvo_edge *)ctx->scratch_c.p,
                                      nR, (const double *)ctx->lines.p, epi_thr, max_disp, orient_thr_deg, stage_mask,
                                      &np)))
        return rc;
    *n_pairs = np;
    EBVO_HIP(ctx, hipMemcpyAsync(row_ptr, ctx->row_ptr.p, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyDeviceToHost,
                                 ctx->stream));
    int ret = EBVO_OK;
    if (col_idx && np <= cap)
    {
        if (np)
            EBVO_HIP(ctx, hipMemcpyAsync(col_idx, ctx->col_idx.p, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost,
                                         ctx->stream));
    }
    else if (np > cap && (col_idx || cap > 0))
        ret = EBVO_ERR_CAPACITY;
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ret;
}

extern "C" int ebvo_edge_patches(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride,
                                 const ebvo_edge *edges, int n, float *patches)
{
    if (!ctx || !img || n < 0 || (n > 0 && (!edges || !patches)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    if (n == 0)
        return EBVO_OK;
    ctx->have_pair = ctx->have_run = false;
    if ((rc = upload_image(ctx, 0, img, h, w, stride)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_b, sizeof(ebvo_edge) * (size_t)n)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_raw, sizeof(float) * 98 * (size_t)n)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, edges, sizeof(ebvo_edge) * (size_t)n, hipMemcpyHostToDevice,
                                 ctx->stream));
    if ((rc = match_patches_device(ctx, ctx->im[0].img, h, w, w, (const ebvo_edge *)ctx->scratch_b.p, n,
                                   (float *)ctx->patches_raw.p, nullptr, nullptr)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(patches, ctx->patches_raw.p, sizeof(float) * 98 * (size_t)n, hipMemcpyDeviceToHost,
                                 ctx->stream));
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EBVO_OK;
}

// shared by the host entry point and the pipeline: left patches + NCC over the CSR pairs
static int ncc_pairs_core(ebvo_ctx *ctx, int h, int w, const ebvo_edge *d_L, int nL, const ebvo_edge *d_Rc,
                          const int32_t *d_row_ptr, int64_t n_pairs, double thr, bool want_sims)
{
    int rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_raw, sizeof(float) * 98 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_norm, sizeof(float) * 98 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_flag, 2 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->match_cnt, 64)))
        return rc;
    if (want_sims && (rc = ebvo_grow(ctx, ctx->sims, sizeof(double) * 4 * ((size_t)n_pairs + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->best, sizeof(double) * ((size_t)n_pairs + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->keep, (size_t)n_pairs + 1)))
        return rc;
    if ((rc = match_patches_device(ctx, ctx->im[0].img, h, w, w, d_L, nL, (float *)ctx->patches_raw.p,
                                   (float *)ctx->patches_norm.p, (uint8_t *)ctx->patches_flag.p)))
        return rc;
    return match_ncc_pairs_device(ctx, ctx->im[1].img, h, w, w, d_Rc, d_row_ptr, nL, n_pairs,
                                  (const float *)ctx->patches_norm.p, (const uint8_t *)ctx->patches_flag.p, thr,
                                  want_sims ? (double *)ctx->sims.p : nullptr, (double *)ctx->best.p,
                                  (uint8_t *)ctx->keep.p, (int32_t *)ctx->match_cnt.p);
}

extern "C" int ebvo_ncc_pairs(ebvo_ctx *ctx, const uint8_t *imgL, const uint8_t *imgR, int h, int w,
                              ptrdiff_t strideL, ptrdiff_t strideR, const ebvo_edge *L, int nL, const ebvo_edge *Rc,
                              const int32_t *row_ptr, double thr, float *left_patches, double *sims, double *best,
                              uint8_t *keep)
{
    if (!ctx || !imgL || !imgR || nL < 0 || !row_ptr || (nL > 0 && !L))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    if (row_ptr[0] != 0)
        return EBVO_ERR_ARG;
    for (int i = 0; i < nL; ++i)
        if (row_ptr[i + 1] < row_ptr[i])
            return EBVO_ERR_ARG;
    const int64_t np = row_ptr[nL];
    if (np > 0 && !Rc)
        return EBVO_ERR_ARG;
    if (nL == 0)
        return EBVO_OK;
    ctx->have_pair = ctx->have_run = false;
    if ((rc = upload_image(ctx, 0, imgL, h, w, strideL)))
        return rc;
    if ((rc = upload_image(ctx, 1, imgR, h, w, strideR)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_b, sizeof(ebvo_edge) * (size_t)nL)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->rc_edges, sizeof(ebvo_edge) * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->row_ptr, sizeof(int32_t) * ((size_t)nL + 1))))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, L, sizeof(ebvo_edge) * (size_t)nL, hipMemcpyHostToDevice,
                                 ctx->stream));
    if (np)
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->rc_edges.p, Rc, sizeof(ebvo_edge) * (size_t)np, hipMemcpyHostToDevice,
                                     ctx->stream));
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice,
                                 ctx->stream));
    if ((rc = ncc_pairs_core(ctx, h, w, (const ebvo_edge *)ctx->scratch_b.p, nL, (const ebvo_edge *)ctx->rc_edges.p,
                             (const int32_t *)ctx->row_ptr.p, np, thr, sims != nullptr)))
        return rc;
    if (left_patches)
        EBVO_HIP(ctx, hipMemcpyAsync(left_patches, ctx->patches_raw.p, sizeof(float) * 98 * (size_t)nL,
                                     hipMemcpyDeviceToHost, ctx->stream));
    if (np)
    {
        if (sims)
            EBVO_HIP(ctx, hipMemcpyAsync(sims, ctx->sims.p, sizeof(double) * 4 * (size_t)np, hipMemcpyDeviceToHost,
                                         ctx->stream));
        if (best)
            EBVO_HIP(ctx, hipMemcpyAsync(best, ctx->best.p, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost,
                                         ctx->stream));
        if (keep)
            EBVO_HIP(ctx, hipMemcpyAsync(keep, ctx->keep.p, (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EBVO_OK;
}

extern "C" int ebvo_ncc_patches(ebvo_ctx *ctx, const float *A, const float *B, int n, double *sim)
{
    if (!ctx || n < 0 || (n > 0 && (!A || !B || !sim)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0)
        return EBVO_OK;
    int rc;
    const size_t pb = sizeof(float) * 49 * (size_t)n;
    if ((rc = ebvo_grow(ctx, ctx->scratch_b, pb)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_c, pb)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->scratch_d, sizeof(double) * (size_t)n)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, A, pb, hipMemcpyHostToDevice, ctx->stream));
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->scratch_c.p, B, pb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = match_ncc_stored_device(ctx, (const float *)ctx->scratch_b.p, (const float *)ctx->scratch_c.p, n,
                                      (double *)ctx->scratch_d.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(sim, ctx->scratch_d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost,
                                 ctx->stream));
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return EBVO_OK;
}

extern "C" int ebvo_ncc_quads(ebvo_ctx *ctx, const float *kfL, const float *kfR, const float *cfL, const float *cfR,
                              int n, double thr, double *sim_left, double *sim_right, uint8_t *keep)
{
    if (!ctx || n < 0 || (n > 0 && (!kfL || !kfR || !cfL || !cfR || !sim_left || !sim_right)))
        return EBVO_ERR_ARG;
    if (n == 0)
        return EBVO_OK;
    // 8 stored-patch NCCs per quad, order of src/Temporal_Matches.cpp:441-450:
    // (first,first) (first,second) (second,first) (second,second), left then right
    std::vector<float> A((size_t)n * 8 * 49), B((size_t)n * 8 * 49);
    for (int k = 0; k < n; ++k)
        for (int m = 0; m < 8; ++m)
        {
            const float *kf = (m < 4 ? kfL : kfR) + (size_t)k * 98 + ((m >> 1) & 1) * 49;
            const float *cf = (m < 4 ? cfL : cfR) + (size_t)k * 98 + (m & 1) * 49;
            memcpy(&A[((size_t)k * 8 + m) * 49], kf, sizeof(float) * 49);
            memcpy(&B[((size_t)k * 8 + m) * 49], cf, sizeof(float) * 49);
        }
    std::vector<double> s((size_t)n * 8);
    int rc = ebvo_ncc_patches(ctx, A.data(), B.data(), n * 8, s.data());
    if (rc)
        return rc;
    for (int k = 0; k < n; ++k)
    {
        double sl = s[(size_t)k * 8], sr = s[(size_t)k * 8 + 4];
        for (int m = 1; m < 4; ++m)
        {
            if (sl < s[(size_t)k * 8 + m]) sl = s[(size_t)k * 8 + m];
            if (sr < s[(size_t)k * 8 + 4 + m]) sr = s[(size_t)k * 8 + 4 + m];
        }
        sim_left[k] = sl;
        sim_right[k] = sr;
        if (keep)
            keep[k] = (sl > thr && sr > thr) ? 1 : 0; // :452
    }
    return EBVO_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" void ebvo_stereo_default_params(ebvo_stereo_params *p)
{
    if (!p)
        return;
    memset(p, 0, sizeof *p);
    p->epi_thr = EBVO_EPIPOLAR_LINE_DIST_THRESH;
    p->max_disp = EBVO_MAX_DISPARITY;
    p->orient_thr_deg = EBVO_ORIENT_THRESH_DEG;
    p->ncc_thr = EBVO_NCC_THRESH;
    p->stage_mask = EBVO_STAGE_ALL;
}

extern "C" int ebvo_stereo_upload(ebvo_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                                  ptrdiff_t stride_left, ptrdiff_t stride_right)
{
    if (!ctx || !img_left || !img_right)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    ctx->have_pair = ctx->have_run = false;
    if ((rc = upload_image(ctx, 0, img_left, h, w, stride_left)))
        return rc;
    if ((rc = upload_image(ctx, 1, img_right, h, w, stride_right)))
        return rc;
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cur_h = h;
    ctx->cur_w = w;
    ctx->have_pair = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_run(ebvo_ctx *ctx, const ebvo_stereo_params *p, ebvo_stereo_counts *counts)
{
    if (!ctx || !p || !counts || (p->stage_mask & ~EBVO_STAGE_ALL) || p->stage_mask == 0)
        return EBVO_ERR_ARG;
    if (!ctx->have_pair)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const int h = ctx->cur_h, w = ctx->cur_w;
    int rc;
    ctx->have_run = false;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->d_params, p->F21, sizeof(double) * 9, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = toed_run_device(ctx, 2, h, w, nullptr, nullptr)))
        return rc;
    const int nL = ctx->im[0].n_kept, nR = ctx->im[1].n_kept;
    if (ctx->im[0].n_total > ctx->cap_edges || ctx->im[1].n_total > ctx->cap_edges)
        return EBVO_ERR_CAPACITY;
    if ((rc = ebvo_grow(ctx, ctx->lines, sizeof(double) * 3 * ((size_t)nL + 1))))
        return rc;
    if ((rc = match_lines_device(ctx, ctx->d_params, ctx->im[0].edges, nL, (double *)ctx->lines.p)))
        return rc;
    int64_t np = 0;
    if ((rc = match_candidates_device(ctx, ctx->im[0].edges, nL, ctx->im[1].edges, nR, (const double *)ctx->lines.p,
                                      p->epi_thr, p->max_disp, p->orient_thr_deg, p->stage_mask, &np)))
        return rc;
    // NCC from banks: left and right patches are sampled and normalised once per edge
    if ((rc = ebvo_grow(ctx, ctx->patches_raw, sizeof(float) * 98 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_norm, sizeof(float) * 98 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_flag, 2 * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_norm_r, sizeof(float) * 98 * ((size_t)nR + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->patches_flag_r, 2 * ((size_t)nR + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->match_cnt, 64)))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->sims, sizeof(double) * 4 * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->best, sizeof(double) * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, ctx->keep, (size_t)np + 1)))
        return rc;
    if ((rc = match_patches_device(ctx, ctx->im[0].img, h, w, w, ctx->im[0].edges, nL, (float *)ctx->patches_raw.p,
                                   (float *)ctx->patches_norm.p, (uint8_t *)ctx->patches_flag.p)))
        return rc;
    if ((rc = match_patches_device(ctx, ctx->im[1].img, h, w, w, ctx->im[1].edges, nR, nullptr,
                                   (float *)ctx->patches_norm_r.p, (uint8_t *)ctx->patches_flag_r.p)))
        return rc;
    if ((rc = match_ncc_banked_device(ctx, (const int32_t *)ctx->row_ptr.p, (const int32_t *)ctx->col_idx.p, nL, np,
                                      (const float *)ctx->patches_norm.p, (const uint8_t *)ctx->patches_flag.p,
                                      (const float *)ctx->patches_norm_r.p, (const uint8_t *)ctx->patches_flag_r.p,
                                      p->ncc_thr, (double *)ctx->sims.p, (double *)ctx->best.p, (uint8_t *)ctx->keep.p,
                                      (int32_t *)ctx->match_cnt.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->h_small + 8, ctx->match_cnt.p, sizeof(int32_t), hipMemcpyDeviceToHost,
                                 ctx->stream));
    EBVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    counts->n_left = nL;
    counts->n_right = nR;
    counts->n_total_left = ctx->im[0].n_total;
    counts->n_total_right = ctx->im[1].n_total;
    counts->n_pairs = np;
    counts->n_matches = ctx->h_small[8];
    ctx->n_pairs = np;
    ctx->n_matches = counts->n_matches;
    ctx->n_left = nL;
    ctx->have_run = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch(ebvo_ctx *ctx, ebvo_edge *left, ebvo_edge *right, int32_t *row_ptr,
                                 int32_t *col_idx, double *sims, double *best, uint8_t *keep, float *left_patches)
{
    if (!ctx)
        return EBVO_ERR_ARG;
    if (!ctx->have_run)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = (size_t)ctx->im[0].n_kept, nR = (size_t)ctx->im[1].n_kept, np = (size_t)ctx->n_pairs;
    hipStream_t s = ctx->stream;
    if (left && nL)
        EBVO_HIP(ctx, hipMemcpyAsync(left, ctx->im[0].edges, sizeof(ebvo_edge) * nL, hipMemcpyDeviceToHost, s));
    if (right && nR)
        EBVO_HIP(ctx, hipMemcpyAsync(right, ctx->im[1].edges, sizeof(ebvo_edge) * nR, hipMemcpyDeviceToHost, s));
    if (row_ptr)
        EBVO_HIP(ctx, hipMemcpyAsync(row_ptr, ctx->row_ptr.p, sizeof(int32_t) * (nL + 1), hipMemcpyDeviceToHost, s));
    if (col_idx && np)
        EBVO_HIP(ctx, hipMemcpyAsync(col_idx, ctx->col_idx.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost, s));
    if (sims && np)
        EBVO_HIP(ctx, hipMemcpyAsync(sims, ctx->sims.p, sizeof(double) * 4 * np, hipMemcpyDeviceToHost, s));
    if (best && np)
        EBVO_HIP(ctx, hipMemcpyAsync(best, ctx->best.p, sizeof(double) * np, hipMemcpyDeviceToHost, s));
    if (keep && np)
        EBVO_HIP(ctx, hipMemcpyAsync(keep, ctx->keep.p, np, hipMemcpyDeviceToHost, s));
    if (left_patches && nL)
        EBVO_HIP(ctx, hipMemcpyAsync(left_patches, ctx->patches_raw.p, sizeof(float) * 98 * nL,
                                     hipMemcpyDeviceToHost, s));
    EBVO_HIP(ctx, hipStreamSynchronize(s));
    return EBVO_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int ebvo_profile_enable(ebvo_ctx *ctx, int on)
{
    if (!ctx)
        return EBVO_ERR_ARG;
    int rc = prof_drain(ctx);
    ctx->prof = on != 0;
    return rc;
}

extern "C" int ebvo_profile_reset(ebvo_ctx *ctx)
{
    if (!ctx)
        return EBVO_ERR_ARG;
    int rc = prof_drain(ctx);
    for (int k = 0; k < K_NUM; ++k)
    {
        ctx->prof_ms[k] = 0;
        ctx->prof_launches[k] = 0;
    }
    return rc;
}

extern "C" int ebvo_profile_get(ebvo_ctx *ctx, ebvo_kernel_time *out, int *n)
{
    if (!ctx || !out || !n)
        return EBVO_ERR_ARG;
    int rc = prof_drain(ctx);
    for (int k = 0; k < K_NUM; ++k)
    {
        out[k].name = g_kernel_names[k];
        out[k].ms = ctx->prof_ms[k];
        out[k].launches = ctx->prof_launches[k];
    }
    *n = K_NUM;
    return rc;
}

extern "C" int ebvo_fp64_peak(ebvo_ctx *ctx, int iters, double *tflops_muladd, double *tflops_fma)
{
    if (!ctx || iters <= 0)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    return misc_fp64_peak(ctx, iters, tflops_muladd, tflops_fma);
}

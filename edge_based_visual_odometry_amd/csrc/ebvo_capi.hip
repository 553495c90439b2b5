// ebvo_capi.hip -- the C ABI of include/ebvo_hip.h: context and slots, host-buffer entry points,
// the sync-free device-resident stereo pipeline, profiling.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>

#include "ebvo_internal.h"

const char *const g_kernel_names[K_NUM] = {
    "toed_conv",   "toed_nms",   "toed_rowscan", "toed_compact", "toed_finalize", "toed_exact_centre", "toed_exact_mags", "cand_boxes", "epi_lines",
    "cand_count",  "scan",       "cand_fill",    "edge_patches", "ncc_pairs",     "ncc_stored", "misc", "sobel", "gn_refine", "sift"};

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

int ebvo_grow(ebvo_ctx *ctx, Slot &s, GrowBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.p)
        return EBVO_OK;
    // the stream may still be using the old allocation
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4 + 256;
    ++s.buf_gen; // a captured chain of this slot may hold the old address
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

bool ebvo_prof_begin(ebvo_ctx *ctx, Slot &s, int kid)
{
    if (!ctx->prof || !s.prof_now || (ctx->prof_only >= 0 && kid != ctx->prof_only))
        return false;
    ProfEvent pe;
    if (!ctx->prof_free.empty())
    {
        pe = ctx->prof_free.back();
        ctx->prof_free.pop_back();
    }
    else
    {
        if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess)
            return false;
    }
    pe.kid = kid;
    (void)hipEventRecord(pe.a, s.stream);
    s.prof_pending.push_back(pe);
    return true;
}

// Events for ONE kernel launch timed by its own dispatch (hipExtLaunchKernelGGL stamps them with the kernel's begin and
// end, the timestamps a kernel trace reports) instead of a pair recorded around the launch.  False when profiling is off.
bool ebvo_prof_kernel(ebvo_ctx *ctx, Slot &s, int kid, hipEvent_t *a, hipEvent_t *b)
{
    if (!ctx->prof || !s.prof_now || (ctx->prof_only >= 0 && kid != ctx->prof_only))
        return false;
    ProfEvent pe;
    if (!ctx->prof_free.empty())
    {
        pe = ctx->prof_free.back();
        ctx->prof_free.pop_back();
    }
    else
    {
        if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess)
            return false;
    }
    pe.kid = kid;
    s.prof_pending.push_back(pe);
    *a = pe.a;
    *b = pe.b;
    return true;
}

void ebvo_prof_end(ebvo_ctx *ctx, Slot &s)
{
    if (!ctx->prof || !s.prof_now || s.prof_pending.empty())
        return;
    (void)hipEventRecord(s.prof_pending.back().b, s.stream);
}

static int prof_drain(ebvo_ctx *ctx)
{
    for (Slot *sp : ctx->slots)
    {
        Slot &s = *sp;
        if (s.prof_pending.empty())
            continue;
        EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
        for (ProfEvent &pe : s.prof_pending)
        {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess)
            {
                ctx->prof_ms[pe.kid] += ms;
                ctx->prof_launches[pe.kid] += 1;
            }
            ctx->prof_free.push_back(pe);
        }
        s.prof_pending.clear();
    }
    return EBVO_OK;
}

static void free_buf(GrowBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

static void slot_destroy(Slot *s)
{
    if (!s)
        return;
    if (s->own_stream) // (a pair still running on a lane is drained by ebvo_ctx_destroy before any slot goes)
        (void)hipStreamSynchronize(s->own_stream);
    if (s->pair_graph)
        (void)hipGraphExecDestroy(s->pair_graph);
    for (ProfEvent &pe : s->prof_pending)
    {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    for (int k = 0; k < 2; ++k)
    {
        ImageWS &ws = s->im[k];
        (void)hipFree(ws.img_base);
        (void)hipFree(ws.pix2);
        (void)hipFree(ws.raw_base);
        (void)hipFree(ws.undist_xs);
        (void)hipFree(ws.maps);
        (void)hipFree(ws.flag);
        (void)hipFree(ws.row_cnt);
        (void)hipFree(ws.row_off);
        (void)hipFree(ws.counts);
        (void)hipFree(ws.src);
        (void)hipFree(ws.edges);
        (void)hipFree(ws.all4);
        (void)hipFree(ws.cand_rec);
        (void)hipFree(ws.cand_flag);
        (void)hipFree(ws.cand_off);
        (void)hipFree(ws.cand_data);
        (void)hipFree(ws.cand_lists);
        (void)hipFree(ws.cand_lcount);
    }
    GrowBuf *bufs[] = {&s->lines,        &s->boxes_chunk,  &s->boxes_group,    &s->cand_cnt,       &s->cand_stage, &s->cand_tileflag, &s->row_ptr, &s->grad_x, &s->grad_y, &s->gn_xy, &s->gn_out, &s->gn_valid, &s->gn_iters, &s->gn_state, &s->gn_lists, &s->gn_pack, &s->sift_used, &s->sift_img, &s->sift_desc, &s->sift_f32, &s->sift_dist, &s->tq_i32, &s->tq_cols, &s->tq_f64, &s->tq_u8, &s->tq_cells, &s->tq_chain, &s->fin_i32, &s->fin_edges, &s->fin_f64, &s->fin_u8, &s->fin_out,
                       &s->scan_tmp,     &s->col_idx,      &s->rc_edges,       &s->sims,           &s->best,
                       &s->keep,         &s->patches_raw,  &s->patches_norm,   &s->patches_flag,   &s->patches_norm_r,
                       &s->patches_flag_r, &s->pair_left,  &s->sincos,         &s->scratch_b,      &s->scratch_c,
                       &s->scratch_d,    &s->fetch_pack};
    for (GrowBuf *b : bufs)
        free_buf(*b);
    (void)hipFree(s->d_total);
    (void)hipFree(s->d_matches);
    (void)hipFree(s->d_sizes);
    (void)hipFree(s->d_F);
    (void)hipFree(s->d_result);
    if (s->h_result)
        (void)hipHostFree(s->h_result);
    if (s->h_arena)
        (void)hipHostFree(s->h_arena);
    if (s->ev_done)
        (void)hipEventDestroy(s->ev_done);
    if (s->ev_rebind)
        (void)hipEventDestroy(s->ev_rebind);
    if (s->ev_upload)
        (void)hipEventDestroy(s->ev_upload);
    if (s->h_push)
        (void)hipHostFree(s->h_push);
    if (s->h_mail)
        (void)hipHostFree(s->h_mail);
    if (s->h_up)
        (void)hipHostFree(s->h_up);
    (void)hipFree(s->d_fin_tot);
    if (s->h_fin_tot)
        (void)hipHostFree(s->h_fin_tot);
    if (s->ev_fin)
        (void)hipEventDestroy(s->ev_fin);
    (void)hipFree(s->d_tq_tot);
    if (s->h_tq_tot)
        (void)hipHostFree(s->h_tq_tot);
    if (s->ev_tq)
        (void)hipEventDestroy(s->ev_tq);
    if (s->own_stream)
        (void)hipStreamDestroy(s->own_stream);
    delete s;
}

// Result copies of ebvo_stereo_fetch_begin run on the context's copy stream: an entry point that writes buffers of the
// slot lets them finish first (the views stay valid; ebvo_stereo_fetch_end then returns at once).
static int drain_fetch(ebvo_ctx *ctx, Slot &s)
{
    if (s.fetch_pending)
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    if (s.upload_pending) // an asynchronous upload (ebvo_stereo_upload_async) still writes the slot's image buffers
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_upload));
    s.upload_pending = false;
    return EBVO_OK;
}

// an earlier stream-form upload of the slot must have landed before the mailbox names other images (host wait; rare)
static int drain_stream_upload(ebvo_ctx *ctx, Slot &s)
{
    if (s.upload_pending)
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_upload));
    s.upload_pending = false;
    return EBVO_OK;
}

static int slot_create(ebvo_ctx *ctx, Slot **out)
{
    Slot *s = new (std::nothrow) Slot();
    if (!s)
        return EBVO_ERR_NOMEM;
#define CK(call)                                                                   \
    do                                                                             \
    {                                                                              \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess)                                                      \
        {                                                                          \
            ebvo_fail_hip(ctx, e_, #call, __FILE__, __LINE__);                     \
            slot_destroy(s);                                                       \
            return e_ == hipErrorOutOfMemory ? EBVO_ERR_NOMEM : EBVO_ERR_HIP;      \
        }                                                                          \
    } while (0)
    CK(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    CK(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&s->ev_rebind, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&s->ev_upload, hipEventDisableTiming));
    const size_t H2 = 2 * (size_t)ctx->max_h, W2 = 2 * (size_t)ctx->max_w, np2 = H2 * W2;
    for (int k = 0; k < 2; ++k)
    {
        ImageWS &ws = s->im[k];
        CK(hipMalloc(&ws.img_base, (size_t)ctx->max_h * ctx->max_w + 128));
        // on the slot's own stream: hipMemset runs asynchronously on the NULL stream, which a non-blocking stream does
        // not wait for -- the zero fill could land after the first upload of a fresh context
        CK(hipMemsetAsync(ws.img_base, 0, (size_t)ctx->max_h * ctx->max_w + 128, s->stream));
        ws.img = ws.img_base + 64;
        CK(hipMalloc(&ws.pix2, sizeof(uint16_t) * ((size_t)ctx->max_h * ctx->max_w + 64)));
        CK(hipMemsetAsync(ws.pix2, 0, sizeof(uint16_t) * ((size_t)ctx->max_h * ctx->max_w + 64), s->stream));
        CK(hipMalloc(&ws.undist_xs, sizeof(double) * (size_t)ctx->max_w));
        if (ctx->undist_on)
        {
            CK(hipMalloc(&ws.raw_base, (size_t)ctx->max_h * ctx->max_w + 128));
            CK(hipMemsetAsync(ws.raw_base, 0, (size_t)ctx->max_h * ctx->max_w + 128, s->stream));
            ws.raw = ws.raw_base + 64;
        }
        CK(hipMalloc(&ws.maps, sizeof(double) * np2 * PL_NUM));
        CK(hipMalloc(&ws.flag, np2));
        CK(hipMalloc(&ws.row_cnt, sizeof(int32_t) * 2 * H2));
        CK(hipMalloc(&ws.row_off, sizeof(int32_t) * 3 * (H2 + 1)));
        CK(hipMalloc(&ws.counts, sizeof(int32_t) * 8));
        CK(hipMemsetAsync(ws.counts, 0, sizeof(int32_t) * 8, s->stream));
        CK(hipMalloc(&ws.cand_rec, 40 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.cand_flag, sizeof(int32_t) * 2 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.cand_off, sizeof(int32_t) * 2 * ((size_t)ctx->cap_edges + 1)));
        CK(hipMalloc(&ws.cand_data, 64 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.cand_lists, sizeof(int32_t) * 12 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.cand_lcount, sizeof(int32_t) * 12));
        CK(hipMalloc(&ws.src, sizeof(int32_t) * 2 * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.edges, sizeof(ebvo_edge) * (size_t)ctx->cap_edges));
        CK(hipMalloc(&ws.all4, sizeof(double) * 4 * (size_t)ctx->cap_edges));
    }
    CK(hipMalloc(&s->d_total, sizeof(unsigned long long) * (1 + EBVO_TOTAL_PARTS)));
    CK(hipMalloc(&s->d_matches, sizeof(int32_t) * EBVO_MATCH_PARTS));
    CK(hipMalloc(&s->d_sizes, sizeof(int32_t) * 4));
    CK(hipMalloc(&s->d_F, sizeof(double) * 9));
    CK(hipHostMalloc(reinterpret_cast<void **>(&s->h_mail), sizeof(Slot::PullMail)));
    memset(s->h_mail, 0, sizeof(Slot::PullMail));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_mail), s->h_mail, 0));
    CK(hipMalloc(&s->d_result, sizeof(PairResult)));
    CK(hipHostMalloc(&s->h_result, sizeof(PairResult)));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_result_host), s->h_result, 0));
    CK(hipMalloc(&s->d_fin_tot, sizeof(int32_t) * 8));
    CK(hipHostMalloc(&s->h_fin_tot, sizeof(int32_t) * 8));
    CK(hipEventCreateWithFlags(&s->ev_fin, hipEventDisableTiming));
    CK(hipMalloc(&s->d_tq_tot, sizeof(unsigned long long) * 2));
    CK(hipHostMalloc(&s->h_tq_tot, sizeof(unsigned long long) * 2));
    CK(hipEventCreateWithFlags(&s->ev_tq, hipEventDisableTiming));
    CK(hipStreamSynchronize(s->stream));
#undef CK
    *out = s;
    return EBVO_OK;
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

extern "C" int ebvo_set_toed_mode(ebvo_ctx *ctx, int mode)
{
    if (!ctx || (mode != EBVO_TOED_STRICT && mode != EBVO_TOED_HYBRID))
        return EBVO_ERR_ARG;
    ++ctx->graph_gen;
    ctx->toed_mode = mode;
    return EBVO_OK;
}

extern "C" int ebvo_get_toed_mode(const ebvo_ctx *ctx) { return ctx ? ctx->toed_mode : EBVO_ERR_ARG; }

extern "C" int64_t ebvo_toed_fallbacks(const ebvo_ctx *ctx) { return ctx ? ctx->toed_fallbacks : -1; }

extern "C" int64_t ebvo_graph_launches(const ebvo_ctx *ctx) { return ctx ? ctx->graph_launches : -1; }

extern "C" int ebvo_toed_stats(ebvo_ctx *ctx, int slot, int32_t out[8])
{
    if (!ctx || !out || slot < 0 || slot >= (int)ctx->slots.size())
        return EBVO_ERR_ARG;
    Slot &s = *ctx->slots[(size_t)slot];
    if (s.in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 2; ++k)
        EBVO_HIP(ctx, hipMemcpyAsync(out + 4 * k, s.im[k].counts, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

extern "C" void ebvo_ctx_destroy(ebvo_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    for (hipStream_t st : ctx->lane_streams) // pairs still running on a lane use the buffers of their slot
        (void)hipStreamSynchronize(st);
    if (ctx->copy_stream)
        (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->upload_stream)
        (void)hipStreamSynchronize(ctx->upload_stream);
    for (Slot *s : ctx->slots)
        slot_destroy(s);
    if (ctx->upload_stream)
        (void)hipStreamDestroy(ctx->upload_stream);
    for (const ebvo_ctx::HostRange &r : ctx->host_ranges) // (the caller should have unregistered them)
        if (r.ours)
            (void)hipHostUnregister(const_cast<uint8_t *>(r.host));
    for (hipStream_t st : ctx->lane_streams)
        (void)hipStreamDestroy(st);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    for (ProfEvent &pe : ctx->prof_free)
    {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    (void)hipFree(ctx->kf_L);
    (void)hipFree(ctx->kf_R);
    (void)hipFree(ctx->kf_Ln);
    (void)hipFree(ctx->kf_Rn);
    (void)hipFree(ctx->kf_Lf);
    (void)hipFree(ctx->kf_Ld);
    (void)hipFree(ctx->kf_Rd);
    (void)hipFree(ctx->kf_imgL);
    (void)hipFree(ctx->kf_imgR);
    (void)hipFree(ctx->kf_Rf);
    for (PinnedBuf *b : {&ctx->sw_toed[0], &ctx->sw_toed[1], &ctx->sw_cand, &ctx->sw_ncc, &ctx->sw_up})
        if (b->p)
            (void)hipHostFree(b->p);
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
    {
        // both modes return the same bits (every TOED parity test runs in both); hybrid does a third of the work
        const char *m = getenv("EBVO_TOED_MODE");
        ctx->toed_mode = (m && strcmp(m, "strict") == 0) ? EBVO_TOED_STRICT : EBVO_TOED_HYBRID;
        const char *g = getenv("EBVO_GRAPHS"); // "0": the pair chain as direct launches (hipGraph capture is the default)
        ctx->use_graphs = (g && strcmp(g, "0") == 0) ? 0 : 1;
    }
    int rc;
    auto fail = [&](int code) {
        fprintf(stderr, "[ebvo] ebvo_ctx_create: %s (%s)\n", ebvo_strerror(code), ctx->last_error.c_str());
        ebvo_ctx_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess)
        return fail(EBVO_ERR_HIP);
    Slot *s0 = nullptr;
    if ((rc = slot_create(ctx, &s0)))
        return fail(rc);
    ctx->slots.push_back(s0);
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

static int upload_image(ebvo_ctx *ctx, Slot &s, int k, const uint8_t *img, int h, int w, ptrdiff_t stride, bool to_raw = false,
                        hipStream_t on = nullptr)
{
    if (stride < w)
        return EBVO_ERR_ARG;
    uint8_t *dst = to_raw ? s.im[k].raw : s.im[k].img;
    const hipStream_t st = on ? on : s.stream;
    // a tightly packed image is ONE linear copy: the 2-D copy of pageable memory is staged row by row (measured 3 ms
    // per KITTI image against 0.1 ms)
    if (stride == (ptrdiff_t)w)
        EBVO_HIP(ctx, hipMemcpyAsync(dst, img, (size_t)w * h, hipMemcpyHostToDevice, st));
    else
        EBVO_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h,
                                       hipMemcpyHostToDevice, st));
    return EBVO_OK;
}

// host-buffer entry points use slot 0; they must not run while a submitted pair is in flight on it
static int host_slot(ebvo_ctx *ctx, Slot **out)
{
    Slot &s = *ctx->slots[0];
    if (s.in_flight || s.fin_in_flight || s.tq_in_flight)
    {
        ctx->last_error = "slot 0 has submitted work in flight; call ebvo_stereo_wait / ebvo_stereo_finalize_wait / ebvo_temporal_match_wait first";
        return EBVO_ERR_STATE;
    }
    if (int rc = drain_fetch(ctx, s)) // result copies of the previous pair (copy stream) still read the slot's buffers
        return rc;
    s.fetch_pending = false;
    s.have_pair = s.have_run = s.have_refined = s.have_final = false;
    s.sift_left_valid = false;
    s.undist_pair = false;
    s.fetch_what = 0;
    *out = &s;
    return EBVO_OK;
}

// run TOED on n_img resident images of the slot and read the counts back (synchronises)
namespace
{
struct EventTriple // destroyed on every return path
{
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    ~EventTriple()
    {
        for (hipEvent_t ev : e)
            if (ev)
                (void)hipEventDestroy(ev);
    }
};
} // namespace

static int toed_sync(ebvo_ctx *ctx, Slot &s, int n_img, int h, int w, float *ms_conv, float *ms_nms)
{
    EventTriple ev;
    const bool timed = ms_conv || ms_nms;
    if (timed)
        for (hipEvent_t &e : ev.e)
            EBVO_HIP(ctx, hipEventCreate(&e));
    // the counts travel through the slot's pinned record, never through this stack frame: a failure between the
    // enqueue of a copy and the synchronisation cannot leave a DMA pointing at dead memory
    int32_t *hc = reinterpret_cast<int32_t *>(s.h_result);
    static_assert(sizeof(PairResult) >= 6 * sizeof(int32_t), "pinned record too small for the TOED counts");
    int mode = ctx->toed_mode;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        int rc = toed_enqueue(ctx, s, n_img, h, w, ev.e[0], ev.e[1], ev.e[2], mode);
        if (rc)
            return rc;
        hipError_t e = hipSuccess;
        for (int k = 0; k < n_img && e == hipSuccess; ++k)
        {
            e = hipMemcpyAsync(hc + 3 * k, s.im[k].counts, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(hc + 3 * k + 2, s.im[k].counts + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s.stream);
        }
        const hipError_t es = hipStreamSynchronize(s.stream); // also on failure: nothing may stay in flight
        if (e != hipSuccess || es != hipSuccess)
            return ebvo_fail_hip(ctx, e != hipSuccess ? e : es, "TOED count read-back", __FILE__, __LINE__);
        bool fits = true;
        for (int k = 0; k < n_img; ++k)
        {
            s.im[k].n_total = hc[3 * k];
            s.im[k].n_kept = hc[3 * k + 1];
            if (mode == EBVO_TOED_HYBRID && hc[3 * k + 2] > ctx->cap_edges)
                fits = false;
        }
        if (fits)
            break;
        // the screen flagged more grid points than the candidate buffers hold (an image of ties): the strict path has no
        // candidate lists and handles any image
        mode = EBVO_TOED_STRICT;
        ++ctx->toed_fallbacks;
    }
    if (timed)
    {
        float a = 0, b = 0;
        EBVO_HIP(ctx, hipEventElapsedTime(&a, ev.e[0], ev.e[1]));
        EBVO_HIP(ctx, hipEventElapsedTime(&b, ev.e[1], ev.e[2]));
        if (ms_conv) *ms_conv = a;
        if (ms_nms) *ms_nms = b;
    }
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
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    ctx->sw_tag[0] = ctx->sw_tag[1] = 0; // the edge lists of the resident stage-wise calls live in these workspaces
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)))
        return rc;
    float ms_c = 0, ms_n = 0;
    const bool timed = t_conv || t_nms;
    if ((rc = toed_sync(ctx, s, 1, h, w, timed ? &ms_c : nullptr, timed ? &ms_n : nullptr)))
        return rc;
    const ImageWS &ws = s.im[0];
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
                                     s.stream));
    if (all4 && ws.n_total)
        EBVO_HIP(ctx, hipMemcpyAsync(all4, ws.all4, sizeof(double) * 4 * (size_t)ws.n_total, hipMemcpyDeviceToHost,
                                     s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_toed_screen_audit(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, ebvo_screen_audit *out)
{
    if (!ctx || !img || !out)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    ctx->sw_tag[0] = ctx->sw_tag[1] = 0;
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)))
        return rc;
    ctx->screen_audit = true; // the next toed_enqueue keeps the screen's values and runs the audit kernel behind the exact stage
    rc = toed_enqueue(ctx, s, 1, h, w, nullptr, nullptr, nullptr, EBVO_TOED_HYBRID);
    ctx->screen_audit = false;
    // a diagnostic: drain the stream, then plain synchronous copies into this frame
    const hipError_t es = hipStreamSynchronize(s.stream);
    if (rc)
        return rc;
    EBVO_HIP(ctx, es);
    unsigned long long hw[6];
    int32_t hc[5];
    EBVO_HIP(ctx, hipMemcpy(hw, toed_screen_audit_result(s, 0, h, w), sizeof hw, hipMemcpyDeviceToHost));
    EBVO_HIP(ctx, hipMemcpy(hc, s.im[0].counts, sizeof hc, hipMemcpyDeviceToHost));
    if (hc[4] > ctx->cap_edges)
    {
        ctx->last_error = "the screen flagged more grid points than the context holds: nothing to audit";
        return EBVO_ERR_CAPACITY;
    }
    double b[5];
    toed_screen_budget(b);
    auto dbl = [](unsigned long long v) { double d; memcpy(&d, &v, sizeof d); return d; };
    out->n_maxima = hc[0];
    out->n_kept = hc[1];
    out->n_candidates = hc[2];
    out->n_neighbour_points = hc[3];
    out->max_err_gx = dbl(hw[0]);
    out->max_err_gy = dbl(hw[1]);
    out->max_err_mag = dbl(hw[2]);
    out->max_err_mag_neighbours = dbl(hw[3]);
    out->bound_g = b[0];
    out->bound_mag = b[1];
    out->bound_slope = b[2];
    out->tol_mag = b[3];
    out->tol_slope = b[4];
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
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    ctx->sw_tag[0] = ctx->sw_tag[1] = 0;
    if ((rc = upload_image(ctx, s, 0, img_left, h, w, stride_left)))
        return rc;
    if ((rc = upload_image(ctx, s, 1, img_right, h, w, stride_right)))
        return rc;
    if ((rc = toed_sync(ctx, s, 2, h, w, nullptr, nullptr)))
        return rc;
    ebvo_edge *outs[2] = {out_left, out_right};
    bool too_small = false;
    for (int k = 0; k < 2; ++k)
    {
        n_kept[k] = s.im[k].n_kept;
        n_total[k] = s.im[k].n_total;
        if (s.im[k].n_total > ctx->cap_edges)
            return EBVO_ERR_CAPACITY;
        if (outs[k] && s.im[k].n_kept > cap)
            too_small = true;
    }
    if (too_small)
        return EBVO_ERR_CAPACITY;
    for (int k = 0; k < 2; ++k)
        if (outs[k] && s.im[k].n_kept)
            EBVO_HIP(ctx, hipMemcpyAsync(outs[k], s.im[k].edges, sizeof(ebvo_edge) * (size_t)s.im[k].n_kept,
                                         hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
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

static int epi_candidates_impl(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR, const double *lines,
                               double epi_thr, double max_disp, double orient_thr_deg, int stage_mask, int32_t *row_ptr,
                               int32_t *col_idx, uint8_t *orient_ok, int64_t cap, int64_t *n_pairs);

extern "C" int ebvo_epi_candidates(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR,
                                   const double *lines, double epi_thr, double max_disp, double orient_thr_deg,
                                   int stage_mask, int32_t *row_ptr, int32_t *col_idx, int64_t cap,
                                   int64_t *n_pairs)
{
    return epi_candidates_impl(ctx, L, nL, R, nR, lines, epi_thr, max_disp, orient_thr_deg, stage_mask, row_ptr, col_idx, nullptr,
                               cap, n_pairs);
}

extern "C" int ebvo_epi_candidates_staged(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR,
                                          const double *lines, double epi_thr, double max_disp, double orient_thr_deg,
                                          int32_t *row_ptr, int32_t *col_idx, uint8_t *orient_ok, int64_t cap,
                                          int64_t *n_pairs)
{
    if (cap > 0 && !orient_ok)
        return EBVO_ERR_ARG;
    return epi_candidates_impl(ctx, L, nL, R, nR, lines, epi_thr, max_disp, orient_thr_deg,
                               EBVO_STAGE_EPIPOLAR | EBVO_STAGE_DISPARITY, row_ptr, col_idx, orient_ok, cap, n_pairs);
}

static int epi_candidates_impl(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR, const double *lines,
                               double epi_thr, double max_disp, double orient_thr_deg, int stage_mask, int32_t *row_ptr,
                               int32_t *col_idx, uint8_t *orient_ok, int64_t cap, int64_t *n_pairs)
{
    if (!ctx || nL < 0 || nR < 0 || !row_ptr || !n_pairs || cap < 0 || (cap > 0 && !col_idx) ||
        (nL > 0 && (!L || !lines)) || (nR > 0 && !R) || (stage_mask & ~EBVO_STAGE_ALL) || stage_mask == 0)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    *n_pairs = 0;
    if (nL == 0 || nR == 0)
    {
        memset(row_ptr, 0, sizeof(int32_t) * ((size_t)nL + 1));
        return EBVO_OK;
    }
    if ((rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * (size_t)nL)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.scratch_c, sizeof(ebvo_edge) * (size_t)nR)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.lines, sizeof(double) * 3 * (size_t)nL)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, L, sizeof(ebvo_edge) * (size_t)nL, hipMemcpyHostToDevice, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(s.lines.p, lines, sizeof(double) * 3 * (size_t)nL, hipMemcpyHostToDevice, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_c.p, R, sizeof(ebvo_edge) * (size_t)nR, hipMemcpyHostToDevice, s.stream));
    const ebvo_edge *dL = (const ebvo_edge *)s.scratch_b.p, *dR = (const ebvo_edge *)s.scratch_c.p;
    if ((rc = match_candidates_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p, epi_thr,
                                       max_disp, orient_thr_deg, stage_mask, false)))
        return rc;
    // the total lands in the slot's pinned record (not in this stack frame); on any failure the stream is drained
    // before returning so that no copy into the caller's row_ptr stays in flight
    unsigned long long *h_total_p = reinterpret_cast<unsigned long long *>(s.h_result);
    {
        hipError_t e1 = hipMemcpyAsync(h_total_p, s.d_total, sizeof(*h_total_p), hipMemcpyDeviceToHost, s.stream);
        hipError_t e2 = e1 == hipSuccess ? hipMemcpyAsync(row_ptr, s.row_ptr.p, sizeof(int32_t) * ((size_t)nL + 1),
                                                          hipMemcpyDeviceToHost, s.stream)
                                         : hipSuccess;
        hipError_t e3 = hipStreamSynchronize(s.stream);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
            return ebvo_fail_hip(ctx, e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "candidate total read-back",
                                 __FILE__, __LINE__);
    }
    const unsigned long long h_total = *h_total_p;
    if (h_total > 0x7fffffffull)
    {
        ctx->last_error = "candidate list exceeds 2^31-1 pairs";
        return EBVO_ERR_CAPACITY;
    }
    const int64_t np = (int64_t)h_total;
    *n_pairs = np;
    if (np > cap)
        return (col_idx || cap > 0) ? EBVO_ERR_CAPACITY : EBVO_OK;
    if (np == 0 || !col_idx)
        return EBVO_OK;
    if ((rc = ebvo_grow(ctx, s, s.col_idx, sizeof(int32_t) * (size_t)np)))
        return rc;
    s.cap_pairs = np;
    if ((rc = match_candidates_fill_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p,
                                            epi_thr, max_disp, orient_thr_deg, stage_mask)))
        return rc;
    // from here on copies into caller-owned memory are in flight: whatever fails, the stream is drained before returning
    rc = [&]() -> int {
        int r;
        if (orient_ok)
        {
            // the third reference stage (apply_orientation_filter, src/Stereo_Matches.cpp:863-915) as one flag per listed pair
            if ((r = ebvo_grow(ctx, s, s.keep, (size_t)np)) ||
                (r = match_orient_flags_enqueue(ctx, s, dL, nL, dR, (const int32_t *)s.row_ptr.p, (const int32_t *)s.col_idx.p,
                                                np, orient_thr_deg, (uint8_t *)s.keep.p)))
                return r;
        }
        EBVO_HIP(ctx, hipMemcpyAsync(col_idx, s.col_idx.p, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, s.stream));
        if (orient_ok)
            EBVO_HIP(ctx, hipMemcpyAsync(orient_ok, s.keep.p, (size_t)np, hipMemcpyDeviceToHost, s.stream));
        return EBVO_OK;
    }();
    if (rc)
    {
        (void)hipStreamSynchronize(s.stream);
        return rc;
    }
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    s.cap_pairs = 0; // the pipeline re-establishes its own capacity
    return EBVO_OK;
}

extern "C" void ebvo_gn_default_params(ebvo_gn_params *p)
{
    if (!p)
        return;
    p->max_iter = EBVO_GN_MAX_ITER;
    p->tol = EBVO_GN_TOL;
    p->huber_delta = EBVO_GN_HUBER_DELTA;
}

extern "C" int ebvo_sobel_gradients(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, float *gx,
                                    float *gy)
{
    if (!ctx || !img || !gx || !gy)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    const size_t bytes = sizeof(float) * (size_t)h * w;
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)) || (rc = ebvo_grow(ctx, s, s.grad_x, bytes)) ||
        (rc = ebvo_grow(ctx, s, s.grad_y, bytes)))
        return rc;
    if ((rc = refine_sobel_enqueue(ctx, s, s.im[0].img, h, w, w, (float *)s.grad_x.p, (float *)s.grad_y.p, nullptr)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(gx, s.grad_x.p, bytes, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(gy, s.grad_y.p, bytes, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_gn_refine_stereo(ebvo_ctx *ctx, const uint8_t *imgL, const uint8_t *imgR, int h, int w,
                                     ptrdiff_t strideL, ptrdiff_t strideR, const ebvo_edge *L, int nL,
                                     const double *lines, const int32_t *row_ptr, const double *cand_xy,
                                     const ebvo_gn_params *params, double *alpha, double *score, double *confidence,
                                     uint8_t *validity, int32_t *iters, double *refined_xy)
{
    if (!ctx || !imgL || !imgR || nL < 0 || !row_ptr || !params || (nL > 0 && (!L || !lines)) || params->max_iter < 1 ||
        !(params->tol >= 0) || !(params->huber_delta > 0))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (row_ptr[0] != 0)
        return EBVO_ERR_ARG;
    for (int i = 0; i < nL; ++i)
        if (row_ptr[i + 1] < row_ptr[i])
            return EBVO_ERR_ARG;
    const int64_t np = row_ptr[nL];
    if (np == 0)
        return EBVO_OK;
    if (!cand_xy || !alpha || !score || !confidence || !validity || !iters || !refined_xy)
        return EBVO_ERR_ARG;
    const size_t npz = (size_t)np;
    if ((rc = upload_image(ctx, s, 0, imgL, h, w, strideL)) || (rc = upload_image(ctx, s, 1, imgR, h, w, strideR)) ||
        (rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * (size_t)nL)) ||
        (rc = ebvo_grow(ctx, s, s.lines, sizeof(double) * 3 * (size_t)nL)) ||
        (rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * npz)) || (rc = ebvo_grow(ctx, s, s.gn_xy, sizeof(double) * 2 * npz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_out, sizeof(double) * 5 * npz)) || (rc = ebvo_grow(ctx, s, s.gn_valid, npz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_iters, sizeof(int32_t) * npz)))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, L, sizeof(ebvo_edge) * (size_t)nL, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.lines.p, lines, sizeof(double) * 3 * (size_t)nL, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.gn_xy.p, cand_xy, sizeof(double) * 2 * npz, hipMemcpyHostToDevice, st));
    double *out = (double *)s.gn_out.p;
    if ((rc = match_expand_rows_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, nL, np, (int32_t *)s.pair_left.p)) ||
        (rc = refine_gn_stereo_enqueue(ctx, s, s.im[0].img, s.im[1].img, nullptr, h, w, (const ebvo_edge *)s.scratch_b.p, nL,
                                       (const double *)s.lines.p, (const int32_t *)s.pair_left.p,
                                       (const double *)s.gn_xy.p, nullptr, nullptr, nullptr, np, params->max_iter,
                                       params->tol, params->huber_delta,
                                       out, out + npz, out + 2 * npz, (uint8_t *)s.gn_valid.p, (int32_t *)s.gn_iters.p,
                                       out + 3 * npz)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(alpha, out, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(score, out + npz, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(confidence, out + 2 * npz, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(refined_xy, out + 3 * npz, sizeof(double) * 2 * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(validity, s.gn_valid.p, npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(iters, s.gn_iters.p, sizeof(int32_t) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_edge_patches(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride,
                                 const ebvo_edge *edges, int n, float *patches)
{
    if (!ctx || !img || n < 0 || (n > 0 && (!edges || !patches)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (n == 0)
        return EBVO_OK;
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * (size_t)n)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_raw, sizeof(float) * 98 * (size_t)n)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, edges, sizeof(ebvo_edge) * (size_t)n, hipMemcpyHostToDevice, s.stream));
    if ((rc = match_patches_enqueue(ctx, s, s.im[0].img, h, w, w, (const ebvo_edge *)s.scratch_b.p, n, nullptr, 0,
                                    (float *)s.patches_raw.p, nullptr, nullptr)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(patches, s.patches_raw.p, sizeof(float) * 98 * (size_t)n, hipMemcpyDeviceToHost,
                                 s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
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
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
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
    if ((rc = upload_image(ctx, s, 0, imgL, h, w, strideL)))
        return rc;
    if ((rc = upload_image(ctx, s, 1, imgR, h, w, strideR)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * (size_t)nL)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.rc_edges, sizeof(ebvo_edge) * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)nL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_raw, sizeof(float) * 98 * (size_t)nL)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_norm, sizeof(float) * 98 * (size_t)nL)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_flag, 2 * (size_t)nL)))
        return rc;
    if (sims && (rc = ebvo_grow(ctx, s, s.sims, sizeof(double) * 4 * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.best, sizeof(double) * ((size_t)np + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.keep, (size_t)np + 1)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, L, sizeof(ebvo_edge) * (size_t)nL, hipMemcpyHostToDevice, s.stream));
    if (np)
        EBVO_HIP(ctx, hipMemcpyAsync(s.rc_edges.p, Rc, sizeof(ebvo_edge) * (size_t)np, hipMemcpyHostToDevice, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice,
                                 s.stream));
    if ((rc = match_patches_enqueue(ctx, s, s.im[0].img, h, w, w, (const ebvo_edge *)s.scratch_b.p, nL, nullptr, 0,
                                    (float *)s.patches_raw.p, (float *)s.patches_norm.p, (uint8_t *)s.patches_flag.p)))
        return rc;
    if ((rc = match_ncc_pairs_enqueue(ctx, s, s.im[1].img, h, w, w, (const ebvo_edge *)s.rc_edges.p,
                                      (const int32_t *)s.row_ptr.p, nL, np, (const float *)s.patches_norm.p,
                                      (const uint8_t *)s.patches_flag.p, thr, sims ? (double *)s.sims.p : nullptr,
                                      (double *)s.best.p, (uint8_t *)s.keep.p)))
        return rc;
    if (left_patches)
        EBVO_HIP(ctx, hipMemcpyAsync(left_patches, s.patches_raw.p, sizeof(float) * 98 * (size_t)nL,
                                     hipMemcpyDeviceToHost, s.stream));
    if (np)
    {
        if (sims)
            EBVO_HIP(ctx, hipMemcpyAsync(sims, s.sims.p, sizeof(double) * 4 * (size_t)np, hipMemcpyDeviceToHost, s.stream));
        if (best)
            EBVO_HIP(ctx, hipMemcpyAsync(best, s.best.p, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, s.stream));
        if (keep)
            EBVO_HIP(ctx, hipMemcpyAsync(keep, s.keep.p, (size_t)np, hipMemcpyDeviceToHost, s.stream));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
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
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    const size_t pb = sizeof(float) * 49 * (size_t)n;
    if ((rc = ebvo_grow(ctx, s, s.scratch_b, pb)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.scratch_c, pb)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.scratch_d, sizeof(double) * (size_t)n)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, A, pb, hipMemcpyHostToDevice, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_c.p, B, pb, hipMemcpyHostToDevice, s.stream));
    if ((rc = match_ncc_stored_enqueue(ctx, s, (const float *)s.scratch_b.p, (const float *)s.scratch_c.p, n,
                                       (double *)s.scratch_d.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(sim, s.scratch_d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
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
    std::vector<double> sv((size_t)n * 8);
    int rc = ebvo_ncc_patches(ctx, A.data(), B.data(), n * 8, sv.data());
    if (rc)
        return rc;
    for (int k = 0; k < n; ++k)
    {
        double sl = sv[(size_t)k * 8], sr = sv[(size_t)k * 8 + 4];
        for (int m = 1; m < 4; ++m)
        {
            if (sl < sv[(size_t)k * 8 + m]) sl = sv[(size_t)k * 8 + m];
            if (sr < sv[(size_t)k * 8 + 4 + m]) sr = sv[(size_t)k * 8 + 4 + m];
        }
        sim_left[k] = sl;
        sim_right[k] = sr;
        if (keep)
            keep[k] = (sl > thr && sr > thr) ? 1 : 0; // :452
    }
    return EBVO_OK;
}

// ------------------------------------------------------------------------------------------
// Device-resident pipeline.  A pair is ENQUEUED without any host synchronisation: every size lives in device
// memory, every launch is sized by a capacity, the last kernel gathers the counts, one async copy brings them to
// pinned host memory.  ebvo_stereo_wait synchronises the slot's stream, and only if the candidate list did not
// fit the pair-indexed buffers does it grow them and re-enqueue the matching half.
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

extern "C" void ebvo_finalize_default_params(ebvo_finalize_params *p)
{
    if (!p)
        return;
    memset(p, 0, sizeof *p);
    p->bnb_ratio = EBVO_BNB_NCC;
    p->ncc_thr = EBVO_NCC_THRESH;
    ebvo_gn_default_params(&p->gn);
    p->use_sift = 0;
    p->sift_thr = EBVO_SIFT_THRESHOLD;
    p->bnb_sift = EBVO_BNB_SIFT;
}

static int get_slot(ebvo_ctx *ctx, int slot, Slot **out)
{
    if (!ctx || slot < 0 || slot >= (int)ctx->slots.size())
        return EBVO_ERR_ARG;
    *out = ctx->slots[(size_t)slot];
    return EBVO_OK;
}

extern "C" int ebvo_stereo_set_slots(ebvo_ctx *ctx, int n_slots)
{
    if (!ctx || n_slots < 1 || n_slots > 64)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->slots.size() < n_slots)
    {
        Slot *s = nullptr;
        int rc = slot_create(ctx, &s);
        if (rc)
            return rc;
        ctx->slots.push_back(s);
    }
    return EBVO_OK;
}

extern "C" int ebvo_stereo_upload_slot(ebvo_ctx *ctx, int slot, const uint8_t *img_left, const uint8_t *img_right,
                                       int h, int w, ptrdiff_t stride_left, ptrdiff_t stride_right)
{
    Slot *sp;
    int rc;
    if (!img_left || !img_right || (rc = get_slot(ctx, slot, &sp)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = check_size(ctx, h, w)))
        return rc;
    Slot &s = *sp;
    if (s.in_flight || s.fin_in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    s.have_pair = s.have_run = s.have_refined = s.have_final = false; // results of the previous pair are gone
    s.have_push = false;
    if (slot == 0)
        ctx->sw_tag[0] = ctx->sw_tag[1] = 0; // ... and so are the TOED results of the resident stage-wise calls
    s.tq_n = -1;
    s.tq_final.n = -1;
    s.sift_left_valid = false;
    if ((rc = drain_fetch(ctx, s))) // a result copy of the previous pair still reads the buffers / an asynchronous upload writes them
        return rc;
    s.fetch_pending = false;
    s.fetch_what = 0;
    s.undist_pair = ctx->undist_on; // the pair goes to the raw buffers; submit undistorts it into img
    s.toed_strict_override = false;
    s.pull = false;
    if ((rc = upload_image(ctx, s, 0, img_left, h, w, stride_left, s.undist_pair)))
        return rc;
    if ((rc = upload_image(ctx, s, 1, img_right, h, w, stride_right, s.undist_pair)))
        return rc;
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    s.cur_h = h;
    s.cur_w = w;
    s.have_pair = true;
    return EBVO_OK;
}

// The same upload without blocking the calling thread: the two copies go to the context's UPLOAD stream (its own hardware
// queue: they never sit in front of a pair's kernels), an event marks their end, and the slot's next submission makes its
// stream wait for that event -- satisfied long before when the caller uploads a slot a frame ahead of submitting it
// (src/Pipeline.cpp:77-99 reads frame k + 1 while frame k is matched).  Asynchronous only for page-locked sources
// (ebvo_host_register / hipHostMalloc); pageable images are copied into the slot's page-locked staging before the call returns.
extern "C" int ebvo_stereo_upload_async(ebvo_ctx *ctx, int slot, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                                        ptrdiff_t stride_left, ptrdiff_t stride_right)
{
    Slot *sp;
    int rc;
    if (!img_left || !img_right || (rc = get_slot(ctx, slot, &sp)))
        return EBVO_ERR_ARG;
    if ((rc = check_size(ctx, h, w)))
        return rc;
    Slot &s = *sp;
    if (s.in_flight || s.fin_in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    s.have_pair = s.have_run = s.have_refined = s.have_final = false;
    if (slot == 0)
        ctx->sw_tag[0] = ctx->sw_tag[1] = 0;
    s.tq_n = -1;
    s.tq_final.n = -1;
    s.sift_left_valid = false;
    s.pull = false;
    if (stride_left < w || stride_right < w)
        return EBVO_ERR_ARG;
    if (!ctx->ingest_stream) // (the pull form makes no runtime call at all)
    {
        // pull form: both images entirely in page-locked memory the device can address?  Then the pair's own chain reads them
        // (its first kernel; the pointers go through the slot's mailbox) -- no copy engine, no second stream, no event: nothing
        // that could queue behind another pair's kernels.  Page-locked = inside a range registered with ebvo_host_register.
        const uint8_t *imgs[2] = {img_left, img_right};
        const ptrdiff_t strides[2] = {stride_left, stride_right};
        const uint8_t *dev[2] = {nullptr, nullptr};
        bool pinned = true;
        for (int k = 0; k < 2 && pinned; ++k)
        {
            const size_t span = (size_t)(h - 1) * (size_t)strides[k] + (size_t)w;
            pinned = false;
            for (const ebvo_ctx::HostRange &r : ctx->host_ranges)
                if (imgs[k] >= r.host && imgs[k] + span <= r.host + r.bytes)
                {
                    dev[k] = r.dev + (imgs[k] - r.host);
                    pinned = true;
                    break;
                }
        }
        if (pinned)
        {
            if ((rc = drain_stream_upload(ctx, s)))
                return rc;
            for (int k = 0; k < 2; ++k)
            {
                s.h_mail->src[k] = dev[k];
                s.h_mail->stride[k] = (long long)strides[k];
            }
            s.pull = true;
            ++ctx->pull_uploads;
            s.undist_pair = ctx->undist_on;
            s.toed_strict_override = false;
            s.cur_h = h;
            s.cur_w = w;
            s.have_pair = true;
            return EBVO_OK;
        }
    }
    ++ctx->stream_uploads;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->upload_stream)
        EBVO_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    // ordering on the device, not on the host: result copies of the previous pair (copy stream) still read the edge lists --
    // not the images -- so only a previous asynchronous upload of this slot (same stream: ordered) and the slot's own stream
    // (later stages of the previous pair sample the images) matter
    // The copies below wait ON THE DEVICE for the slot's earlier work, so the runtime cannot read a pageable source while this
    // call runs: it would read it whenever the copy gets its turn -- after the caller has, rightly, freed the images (seen as
    // a GPU memory fault at a host heap address in a later test, once in five full test runs).  Images that are not inside a
    // registered range are therefore copied into the slot's own page-locked staging first (a host memcpy of 2 h w bytes); the
    // previous asynchronous upload of the slot has left the staging when drain_stream_upload returns.
    const uint8_t *src[2] = {img_left, img_right};
    ptrdiff_t sstride[2] = {stride_left, stride_right};
    {
        const size_t npx = (size_t)h * w;
        bool staged = false;
        for (int k = 0; k < 2; ++k)
        {
            const size_t span = (size_t)(h - 1) * (size_t)sstride[k] + (size_t)w;
            bool pinned = false;
            for (const ebvo_ctx::HostRange &r : ctx->host_ranges)
                if (src[k] >= r.host && src[k] + span <= r.host + r.bytes)
                    pinned = true;
            if (pinned)
                continue;
            if (!staged)
            {
                if ((rc = drain_stream_upload(ctx, s)))
                    return rc;
                if (s.h_up_bytes < 2 * npx)
                {
                    if (s.h_up)
                        (void)hipHostFree(s.h_up);
                    s.h_up = nullptr;
                    s.h_up_bytes = 0;
                    if (hipHostMalloc(reinterpret_cast<void **>(&s.h_up), 2 * npx) != hipSuccess)
                    {
                        (void)hipGetLastError();
                        ctx->last_error = "hipHostMalloc failed (staging of ebvo_stereo_upload_async)";
                        return EBVO_ERR_NOMEM;
                    }
                    s.h_up_bytes = 2 * npx;
                }
                staged = true;
            }
            uint8_t *stage = s.h_up + (size_t)k * npx;
            if (sstride[k] == (ptrdiff_t)w)
                memcpy(stage, src[k], npx);
            else
                for (int y = 0; y < h; ++y)
                    memcpy(stage + (size_t)y * w, src[k] + (size_t)y * sstride[k], (size_t)w);
            src[k] = stage;
            sstride[k] = w;
        }
    }
    EBVO_HIP(ctx, hipEventRecord(s.ev_upload, s.own_stream));
    EBVO_HIP(ctx, hipStreamWaitEvent(ctx->upload_stream, s.ev_upload, 0));
    s.undist_pair = ctx->undist_on;
    s.toed_strict_override = false;
    if ((rc = upload_image(ctx, s, 0, src[0], h, w, sstride[0], s.undist_pair, ctx->upload_stream)) ||
        (rc = upload_image(ctx, s, 1, src[1], h, w, sstride[1], s.undist_pair, ctx->upload_stream)))
        return rc;
    EBVO_HIP(ctx, hipEventRecord(s.ev_upload, ctx->upload_stream));
    s.upload_pending = true;
    s.cur_h = h;
    s.cur_w = w;
    s.have_pair = true;
    return EBVO_OK;
}

// Page-locks caller memory (a frame ring, say) so that uploads from it are asynchronous DMA and result copies into it need no
// staging.  Thin wrappers: the host side of the boundary needs no HIP header.
extern "C" int ebvo_host_register(ebvo_ctx *ctx, void *p, size_t bytes)
{
    if (!ctx || !p || !bytes)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    ebvo_ctx::HostRange r;
    r.host = static_cast<const uint8_t *>(p);
    r.bytes = bytes;
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipSuccess)
        r.ours = true;
    else
    {
        // memory that is page-locked already (hipHostMalloc, an earlier registration) is accepted as it is
        (void)hipGetLastError();
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, p) != hipSuccess || a.type != hipMemoryTypeHost)
        {
            (void)hipGetLastError();
            return ebvo_fail_hip(ctx, e, "hipHostRegister", __FILE__, __LINE__);
        }
    }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d)
    {
        (void)hipGetLastError();
        if (r.ours)
            (void)hipHostUnregister(p);
        ctx->last_error = "hipHostGetDevicePointer failed for the registered range";
        return EBVO_ERR_HIP;
    }
    r.dev = static_cast<const uint8_t *>(d);
    ctx->host_ranges.push_back(r);
    return EBVO_OK;
}

// how many ebvo_stereo_upload_async calls took the pull form / the upload stream so far (a frame loop checks that its ring is
// really registered: a pageable source silently takes the slow form)
extern "C" int ebvo_ingest_stats(const ebvo_ctx *ctx, int64_t out[2])
{
    if (!ctx || !out)
        return EBVO_ERR_ARG;
    out[0] = ctx->pull_uploads;
    out[1] = ctx->stream_uploads;
    return EBVO_OK;
}

extern "C" int ebvo_host_unregister(ebvo_ctx *ctx, void *p)
{
    if (!ctx || !p)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    // a slot whose chain pulls its images from page-locked caller memory must not be submitted again with them: it needs a new
    // upload first (conservative: the library does not know which registration a mailbox points into)
    for (Slot *sl : ctx->slots)
        if (sl->pull)
        {
            if (sl->in_flight)
                EBVO_HIP(ctx, hipEventSynchronize(sl->ev_done));
            sl->pull = false;
            if (!sl->in_flight)
                sl->have_pair = false;
        }
    bool ours = false, found = false;
    for (size_t k = 0; k < ctx->host_ranges.size(); ++k)
        if (ctx->host_ranges[k].host == p)
        {
            ours = ctx->host_ranges[k].ours;
            found = true;
            ctx->host_ranges.erase(ctx->host_ranges.begin() + (ptrdiff_t)k);
            break;
        }
    if (!found)
        return EBVO_ERR_ARG;
    if (ours)
        EBVO_HIP(ctx, hipHostUnregister(p));
    return EBVO_OK;
}

extern "C" int ebvo_stereo_upload(ebvo_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                                  ptrdiff_t stride_left, ptrdiff_t stride_right)
{
    return ebvo_stereo_upload_slot(ctx, 0, img_left, img_right, h, w, stride_left, stride_right);
}

// capacity-sized buffers of the pipeline
static int ensure_pipeline_buffers(ebvo_ctx *ctx, Slot &s, int64_t cap_pairs)
{
    const size_t ce = (size_t)ctx->cap_edges;
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.lines, sizeof(double) * 3 * ce)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_raw, sizeof(float) * 98 * ce)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_norm, sizeof(float) * 98 * ce)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_flag, 2 * ce)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.patches_norm_r, match_right_bank_bytes(ctx->cap_edges))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.sincos, sizeof(double) * 2 * ce)))
        return rc;
    const size_t cp = (size_t)cap_pairs;
    if ((rc = ebvo_grow(ctx, s, s.col_idx, sizeof(int32_t) * cp)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * cp)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.sims, sizeof(double) * 4 * cp)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.best, sizeof(double) * cp)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.keep, cp)))
        return rc;
    s.cap_pairs = cap_pairs;
    s.pipe_cap = cap_pairs;
    return EBVO_OK;
}

// lines -> candidates (count, scan, fill) -> right patch bank -> LDS-tiled NCC -> result record; no host synchronisation
static int enqueue_matching(ebvo_ctx *ctx, Slot &s, int toed_mode = -1)
{
    const ebvo_stereo_params &p = s.params;
    const int h = s.cur_h, w = s.cur_w, ce = ctx->cap_edges;
    const int32_t *d_nL = s.im[0].counts + 1, *d_nR = s.im[1].counts + 1;
    int rc;
    // lines, boxes, sin / cos and row-pair images in one launch (round 4: four launches of ~5 us each before)
    const bool prep = !ctx->no_prep;
    const int stop = ctx->stop_stage; // (developer key 16: the counts of the last whole run stay in the pair's record)
    if (stop && stop <= 7)
        return EBVO_OK;
    if (prep ? (rc = match_prep_enqueue(ctx, s, h, w, ce))
             : (rc = match_lines_enqueue(ctx, s, s.d_F, s.im[0].edges, 0, d_nL, ce, (double *)s.lines.p)))
        return rc;
    if (stop == 8)
        return EBVO_OK;
    if ((rc = match_candidates_enqueue(ctx, s, s.im[0].edges, 0, d_nL, s.im[1].edges, 0, d_nR, ce,
                                       (const double *)s.lines.p, p.epi_thr, p.max_disp, p.orient_thr_deg, p.stage_mask,
                                       true, prep)))
        return rc;
    if (stop && stop <= 11)
        return EBVO_OK;
    if ((rc = match_ncc_resident_enqueue(ctx, s, h, w, ce, p.ncc_thr, 0, !(p.reserved & EBVO_PAIR_NO_SIMS), prep)))
        return rc;
    if (stop)
        return EBVO_OK;
    // a hybrid TOED run reports candidate lists that did not fit through the result record (bit 1, value 2, of `overflow`)
    if ((rc = match_pair_result_enqueue(ctx, s, (toed_mode < 0 ? ctx->toed_mode : toed_mode) == EBVO_TOED_HYBRID ? ce : 0)))
        return rc;
    return EBVO_OK; // the caller records s.ev_done behind it (outside a stream capture)
}

// undistortion (if the pair was uploaded raw) + TOED of both images + matching: every launch of one pair
// EBVO_PAIR_PUSH: the compact results of a pair written by the device into page-locked host memory, as the last kernel of the
// pair's chain (same arrays as ebvo_stereo_fetch_compact_begin; the sizes are read on the device).  Coalesced 16-byte stores
// over PCIe; the kernel holds a handful of CUs while the lanes' other pairs compute.
namespace
{
// where the eight arrays lie in a packed result block, from the ACTUAL counts (the device reads them where the host computes
// the same offsets after ebvo_stereo_wait): xyL, xyR, thL, thR, row_ptr, col_idx, best, keep bits; 256-byte aligned
struct PackLayout
{
    size_t off[8], total;
};
__host__ __device__ inline PackLayout pack_layout(size_t nL, size_t nR, size_t np, bool theta)
{
    const size_t sizes[8] = {16 * nL, 16 * nR, theta ? 8 * nL : 0, theta ? 8 * nR : 0, 4 * (nL + 1), 4 * np, 8 * np, 8 * ((np + 63) >> 6)};
    PackLayout l;
    size_t t = 0;
    for (int k = 0; k < 8; ++k)
    {
        l.off[k] = t;
        t += (sizes[k] + 255) & ~(size_t)255;
    }
    l.total = t;
    return l;
}

__global__ __launch_bounds__(256) void push_results_kernel(const ebvo_edge *__restrict__ L, const ebvo_edge *__restrict__ R,
                                                           const int32_t *__restrict__ nLp, const int32_t *__restrict__ nRp, int cap_edges,
                                                           const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
                                                           const double *__restrict__ best, const uint8_t *__restrict__ keep,
                                                           int64_t cap_pairs, char *__restrict__ base, int theta)
{
    const int nL = min(*nLp, cap_edges), nR = min(*nRp, cap_edges);
    int64_t np = nL > 0 ? (int64_t)row_ptr[nL] : 0;
    np = np < cap_pairs ? np : cap_pairs;
    const PackLayout lay = pack_layout((size_t)nL, (size_t)nR, (size_t)np, theta != 0);
    double2 *xyL = (double2 *)(base + lay.off[0]), *xyR = (double2 *)(base + lay.off[1]);
    double *thL = theta ? (double *)(base + lay.off[2]) : nullptr, *thR = theta ? (double *)(base + lay.off[3]) : nullptr;
    int32_t *o_rp = (int32_t *)(base + lay.off[4]), *o_ci = (int32_t *)(base + lay.off[5]);
    double *o_best = (double *)(base + lay.off[6]);
    uint32_t *o_bits = (uint32_t *)(base + lay.off[7]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nL; i += stride)
    {
        const ebvo_edge e = L[i];
        xyL[i] = make_double2(e.x, e.y);
        if (thL)
            thL[i] = e.theta;
    }
    for (int64_t i = t0; i < nR; i += stride)
    {
        const ebvo_edge e = R[i];
        xyR[i] = make_double2(e.x, e.y);
        if (thR)
            thR[i] = e.theta;
    }
    for (int64_t i = t0; i <= nL; i += stride)
        o_rp[i] = row_ptr[i];
    // four CSR entries / two scores per thread: 16-byte stores
    const int64_t n4 = np >> 2;
    for (int64_t i = t0; i < n4; i += stride)
        reinterpret_cast<int4 *>(o_ci)[i] = reinterpret_cast<const int4 *>(col_idx)[i];
    for (int64_t i = (n4 << 2) + t0; i < np; i += stride)
        o_ci[i] = col_idx[i];
    const int64_t n2 = np >> 1;
    for (int64_t i = t0; i < n2; i += stride)
        reinterpret_cast<double2 *>(o_best)[i] = reinterpret_cast<const double2 *>(best)[i];
    if ((np & 1) && t0 == 0)
        o_best[np - 1] = best[np - 1];
    const int64_t groups = (np + 63) >> 6, wave = t0 >> 6, waves = stride >> 6;
    const int lane = threadIdx.x & 63;
    for (int64_t g = wave; g < groups; g += waves)
    {
        const int64_t k = (g << 6) + lane;
        const unsigned long long m = __ballot(k < np && keep[k] != 0);
        if (lane == 0)
        {
            o_bits[2 * g] = (uint32_t)m;
            o_bits[2 * g + 1] = (uint32_t)(m >> 32);
        }
    }
}
} // namespace

// (re)sizes the destination of the chain's pack kernel for the capacities the chain is enqueued with: the slot's page-locked
// arena (EBVO_PAIR_PUSH) or its device staging (EBVO_PAIR_PACK).  A re-allocation bumps buf_gen (the address is an argument of
// a captured launch).
static int ensure_push_arena(ebvo_ctx *ctx, Slot &s)
{
    const bool theta = s.params.reserved & EBVO_PAIR_PUSH_THETA;
    const size_t total = pack_layout((size_t)ctx->cap_edges, (size_t)ctx->cap_edges, (size_t)s.cap_pairs, theta).total;
    if (s.params.reserved & EBVO_PAIR_PUSH)
    {
        if (total > s.push_bytes)
        {
            if (s.h_push)
                (void)hipHostFree(s.h_push);
            s.h_push = s.d_push = nullptr;
            s.push_bytes = 0;
            if (hipHostMalloc(&s.h_push, total) != hipSuccess)
            {
                (void)hipGetLastError();
                ctx->last_error = "hipHostMalloc failed (page-locked result arena of EBVO_PAIR_PUSH)";
                return EBVO_ERR_NOMEM;
            }
            EBVO_HIP(ctx, hipHostGetDevicePointer(&s.d_push, s.h_push, 0));
            s.push_bytes = total;
            ++s.buf_gen;
        }
        return EBVO_OK;
    }
    if (total > s.fetch_pack.bytes && s.fetch_pending) // copies out of the old staging must be over before it is replaced
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    return ebvo_grow(ctx, s, s.fetch_pack, total);
}

// the images of a pair, read by the GPU from the caller's page-locked memory (ebvo_stereo_upload_async): 16 bytes per thread
namespace
{
__global__ __launch_bounds__(256) void pull_images_kernel(const Slot::PullMail *__restrict__ mail, uint8_t *__restrict__ d0,
                                                          uint8_t *__restrict__ d1, int h, int w)
{
    const uint8_t *__restrict__ src = mail->src[blockIdx.y];
    const long long stride = mail->stride[blockIdx.y];
    uint8_t *__restrict__ dst = blockIdx.y ? d1 : d0;
    const int per_row = (w + 15) >> 4; // 16-byte pieces per row (the last one may be short)
    const long long pieces = (long long)h * per_row;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < pieces; t += (long long)gridDim.x * blockDim.x)
    {
        const int y = (int)(t / per_row), c = (int)(t - (long long)y * per_row) << 4;
        const uint8_t *sp = src + (long long)y * stride + c;
        uint8_t *dp = dst + (size_t)y * w + c;
        if (c + 16 <= w)
        {
            uint4 v;
            __builtin_memcpy(&v, sp, 16);
            __builtin_memcpy(dp, &v, 16);
        }
        else
            for (int k = 0; c + k < w; ++k)
                dp[k] = sp[k];
    }
}
} // namespace

static int enqueue_push(ebvo_ctx *ctx, Slot &s)
{
    if (!(s.params.reserved & (EBVO_PAIR_PUSH | EBVO_PAIR_PACK)))
        return EBVO_OK;
    char *base = static_cast<char *>((s.params.reserved & EBVO_PAIR_PUSH) ? s.d_push : s.fetch_pack.p);
    hipLaunchKernelGGL(push_results_kernel, dim3(64), dim3(256), 0, s.stream, (const ebvo_edge *)s.im[0].edges,
                       (const ebvo_edge *)s.im[1].edges, (const int32_t *)(s.im[0].counts + 1), (const int32_t *)(s.im[1].counts + 1),
                       ctx->cap_edges, (const int32_t *)s.row_ptr.p, (const int32_t *)s.col_idx.p, (const double *)s.best.p,
                       (const uint8_t *)s.keep.p, s.cap_pairs, base, (s.params.reserved & EBVO_PAIR_PUSH_THETA) ? 1 : 0);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

static int enqueue_pair_chain(ebvo_ctx *ctx, Slot &s)
{
    int rc;
    if (s.pull)
    {
        hipLaunchKernelGGL(pull_images_kernel, dim3(64, 2), dim3(256), 0, s.stream, (const Slot::PullMail *)s.d_mail,
                           s.undist_pair ? s.im[0].raw : s.im[0].img, s.undist_pair ? s.im[1].raw : s.im[1].img, s.cur_h, s.cur_w);
        EBVO_HIP(ctx, hipGetLastError());
    }
    if (s.undist_pair) // cv::undistort of both raw images (src/Pipeline.cpp:78-79); TOED runs on the result (:93, :97)
    {
        const ebvo_undistort_params &u = ctx->undist;
        if ((rc = refine_undistort_enqueue(ctx, s, s.im[0].raw, s.cur_w, s.cur_h, s.cur_w, u.K_left, u.dist_left, u.n_dist,
                                           s.im[0].undist_xs, s.im[0].img, s.cur_w)) ||
            (rc = refine_undistort_enqueue(ctx, s, s.im[1].raw, s.cur_w, s.cur_h, s.cur_w, u.K_right, u.dist_right, u.n_dist,
                                           s.im[1].undist_xs, s.im[1].img, s.cur_w)))
            return rc;
    }
    // (no subpix_edge_pts_final in the resident pipeline: nothing downstream reads it, 8 MB of scattered stores per pair)
    const int mode = s.toed_strict_override ? EBVO_TOED_STRICT : -1;
    if ((rc = toed_enqueue(ctx, s, 2, s.cur_h, s.cur_w, nullptr, nullptr, nullptr, mode, false)))
        return rc;
    if ((rc = enqueue_matching(ctx, s, mode)))
        return rc;
    return enqueue_push(ctx, s);
}

static void pair_graph_drop(Slot &s)
{
    if (s.pair_graph)
        (void)hipGraphExecDestroy(s.pair_graph);
    s.pair_graph = nullptr;
}

// The chain of one pair is 18 launches whose arguments depend only on PairGraphKey and on the slot's buffers (the sizes
// of everything live on the device): the second submission with an unchanged key captures it, later ones launch the graph
// (measured, tools/ubench/graph_launch.hip: 7-20 us of host time for a 30-kernel chain against 86 us of direct launches,
// and ~2 us instead of ~2.9 us between dependent kernels on the device).  Anything unusual -- profiling markers, a key
// that changed, a capture that fails -- takes the direct path; `first` reports that to the caller for its statistics.
static int submit_pair_chain(ebvo_ctx *ctx, Slot &s)
{
    PairGraphKey key;
    memset(&key, 0, sizeof key);
    key.h = s.cur_h;
    key.w = s.cur_w;
    key.toed_mode = s.toed_strict_override ? EBVO_TOED_STRICT : ctx->toed_mode;
    key.undist = (s.undist_pair ? 1 : 0) | (s.pull ? 2 : 0);
    key.cap_pairs = s.cap_pairs;
    key.gen = ctx->graph_gen + (s.buf_gen << 20);
    key.epi_thr = s.params.epi_thr;
    key.max_disp = s.params.max_disp;
    key.orient_thr_deg = s.params.orient_thr_deg;
    key.ncc_thr = s.params.ncc_thr;
    key.stage_mask = s.params.stage_mask;
    key.flags = s.params.reserved;
    const bool same = memcmp(&key, &s.pair_key, sizeof key) == 0;
    const bool eligible = ctx->use_graphs && !s.pair_graph_off && !ctx->prof && !s.prof_now;
    if (!same)
    {
        pair_graph_drop(s);
        s.pair_key = key;
        s.pair_key_warm = false;
    }
    int rc;
    if (eligible && s.pair_graph)
    {
        EBVO_HIP(ctx, hipGraphLaunch(s.pair_graph, s.stream));
        ++ctx->graph_launches;
        return EBVO_OK;
    }
    if (eligible && s.pair_key_warm)
    {
        // capture: thread-local mode, so that other host threads keep using their own streams meanwhile
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess)
        {
            const uint64_t gen_before = s.buf_gen;
            rc = enqueue_pair_chain(ctx, s);
            e = hipStreamEndCapture(s.stream, &g); // always: the stream must leave capture mode
            if (rc == EBVO_OK && e == hipSuccess && g && s.buf_gen == gen_before)
                e = hipGraphInstantiate(&s.pair_graph, g, nullptr, nullptr, 0);
            else if (e == hipSuccess)
                e = hipErrorUnknown;
            if (g)
                (void)hipGraphDestroy(g);
        }
        if (e == hipSuccess && s.pair_graph)
        {
            EBVO_HIP(ctx, hipGraphLaunch(s.pair_graph, s.stream));
            ++ctx->graph_launches;
            return EBVO_OK;
        }
        (void)hipGetLastError();
        pair_graph_drop(s);
        s.pair_graph_off = true; // and fall through to the direct launches
    }
    if ((rc = enqueue_pair_chain(ctx, s)))
        return rc;
    if (ctx->graph_gen + (s.buf_gen << 20) == s.pair_key.gen)
        s.pair_key_warm = true; // nothing was (re)allocated on the way: the next submission may capture
    return EBVO_OK;
}

extern "C" int ebvo_stereo_submit(ebvo_ctx *ctx, int slot, const ebvo_stereo_params *p)
{
    Slot *sp;
    if (!p || (p->stage_mask & ~EBVO_STAGE_ALL) || p->stage_mask == 0 || (p->reserved & ~(EBVO_PAIR_NO_SIMS | EBVO_PAIR_PUSH | EBVO_PAIR_PUSH_THETA | EBVO_PAIR_PACK)) ||
        ((p->reserved & EBVO_PAIR_PUSH) && (p->reserved & EBVO_PAIR_PACK)) || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_pair || s.in_flight || s.fin_in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    s.have_run = s.have_refined = s.have_final = false;
    const bool fetch_was_pending = s.fetch_pending;
    s.fetch_pending = false; // stream order (or the event below): the kernels run after any copy still enqueued
    s.fetch_what = 0;
    s.params = *p;
    s.have_sims = !(p->reserved & EBVO_PAIR_NO_SIMS);
    s.prof_now = ctx->prof && (ctx->prof_submits++ % ctx->prof_every == 0);
    // lanes in use: none up to three slots (one stream each), never as many lanes as slots (a lane must be able to hold
    // a queued pair behind the running one: four slots on four streams is the 2500 pairs/s dip)
    const int n_lanes = ((int)ctx->slots.size() >= 4 && ctx->lanes > 0)
                            ? (ctx->lanes < (int)ctx->slots.size() - 1 ? ctx->lanes : (int)ctx->slots.size() - 1)
                            : 0;
    if (n_lanes > 0)
    {
        // Four or more pairs in flight: their KERNELS are dealt round-robin to a few streams of the context ("lanes")
        // instead of one stream per slot.  Measured (tools/gpu_streams_sweep.py, tools/gpu_lanes_sweep.py, KITTI pair): one
        // stream per slot gives 2010 / 2540 / 2800 pairs/s for 1 / 2 / 3 slots and then DROPS (2500 at 4 slots, 2700 at 5;
        // GPU_MAX_HW_QUEUES and stream priorities do not change that); three lanes give 2850-2880 for 4 ... 12 slots, four
        // lanes 2920-2945 for 5 ... 16 slots, five and six lanes less.  Everything else a slot does (uploads, fetch copies,
        // the later stages) stays on its own stream: ebvo_stereo_wait hands the slot back to it once the host has seen
        // the pair complete, and the only work a slot can leave pending there -- the copies of ebvo_stereo_fetch_begin --
        // carries an event.
        while ((int)ctx->lane_streams.size() < n_lanes)
        {
            hipStream_t st = nullptr;
            EBVO_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            ctx->lane_streams.push_back(st);
        }
        hipStream_t lane = ctx->lane_streams[ctx->submit_seq++ % (uint64_t)n_lanes];
        s.stream = lane;
    }
    if (fetch_was_pending) // result copies of the previous pair (on the copy stream) still read the buffers
        EBVO_HIP(ctx, hipStreamWaitEvent(s.stream, s.ev_rebind, 0));
    if (s.upload_pending) // the pair's images arrive on the upload stream (ebvo_stereo_upload_async)
        EBVO_HIP(ctx, hipStreamWaitEvent(s.stream, s.ev_upload, 0));
    s.upload_pending = false;
    struct Unbind // a submission that fails leaves the slot on its own stream
    {
        Slot &s;
        ~Unbind()
        {
            if (!s.in_flight)
                s.stream = s.own_stream;
        }
    } unbind{s};
    {
        int64_t want = s.pipe_cap > 0 ? s.pipe_cap : 8 * (int64_t)ctx->cap_edges;
        if (want < 4096)
            want = 4096;
        if ((rc = ensure_pipeline_buffers(ctx, s, want)))
            return rc;
    }
    s.have_push = s.have_pack = false;
    if ((s.params.reserved & (EBVO_PAIR_PUSH | EBVO_PAIR_PACK)) && (rc = ensure_push_arena(ctx, s)))
        return rc;
    // the fundamental matrix goes up only when it differs from what the device holds (a sequence has ONE; the copy of 72 bytes
    // from pageable memory is a blit kernel of ~5 us at the head of the chain otherwise)
    if (!s.F_dev_valid || memcmp(s.F_dev, s.params.F21, sizeof s.F_dev) != 0)
    {
        EBVO_HIP(ctx, hipMemcpyAsync(s.d_F, s.params.F21, sizeof(double) * 9, hipMemcpyHostToDevice, s.stream));
        memcpy(s.F_dev, s.params.F21, sizeof s.F_dev);
        s.F_dev_valid = true;
    }
    if ((rc = submit_pair_chain(ctx, s)))
        return rc;
    EBVO_HIP(ctx, hipEventRecord(s.ev_done, s.stream));
    s.in_flight = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_wait(ebvo_ctx *ctx, int slot, ebvo_stereo_counts *counts)
{
    Slot *sp;
    if (!counts || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc = EBVO_OK;
    struct Restore
    {
        Slot &s;
        ~Restore()
        {
            s.prof_now = true; // host-buffer calls on this slot are always bracketed
            if (!s.in_flight)
                s.stream = s.own_stream; // the pair ran on a lane (ebvo_stereo_submit); the host has seen it complete
        }
    } restore{s};
    bool have_result = false;
    const int max_attempts = ctx->wait_attempts > 0 ? ctx->wait_attempts : 4;
    for (int attempt = 0; attempt < max_attempts; ++attempt)
    {
        // the event behind this pair's last kernel, not the stream: a lane may already hold the next pair of another slot
        hipError_t e = hipEventSynchronize(s.ev_done);
        if (e != hipSuccess)
        {
            s.in_flight = false;
            return ebvo_fail_hip(ctx, e, "hipEventSynchronize", __FILE__, __LINE__);
        }
        const PairResult r = *s.h_result;
        if (r.overflow & 2)
        {
            // the hybrid screen flagged more candidates than fit (toed_compact_phase_kernel): the whole pair again, strict
            ++ctx->toed_fallbacks;
            s.toed_strict_override = true; // remembered until the next upload: these images go strict at once from now on
            if ((rc = toed_enqueue(ctx, s, 2, s.cur_h, s.cur_w, nullptr, nullptr, nullptr, EBVO_TOED_STRICT, false)) ||
                (rc = enqueue_matching(ctx, s, EBVO_TOED_STRICT)) || (rc = enqueue_push(ctx, s)))
            {
                s.in_flight = false;
                return rc;
            }
            EBVO_HIP(ctx, hipEventRecord(s.ev_done, s.stream));
            --attempt; // not a regrowth: the fallback does not use up one of the attempts (it happens at most once per wait)
            continue;
        }
        if (r.n_total_left > ctx->cap_edges || r.n_total_right > ctx->cap_edges)
        {
            s.in_flight = false;
            ctx->last_error = "internal edge capacity exceeded";
            return EBVO_ERR_CAPACITY;
        }
        if (!r.overflow && !(ctx->force_overflow > 0 && ctx->force_overflow-- > 0))
        {
            s.result = r;
            have_result = true;
            break;
        }
        if (r.n_pairs > 0x7fffffffll)
        {
            s.in_flight = false;
            ctx->last_error = "candidate list exceeds 2^31-1 pairs";
            return EBVO_ERR_CAPACITY;
        }
        // more candidates than the buffers hold: grow them and redo the matching half (TOED results are intact)
        int64_t want = r.n_pairs + r.n_pairs / 4 + 1024;
        if ((rc = ensure_pipeline_buffers(ctx, s, want)) ||
            ((s.params.reserved & (EBVO_PAIR_PUSH | EBVO_PAIR_PACK)) && (rc = ensure_push_arena(ctx, s))) ||
            (rc = enqueue_matching(ctx, s)) || (rc = enqueue_push(ctx, s)))
        {
            s.in_flight = false;
            return rc;
        }
        EBVO_HIP(ctx, hipEventRecord(s.ev_done, s.stream));
    }
    if (!have_result)
    {
        // every attempt reported an overflow: the last re-enqueued matching half is still running; never publish a
        // stale s.result
        (void)hipEventSynchronize(s.ev_done);
        s.in_flight = false;
        ctx->last_error = "candidate buffers still too small after regrowing (ebvo_stereo_wait gave up)";
        return EBVO_ERR_CAPACITY;
    }
    s.in_flight = false;
    s.im[0].n_kept = s.result.n_left;
    s.im[0].n_total = s.result.n_total_left;
    s.im[1].n_kept = s.result.n_right;
    s.im[1].n_total = s.result.n_total_right;
    counts->n_left = s.result.n_left;
    counts->n_right = s.result.n_right;
    counts->n_total_left = s.result.n_total_left;
    counts->n_total_right = s.result.n_total_right;
    counts->n_pairs = s.result.n_pairs;
    counts->n_matches = s.result.n_matches;
    s.have_run = true;
    s.have_push = (s.params.reserved & EBVO_PAIR_PUSH) != 0;
    s.have_pack = (s.params.reserved & EBVO_PAIR_PACK) != 0;
    return EBVO_OK;
}

static void fill_compact_view(ebvo_stereo_compact_view *view, const char *b, const PackLayout &lay, const PairResult &r, bool theta)
{
    memset(view, 0, sizeof *view);
    view->n_left = r.n_left;
    view->n_right = r.n_right;
    view->n_pairs = r.n_pairs;
    view->n_matches = r.n_matches;
    view->left_xy = reinterpret_cast<const double *>(b + lay.off[0]);
    view->right_xy = reinterpret_cast<const double *>(b + lay.off[1]);
    if (theta)
    {
        view->left_theta = reinterpret_cast<const double *>(b + lay.off[2]);
        view->right_theta = reinterpret_cast<const double *>(b + lay.off[3]);
    }
    view->row_ptr = reinterpret_cast<const int32_t *>(b + lay.off[4]);
    view->col_idx = reinterpret_cast<const int32_t *>(b + lay.off[5]);
    view->best = reinterpret_cast<const double *>(b + lay.off[6]);
    view->keep_bits = reinterpret_cast<const uint32_t *>(b + lay.off[7]);
}

// the compact results a pair submitted with EBVO_PAIR_PUSH left in the slot's page-locked arena: valid from the pair's
// ebvo_stereo_wait until the slot's next submission
extern "C" int ebvo_stereo_pushed_view(ebvo_ctx *ctx, int slot, ebvo_stereo_compact_view *view)
{
    Slot *sp;
    if (!view || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (s.in_flight || !s.have_push || !s.h_push)
        return EBVO_ERR_STATE;
    const char *b = static_cast<const char *>(s.h_push);
    const bool theta = s.params.reserved & EBVO_PAIR_PUSH_THETA;
    const PackLayout lay = pack_layout((size_t)s.result.n_left, (size_t)s.result.n_right, (size_t)s.result.n_pairs, theta);
    fill_compact_view(view, b, lay, s.result, theta);
    return EBVO_OK;
}

extern "C" int ebvo_stereo_run(ebvo_ctx *ctx, const ebvo_stereo_params *p, ebvo_stereo_counts *counts)
{
    if (!ctx || !p || !counts)
        return EBVO_ERR_ARG;
    int rc = ebvo_stereo_submit(ctx, 0, p);
    if (rc)
        return rc;
    return ebvo_stereo_wait(ctx, 0, counts);
}

// ------------------------------------------------------------------------------------------
// Resident stage-wise path.  main_VO calls the stages one after the other (src/Pipeline.cpp:24-29, :93-97,
// src/Stereo_Matches.cpp:1374-1427) and hands every stage the vectors the previous one returned.  The calls below keep
// what a stage produced on the device (slot 0's two image workspaces: edges, counts) under a TAG, take tags instead of
// edge arrays, and return their results as pointers into page-locked memory of the context: nothing that the device
// already holds is uploaded again, no result passes through pageable memory.  A tag dies when its workspace gets another
// edge list: the next ebvo_toed_resident on it, ebvo_toed / ebvo_toed_pair, a pair uploaded into slot 0.  The other
// host-buffer entry points (SIFT, Best-Nearly-Best, refinement, ... -- what get_Stereo_Edge_Pairs calls between the stages)
// use slot 0's scratch buffers and image planes only and leave the tags valid.  A caller that cannot present valid tags
// gets EBVO_ERR_STATE and uses the host-buffer calls.
static int pinned_grow(ebvo_ctx *ctx, PinnedBuf &b, size_t bytes)
{
    if (bytes <= b.bytes)
        return EBVO_OK;
    if (b.p)
        EBVO_HIP(ctx, hipHostFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&b.p, want);
    if (e != hipSuccess)
    {
        b.p = nullptr;
        ebvo_fail_hip(ctx, e, "hipHostMalloc (stage-wise result arena)", __FILE__, __LINE__);
        return e == hipErrorOutOfMemory ? EBVO_ERR_NOMEM : EBVO_ERR_HIP;
    }
    b.bytes = want;
    return EBVO_OK;
}

static size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

// Host-to-device copy of caller memory through the context's page-locked staging (`off` bytes into sw_up, which the caller
// has sized).  The caller's buffers are ordinary pageable memory (a cv::Mat, a std::vector) that it frees or reuses right
// after the call: copying out of them directly makes the runtime lock their pages for the transfer, and unmapping locked
// pages later stalls the next submission (measured: ~1 ms on the first call of the next frame).
static int upload_staged(ebvo_ctx *ctx, Slot &s, void *d_dst, const void *src, size_t bytes, size_t off)
{
    if (!bytes)
        return EBVO_OK;
    char *stage = (char *)ctx->sw_up.p + off;
    memcpy(stage, src, bytes);
    EBVO_HIP(ctx, hipMemcpyAsync(d_dst, stage, bytes, hipMemcpyHostToDevice, s.stream));
    return EBVO_OK;
}
static int upload_image_staged(ebvo_ctx *ctx, Slot &s, int k, const uint8_t *img, int h, int w, ptrdiff_t stride, size_t off)
{
    if (stride < w)
        return EBVO_ERR_ARG;
    uint8_t *stage = (uint8_t *)ctx->sw_up.p + off;
    if (stride == (ptrdiff_t)w)
        memcpy(stage, img, (size_t)w * h);
    else
        for (int y = 0; y < h; ++y)
            memcpy(stage + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
    EBVO_HIP(ctx, hipMemcpyAsync(s.im[k].img, stage, (size_t)w * h, hipMemcpyHostToDevice, s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_toed_resident(ebvo_ctx *ctx, int which, const uint8_t *img, int h, int w, ptrdiff_t stride,
                                  int want_all4, ebvo_toed_view *view)
{
    if (!ctx || !img || !view || which < 0 || which > 1)
        return EBVO_ERR_ARG;
    memset(view, 0, sizeof *view);
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (ctx->sw_h != h || ctx->sw_w != w) // the two workspaces of a slot hold images of ONE size
        ctx->sw_tag[0] = ctx->sw_tag[1] = 0;
    ctx->sw_h = h;
    ctx->sw_w = w;
    ctx->sw_tag[which] = 0;
    static const bool trace = getenv("EBVO_TRACE_STAGEWISE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[6] = {0};
    if (trace)
        tt[0] = tnow();
    if ((rc = pinned_grow(ctx, ctx->sw_up, (size_t)h * w + 64)) || (rc = upload_image_staged(ctx, s, which, img, h, w, stride, 0)))
        return rc;
    if (trace)
    {
        tt[1] = tnow();
        (void)hipStreamSynchronize(s.stream);
        tt[2] = tnow();
    }
    float ms_c = 0, ms_n = 0;
    {
        // TOED of ONE image, in place in workspace `which`: the enqueue walks workspaces 0 .. n - 1, so the two are swapped
        // around it (plain pointer records; the kernels hold their own copies of the pointers)
        struct Swap
        {
            Slot &s;
            bool on;
            Swap(Slot &s_, bool on_) : s(s_), on(on_)
            {
                if (on)
                    std::swap(s.im[0], s.im[1]);
            }
            ~Swap()
            {
                if (on)
                    std::swap(s.im[0], s.im[1]);
            }
        } swap_ws(s, which == 1);
        if ((rc = toed_sync(ctx, s, 1, h, w, &ms_c, &ms_n)))
            return rc;
    }
    if (trace)
        tt[3] = tnow();
    const ImageWS &ws = s.im[which];
    view->n_kept = ws.n_kept;
    view->n_total = ws.n_total;
    view->t_conv = ms_c * 1e-3;
    view->t_nms = ms_n * 1e-3;
    if (ws.n_total > ctx->cap_edges)
    {
        ctx->last_error = "internal edge capacity exceeded";
        return EBVO_ERR_CAPACITY;
    }
    const size_t eb = align64(sizeof(ebvo_edge) * (size_t)ws.n_kept), ab = want_all4 ? sizeof(double) * 4 * (size_t)ws.n_total : 0;
    PinnedBuf &pb = ctx->sw_toed[which];
    if ((rc = pinned_grow(ctx, pb, eb + ab + 64)))
        return rc;
    if (ws.n_kept)
        EBVO_HIP(ctx, hipMemcpyAsync(pb.p, ws.edges, sizeof(ebvo_edge) * (size_t)ws.n_kept, hipMemcpyDeviceToHost, s.stream));
    if (ab)
        EBVO_HIP(ctx, hipMemcpyAsync((char *)pb.p + eb, ws.all4, ab, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    if (trace)
    {
        tt[4] = tnow();
        fprintf(stderr, "[ebvo] toed_resident(%d): upload call %.3f, upload done %.3f, toed + counts %.3f, result copies %.3f ms (device %.3f)\n",
                which, (tt[1] - tt[0]) * 1e3, (tt[2] - tt[1]) * 1e3, (tt[3] - tt[2]) * 1e3, (tt[4] - tt[3]) * 1e3, ms_c + ms_n);
    }
    view->edges = (const ebvo_edge *)pb.p;
    view->all4 = want_all4 ? (const double *)((char *)pb.p + eb) : nullptr;
    view->tag = ctx->sw_tag[which] = ++ctx->sw_seq;
    return EBVO_OK;
}

// which workspace holds `tag` (0 / 1), -1 if neither does
static int resident_ws(const ebvo_ctx *ctx, uint64_t tag)
{
    if (tag != 0 && ctx->sw_tag[0] == tag)
        return 0;
    if (tag != 0 && ctx->sw_tag[1] == tag)
        return 1;
    return -1;
}

extern "C" int ebvo_epi_candidates_resident(ebvo_ctx *ctx, uint64_t tag_left, uint64_t tag_right, const double *lines,
                                            double epi_thr, double max_disp, double orient_thr_deg, int stage_mask,
                                            int want_orient_flags, ebvo_candidates_view *view)
{
    if (!ctx || !view || (stage_mask & ~EBVO_STAGE_ALL) || stage_mask == 0)
        return EBVO_ERR_ARG;
    memset(view, 0, sizeof *view);
    const int iL = resident_ws(ctx, tag_left), iR = resident_ws(ctx, tag_right);
    if (iL < 0 || iR < 0 || iL == iR)
    {
        ctx->last_error = "the edge lists named by the tags are no longer resident";
        return EBVO_ERR_STATE;
    }
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    const int nL = s.im[iL].n_kept, nR = s.im[iR].n_kept;
    if (nL > 0 && !lines)
        return EBVO_ERR_ARG;
    const size_t rb = align64(sizeof(int32_t) * ((size_t)nL + 1));
    if ((rc = pinned_grow(ctx, ctx->sw_cand, rb + 64)))
        return rc;
    view->row_ptr = (const int32_t *)ctx->sw_cand.p;
    if (nL == 0 || nR == 0)
    {
        memset(ctx->sw_cand.p, 0, rb);
        if (want_orient_flags)
            view->row_ptr_final = view->row_ptr; // all zeros
        return EBVO_OK;
    }
    if ((rc = ebvo_grow(ctx, s, s.lines, sizeof(double) * 3 * (size_t)nL)))
        return rc;
    if ((rc = pinned_grow(ctx, ctx->sw_up, sizeof(double) * 3 * (size_t)nL + 64)) ||
        (rc = upload_staged(ctx, s, s.lines.p, lines, sizeof(double) * 3 * (size_t)nL, 0)))
        return rc;
    const ebvo_edge *dL = s.im[iL].edges, *dR = s.im[iR].edges;
    s.cap_pairs = 0;
    if ((rc = match_candidates_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p, epi_thr,
                                       max_disp, orient_thr_deg, stage_mask, false)))
    {
        (void)hipStreamSynchronize(s.stream); // the copy out of the caller's `lines` may still be in flight
        return rc;
    }
    unsigned long long *h_total_p = reinterpret_cast<unsigned long long *>(s.h_result);
    {
        hipError_t e1 = hipMemcpyAsync(h_total_p, s.d_total, sizeof(*h_total_p), hipMemcpyDeviceToHost, s.stream);
        hipError_t e2 = e1 == hipSuccess ? hipMemcpyAsync(ctx->sw_cand.p, s.row_ptr.p, sizeof(int32_t) * ((size_t)nL + 1),
                                                          hipMemcpyDeviceToHost, s.stream)
                                         : hipSuccess;
        hipError_t e3 = hipStreamSynchronize(s.stream);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
            return ebvo_fail_hip(ctx, e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "candidate total read-back",
                                 __FILE__, __LINE__);
    }
    const unsigned long long h_total = *h_total_p;
    if (h_total > 0x7fffffffull)
    {
        ctx->last_error = "candidate list exceeds 2^31-1 pairs";
        return EBVO_ERR_CAPACITY;
    }
    const int64_t np = (int64_t)h_total;
    view->n_pairs = np;
    if (np == 0)
    {
        if (want_orient_flags)
            view->row_ptr_final = view->row_ptr; // all zeros
        return EBVO_OK;
    }
    // arena: row_ptr | col_idx | flags | row_ptr_final | col_idx_final (the final list is a sub-list: np bounds it)
    const bool staged = want_orient_flags != 0;
    const size_t cb = align64(sizeof(int32_t) * (size_t)np), fb = staged ? align64((size_t)np) : 0;
    const size_t rb2 = staged ? rb : 0, cb2 = staged ? cb : 0;
    {
        // growing the arena moves it: the row offsets already copied are kept
        PinnedBuf &pb = ctx->sw_cand;
        const size_t need = rb + cb + fb + rb2 + cb2 + 64;
        if (pb.bytes < need)
        {
            PinnedBuf fresh;
            if ((rc = pinned_grow(ctx, fresh, need)))
                return rc;
            memcpy(fresh.p, pb.p, rb);
            (void)hipHostFree(pb.p);
            pb = fresh;
        }
        view->row_ptr = (const int32_t *)pb.p;
    }
    if ((rc = ebvo_grow(ctx, s, s.col_idx, sizeof(int32_t) * (size_t)np)))
        return rc;
    s.cap_pairs = np;
    char *base = (char *)ctx->sw_cand.p;
    unsigned long long *h_final_p = h_total_p + 1; // second word of the slot's pinned record
    static_assert(sizeof(PairResult) >= 2 * sizeof(unsigned long long), "pinned record too small for two totals");
    rc = [&]() -> int {
        int r;
        if ((r = match_candidates_fill_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p, epi_thr,
                                               max_disp, orient_thr_deg, stage_mask)))
            return r;
        if (staged)
        {
            if ((r = ebvo_grow(ctx, s, s.keep, (size_t)np)) ||
                (r = match_orient_flags_enqueue(ctx, s, dL, nL, dR, (const int32_t *)s.row_ptr.p, (const int32_t *)s.col_idx.p,
                                                np, orient_thr_deg, (uint8_t *)s.keep.p)))
                return r;
        }
        EBVO_HIP(ctx, hipMemcpyAsync(base + rb, s.col_idx.p, sizeof(int32_t) * (size_t)np, hipMemcpyDeviceToHost, s.stream));
        if (staged)
        {
            EBVO_HIP(ctx, hipMemcpyAsync(base + rb + cb, s.keep.p, (size_t)np, hipMemcpyDeviceToHost, s.stream));
            // the list the flags select = the search under all three predicates (stream order: the copies above have read
            // row_ptr / col_idx before these kernels overwrite them); it fits the buffers of the longer list
            if ((r = match_candidates_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p, epi_thr,
                                              max_disp, orient_thr_deg, stage_mask | EBVO_STAGE_ORIENTATION, false)) ||
                (r = match_candidates_fill_enqueue(ctx, s, dL, nL, nullptr, dR, nR, nullptr, 0, (const double *)s.lines.p, epi_thr,
                                                   max_disp, orient_thr_deg, stage_mask | EBVO_STAGE_ORIENTATION)))
                return r;
            EBVO_HIP(ctx, hipMemcpyAsync(h_final_p, s.d_total, sizeof(*h_final_p), hipMemcpyDeviceToHost, s.stream));
            EBVO_HIP(ctx, hipMemcpyAsync(base + rb + cb + fb, s.row_ptr.p, sizeof(int32_t) * ((size_t)nL + 1),
                                         hipMemcpyDeviceToHost, s.stream));
        }
        return EBVO_OK;
    }();
    hipError_t es = hipStreamSynchronize(s.stream);
    int64_t nf = 0;
    if (!rc && es == hipSuccess && staged)
    {
        nf = (int64_t)*h_final_p;
        if (nf > np) // cannot happen: a sub-list
        {
            ctx->last_error = "internal: the orientation-filtered list is longer than the list it filters";
            rc = EBVO_ERR_HIP;
        }
        else if (nf > 0)
        {
            hipError_t e = hipMemcpyAsync(base + rb + cb + fb + rb2, s.col_idx.p, sizeof(int32_t) * (size_t)nf, hipMemcpyDeviceToHost,
                                          s.stream);
            es = hipStreamSynchronize(s.stream);
            if (e != hipSuccess)
                es = e;
        }
    }
    s.cap_pairs = 0; // the pipeline re-establishes its own capacity
    if (rc)
        return rc;
    EBVO_HIP(ctx, es);
    view->col_idx = (const int32_t *)(base + rb);
    view->orient_ok = staged ? (const uint8_t *)(base + rb + cb) : nullptr;
    if (staged)
    {
        view->row_ptr_final = (const int32_t *)(base + rb + cb + fb);
        view->col_idx_final = (const int32_t *)(base + rb + cb + fb + rb2);
        view->n_final = nf;
    }
    return EBVO_OK;
}

extern "C" int ebvo_ncc_pairs_resident(ebvo_ctx *ctx, uint64_t tag_left, uint64_t tag_right, const uint8_t *imgL,
                                       const uint8_t *imgR, int h, int w, ptrdiff_t strideL, ptrdiff_t strideR,
                                       const int32_t *row_ptr, const int32_t *col_idx, double thr, int want,
                                       ebvo_ncc_view *view)
{
    if (!ctx || !view || !imgL || !imgR || !row_ptr || (want & ~(EBVO_NCC_WANT_LEFT_PATCHES | EBVO_NCC_WANT_SIMS)))
        return EBVO_ERR_ARG;
    memset(view, 0, sizeof *view);
    const int iL = resident_ws(ctx, tag_left), iR = resident_ws(ctx, tag_right);
    if (iL < 0 || iR < 0 || iL == iR)
    {
        ctx->last_error = "the edge lists named by the tags are no longer resident";
        return EBVO_ERR_STATE;
    }
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (h != ctx->sw_h || w != ctx->sw_w)
    {
        ctx->last_error = "image size differs from the size the resident edges were detected on";
        return EBVO_ERR_ARG;
    }
    const int nL = s.im[iL].n_kept, nR = s.im[iR].n_kept;
    if (row_ptr[0] != 0)
        return EBVO_ERR_ARG;
    const int64_t np = nL > 0 ? row_ptr[nL] : 0;
    if (np < 0 || (np > 0 && !col_idx))
        return EBVO_ERR_ARG;
    if (nL == 0)
        return EBVO_OK;
    // the lists go through page-locked staging and are validated on the way: the kernels index the right patch bank with
    // col_idx and walk the pair arrays with row_ptr
    const size_t rb = align64(sizeof(int32_t) * ((size_t)nL + 1)), cb = align64(sizeof(int32_t) * (size_t)np);
    const size_t ib = align64((size_t)h * w);
    if ((rc = pinned_grow(ctx, ctx->sw_up, rb + cb + 2 * ib + 64)))
        return rc;
    int32_t *h_rp = (int32_t *)ctx->sw_up.p, *h_ci = (int32_t *)((char *)ctx->sw_up.p + rb);
    {
        int32_t prev = 0, bad = 0;
        for (int i = 0; i <= nL; ++i)
        {
            const int32_t v = row_ptr[i];
            bad |= (v < prev);
            prev = v;
            h_rp[i] = v;
        }
        uint32_t over = 0;
        for (int64_t k = 0; k < np; ++k)
        {
            const int32_t v = col_idx[k];
            over |= ((uint32_t)v >= (uint32_t)nR);
            h_ci[k] = v;
        }
        if (bad || over)
        {
            ctx->last_error = bad ? "row_ptr is not non-decreasing" : "col_idx holds an index outside the resident right edge list";
            return EBVO_ERR_ARG;
        }
    }
    int64_t want_cap = s.pipe_cap > np ? s.pipe_cap : np + np / 4 + 1024;
    if (want_cap < 4096)
        want_cap = 4096;
    if ((rc = ensure_pipeline_buffers(ctx, s, want_cap)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)ctx->cap_edges + 1))))
        return rc;
    const size_t lb = (want & EBVO_NCC_WANT_LEFT_PATCHES) ? align64(sizeof(float) * 98 * (size_t)nL) : 0;
    const size_t sb = (want & EBVO_NCC_WANT_SIMS) ? align64(sizeof(double) * 4 * (size_t)np) : 0;
    const size_t bb = align64(sizeof(double) * (size_t)np), kb = align64((size_t)np);
    if ((rc = pinned_grow(ctx, ctx->sw_ncc, lb + sb + bb + kb + 64)))
        return rc;
    // the images the NCC samples are the RAW ones the caller passes (src/Stereo_Matches.cpp:562-563); they replace the
    // images in the workspaces, the edges stay
    if ((rc = upload_image_staged(ctx, s, iL, imgL, h, w, strideL, rb + cb)) ||
        (rc = upload_image_staged(ctx, s, iR, imgR, h, w, strideR, rb + cb + ib)))
    {
        (void)hipStreamSynchronize(s.stream);
        return rc;
    }
    bool patches_on_copy_stream = false;
    rc = [&]() -> int {
        int r;
        EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, h_rp, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice, s.stream));
        if (np)
            EBVO_HIP(ctx, hipMemcpyAsync(s.col_idx.p, h_ci, sizeof(int32_t) * (size_t)np, hipMemcpyHostToDevice, s.stream));
        if (lb && (r = match_patches_enqueue(ctx, s, s.im[iL].img, h, w, w, s.im[iL].edges, nL, nullptr, 0,
                                             (float *)s.patches_raw.p, (float *)s.patches_norm.p, (uint8_t *)s.patches_flag.p)))
            return r;
        char *base = (char *)ctx->sw_ncc.p;
        if (lb)
        {
            // the left patches (the largest result: 392 bytes per left edge) travel on the copy stream while the pairs are scored
            if (!ctx->copy_stream)
                EBVO_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            EBVO_HIP(ctx, hipEventRecord(s.ev_done, s.stream));
            EBVO_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, s.ev_done, 0));
            EBVO_HIP(ctx, hipMemcpyAsync(base, s.patches_raw.p, sizeof(float) * 98 * (size_t)nL, hipMemcpyDeviceToHost,
                                         ctx->copy_stream));
            patches_on_copy_stream = true;
        }
        if (np)
        {
            if ((r = match_ncc_resident_enqueue(ctx, s, h, w, ctx->cap_edges, thr, iL)))
                return r;
            if (sb)
                EBVO_HIP(ctx, hipMemcpyAsync(base + lb, s.sims.p, sizeof(double) * 4 * (size_t)np, hipMemcpyDeviceToHost, s.stream));
            EBVO_HIP(ctx, hipMemcpyAsync(base + lb + sb, s.best.p, sizeof(double) * (size_t)np, hipMemcpyDeviceToHost, s.stream));
            EBVO_HIP(ctx, hipMemcpyAsync(base + lb + sb + bb, s.keep.p, (size_t)np, hipMemcpyDeviceToHost, s.stream));
        }
        return EBVO_OK;
    }();
    const hipError_t es = hipStreamSynchronize(s.stream);
    const hipError_t ec = patches_on_copy_stream ? hipStreamSynchronize(ctx->copy_stream) : hipSuccess;
    if (rc)
        return rc;
    EBVO_HIP(ctx, es);
    EBVO_HIP(ctx, ec);
    char *base = (char *)ctx->sw_ncc.p;
    view->n_left = nL;
    view->n_pairs = np;
    view->left_patches = lb ? (const float *)base : nullptr;
    view->sims = sb ? (const double *)(base + lb) : nullptr;
    view->best = np ? (const double *)(base + lb + sb) : nullptr;
    view->keep = np ? (const uint8_t *)(base + lb + sb + bb) : nullptr;
    return EBVO_OK;
}

extern "C" int ebvo_gn_refine_temporal(ebvo_ctx *ctx, const uint8_t *imgKF, const uint8_t *imgCF, int h, int w,
                                       ptrdiff_t strideKF, ptrdiff_t strideCF, const ebvo_edge *kf, const ebvo_edge *cf,
                                       const double *init_disp, int n, const ebvo_gn_params *params, double *disp,
                                       double *score, uint8_t *validity, int32_t *iters)
{
    if (!ctx || !imgKF || !imgCF || n < 0 || !params || params->max_iter < 1 || !(params->tol >= 0) ||
        !(params->huber_delta > 0))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (n == 0)
        return EBVO_OK;
    if (!kf || !cf || !init_disp || !disp || !score || !validity || !iters)
        return EBVO_ERR_ARG;
    const size_t nz = (size_t)n;
    if ((rc = upload_image(ctx, s, 0, imgKF, h, w, strideKF)) || (rc = upload_image(ctx, s, 1, imgCF, h, w, strideCF)) ||
        (rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * nz)) ||
        (rc = ebvo_grow(ctx, s, s.scratch_c, sizeof(ebvo_edge) * nz)) || (rc = ebvo_grow(ctx, s, s.gn_xy, sizeof(double) * 2 * nz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_out, sizeof(double) * 3 * nz)) || (rc = ebvo_grow(ctx, s, s.gn_valid, nz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_iters, sizeof(int32_t) * nz)))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, kf, sizeof(ebvo_edge) * nz, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_c.p, cf, sizeof(ebvo_edge) * nz, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.gn_xy.p, init_disp, sizeof(double) * 2 * nz, hipMemcpyHostToDevice, st));
    double *out = (double *)s.gn_out.p;
    // (the refinement packs the current-frame image itself: intensity + Sobel gradients per bilinear cell)
    if ((rc = refine_gn_temporal_enqueue(ctx, s, s.im[0].img, s.im[1].img, nullptr, h, w,
                                         (const ebvo_edge *)s.scratch_b.p, (const ebvo_edge *)s.scratch_c.p,
                                         (const double *)s.gn_xy.p, n, params->max_iter, params->tol, params->huber_delta,
                                         out, out + 2 * nz, (uint8_t *)s.gn_valid.p, (int32_t *)s.gn_iters.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(disp, out, sizeof(double) * 2 * nz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(score, out + 2 * nz, sizeof(double) * nz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(validity, s.gn_valid.p, nz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipMemcpyAsync(iters, s.gn_iters.p, sizeof(int32_t) * nz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

// ---- stage glue on CSR candidate lists -------------------------------------------------------------------------
static int check_csr(const int32_t *row_ptr, int nL, int64_t *np)
{
    if (nL < 0 || !row_ptr || row_ptr[0] != 0)
        return EBVO_ERR_ARG;
    for (int i = 0; i < nL; ++i)
        if (row_ptr[i + 1] < row_ptr[i])
            return EBVO_ERR_ARG;
    *np = row_ptr[nL];
    return EBVO_OK;
}

static int row_select(ebvo_ctx *ctx, const int32_t *row_ptr, int nL, const double *scores, double thr, int mode,
                      int32_t *new_count, int32_t *order)
{
    int64_t np = 0;
    if (!ctx || check_csr(row_ptr, nL, &np) || (nL > 0 && !new_count) || (np > 0 && (!scores || !order)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (nL == 0)
        return EBVO_OK;
    const size_t npz = (size_t)np;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.best, sizeof(double) * (npz + 1))) ||
        (rc = ebvo_grow(ctx, s, s.cand_cnt, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * (npz + 1))))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice, st));
    if (npz)
        EBVO_HIP(ctx, hipMemcpyAsync(s.best.p, scores, sizeof(double) * npz, hipMemcpyHostToDevice, st));
    if (mode == 2)
        rc = glue_keep_best_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, nL, (const double *)s.best.p,
                                    (int32_t *)s.cand_cnt.p, (int32_t *)s.pair_left.p);
    else
        rc = glue_bnb_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, nL, (const double *)s.best.p, thr, mode,
                              (int32_t *)s.cand_cnt.p, (int32_t *)s.pair_left.p);
    if (rc)
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(new_count, s.cand_cnt.p, sizeof(int32_t) * (size_t)nL, hipMemcpyDeviceToHost, st));
    if (npz)
        EBVO_HIP(ctx, hipMemcpyAsync(order, s.pair_left.p, sizeof(int32_t) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_bnb_test(ebvo_ctx *ctx, const int32_t *row_ptr, int nL, const double *scores, double ratio_thr,
                             int higher_is_better, int32_t *new_count, int32_t *order)
{
    return row_select(ctx, row_ptr, nL, scores, ratio_thr, higher_is_better ? 1 : 0, new_count, order);
}

extern "C" int ebvo_keep_best(ebvo_ctx *ctx, const int32_t *row_ptr, int nL, const double *scores, int32_t *new_count,
                              int32_t *order)
{
    return row_select(ctx, row_ptr, nL, scores, 0.0, 2, new_count, order);
}

extern "C" int ebvo_epipolar_shift(ebvo_ctx *ctx, const ebvo_edge *cand, const double *lines, const int32_t *row_ptr, int nL,
                                   ebvo_edge *shifted)
{
    int64_t np = 0;
    if (!ctx || check_csr(row_ptr, nL, &np) || (np > 0 && (!cand || !lines || !shifted)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (np == 0)
        return EBVO_OK;
    const size_t npz = (size_t)np;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.lines, sizeof(double) * 3 * (size_t)nL)) ||
        (rc = ebvo_grow(ctx, s, s.rc_edges, sizeof(ebvo_edge) * npz)) || (rc = ebvo_grow(ctx, s, s.scratch_c, sizeof(ebvo_edge) * npz)) ||
        (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * npz)))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.lines.p, lines, sizeof(double) * 3 * (size_t)nL, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.rc_edges.p, cand, sizeof(ebvo_edge) * npz, hipMemcpyHostToDevice, st));
    if ((rc = match_expand_rows_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, nL, np, (int32_t *)s.pair_left.p)) ||
        (rc = glue_shift_enqueue(ctx, s, (const ebvo_edge *)s.rc_edges.p, (const double *)s.lines.p,
                                 (const int32_t *)s.pair_left.p, np, (ebvo_edge *)s.scratch_c.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(shifted, s.scratch_c.p, sizeof(ebvo_edge) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_cluster_rows(ebvo_ctx *ctx, const ebvo_edge *cand, const int32_t *row_ptr, int nL, int by_orientation,
                                 int skip_single, int32_t *new_count, ebvo_edge *centres, int32_t *cluster_of)
{
    int64_t np = 0;
    if (!ctx || check_csr(row_ptr, nL, &np) || (nL > 0 && !new_count) || (np > 0 && (!cand || !centres || !cluster_of)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (nL == 0)
        return EBVO_OK;
    const size_t npz = (size_t)np;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.cand_cnt, sizeof(int32_t) * ((size_t)nL + 1))) ||
        (rc = ebvo_grow(ctx, s, s.rc_edges, sizeof(ebvo_edge) * (npz + 1))) ||
        (rc = ebvo_grow(ctx, s, s.scratch_c, sizeof(ebvo_edge) * (npz + 1))) ||
        (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * (npz + 1))))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * ((size_t)nL + 1), hipMemcpyHostToDevice, st));
    if (npz)
        EBVO_HIP(ctx, hipMemcpyAsync(s.rc_edges.p, cand, sizeof(ebvo_edge) * npz, hipMemcpyHostToDevice, st));
    if ((rc = glue_cluster_enqueue(ctx, s, (const ebvo_edge *)s.rc_edges.p, (const int32_t *)s.row_ptr.p, nL, by_orientation,
                                   skip_single, (int32_t *)s.cand_cnt.p, (ebvo_edge *)s.scratch_c.p, (int32_t *)s.pair_left.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(new_count, s.cand_cnt.p, sizeof(int32_t) * (size_t)nL, hipMemcpyDeviceToHost, st));
    if (npz)
    {
        EBVO_HIP(ctx, hipMemcpyAsync(centres, s.scratch_c.p, sizeof(ebvo_edge) * npz, hipMemcpyDeviceToHost, st));
        EBVO_HIP(ctx, hipMemcpyAsync(cluster_of, s.pair_left.p, sizeof(int32_t) * npz, hipMemcpyDeviceToHost, st));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_finalize_pairs(ebvo_ctx *ctx, const ebvo_stereo_calib *calib, const ebvo_edge *left,
                                   const ebvo_edge *right, int n, double *out16)
{
    if (!ctx || !calib || n < 0 || (n > 0 && (!left || !right || !out16)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (n == 0)
        return EBVO_OK;
    const size_t nz = (size_t)n;
    if ((rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * nz)) || (rc = ebvo_grow(ctx, s, s.scratch_c, sizeof(ebvo_edge) * nz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_out, sizeof(double) * 16 * nz)))
        return rc;
    hipStream_t st = s.stream;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, left, sizeof(ebvo_edge) * nz, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_c.p, right, sizeof(ebvo_edge) * nz, hipMemcpyHostToDevice, st));
    if ((rc = refine_finalize_pairs_enqueue(ctx, s, calib->K_left, calib->K_right, calib->R21, calib->T21,
                                            (const ebvo_edge *)s.scratch_b.p, (const ebvo_edge *)s.scratch_c.p, n,
                                            (double *)s.gn_out.p)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(out16, s.gn_out.p, sizeof(double) * 16 * nz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

// ---- photometric refinement of the kept matches of a resident pair ---------------------------------------------
extern "C" int ebvo_stereo_refine(ebvo_ctx *ctx, int slot, const ebvo_gn_params *params)
{
    Slot *sp;
    if (!params || params->max_iter < 1 || !(params->tol >= 0) || !(params->huber_delta > 0) || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_run || s.in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc_f = drain_fetch(ctx, s))
        return rc_f;
    const int64_t np = s.result.n_pairs;
    s.have_refined = false;
    if (np == 0)
    {
        s.have_refined = true;
        return EBVO_OK;
    }
    const int h = s.cur_h, w = s.cur_w;
    const size_t npz = (size_t)np;
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.gn_out, sizeof(double) * 5 * npz)) || (rc = ebvo_grow(ctx, s, s.gn_valid, npz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_iters, sizeof(int32_t) * npz)))
        return rc;
    double *out = (double *)s.gn_out.p;
    // the pipeline's own device arrays: left / right TOED edges, epipolar lines, NCC keep flags; the pair -> left edge
    // expansion of the CSR is formed here (the matching kernels themselves work per tile of rows)
    if ((rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * npz)) ||
        (rc = match_expand_rows_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, s.result.n_left, np, (int32_t *)s.pair_left.p)) ||
        (rc = refine_gn_stereo_enqueue(ctx, s, s.im[0].img, s.im[1].img, nullptr, h, w, s.im[0].edges, s.result.n_left,
                                       (const double *)s.lines.p, (const int32_t *)s.pair_left.p, nullptr, s.im[1].edges,
                                       (const int32_t *)s.col_idx.p, (const uint8_t *)s.keep.p, np, params->max_iter,
                                       params->tol, params->huber_delta, out, out + npz, out + 2 * npz,
                                       (uint8_t *)s.gn_valid.p, (int32_t *)s.gn_iters.p, out + 3 * npz)))
        return rc;
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    s.have_refined = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch_refined(ebvo_ctx *ctx, int slot, double *alpha, double *score, double *confidence,
                                         uint8_t *validity, int32_t *iters, double *refined_xy)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_refined || s.in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npz = (size_t)s.result.n_pairs;
    if (!npz)
        return EBVO_OK;
    const double *out = (const double *)s.gn_out.p;
    hipStream_t st = s.stream;
    if (alpha)
        EBVO_HIP(ctx, hipMemcpyAsync(alpha, out, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    if (score)
        EBVO_HIP(ctx, hipMemcpyAsync(score, out + npz, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    if (confidence)
        EBVO_HIP(ctx, hipMemcpyAsync(confidence, out + 2 * npz, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    if (refined_xy)
        EBVO_HIP(ctx, hipMemcpyAsync(refined_xy, out + 3 * npz, sizeof(double) * 2 * npz, hipMemcpyDeviceToHost, st));
    if (validity)
        EBVO_HIP(ctx, hipMemcpyAsync(validity, s.gn_valid.p, npz, hipMemcpyDeviceToHost, st));
    if (iters)
        EBVO_HIP(ctx, hipMemcpyAsync(iters, s.gn_iters.p, sizeof(int32_t) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

// ---- the stages after the NCC pass, on the resident pair: [SIFT filter ->] BNB -> shift -> refine -> cluster -> NCC -> best
// Every stage is the kernel behind the corresponding host-buffer entry point; only the CSR bookkeeping between them is
// new (glue_kernels.hip).  The chain is ENQUEUED: a stage's survivor count stays on the device (the last entry of the
// scanned row offsets), the next stage's kernels are launched for the upper bound (the pair count of the run: every stage
// only shrinks the lists) and read the count there.  The six stage totals travel to page-locked memory in one copy behind
// the last kernel; ebvo_stereo_finalize_wait is the only host synchronisation.  No candidate data leaves HBM.
static int read_i32(ebvo_ctx *ctx, Slot &s, const int32_t *d, int32_t *h)
{
    EBVO_HIP(ctx, hipMemcpyAsync(h, d, sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

static int finalize_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_finalize_params *p, const ebvo_stereo_calib *calib)
{
    const int nL = s.result.n_left, h = s.cur_h, w = s.cur_w;
    const int64_t n0 = s.result.n_pairs;
    const size_t nz = (size_t)n0, nLz = (size_t)nL + 1;
    int rc;
    // carve the work buffers: everything is bounded by the pair count of the run (each stage only shrinks the lists)
    if ((rc = ebvo_grow(ctx, s, s.fin_i32, sizeof(int32_t) * (4 * nLz + 4 * nz))) ||
        (rc = ebvo_grow(ctx, s, s.fin_edges, sizeof(ebvo_edge) * (3 * nz + 2 * nLz))) ||
        (rc = ebvo_grow(ctx, s, s.fin_f64, sizeof(double) * (5 * nz + nLz) + 16 * nz)) ||
        (rc = ebvo_grow(ctx, s, s.fin_u8, nz)) || (rc = ebvo_grow(ctx, s, s.fin_out, sizeof(double) * 16 * nLz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_xy, sizeof(double) * 2 * nz)) || (rc = ebvo_grow(ctx, s, s.gn_out, sizeof(double) * 5 * nz)) ||
        (rc = ebvo_grow(ctx, s, s.gn_valid, nz)) || (rc = ebvo_grow(ctx, s, s.gn_iters, sizeof(int32_t) * nz)))
        return rc;
    // (the final lists keep their places: ebvo_stereo_fetch_final and the temporal stage read final_left, fin_l / fin_r and
    // fin_score at these offsets)
    int32_t *cnt = (int32_t *)s.fin_i32.p, *rpA = cnt + nLz, *rpB = rpA + nLz, *final_left = rpB + nLz;
    int32_t *order = final_left + nLz, *left_of = order + nz, *cluster_of = left_of + nz, *ncc_left = cluster_of + nz;
    ebvo_edge *candA = (ebvo_edge *)s.fin_edges.p, *candB = candA + nz, *candC = candB + nz, *fin_l = candC + nz,
              *fin_r = fin_l + nLz;
    double *scoreA = (double *)s.fin_f64.p, *scoreB = scoreA + nz, *best2 = scoreB + nz, *fin_score = best2 + nz;
    double *confA = fin_score + nLz, *confB = confA + nz; // SIFT distances carried through the Best-Nearly-Best tests
    void *sincos2 = confB + nz; // n0 double2
    uint8_t *keep2 = (uint8_t *)s.fin_u8.p;
    const int32_t *rp0 = (const int32_t *)s.row_ptr.p;
    hipStream_t st = s.stream;
    int32_t *tot = s.d_fin_tot;
    EBVO_HIP(ctx, hipMemsetAsync(tot, 0, sizeof(int32_t) * 8, st));
    auto scan_counts = [&](int32_t *rp_out, int which) -> int {
        // exclusive scan of cnt[0 .. nL) with the total at rp_out[nL]; the scan also stores it as stage total `which`
        // (no memset of a trailing zero, no copy of the total: two launches less per stage)
        return ebvo_device_scan(ctx, s, cnt, rp_out, nL, nullptr, 1, nL + 1, tot + which);
    };
    enum { T_SIFT = 0, T_NCC, T_BNB, T_CLUSTERS, T_NCC2, T_FINAL };
    const uint8_t *keep1 = (const uint8_t *)s.keep.p;
    const double *conf0 = nullptr; // SIFT distance per pair of the run (refine_confidences, :757)
    if (p->use_sift)
    {
        // 0. augment_Edge_Data (:1410) + apply_SIFT_filtering (:1414): descriptors of every left and right TOED edge on the
        // undistorted images (each right edge once, not once per left edge that lists it), the smallest of the four
        // distances per candidate pair, dist < thr.  NCC scores do not depend on which pairs survive, so the SIFT filter
        // applied to the NCC-scored pairs of the run selects the reference's SIFT-then-NCC survivors.
        const int nR = s.result.n_right;
        const size_t npx = (size_t)h * w;
        if ((rc = ebvo_grow(ctx, s, s.sift_img, sizeof(float) * 3 * npx)) ||
            (rc = ebvo_grow(ctx, s, s.sift_desc, 256 * ((size_t)nL + (size_t)nR))) ||
            (rc = ebvo_grow(ctx, s, s.sift_dist, sizeof(double) * nz + 2 * nz)) ||
            (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * nz)) ||
            (rc = ebvo_grow(ctx, s, s.sift_used, sizeof(int32_t) * (4 + (size_t)nL + (size_t)nR) + (size_t)nR)))
            return rc;
        float *tmp = (float *)s.sift_img.p, *base = tmp + npx, *base1 = base + npx; // (both octave bases: one descriptor launch)
        uint8_t *dL = (uint8_t *)s.sift_desc.p, *dR = dL + 256 * (size_t)nL;
        double *dist = (double *)s.sift_dist.p;
        uint8_t *ok = (uint8_t *)(dist + nz), *both = ok + nz;
        // Only an edge that appears in a candidate pair has its descriptor read (by the distances below; the temporal stage
        // reuses those of the final mates' left edges): on the EuRoC-shaped pair that is 45 % of the edges of either image
        // (1.2 candidates per left edge; KITTI-shaped: 97 %), and the descriptors are the dearest kernel of this chain.
        int32_t *used_cnt = (int32_t *)s.sift_used.p, *usedL = used_cnt + 4, *usedR = usedL + nL;
        uint8_t *used_flags = (uint8_t *)(usedR + nR);
        if ((rc = sift_used_edges_enqueue(ctx, s, rp0, nL, (const int32_t *)s.col_idx.p, n0, nR, used_cnt, usedL, usedR, used_flags)) ||
            (rc = sift_base_enqueue(ctx, s, s.im[0].img, h, w, w, tmp, base)) ||
            (rc = sift_base_enqueue(ctx, s, s.im[1].img, h, w, w, tmp, base1)) ||
            (rc = sift_descriptors_listed_pair_enqueue(ctx, s, base, base1, h, w, s.im[0].edges, s.im[1].edges, usedL, usedR, used_cnt, nL,
                                                       nR, dL, dR)) ||
            (rc = match_expand_rows_enqueue(ctx, s, rp0, nL, n0, (int32_t *)s.pair_left.p)) ||
            (rc = sift_distances_enqueue(ctx, s, dL, dR, (const int32_t *)s.pair_left.p, (const int32_t *)s.col_idx.p, n0,
                                         p->sift_thr, dist, ok)) ||
            (rc = sift_and_flags_enqueue(ctx, s, ok, (const uint8_t *)s.keep.p, n0, both)))
            return rc;
        if ((rc = glue_rows_from_flags_enqueue(ctx, s, rp0, nL, ok, cnt, order)) || (rc = scan_counts(rpA, T_SIFT)))
            return rc;
        keep1 = both;
        conf0 = dist;
        s.sift_left_valid = true; // dL: the descriptors of every left TOED edge that has a candidate (ebvo_temporal_* reuse the final mates')
    }
    // 1. the kept NCC matches as a CSR list of right TOED edges with their scores (apply_NCC_Filtering's output, :597-607)
    if ((rc = glue_rows_from_flags_enqueue(ctx, s, rp0, nL, keep1, cnt, order)) || (rc = scan_counts(rpA, T_NCC)) ||
        (rc = glue_gather_rows_enqueue(ctx, s, rp0, cnt, order, rpA, nL, s.im[1].edges, (const int32_t *)s.col_idx.p, candA,
                                       (const double *)s.best.p, scoreA)) ||
        (conf0 && (rc = glue_gather_rows_enqueue(ctx, s, rp0, cnt, order, rpA, nL, nullptr, nullptr, nullptr, conf0, confA))))
        return rc;
    // 2. Best-Nearly-Best test on the NCC scores (:1440) ...
    if ((rc = glue_bnb_enqueue(ctx, s, rpA, nL, scoreA, p->bnb_ratio, 1, cnt, order)) || (rc = scan_counts(rpB, T_BNB)) ||
        (rc = glue_gather_rows_enqueue(ctx, s, rpA, cnt, order, rpB, nL, candA, nullptr, candB, scoreA, scoreB)) ||
        (conf0 && (rc = glue_gather_rows_enqueue(ctx, s, rpA, cnt, order, rpB, nL, nullptr, nullptr, nullptr, confA, confB))))
        return rc;
    if (conf0)
    {
        // ... and on the SIFT distances, lower is better (:1452).  The survivors land in candA / rpA; the two sets of buffers
        // swap their roles instead of being copied back.
        if ((rc = glue_bnb_enqueue(ctx, s, rpB, nL, confB, p->bnb_sift, 0, cnt, order)) || (rc = scan_counts(rpA, T_BNB)) ||
            (rc = glue_gather_rows_enqueue(ctx, s, rpB, cnt, order, rpA, nL, candB, nullptr, candA, scoreB, scoreA)))
            return rc;
        std::swap(rpA, rpB);
        std::swap(candA, candB);
        std::swap(scoreA, scoreB);
    }
    const int32_t *d_nB = rpB + nL; // pairs that enter the refinement: a subset of the run's kept NCC matches
    const int64_t n_ref = s.result.n_matches < n0 ? (s.result.n_matches > 0 ? s.result.n_matches : 1) : n0;
    {
        // 3. epipolar shift (:1436) and 4. photometric refinement along the epipolar line (:1438)
        double *out = (double *)s.gn_out.p;
        if ((rc = match_expand_rows_enqueue(ctx, s, rpB, nL, n0, left_of)) ||
            (rc = glue_shift_enqueue(ctx, s, candB, (const double *)s.lines.p, left_of, n0, candC, d_nB)) ||
            (rc = glue_xy_enqueue(ctx, s, candC, (double *)s.gn_xy.p, n0, false, d_nB)) ||
            (rc = refine_gn_stereo_enqueue(ctx, s, s.im[0].img, s.im[1].img, nullptr, h, w, s.im[0].edges, nL,
                                           (const double *)s.lines.p, left_of, (const double *)s.gn_xy.p, nullptr, nullptr,
                                           nullptr, n_ref, p->gn.max_iter, p->gn.tol, p->gn.huber_delta, out, out + nz,
                                           out + 2 * nz, (uint8_t *)s.gn_valid.p, (int32_t *)s.gn_iters.p, out + 3 * nz, d_nB)) ||
            (rc = glue_xy_enqueue(ctx, s, candC, out + 3 * nz, n0, true, d_nB)))
            return rc;
        // 5. consolidate_redundant_edge_hypothesis(pairs, false, true) (:1483).  Against the signature (pairs, frame_idx,
        // b_do_epipolar_shift = true, b_do_clustering = true) this binds frame_idx = 0, shift = true, cluster = true: the
        // refined centres are shifted to the epipolar line AGAIN (:981-998) and clustered with b_cluster_by_orientation =
        // b_do_epipolar_shift = true (:1031); rows with one candidate are not skipped on the shift branch (:1001 is the
        // else branch only).  -> candA in the rows of rpA
        if ((rc = glue_shift_enqueue(ctx, s, candC, (const double *)s.lines.p, left_of, n0, candA, d_nB)) ||
            (rc = glue_cluster_enqueue(ctx, s, candA, rpB, nL, 1, 0, cnt, candB, cluster_of)) ||
            (rc = scan_counts(rpA, T_CLUSTERS)) ||
            (rc = glue_gather_rows_enqueue(ctx, s, rpB, cnt, nullptr, rpA, nL, candB, nullptr, candA, nullptr, nullptr)))
            return rc;
    }
    {
        // 6. second NCC pass on the cluster centres (:1500) -> keep2 / best2, 7. the best survivor of every row (:1513).
        // The normalised left patches (left_edge_patches, :578) are sampled here: the first pass keeps them in LDS only.
        if ((rc = match_patches_enqueue(ctx, s, ncc_img(s, 0), h, w, w, s.im[0].edges, nL, nullptr, 0, nullptr,
                                        (float *)s.patches_norm.p, (uint8_t *)s.patches_flag.p)) ||
            (rc = match_ncc_pairs_enqueue(ctx, s, ncc_img(s, 1), h, w, w, candA, rpA, nL, n0, (const float *)s.patches_norm.p,
                                          (const uint8_t *)s.patches_flag.p, p->ncc_thr, nullptr, best2, keep2, ncc_left,
                                          sincos2, rpA + nL)) ||
            (rc = glue_rows_from_flags_enqueue(ctx, s, rpA, nL, keep2, cnt, order)) || (rc = scan_counts(rpB, T_NCC2)) ||
            (rc = glue_gather_rows_enqueue(ctx, s, rpA, cnt, order, rpB, nL, candA, nullptr, candC, best2, scoreB)))
            return rc;
        if ((rc = glue_keep_best_enqueue(ctx, s, rpB, nL, scoreB, cnt, order)) || (rc = scan_counts(rpA, T_FINAL)) ||
            (rc = glue_final_pairs_enqueue(ctx, s, rpB, cnt, order, rpA, nL, s.im[0].edges, candC, scoreB, final_left,
                                           fin_l, fin_r, fin_score)))
            return rc;
    }
    // 8. the rows of the output file (at most one final pair per left edge)
    if (calib && (rc = refine_finalize_pairs_enqueue(ctx, s, calib->K_left, calib->K_right, calib->R21, calib->T21, fin_l, fin_r,
                                                     nL, (double *)s.fin_out.p, tot + T_FINAL)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.h_fin_tot, tot, sizeof(int32_t) * 8, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipEventRecord(s.ev_fin, st));
    return EBVO_OK;
}

extern "C" int ebvo_stereo_finalize_submit(ebvo_ctx *ctx, int slot, const ebvo_finalize_params *p, const ebvo_stereo_calib *calib)
{
    Slot *sp;
    if (!p || !(p->bnb_ratio >= 0) || !(p->ncc_thr == p->ncc_thr) || p->gn.max_iter < 1 || !(p->gn.tol >= 0) ||
        !(p->gn.huber_delta > 0) || (p->use_sift && (!(p->sift_thr > 0) || !(p->bnb_sift >= 0))) || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_run || s.in_flight || s.fin_in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc_f = drain_fetch(ctx, s))
        return rc_f;
    s.have_final = s.have_refined = false; // the refinement buffers are reused
    s.tq_n = -1;
    s.tq_final.n = -1;
    s.sift_left_valid = false;
    s.n_final = 0;
    s.final_has_rows = calib != nullptr;
    memset(s.h_fin_tot, 0, sizeof(int32_t) * 8);
    if (s.result.n_left == 0 || s.result.n_pairs == 0)
    {
        EBVO_HIP(ctx, hipEventRecord(s.ev_fin, s.stream)); // nothing to do: the wait returns zero counts
        s.fin_in_flight = true;
        return EBVO_OK;
    }
    int rc = finalize_enqueue(ctx, s, p, calib);
    if (rc)
    {
        (void)hipStreamSynchronize(s.stream); // part of the chain may be enqueued
        return rc;
    }
    s.fin_in_flight = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_finalize_wait(ebvo_ctx *ctx, int slot, ebvo_finalize_counts *counts)
{
    Slot *sp;
    if (!counts || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.fin_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    s.fin_in_flight = false;
    EBVO_HIP(ctx, hipEventSynchronize(s.ev_fin));
    const int32_t *t = s.h_fin_tot;
    memset(counts, 0, sizeof *counts);
    counts->n_sift = t[0];
    counts->n_ncc = t[1];
    counts->n_bnb = t[2];
    counts->n_clusters = t[3];
    counts->n_ncc2 = t[4];
    counts->n_final = t[5];
    s.n_final = t[5];
    s.have_final = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_finalize(ebvo_ctx *ctx, int slot, const ebvo_finalize_params *p, const ebvo_stereo_calib *calib,
                                    ebvo_finalize_counts *counts)
{
    if (!counts)
        return EBVO_ERR_ARG;
    int rc = ebvo_stereo_finalize_submit(ctx, slot, p, calib);
    return rc ? rc : ebvo_stereo_finalize_wait(ctx, slot, counts);
}

extern "C" int ebvo_stereo_fetch_final(ebvo_ctx *ctx, int slot, int32_t *left_index, ebvo_edge *right_edge, double *ncc_score,
                                       double *out16)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_final || s.in_flight)
        return EBVO_ERR_STATE;
    if (out16 && !s.final_has_rows)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)s.n_final;
    if (!n)
        return EBVO_OK;
    const size_t nz = (size_t)s.result.n_pairs, nLz = (size_t)s.result.n_left + 1;
    const int32_t *final_left = (const int32_t *)s.fin_i32.p + 3 * nLz;
    const ebvo_edge *fin_r = (const ebvo_edge *)s.fin_edges.p + 3 * nz + nLz;
    const double *fin_score = (const double *)s.fin_f64.p + 3 * nz;
    hipStream_t st = s.stream;
    if (left_index)
        EBVO_HIP(ctx, hipMemcpyAsync(left_index, final_left, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    if (right_edge)
        EBVO_HIP(ctx, hipMemcpyAsync(right_edge, fin_r, sizeof(ebvo_edge) * n, hipMemcpyDeviceToHost, st));
    if (ncc_score)
        EBVO_HIP(ctx, hipMemcpyAsync(ncc_score, fin_score, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (out16)
        EBVO_HIP(ctx, hipMemcpyAsync(out16, s.fin_out.p, sizeof(double) * 16 * n, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch_slot(ebvo_ctx *ctx, int slot, ebvo_edge *left, ebvo_edge *right, int32_t *row_ptr,
                                      int32_t *col_idx, double *sims, double *best, uint8_t *keep, float *left_patches)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_run || s.in_flight)
        return EBVO_ERR_STATE;
    if (sims && !s.have_sims)
    {
        ctx->last_error = "the pair was submitted with EBVO_PAIR_NO_SIMS: the four scores were not stored";
        return EBVO_ERR_STATE;
    }
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = (size_t)s.result.n_left, nR = (size_t)s.result.n_right, np = (size_t)s.result.n_pairs;
    hipStream_t st = s.stream;
    if (left && nL)
        EBVO_HIP(ctx, hipMemcpyAsync(left, s.im[0].edges, sizeof(ebvo_edge) * nL, hipMemcpyDeviceToHost, st));
    if (right && nR)
        EBVO_HIP(ctx, hipMemcpyAsync(right, s.im[1].edges, sizeof(ebvo_edge) * nR, hipMemcpyDeviceToHost, st));
    if (row_ptr)
        EBVO_HIP(ctx, hipMemcpyAsync(row_ptr, s.row_ptr.p, sizeof(int32_t) * (nL + 1), hipMemcpyDeviceToHost, st));
    if (col_idx && np)
        EBVO_HIP(ctx, hipMemcpyAsync(col_idx, s.col_idx.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost, st));
    if (sims && np)
        EBVO_HIP(ctx, hipMemcpyAsync(sims, s.sims.p, sizeof(double) * 4 * np, hipMemcpyDeviceToHost, st));
    if (best && np)
        EBVO_HIP(ctx, hipMemcpyAsync(best, s.best.p, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    if (keep && np)
        EBVO_HIP(ctx, hipMemcpyAsync(keep, s.keep.p, np, hipMemcpyDeviceToHost, st));
    if (left_patches && nL)
    {
        // the pipeline keeps only the normalised banks; the raw left patches (what the reference stores per match,
        // src/Stereo_Matches.cpp:1622) are sampled when they are asked for
        int rc = match_patches_enqueue(ctx, s, ncc_img(s, 0), s.cur_h, s.cur_w, s.cur_w, s.im[0].edges, (int)nL, nullptr, 0,
                                       (float *)s.patches_raw.p, nullptr, nullptr);
        if (rc)
            return rc;
        EBVO_HIP(ctx, hipMemcpyAsync(left_patches, s.patches_raw.p, sizeof(float) * 98 * nL, hipMemcpyDeviceToHost, st));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}




// ---- fixed-scale SIFT descriptors and the descriptor-distance score ---------------------------------------------------
extern "C" int ebvo_sift_descriptors(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, const ebvo_edge *edges,
                                     int n, float *desc)
{
    if (!ctx || !img || n < 0 || (n > 0 && (!edges || !desc)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (n == 0)
        return EBVO_OK;
    const size_t npx = (size_t)h * w, nz = (size_t)n;
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)) || (rc = ebvo_grow(ctx, s, s.sift_img, sizeof(float) * 2 * npx)) ||
        (rc = ebvo_grow(ctx, s, s.scratch_b, sizeof(ebvo_edge) * nz)) ||
        (rc = ebvo_grow(ctx, s, s.sift_f32, sizeof(float) * 256 * nz)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(s.scratch_b.p, edges, sizeof(ebvo_edge) * nz, hipMemcpyHostToDevice, s.stream));
    float *tmp = (float *)s.sift_img.p, *base = tmp + npx;
    if ((rc = sift_base_enqueue(ctx, s, s.im[0].img, h, w, w, tmp, base)) ||
        (rc = sift_descriptors_enqueue(ctx, s, base, h, w, (const ebvo_edge *)s.scratch_b.p, n, (float *)s.sift_f32.p, nullptr)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(desc, s.sift_f32.p, sizeof(float) * 256 * nz, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_sift_min_distances(ebvo_ctx *ctx, const float *left_desc, int nL, const float *cand_desc,
                                       const int32_t *row_ptr, double *dist)
{
    int64_t np = 0;
    if (!ctx || check_csr(row_ptr, nL, &np) || (np > 0 && (!left_desc || !cand_desc || !dist)))
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if (np == 0)
        return EBVO_OK;
    const size_t npz = (size_t)np, nLz = (size_t)nL;
    if ((rc = ebvo_grow(ctx, s, s.sift_f32, sizeof(float) * 256 * (npz + nLz))) ||
        (rc = ebvo_grow(ctx, s, s.sift_desc, 256 * (npz + nLz))) ||
        (rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * (nLz + 1))) ||
        (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * npz)) || (rc = ebvo_grow(ctx, s, s.sift_dist, sizeof(double) * npz)))
        return rc;
    hipStream_t st = s.stream;
    float *fl = (float *)s.sift_f32.p, *fc = fl + 256 * nLz;
    uint8_t *ul = (uint8_t *)s.sift_desc.p, *uc = ul + 256 * nLz;
    EBVO_HIP(ctx, hipMemcpyAsync(fl, left_desc, sizeof(float) * 256 * nLz, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(fc, cand_desc, sizeof(float) * 256 * npz, hipMemcpyHostToDevice, st));
    EBVO_HIP(ctx, hipMemcpyAsync(s.row_ptr.p, row_ptr, sizeof(int32_t) * (nLz + 1), hipMemcpyHostToDevice, st));
    if ((rc = sift_to_u8_enqueue(ctx, s, fl, (int64_t)(256 * (nLz + npz)), ul)) ||
        (rc = match_expand_rows_enqueue(ctx, s, (const int32_t *)s.row_ptr.p, nL, np, (int32_t *)s.pair_left.p)) ||
        (rc = sift_distances_enqueue(ctx, s, ul, uc, (const int32_t *)s.pair_left.p, nullptr, np, EBVO_SIFT_THRESHOLD,
                                     (double *)s.sift_dist.p, nullptr)))
        return rc;
    EBVO_HIP(ctx, hipMemcpyAsync(dist, s.sift_dist.p, sizeof(double) * npz, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

// ---- input side: cv::undistort ---------------------------------------------------------------------------------
extern "C" int ebvo_undistort(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, const double K[4],
                              const double *dist, int n_dist, uint8_t *out, ptrdiff_t out_stride)
{
    if (!ctx || !img || !out || !K || n_dist < 0 || n_dist > 5 || (n_dist > 0 && !dist) || out_stride < w)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    Slot *sp;
    if ((rc = check_size(ctx, h, w)) || (rc = host_slot(ctx, &sp)))
        return rc;
    Slot &s = *sp;
    if ((rc = upload_image(ctx, s, 0, img, h, w, stride)) ||
        (rc = refine_undistort_enqueue(ctx, s, s.im[0].img, w, h, w, K, dist, n_dist, s.im[0].undist_xs, s.im[1].img, w)))
        return rc;
    if (out_stride == (ptrdiff_t)w)
        EBVO_HIP(ctx, hipMemcpyAsync(out, s.im[1].img, (size_t)w * h, hipMemcpyDeviceToHost, s.stream));
    else
        EBVO_HIP(ctx, hipMemcpy2DAsync(out, (size_t)out_stride, s.im[1].img, (size_t)w, (size_t)w, (size_t)h,
                                       hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_stereo_set_undistort(ebvo_ctx *ctx, const ebvo_undistort_params *p)
{
    if (ctx)
        ++ctx->graph_gen; // the coefficients are launch arguments of a captured chain

    if (!ctx || (p && (p->n_dist < 0 || p->n_dist > 5)))
        return EBVO_ERR_ARG;
    for (Slot *s : ctx->slots)
        if (s->in_flight)
            return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (!p)
    {
        ctx->undist_on = false;
        return EBVO_OK;
    }
    for (Slot *s : ctx->slots)
        for (int k = 0; k < 2; ++k)
            if (!s->im[k].raw_base)
            {
                const size_t bytes = (size_t)ctx->max_h * ctx->max_w + 128;
                if (hipMalloc(&s->im[k].raw_base, bytes) != hipSuccess)
                {
                    (void)hipGetLastError();
                    return EBVO_ERR_NOMEM;
                }
                EBVO_HIP(ctx, hipMemsetAsync(s->im[k].raw_base, 0, bytes, s->stream));
                EBVO_HIP(ctx, hipStreamSynchronize(s->stream));
                s->im[k].raw = s->im[k].raw_base + 64;
            }
    ctx->undist = *p;
    ctx->undist_on = true;
    for (Slot *s : ctx->slots) // pairs uploaded before the switch went to img: they must be uploaded again
        s->have_pair = s->have_run = s->have_refined = s->have_final = false;
    return EBVO_OK;
}


// ---- temporal quads against the keyframe ---------------------------------------------------------------------------
extern "C" void ebvo_temporal_default_params(ebvo_temporal_params *p)
{
    if (!p)
        return;
    memset(p, 0, sizeof *p);
    p->cell_size = 15;
    p->grid_radius = 30.0;
    p->orient_thr_deg = 10.0;
    p->ncc_thr = EBVO_NCC_THRESH_TEMPORAL;
    p->stages = 0;
    p->sift_thr = 200.0; // src/Temporal_Matches.cpp:196
    p->bnb_ncc = 0.8;    // :200
    p->bnb_sift = 0.8;   // :204
    ebvo_gn_default_params(&p->gn); // 20, 1e-3, 3.0 (:612, :615)
}

// the final mates of a finalized slot: left edges, right centre edges (carved by ebvo_stereo_finalize)
static void final_mates(Slot &s, const ebvo_edge **fl, const ebvo_edge **fr)
{
    const size_t nz = (size_t)s.result.n_pairs, nLz = (size_t)s.result.n_left + 1;
    *fl = (const ebvo_edge *)s.fin_edges.p + 3 * nz;
    *fr = *fl + nLz;
}

// SIFT descriptors of n stereo mates of the slot's pair: left edges on the undistorted left image, right edges on the
// undistorted right image (the octave base of each image is rebuilt here: one separable blur per image)
static int mate_descriptors(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, const ebvo_edge *d_R, int n, uint8_t *d_descL,
                            uint8_t *d_descR)
{
    const int h = s.cur_h, w = s.cur_w;
    const size_t npx = (size_t)h * w;
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.sift_img, sizeof(float) * 2 * npx)))
        return rc;
    float *tmp = (float *)s.sift_img.p, *base = tmp + npx;
    // a mate's left edge IS its left TOED edge: if ebvo_stereo_finalize ran the SIFT stages on this pair, the descriptors of
    // every left TOED edge are still there and the mates' are picked from them
    const size_t nLz = (size_t)s.result.n_left + 1;
    if (s.sift_left_valid && s.have_final && d_L == (const ebvo_edge *)s.fin_edges.p + 3 * (size_t)s.result.n_pairs)
    {
        if ((rc = sift_gather_enqueue(ctx, s, (const uint8_t *)s.sift_desc.p, (const int32_t *)s.fin_i32.p + 3 * nLz, n, d_descL)))
            return rc;
    }
    else if ((rc = sift_base_enqueue(ctx, s, s.im[0].img, h, w, w, tmp, base)) ||
             (rc = sift_descriptors_enqueue(ctx, s, base, h, w, d_L, n, nullptr, d_descL)))
        return rc;
    if ((rc = sift_base_enqueue(ctx, s, s.im[1].img, h, w, w, tmp, base)) ||
        (rc = sift_descriptors_enqueue(ctx, s, base, h, w, d_R, n, nullptr, d_descR)))
        return rc;
    return EBVO_OK;
}

extern "C" int ebvo_temporal_set_keyframe(ebvo_ctx *ctx, int slot)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_final || s.in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc_f = drain_fetch(ctx, s))
        return rc_f;
    const size_t n = (size_t)s.n_final;
    if (n > ctx->kf_cap)
    {
        EBVO_HIP(ctx, hipDeviceSynchronize());
        for (void *p : {(void *)ctx->kf_L, (void *)ctx->kf_R, (void *)ctx->kf_Ln, (void *)ctx->kf_Rn, (void *)ctx->kf_Lf, (void *)ctx->kf_Rf,
                        (void *)ctx->kf_Ld, (void *)ctx->kf_Rd})
            (void)hipFree(p);
        ctx->kf_L = ctx->kf_R = nullptr;
        ctx->kf_Ln = ctx->kf_Rn = nullptr;
        ctx->kf_Lf = ctx->kf_Rf = nullptr;
        ctx->kf_Ld = ctx->kf_Rd = nullptr;
        ctx->kf_cap = 0;
        const size_t cap = n + n / 4 + 64;
        if (hipMalloc(&ctx->kf_L, sizeof(ebvo_edge) * cap) != hipSuccess || hipMalloc(&ctx->kf_R, sizeof(ebvo_edge) * cap) != hipSuccess ||
            hipMalloc(&ctx->kf_Ln, sizeof(float) * 98 * cap) != hipSuccess || hipMalloc(&ctx->kf_Rn, sizeof(float) * 98 * cap) != hipSuccess ||
            hipMalloc(&ctx->kf_Lf, 2 * cap) != hipSuccess || hipMalloc(&ctx->kf_Rf, 2 * cap) != hipSuccess ||
            hipMalloc(&ctx->kf_Ld, 256 * cap) != hipSuccess || hipMalloc(&ctx->kf_Rd, 256 * cap) != hipSuccess)
        {
            (void)hipGetLastError();
            ctx->last_error = "hipMalloc failed (keyframe store)";
            return EBVO_ERR_NOMEM;
        }
        ctx->kf_cap = cap;
    }
    ctx->kf_n = (int)n;
    {
        // the keyframe's undistorted images stay with it: the photometric refinement of the quads samples them
        const size_t bytes = (size_t)s.cur_h * s.cur_w;
        if (bytes > ctx->kf_img_bytes)
        {
            EBVO_HIP(ctx, hipDeviceSynchronize());
            (void)hipFree(ctx->kf_imgL);
            (void)hipFree(ctx->kf_imgR);
            ctx->kf_imgL = ctx->kf_imgR = nullptr;
            ctx->kf_img_bytes = 0;
            if (hipMalloc(&ctx->kf_imgL, bytes) != hipSuccess || hipMalloc(&ctx->kf_imgR, bytes) != hipSuccess)
            {
                (void)hipGetLastError();
                ctx->last_error = "hipMalloc failed (keyframe images)";
                return EBVO_ERR_NOMEM;
            }
            ctx->kf_img_bytes = bytes;
        }
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->kf_imgL, s.im[0].img, bytes, hipMemcpyDeviceToDevice, s.stream));
        EBVO_HIP(ctx, hipMemcpyAsync(ctx->kf_imgR, s.im[1].img, bytes, hipMemcpyDeviceToDevice, s.stream));
    }
    if (n == 0)
    {
        EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
        return EBVO_OK;
    }
    const ebvo_edge *fl, *fr;
    final_mates(s, &fl, &fr);
    int rc;
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->kf_L, fl, sizeof(ebvo_edge) * n, hipMemcpyDeviceToDevice, s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(ctx->kf_R, fr, sizeof(ebvo_edge) * n, hipMemcpyDeviceToDevice, s.stream));
    // left_edge_patches: raw left image (src/Stereo_Matches.cpp:578, :1621); right_edge_patches: undistorted right (:1622)
    if ((rc = match_patches_enqueue(ctx, s, ncc_img(s, 0), s.cur_h, s.cur_w, s.cur_w, ctx->kf_L, (int)n, nullptr, 0, nullptr, ctx->kf_Ln,
                                    ctx->kf_Lf)) ||
        (rc = match_patches_enqueue(ctx, s, s.im[1].img, s.cur_h, s.cur_w, s.cur_w, ctx->kf_R, (int)n, nullptr, 0, nullptr, ctx->kf_Rn,
                                    ctx->kf_Rf)))
        return rc;
    // left_edge_descriptors (augment_Edge_Data, src/Stereo_Matches.cpp:655-689: the left TOED edge on the undistorted left
    // image) and right_edge_descriptors (finalize_stereo_edge_mates, :1627-1635: the final right edge on the undistorted
    // right image) of every mate
    if ((rc = mate_descriptors(ctx, s, fl, fr, (int)n, ctx->kf_Ld, ctx->kf_Rd)))
        return rc;
    EBVO_HIP(ctx, hipStreamSynchronize(s.stream));
    return EBVO_OK;
}

// get_Temporal_Edge_Pairs_from_Quads after the NCC filter (src/Temporal_Matches.cpp:196-215) on the candidate quads of the
// slot's pair: rp / col / quad_kf / sim_l / keep are the CSR candidate lists, the current-frame mate and the keyframe mate
// of every candidate, its left NCC maximum and the NCC keep flag, as ebvo_temporal_match left them on the device.
static int temporal_chain(ebvo_ctx *ctx, Slot &s, const ebvo_temporal_params &P, const int32_t *rp, const int32_t *col,
                          const int32_t *quad_kf, const double *sim_l, const uint8_t *keep, int n_kf, const ebvo_edge *cfL,
                          const ebvo_edge *cfR, int n_cf, int64_t n_kept, ebvo_temporal_counts *counts)
{
    const int h = s.cur_h, w = s.cur_w;
    const size_t nk1 = (size_t)n_kf + 1, nK = (size_t)(n_kept > 0 ? n_kept : 1), ncz = (size_t)n_cf;
    hipStream_t st = s.stream;
    int rc;
    // one buffer, carved: two quad sets (ping-pong), row bookkeeping, refinement and clustering arrays, the final lists
    struct QuadSet
    {
        int32_t *rp, *cf, *kf;
        double *simL, *siftL, *siftR;
    } A, B;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 63) & ~(size_t)63;
        return o;
    };
    size_t o_set[2][6];
    for (int k = 0; k < 2; ++k)
    {
        o_set[k][0] = carve(sizeof(int32_t) * nk1);
        o_set[k][1] = carve(sizeof(int32_t) * nK);
        o_set[k][2] = carve(sizeof(int32_t) * nK);
        o_set[k][3] = carve(sizeof(double) * nK);
        o_set[k][4] = carve(sizeof(double) * nK);
        o_set[k][5] = carve(sizeof(double) * nK);
    }
    // `order` is indexed like the INPUT list of a selection: the first one selects from all nq candidate quads
    const size_t nq = (size_t)(s.tq_n > 0 ? s.tq_n : 1);
    const size_t o_cnt = carve(sizeof(int32_t) * (nk1 + 1)), o_order = carve(sizeof(int32_t) * (nq > nK ? nq : nK)),
                 o_idx = carve(sizeof(int32_t) * nK),
                 o_ok = carve(3 * nK), o_desc = carve(512 * ncz), o_kfe = carve(sizeof(ebvo_edge) * 2 * nK),
                 o_cfe = carve(sizeof(ebvo_edge) * 2 * nK), o_init = carve(sizeof(double) * 4 * nK),
                 o_disp = carve(sizeof(double) * 4 * nK), o_score = carve(sizeof(double) * 2 * nK), o_valid = carve(3 * nK),
                 o_iters = carve(sizeof(int32_t) * 2 * nK), o_cen = carve(sizeof(ebvo_edge) * 2 * nK),
                 o_centres = carve(sizeof(ebvo_edge) * nK), o_clof = carve(sizeof(int32_t) * nK),
                 o_frp = carve(sizeof(int32_t) * nk1), o_fsrc = carve(sizeof(int32_t) * nK), o_fcf = carve(sizeof(int32_t) * nK),
                 o_fL = carve(sizeof(ebvo_edge) * nK), o_fR = carve(sizeof(ebvo_edge) * nK), o_fd = carve(sizeof(double) * 4 * nK),
                 o_fvalid = carve(nK);
    if ((rc = ebvo_grow(ctx, s, s.tq_chain, off + 64)))
        return rc;
    char *base = (char *)s.tq_chain.p;
    QuadSet *sets[2] = {&A, &B};
    for (int k = 0; k < 2; ++k)
    {
        sets[k]->rp = (int32_t *)(base + o_set[k][0]);
        sets[k]->cf = (int32_t *)(base + o_set[k][1]);
        sets[k]->kf = (int32_t *)(base + o_set[k][2]);
        sets[k]->simL = (double *)(base + o_set[k][3]);
        sets[k]->siftL = (double *)(base + o_set[k][4]);
        sets[k]->siftR = (double *)(base + o_set[k][5]);
    }
    int32_t *cnt = (int32_t *)(base + o_cnt), *order = (int32_t *)(base + o_order), *idx = (int32_t *)(base + o_idx);
    uint8_t *okL = (uint8_t *)(base + o_ok), *okR = okL + nK, *ok = okR + nK;
    uint8_t *cf_Ld = (uint8_t *)(base + o_desc), *cf_Rd = cf_Ld + 256 * ncz;
    auto scan_counts = [&](int32_t *rp_out, int32_t *total, int64_t known = -1) -> int {
        int r = ebvo_device_scan(ctx, s, cnt, rp_out, n_kf, nullptr, 1, n_kf + 1);
        if (r || known >= 0) // (a total the host already holds is not read back: one synchronisation less)
        {
            *total = (int32_t)(known >= 0 ? known : 0);
            return r;
        }
        return read_i32(ctx, s, rp_out + n_kf, total);
    };
    // select rows of `from` (counts / order just formed against from.rp) into `to`
    auto compact = [&](const QuadSet &from, QuadSet &to, int32_t *total, bool with_sift, int64_t known = -1) -> int {
        int r;
        if ((r = scan_counts(to.rp, total, known)) || *total == 0)
            return r;
        if ((r = glue_row_index_enqueue(ctx, s, from.rp, cnt, order, to.rp, n_kf, idx)))
            return r;
        GlueGather g;
        g.i_src[0] = from.cf, g.i_dst[0] = to.cf;
        g.i_src[1] = from.kf, g.i_dst[1] = to.kf;
        g.d_src[0] = from.simL, g.d_dst[0] = to.simL;
        if (with_sift)
        {
            g.d_src[1] = from.siftL, g.d_dst[1] = to.siftL;
            g.d_src[2] = from.siftR, g.d_dst[2] = to.siftR;
        }
        return glue_gather_enqueue(ctx, s, idx, *total, g);
    };
    auto finish_empty = [&]() -> int { // nothing left: an empty final list
        s.tq_final.rp = (int32_t *)(base + o_frp);
        EBVO_HIP(ctx, hipMemsetAsync(s.tq_final.rp, 0, sizeof(int32_t) * nk1, st));
        EBVO_HIP(ctx, hipStreamSynchronize(st));
        s.tq_final.n = 0;
        return EBVO_OK;
    };
    // 0. the quads that passed the NCC filter
    int32_t n0 = 0;
    {
        QuadSet G0{const_cast<int32_t *>(rp), const_cast<int32_t *>(col), const_cast<int32_t *>(quad_kf), const_cast<double *>(sim_l),
                   nullptr, nullptr};
        // (their number came with the NCC counts: n_kept)
        if ((rc = glue_rows_from_flags_enqueue(ctx, s, rp, n_kf, keep, cnt, order)) || (rc = compact(G0, A, &n0, false, n_kept)))
            return rc;
    }
    if (n0 == 0)
        return finish_empty();
    // 1. apply_SIFT_filtering_quads (:471-515): the smaller of the four descriptor distances of the left edges AND of the right
    // edges below the threshold; the distances become the quad's SIFT scores
    if ((rc = mate_descriptors(ctx, s, cfL, cfR, n_cf, cf_Ld, cf_Rd)) ||
        (rc = sift_distances_enqueue(ctx, s, ctx->kf_Ld, cf_Ld, A.kf, A.cf, n0, P.sift_thr, A.siftL, okL)) ||
        (rc = sift_distances_enqueue(ctx, s, ctx->kf_Rd, cf_Rd, A.kf, A.cf, n0, P.sift_thr, A.siftR, okR)) ||
        (rc = sift_and_flags_enqueue(ctx, s, okL, okR, n0, ok)) || (rc = glue_rows_from_flags_enqueue(ctx, s, A.rp, n_kf, ok, cnt, order)))
        return rc;
    int32_t n1 = 0;
    if ((rc = compact(A, B, &n1, true)))
        return rc;
    counts->n_sift = n1;
    if (n1 == 0)
        return finish_empty();
    // 2. apply_best_nearly_best_filtering_quads on the left NCC scores, then on the left SIFT scores (:517-570): rows of two or
    // more quads come out sorted by the score
    int32_t n2 = 0, n3 = 0;
    if ((rc = glue_bnb_enqueue(ctx, s, B.rp, n_kf, B.simL, P.bnb_ncc, 1 | 2, cnt, order)) || (rc = compact(B, A, &n2, true)))
        return rc;
    counts->n_bnb_ncc = n2;
    if ((rc = glue_bnb_enqueue(ctx, s, A.rp, n_kf, A.siftL, P.bnb_sift, 0 | 2, cnt, order)) || (rc = compact(A, B, &n3, true)))
        return rc;
    counts->n_bnb_sift = n3;
    if (n3 == 0)
        return finish_empty();
    // 3. apply_photometric_refinement_quads (:572-634): both cameras of every quad, keyframe image against the current image
    // (undistorted), gradients of the current image; the quad's centres move where the refinement is valid
    const size_t n3z = (size_t)n3;
    ebvo_edge *kfeL = (ebvo_edge *)(base + o_kfe), *kfeR = kfeL + n3z, *cfeL = (ebvo_edge *)(base + o_cfe), *cfeR = cfeL + n3z;
    double *initL = (double *)(base + o_init), *initR = initL + 2 * n3z, *dispL = (double *)(base + o_disp), *dispR = dispL + 2 * n3z;
    double *scoreL = (double *)(base + o_score), *scoreR = scoreL + n3z;
    uint8_t *validL = (uint8_t *)(base + o_valid), *validR = validL + n3z, *valid = validR + n3z;
    int32_t *itersL = (int32_t *)(base + o_iters); // the second camera's counts follow at + n3z (one batch, refine_gn_temporal_enqueue)
    ebvo_edge *cenL = (ebvo_edge *)(base + o_cen), *cenR = cenL + n3z;
    if ((rc = glue_quad_refine_inputs_enqueue(ctx, s, ctx->kf_L, B.kf, cfL, B.cf, n3, kfeL, cfeL, initL)) ||
        (rc = glue_quad_refine_inputs_enqueue(ctx, s, ctx->kf_R, B.kf, cfR, B.cf, n3, kfeR, cfeR, initR)) ||
        // both cameras in one batch: items 0 .. n3 - 1 the left images, n3 .. 2 n3 - 1 the right ones (the arrays are contiguous)
        (rc = refine_gn_temporal_enqueue(ctx, s, ctx->kf_imgL, s.im[0].img, nullptr, h, w, kfeL, cfeL, initL, 2 * (int64_t)n3,
                                         P.gn.max_iter, P.gn.tol, P.gn.huber_delta, dispL, scoreL, validL, itersL, n3, ctx->kf_imgR,
                                         s.im[1].img, nullptr)) ||
        (rc = glue_quad_apply_refine_enqueue(ctx, s, kfeL, cfeL, dispL, validL, kfeR, cfeR, dispR, validR, n3, cenL, cenR, valid)))
        return rc;
    unsigned long long *d_nvalid = (unsigned long long *)(base + o_ok); // (the SIFT flags are no longer needed)
    EBVO_HIP(ctx, hipMemsetAsync(d_nvalid, 0, sizeof *d_nvalid, st));
    if ((rc = match_count_flags_enqueue(ctx, s, valid, n3, d_nvalid)))
        return rc;
    // 4. apply_temporal_edge_clustering_quads (:636-733): EdgeClusterer by orientation on the refined left edges of every row of
    // two or more quads, the right centre and the record of a merged quad from its members
    ebvo_edge *centres = (ebvo_edge *)(base + o_centres);
    int32_t *cluster_of = (int32_t *)(base + o_clof);
    Slot::TqFinal F;
    F.rp = (int32_t *)(base + o_frp);
    int32_t *fsrc = (int32_t *)(base + o_fsrc);
    F.cf = (int32_t *)(base + o_fcf);
    F.L = (ebvo_edge *)(base + o_fL);
    F.R = (ebvo_edge *)(base + o_fR);
    F.ncc = (double *)(base + o_fd);
    int32_t n4 = 0;
    if ((rc = glue_cluster_enqueue(ctx, s, cenL, B.rp, n_kf, 1, 1, cnt, centres, cluster_of)) || (rc = scan_counts(F.rp, &n4)))
        return rc;
    const size_t n4z = (size_t)n4;
    F.sift = F.ncc + n4z;
    F.sL = F.sift + n4z;
    F.sR = F.sL + n4z;
    F.valid = (uint8_t *)(base + o_fvalid);
    if ((rc = glue_quad_cluster_post_enqueue(ctx, s, B.rp, n_kf, cnt, cluster_of, centres, cenL, cenR, F.rp, F.L, F.R, fsrc)))
        return rc;
    {
        GlueGather g;
        g.i_src[0] = B.cf, g.i_dst[0] = F.cf;
        g.d_src[0] = B.simL, g.d_dst[0] = F.ncc;
        g.d_src[1] = B.siftL, g.d_dst[1] = F.sift;
        g.d_src[2] = scoreL, g.d_dst[2] = F.sL;
        g.d_src[3] = scoreR, g.d_dst[3] = F.sR;
        g.b_src[0] = valid, g.b_dst[0] = F.valid;
        if ((rc = glue_gather_enqueue(ctx, s, fsrc, n4, g)))
            return rc;
    }
    unsigned long long nvalid = 0;
    EBVO_HIP(ctx, hipMemcpyAsync(s.h_result, d_nvalid, sizeof nvalid, hipMemcpyDeviceToHost, st));
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    memcpy(&nvalid, s.h_result, sizeof nvalid);
    counts->n_refined_valid = (int64_t)nvalid;
    counts->n_final = n4;
    s.tq_final.rp = F.rp;
    s.tq_final.cf = F.cf;
    s.tq_final.L = F.L;
    s.tq_final.R = F.R;
    s.tq_final.ncc = F.ncc;
    s.tq_final.sift = F.sift;
    s.tq_final.sL = F.sL;
    s.tq_final.sR = F.sR;
    s.tq_final.valid = F.valid;
    s.tq_final.n = n4;
    return EBVO_OK;
}

// Candidate quads + NCC of the slot's final mates against the keyframe's.  The buffers indexed by candidate quad are sized
// for a CAPACITY (s.tq_cap: what the last frame needed, with headroom), every kernel takes the number of quads from the
// device (the last entry of the scanned row offsets), and the two counts -- candidates, kept -- travel to page-locked memory
// behind the last kernel: no host synchronisation between the stages.  Only a first frame (no capacity yet) or a frame
// with more candidates than the capacity reads the count back first.
static int temporal_stage0_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_temporal_params *p, bool size_first)
{
    const int n_kf = ctx->kf_n, n_cf = s.n_final, h = s.cur_h, w = s.cur_w;
    int rc;
    const size_t nk1 = (size_t)n_kf + 1;
    if ((rc = ebvo_grow(ctx, s, s.tq_i32, sizeof(int32_t) * (2 * nk1 + 2) + 16)))
        return rc;
    int32_t *cnt = (int32_t *)s.tq_i32.p, *rp = cnt + nk1;
    EBVO_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(int32_t) * 2 * nk1, s.stream));
    const ebvo_edge *cfL, *cfR;
    final_mates(s, &cfL, &cfR);
    const int cell = p->cell_size, gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    const int sr = (int)ceil(p->grid_radius / cell);
    if ((rc = ebvo_grow(ctx, s, s.tq_cells, match_temporal_grid_bytes(n_cf, gw * gh))))
        return rc;
    void *grid = s.tq_cells.p;
    if ((rc = match_temporal_cells_enqueue(ctx, s, cfL, cfR, n_cf, cell, gw, gh, grid)) ||
        (rc = match_temporal_candidates_enqueue(ctx, s, ctx->kf_L, ctx->kf_R, n_kf, cfL, cfR, grid, n_cf, cell, sr, gw, gh,
                                                p->orient_thr_deg, cnt, nullptr, nullptr, 0)) ||
        (rc = ebvo_device_scan(ctx, s, cnt, rp, n_kf, nullptr, 1, n_kf + 1)))
        return rc;
    if (size_first)
    {
        int32_t nq = 0;
        if ((rc = read_i32(ctx, s, rp + n_kf, &nq)))
            return rc;
        s.tq_cap = (int64_t)nq + nq / 4 + 1024;
    }
    const size_t capz = (size_t)s.tq_cap, ncz = (size_t)n_cf;
    if ((rc = ebvo_grow(ctx, s, s.tq_cols, sizeof(int32_t) * 2 * capz)) || (rc = ebvo_grow(ctx, s, s.tq_f64, sizeof(double) * 2 * capz)) ||
        (rc = ebvo_grow(ctx, s, s.tq_u8, 2 * ncz + capz + 64)) || (rc = ebvo_grow(ctx, s, s.patches_raw, sizeof(float) * 98 * ncz)) ||
        (rc = ebvo_grow(ctx, s, s.patches_norm, sizeof(float) * 98 * ncz)) || (rc = ebvo_grow(ctx, s, s.patches_flag, 2 * ncz)))
        return rc;
    int32_t *col = (int32_t *)s.tq_cols.p, *quad_kf = col + capz;
    double *sim_l = (double *)s.tq_f64.p, *sim_r = sim_l + capz;
    uint8_t *flagR = (uint8_t *)s.tq_u8.p, *keep = flagR + ((2 * ncz + 63) & ~(size_t)63);
    float *cfLn = (float *)s.patches_norm.p, *cfRn = (float *)s.patches_raw.p;
    uint8_t *cfLf = (uint8_t *)s.patches_flag.p;
    const int32_t *d_nq = rp + n_kf;
    if ((rc = match_temporal_candidates_enqueue(ctx, s, ctx->kf_L, ctx->kf_R, n_kf, cfL, cfR, grid, n_cf, cell, sr, gw, gh,
                                                p->orient_thr_deg, nullptr, rp, col, s.tq_cap)) ||
        (rc = match_expand_rows_enqueue(ctx, s, rp, n_kf, s.tq_cap, quad_kf)) ||
        (rc = match_patches_enqueue(ctx, s, ncc_img(s, 0), h, w, w, cfL, n_cf, nullptr, 0, nullptr, cfLn, cfLf)) ||
        (rc = match_patches_enqueue(ctx, s, s.im[1].img, h, w, w, cfR, n_cf, nullptr, 0, nullptr, cfRn, flagR)) ||
        (rc = match_ncc_quads_indexed_enqueue(ctx, s, ctx->kf_Ln, ctx->kf_Lf, ctx->kf_Rn, ctx->kf_Rf, cfLn, cfLf, cfRn, flagR, quad_kf,
                                              col, s.tq_cap, p->ncc_thr, sim_l, sim_r, keep, d_nq)) ||
        (rc = match_count_flags_enqueue(ctx, s, keep, s.tq_cap, s.d_tq_tot + 1, d_nq)))
        return rc;
    // candidate count (an int32 of the row offsets) widened into the first word on the device side of the copy below
    EBVO_HIP(ctx, hipMemsetAsync(s.d_tq_tot, 0, sizeof(unsigned long long), s.stream));
    EBVO_HIP(ctx, hipMemcpyAsync(s.d_tq_tot, d_nq, sizeof(int32_t), hipMemcpyDeviceToDevice, s.stream)); // little-endian low word
    EBVO_HIP(ctx, hipMemcpyAsync(s.h_tq_tot, s.d_tq_tot, sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, s.stream));
    EBVO_HIP(ctx, hipEventRecord(s.ev_tq, s.stream));
    return EBVO_OK;
}

extern "C" int ebvo_temporal_match_submit(ebvo_ctx *ctx, int slot, const ebvo_temporal_params *p)
{
    Slot *sp;
    if (!p || p->cell_size < 1 || !(p->grid_radius >= 0) || !(p->orient_thr_deg >= 0) || (p->stages & ~1) ||
        (p->stages && (!(p->sift_thr > 0) || !(p->bnb_ncc >= 0) || !(p->bnb_sift >= 0) || p->gn.max_iter < 1 || !(p->gn.tol >= 0) ||
                       !(p->gn.huber_delta > 0))) ||
        get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_final || s.in_flight || s.fin_in_flight || s.tq_in_flight || ctx->kf_n < 0)
    {
        ctx->last_error = !s.have_final ? "the slot holds no final stereo mates (ebvo_stereo_finalize first)"
                          : ctx->kf_n < 0 ? "no keyframe (ebvo_temporal_set_keyframe first)"
                                          : "the slot has work in flight (wait for it first)";
        return EBVO_ERR_STATE;
    }
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc_f = drain_fetch(ctx, s))
        return rc_f;
    s.tq_n = -1;
    s.tq_n_kf = ctx->kf_n;
    s.tq_final = Slot::TqFinal();
    s.tq_params = *p;
    s.tq_empty = ctx->kf_n == 0 || s.n_final == 0;
    s.h_tq_tot[0] = s.h_tq_tot[1] = 0;
    if (s.tq_empty)
    {
        // no quads at all: zeroed row offsets, nothing else to run
        const size_t nk1 = (size_t)ctx->kf_n + 1;
        int rc = ebvo_grow(ctx, s, s.tq_i32, sizeof(int32_t) * (2 * nk1 + 2) + 16);
        if (rc)
            return rc;
        EBVO_HIP(ctx, hipMemsetAsync(s.tq_i32.p, 0, sizeof(int32_t) * 2 * nk1, s.stream));
        EBVO_HIP(ctx, hipEventRecord(s.ev_tq, s.stream));
        s.tq_in_flight = true;
        return EBVO_OK;
    }
    int rc = temporal_stage0_enqueue(ctx, s, p, s.tq_cap == 0);
    if (rc)
    {
        (void)hipStreamSynchronize(s.stream);
        return rc;
    }
    s.tq_in_flight = true;
    return EBVO_OK;
}

extern "C" int ebvo_temporal_match_wait(ebvo_ctx *ctx, int slot, ebvo_temporal_counts *counts)
{
    Slot *sp;
    if (!counts || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    s.tq_in_flight = false;
    memset(counts, 0, sizeof *counts);
    const ebvo_temporal_params &P = s.tq_params;
    const int n_kf = s.tq_n_kf, n_cf = s.n_final;
    counts->n_kf = n_kf;
    counts->n_cf = n_cf;
    EBVO_HIP(ctx, hipEventSynchronize(s.ev_tq));
    const size_t nk1 = (size_t)n_kf + 1;
    int32_t *rp = (int32_t *)s.tq_i32.p + nk1;
    if (s.tq_empty)
    {
        s.tq_n = 0;
        if (P.stages)
        {
            s.tq_final.rp = rp;
            s.tq_final.n = 0;
        }
        return EBVO_OK;
    }
    int64_t nq = (int64_t)s.h_tq_tot[0];
    if (nq > s.tq_cap)
    {
        // more candidate quads than the buffers were sized for: size them for this frame and run the stages again
        s.tq_cap = 0;
        int rc = temporal_stage0_enqueue(ctx, s, &P, true);
        if (rc)
        {
            (void)hipStreamSynchronize(s.stream);
            return rc;
        }
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_tq));
        nq = (int64_t)s.h_tq_tot[0];
    }
    const int64_t kept = (int64_t)s.h_tq_tot[1];
    counts->n_candidates = nq;
    counts->n_kept = kept;
    s.tq_n = nq;
    s.tq_final.n = -1;
    if (nq == 0)
    {
        if (P.stages)
        {
            s.tq_final.rp = rp; // the scanned counts: all zero
            s.tq_final.n = 0;
        }
        return EBVO_OK;
    }
    if (P.stages == 0)
        return EBVO_OK;
    const size_t capz = (size_t)s.tq_cap, ncz = (size_t)n_cf;
    const ebvo_edge *cfL, *cfR;
    final_mates(s, &cfL, &cfR);
    const int32_t *col = (const int32_t *)s.tq_cols.p, *quad_kf = col + capz;
    const double *sim_l = (const double *)s.tq_f64.p;
    const uint8_t *keep = (const uint8_t *)s.tq_u8.p + ((2 * ncz + 63) & ~(size_t)63);
    return temporal_chain(ctx, s, P, rp, col, quad_kf, sim_l, keep, n_kf, cfL, cfR, n_cf, kept, counts);
}

extern "C" int ebvo_temporal_match(ebvo_ctx *ctx, int slot, const ebvo_temporal_params *p, ebvo_temporal_counts *counts)
{
    if (!counts)
        return EBVO_ERR_ARG;
    int rc = ebvo_temporal_match_submit(ctx, slot, p);
    return rc ? rc : ebvo_temporal_match_wait(ctx, slot, counts);
}

extern "C" int ebvo_temporal_fetch_final(ebvo_ctx *ctx, int slot, int32_t *row_ptr, int32_t *cf_index, ebvo_edge *left,
                                         ebvo_edge *right, double *ncc_left, double *sift_left, double *score_left,
                                         double *score_right, uint8_t *valid)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (s.tq_final.n < 0 || s.in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const Slot::TqFinal &F = s.tq_final;
    const size_t nk1 = (size_t)s.tq_n_kf + 1, n = (size_t)F.n;
    hipStream_t st = s.stream;
    if (row_ptr)
        EBVO_HIP(ctx, hipMemcpyAsync(row_ptr, F.rp, sizeof(int32_t) * nk1, hipMemcpyDeviceToHost, st));
    if (n)
    {
        const struct
        {
            void *dst;
            const void *src;
            size_t bytes;
        } copies[] = {{cf_index, F.cf, sizeof(int32_t) * n}, {left, F.L, sizeof(ebvo_edge) * n}, {right, F.R, sizeof(ebvo_edge) * n},
                      {ncc_left, F.ncc, sizeof(double) * n}, {sift_left, F.sift, sizeof(double) * n},
                      {score_left, F.sL, sizeof(double) * n}, {score_right, F.sR, sizeof(double) * n}, {valid, F.valid, n}};
        for (const auto &c : copies)
            if (c.dst)
                EBVO_HIP(ctx, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, st));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

extern "C" int ebvo_temporal_fetch(ebvo_ctx *ctx, int slot, int32_t *row_ptr, int32_t *col_idx, double *sim_left, double *sim_right,
                                   uint8_t *keep)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (s.tq_n < 0 || s.in_flight || s.tq_in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nk1 = (size_t)s.tq_n_kf + 1, nq = (size_t)s.tq_n, ncz = (size_t)s.n_final, capz = (size_t)s.tq_cap;
    hipStream_t st = s.stream;
    if (row_ptr)
        EBVO_HIP(ctx, hipMemcpyAsync(row_ptr, (const int32_t *)s.tq_i32.p + nk1, sizeof(int32_t) * nk1, hipMemcpyDeviceToHost, st));
    if (nq)
    {
        if (col_idx)
            EBVO_HIP(ctx, hipMemcpyAsync(col_idx, s.tq_cols.p, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, st));
        if (sim_left)
            EBVO_HIP(ctx, hipMemcpyAsync(sim_left, s.tq_f64.p, sizeof(double) * nq, hipMemcpyDeviceToHost, st));
        if (sim_right)
            EBVO_HIP(ctx, hipMemcpyAsync(sim_right, (const double *)s.tq_f64.p + capz, sizeof(double) * nq, hipMemcpyDeviceToHost, st));
        if (keep)
            EBVO_HIP(ctx, hipMemcpyAsync(keep, (const uint8_t *)s.tq_u8.p + ((2 * ncz + 63) & ~(size_t)63), nq, hipMemcpyDeviceToHost, st));
    }
    EBVO_HIP(ctx, hipStreamSynchronize(st));
    return EBVO_OK;
}

// the slot's page-locked result arena holds at least `bytes` (a re-allocation waits for copies still heading into the old one)
static int ensure_arena(ebvo_ctx *ctx, Slot &s, size_t bytes)
{
    if (bytes <= s.h_arena_bytes)
        return EBVO_OK;
    if (s.fetch_pending)
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    if (s.h_arena)
        (void)hipHostFree(s.h_arena);
    s.h_arena = nullptr;
    s.h_arena_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&s.h_arena, want) != hipSuccess)
    {
        (void)hipGetLastError();
        ctx->last_error = "hipHostMalloc failed (page-locked result staging)";
        return EBVO_ERR_NOMEM;
    }
    s.h_arena_bytes = want;
    return EBVO_OK;
}

// ---- results through page-locked memory, no staging copy on the host ------------------------------------------------
extern "C" int ebvo_stereo_fetch_begin(ebvo_ctx *ctx, int slot, int what)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp) || what <= 0 || (what & ~EBVO_FETCH_ALL))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_run || s.in_flight)
        return EBVO_ERR_STATE;
    if ((what & EBVO_FETCH_SIMS) && !s.have_sims)
    {
        ctx->last_error = "the pair was submitted with EBVO_PAIR_NO_SIMS: the four scores were not stored";
        return EBVO_ERR_STATE;
    }
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = (size_t)s.result.n_left, nR = (size_t)s.result.n_right, np = (size_t)s.result.n_pairs;
    const size_t sizes[7] = {(what & EBVO_FETCH_EDGES) ? sizeof(ebvo_edge) * nL : 0,
                             (what & EBVO_FETCH_EDGES) ? sizeof(ebvo_edge) * nR : 0,
                             (what & EBVO_FETCH_CSR) ? sizeof(int32_t) * (nL + 1) : 0,
                             (what & EBVO_FETCH_CSR) ? sizeof(int32_t) * np : 0,
                             (what & EBVO_FETCH_SIMS) ? sizeof(double) * 4 * np : 0,
                             (what & EBVO_FETCH_BEST) ? sizeof(double) * np : 0,
                             (what & EBVO_FETCH_KEEP) ? np : 0};
    size_t total = 0;
    for (int k = 0; k < 7; ++k)
    {
        s.fetch_off[k] = total;
        total += (sizes[k] + 63) & ~(size_t)63;
    }
    if (int rc_arena = ensure_arena(ctx, s, total))
        return rc_arena;
    const void *src[7] = {s.im[0].edges, s.im[1].edges, s.row_ptr.p, s.col_idx.p, s.sims.p, s.best.p, s.keep.p};
    char *base = static_cast<char *>(s.h_arena);
    // One copy stream for all slots: the pair is complete (the host has waited for it), so the copies need no ordering
    // against kernels, and on a stream of their own they never sit in front of another pair's kernels in a hardware queue.
    // The event marks the end of THIS slot's copies for ebvo_stereo_fetch_end and for the slot's next submission.
    if (!ctx->copy_stream)
        EBVO_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 7; ++k)
        if (sizes[k])
            EBVO_HIP(ctx, hipMemcpyAsync(base + s.fetch_off[k], src[k], sizes[k], hipMemcpyDeviceToHost, ctx->copy_stream));
    EBVO_HIP(ctx, hipEventRecord(s.ev_rebind, ctx->copy_stream));
    s.fetch_what = what;
    s.fetch_compact = false;
    s.fetch_pending = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch_end(ebvo_ctx *ctx, int slot, ebvo_stereo_view *view)
{
    Slot *sp;
    if (!view || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    // (no have_run test: an asynchronous upload of the slot's NEXT images may already have been enqueued -- the copies started by
    // _begin read the result buffers, which only the next submission overwrites, and that clears fetch_what)
    if (s.in_flight || !s.fetch_what || s.fetch_compact)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (s.fetch_pending)
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    s.fetch_pending = false;
    const char *base = static_cast<const char *>(s.h_arena);
    const int what = s.fetch_what;
    memset(view, 0, sizeof *view);
    view->n_left = s.result.n_left;
    view->n_right = s.result.n_right;
    view->n_pairs = s.result.n_pairs;
    if (what & EBVO_FETCH_EDGES)
    {
        view->left = reinterpret_cast<const ebvo_edge *>(base + s.fetch_off[0]);
        view->right = reinterpret_cast<const ebvo_edge *>(base + s.fetch_off[1]);
    }
    if (what & EBVO_FETCH_CSR)
    {
        view->row_ptr = reinterpret_cast<const int32_t *>(base + s.fetch_off[2]);
        view->col_idx = reinterpret_cast<const int32_t *>(base + s.fetch_off[3]);
    }
    if (what & EBVO_FETCH_SIMS)
        view->sims = reinterpret_cast<const double *>(base + s.fetch_off[4]);
    if (what & EBVO_FETCH_BEST)
        view->best = reinterpret_cast<const double *>(base + s.fetch_off[5]);
    if (what & EBVO_FETCH_KEEP)
        view->keep = reinterpret_cast<const uint8_t *>(base + s.fetch_off[6]);
    return EBVO_OK;
}

// ---- compact results: what a consumer of the NCC stage reads, in fewer bytes (round 4) -------------------------------------
// The 32-byte ABI record of an edge carries its index (= its position in the list) and padding; (x, y) as 16 bytes and the
// orientation as a separate array on demand halve the edge traffic, and the keep flag of a pair is one bit.  best stays fp64.
namespace
{
__global__ __launch_bounds__(256) void pack_results_kernel(const ebvo_edge *__restrict__ L, int nL, const ebvo_edge *__restrict__ R,
                                                           int nR, const uint8_t *__restrict__ keep, int64_t np,
                                                           double2 *__restrict__ xyL, double2 *__restrict__ xyR,
                                                           double *__restrict__ thL, double *__restrict__ thR,
                                                           uint32_t *__restrict__ bits)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nL; i += stride)
    {
        const ebvo_edge e = L[i];
        xyL[i] = make_double2(e.x, e.y);
        if (thL)
            thL[i] = e.theta;
    }
    for (int64_t i = t0; i < nR; i += stride)
    {
        const ebvo_edge e = R[i];
        xyR[i] = make_double2(e.x, e.y);
        if (thR)
            thR[i] = e.theta;
    }
    if (bits) // one 64-lane ballot = two words; every wave walks whole 64-pair groups (uniform trip count)
    {
        const int64_t groups = (np + 63) >> 6;
        const int64_t wave = t0 >> 6, waves = stride >> 6;
        const int lane = threadIdx.x & 63;
        for (int64_t g = wave; g < groups; g += waves)
        {
            const int64_t k = (g << 6) + lane;
            const unsigned long long m = __ballot(k < np && keep[k] != 0);
            if (lane == 0)
            {
                bits[2 * g] = (uint32_t)m;
                bits[2 * g + 1] = (uint32_t)(m >> 32);
            }
        }
    }
}
} // namespace

extern "C" int ebvo_stereo_fetch_compact_begin(ebvo_ctx *ctx, int slot, int what)
{
    Slot *sp;
    if (get_slot(ctx, slot, &sp) || what <= 0 || (what & ~EBVO_COMPACT_ALL))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (!s.have_run || s.in_flight)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = (size_t)s.result.n_left, nR = (size_t)s.result.n_right, np = (size_t)s.result.n_pairs;
    s.fetch_packed = false;
    if (s.have_pack && (!(what & EBVO_COMPACT_THETA) || (s.params.reserved & EBVO_PAIR_PUSH_THETA)))
    {
        // the pair's own chain has packed the arrays (EBVO_PAIR_PACK): ONE copy of the block, sized by the actual counts
        const PackLayout lay = pack_layout(nL, nR, np, s.params.reserved & EBVO_PAIR_PUSH_THETA);
        if (int rc_arena = ensure_arena(ctx, s, lay.total))
            return rc_arena;
        if (!ctx->copy_stream)
            EBVO_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        EBVO_HIP(ctx, hipMemcpyAsync(s.h_arena, s.fetch_pack.p, lay.total, hipMemcpyDeviceToHost, ctx->copy_stream));
        EBVO_HIP(ctx, hipEventRecord(s.ev_rebind, ctx->copy_stream));
        s.fetch_what = what;
        s.fetch_compact = true;
        s.fetch_packed = true;
        s.fetch_packed_theta = (s.params.reserved & EBVO_PAIR_PUSH_THETA) != 0;
        s.fetch_result = s.result;
        s.fetch_pending = true;
        return EBVO_OK;
    }
    const size_t nwords = 2 * ((np + 63) >> 6);
    const bool xy = what & EBVO_COMPACT_XY, th = what & EBVO_COMPACT_THETA, kb = what & EBVO_COMPACT_KEEP_BITS;
    const size_t sizes[8] = {xy ? 16 * nL : 0, xy ? 16 * nR : 0, th ? 8 * nL : 0, th ? 8 * nR : 0,
                             (what & EBVO_COMPACT_CSR) ? sizeof(int32_t) * (nL + 1) : 0,
                             (what & EBVO_COMPACT_CSR) ? sizeof(int32_t) * np : 0,
                             (what & EBVO_COMPACT_BEST) ? sizeof(double) * np : 0, kb ? sizeof(uint32_t) * nwords : 0};
    size_t total = 0;
    for (int k = 0; k < 8; ++k)
    {
        s.fetch_off[k] = total;
        total += (sizes[k] + 63) & ~(size_t)63;
    }
    if (int rc_arena = ensure_arena(ctx, s, total))
        return rc_arena;
    if (!ctx->copy_stream)
        EBVO_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // device staging of the packed arrays, laid out like the first four and the last array of the arena
    size_t poff[5], ptotal = 0;
    const size_t psizes[5] = {sizes[0], sizes[1], sizes[2], sizes[3], sizes[7]};
    for (int k = 0; k < 5; ++k)
    {
        poff[k] = ptotal;
        ptotal += (psizes[k] + 63) & ~(size_t)63;
    }
    if (ptotal > s.fetch_pack.bytes && s.fetch_pending) // (re)allocation: the previous copies out of the old buffer must be over
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    int rc;
    if (ptotal && (rc = ebvo_grow(ctx, s, s.fetch_pack, ptotal)))
        return rc;
    char *pk = static_cast<char *>(s.fetch_pack.p);
    if (xy || th || kb)
    {
        const size_t work = nL > nR ? nL : nR;
        const size_t most = work > ((np + 63) >> 6) * 64 ? work : ((np + 63) >> 6) * 64;
        int blocks = (int)((most + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
        // the pair is complete (the host has waited for it) and the copy stream orders the kernel behind the previous copies
        // out of the staging buffer
        hipLaunchKernelGGL(pack_results_kernel, dim3(blocks), dim3(256), 0, ctx->copy_stream, (const ebvo_edge *)s.im[0].edges,
                           xy || th ? (int)nL : 0, (const ebvo_edge *)s.im[1].edges, xy || th ? (int)nR : 0,
                           (const uint8_t *)s.keep.p, kb ? (int64_t)np : 0, (double2 *)(pk + poff[0]), (double2 *)(pk + poff[1]),
                           th ? (double *)(pk + poff[2]) : nullptr, th ? (double *)(pk + poff[3]) : nullptr,
                           kb ? (uint32_t *)(pk + poff[4]) : nullptr);
        EBVO_HIP(ctx, hipGetLastError());
    }
    const void *src[8] = {pk + poff[0], pk + poff[1], pk + poff[2], pk + poff[3], s.row_ptr.p, s.col_idx.p, s.best.p, pk + poff[4]};
    char *base = static_cast<char *>(s.h_arena);
    for (int k = 0; k < 8; ++k)
        if (sizes[k])
            EBVO_HIP(ctx, hipMemcpyAsync(base + s.fetch_off[k], src[k], sizes[k], hipMemcpyDeviceToHost, ctx->copy_stream));
    EBVO_HIP(ctx, hipEventRecord(s.ev_rebind, ctx->copy_stream));
    s.fetch_what = what;
    s.fetch_compact = true;
    s.fetch_pending = true;
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch_compact_end(ebvo_ctx *ctx, int slot, ebvo_stereo_compact_view *view)
{
    Slot *sp;
    if (!view || get_slot(ctx, slot, &sp))
        return EBVO_ERR_ARG;
    Slot &s = *sp;
    if (s.in_flight || !s.fetch_what || !s.fetch_compact) // (see ebvo_stereo_fetch_end)
        return EBVO_ERR_STATE;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    if (s.fetch_pending)
        EBVO_HIP(ctx, hipEventSynchronize(s.ev_rebind));
    s.fetch_pending = false;
    const char *base = static_cast<const char *>(s.h_arena);
    const int what = s.fetch_what;
    if (s.fetch_packed)
    {
        const bool theta = s.fetch_packed_theta;
        const PackLayout lay = pack_layout((size_t)s.fetch_result.n_left, (size_t)s.fetch_result.n_right, (size_t)s.fetch_result.n_pairs, theta);
        fill_compact_view(view, base, lay, s.fetch_result, theta);
        if (!(what & EBVO_COMPACT_XY))
            view->left_xy = view->right_xy = nullptr;
        if (!(what & EBVO_COMPACT_THETA))
            view->left_theta = view->right_theta = nullptr;
        if (!(what & EBVO_COMPACT_CSR))
            view->row_ptr = view->col_idx = nullptr;
        if (!(what & EBVO_COMPACT_BEST))
            view->best = nullptr;
        if (!(what & EBVO_COMPACT_KEEP_BITS))
            view->keep_bits = nullptr;
        return EBVO_OK;
    }
    memset(view, 0, sizeof *view);
    view->n_left = s.result.n_left;
    view->n_right = s.result.n_right;
    view->n_pairs = s.result.n_pairs;
    view->n_matches = s.result.n_matches;
    if (what & EBVO_COMPACT_XY)
    {
        view->left_xy = reinterpret_cast<const double *>(base + s.fetch_off[0]);
        view->right_xy = reinterpret_cast<const double *>(base + s.fetch_off[1]);
    }
    if (what & EBVO_COMPACT_THETA)
    {
        view->left_theta = reinterpret_cast<const double *>(base + s.fetch_off[2]);
        view->right_theta = reinterpret_cast<const double *>(base + s.fetch_off[3]);
    }
    if (what & EBVO_COMPACT_CSR)
    {
        view->row_ptr = reinterpret_cast<const int32_t *>(base + s.fetch_off[4]);
        view->col_idx = reinterpret_cast<const int32_t *>(base + s.fetch_off[5]);
    }
    if (what & EBVO_COMPACT_BEST)
        view->best = reinterpret_cast<const double *>(base + s.fetch_off[6]);
    if (what & EBVO_COMPACT_KEEP_BITS)
        view->keep_bits = reinterpret_cast<const uint32_t *>(base + s.fetch_off[7]);
    return EBVO_OK;
}

extern "C" int ebvo_stereo_fetch(ebvo_ctx *ctx, ebvo_edge *left, ebvo_edge *right, int32_t *row_ptr,
                                 int32_t *col_idx, double *sims, double *best, uint8_t *keep, float *left_patches)
{
    return ebvo_stereo_fetch_slot(ctx, 0, left, right, row_ptr, col_idx, sims, best, keep, left_patches);
}

// ------------------------------------------------------------------------------------------
extern "C" int ebvo_profile_enable(ebvo_ctx *ctx, int on)
{
    if (!ctx)
        return EBVO_ERR_ARG;
    int rc = prof_drain(ctx);
    ctx->prof = on != 0;
    ctx->prof_every = on > 1 ? on : 1; // on = N > 1: sample every N-th pair submitted to the device pipeline
    ctx->prof_submits = 0;
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

extern "C" int ebvo_debug_set(ebvo_ctx *ctx, int key, int value)
{
    if (!ctx || value < 0)
        return EBVO_ERR_ARG;
    ++ctx->graph_gen;
    if (key == 10 && value <= 1)
        ctx->use_graphs = value; // the pair chain as a hipGraph (default) or as direct launches
    else if (key == 14 && value <= 1)
        ctx->no_prep = value;
    else if (key == 11 || key == 12)
        ctx->exact_blocks[key - 11] = value;
    else if (key == 15)
        ctx->repeat_mask = value;
    else if (key == 16)
        ctx->stop_stage = value;
    else if (key == 17)
        ctx->ncc_blocks = value;
    else if (key == 18)
        ctx->small_div = value;
    else if (key == 19 && value <= EBVO_TOTAL_PARTS)
        ctx->cand_blocks = value;
    else if (key == 13 && value <= 1)
        ctx->ingest_stream = value;
    else if (key == 0)
        ctx->wait_attempts = value;
    else if (key == 1)
        ctx->force_overflow = value;
    else if (key == 2)
        ctx->lanes = value; // 0 = one stream per slot whatever their number
    else if (key == 3)
        ctx->prof_only = value - 1; // 0 = every stage; id + 1 = that stage alone: no event markers between the other kernels
    else if (key == 4 && value <= 1)
        ctx->gn_no_rows = value; // refinement launch layout (same bits either way, tests/test_gpu_refine.py)
    else if (key == 5)
        ctx->gn_rows_below = value;
    else if (key == 9)
        ctx->gn_persist_blocks = value;
    else if (key == 8 && (value == 2 || value == 3))
        ctx->gn_persist_waves = value;
    else if (key == 7 && value <= 1)
        ctx->gn_per_iteration_rows = value; // the stereo refinement's row layout as a launch per iteration (A/B and parity tests)
    else if (key == 6 && value >= 1)
    {
        for (Slot *sl : ctx->slots) // test hook: the next temporal match of every slot finds its quad buffers too small
            if (sl->tq_cap > 0 && !sl->tq_in_flight)
                sl->tq_cap = value;
    }
    else
        return EBVO_ERR_ARG;
    return EBVO_OK;
}

extern "C" int ebvo_fp64_peak(ebvo_ctx *ctx, int iters, double *tflops_muladd, double *tflops_fma)
{
    if (!ctx || iters <= 0)
        return EBVO_ERR_ARG;
    EBVO_HIP(ctx, hipSetDevice(ctx->device));
    return misc_fp64_peak(ctx, *ctx->slots[0], iters, tflops_muladd, tflops_fma);
}

// ebvo_internal.h -- shared declarations of the HIP implementation behind include/ebvo_hip.h.
#ifndef EBVO_INTERNAL_H
#define EBVO_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ebvo_hip.h"

enum KernelId
{
    K_CONV = 0,
    K_NMS,
    K_ROWSCAN,
    K_COMPACT,
    K_FINALIZE,
    K_EXACT_CENTRE,
    K_EXACT_MAGS,
    K_BOXES,
    K_LINES,
    K_CAND_COUNT,
    K_SCAN,
    K_CAND_FILL,
    K_PATCHES,
    K_NCC_PAIRS,
    K_NCC_STORED,
    K_MISC,
    K_SOBEL,
    K_GN_REFINE,
    K_SIFT,
    K_NUM
};
static_assert(K_NUM <= EBVO_MAX_KERNELS, "grow EBVO_MAX_KERNELS");

extern const char *const g_kernel_names[K_NUM];

// Interpolated-grid planes written by the convolution, per image.  Each plane holds the four sub-pixel
// phases as separate H x W sub-planes: [sy][sx][H][W] (toed_kernels.hip: midx).
enum
{
    PL_IX = 0,
    PL_IY,
    PL_MAG,
    PL_TOX, // third-order orientation vector, unnormalised
    PL_TOY,
    PL_NUM
};

struct GrowBuf
{
    void *p = nullptr;
    size_t bytes = 0;
};

// Per-image device workspace.
struct ImageWS
{
    uint8_t *img = nullptr;     // h*w, tightly packed; = img_base + 64: 64 readable bytes before and after (wide loads)
    uint8_t *img_base = nullptr;
    uint8_t *raw_base = nullptr, *raw = nullptr; // the image as uploaded when the context undistorts (ebvo_stereo_set_undistort):
                                                 // TOED and the refinement run on img (undistorted), NCC samples raw
    double *undist_xs = nullptr;                 // [max_w] row sequence of the undistortion map
    uint16_t *pix2 = nullptr;   // row-pair image: pix2[y*w + x] = img(y, x) | img(y + 1, x) << 8 (one 4-byte load per bilinear sample)
    double *maps = nullptr;     // PL_NUM planes of 4 x H x W
    uint8_t *flag = nullptr;    // 2H x 2W: 0 none, 1 NMS maximum, 3 maximum inside the 10-px border
    int32_t *row_cnt = nullptr; // [2][H2]   per interpolated row: all maxima, kept maxima (hybrid: even / odd column candidates)
    int32_t *row_off = nullptr; // [3][H2+1] exclusive prefix of row_cnt (strict path; the hybrid compaction sums the counts itself)
    int32_t *counts = nullptr;  // [8] n_total, n_kept (device-side sizes of everything downstream), n_candidates (0 when they
                                // did not fit), neighbour points, screened candidates before the capacity check
    int32_t *src = nullptr;     // [cap][2] (pixel offset, kept rank or -1) per maximum, raster order
    ebvo_edge *edges = nullptr; // [cap] kept edges
    double *all4 = nullptr;     // [cap][4] every maximum (x, y, theta, mag)
    void *cand_rec = nullptr;       // hybrid TOED: [cap] exact records of the screened candidates
    int32_t *cand_flag = nullptr;   // [2][cap]   is-maximum / is-kept flags per candidate
    int32_t *cand_off = nullptr;    // [2][cap+1] ints; hybrid: (maxima, kept maxima) per chunk of 256 candidates (toed_exact_decide_kernel)
    double *cand_data = nullptr;    // [cap] 64-byte records: exact gx, gy, |g|, TOx, TOy and the packed NMS sector of a candidate
    int32_t *cand_lists = nullptr;  // [12][cap] candidate indices by phase (4) and by (phase, axis) (8)
    int32_t *cand_lcount = nullptr; // [12]
    int n_total = 0, n_kept = 0; // host copies, valid after a synchronising call
};

// What the last kernel of a device-resident pair reports (match_kernels.hip: pair_result_kernel).
struct PairResult
{
    int32_t n_left, n_right, n_total_left, n_total_right;
    int64_t n_pairs;
    int64_t n_matches;
    int32_t overflow; // bit 0: more candidate pairs than the buffers hold; bit 1: the hybrid TOED screen overflowed
    int32_t pad;
};

struct ProfEvent
{
    hipEvent_t a, b;
    int kid;
};

constexpr int EBVO_CLEAR_MAX = 8;      // arrays one clear launch zeroes
constexpr int EBVO_TOTAL_PARTS = 4096;  // most blocks the counting pass of the candidate search is launched with
constexpr int EBVO_MATCH_PARTS = 4096; // most blocks ncc_tile_kernel is launched with

// Everything that belongs to one HIP stream: a stereo pair in flight (or the workspace of a host-buffer call).
// What a captured pair chain (ebvo_stereo_submit) depends on besides the slot's buffers: every value a launch carries as an
// argument.  `gen` is the context's settings generation (a mode or debug switch changed since the capture) plus the slot's
// own allocation generation (one of ITS buffers re-allocated): either makes the graph stale.  Another slot's first
// allocations do not.
struct PairGraphKey
{
    int32_t h, w, toed_mode, undist;
    int64_t cap_pairs;
    uint64_t gen;
    double epi_thr, max_disp, orient_thr_deg, ncc_thr;
    int32_t stage_mask, flags;
};

struct Slot
{
    hipStream_t stream = nullptr;     // the stream the slot's work is enqueued on: its own, or a lane (ebvo_stereo_submit)
    // the pair chain as a hipGraph: captured at the THIRD submission with the same key (the first one runs directly, performs
    // every allocation and bumps buf_gen, i.e. changes the key; the second one runs directly with the final key), launched from then on -- ~7 us of host time instead of ~90 for the 31 launches
    hipGraphExec_t pair_graph = nullptr;
    PairGraphKey pair_key{};
    uint64_t buf_gen = 1;         // bumped whenever one of this slot's buffers is (re)allocated: part of the key
    bool pair_key_warm = false;   // a direct submission with pair_key has run
    bool pair_graph_off = false;  // capture failed once on this slot: stay on direct launches
    hipStream_t own_stream = nullptr; // created with the slot
    hipEvent_t ev_done = nullptr;     // recorded behind the last kernel of a submitted pair
    hipEvent_t ev_rebind = nullptr;   // orders a slot's earlier work before its first work on another lane
    // ebvo_stereo_upload_async, pull form: the pair's chain starts with a kernel that READS the two images from the caller's
    // page-locked memory (the pointers travel in this page-locked mailbox, which the kernel reads over PCIe -- 32 bytes -- so
    // the captured graph of the chain never changes and a submission issues no copy for it)
    struct PullMail
    {
        const uint8_t *src[2];
        long long stride[2];
    };
    PullMail *h_mail = nullptr, *d_mail = nullptr; // page-locked mailbox: host address / the address the pull kernel reads it at
    bool pull = false;                             // the resident pair is pulled by the chain itself
    hipEvent_t ev_upload = nullptr;   // end of the slot's asynchronous image upload (ebvo_stereo_upload_async, the context's upload stream)
    bool upload_pending = false;      // ... recorded and not yet waited for by a submission or a host call
    uint8_t *h_up = nullptr;          // page-locked staging of the stream form: a pair of images from PAGEABLE caller memory is copied
    size_t h_up_bytes = 0;            // here before the call returns (the caller may free its images; the runtime reads pageable sources late)
    ImageWS im[2];
    int cur_h = 0, cur_w = 0;
    bool have_pair = false, have_run = false, in_flight = false, have_refined = false;
    bool have_sims = true; // the last pair stored all four scores (not submitted with EBVO_PAIR_NO_SIMS)
    bool toed_strict_override = false; // the hybrid screen overflowed on the resident pair (ebvo_stereo_wait re-ran it strict): later
                                       // submissions of the SAME images go strict at once; cleared by the next upload
    bool undist_pair = false; // the resident pair was uploaded raw and is undistorted by the pipeline (im[k].raw -> im[k].img)

    // matching workspace
    GrowBuf grad_x, grad_y, gn_xy, gn_out, gn_valid, gn_iters, gn_state, gn_lists, gn_pack; // photometric refinement (refine_kernels.hip)
    GrowBuf tq_i32, tq_cols, tq_f64, tq_u8, tq_cells; // temporal quads of the slot's pair against the keyframe
    GrowBuf tq_chain;                                  // ... and what the stages after the NCC filter need
    bool sift_left_valid = false;                      // sift_desc holds the descriptors of every left TOED edge of this pair
    struct TqFinal                                     // the quads that leave the chain (pointers into tq_chain)
    {
        int32_t *rp = nullptr, *cf = nullptr;
        ebvo_edge *L = nullptr, *R = nullptr;
        double *ncc = nullptr, *sift = nullptr, *sL = nullptr, *sR = nullptr;
        uint8_t *valid = nullptr;
        int64_t n = -1;
    } tq_final;
    int tq_n_kf = 0;
    int64_t tq_n = -1;                       // -1: none
    // ebvo_temporal_match_submit / _wait: the candidate and NCC stages are enqueued for a capacity of quads (what the last
    // frame needed, with headroom); the two counts travel behind the last kernel
    int64_t tq_cap = 0;
    unsigned long long *d_tq_tot = nullptr, *h_tq_tot = nullptr; // [2] device / page-locked: candidate quads, kept quads
    hipEvent_t ev_tq = nullptr;
    bool tq_in_flight = false, tq_empty = false;
    ebvo_temporal_params tq_params{};
    GrowBuf sift_used; // [2] list lengths, then the lists of the left / right edges that appear in a candidate pair, then nR flag bytes
    GrowBuf sift_img, sift_desc, sift_f32, sift_dist; // SIFT: blurred levels, descriptor banks, per-pair distances (sift_kernels.hip)
    GrowBuf fin_i32, fin_edges, fin_f64, fin_u8, fin_out; // ebvo_stereo_finalize: CSRs, candidate lists, scores, final rows
    int n_final = 0;
    bool have_final = false, final_has_rows = false;
    // ebvo_stereo_finalize_submit / _wait: the chain is enqueued without reading a count back; the stage totals travel in
    // one copy behind its last kernel
    int32_t *d_fin_tot = nullptr, *h_fin_tot = nullptr; // [8] device / page-locked: n_sift, n_ncc, n_bnb, n_clusters, n_ncc2, n_final
    hipEvent_t ev_fin = nullptr;
    bool fin_in_flight = false;
    GrowBuf lines, boxes_chunk, boxes_group, cand_cnt, cand_stage, cand_tileflag, row_ptr, scan_tmp, col_idx, rc_edges, sims, best, keep,
        patches_raw, patches_norm, patches_flag, patches_norm_r, patches_flag_r, pair_left, sincos, scratch_b, scratch_c,
        scratch_d;
    int64_t cap_pairs = 0;               // capacity of col_idx & co. as the kernels of the current call see it
    int64_t pipe_cap = 0;                // capacity the device pipeline keeps between pairs
    unsigned long long *d_total = nullptr; // [0]: 64-bit candidate total; [1 ..]: per-block partial totals
    int n_total_part = 0;
    int32_t *d_matches = nullptr; // [EBVO_MATCH_PARTS] per-block kept-pair counts of ncc_tile_kernel
    int n_match_part = 0;
    int32_t *d_sizes = nullptr;          // [4] host-provided sizes for the host-buffer entry points
    double *d_F = nullptr;               // 9 doubles
    PairResult *d_result = nullptr, *h_result = nullptr; // h_result is pinned
    PairResult *d_result_host = nullptr;                 // the device's address of h_result: pair_result_kernel writes the record
                                                         // straight into host memory (round 4: no copy node behind it)
    double F_dev[9] = {0};                               // what d_F holds (uploaded again only when the caller's matrix changes)
    bool F_dev_valid = false;
    void *h_arena = nullptr;             // pinned staging of ebvo_stereo_fetch_begin / _end
    size_t h_arena_bytes = 0;
    int fetch_what = 0;                  // arrays in flight to / present in h_arena
    bool fetch_pending = false;          // copies enqueued, not yet waited for
    size_t fetch_off[8] = {0};           // left, right, row_ptr, col_idx, sims, best, keep (compact: xyL, xyR, thL, thR, row_ptr, col_idx, best, keep bits)
    bool fetch_compact = false;          // the arena holds the arrays of ebvo_stereo_fetch_compact_begin
    // EBVO_PAIR_PUSH: the pair's chain ends with a kernel that WRITES the compact results into this page-locked arena (sized for
    // the capacities the chain was enqueued with); no copy, no event: they are there when the pair's ebvo_stereo_wait returns
    void *h_push = nullptr, *d_push = nullptr;
    size_t push_bytes = 0;
    bool have_push = false;              // the last completed pair pushed its results (ebvo_stereo_pushed_view)
    bool fetch_packed = false, fetch_packed_theta = false; // the arena holds one packed block (layout from fetch_result's counts)
    PairResult fetch_result{};
    bool have_pack = false;              // ... packed them into fetch_pack (EBVO_PAIR_PACK): the compact fetch is ONE copy
    GrowBuf fetch_pack;                  // device staging of the compact fetch: (x, y) pairs, orientations, keep bits
    ebvo_stereo_params params{};
    PairResult result{};                 // last completed result

    std::vector<ProfEvent> prof_pending;
    bool prof_now = true; // this submission is one of the sampled ones
};

// the image the NCC passes sample: the RAW one (src/Stereo_Matches.cpp:562-563)
inline const uint8_t *ncc_img(const Slot &s, int k) { return s.undist_pair ? s.im[k].raw : s.im[k].img; }

// page-locked host memory owned by the context (results of the resident stage-wise calls), grown on demand
struct PinnedBuf
{
    void *p = nullptr;
    size_t bytes = 0;
};

struct ebvo_ctx
{
    int device = 0;
    int max_h = 0, max_w = 0;
    int cap_edges = 0; // per image
    int toed_mode = EBVO_TOED_STRICT;
    std::string last_error;
    std::vector<Slot *> slots; // slot 0 always exists; it also serves the host-buffer entry points
    // keyframe of the temporal stage (ebvo_temporal_set_keyframe): final stereo mates and their stored patches, device-resident
    int kf_n = -1;             // -1: no keyframe
    ebvo_edge *kf_L = nullptr, *kf_R = nullptr;
    float *kf_Ln = nullptr, *kf_Rn = nullptr;   // [n][2][49] normalised patches (left image raw, right image undistorted)
    uint8_t *kf_Lf = nullptr, *kf_Rf = nullptr; // [n][2] sentinel flags
    uint8_t *kf_Ld = nullptr, *kf_Rd = nullptr; // [n][2][128] SIFT descriptors of the mates (left TOED edge / final right edge)
    uint8_t *kf_imgL = nullptr, *kf_imgR = nullptr; // the keyframe's undistorted images (photometric refinement of the quads)
    size_t kf_img_bytes = 0;
    size_t kf_cap = 0;
    bool undist_on = false;    // ebvo_stereo_set_undistort
    ebvo_undistort_params undist{};
    int lanes = 4;             // streams the kernels of submitted pairs are dealt to from four slots on (ebvo_stereo_submit)
    int prof_only = -1; // >= 0: the profiler instruments this stage id alone (ebvo_debug_set key 3 = id + 1)
    uint64_t submit_seq = 0;
    std::vector<hipStream_t> lane_streams; // created on first use, owned by the context
    hipStream_t copy_stream = nullptr;     // result copies of ebvo_stereo_fetch_begin (all slots), created on first use
    // memory page-locked through ebvo_host_register: host range and the address the device uses for it.  ebvo_stereo_upload_async
    // takes its pull form only for images inside one of these ranges (a lookup in this list: no runtime call per frame --
    // hipPointerGetAttributes costs 5 us in a bare process and 300 us in one where PyTorch has initialised the device)
    struct HostRange
    {
        const uint8_t *host = nullptr, *dev = nullptr;
        size_t bytes = 0;
        bool ours = false; // registered by the library (to be unregistered by it)
    };
    std::vector<HostRange> host_ranges;
    int64_t pull_uploads = 0, stream_uploads = 0; // ebvo_stereo_upload_async calls by form (ebvo_ingest_stats)
    hipStream_t upload_stream = nullptr;   // image uploads of ebvo_stereo_upload_async (all slots), created on first use
    // resident stage-wise path (ebvo_toed_resident / ebvo_epi_candidates_resident / ebvo_ncc_pairs_resident): the TOED results
    // of the last two images stay in slot 0's image workspaces; tag 0 = that workspace holds nothing a caller may refer to
    uint64_t sw_seq = 0;
    uint64_t sw_tag[2] = {0, 0};
    int sw_h = 0, sw_w = 0;
    PinnedBuf sw_toed[2], sw_cand, sw_ncc, sw_up;
    int gn_no_rows = 0;        // developer key (ebvo_debug_set 4): 1 = the refinements never use the eight-lanes-per-pair layout
    int gn_persist_waves = 2;      // developer key (ebvo_debug_set 8): waves per SIMD the persistent refinement kernel is built for (2 | 3)
    int gn_persist_blocks = 0;     // developer key (ebvo_debug_set 9): most workgroups of the persistent refinement launch, 0 = default
    int gn_per_iteration_rows = 0; // developer key (ebvo_debug_set 7): 1 = the eight-lanes layout as a launch per iteration
    int gn_rows_below = 0;     // developer key (ebvo_debug_set 5): active-pair count below which an iteration uses it, 0 = default
    int wait_attempts = 0;     // test hook (ebvo_debug_set): attempts of ebvo_stereo_wait's regrow loop, 0 = default (4)
    int force_overflow = 0;    // test hook: treat the next N results as overflowed
    uint64_t graph_gen = 1;     // bumped by every mode / debug change: invalidates the captured graphs of every slot
    int use_graphs = 1;         // EBVO_GRAPHS=0 or ebvo_debug_set(10, 0): direct launches only
    int64_t graph_launches = 0; // pairs submitted as a graph launch
    int exact_blocks[2] = {0, 0}; // developer keys (ebvo_debug_set 11, 12): grid of the exact centre / mags kernel in blocks (0 = what the device keeps resident)
    int cand_blocks = 0;        // developer key (ebvo_debug_set 19): most blocks of candidates<count> (0 = 1024; at most EBVO_TOTAL_PARTS)
    int small_div = 0;          // developer key (ebvo_debug_set 18): the grids of the latency-bound decide / cand_scatter / candidates<fill> are
                                // 512 / 512 / 4096 blocks divided by this (0 = 4: +1.0 % pairs/s against 1, +0.7 % for 2, two runs; 8: -0.3 %)
    int ncc_blocks = 0;         // developer key (ebvo_debug_set 17): grid of ncc_tile_kernel in blocks (0 = what the device keeps resident)
    int stop_stage = 0;         // developer key (ebvo_debug_set 16): the resident pair's chain ends after stage N (tools/gpu_prefix_chain.py:
                                // the pair rate of every prefix of the chain = what each stage costs in the steady state); 0 = whole chain
    int repeat_mask = 0;        // developer key (ebvo_debug_set 15): bit 0 centre, 1 mags, 2 right bank, 3 NCC tile launched TWICE (idempotent kernels: what
                                // one more launch of each costs the pair rate, tools/gpu_marginal_cost.py)
    int no_prep = 0;            // developer key (ebvo_debug_set 14): 1 = lines, boxes, sincos and row pairs as four launches (A/B)
    int ingest_stream = 0;      // developer key (ebvo_debug_set 13): 1 = ebvo_stereo_upload_async copies on the upload stream (A/B)
    bool screen_audit = false;  // ebvo_toed_screen_audit is running: the screen keeps its gx, gy, |g| (toed_kernels.hip)
    int64_t toed_fallbacks = 0; // hybrid TOED runs repeated on the strict path (more screened candidates than cap_edges)

    // profiling (accumulated over all slots)
    bool prof = false;
    int prof_every = 1;          // device pipeline: bracket the kernels of every N-th submitted pair only
    int64_t prof_submits = 0;
    std::vector<ProfEvent> prof_free;
    double prof_ms[K_NUM] = {0};
    int64_t prof_launches[K_NUM] = {0};
};

int ebvo_fail_hip(ebvo_ctx *ctx, hipError_t e, const char *what, const char *file, int line);

// A pair / item count that is either known on the host (dev == nullptr) or lives in device memory (an int32: the last
// entry of a CSR row_ptr, a stage total).  With a device pointer `host` is the UPPER BOUND the launch and the buffers are
// sized for: the chains after the first NCC pass enqueue every stage without reading a count back.
struct DevCount
{
    int64_t host;
    const int32_t *dev;
};
#ifdef __HIPCC__
__device__ inline int64_t devcount(const DevCount &c)
{
    if (!c.dev)
        return c.host;
    const int64_t v = *c.dev;
    return v < c.host ? v : c.host;
}
#endif

#define EBVO_HIP(ctx, call)                                              \
    do                                                                   \
    {                                                                    \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess)                                            \
            return ebvo_fail_hip((ctx), e_, #call, __FILE__, __LINE__);  \
    } while (0)

// grow a slot buffer (synchronises the slot's stream before freeing the old allocation)
int ebvo_grow(ebvo_ctx *ctx, Slot &s, GrowBuf &b, size_t bytes);

// profiling brackets around kernel launches on a slot's stream
bool ebvo_prof_begin(ebvo_ctx *ctx, Slot &s, int kid); // false: nothing was recorded (profiler off, or another stage selected)
void ebvo_prof_end(ebvo_ctx *ctx, Slot &s);

bool ebvo_prof_kernel(ebvo_ctx *ctx, Slot &s, int kid, hipEvent_t *a, hipEvent_t *b);

struct ProfScope
{
    ebvo_ctx *c;
    Slot &s;
    bool on;
    ProfScope(ebvo_ctx *ctx, Slot &slot, int kid) : c(ctx), s(slot), on(ebvo_prof_begin(ctx, slot, kid)) {}
    ~ProfScope()
    {
        if (on)
            ebvo_prof_end(c, s);
    }
};

// ---- device-level stages (device pointers; asynchronous on the slot's stream; NO host synchronisation) ----
// A size argument is a host value plus an optional device pointer; when the pointer is non-null the kernels read
// the size from device memory and `*_cap` bounds the launch.

// toed_kernels.hip
int toed_init_constants(ebvo_ctx *ctx);
// conv + NMS + compaction of n_img (1 or 2) resident images of slot s; counts stay in s.im[k].counts.  mode < 0: the
// context's mode.  A hybrid run whose screen flags more candidates than the buffers hold leaves counts[4] > cap_edges and
// empty results: the caller re-runs with EBVO_TOED_STRICT.
int toed_enqueue(ebvo_ctx *ctx, Slot &s, int n_img, int h, int w, hipEvent_t ev_conv_begin, hipEvent_t ev_conv_end,
                 hipEvent_t ev_end, int mode = -1, bool want_all4 = true);
// ebvo_toed_screen_audit: where the audit kernel leaves the 8 words of image k; the screen's budget {E_G, E_M, E_S, TOL_M, TOL_S}
unsigned long long *toed_screen_audit_result(Slot &s, int k, int h, int w);
void toed_screen_budget(double out[5]);

// match_kernels.hip
int match_lines_enqueue(ebvo_ctx *ctx, Slot &s, const double *d_F, const ebvo_edge *d_edges, int n,
                        const int32_t *d_n, int cap_n, double *d_lines);
// candidate search into s.row_ptr / s.col_idx (capacity s.cap_pairs must be set and the buffers allocated);
// s.d_total receives the 64-bit number of pairs found
int match_candidates_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const int32_t *d_nL,
                             const ebvo_edge *d_R, int nR, const int32_t *d_nR, int cap_edges, const double *d_lines,
                             double epi_thr, double max_disp, double orient_thr_deg, int stage_mask, bool fill, bool prep_done = false);
int match_candidates_fill_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const int32_t *d_nL,
                                  const ebvo_edge *d_R, int nR, const int32_t *d_nR, int cap_edges,
                                  const double *d_lines, double epi_thr, double max_disp, double orient_thr_deg,
                                  int stage_mask);
int match_patches_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch,
                          const ebvo_edge *d_edges, int n, const int32_t *d_n, int cap_n, float *d_raw, float *d_norm,
                          uint8_t *d_flag);
int match_ncc_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgR, int h, int w, int pitchR,
                            const ebvo_edge *d_Rc, const int32_t *d_row_ptr, int nL, int64_t n_pairs,
                            const float *d_left_norm, const uint8_t *d_left_flag, double thr, double *d_sims,
                            double *d_best, uint8_t *d_keep, int32_t *d_pair_left_scratch = nullptr,
                            void *d_sincos_scratch = nullptr /* n_pairs double2; NULL: the slot's own buffers */,
                            const int32_t *d_n_pairs = nullptr /* the count on the device; n_pairs is then its bound */);
// resident pipeline: sin/cos, right patch bank, LDS-tiled NCC of every CSR pair (sizes read on the device)
// left = index of the slot's image workspace that holds the LEFT image and edges (the right one is the other)
int match_ncc_resident_enqueue(ebvo_ctx *ctx, Slot &s, int h, int w, int cap_edges, double thr, int left = 0, bool want_sims = true,
                               bool prep_done = false);
// epipolar lines, right-edge boxes (+ tile flags), sin / cos of both edge lists and both row-pair images of a resident pair in
// ONE launch; pass prep_done = true to the two calls above / below afterwards
int match_prep_enqueue(ebvo_ctx *ctx, Slot &s, int h, int w, int cap_edges);
size_t match_right_bank_bytes(int cap_edges);
int match_pair_result_enqueue(ebvo_ctx *ctx, Slot &s, int cand_cap); // cand_cap > 0: hybrid TOED, report candidates > cand_cap
int match_orient_flags_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const ebvo_edge *d_R, const int32_t *d_row_ptr,
                               const int32_t *d_col_idx, int64_t n_pairs, double orient_thr_deg, uint8_t *d_ok);
// temporal quads (Temporal_Matches): cells + chunk boxes of the current-frame mates, candidate count / fill, indexed NCC
int match_temporal_cells_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cfL, const ebvo_edge *d_cfR, int n_cf, int cell, int gw,
                                 int gh, void *d_grid);
size_t match_temporal_grid_bytes(int n_cf, int n_cells);
int match_temporal_candidates_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfL, const ebvo_edge *d_kfR, int n_kf,
                                      const ebvo_edge *d_cfL, const ebvo_edge *d_cfR, const void *d_grid, int n_cf, int cell,
                                      int sr, int gw, int gh, double orient_thr, int32_t *d_cnt, const int32_t *d_row_ptr,
                                      int32_t *d_col_idx, int64_t cap);
int match_count_flags_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_flags, int64_t n, unsigned long long *d_out,
                              const int32_t *d_n = nullptr);
int match_ncc_quads_indexed_enqueue(ebvo_ctx *ctx, Slot &s, const float *kfLn, const uint8_t *kfLf, const float *kfRn,
                                    const uint8_t *kfRf, const float *cfLn, const uint8_t *cfLf, const float *cfRn,
                                    const uint8_t *cfRf, const int32_t *d_quad_kf, const int32_t *d_quad_cf, int64_t n_quads,
                                    double thr, double *d_sim_left, double *d_sim_right, uint8_t *d_keep,
                                    const int32_t *d_n_quads = nullptr);
// exclusive scan of n (+ n_add) int32 on the slot's stream; n_dev != nullptr: the count lives on the device, cap_n bounds it
// zero n (<= EBVO_CLEAR_MAX) int32 arrays with one launch
int ebvo_clear_enqueue(ebvo_ctx *ctx, Slot &s, int32_t *const ptrs[], const int counts[], int n);
// n_add = 1: the scan covers one more element that counts as zero whatever the memory behind `in` holds (out[n] = total);
// d_total (optional, n_add = 1): the total is stored there as well
int ebvo_device_scan(ebvo_ctx *ctx, Slot &s, const int32_t *in, int32_t *out, int n_host, const int32_t *n_dev, int n_add,
                     int cap_n, int32_t *d_total = nullptr);
// glue_kernels.hip
int glue_bnb_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const double *d_scores, double thr,
                     int higher_is_better, int32_t *d_new_count, int32_t *d_order);
// temporal quads after the NCC filter
struct GlueGather // null-terminated lists of (source, destination) arrays gathered through one index list
{
    const int32_t *i_src[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *i_dst[4] = {nullptr, nullptr, nullptr, nullptr};
    const double *d_src[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_dst[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const uint8_t *b_src[2] = {nullptr, nullptr};
    uint8_t *b_dst[2] = {nullptr, nullptr};
    const ebvo_edge *e_src[2] = {nullptr, nullptr};
    ebvo_edge *e_dst[2] = {nullptr, nullptr};
};
int glue_row_index_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                           const int32_t *d_rp_out, int nL, int32_t *d_idx);
int glue_gather_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_idx, int64_t n, const GlueGather &g);
int glue_quad_refine_inputs_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfE, const int32_t *d_quad_kf,
                                    const ebvo_edge *d_cfE, const int32_t *d_quad_cf, int64_t n, ebvo_edge *d_kf_out,
                                    ebvo_edge *d_cf_out, double *d_init);
int glue_quad_apply_refine_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfL, const ebvo_edge *d_cfL, const double *d_dispL,
                                   const uint8_t *d_validL, const ebvo_edge *d_kfR, const ebvo_edge *d_cfR,
                                   const double *d_dispR, const uint8_t *d_validR, int64_t n, ebvo_edge *d_cenL,
                                   ebvo_edge *d_cenR, uint8_t *d_valid);
int glue_quad_cluster_post_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, int nL, const int32_t *d_new_count,
                                   const int32_t *d_cluster_of, const ebvo_edge *d_centres, const ebvo_edge *d_cenL,
                                   const ebvo_edge *d_cenR, const int32_t *d_rp_out, ebvo_edge *d_outL, ebvo_edge *d_outR,
                                   int32_t *d_src);
int glue_keep_best_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const double *d_scores,
                           int32_t *d_new_count, int32_t *d_order);
// n / d_n: item count on the host, or (d_n != nullptr) its upper bound and where the device holds the count
int glue_shift_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cand, const double *d_lines, const int32_t *d_pair_left,
                       int64_t n, ebvo_edge *d_out, const int32_t *d_n = nullptr);
int glue_cluster_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cand, const int32_t *d_row_ptr, int nL,
                         int by_orientation, int skip_single, int32_t *d_new_count, ebvo_edge *d_centres,
                         int32_t *d_cluster_of);
int glue_rows_from_flags_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const uint8_t *d_flags,
                                 int32_t *d_new_count, int32_t *d_order);
int glue_gather_rows_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                             const int32_t *d_rp_out, int nL, const ebvo_edge *d_E_src, const int32_t *d_emap,
                             ebvo_edge *d_E_dst, const double *d_D_src, double *d_D_dst);
int glue_xy_enqueue(ebvo_ctx *ctx, Slot &s, ebvo_edge *d_edges, double *d_xy, int64_t n, bool to_edges, const int32_t *d_n = nullptr);
int glue_final_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                             const int32_t *d_rp_out, int nL, const ebvo_edge *d_L, const ebvo_edge *d_cand,
                             const double *d_score, int32_t *d_left_index, ebvo_edge *d_left_edge, ebvo_edge *d_right_edge,
                             double *d_final_score);
// refine_kernels.hip
int refine_sobel_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch, float *d_gx, float *d_gy,
                         void *d_gxy /* optional interleaved float2 plane */);
int refine_undistort_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_src, int pitch, int h, int w, const double K[4],
                             const double *dist, int n_dist, double *d_xs, uint8_t *d_dst, int dpitch);
int refine_gn_stereo_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgL, const uint8_t *d_imgR, const void *d_gxy,
                             int h, int w, const ebvo_edge *d_L, int nL, const double *d_lines,
                             const int32_t *d_pair_left, const double *d_cand_xy /* or NULL with R + col_idx */,
                             const ebvo_edge *d_R, const int32_t *d_col_idx, const uint8_t *d_keep /* optional */,
                             int64_t n_pairs, int max_iter, double tol, double huber, double *d_alpha, double *d_score,
                             double *d_conf, uint8_t *d_valid, int32_t *d_iters, double *d_refined_xy,
                             const int32_t *d_n_pairs = nullptr /* the count on the device; n_pairs is then its bound */);
int refine_finalize_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const double *K_left, const double *K_right, const double *R21,
                                  const double *T21, const ebvo_edge *d_L, const ebvo_edge *d_R, int n, double *d_out,
                                  const int32_t *d_n = nullptr);
int refine_gn_temporal_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgK, const uint8_t *d_imgC, const void *d_gxy,
                               int h, int w, const ebvo_edge *d_kf, const ebvo_edge *d_cf, const double *d_init, int64_t n,
                               int max_iter, double tol, double huber, double *d_disp, double *d_score, uint8_t *d_valid,
                               int32_t *d_iters, int64_t n_first = -1, const uint8_t *d_imgK2 = nullptr,
                               const uint8_t *d_imgC2 = nullptr, const void *d_gxy2 = nullptr);
// sift_kernels.hip
int sift_base_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch, float *d_tmp, float *d_base);
int sift_descriptors_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base, int h, int w, const ebvo_edge *d_edges, int n,
                             float *d_desc_f, uint8_t *d_desc_u8);
// the descriptors of the edges list[0 .. *d_n) only (n_max bounds the launch), written at the edges' own places in d_desc_u8
int sift_descriptors_listed_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base, int h, int w, const ebvo_edge *d_edges,
                                    const int32_t *d_list, const int32_t *d_n, int n_max, uint8_t *d_desc_u8);
int sift_descriptors_listed_pair_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base0, const float *d_base1, int h, int w,
                                         const ebvo_edge *d_edges0, const ebvo_edge *d_edges1, const int32_t *d_list0,
                                         const int32_t *d_list1, const int32_t *d_n, int n_max0, int n_max1, uint8_t *d_desc0,
                                         uint8_t *d_desc1);
// which left (a row of the CSR with at least one pair) and right (named by some pair) edges need a descriptor: two index lists,
// unordered, and their lengths in d_counts[0 .. 1]; d_flags: nR scratch bytes
int sift_used_edges_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const int32_t *d_col_idx, int64_t n_pairs,
                            int nR, int32_t *d_counts, int32_t *d_list_left, int32_t *d_list_right, uint8_t *d_flags);
int sift_gather_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_src, const int32_t *d_index, int n, uint8_t *d_dst);
int sift_to_u8_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_f, int64_t n, uint8_t *d_u8);
int sift_distances_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_left, const uint8_t *d_cand, const int32_t *d_pair_left,
                           const int32_t *d_cand_index, int64_t n_pairs, double thr, double *d_dist, uint8_t *d_ok);
int sift_and_flags_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_a, const uint8_t *d_b, int64_t n, uint8_t *d_out);
int match_expand_rows_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, int64_t n_pairs, int32_t *d_pair_left);
int match_ncc_stored_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_A, const float *d_B, int n, double *d_sim);
int misc_fp64_peak(ebvo_ctx *ctx, Slot &s, int iters, double *tf_muladd, double *tf_fma);

#endif

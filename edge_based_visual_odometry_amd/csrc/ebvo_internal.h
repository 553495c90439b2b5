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
    K_BOXES,
    K_LINES,
    K_CAND_COUNT,
    K_SCAN,
    K_CAND_FILL,
    K_PATCHES,
    K_NCC_PAIRS,
    K_NCC_STORED,
    K_MISC,
    K_NUM
};
static_assert(K_NUM <= EBVO_MAX_KERNELS, "grow EBVO_MAX_KERNELS");

extern const char *const g_kernel_names[K_NUM];

// Interpolated-grid planes written by the convolution, per image.
enum
{
    PL_IX = 0,
    PL_IY,
    PL_MAG,
    PL_TOX,
    PL_TOY,
    PL_NUM
};

struct GrowBuf
{
    void *p = nullptr;
    size_t bytes = 0;
};

// Per-image device workspace (two of them: left / right).
struct ImageWS
{
    uint8_t *img = nullptr;     // h*w, tightly packed
    double *maps = nullptr;     // PL_NUM planes of 2H x 2W
    uint8_t *flag = nullptr;    // 2H x 2W: 0 none, 1 NMS maximum, 3 maximum inside the 10-px border
    int32_t *row_cnt = nullptr; // [2][H2]   per interpolated row: all maxima, kept maxima
    int32_t *row_off = nullptr; // [2][H2+1] exclusive prefix of row_cnt
    int32_t *counts = nullptr;  // [2] n_total, n_kept
    int32_t *src = nullptr;     // [cap][2] (pixel offset, kept rank or -1) per maximum, raster order
    ebvo_edge *edges = nullptr; // [cap] kept edges
    double *all4 = nullptr;     // [cap][4] every maximum (x, y, theta, mag)
    int n_total = 0, n_kept = 0;
};

struct ProfEvent
{
    hipEvent_t a, b;
    int kid;
};

struct ebvo_ctx
{
    int device = 0;
    int max_h = 0, max_w = 0;
    int cap_edges = 0; // per image
    hipStream_t stream = nullptr;
    std::string last_error;

    ImageWS im[2];
    int cur_h = 0, cur_w = 0; // size of the resident stereo pair
    bool have_pair = false, have_run = false;

    // matching workspace (grown on demand)
    GrowBuf lines, boxes_chunk, boxes_group, cand_cnt, row_ptr, scan_tmp, col_idx, rc_edges, sims, best, keep,
        patches_raw, patches_norm, patches_flag, patches_norm_r, patches_flag_r, match_cnt, scratch_a, scratch_b, scratch_c, scratch_d;
    int64_t n_pairs = 0, n_matches = 0;
    int n_left = 0;
    double *d_params = nullptr; // F21 for the device line kernel (9 doubles)

    // pinned host staging for small read-backs
    int32_t *h_small = nullptr; // 64 ints

    // profiling
    bool prof = false;
    std::vector<ProfEvent> prof_pending;
    std::vector<ProfEvent> prof_free;
    double prof_ms[K_NUM] = {0};
    int64_t prof_launches[K_NUM] = {0};
};

int ebvo_fail_hip(ebvo_ctx *ctx, hipError_t e, const char *what, const char *file, int line);

#define EBVO_HIP(ctx, call)                                              \
    do                                                                   \
    {                                                                    \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess)                                            \
            return ebvo_fail_hip((ctx), e_, #call, __FILE__, __LINE__);  \
    } while (0)

int ebvo_grow(ebvo_ctx *ctx, GrowBuf &b, size_t bytes);

// profiling brackets around a kernel launch
void ebvo_prof_begin(ebvo_ctx *ctx, int kid);
void ebvo_prof_end(ebvo_ctx *ctx);

struct ProfScope
{
    ebvo_ctx *c;
    ProfScope(ebvo_ctx *ctx, int kid) : c(ctx) { ebvo_prof_begin(c, kid); }
    ~ProfScope() { ebvo_prof_end(c); }
};

// ---- device-level stages (all pointers are device pointers; asynchronous on ctx->stream) ----

// toed_kernels.hip
int toed_init_constants(ebvo_ctx *ctx);
// runs conv + NMS + compaction for n_img (1 or 2) resident images ctx->im[0..n_img-1];
// fills im[k].n_total / n_kept (synchronises once to read the counts).
int toed_run_device(ebvo_ctx *ctx, int n_img, int h, int w, float *ms_conv, float *ms_nms);

// match_kernels.hip
int match_lines_device(ebvo_ctx *ctx, const double *d_F, const ebvo_edge *d_edges, int n, double *d_lines);
// candidate search: fills ctx->row_ptr / ctx->col_idx (device), returns n_pairs (synchronises)
int match_candidates_device(ebvo_ctx *ctx, const ebvo_edge *d_L, int nL, const ebvo_edge *d_R, int nR,
                            const double *d_lines, double epi_thr, double max_disp, double orient_thr_deg,
                            int stage_mask, int64_t *n_pairs);
int match_gather_edges_device(ebvo_ctx *ctx, const ebvo_edge *d_R, const int32_t *d_col_idx, int64_t n,
                              ebvo_edge *d_out);
int match_patches_device(ebvo_ctx *ctx, const uint8_t *d_img, int h, int w, int pitch, const ebvo_edge *d_edges,
                         int n, float *d_raw, float *d_norm, uint8_t *d_flag);
int match_ncc_pairs_device(ebvo_ctx *ctx, const uint8_t *d_imgR, int h, int w, int pitchR, const ebvo_edge *d_Rc,
                           const int32_t *d_row_ptr, int nL, int64_t n_pairs, const float *d_left_norm,
                           const uint8_t *d_left_flag, double thr, double *d_sims, double *d_best,
                           uint8_t *d_keep, int32_t *d_match_cnt);
int match_ncc_banked_device(ebvo_ctx *ctx, const int32_t *d_row_ptr, const int32_t *d_col_idx, int nL, int64_t n_pairs,
                            const float *d_left_norm, const uint8_t *d_left_flag, const float *d_right_norm,
                            const uint8_t *d_right_flag, double thr, double *d_sims, double *d_best, uint8_t *d_keep,
                            int32_t *d_match_cnt);
int match_ncc_stored_device(ebvo_ctx *ctx, const float *d_A, const float *d_B, int n, double *d_sim);
int misc_fp64_peak(ebvo_ctx *ctx, int iters, double *tf_muladd, double *tf_fma);

#endif

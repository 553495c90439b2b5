/*
 * ebvo_hip.h -- C ABI of the MI355X-native edge-extraction-and-matching hot path.
 *
 * Drop-in boundary for Brown-LEMS/Edge_Based_Visual_Odometry (C++17, no FFI of its own): the
 * reference reaches this path through three C++ objects held by Pipeline
 * (include/Pipeline.h:193-195).  Each entry point below names the reference interface it
 * replaces (file:line); INTEGRATION.md shows the adapter classes a maintainer drops in.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer, the library owns the ctx;
 *   - return 0 (EBVO_OK) or a negative ebvo_status; nothing throws, nothing is retained
 *     across calls; numeric sentinels of the reference are reproduced (NaN for out-of-image
 *     or integer-coordinate bilinear samples, -1.0 for a zero-variance NCC patch);
 *   - one ctx per host thread and per GPU; a ctx is not thread-safe; calls are synchronous
 *     for the caller (internally ordered on the ctx's HIP stream);
 *   - there is NO CPU fallback: without a usable HIP device ebvo_ctx_create fails.
 *
 * Arithmetic contract: IEEE double, separate multiply and add (no FMA), the reference's
 * operation order; atan2 / sin / cos are the correctly rounding routines of
 * csrc/ebvo_math.h.  Edge positions and indices, candidate lists and match-pair IDs are
 * bit-exact against the CPU path; orientations are the correctly rounded value (glibc differs
 * from it by 1 ulp on ~0.06 % of inputs).
 */
#ifndef EBVO_HIP_H
#define EBVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBVO_ABI_VERSION 6 /* 2: photometric refinement, stage glue, finalisation, resident chain; 3: pinned result views,
                              undistortion, SIFT descriptors, SIFT stages of the chain; 4: the temporal chain after the
                              NCC filter (ebvo_temporal_params / _counts grew, ebvo_temporal_fetch_final); 5: the resident
                              stage-wise calls (ebvo_toed_resident, ebvo_epi_candidates_resident, ebvo_ncc_pairs_resident); 6: ebvo_toed_screen_audit, ebvo_stereo_upload_async, ebvo_host_register, ebvo_stereo_fetch_compact_begin / _end, ebvo_stereo_pushed_view */

typedef struct ebvo_ctx ebvo_ctx;

/* struct Edge (include/toed/cpu_toed.hpp:26-48) without the cv:: type: location, orientation,
 * index.  b_isEmpty / frame_source are filled by the adapter exactly as TOED leaves them. */
typedef struct ebvo_edge
{
    double x, y, theta;
    int32_t index;
    int32_t pad;
} ebvo_edge;

typedef enum ebvo_status
{
    EBVO_OK = 0,
    EBVO_ERR_ARG = -1,      /* null pointer, non-positive size, size above the ctx maximum */
    EBVO_ERR_CAPACITY = -2, /* an output buffer is too small; the required size is reported */
    EBVO_ERR_HIP = -3,      /* a HIP call failed; see ebvo_last_error */
    EBVO_ERR_NOMEM = -4,
    EBVO_ERR_STATE = -5     /* call made before the data it needs exists */
} ebvo_status;

/* stage_mask bits of ebvo_epi_candidates */
enum
{
    EBVO_STAGE_EPIPOLAR = 1,    /* apply_Epipolar_Line_Distance_Filtering, src/Stereo_Matches.cpp:381-419 */
    EBVO_STAGE_DISPARITY = 2,   /* apply_Disparity_Filtering, :534-553 */
    EBVO_STAGE_ORIENTATION = 4, /* apply_orientation_filter, :863-915 */
    EBVO_STAGE_ALL = 7
};

/* Defaults = the reference's compile-time macros (include/definitions.h:17-24). */
#define EBVO_EPIPOLAR_LINE_DIST_THRESH 0.5
#define EBVO_MAX_DISPARITY 25.0
#define EBVO_ORIENT_THRESH_DEG 10.0
#define EBVO_NCC_THRESH 0.6
#define EBVO_NCC_THRESH_TEMPORAL 0.8
#define EBVO_PATCH_SIZE 7
#define EBVO_PATCH_ELEMS 49

const char *ebvo_strerror(int status);
const char *ebvo_last_error(const ebvo_ctx *ctx); /* text of the last HIP failure, "" if none */
int ebvo_abi_version(void);

/* Replaces ThirdOrderEdgeDetectionCPU::ThirdOrderEdgeDetectionCPU(int H, int W) and its
 * destructor (src/toed/cpu_toed.cpp:24-64, :649-663): device workspace for images up to
 * max_h x max_w on HIP device `device`. */
int ebvo_ctx_create(int device, int max_h, int max_w, ebvo_ctx **out);
void ebvo_ctx_destroy(ebvo_ctx *ctx);

/* How the third-order detector reaches its (identical) result:
 *   EBVO_TOED_STRICT  the reference's direct-form convolution at every pixel (27.6 k fp64 operations per pixel);
 *   EBVO_TOED_HYBRID  a separable fp32 screen selects a superset of the NMS maxima (tolerances 1.9 - 2.3 x the screen's
 *                     WORST-CASE rounding-error bound, which tools/screen_error_bound.py derives from the tap tables and
 *                     toed_kernels.hip asserts at compile time; two orders of magnitude above the error observed on
 *                     images, see ebvo_toed_screen_audit), and only those pixels are evaluated in the
 *                     reference's exact arithmetic.  Same bits out, ~3x less work; see toed_kernels.hip.  An image on
 *                     which the screen flags more grid points than the context's max_h * max_w (possible only when
 *                     most of the image is exact ties, e.g. a one-pixel checkerboard) is re-run on the strict path by
 *                     the library; ebvo_toed_fallbacks counts those.
 * The default is EBVO_TOED_HYBRID unless the environment variable EBVO_TOED_MODE is "strict". */
enum
{
    EBVO_TOED_STRICT = 0,
    EBVO_TOED_HYBRID = 1
};
int ebvo_set_toed_mode(ebvo_ctx *ctx, int mode);
int ebvo_get_toed_mode(const ebvo_ctx *ctx);
int64_t ebvo_toed_fallbacks(const ebvo_ctx *ctx); /* hybrid runs the library repeated on the strict path so far */
/* ebvo_stereo_submit captures the 18 launches of a pair into a hipGraph at the THIRD submission of a slot with unchanged
 * size, parameters and buffers, and launches that graph afterwards (environment EBVO_GRAPHS=0 or ebvo_debug_set(ctx, 10, 0):
 * direct launches).  Number of pairs submitted as a graph launch so far: */
int64_t ebvo_graph_launches(const ebvo_ctx *ctx);
/* diagnostics of the last TOED run on a slot: per image {all NMS maxima, kept edges, screened candidates (hybrid),
 * distinct neighbour grid points whose exact magnitude was evaluated (hybrid)} */
int ebvo_toed_stats(ebvo_ctx *ctx, int slot, int32_t out[8]);

/* Diagnostic of the hybrid detector's FP32 screen (not on any hot path): runs the detector on one image with a variant of the
 * screen kernel that keeps its gx, gy, |g|, and compares them with the exact stage's values at every candidate (and |g| at
 * every neighbour grid point of the screened interior).  The superset property of the screen (hybrid == strict, bit for bit:
 * src/toed/cpu_toed.cpp:406-483 decided from exact values only) rests on max_err_* <= bound_*; tests assert it on full-size
 * images.  Returns EBVO_ERR_CAPACITY when the screen flagged more grid points than the context holds (nothing to audit). */
typedef struct ebvo_screen_audit
{
    int32_t n_candidates;       /* grid points the screen selected */
    int32_t n_maxima;           /* ... of which the exact NMS accepted (Total_Num_Of_TOED) */
    int32_t n_kept;             /* ... inside the 10-px border (toed_edges.size()) */
    int32_t n_neighbour_points; /* distinct neighbour grid points evaluated exactly */
    double max_err_gx, max_err_gy, max_err_mag; /* over the candidates: |screen - exact| */
    double max_err_mag_neighbours;              /* over the neighbour points */
    double bound_g, bound_mag, bound_slope;     /* worst-case budget E_G, E_M, E_S (tools/screen_error_bound.py) */
    double tol_mag, tol_slope;                  /* the relaxed test's tolerances TOL_M, TOL_S */
} ebvo_screen_audit;
int ebvo_toed_screen_audit(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, ebvo_screen_audit *out);

/*
 * Replaces ThirdOrderEdgeDetectionCPU::get_Third_Order_Edges(cv::Mat)
 * (src/toed/cpu_toed.cpp:66-77: preprocessing + convolve_img + non_maximum_suppresion), called
 * by Pipeline::ProcessEdges (src/Pipeline.cpp:24-29).
 *   img/h/w/stride : CV_8UC1 image (stride in bytes)
 *   out/cap        : toed_edges -- edges inside the 10-px border, raster order, index = position
 *   n_kept         : toed_edges.size();  n_total : Total_Num_Of_TOED (all NMS maxima)
 *   all4/cap_all   : optional subpix_edge_pts_final rows (x, y, theta, sub-pixel magnitude)
 *   t_conv/t_nms   : optional time_conv / time_nms in seconds (device time)
 * On EBVO_ERR_CAPACITY n_kept / n_total hold the required sizes.
 */
int ebvo_toed(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, ebvo_edge *out, int cap,
              int *n_kept, int *n_total, double *all4, int cap_all, double *t_conv, double *t_nms);

/* Two images of one size in one launch (left + right of a stereo frame, src/Pipeline.cpp:93,97). */
int ebvo_toed_pair(ebvo_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                   ptrdiff_t stride_left, ptrdiff_t stride_right, ebvo_edge *out_left, ebvo_edge *out_right,
                   int cap, int n_kept[2], int n_total[2]);

/* Replaces Stereo_Matches::CalculateEpipolarLine (src/Stereo_Matches.cpp:10-20): l = F (x, y, 1),
 * F row-major, evaluated (F_i0*x + F_i1*y) + F_i2.  Host-side helper (3 flops per line); the
 * adapter may instead pass the lines Eigen produced. */
int ebvo_epipolar_lines(const double F[9], const ebvo_edge *edges, int n, double *lines /* n x 3 */);

/*
 * Replaces extract_Epipolar_Edge_Indices + apply_Epipolar_Line_Distance_Filtering
 * (src/Stereo_Matches.cpp:91-109, :381-419), apply_Disparity_Filtering (:534-553) and
 * apply_orientation_filter (:863-915); stage_mask selects which predicates apply, so the
 * adapter can expose the three stages separately or fused.
 *   L/nL, R/nR : focused (left) edges and candidate (right) edges
 *   lines      : nL x 3 epipolar line coefficients (a, b, c) of the left edges
 *   row_ptr    : nL + 1 CSR offsets;  col_idx/cap : right-edge indices, ascending per row
 *   n_pairs    : number of pairs (required cap on EBVO_ERR_CAPACITY)
 * col_idx may be NULL with cap 0 to size the output.
 */
int ebvo_epi_candidates(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR,
                        const double *lines, double epi_thr, double max_disp, double orient_thr_deg,
                        int stage_mask, int32_t *row_ptr, int32_t *col_idx, int64_t cap, int64_t *n_pairs);

/* The three geometric stages for a stage-wise caller from ONE device search: the list under the epipolar AND disparity
 * predicates (apply_Epipolar_Line_Distance_Filtering + apply_Disparity_Filtering, :381-419, :534-553) plus one flag per
 * listed pair, orient_ok[k] = the pair passes apply_orientation_filter (:863-915).  The pairs with the flag, in order, are
 * exactly the list ebvo_epi_candidates returns for EBVO_STAGE_ALL; the epipolar-only list (hundreds of candidates per
 * edge) is never formed.  Same calling convention as ebvo_epi_candidates (cap = 0 sizes the list). */
int ebvo_epi_candidates_staged(ebvo_ctx *ctx, const ebvo_edge *L, int nL, const ebvo_edge *R, int nR, const double *lines,
                               double epi_thr, double max_disp, double orient_thr_deg, int32_t *row_ptr, int32_t *col_idx,
                               uint8_t *orient_ok, int64_t cap, int64_t *n_pairs);

/*
 * The same three stages for a caller that runs them ONE AFTER THE OTHER on what the previous stage returned -- main_VO as it
 * is: Pipeline::ProcessEdges x 2 (src/Pipeline.cpp:24-29, :93-97), then apply_Epipolar_Line_Distance_Filtering /
 * apply_Disparity_Filtering / apply_orientation_filter (src/Stereo_Matches.cpp:1374-1399), then apply_NCC_Filtering (:1427).
 * What a stage produced STAYS on the device under a tag; the next stage names it by tag instead of uploading it again, and
 * every result is returned as pointers into page-locked memory owned by the context (valid until the next call of the
 * same function on this context; nothing to free).  The tags of the two most recent ebvo_toed_resident results (one per
 * `which`) are valid until that workspace receives another edge list (ebvo_toed_resident on it, ebvo_toed, ebvo_toed_pair,
 * a pair uploaded into slot 0); the other host-buffer entry points leave them valid, so the SIFT / Best-Nearly-Best /
 * refinement calls get_Stereo_Edge_Pairs makes between the stages do not cost the residency.  A stale tag gives
 * EBVO_ERR_STATE and the caller falls back to the host-buffer entry points above (include/ebvo/adapters.hpp does).
 * Results are bit-identical to ebvo_toed / ebvo_epi_candidates(_staged) / ebvo_ncc_pairs on the same inputs.
 */
typedef struct ebvo_toed_view
{
    const ebvo_edge *edges; /* toed_edges: n_kept records */
    const double *all4;     /* subpix_edge_pts_final: n_total x 4, NULL unless asked for */
    int32_t n_kept, n_total;
    uint64_t tag;           /* names this result in the calls below; never 0 */
    double t_conv, t_nms;   /* seconds of device time */
} ebvo_toed_view;
/* ThirdOrderEdgeDetectionCPU::get_Third_Order_Edges (src/toed/cpu_toed.cpp:66-77) into workspace `which` (0 or 1): the
 * image is uploaded, the edges are detected, copied to page-locked memory AND kept on the device. */
int ebvo_toed_resident(ebvo_ctx *ctx, int which, const uint8_t *img, int h, int w, ptrdiff_t stride, int want_all4,
                       ebvo_toed_view *view);

typedef struct ebvo_candidates_view
{
    const int32_t *row_ptr;   /* n_left + 1 */
    const int32_t *col_idx;   /* n_pairs right-edge indices, ascending per row */
    const uint8_t *orient_ok; /* n_pairs flags (see ebvo_epi_candidates_staged), NULL unless asked for */
    int64_t n_pairs;
    /* with the flags also the list they select, i.e. the lists after apply_orientation_filter (= ebvo_epi_candidates with
     * EBVO_STAGE_ALL): what the next stage takes as its row_ptr / col_idx */
    const int32_t *row_ptr_final; /* n_left + 1 */
    const int32_t *col_idx_final; /* n_final */
    int64_t n_final;
} ebvo_candidates_view;
/* ebvo_epi_candidates / ebvo_epi_candidates_staged on two resident edge lists (left and right may sit in either
 * workspace); only the nL x 3 line coefficients travel to the device. */
int ebvo_epi_candidates_resident(ebvo_ctx *ctx, uint64_t tag_left, uint64_t tag_right, const double *lines, double epi_thr,
                                 double max_disp, double orient_thr_deg, int stage_mask, int want_orient_flags,
                                 ebvo_candidates_view *view);

enum
{
    EBVO_NCC_WANT_LEFT_PATCHES = 1, /* left_edge_patches (src/Stereo_Matches.cpp:578): n_left x 2 x 49 floats */
    EBVO_NCC_WANT_SIMS = 2          /* the four scores of every pair (:592-595) */
};
typedef struct ebvo_ncc_view
{
    const float *left_patches; /* n_left x 98 */
    const double *sims;        /* n_pairs x 4 */
    const double *best;        /* n_pairs: final_SIM_score (:596) */
    const uint8_t *keep;       /* n_pairs: best > thr (:597) */
    int32_t n_left;
    int64_t n_pairs;
} ebvo_ncc_view;
/* apply_NCC_Filtering in its FIRST pass (:1427), where every candidate is a right TOED edge: the pairs are given as CSR
 * indices into the resident right edge list (contributing_edges_toed_indices[0], :413) instead of 32-byte edge records.
 * imgL / imgR are the RAW images (:562-563); they are uploaded (0.5 MB each), the edges are not. */
int ebvo_ncc_pairs_resident(ebvo_ctx *ctx, uint64_t tag_left, uint64_t tag_right, const uint8_t *imgL, const uint8_t *imgR,
                            int h, int w, ptrdiff_t strideL, ptrdiff_t strideR, const int32_t *row_ptr,
                            const int32_t *col_idx, double thr, int want, ebvo_ncc_view *view);

/*
 * Replaces Stereo_Matches::apply_NCC_Filtering (src/Stereo_Matches.cpp:555-616) and under it
 * Utility::get_edge_patches / get_patch_similarity (src/utility.cpp:182-212, :163-180).
 *   imgL/imgR     : the raw left / right CV_8UC1 images (:562-563)
 *   L/nL          : left edges;  row_ptr : nL + 1 CSR offsets into the pair arrays
 *   Rc            : one explicit candidate edge per pair (a TOED edge in the first pass, a
 *                   cluster centre in the second, :588)
 *   left_patches  : optional nL x 2 x 49 floats (plus, minus) -- left_edge_patches (:578)
 *   sims          : optional n_pairs x 4 doubles (pp, nn, pn, np) (:592-595)
 *   best          : optional n_pairs doubles, std::max of the four (:596)
 *   keep          : optional n_pairs bytes, best > thr (:597)
 */
int ebvo_ncc_pairs(ebvo_ctx *ctx, const uint8_t *imgL, const uint8_t *imgR, int h, int w,
                   ptrdiff_t strideL, ptrdiff_t strideR, const ebvo_edge *L, int nL, const ebvo_edge *Rc,
                   const int32_t *row_ptr, double thr, float *left_patches, double *sims, double *best,
                   uint8_t *keep);

/* ---- input side (OpenCV is not part of the reference tree: parity unpinned) -------------- */

/* cv::undistort(img, out, K, dist) as Pipeline::prepare_Stereo_Images calls it (src/Pipeline.cpp:78-79: no new camera
 * matrix; K = fx fy cx cy of include/Dataset.h get_*_calib_matrix_cvMat, dist = k1 k2 p1 p2 [k3], n_dist = 4 or 5,
 * include/Dataset.h:396-397).  OpenCV 4.x restated: stripe-wise maps with 5 fractional bits, fixed-point bilinear
 * remap, constant-0 border.  Zero distortion returns the input image. */
int ebvo_undistort(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, const double K[4], const double *dist,
                   int n_dist, uint8_t *out, ptrdiff_t out_stride);

/* The resident pipeline undistorts the uploaded pair itself: TOED, the refinement and the output geometry then run on the
 * undistorted images while the NCC passes sample the RAW ones, exactly as the reference does (src/Pipeline.cpp:93-97 vs
 * src/Stereo_Matches.cpp:562-563).  p = NULL switches it off (KITTI, ETH3D: zero distortion, same result either way). */
typedef struct ebvo_undistort_params
{
    double K_left[4], K_right[4];       /* fx fy cx cy */
    double dist_left[5], dist_right[5]; /* k1 k2 p1 p2 k3 */
    int n_dist;                         /* 4 or 5 */
    int reserved;
} ebvo_undistort_params;
int ebvo_stereo_set_undistort(ebvo_ctx *ctx, const ebvo_undistort_params *p);

/* ---- fixed-scale SIFT descriptors (cv::SIFT is third-party: parity unpinned) --------------- */

#define EBVO_SIFT_THRESHOLD 500.0 /* include/definitions.h:40 */
#define EBVO_SIFT_DESC_LEN 128

/*
 * Replaces the cv::SIFT::create()->compute(image, {kp1, kp2}, desc) calls of Stereo_Matches::augment_Edge_Data
 * (src/Stereo_Matches.cpp:655-689), apply_SIFT_filtering (:716-727) and finalize_stereo_edge_mates (:1627-1635): the
 * descriptors at the two points 8 px either side of every edge (get_Orthogonal_Shifted_Points(edge, 8)), keypoint size 1,
 * angle 180 / pi * theta.  OpenCV 4.x restated: first pyramid level = GaussianBlur(sigma sqrt(1.6^2 - 0.5^2)) of the
 * image, formed ONCE per call instead of once per edge; 4 x 4 x 8 histogram over an 11 x 11 window; values 0 .. 255
 * stored as floats like cv::SIFT's CV_32F descriptors.
 *   img : the undistorted CV_8UC1 image;  desc : n x 2 x 128 floats (plus point, minus point)
 */
int ebvo_sift_descriptors(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, const ebvo_edge *edges, int n,
                          float *desc);

/* The score of Stereo_Matches::apply_SIFT_filtering (:736-740): per candidate pair the smallest of the four L2 distances
 * between the two descriptors of the left edge and the two of the candidate.  left_desc: nL x 2 x 128, cand_desc: one
 * descriptor pair per PAIR (n_pairs x 2 x 128, the order of the CSR lists); dist: n_pairs doubles.  The filter keeps
 * dist < EBVO_SIFT_THRESHOLD and stores dist as refine_confidences (:752-757). */
int ebvo_sift_min_distances(ebvo_ctx *ctx, const float *left_desc, int nL, const float *cand_desc, const int32_t *row_ptr,
                            double *dist);

/* ---- photometric refinement (no fixture of the reference pins it) -------------------- */

/* Defaults of Stereo_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton_along_EpipolarLine
 * (include/Stereo_Matches.h:79-84). */
#define EBVO_GN_MAX_ITER 20
#define EBVO_GN_TOL 1e-3
#define EBVO_GN_HUBER_DELTA 3.0

typedef struct
{
    int max_iter;       /* >= 1 */
    double tol;         /* |delta| below which the iteration stops */
    double huber_delta; /* Huber threshold; outlier if rms > 2 * huber_delta */
} ebvo_gn_params;

void ebvo_gn_default_params(ebvo_gn_params *p);

/* util_compute_Img_Gradients (include/utility.h:131-141; called from src/Pipeline.cpp:83-84): cv::Sobel 3x3 with
 * scale 1/8 and OpenCV's default border (reflect-101) of the CV_8UC1 image converted to CV_32F.
 * gx, gy: h x w floats, tightly packed. */
int ebvo_sobel_gradients(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride, float *gx, float *gy);

/*
 * Replaces the per-candidate body of Stereo_Matches::refine_edge_disparity (src/Stereo_Matches.cpp:1290-1358) and
 * under it min_Edge_Photometric_Residual_by_Gauss_Newton_along_EpipolarLine (:1159-1288): 1-D Gauss-Newton on the
 * photometric residual of the two 7x7 side patches, the right patch pair sliding along the epipolar direction
 * (-b, a)/|(-b, a)| of the left edge's line (:1331-1336), Huber weights, init_alpha 0.
 *   imgL/imgR   : the undistorted left / right CV_8UC1 images (the reference converts them to CV_32F, :1292-1294;
 *                 pass them swapped for is_left = false).  The gradients of imgR are formed internally.
 *   L/nL, lines : left edges and their nL x 3 line coefficients (epip_line_coeffs_of_left_edges)
 *   row_ptr     : nL + 1 CSR offsets into the pair arrays;  cand_xy : n_pairs x 2 candidate locations
 *                 (EdgeCluster::center_edge.location)
 * Outputs, one per pair: alpha (refined_alpha), score (refine_final_scores: final RMS), confidence
 * (refine_confidences: exp(-rms / huber_delta)), validity (refine_validities: 0 / 1), iters (iterations
 * executed) and refined_xy (the updated centre location, :1349-1351).  validity = 2 marks the case the reference
 * leaves undefined: it stops on H < 1e-8 without assigning its outputs (:1255); score and confidence are NaN there.
 */
int ebvo_gn_refine_stereo(ebvo_ctx *ctx, const uint8_t *imgL, const uint8_t *imgR, int h, int w, ptrdiff_t strideL,
                          ptrdiff_t strideR, const ebvo_edge *L, int nL, const double *lines, const int32_t *row_ptr,
                          const double *cand_xy, const ebvo_gn_params *params, double *alpha, double *score,
                          double *confidence, uint8_t *validity, int32_t *iters, double *refined_xy);

/*
 * Replaces Temporal_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton (src/Temporal_Matches.cpp:735-851) as
 * apply_photometric_refinement_quads calls it (:572-634, once for the left and once for the right image of a quad):
 * 2-D Gauss-Newton on the photometric residual between the keyframe edge's side patches and the current-frame edge's,
 * the current-frame patch pair (oriented by the CURRENT-frame edge) placed at kf.location - d.
 *   imgKF / imgCF : undistorted keyframe / current-frame CV_8UC1 images of one camera; gradients of imgCF are formed
 *                   internally (current_frame.*_image_gradients_*)
 *   kf, cf, init_disp : n keyframe edges, n current-frame edges, n x 2 initial disparities (kf.location - cf.location)
 * Outputs per item: disp (refined_disparity, n x 2), score (final RMS), validity (0 / 1), iters.  The update solves the
 * 2x2 system with Eigen 3.4's pivoted LDL^T as published (H.ldlt().solve(b), :818).
 */
int ebvo_gn_refine_temporal(ebvo_ctx *ctx, const uint8_t *imgKF, const uint8_t *imgCF, int h, int w, ptrdiff_t strideKF,
                            ptrdiff_t strideCF, const ebvo_edge *kf, const ebvo_edge *cf, const double *init_disp, int n,
                            const ebvo_gn_params *params, double *disp, double *score, uint8_t *validity, int32_t *iters);

/* ---- stage glue on CSR candidate lists (no fixture of the reference pins it) ----------- */

#define EBVO_BNB_NCC 0.9  /* include/definitions.h:34 */
#define EBVO_BNB_SIFT 0.4 /* :33 */

/*
 * Stereo_Matches::apply_Best_Nearly_Best_Test (src/Stereo_Matches.cpp:789-862) on every row of a CSR list.
 *   scores           : one per pair -- refine_final_scores with higher_is_better = 1 (is_NCC), refine_confidences
 *                      (SIFT distances) with higher_is_better = 0
 *   new_count[nL]    : survivors per row;  order[n_pairs] : order[row_ptr[i] + k] = pair index of the k-th survivor of
 *                      row i (k < new_count[i]), in the order the reference leaves them: by score when something was
 *                      dropped, untouched otherwise (:840).  The order among equal scores is std::sort's, i.e. libstdc++'s
 *                      introsort restated move for move (csrc/ebvo_sort.h): original position for rows of at most 16
 *                      candidates, wherever its partitioning leaves them for longer rows.
 */
int ebvo_bnb_test(ebvo_ctx *ctx, const int32_t *row_ptr, int nL, const double *scores, double ratio_thr,
                  int higher_is_better, int32_t *new_count, int32_t *order);

/* Stereo_Matches::apply_Lowe_Ratio_Test as written (:916-964): only the best-scoring candidate of each row survives
 * (the first one on ties; candidate 0 if no score exceeds -1).  Same outputs as ebvo_bnb_test. */
int ebvo_keep_best(ebvo_ctx *ctx, const int32_t *row_ptr, int nL, const double *scores, int32_t *new_count,
                   int32_t *order);

/*
 * Stereo_Matches::shift_Edge_to_Epipolar_Line (:26-89) for every candidate of every row, i.e. the shift-only pass of
 * consolidate_redundant_edge_hypothesis (:976-996): cand / shifted hold n_pairs edges (x, y, theta; index is returned
 * 0 like the reference's Edge{loc, theta, false, 0}), lines the nL x 3 epipolar coefficients of the left edges.
 * tan(theta) is formed as sin / cos of the library's correctly rounded pair, pow(x, 2) as x * x.
 */
int ebvo_epipolar_shift(ebvo_ctx *ctx, const ebvo_edge *cand, const double *lines, const int32_t *row_ptr, int nL,
                        ebvo_edge *shifted);

/*
 * EdgeClusterer::performClustering (src/EdgeClusterer.cpp:119-302) on every row of a CSR list, as the clustering pass of
 * consolidate_redundant_edge_hypothesis runs it (src/Stereo_Matches.cpp:1006-1034): single-linkage merging of the
 * candidates' locations below CLUSTER_DIST_THRESH = 1 px (and |dtheta| < 20 deg if by_orientation), clusters capped at
 * MAX_CLUSTER_SIZE = 10, one Gaussian-weighted average edge per cluster.  skip_single = 1 leaves rows with one
 * candidate untouched (the cluster-only call, :998-999).
 *   new_count[nL]      : clusters per row
 *   centres[n_pairs]   : centres[row_ptr[i] + c] = centre edge of cluster c of row i (c < new_count[i]); the rest of a
 *                        row's slots is unspecified
 *   cluster_of[n_pairs]: cluster index (0 .. new_count[i] - 1) of every input candidate: its contributing_edges
 * Merging decisions are exact; the centres carry the device exp() in their weights (within 1e-12 px of glibc's).
 */
int ebvo_cluster_rows(ebvo_ctx *ctx, const ebvo_edge *cand, const int32_t *row_ptr, int nL, int by_orientation,
                      int skip_single, int32_t *new_count, ebvo_edge *centres, int32_t *cluster_of);

/* ---- finalisation geometry (what the output file holds) ----------------- */

typedef struct
{
    double K_left[9], K_right[9]; /* row-major 3x3 calibration matrices (Dataset::get_left/right_calib_matrix) */
    double R21[9], T21[3];        /* get_relative_rot_left_to_right / get_relative_transl_left_to_right */
} ebvo_stereo_calib;

/*
 * The numeric body of Stereo_Matches::write_finalized_stereo_edge_pairs_to_file (src/Stereo_Matches.cpp:1656-1699)
 * with Utility::backproject_2D_point_to_3D_point_using_rays, reconstruct_3D_Tangent_through_intersection_of_planes
 * and project_3D_Tangent_to_2D_Tangent (src/utility.cpp:95-119): per final (left edge, right edge) pair the 16
 * numbers of one output row, out16[k] = {lx, ly, ltheta, rx, ry, rtheta, Gamma.x, .y, .z, T.x, .y, .z,
 * projected_T_1.x, .y, projected_T_2.x, .y}.  The text itself (header line, 6 significant digits, spaces) is written
 * by the caller (ebvo::write_finalized_stereo_edge_pairs in include/ebvo/adapters.hpp uses the reference's iostream
 * calls).
 */
int ebvo_finalize_pairs(ebvo_ctx *ctx, const ebvo_stereo_calib *calib, const ebvo_edge *left, const ebvo_edge *right,
                        int n, double *out16);

/*
 * The same refinement on the pair resident in `slot` after ebvo_stereo_run / ebvo_stereo_wait, without leaving the
 * device: every candidate pair the NCC filter kept (keep[k] = 1) is refined against its right TOED edge, using the
 * pipeline's own edge lists, lines and images.  Outputs are indexed like sims / keep (n_pairs entries); pairs that
 * were not kept get validity 255, alpha 0 and their unrefined location.  Any output pointer may be NULL.
 */
int ebvo_stereo_refine(ebvo_ctx *ctx, int slot, const ebvo_gn_params *params);
int ebvo_stereo_fetch_refined(ebvo_ctx *ctx, int slot, double *alpha, double *score, double *confidence,
                              uint8_t *validity, int32_t *iters, double *refined_xy);

/*
 * The stages of get_Stereo_Edge_Pairs after the NCC pass (src/Stereo_Matches.cpp:1418-1481), without SIFT, on the pair
 * resident in `slot` and without the candidate lists leaving the device:
 *   [SIFT filter ->] kept NCC matches -> apply_Best_Nearly_Best_Test(BNB_NCC) [-> (BNB_SIFT)] -> epipolar shift ->
 *   refine_edge_disparity -> epipolar shift + clustering by orientation (:1483 as its arguments bind) ->
 *   apply_NCC_Filtering on the cluster centres -> best candidate per row -> rows with a match
 * and, if calib is given, the 16 numbers of the output file per final pair.  Each stage is the kernel behind the
 * corresponding host-buffer entry point (ebvo_bnb_test, ebvo_epipolar_shift, ebvo_gn_refine_stereo,
 * ebvo_cluster_rows, ebvo_ncc_pairs, ebvo_keep_best, ebvo_finalize_pairs), so the result equals chaining those calls
 * on the fetched data.  With use_sift the SIFT filter (:1414) and the BNB test on the SIFT distances (:1452) run too
 * (ebvo_sift_descriptors / ebvo_sift_min_distances on the resident pair): the chain is then get_Stereo_Edge_Pairs stage for
 * stage.  (The NCC of a pair does not depend on which other pairs survive, so filtering by SIFT after the first NCC pass
 * of ebvo_stereo_run selects the same pairs, with the same scores, as the reference's SIFT-then-NCC order.)
 */
typedef struct
{
    double bnb_ratio; /* EBVO_BNB_NCC */
    double ncc_thr;   /* EBVO_NCC_THRESH, second pass */
    ebvo_gn_params gn;
    int use_sift;     /* 1: SIFT filter (:1414) and Best-Nearly-Best on the SIFT distances (:1452) are part of the chain */
    int reserved;
    double sift_thr;  /* EBVO_SIFT_THRESHOLD */
    double bnb_sift;  /* EBVO_BNB_SIFT */
} ebvo_finalize_params;

typedef struct
{
    int32_t n_ncc, n_bnb, n_clusters, n_ncc2, n_final; /* candidates surviving each stage (n_ncc: SIFT and NCC filter) */
    int32_t n_sift;                                    /* candidates passing the SIFT filter alone (use_sift) */
} ebvo_finalize_counts;

void ebvo_finalize_default_params(ebvo_finalize_params *p); /* the reference's constants, use_sift = 0 */
int ebvo_stereo_finalize(ebvo_ctx *ctx, int slot, const ebvo_finalize_params *params, const ebvo_stereo_calib *calib,
                         ebvo_finalize_counts *counts);
/* The same chain without blocking the caller: _submit ENQUEUES every stage on the slot's stream (the stage totals stay on
 * the device; no count is read back in between) and returns; _wait blocks until the chain has run and returns the
 * counts.  A frame loop keeps several slots in flight: the chains of different slots run concurrently on the device.
 * Between the two calls the slot accepts no other call (EBVO_ERR_STATE). */
int ebvo_stereo_finalize_submit(ebvo_ctx *ctx, int slot, const ebvo_finalize_params *p, const ebvo_stereo_calib *calib);
int ebvo_stereo_finalize_wait(ebvo_ctx *ctx, int slot, ebvo_finalize_counts *counts);
/* n_final entries each: index of the left TOED edge, the matched right centre edge, its NCC score, and (if calib was
 * given) the 16 numbers of the output row.  Any pointer may be NULL. */
int ebvo_stereo_fetch_final(ebvo_ctx *ctx, int slot, int32_t *left_index, ebvo_edge *right_edge, double *ncc_score,
                            double *out16);

/* Utility::get_edge_patches for n edges on one image: patches = n x 2 x 49 floats
 * (src/utility.cpp:182-212; used again by finalize_stereo_edge_mates, src/Stereo_Matches.cpp:1622). */
int ebvo_edge_patches(ebvo_ctx *ctx, const uint8_t *img, int h, int w, ptrdiff_t stride,
                      const ebvo_edge *edges, int n, float *patches);

/* Utility::get_patch_similarity over n explicit pairs of stored 7x7 float patches
 * (src/utility.cpp:163-180); also the MatlabNCCComputer::computeNCC-shaped entry
 * (include/MatlabNCCComputer.h:41): sim[k] = ncc(A[k], B[k]). */
int ebvo_ncc_patches(ebvo_ctx *ctx, const float *A, const float *B, int n, double *sim);

/* Replaces the scoring loop of Temporal_Matches::apply_NCC_filtering_quads
 * (src/Temporal_Matches.cpp:426-468): per quad k, stored (plus, minus) patches of the keyframe
 * mate and the current-frame mate on the left and on the right (n x 2 x 49 floats each);
 * sim_left / sim_right = max of the four NCCs in the order (:441-450); keep = both > thr. */
int ebvo_ncc_quads(ebvo_ctx *ctx, const float *kfL, const float *kfR, const float *cfL, const float *cfR,
                   int n, double thr, double *sim_left, double *sim_right, uint8_t *keep);

/* ---- temporal quads on the resident pairs (Temporal_Matches; BASELINE configs[2]) -------------------------------- */
/* The part of Temporal_Matches::get_Temporal_Edge_Pairs_from_Quads (src/Temporal_Matches.cpp:168-218) that does not
 * need ground-truth poses: for every stereo mate of the keyframe the candidate mates of the current frame
 * (apply_spatial_grid_filtering_quads :335-383: left edge within +-ceil(radius / cell) grid cells of the keyframe
 * mate's left edge AND right edge likewise, whole cells as SpatialGrid::getCandidatesWithinRadius returns them;
 * apply_orientation_filtering_quads :385-414: both orientation differences within the threshold) and
 * apply_NCC_filtering_quads (:416-469) on the mates' stored patches: left_edge_patches from the raw left image, right
 * patches from the undistorted right image at the final right edge (src/Stereo_Matches.cpp:1621-1622).
 * Mates = the final pairs of ebvo_stereo_finalize on a slot.  The candidates of a keyframe mate are listed in the order
 * of the reference's `left_candidates`: neighbour cell by neighbour cell (dy outer, dx inner), ascending mate index
 * within a cell (include/Dataset.h:92-113). */
typedef struct ebvo_temporal_params
{
    int cell_size;         /* GRID_SIZE 15, include/definitions.h:45 */
    int stages;            /* 0: through the NCC filter (src/Temporal_Matches.cpp:184-192);
                              1: the rest of get_Temporal_Edge_Pairs_from_Quads as well (:196-215): SIFT filter, Best-Nearly-Best
                                 on the NCC and on the SIFT scores, photometric refinement of both cameras, edge clustering */
    double grid_radius;    /* 30, src/Temporal_Matches.cpp:184 */
    double orient_thr_deg; /* 10, :188 */
    double ncc_thr;        /* EBVO_NCC_THRESH_TEMPORAL 0.8, :192 */
    double sift_thr;       /* 200, :196: both min-of-four descriptor distances (left and right camera) below it */
    double bnb_ncc;        /* 0.8, :200 */
    double bnb_sift;       /* 0.8, :204 */
    ebvo_gn_params gn;     /* 20 iterations, 1e-3, Huber 3.0, :612-617 */
} ebvo_temporal_params;
typedef struct ebvo_temporal_counts
{
    int32_t n_kf, n_cf;   /* keyframe / current-frame mates */
    int64_t n_candidates; /* quads after the grid and orientation filters */
    int64_t n_kept;       /* quads with both NCC maxima above the threshold */
    int64_t n_sift, n_bnb_ncc, n_bnb_sift; /* stages = 1: quads surviving each later filter */
    int64_t n_refined_valid;               /* quads whose refinement is valid in both cameras (they all stay, :618) */
    int64_t n_final;                       /* quads after the edge clustering */
} ebvo_temporal_counts;
void ebvo_temporal_default_params(ebvo_temporal_params *p);
/* the final mates of `slot` (after ebvo_stereo_finalize) become the keyframe (src/Pipeline.cpp:133-138: frame 0) */
int ebvo_temporal_set_keyframe(ebvo_ctx *ctx, int slot);
/* quads of the keyframe against the final mates of `slot` */
int ebvo_temporal_match(ebvo_ctx *ctx, int slot, const ebvo_temporal_params *p, ebvo_temporal_counts *counts);
/* The same without blocking the caller (see ebvo_stereo_finalize_submit): _submit enqueues the candidate search and the NCC
 * of the stored patches on the slot's stream and returns, _wait returns the counts (and, for stages = 1, runs the rest of
 * the chain).  Frames in different slots overlap on the device. */
int ebvo_temporal_match_submit(ebvo_ctx *ctx, int slot, const ebvo_temporal_params *p);
int ebvo_temporal_match_wait(ebvo_ctx *ctx, int slot, ebvo_temporal_counts *counts);
/* The quads that leave the whole chain (stages = 1), CSR over the keyframe mates: row_ptr n_kf + 1; per final quad the
 * current-frame mate it is copied from (`best_idx` of :713), the left cluster centre (EdgeClusterer's weighted average, or
 * the refined left edge of an unclustered quad), the right centre (the mean of the members' refined right edges), the left
 * NCC and SIFT scores, the final residuals of both refinements and refine_validity.  Any pointer may be NULL. */
int ebvo_temporal_fetch_final(ebvo_ctx *ctx, int slot, int32_t *row_ptr, int32_t *cf_index, ebvo_edge *left, ebvo_edge *right,
                              double *ncc_left, double *sift_left, double *score_left, double *score_right, uint8_t *valid);
/* row_ptr: n_kf + 1; col_idx / sim_left / sim_right / keep: n_candidates.  Any pointer may be NULL. */
int ebvo_temporal_fetch(ebvo_ctx *ctx, int slot, int32_t *row_ptr, int32_t *col_idx, double *sim_left, double *sim_right,
                        uint8_t *keep);

/* ---------------------------------------------------------------------------------------- */
/* Device-resident stereo pipeline: TOED(left) + TOED(right) + candidates + NCC of one pair,  */
/* images and every intermediate in HBM.  This is what bench.py times.                        */
/* ---------------------------------------------------------------------------------------- */

typedef struct ebvo_stereo_params
{
    double F21[9]; /* row-major fundamental matrix, Dataset::get_fund_mat_21 */
    double epi_thr, max_disp, orient_thr_deg, ncc_thr;
    int stage_mask;
    int reserved; /* flags: 0, or EBVO_PAIR_NO_SIMS | EBVO_PAIR_PUSH / EBVO_PAIR_PACK | EBVO_PAIR_PUSH_THETA */
} ebvo_stereo_params;
/* ebvo_stereo_params.reserved = EBVO_PAIR_NO_SIMS: the resident pipeline stores, per candidate pair, the final score (the
 * maximum of the four similarities: what the reference keeps, refine_final_scores, src/Stereo_Matches.cpp:596-600) and the
 * keep flag, not the four similarities themselves (32 of the 41 bytes the NCC stage writes per pair).  The arithmetic is the
 * same; asking for `sims` of such a pair is EBVO_ERR_STATE. */
#define EBVO_PAIR_NO_SIMS 1
/* EBVO_PAIR_PUSH: the pair's chain ENDS with a kernel that writes the compact results (the arrays of
 * ebvo_stereo_fetch_compact_begin's default selection; + EBVO_PAIR_PUSH_THETA: the orientations too) straight into a page-locked
 * arena of the slot, over PCIe: when ebvo_stereo_wait returns they are in host memory -- no copy call, no second stream, no event.
 * ebvo_stereo_pushed_view hands out the pointers; they stay valid until the slot's next submission.  What a frame loop that
 * consumes the first NCC pass on the host wants (src/Stereo_Matches.cpp:1427 onwards run there). */
#define EBVO_PAIR_PUSH 2
#define EBVO_PAIR_PUSH_THETA 4
/* EBVO_PAIR_PACK: the chain ends with the same kernel, writing the block into DEVICE staging; ebvo_stereo_fetch_compact_begin is
 * then ONE device-to-host copy of exactly the bytes the pair produced (the copy engine moves a contiguous block ~1.3x faster than
 * a kernel's stores cross PCIe on the boxes measured).  Exclusive with EBVO_PAIR_PUSH; EBVO_PAIR_PUSH_THETA adds the orientations. */
#define EBVO_PAIR_PACK 8

typedef struct ebvo_stereo_counts
{
    int32_t n_left, n_right;             /* toed_edges.size() of each image */
    int32_t n_total_left, n_total_right; /* Total_Num_Of_TOED of each image */
    int64_t n_pairs;                     /* candidates after the geometric filters */
    int64_t n_matches;                   /* pairs with max NCC > ncc_thr */
} ebvo_stereo_counts;

void ebvo_stereo_default_params(ebvo_stereo_params *p);
/* copy one stereo pair into HBM (not part of the timed region) */
int ebvo_stereo_upload(ebvo_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                       ptrdiff_t stride_left, ptrdiff_t stride_right);
/* run the whole hot path on the resident pair (= submit + wait on slot 0) */
int ebvo_stereo_run(ebvo_ctx *ctx, const ebvo_stereo_params *p, ebvo_stereo_counts *counts);
/* fetch the results of the last ebvo_stereo_run; any pointer may be NULL */
int ebvo_stereo_fetch(ebvo_ctx *ctx, ebvo_edge *left, ebvo_edge *right, int32_t *row_ptr, int32_t *col_idx,
                      double *sims, double *best, uint8_t *keep, float *left_patches);

/* Several pairs in flight from one host thread: a slot is a stereo pair's workspace plus its own HIP stream.
 * ebvo_stereo_submit enqueues the whole hot path of the slot's resident pair WITHOUT any host synchronisation
 * (every size stays in device memory; launches are sized by capacities) and returns at once; ebvo_stereo_wait
 * blocks until that pair is done and returns its counts.  Kernels of different slots overlap on the GPU.
 * Slot 0 always exists and is the one the host-buffer entry points above use. */
int ebvo_stereo_set_slots(ebvo_ctx *ctx, int n_slots);
int ebvo_stereo_upload_slot(ebvo_ctx *ctx, int slot, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                            ptrdiff_t stride_left, ptrdiff_t stride_right);
/* The same upload without blocking the caller (round 4), for a frame loop that reads frame k + 1 while frame k is matched
 * (src/Pipeline.cpp:77-99, cmd/main_VO.cpp:99-113).
 * PULL FORM (both images inside a range registered with ebvo_host_register -- the caller's frame ring, registered once;
 * memory that is page-locked already, hipHostMalloc, may be registered too): the call only records where the images lie (no
 * runtime call, < 1 us) and the pair's own chain starts with a kernel that READS them from the caller's memory -- no copy
 * engine, no second stream, no event.  The images must stay unchanged until the pair's ebvo_stereo_wait has returned, and
 * registered as long as the slot is submitted with them.  Images anywhere else (pageable memory) are copied into the slot's
 * own page-locked staging BEFORE the call returns -- the caller may free or overwrite them at once -- and go up from there on
 * the context's upload stream (ebvo_ingest_stats tells which form the calls took). */
int ebvo_stereo_upload_async(ebvo_ctx *ctx, int slot, const uint8_t *img_left, const uint8_t *img_right, int h, int w,
                             ptrdiff_t stride_left, ptrdiff_t stride_right);
/* page-lock / release caller memory (hipHostRegister / hipHostUnregister): no HIP header needed on the host side */
int ebvo_host_register(ebvo_ctx *ctx, void *p, size_t bytes);
int ebvo_host_unregister(ebvo_ctx *ctx, void *p);
/* diagnostics: ebvo_stereo_upload_async calls so far that took {the pull form, the upload stream} */
int ebvo_ingest_stats(const ebvo_ctx *ctx, int64_t out[2]);
int ebvo_stereo_submit(ebvo_ctx *ctx, int slot, const ebvo_stereo_params *p);
int ebvo_stereo_wait(ebvo_ctx *ctx, int slot, ebvo_stereo_counts *counts);
int ebvo_stereo_fetch_slot(ebvo_ctx *ctx, int slot, ebvo_edge *left, ebvo_edge *right, int32_t *row_ptr,
                           int32_t *col_idx, double *sims, double *best, uint8_t *keep, float *left_patches);

/* Results of a finished pair WITHOUT a staging copy on the host: ebvo_stereo_fetch_begin enqueues the device-to-host copies
 * of the selected arrays into page-locked memory owned by the slot and returns at once (the copy engine works while
 * other slots compute); ebvo_stereo_fetch_end waits for them and hands out read-only pointers, valid until the next
 * upload / submit / host-buffer call on that slot.  What the stage-wise reference code consumes per pair is the default
 * selection (both edge lists, the CSR candidate lists, best score and keep flag per pair: ~16 MB at 126 k edges per
 * image); the four individual scores are another 32 bytes per pair. */
enum
{
    EBVO_FETCH_EDGES = 1, /* left, right */
    EBVO_FETCH_CSR = 2,   /* row_ptr, col_idx */
    EBVO_FETCH_BEST = 4,
    EBVO_FETCH_KEEP = 8,
    EBVO_FETCH_SIMS = 16,
    EBVO_FETCH_DEFAULT = 15,
    EBVO_FETCH_ALL = 31
};
typedef struct ebvo_stereo_view
{
    const ebvo_edge *left, *right; /* n_left, n_right */
    const int32_t *row_ptr;        /* n_left + 1 */
    const int32_t *col_idx;        /* n_pairs */
    const double *sims;            /* n_pairs x 4 (pp, nn, pn, np) */
    const double *best;            /* n_pairs */
    const uint8_t *keep;           /* n_pairs */
    int32_t n_left, n_right;
    int64_t n_pairs;
} ebvo_stereo_view; /* pointers of arrays that were not selected are NULL */
int ebvo_stereo_fetch_begin(ebvo_ctx *ctx, int slot, int what);
int ebvo_stereo_fetch_end(ebvo_ctx *ctx, int slot, ebvo_stereo_view *view);

/* The same results in fewer bytes (round 4): (x, y) of every edge as 16 bytes (an edge's index is its position in the list),
 * the orientations as separate arrays on demand, the keep flag of pair k as bit k & 31 of word k >> 5; best stays fp64.
 * 11.6 MB instead of 16.2 MB per KITTI pair for the default selection.  Same protocol as ebvo_stereo_fetch_begin / _end (the two
 * share the slot's page-locked arena: one selection at a time). */
enum
{
    EBVO_COMPACT_XY = 1,
    EBVO_COMPACT_THETA = 2,
    EBVO_COMPACT_CSR = 4,
    EBVO_COMPACT_BEST = 8,
    EBVO_COMPACT_KEEP_BITS = 16,
    EBVO_COMPACT_DEFAULT = 1 | 4 | 8 | 16,
    EBVO_COMPACT_ALL = 31
};
typedef struct ebvo_stereo_compact_view
{
    const double *left_xy, *right_xy;       /* n_left x 2, n_right x 2 */
    const double *left_theta, *right_theta; /* n_left, n_right */
    const int32_t *row_ptr;                 /* n_left + 1 */
    const int32_t *col_idx;                 /* n_pairs */
    const double *best;                     /* n_pairs */
    const uint32_t *keep_bits;              /* 2 * ceil(n_pairs / 64) words */
    int32_t n_left, n_right;
    int64_t n_pairs, n_matches;
} ebvo_stereo_compact_view; /* pointers of arrays that were not selected are NULL */
int ebvo_stereo_pushed_view(ebvo_ctx *ctx, int slot, ebvo_stereo_compact_view *view); /* after a pair submitted with EBVO_PAIR_PUSH */
int ebvo_stereo_fetch_compact_begin(ebvo_ctx *ctx, int slot, int what);
int ebvo_stereo_fetch_compact_end(ebvo_ctx *ctx, int slot, ebvo_stereo_compact_view *view);

/* Per-kernel device timing (HIP events on the slots' streams, accumulated). */
#define EBVO_MAX_KERNELS 24 /* >= the number of kernel ids (ebvo_internal.h) */
typedef struct ebvo_kernel_time
{
    const char *name;
    double ms;        /* accumulated */
    int64_t launches; /* accumulated */
} ebvo_kernel_time;
/* on = 0: off; 1: bracket every kernel launch with HIP events; N > 1: in the device pipeline bracket the kernels of every
 * N-th submitted pair only (the event records themselves cost ~0.2 ms of device timeline per pair). */
int ebvo_profile_enable(ebvo_ctx *ctx, int on);
int ebvo_profile_reset(ebvo_ctx *ctx);
int ebvo_profile_get(ebvo_ctx *ctx, ebvo_kernel_time *out /* EBVO_MAX_KERNELS */, int *n);

/* Test hooks, not part of the drop-in surface.  key 0: attempts of the regrow loop of ebvo_stereo_wait (0 = default 4);
 * key 1: treat the next `value` pair results as "candidate buffers overflowed" (exercises the regrow / give-up path);
 * key 2: lanes -- from four slots on, ebvo_stereo_submit deals the kernels of the pairs round-robin to
 *        min(lanes, slots - 1) streams of the context instead of one stream per slot (default 4, the measured optimum
 *        on MI355X; 0 = one stream per slot whatever their number).
 * key 3: the profiler instruments ONE stage (value = its index in ebvo_profile_get's order + 1; 0 = every stage): no
 *        event markers between the other kernels, so the stage is timed as it runs in the unprofiled pipeline.
 * key 4: 1 = the photometric refinements never use their eight-lanes-per-pair launch layout (0 = default: chosen per
 *        iteration from the number of active pairs); key 5: that threshold (0 = built-in).  Same bits either way.
 * key 7: 1 = the stereo refinement's eight-lanes layout as a launch per iteration instead of one persistent launch;
 * key 8: waves per SIMD the persistent launch is built for (2 or 3).  Same bits either way.
 * key 6: set the candidate-quad capacity of every slot's temporal stage to `value` (>= 1): the next ebvo_temporal_match
 *        finds more quads than its buffers hold and takes the regrow path.
 * key 10: 0 = ebvo_stereo_submit enqueues the pair as direct launches, 1 = as a captured hipGraph (default).
 * keys 11, 12: grid of toed_exact_centre / toed_exact_mags in blocks (0 = what the device keeps resident);
 * key 13: 1 = ebvo_stereo_upload_async copies on the upload stream instead of the pull kernel; key 14: 1 = lines, boxes,
 *        sin / cos and row pairs as four launches instead of one; key 17: grid of ncc_tile_kernel in blocks (0 = resident);
 *        key 18: grids of decide / cand_scatter / candidates<fill> (512 / 512 / 4096 blocks) divided by `value` (0 = the
 *        default, 4); key 19: most blocks of candidates<count> (0 = the default, 1024).  Same bits either way (A/B switches).
 * key 15: bit mask -- an idempotent kernel of the resident pair's chain is launched TWICE (1 centre, 2 mags, 4 right bank,
 *        8 NCC tile): what one more launch costs the pair rate (tools/gpu_marginal_cost.py).
 * key 16: the resident pair's chain ENDS after stage `value` (0 = whole chain; the pair's record keeps the counts of the last
 *        whole run, the buffers behind the stage are stale): the pair rate of every prefix of the chain
 *        (tools/gpu_prefix_chain.py).  Measurement only. */
int ebvo_debug_set(ebvo_ctx *ctx, int key, int value);

/* Raw FP64 vector-ALU microbenchmark (mul + add, no FMA) used to anchor the compute roofline:
 * returns achieved TFLOP/s over `iters` launches. */
int ebvo_fp64_peak(ebvo_ctx *ctx, int iters, double *tflops_muladd, double *tflops_fma);

#ifdef __cplusplus
}
#endif
#endif /* EBVO_HIP_H */

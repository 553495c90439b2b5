/*
 * ebvo_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (edge_based_visual_odometry_amd/) never does.
 *
 * Every function cites the file:line of Brown-LEMS/Edge_Based_Visual_Odometry it follows.
 * Parity pin: the TOED restatement reproduces, in math_mode = ORC_MATH_LIBM, every
 * known-answer hash recorded from the unmodified reference source in SURVEY.md section 8(c)
 * (tests/test_oracle_kat.py); the NCC restatement reproduces the reference fixture
 * test/ncc_debug_frame1_edge8 (tests/test_oracle_ncc_fixture.py) to its 8-bit precision.
 * PARITY UNPINNED for orc_gn_refine_stereo / orc_sobel_gradients (the reference holds no fixture for its photometric
 * refinement): see tests/test_oracle_gn.py for what anchors them instead.
 */
#ifndef EBVO_ORACLE_H
#define EBVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as ebvo_edge in include/ebvo_hip.h (struct Edge, include/toed/cpu_toed.hpp:26-48). */
typedef struct
{
    double x, y, theta;
    int32_t index;
    int32_t pad;
} orc_edge;

enum
{
    ORC_MATH_PORTABLE = 0, /* csrc/ebvo_math.h atan2 / sincos: bit-identical to the HIP path */
    ORC_MATH_LIBM = 1      /* glibc atan2 / sin / cos, exactly what the reference calls */
};

enum
{
    ORC_STAGE_EPIPOLAR = 1,
    ORC_STAGE_DISPARITY = 2,
    ORC_STAGE_ORIENTATION = 4,
    ORC_STAGE_ALL = 7
};

/*
 * Third-order edge detection of one H x W u8 image.
 *   kept/cap_kept   : edges inside the 10-px border, Edge::index = position (may be NULL to count only)
 *   all4/cap_all    : every NMS maximum as (x, y, theta, subpixel magnitude) rows (may be NULL)
 *   maps            : optional 5 planes of 2H x 2W doubles: Ix, Iy, |grad|, TOx/|TO|, TOy/|TO|
 *   t_conv, t_nms   : seconds (may be NULL)
 * Returns 0, or -1 if a capacity is too small (n_kept / n_total still hold the required sizes).
 */
int orc_toed(const uint8_t *img, int h, int w, ptrdiff_t stride, int math_mode, int nthreads,
             orc_edge *kept, int cap_kept, double *all4, int cap_all, int *n_kept, int *n_total,
             double *maps, double *t_conv, double *t_nms);

/* l = F * (x, y, 1) for each edge; F row-major 3x3; lines = n x 3. */
void orc_epipolar_lines(const double *F, const orc_edge *edges, int n, double *lines);

/*
 * Candidate search + geometric filters, brute force exactly as the reference.
 * row_ptr has nL + 1 entries; col_idx receives right indices in ascending order per row.
 * Returns 0, or -1 if cap is too small (*n_pairs = required size).
 */
/* a later geometric stage (disparity and / or orientation) on existing candidate lists: keep[k] per listed pair */
int orc_filter_pairs(const orc_edge *L, int nL, const orc_edge *R, const int32_t *row_ptr, const int32_t *col_idx,
                     double max_disp, double orient_thr_deg, int mask, int nthreads, uint8_t *keep);
int orc_epi_candidates(const orc_edge *L, int nL, const orc_edge *R, int nR, const double *lines,
                       double epi_thr, double max_disp, double orient_thr_deg, int stage_mask,
                       int nthreads, int32_t *row_ptr, int32_t *col_idx, int64_t cap, int64_t *n_pairs);

/* (plus, minus) 7x7 float patches of n edges: patches = n x 2 x 49. */
void orc_edge_patches(const uint8_t *img, int h, int w, ptrdiff_t stride, const orc_edge *edges, int n,
                      int math_mode, int nthreads, float *patches);

/* NCC of two 7x7 float patches (canonical arithmetic, see ebvo_oracle.c). */
double orc_patch_similarity(const float *a, const float *b);

/* NCC over explicit pairs of stored patches: sim[k] = ncc(A[k], B[k]). */
void orc_ncc_patches(const float *A, const float *B, int n, int nthreads, double *sim);

/*
 * Stereo NCC scoring: for CSR rows i (left edge L[i]) and pairs k in [row_ptr[i], row_ptr[i+1]),
 * candidate edge Rc[k] (explicit edges).  sims = n_pairs x 4 (pp, nn, pn, np), best = n_pairs,
 * keep[k] = best > thr.  left_patches (nL x 2 x 49 floats) may be NULL.
 */
void orc_ncc_pairs(const uint8_t *imgL, const uint8_t *imgR, int h, int w, ptrdiff_t strideL,
                   ptrdiff_t strideR, const orc_edge *L, int nL, const orc_edge *Rc,
                   const int32_t *row_ptr, int math_mode, int nthreads, double thr,
                   float *left_patches, double *sims, double *best, uint8_t *keep);

/* Temporal quad scorer: left = max of 4, right = max of 4 over stored patches (n x 2 x 49 each). */
void orc_ncc_quads(const float *kfL, const float *kfR, const float *cfL, const float *cfR, int n,
                   int nthreads, double thr, double *sim_left, double *sim_right, uint8_t *keep);

/* element-wise math in the chosen mode (for tests of ebvo_math.h) */
void orc_atan2_v(const double *y, const double *x, int n, int math_mode, double *out);
void orc_sincos_v(const double *t, int n, int math_mode, double *s, double *c);
void orc_exp_v(const double *x, int n, int math_mode, double *out);

/* FNV-1a-64 helpers used by the known-answer tests. */
uint64_t orc_fnv1a64(const uint8_t *bytes, size_t n);
/* with_theta != 0: "xy-theta-i" hash (24 raw bytes + index); else "xyi" (16 raw bytes + index). */
uint64_t orc_edge_hash(const orc_edge *e, int n, int with_theta);

/* util_compute_Img_Gradients (include/utility.h:131-141): Sobel 3x3 / 8 with BORDER_REFLECT_101, float planes h x w. */
void orc_sobel_gradients(const uint8_t *img, int h, int w, ptrdiff_t stride, float *gx, float *gy);

/*
 * Photometric Gauss-Newton refinement of each (left edge i, candidate k) pair along the epipolar line of edge i
 * (src/Stereo_Matches.cpp:1159-1358).  cand_xy = n_pairs x 2 right-image locations; outputs per pair: alpha, final
 * RMS score, confidence exp(-rms/huber), validity (0 / 1; 2 = the reference stops on H < 1e-8 without setting its
 * outputs -- score and confidence are NaN here), iterations executed, refined right location.  PARITY UNPINNED.
 */
void orc_gn_refine_stereo(const uint8_t *imgL, const uint8_t *imgR, int h, int w, ptrdiff_t strideL, ptrdiff_t strideR,
                          const orc_edge *L, const double *lines, const int32_t *row_ptr, int nL, const double *cand_xy,
                          int max_iter, double tol, double huber_delta, int math_mode, int nthreads, double *alpha,
                          double *score, double *confidence, uint8_t *validity, int32_t *iters, double *refined_xy);

/*
 * Temporal 2-D photometric refinement (src/Temporal_Matches.cpp:735-851, driven by apply_photometric_refinement_quads
 * :572-634): n independent (keyframe edge, current-frame edge, initial disparity) triples; outputs the refined
 * disparity d (the current-frame location is kf - d), final RMS score, validity, iterations.  PARITY UNPINNED.
 */
void orc_gn_refine_temporal(const uint8_t *imgKF, const uint8_t *imgCF, int h, int w, ptrdiff_t strideKF,
                            ptrdiff_t strideCF, const orc_edge *kf, const orc_edge *cf, const double *init_disp, int n,
                            int max_iter, double tol, double huber_delta, int math_mode, int nthreads, double *disp,
                            double *score, uint8_t *validity, int32_t *iters);

/*
 * The 16 numbers write_finalized_stereo_edge_pairs_to_file (src/Stereo_Matches.cpp:1656-1699) prints per final pair:
 * lx ly ltheta rx ry rtheta Gamma(3) T(3) projected_T_1(2) projected_T_2(2).  K, R21 row-major 3x3, T21 3-vector.
 * PARITY UNPINNED (the reference's sample outputs are missing blobs).
 */
void orc_inverse3(const double *m, double *inv);
void orc_finalize_pairs(const double *K_left, const double *K_right, const double *R21, const double *T21,
                        const orc_edge *L, const orc_edge *R, int n, int math_mode, double *out16);

/* Stage glue on CSR candidate lists (src/Stereo_Matches.cpp:789-862, :916-964, :26-89 + :976-996).  PARITY UNPINNED.
 * Selections are returned as new_count[nL] + order[n_pairs] (order[row_ptr[i] + k] = pair index of the k-th survivor). */
void orc_bnb_test(const int32_t *row_ptr, int nL, const double *scores, double thr, int higher_is_better,
                  int32_t *new_count, int32_t *order);
void orc_keep_best(const int32_t *row_ptr, int nL, const double *scores, int32_t *new_count, int32_t *order);
void orc_epipolar_shift(const orc_edge *cand, const double *lines, const int32_t *row_ptr, int nL, int math_mode,
                        orc_edge *out);

/* EdgeClusterer::performClustering per CSR row (src/EdgeClusterer.cpp:119-302; consolidate_redundant_edge_hypothesis
 * :1006-1034).  PARITY UNPINNED. */
void orc_cluster_rows(const orc_edge *cand, const int32_t *row_ptr, int nL, int by_orientation, int skip_single, int math_mode,
                      int32_t *new_count, orc_edge *centres, int32_t *cluster_of);

/* cv::undistort as src/Pipeline.cpp:78-79 calls it (OpenCV 4.x restated; PARITY UNPINNED).  K = fx fy cx cy. */
void orc_undistort(const uint8_t *img, int h, int w, ptrdiff_t stride, const double K[4], const double *dist, int n_dist,
                   uint8_t *out, ptrdiff_t out_stride);

/* cv::SIFT descriptors at the +-8 px points of every edge and apply_SIFT_filtering's score (OpenCV 4.x restated; PARITY
 * UNPINNED).  desc: n x 2 x 128 floats. */
void orc_sift_kernel13(float k[13]);
void orc_sift_base(const uint8_t *img, int h, int w, ptrdiff_t stride, float *base);
void orc_sift_descriptors(const uint8_t *img, int h, int w, ptrdiff_t stride, const orc_edge *edges, int n, int math_mode,
                          int nthreads, float *desc);
void orc_sift_min_distances(const float *left_desc, const float *cand_desc, const int32_t *row_ptr, int nL, double *dist);

/* temporal candidate quads (src/Temporal_Matches.cpp:335-414), ascending current-frame mate index; PARITY UNPINNED */
int orc_temporal_candidates(const orc_edge *kfL, const orc_edge *kfR, int n_kf, const orc_edge *cfL, const orc_edge *cfR, int n_cf,
                            int img_w, int img_h, int cell, double radius, double orient_thr_deg, int32_t *row_ptr,
                            int32_t *col_idx, int64_t cap, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif

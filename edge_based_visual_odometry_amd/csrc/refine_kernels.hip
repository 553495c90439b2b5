// refine_kernels.hip -- photometric Gauss-Newton refinement of stereo candidates along the epipolar line
// (SURVEY.md 8(f) rank 1), gfx950.
//
//   sobel_kernel      util_compute_Img_Gradients, include/utility.h:131-141 (cv::Sobel 3x3, scale 1/8, reflect-101)
//   gn_init_kernel + gn_iter_kernel
//                     Stereo_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton_along_EpipolarLine,
//                     src/Stereo_Matches.cpp:1159-1288, driven as refine_edge_disparity does (:1290-1358)
//
// Arithmetic contract: every quantity is computed by the same scalar IEEE operations in the same order as the
// reference (doubles, no FMA contraction; the 49-term sums are sequential), so one thread owns one (left edge,
// candidate) pair within an iteration.  Nothing is staged: a thread re-samples instead of storing (the 98 centred left
// samples and the 3 x 98 right samples of an iteration would need ~2 KB per thread); re-evaluating a bilinear sample
// returns the same bits, so the result is unchanged.  exp() (confidence) and cos/sin are csrc/ebvo_math.h's shared
// routines (the oracle's portable mode calls the same code): every output is bit-identical to the restatement.
#include <hip/hip_runtime.h>

#include "ebvo_internal.h"
#include "ebvo_math.h"

namespace
{

__device__ inline int reflect101(int p, int n)
{
    if (n == 1)
        return 0;
    while (p < 0 || p >= n)
        p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// On an image of 8-bit integers the 3x3 sums are integers below 2^11 and the scale is 2^-3: the float result is exact
// whatever order OpenCV's separable filter adds in.
__global__ void sobel_kernel(const uint8_t *__restrict__ img, int h, int w, int pitch, float *__restrict__ gx,
                             float *__restrict__ gy, float2 *__restrict__ gxy)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const uint8_t *r0 = img + (size_t)reflect101(y - 1, h) * pitch, *r1 = img + (size_t)y * pitch,
                  *r2 = img + (size_t)reflect101(y + 1, h) * pitch;
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const int sx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
    const int sy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
    const float fx = (float)sx * 0.125f, fy = (float)sy * 0.125f;
    if (gx)
        gx[(size_t)y * w + x] = fx;
    if (gy)
        gy[(size_t)y * w + x] = fy;
    if (gxy) // interleaved plane for the refinement: both gradients of a corner pair arrive in one 16-byte load
        gxy[(size_t)y * w + x] = make_float2(fx, fy);
}


// ---- cv::undistort (src/Pipeline.cpp:78-79), input side of the path --------------------------
// OpenCV 4.x restated (modules/calib3d/src/undistort.dispatch.cpp: cv::undistort + the scalar loop of
// initUndistortRectifyMap with CV_16SC2 maps; modules/imgproc/src/imgwarp.cpp: remapBilinear, 8-bit, BORDER_CONSTANT 0);
// same restatement as oracle/ebvo_oracle.c: orc_undistort, PARITY UNPINNED.  The maps are never materialised:
//   - the image is cut into stripes of max(1, 4096 / cols) rows, each with its own camera matrix (cy - y0) inverted in
//     closed form (cv::invert, 3x3);
//   - along a row OpenCV advances (_x, _y, _w) by REPEATED ADDITION of the inverse's first column.  With R = I and a
//     pinhole K that column is (ir0, +-0, +-0) and the row start of _x is ir[2] for every row of every stripe, so the
//     running sums _x[j] are one sequence of `cols` doubles: undistort_row_kernel forms it with the same additions (one
//     thread), every pixel thread then picks its entry -- bit-identical to the sequential loop;
//   - u, v -> fixed point with 5 fractional bits (round half to even), integer bilinear weights, (sum + 2^14) >> 15.
struct UndistortArgs
{
    double fx, fy, u0, v0, k1, k2, p1, p2, k3;
    int h, w, ss0;
};

__device__ inline void inv3x3_cv(const double a[9], double b[9])
{
    double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    d = 1. / d;
    b[0] = (a[4] * a[8] - a[5] * a[7]) * d;
    b[1] = (a[2] * a[7] - a[1] * a[8]) * d;
    b[2] = (a[1] * a[5] - a[2] * a[4]) * d;
    b[3] = (a[5] * a[6] - a[3] * a[8]) * d;
    b[4] = (a[0] * a[8] - a[2] * a[6]) * d;
    b[5] = (a[2] * a[3] - a[0] * a[5]) * d;
    b[6] = (a[3] * a[7] - a[4] * a[6]) * d;
    b[7] = (a[1] * a[6] - a[0] * a[7]) * d;
    b[8] = (a[0] * a[4] - a[1] * a[3]) * d;
}

__global__ void undistort_row_kernel(UndistortArgs A, double *__restrict__ xs)
{
    if (blockIdx.x || threadIdx.x)
        return;
    const double Ar[9] = {A.fx, 0, A.u0, 0, A.fy, A.v0, 0, 0, 1};
    double ir[9];
    inv3x3_cv(Ar, ir);
    double _x = 0 * ir[1] + ir[2];
    for (int j = 0; j < A.w; ++j, _x += ir[0])
        xs[j] = _x;
}

__device__ inline int cv_round_sat(double v)
{
    if (!(v > -2147483648.0))
        return (int)(-2147483647 - 1);
    if (!(v < 2147483647.0))
        return 2147483647;
    return (int)rint(v); // round to nearest, ties to even
}

__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t *__restrict__ src, int pitch, UndistortArgs A,
                                                        const double *__restrict__ xs, uint8_t *__restrict__ dst,
                                                        int dpitch)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= A.w || r >= A.h)
        return;
    const int y0 = (r / A.ss0) * A.ss0, i = r - y0;
    const double Ar[9] = {A.fx, 0, A.u0, 0, A.fy, A.v0 - y0, 0, 0, 1};
    double ir[9];
    inv3x3_cv(Ar, ir);
    const double _x = xs[j], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8]; // ir[3], ir[6] are +-0: the row steps leave them
    const double ww = 1. / _w, x = _x * ww, y = _y * ww;
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((A.k3 * r2 + A.k2) * r2 + A.k1) * r2) / (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2);
    const double xd = (x * kr + A.p1 * _2xy + A.p2 * (r2 + 2 * x2) + 0.0 * r2 + 0.0 * r2 * r2);
    const double yd = (y * kr + A.p1 * (r2 + 2 * y2) + A.p2 * _2xy + 0.0 * r2 + 0.0 * r2 * r2);
    const double t0 = 1.0 * xd + 0.0 * yd + 0.0 * 1.0, t1 = 0.0 * xd + 1.0 * yd + 0.0 * 1.0, t2 = 0.0 * xd + 0.0 * yd + 1.0 * 1.0;
    const double invProj = t2 ? 1. / t2 : 1;
    const double u = A.fx * invProj * t0 + A.u0;
    const double v = A.fy * invProj * t1 + A.v0;
    const int iu = cv_round_sat(u * 32), iv = cv_round_sat(v * 32);
    const int sx = (short)(iu >> 5), sy = (short)(iv >> 5);
    const int fxi = iu & 31, fyi = iv & 31;
    const int w00 = (32 - fyi) * (32 - fxi) * 32, w01 = (32 - fyi) * fxi * 32, w10 = fyi * (32 - fxi) * 32, w11 = fyi * fxi * 32;
    const bool cx0 = sx >= 0 && sx < A.w, cx1 = sx + 1 >= 0 && sx + 1 < A.w, ry0 = sy >= 0 && sy < A.h,
               ry1 = sy + 1 >= 0 && sy + 1 < A.h;
    const int v00 = (cx0 && ry0) ? src[(size_t)sy * pitch + sx] : 0;
    const int v01 = (cx1 && ry0) ? src[(size_t)sy * pitch + sx + 1] : 0;
    const int v10 = (cx0 && ry1) ? src[(size_t)(sy + 1) * pitch + sx] : 0;
    const int v11 = (cx1 && ry1) ? src[(size_t)(sy + 1) * pitch + sx + 1] : 0;
    const int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    const int o = (acc + (1 << 14)) >> 15;
    dst[(size_t)r * dpitch + j] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
}

// util_bilinear_Sample_F (include/utility.h:160-173): corner indices and weights, shared by every image sampled at
// one point
__device__ inline void tap_at(double x, double y, int w, int h, int &x0, int &x1, int &y0, int &y1, double &a, double &b)
{
    // std::clamp(x, 0, w - 1) as max / min: the same value for every non-NaN x (a NaN coordinate is undefined behaviour
    // in the reference: it converts floor(NaN) to int); two instructions instead of two compares and four selects
    x = fmin(fmax(x, 0.0), (double)w - 1.0);
    y = fmin(fmax(y, 0.0), (double)h - 1.0);
    const double fx = floor(x), fy = floor(y);
    x0 = (int)fx;
    y0 = (int)fy;
    x1 = min(x0 + 1, w - 1);
    y1 = min(y0 + 1, h - 1);
    a = x - fx; // == x - (double)x0: fx is an integer in [0, w - 1]
    b = y - fy;
}

__device__ inline float blend(double a, double b, float v00, float v10, float v01, float v11)
{
    return (float)((1 - a) * (1 - b) * v00 + a * (1 - b) * v10 + (1 - a) * b * v01 + a * b * v11);
}

// The two corners of a row are adjacent pixels: one 2-byte load (u8 image) or one 16-byte load (interleaved gradient
// plane) fetches both.  The sampling is address-divergent (every lane its own patch), so the number of load
// INSTRUCTIONS is what the texture path charges for; pairing halves it (and the interleaved plane halves it again).
// At the right border x1 == x0 == w - 1: the pair is fetched one pixel to the left and its second element used twice.
struct Corners
{
    float v00, v10, v01, v11;
};

__device__ inline Corners corners_u8(const uint8_t *__restrict__ img, int pitch, int w, int x0, int x1, int y0, int y1)
{
    const int xa = min(x0, max(w - 2, 0));
    unsigned short p0, p1;
    __builtin_memcpy(&p0, img + (size_t)y0 * pitch + xa, 2);
    __builtin_memcpy(&p1, img + (size_t)y1 * pitch + xa, 2);
    const bool shifted = xa != x0; // x0 == x1 == w - 1
    Corners c;
    c.v10 = (float)(p0 >> 8);
    c.v11 = (float)(p1 >> 8);
    c.v00 = shifted ? c.v10 : (float)(p0 & 0xff);
    c.v01 = shifted ? c.v11 : (float)(p1 & 0xff);
    (void)x1;
    return c;
}

__device__ inline float sample_u8(const uint8_t *__restrict__ img, int pitch, int w, int h, double x, double y)
{
    int x0, x1, y0, y1;
    double a, b;
    tap_at(x, y, w, h, x0, x1, y0, y1, a, b);
    if (w < 2)
    {
        const uint8_t *r0 = img + (size_t)y0 * pitch, *r1 = img + (size_t)y1 * pitch;
        return blend(a, b, (float)r0[x0], (float)r0[x1], (float)r1[x0], (float)r1[x1]);
    }
    const Corners c = corners_u8(img, pitch, w, x0, x1, y0, y1);
    return blend(a, b, c.v00, c.v10, c.v01, c.v11);
}

// ---- packed corner planes of the stereo refinement ----------------------------------------------------------------
// The refinement samples three planes (intensity, Sobel gx, Sobel gy) bilinearly at ~3.7 M x 98 points per pair, every
// lane at its own address: the texture addresser retires about one lane address per cycle and CU (GRBM_TA_BUSY 89 %,
// profiles/r02_pmc_sq_chain.txt), so the LOAD COUNT per point is the bound -- it was 6 (two 2-byte intensity loads in
// each of the two passes, two 16-byte gradient loads).  Here every pixel position (y0, x0) gets, precomputed once per
// image, everything a bilinear tap at floor = (x0, y0) reads, with util_bilinear_Sample_F's clamps x1 = min(x0 + 1, w - 1),
// y1 = min(y0 + 1, h - 1) (include/utility.h:166-167) baked in:
//   pix4[y0 * w + x0] = I(x0,y0) | I(x1,y0) << 8 | I(x0,y1) << 16 | I(x1,y1) << 24                 (4 bytes)
//   rec [y0 * w + x0] = one word per corner: I | (8 gx) << 8 | (8 gy) << 19  (two 11-bit signed fields)             (16 bytes)
// (8 x Sobel / 8 is an integer in [-1020, 1020]: 11 bits; the float the reference reads is that integer times 0.125,
// exactly).  The iterations read ONE 16-byte record per sample point (gn_tap) and keep its three floats while the mean
// of the patch forms; pix4 serves the left image, whose patches are sampled once per left edge (gn_left_kernel).
__device__ inline void sobel_at(const uint8_t *__restrict__ img, int h, int w, int pitch, int x, int y, int &sx, int &sy)
{
    const uint8_t *r0 = img + (size_t)reflect101(y - 1, h) * pitch, *r1 = img + (size_t)y * pitch,
                  *r2 = img + (size_t)reflect101(y + 1, h) * pitch;
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    sx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
    sy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
}

__global__ __launch_bounds__(256) void gn_pack_kernel(const uint8_t *__restrict__ img, int h, int w, int pitch,
                                                      uint32_t *__restrict__ pix4, uint4 *__restrict__ rec)
{
    const int x0 = blockIdx.x * 64 + (threadIdx.x & 63), y0 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y0 >= h)
        return;
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const int cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
    unsigned I[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        I[c] = img[(size_t)cy[c] * pitch + cx[c]];
    if (pix4)
        pix4[(size_t)y0 * w + x0] = I[0] | (I[1] << 8) | (I[2] << 16) | (I[3] << 24);
    if (!rec)
        return;
    unsigned wd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        int sx, sy;
        sobel_at(img, h, w, pitch, cx[c], cy[c], sx, sy);
        wd[c] = I[c] | (((unsigned)sx & 0x7ffu) << 8) | (((unsigned)sy & 0x7ffu) << 19);
    }
    rec[(size_t)y0 * w + x0] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
}

// util_bilinear_Sample_F of the intensity at (x, y) from the packed plane
__device__ inline float sample_pix4(const uint32_t *__restrict__ pix4, int w, int h, double x, double y)
{
    int x0, x1, y0, y1;
    double a, b;
    tap_at(x, y, w, h, x0, x1, y0, y1, a, b);
    const uint32_t q = pix4[(size_t)y0 * w + x0];
    return blend(a, b, (float)(q & 0xffu), (float)((q >> 8) & 0xffu), (float)((q >> 16) & 0xffu), (float)(q >> 24));
}

// One bilinear tap of the three right-image planes from the packed corner record: the floats util_bilinear_Sample_F
// returns for the intensity, Sobel gx and Sobel gy at (x, y) -- ONE 16-byte load.
struct GnTap
{
    float v, gx, gy;
};

__device__ inline GnTap gn_tap(const uint4 *__restrict__ rec, int w, int h, double x, double y)
{
    int x0, x1, y0, y1;
    double wa, wb;
    tap_at(x, y, w, h, x0, x1, y0, y1, wa, wb);
    const uint4 q = rec[(size_t)y0 * w + x0]; // the four corners: intensity, 8 gx, 8 gy
    const unsigned wd[4] = {q.x, q.y, q.z, q.w};
    double iv[4], gxc[4], gyc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        iv[c] = (double)(wd[c] & 0xffu);
        gxc[c] = (double)(((int)(wd[c] << 13)) >> 21); // 8 gx: bits 8..18, sign-extended
        gyc[c] = (double)(((int)(wd[c] << 2)) >> 21);  // 8 gy: bits 19..29
    }
    // blend()'s sums with the corner weights formed once.  The gradient planes hold (float)(8 g) * 0.125f; the factor
    // 1/8 goes into the weights instead -- a power of two, so every product and every sum is the same double.
    const double oma = 1 - wa, omb = 1 - wb;
    const double w00 = oma * omb, w10 = wa * omb, w01 = oma * wb, w11 = wa * wb;
    const double s00 = w00 * 0.125, s10 = w10 * 0.125, s01 = w01 * 0.125, s11 = w11 * 0.125;
    GnTap t;
    t.v = (float)(w00 * iv[0] + w10 * iv[1] + w01 * iv[2] + w11 * iv[3]);
    t.gx = (float)(s00 * gxc[0] + s10 * gxc[1] + s01 * gxc[2] + s11 * gxc[3]);
    t.gy = (float)(s00 * gyc[0] + s10 * gyc[1] + s01 * gyc[2] + s11 * gyc[3]);
    return t;
}

struct GnArgs
{
    const uint8_t *imgL, *imgR;
    const float2 *gxy;    // (unused by the stereo kernels: they read the packed planes below)
    const uint32_t *pix4L, *pix4R; // packed corner intensities of the left / right image (gn_pack_kernel)
    const uint4 *recR;             // packed corner intensities + Sobel gradients of the right image
    int h, w;
    const ebvo_edge *L;       // left edges
    const double *lines;      // nL x 3
    const int32_t *pair_left; // pair -> left edge
    const double *cand_xy;    // n_pairs x 2, or NULL: candidate k is the right edge R[col_idx[k]] (device pipeline)
    const ebvo_edge *R;
    const int32_t *col_idx;
    const uint8_t *keep;      // optional: only pairs with keep[k] != 0 are refined, the others get validity 255
    int64_t n_pairs;
    const int32_t *n_pairs_dev; // non-null: the number of pairs lives here, n_pairs is its upper bound
    int max_iter;
    double tol, huber;
    double *alpha, *score, *conf, *refined_xy; // outputs; alpha doubles as the iteration state
    uint8_t *valid;
    int32_t *iters;
    // iteration state
    int nL;
    double *mean_l;   // [2][nL] means of the left plus / minus patches, per LEFT EDGE
    float *left_rec;  // [nL][98] the left samples (the floats util_bilinear_Sample_F returns), sampled once per left edge:
                      // the candidates of one left edge sit in neighbouring lanes and read the same record
    double *sc;       // [2][nL] sin, cos of the left orientation
    int32_t *list[2]; // active pairs, ping-pong
    int32_t *counts;  // [max_iter + 1] active pairs entering iteration it
    int rows_below;   // an iteration entering with at most this many active pairs runs eight lanes per pair
};

// Append to a device list from a block of 256 threads: ONE atomic per block (one per wave made gn_init_kernel last 107 us
// for 581,657 pairs -- 9,088 returning atomics on one address).  Every thread of the block must call it; `turn` is the
// caller's loop count (the scratch words alternate so that a fast wave cannot overwrite what a slow one still reads).
// Returns the slot of this thread's item (meaningful where `put`).
__device__ inline int block_append_slot(bool put, int32_t *counter, int turn)
{
    __shared__ int s_cnt[2][4], s_base[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t = turn & 1;
    const unsigned long long m = __ballot(put);
    if (lane == 0)
        s_cnt[t][wv] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int total = s_cnt[t][0] + s_cnt[t][1] + s_cnt[t][2] + s_cnt[t][3];
        s_base[t] = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int slot = s_base[t] + __popcll(m & ((1ull << lane) - 1ull));
    for (int v = 0; v < wv; ++v)
        slot += s_cnt[t][v];
    return slot;
}

// The pairs need 1 .. max_iter iterations each (mean ~8, an eighth run all 20): one thread looping to its own
// convergence keeps a wave busy until its slowest lane is done (2.6x the work at 64 lanes).  So the iteration is a
// launch: gn_iter_kernel<it> runs iteration `it` of every pair still active and appends the survivors to the next
// list (wave-aggregated append; the order of a list does not influence any result).  State per pair: alpha, the two
// left-patch means and sin/cos of the left orientation (40 bytes).
__device__ inline void gn_geometry(const GnArgs &A, int64_t k, ebvo_edge &le, double &ex, double &ey)
{
    const int li = A.pair_left[k];
    le = A.L[li];
    ex = -A.lines[(size_t)li * 3 + 1]; // :1331
    ey = A.lines[(size_t)li * 3];
    const double en = sqrt(ex * ex + ey * ey);
    ex /= en;
    ey /= en;
}

__device__ inline void gn_candidate(const GnArgs &A, int64_t k, double &rx, double &ry)
{
    if (A.cand_xy)
    {
        rx = A.cand_xy[2 * k];
        ry = A.cand_xy[2 * k + 1];
    }
    else
    {
        const int c = A.col_idx[k];
        rx = A.R[c].x;
        ry = A.R[c].y;
    }
}

// the left side of every pair depends on the left edge only: sin / cos, the 2 x 49 samples and their means, once per edge
// Eight lanes per left edge (lane r < 7 = patch row r of both sides; lane 7 repeats row 6 and writes nothing): the 98
// samples of an edge are stored by neighbouring lanes as one contiguous run, and a lane walks 14 samples instead of 98 (a
// thread per edge issued 98 scattered 4-byte stores and was one long dependent walk: 64 us at KITTI size).  The mean's
// sum visits the lanes in row order, as in gn_iter_rows_kernel: the same additions in the same order.
__global__ __launch_bounds__(256) void gn_left_kernel(GnArgs A)
{
    const int h = A.h, w = A.w;
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    const int groups_per_block = blockDim.x >> 3;
    for (int base = blockIdx.x * groups_per_block; base < A.nL; base += gridDim.x * groups_per_block)
    {
        const int i = base + (threadIdx.x >> 3);
        const bool live = i < A.nL; // uniform in the group; the exchanges below run for every lane of the wave
        const ebvo_edge le = A.L[live ? i : 0];
        double st, ct;
        ebvo_sincos(le.theta, &st, &ct);
        const double nx = -st, ny = ct;      // n(-t.y, t.x), :1172
        const double side = (7 / 2.0) + 1.0; // :1173
        const int a = min(row, 6) - 3;       // this lane's patch row
#pragma unroll 1
        for (int sd = 0; sd < 2; ++sd)
        {
            const double cx = sd ? le.x - nx * side : le.x + nx * side, cy = sd ? le.y - ny * side : le.y + ny * side;
            float v[7];
#pragma unroll
            for (int b = -3; b <= 3; ++b)
                v[b + 3] = sample_pix4(A.pix4L, w, h, cx + ct * a - st * b, cy + st * a + ct * b);
            if (live && row < 7)
            {
                float *dst = A.left_rec + (size_t)i * 98 + sd * 49 + row * 7;
#pragma unroll
                for (int b = 0; b < 7; ++b)
                    dst[b] = v[b];
            }
            double sum = 0;
#pragma unroll 1
            for (int r = 0; r < 7; ++r)
            {
                double t = sum;
#pragma unroll
                for (int b = 0; b < 7; ++b)
                    t += (double)v[b];
                sum = __shfl(t, gbase | r); // the running sum after row r
            }
            if (live && row == 0)
                A.mean_l[(size_t)sd * A.nL + i] = sum / 49; // :1183-1190
        }
        if (live && row == 0)
        {
            A.sc[i] = st;
            A.sc[A.nL + i] = ct;
        }
    }
}

__global__ __launch_bounds__(256) void gn_init_kernel(GnArgs A)
{
    int turn = 0;
    const int64_t span = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_pairs = devcount(DevCount{A.n_pairs, A.n_pairs_dev});
    for (int64_t k0 = (int64_t)blockIdx.x * blockDim.x; k0 < n_pairs; k0 += span)
    {
        const int64_t k = k0 + threadIdx.x;
        const bool inside = k < n_pairs;
        const bool active = inside && (!A.keep || A.keep[k]);
        // the first active list (its order does not influence any result)
        const int slot = block_append_slot(active, &A.counts[0], turn++);
        if (active)
            A.list[0][slot] = (int32_t)k;
        if (inside && !active)
        {
            double rx, ry;
            gn_candidate(A, k, rx, ry);
            A.alpha[k] = 0.0;
            A.score[k] = __builtin_nan("");
            A.conf[k] = __builtin_nan("");
            A.valid[k] = 255; // not a kept match: not refined
            A.iters[k] = 0;
            A.refined_xy[2 * k] = rx;
            A.refined_xy[2 * k + 1] = ry;
        }
        if (!active)
            continue;
        A.alpha[k] = 0.0;
        A.score[k] = __builtin_nan("");
        A.conf[k] = __builtin_nan("");
        A.valid[k] = 2; // the reference leaves its outputs unset when it stops on H < 1e-8 (:1255)
        A.iters[k] = 0;
    }
}

// Which layout runs an iteration is decided on the device, from the number of pairs still active: one thread per pair
// above GN_ROWS_BELOW active pairs (the launch is throughput-bound there and that layout is the cheaper one per pair),
// eight lanes per pair below it (a small launch of the thread layout lasts ~100 us whatever it holds).  Large problems
// launch BOTH kernels per iteration and one of them returns at once (mode 1 / 2); small ones only the row layout (mode 0).
#ifndef GN_TAP_ROWS
#define GN_TAP_ROWS 2 // patch rows whose records are in flight together (2.79 -> 2.72 ms against 1)
#endif
constexpr int GN_ROWS_BELOW = 65536; // tools/gpu_gn_ab.py (persistent eight-lanes launch below it): KITTI chain refinement 0.98 / 0.99 / 1.01 /
                                     // 1.04 / 1.13 / 1.41 ms at 32768 / 49152 / 65536 / 98304 / 131072 / 262144; a 752 x 480 frame (25 k pairs, at
                                     // most 50 k kept matches) is below it from the start: one launch, 0.51 ms (0.94 as a launch per iteration)
__device__ inline bool gn_other_layout(int mode, int n_active, int rows_below)
{
    return (mode == 1 && n_active <= rows_below) || (mode == 2 && n_active > rows_below);
}

// One Gauss-Newton iteration `it` of pair k in the one-thread-per-pair layout; `alpha` is the iteration state (in and out).
// Returns true when the pair is finished (its outputs are written then).  s_tv: the block's tap staging, [49][256].
__device__ inline bool gn_thread_iteration(const GnArgs &A, int64_t k, int it, double &alpha, float (*s_tv)[256])
{
    const int h = A.h, w = A.w;
    ebvo_edge le;
    double ex, ey;
    gn_geometry(A, k, le, ex, ey);
    const int li = A.pair_left[k];
    const double st = A.sc[li], ct = A.sc[A.nL + li];
    const double nx = -st, ny = ct, side = (7 / 2.0) + 1.0;
    double rx, ry;
    gn_candidate(A, k, rx, ry);
    const double meanL[2] = {A.mean_l[li], A.mean_l[A.nL + li]};
    const double shx = ex * alpha, shy = ey * alpha;
    // One pass over the right image per side: every sample point is tapped ONCE (intensity and both gradients from
    // one 16-byte record) and its three floats stay in registers while the mean of the side forms; the residual
    // terms follow from the registers.  (The reference samples the patch twice, :1204-1247; the values are the same.)
    double H = 0.0, b = 0.0, cost = 0.0;
#pragma unroll 1
    for (int sd = 0; sd < 2; ++sd)
    {
        const float *__restrict__ lrec = A.left_rec + (size_t)li * 98 + sd * 49;
        const double cx = (sd ? rx - nx * side : rx + nx * side) + shx; // :1204-1205
        const double cy = (sd ? ry - ny * side : ry + ny * side) + shy;
        float tgx[49], tgy[49];
        double sum = 0;
#pragma unroll
        for (int i = -3; i <= 3; ++i)
        {
#pragma unroll // the seven 16-byte loads of a patch row are independent: all in flight together
            for (int j = -3; j <= 3; ++j)
            {
                const GnTap t = gn_tap(A.recR, w, h, cx + ct * i - st * j, cy + st * i + ct * j);
                const int o = (i + 3) * 7 + (j + 3);
                s_tv[o][threadIdx.x] = t.v;
                tgx[o] = t.gx;
                tgy[o] = t.gy;
                sum += (double)t.v;
            }
            if ((i + 3) % GN_TAP_ROWS == GN_TAP_ROWS - 1) // GN_TAP_ROWS rows of records in flight, not all forty-nine
                __builtin_amdgcn_sched_barrier(0);
        }
        const double meanR = sum / 49;
#pragma unroll
        for (int o = 0; o < 49; ++o)
        {
            if (o % 7 == 0)
                __builtin_amdgcn_sched_barrier(0);
            const double Lf = (double)lrec[o]; // sampled once by gn_left_kernel
            const double Rf = (double)s_tv[o][threadIdx.x], gxv = (double)tgx[o], gyv = (double)tgy[o];
            const double r = (Lf - meanL[sd]) - (Rf - meanR);
            const double g = -gxv * ex + gyv * ey; // :1237
            const double absr = fabs(r);
            const double wgt = (absr <= A.huber) ? 1.0 : A.huber / absr;
            H += wgt * g * g;
            b += wgt * g * r;
            cost += wgt * r * r;
        }
    }
    int done_iters = it; // stop on H < 1e-8: outputs stay unset (:1255)
    bool finished = true;
    if (!(H < 1e-8))
    {
        const double delta = -b / H;
        alpha += delta;
        const double rms = sqrt(cost / 98);
        const bool is_outlier = (rms > A.huber * 2.0) || (it + 1 < 2); // residual_log.size() == it + 1
        if (fabs(delta) < A.tol || it == A.max_iter - 1)
        {
            A.valid[k] = is_outlier ? 0 : 1;
            A.score[k] = rms;
            A.conf[k] = ebvo_exp(-rms / A.huber);
            done_iters = it + 1;
        }
        else
            finished = false;
    }
    A.alpha[k] = alpha;
    if (finished)
    {
        A.iters[k] = done_iters;
        A.refined_xy[2 * k] = rx + ex * alpha; // :1349-1351
        A.refined_xy[2 * k + 1] = ry + ey * alpha;
    }
    return finished;
}

__global__ __launch_bounds__(256, 2) void gn_iter_kernel(GnArgs A, int it, int mode)
{
    const int n_in = A.counts[it];
    if (gn_other_layout(mode, n_in, A.rows_below))
        return;
    const int32_t *__restrict__ lin = A.list[it & 1];
    int32_t *__restrict__ lout = A.list[(it + 1) & 1];
    __shared__ float s_tv[49][256]; // the intensity taps of one side, private to each thread (its gradient taps stay in registers)
    int turn = 0;
    for (int base = blockIdx.x * blockDim.x; base < n_in; base += gridDim.x * blockDim.x)
    {
        const int idx = base + threadIdx.x;
        bool survives = false;
        int64_t k = 0;
        if (idx < n_in)
        {
            k = lin[idx];
            double alpha = A.alpha[k];
            survives = !gn_thread_iteration(A, k, it, alpha, s_tv);
        }
        // append the survivors: one atomic per block
        const int slot = block_append_slot(survives, &A.counts[it + 1], turn++);
        if (survives)
            lout[slot] = (int32_t)k;
    }
}

// The same iteration with EIGHT LANES PER PAIR (lane r < 7 = patch row r of both sides, lane 7 idle), for SMALL problems.
// A launch of the one-thread-per-pair kernel lasts at least ~90-110 us however few pairs it holds (one wave per SIMD
// issues an instruction only every ~10 cycles and a thread walks 196 samples); with hundreds of thousands of pairs that
// floor is irrelevant (KITTI: 314 us for 472,947 pairs, and this layout costs 2x there), with the 50-100 k pairs of a
// 752 x 480 frame every one of the 20 launches sits on it.  Here a lane samples 7 + 7 points per side: eight times the
// waves, an eighth of the chain.  The reference's sums are sequential over the 49 (98) samples and stay so: the running
// sum visits the lanes in row order (lane r adds its seven terms to what lane r - 1 produced).  Bit-identical to
// gn_iter_kernel (the refinement tests run both).
#ifndef GN_ROWS_WAVES
#define GN_ROWS_WAVES 3 // waves per SIMD the row layout is compiled for (166 VGPRs; 4 -> 128 with spills, measured slower)
#endif
// lane i receives lane i - 1's value (rows of 16 lanes; the first lane of a row receives 0): two DPP register moves
__device__ inline double row_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// what an eight-lane group keeps of its pair between iterations (the persistent kernel below loads it once per pair)
struct GnRowsState
{
    double ex, ey, st, ct, rx, ry, meanL0, meanL1, alpha;
    int li;
};
__device__ inline void gn_rows_load(const GnArgs &A, int64_t k, GnRowsState &S)
{
    ebvo_edge le;
    gn_geometry(A, k, le, S.ex, S.ey);
    S.li = A.pair_left[k];
    S.st = A.sc[S.li];
    S.ct = A.sc[A.nL + S.li];
    gn_candidate(A, k, S.rx, S.ry);
    S.meanL0 = A.mean_l[S.li];
    S.meanL1 = A.mean_l[A.nL + S.li];
    S.alpha = A.alpha[k];
}
// One iteration `it` of pair k by the eight lanes of a group (every lane of the wave runs it: the exchanges are wave-wide;
// `live` false = the group holds no pair, nothing is written).  Returns, on the group's lane 0, whether the pair is finished
// (its outputs are written then); S.alpha advances on that lane.
__device__ inline bool gn_rows_iteration(const GnArgs &A, int64_t k, GnRowsState &S, int it, bool live, int row, int gbase)
{
    const int h = A.h, w = A.w;
    const double ex = S.ex, ey = S.ey, st = S.st, ct = S.ct, rx = S.rx, ry = S.ry;
    const int li = S.li;
    const double nx = -st, ny = ct, side = (7 / 2.0) + 1.0;
    const double meanL[2] = {S.meanL0, S.meanL1};
    double alpha = S.alpha;
    const double shx = ex * alpha, shy = ey * alpha;
    const int i = min(row, 6) - 3; // this lane's patch row (lane 7 repeats row 6 and is never selected)
    double H = 0.0, b = 0.0, cost = 0.0;
    // this lane's row of BOTH sides, tapped once (intensity and both gradients): the fourteen record loads are in flight
    // together, the second side's arrive while the first side's sums form
    GnTap tps[2][7];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
    {
        const double cx = (sd ? rx - nx * side : rx + nx * side) + shx; // :1204-1205
        const double cy = (sd ? ry - ny * side : ry + ny * side) + shy;
#pragma unroll
        for (int j = -3; j <= 3; ++j)
            tps[sd][j + 3] = gn_tap(A.recR, w, h, cx + ct * i - st * j, cy + st * i + ct * j);
    }
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
    {
        const float *__restrict__ lrec = A.left_rec + (size_t)li * 98 + sd * 49 + (i + 3) * 7;
        const GnTap *tp = tps[sd];
        // The reference's sum runs over the 49 samples in order, i.e. through the lanes in row order.  In round r lane r adds
        // its seven terms to what lane r - 1 handed it in round r - 1 (a DPP shift by one lane: a register move, where a
        // ds_bpermute per round put 28 LDS round trips per side on the critical path); what the other lanes compute in that
        // round is discarded.  After round 6 lane 6 holds the sum.
        double run = 0.0, t = 0.0;
#pragma unroll 1
        for (int r = 0; r < 7; ++r)
        {
            t = run;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                t += (double)tp[j].v;
            run = row_shr1(t);
        }
        const double sum = __shfl(t, gbase | 6);
        const double meanR = sum / 49;
        double tH[7], tb[7], tc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j)
        {
            const double Lf = (double)lrec[j]; // sampled once by gn_left_kernel
            const double Rf = (double)tp[j].v, gxv = (double)tp[j].gx, gyv = (double)tp[j].gy;
            const double r = (Lf - meanL[sd]) - (Rf - meanR);
            const double g = -gxv * ex + gyv * ey; // :1237
            const double absr = fabs(r);
            const double wgt = (absr <= A.huber) ? 1.0 : A.huber / absr;
            tH[j] = wgt * g * g; // the addends of gn_iter_kernel's three sums, formed by the same operations
            tb[j] = wgt * g * r;
            tc[j] = wgt * r * r;
        }
        // (H, b, cost enter the side on lane 0: every lane holds the totals of the previous side)
        double rH = H, rb = b, rc = cost, uH = 0.0, ub = 0.0, uc = 0.0;
#pragma unroll 1
        for (int r = 0; r < 7; ++r)
        {
            uH = rH, ub = rb, uc = rc;
#pragma unroll
            for (int j = 0; j < 7; ++j)
            {
                uH += tH[j];
                ub += tb[j];
                uc += tc[j];
            }
            rH = row_shr1(uH);
            rb = row_shr1(ub);
            rc = row_shr1(uc);
        }
        H = __shfl(uH, gbase | 6);
        b = __shfl(ub, gbase | 6);
        cost = __shfl(uc, gbase | 6);
    }
    bool finished = true;
    if (live && row == 0)
    {
        int done_iters = it; // stop on H < 1e-8: outputs stay unset (:1255)
        if (!(H < 1e-8))
        {
            const double delta = -b / H;
            alpha += delta;
            const double rms = sqrt(cost / 98);
            const bool is_outlier = (rms > A.huber * 2.0) || (it + 1 < 2); // residual_log.size() == it + 1
            if (fabs(delta) < A.tol || it == A.max_iter - 1)
            {
                A.valid[k] = is_outlier ? 0 : 1;
                A.score[k] = rms;
                A.conf[k] = ebvo_exp(-rms / A.huber);
                done_iters = it + 1;
            }
            else
                finished = false;
        }
        A.alpha[k] = alpha;
        S.alpha = alpha;
        if (finished)
        {
            A.iters[k] = done_iters;
            A.refined_xy[2 * k] = rx + ex * alpha; // :1349-1351
            A.refined_xy[2 * k + 1] = ry + ey * alpha;
        }
    }
    return finished;
}

__global__ __launch_bounds__(256, 2) void gn_iter_rows_kernel(GnArgs A, int it, int mode) // (two waves per SIMD: both sides' taps are live)
{
    const int n_in = A.counts[it];
    if (gn_other_layout(mode, n_in, A.rows_below))
        return;
    const int32_t *__restrict__ lin = A.list[it & 1];
    int32_t *__restrict__ lout = A.list[(it + 1) & 1];
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    const int groups_per_block = blockDim.x >> 3;
    for (int base = blockIdx.x * groups_per_block; base < n_in; base += gridDim.x * groups_per_block)
    {
        const int idx = base + (threadIdx.x >> 3);
        const bool live = idx < n_in; // uniform in the group; the cross-lane exchanges below run for every lane of the wave
        const int64_t k = live ? lin[idx] : 0;
        GnRowsState S;
        gn_rows_load(A, k, S);
        const bool finished = gn_rows_iteration(A, k, S, it, live, row, gbase);
        const bool survives = live && row == 0 && !finished;
        // append the survivors: one atomic per wave
        const unsigned long long m = __ballot(survives);
        int wbase = 0;
        if (lane == 0 && m)
            wbase = atomicAdd(&A.counts[it + 1], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (survives)
            lout[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)k;
    }
}

// The remaining iterations of every pair still active in ONE launch, eight lanes per pair: a group draws a pair from the
// list, runs its iterations back to back (the pair's geometry, left-edge record address and means stay in registers: a
// launch per iteration reloads them twenty times through a chain of dependent loads) and draws the next one when the pair
// has converged -- no launch between two iterations of a pair, no list to compact, and a wave is never idle while the list
// holds work.  The groups of a wave are at different iterations of different pairs; the iteration body does not depend on
// the iteration number except through two wave-uniform-per-group scalars, so they run in lock step without divergence.
// Which list: the last one the per-iteration launches of the one-thread-per-pair layout left non-empty (they stop as soon
// as at most rows_below pairs are active, see gn_iter_kernel), or list 0 when no such launch was made.  Same arithmetic per
// pair and iteration as gn_iter_rows_kernel (one device function), hence the same bits.
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void gn_rows_persistent_kernel(GnArgs A)
{
    int it0 = -1;
    for (int it = A.max_iter - 1; it >= 0; --it) // wave-uniform: the last list that holds pairs
        if (A.counts[it] > 0)
        {
            it0 = it;
            break;
        }
    if (it0 < 0)
        return;
    const int n_in = A.counts[it0];
    if (n_in > A.rows_below) // the one-thread-per-pair launches ran this list (and it was the last iteration, or nothing survived)
        return;
    const int32_t *__restrict__ lin = A.list[it0 & 1];
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    bool have = false;
    int it = it0;
    int64_t k = lin[0]; // a group without a pair computes on pair lin[0] and writes nothing
    GnRowsState S;
    gn_rows_load(A, k, S);
    // Every 8-lane group walks its own arithmetic sequence of list entries (group g: g, g + G, g + 2 G, ...).  Round 3 drew
    // the entries from a shared cursor, one returning atomic per wave and loop turn: with one wave per SIMD its round trip
    // (and the serialisation of thousands of same-address atomics across the XCDs) sat in every turn of the loop.
    const int n_groups = (int)gridDim.x * 32;
    int my_next = ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 8 + (lane >> 3);
    for (;;)
    {
        if (!have)
        {
            const int idx = my_next;
            if (idx < n_in)
            {
                my_next += n_groups;
                k = lin[idx];
                gn_rows_load(A, k, S);
                it = it0;
                have = true;
            }
        }
        if (!__ballot(have))
            break; // the list is drained and every pair of this wave has converged
        const bool finished = gn_rows_iteration(A, k, S, it, have, row, gbase);
        const bool fin_g = __shfl((int)finished, gbase) != 0; // the group's verdict (formed on its lane 0)
        S.alpha = __shfl(S.alpha, gbase);
        if (have && fin_g)
            have = false;
        ++it;
    }
}

constexpr int64_t GN_ROWS_MAX_PAIRS = 131072; // temporal batches of up to this many items run eight lanes per item

// ------------------------------------------------------------------------------------------
// Temporal 2-D refinement: Temporal_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton,
// src/Temporal_Matches.cpp:735-851.  Same launch-per-iteration scheme; differences from the stereo variant: the
// current-frame patches use the current-frame edge's own orientation, the Huber test is strict, H is 2x2 with the 1e-6
// regulariser added once per sample (:809), and the update solves H x = b with Eigen's pivoted LDL^T.
struct Gn2Args
{
    const uint8_t *imgK, *imgC;
    const float2 *gxy; // (unused since the packed records: interleaved Sobel planes of the current-frame image)
    // items n_first .. n - 1 belong to a second camera (the right images of the quads): both cameras of a batch of quads
    // run in the same launches (a launch of a few thousand items lasts ~100 us whatever their number)
    const uint8_t *imgK2, *imgC2;
    const float2 *gxy2;
    int64_t n_first;
    int h, w;
    const ebvo_edge *kf, *cf;
    const double *init; // n x 2
    int64_t n;
    int max_iter;
    double tol, huber;
    double *disp, *score; // disp (n x 2) doubles as the iteration state
    uint8_t *valid;
    int32_t *iters;
    double *mean_l; // [2][n]
    double *sc;     // [4][n]: sin, cos of the keyframe edge, sin, cos of the current-frame edge
    float *lrec;    // [n][98] the keyframe-side samples of every item (they do not change with the iterations): gn2_init_kernel
    const uint4 *recC, *recC2; // packed corner records (intensity + Sobel gradients, gn_pack_kernel) of the current-frame images
    int32_t *list[2];
    int32_t *counts;
};

// Eigen 3.4.0 H.ldlt().solve(b) for a 2x2 double H (Eigen/src/Cholesky/LDLT.h, restated; see oracle/ebvo_oracle.c)
__device__ inline void ldlt2_solve(double h00, double h10, double h11, double b0, double b1, double &x0, double &x1)
{
    bool swap = fabs(h11) > fabs(h00);
    if (swap)
    {
        const double t = h00;
        h00 = h11;
        h11 = t;
    }
    double l10 = h10;
    if (!(fabs(h00) > 0.0))
        swap = false;
    else
    {
        l10 = h10 / h00;
        const double temp = h00 * l10;
        h11 -= l10 * temp;
    }
    double y0 = swap ? b1 : b0, y1 = swap ? b0 : b1;
    y1 -= l10 * y0;
    const double tiny = 2.2250738585072014e-308;
    y0 = (fabs(h00) > tiny) ? y0 / h00 : 0.0;
    y1 = (fabs(h11) > tiny) ? y1 / h11 : 0.0;
    y0 -= l10 * y1;
    x0 = swap ? y1 : y0;
    x1 = swap ? y0 : y1;
}

// (eight lanes per item, as gn_left_kernel: the keyframe-side samples leave as one contiguous run per item)
__global__ __launch_bounds__(256) void gn2_init_kernel(Gn2Args A)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        A.counts[0] = (int32_t)A.n;
    const int h = A.h, w = A.w;
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    const int64_t groups_per_block = blockDim.x >> 3;
    for (int64_t base = (int64_t)blockIdx.x * groups_per_block; base < A.n; base += (int64_t)gridDim.x * groups_per_block)
    {
        const int64_t k = base + (threadIdx.x >> 3);
        const bool live = k < A.n; // uniform in the group
        const int64_t kk = live ? k : 0;
        const ebvo_edge ke = A.kf[kk];
        const bool second = kk >= A.n_first;
        const uint8_t *__restrict__ imgK = second ? A.imgK2 : A.imgK;
        double st, ct, stc, ctc;
        ebvo_sincos(ke.theta, &st, &ct);
        ebvo_sincos(A.cf[kk].theta, &stc, &ctc);
        const double nx = -st, ny = ct, side = (7 / 2.0) + 1.0;
        const int i = min(row, 6) - 3; // this lane's patch row (lane 7 repeats row 6 and writes nothing)
#pragma unroll 1
        for (int sd = 0; sd < 2; ++sd)
        {
            const double cx = sd ? ke.x - nx * side : ke.x + nx * side, cy = sd ? ke.y - ny * side : ke.y + ny * side;
            float v[7];
#pragma unroll
            for (int j = -3; j <= 3; ++j)
                v[j + 3] = sample_u8(imgK, w, w, h, cx + ct * i - st * j, cy + st * i + ct * j);
            if (live && row < 7)
            {
                float *dst = A.lrec + (size_t)k * 98 + sd * 49 + row * 7; // read back by every iteration
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    dst[j] = v[j];
            }
            double sum = 0;
#pragma unroll 1
            for (int r = 0; r < 7; ++r)
            {
                double t = sum;
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    t += (double)v[j];
                sum = __shfl(t, gbase | r);
            }
            if (live && row == 0)
                A.mean_l[(size_t)sd * A.n + k] = sum / 49;
        }
        if (live && row == 0)
        {
            A.sc[k] = st;
            A.sc[A.n + k] = ct;
            A.sc[2 * A.n + k] = stc;
            A.sc[3 * A.n + k] = ctc;
            A.disp[2 * k] = A.init[2 * k];
            A.disp[2 * k + 1] = A.init[2 * k + 1];
            A.score[k] = __builtin_nan("");
            A.valid[k] = 2;
            A.iters[k] = 0;
            A.list[0][k] = (int32_t)k;
        }
    }
}

__global__ __launch_bounds__(256) void gn2_iter_kernel(Gn2Args A, int it)
{
    const int h = A.h, w = A.w;
    const int n_in = A.counts[it];
    const int32_t *__restrict__ lin = A.list[it & 1];
    int32_t *__restrict__ lout = A.list[(it + 1) & 1];
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * blockDim.x; base < n_in; base += gridDim.x * blockDim.x)
    {
        const int idx = base + threadIdx.x;
        bool survives = false;
        int64_t k = 0;
        if (idx < n_in)
        {
            k = lin[idx];
            const ebvo_edge ke = A.kf[k];
            const bool second = k >= A.n_first;
            const uint8_t *__restrict__ imgC = second ? A.imgC2 : A.imgC;
            const uint4 *__restrict__ recC = second ? A.recC2 : A.recC;
            const double stc = A.sc[2 * A.n + k], ctc = A.sc[3 * A.n + k];
            const double ncx = -stc, ncy = ctc, side = (7 / 2.0) + 1.0;
            const double meanL[2] = {A.mean_l[k], A.mean_l[A.n + k]};
            double d0 = A.disp[2 * k], d1 = A.disp[2 * k + 1];
            const double lx = ke.x - d0, ly = ke.y - d1; // :786
            double meanR[2];
#pragma unroll 1
            for (int sd = 0; sd < 2; ++sd)
            {
                const double cx = sd ? lx - ncx * side : lx + ncx * side, cy = sd ? ly - ncy * side : ly + ncy * side;
                double sum = 0;
#pragma unroll 1
                for (int i = -3; i <= 3; ++i)
#pragma unroll
                    for (int j = -3; j <= 3; ++j)
                        sum += (double)sample_u8(imgC, w, w, h, cx + ctc * i - stc * j, cy + stc * i + ctc * j);
                meanR[sd] = sum / 49;
            }
            double H00 = 0, H10 = 0, H11 = 0, b0 = 0, b1 = 0, cost = 0;
#pragma unroll 1
            for (int sd = 0; sd < 2; ++sd)
            {
                const float *__restrict__ lrec = A.lrec + (size_t)k * 98 + sd * 49;
                const double cx = sd ? lx - ncx * side : lx + ncx * side, cy = sd ? ly - ncy * side : ly + ncy * side;
#pragma unroll 1
                for (int i = -3; i <= 3; ++i)
#pragma unroll 1
                    for (int j = -3; j <= 3; ++j)
                    {
                        const double Lf = (double)lrec[(i + 3) * 7 + (j + 3)]; // sampled once by gn2_init_kernel
                        const GnTap tp = gn_tap(recC, w, h, cx + ctc * i - stc * j, cy + stc * i + ctc * j);
                        const double Rf = (double)tp.v, J0 = (double)tp.gx, J1 = (double)tp.gy;
                        const double r = (Lf - meanL[sd]) - (Rf - meanR[sd]);
                        const double absr = fabs(r);
                        const double wgt = (absr < A.huber) ? 1.0 : A.huber / absr; // strict, :806
                        const double wJ0 = wgt * J0, wJ1 = wgt * J1;
                        H00 += wJ0 * J0; // H += w * J * J^T, coefficient (i, j) = (w J(i)) J(j)
                        H10 += wJ1 * J0;
                        H11 += wJ1 * J1;
                        H00 += 1e-6;     // H += 1e-6 * I (:809)
                        H10 += 0.0;
                        H11 += 1e-6;
                        b0 += wJ0 * r;
                        b1 += wJ1 * r;
                        cost += wgt * r * r;
                    }
            }
            double s0, s1;
            ldlt2_solve(H00, H10, H11, b0, b1, s0, s1);
            const double delta0 = -s0, delta1 = -s1;
            d0 += delta0;
            d1 += delta1;
            const double rms = sqrt(cost / 98);
            const bool is_outlier = (rms > A.huber * 2.0) || (it + 1 < 2);
            const bool finished = sqrt(delta0 * delta0 + delta1 * delta1) < A.tol || it == A.max_iter - 1;
            A.disp[2 * k] = d0;
            A.disp[2 * k + 1] = d1;
            if (finished)
            {
                A.valid[k] = is_outlier ? 0 : 1;
                A.score[k] = rms;
                A.iters[k] = it + 1;
            }
            survives = !finished;
        }
        const unsigned long long m = __ballot(survives);
        int wbase = 0;
        if (lane == 0 && m)
            wbase = atomicAdd(&A.counts[it + 1], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (survives)
            lout[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)k;
    }
}

// Eight lanes per item for small batches, as gn_iter_rows_kernel: lane r < 7 = patch row r of both sides; the nine
// running sums (H gets its 1e-6 I after every sample, :809) visit the lanes in row order.  Bit-identical to gn2_iter_kernel.
// One Gauss-Newton iteration of item k on the eight lanes of a group (lane `row` = patch row; `live` and k uniform in the
// group).  (d0, d1) is the item's state, held by the caller; lane 0 of the group updates it, stores it, records the verdict
// when the item has finished, and returns whether it goes on (the other lanes return false: the caller broadcasts).
__device__ __forceinline__ bool gn2_rows_iteration(const Gn2Args &A, int64_t k, bool live, int it, int row, int gbase, double &d0,
                                                   double &d1)
{
    const int h = A.h, w = A.w;
    bool survives = false;
    const ebvo_edge ke = A.kf[k];
    const bool second = k >= A.n_first;
    const uint4 *__restrict__ recC = second ? A.recC2 : A.recC;
    const double stc = A.sc[2 * A.n + k], ctc = A.sc[3 * A.n + k];
    const double ncx = -stc, ncy = ctc, side = (7 / 2.0) + 1.0;
    const double meanL[2] = {A.mean_l[k], A.mean_l[A.n + k]};
    const double lx = ke.x - d0, ly = ke.y - d1; // :786
    const int i = min(row, 6) - 3; // this lane's patch row (lane 7 repeats row 6 and is never selected)
    double H00 = 0, H10 = 0, H11 = 0, b0 = 0, b1 = 0, cost = 0;
#pragma unroll 1
    for (int sd = 0; sd < 2; ++sd)
    {
        const float *__restrict__ lrec = A.lrec + (size_t)k * 98 + sd * 49 + (i + 3) * 7;
        const double cx = sd ? lx - ncx * side : lx + ncx * side, cy = sd ? ly - ncy * side : ly + ncy * side;
        GnTap tp[7]; // this lane's row of the side: every point tapped once (intensity and both gradients)
#pragma unroll
        for (int j = -3; j <= 3; ++j)
            tp[j + 3] = gn_tap(recC, w, h, cx + ctc * i - stc * j, cy + stc * i + ctc * j);
        double sum = 0;
#pragma unroll 1
        for (int r = 0; r < 7; ++r)
        {
            double t = sum;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                t += (double)tp[j].v;
            sum = __shfl(t, gbase | r);
        }
        const double meanR = sum / 49;
        double t00[7], t10[7], t11[7], tb0[7], tb1[7], tc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j)
        {
            const double Lf = (double)lrec[j]; // sampled once by gn2_init_kernel
            const double Rf = (double)tp[j].v, J0 = (double)tp[j].gx, J1 = (double)tp[j].gy;
            const double r = (Lf - meanL[sd]) - (Rf - meanR);
            const double absr = fabs(r);
            const double wgt = (absr < A.huber) ? 1.0 : A.huber / absr; // strict, :806
            const double wJ0 = wgt * J0, wJ1 = wgt * J1;
            t00[j] = wJ0 * J0; // the addends of gn2_iter_kernel's sums, formed by the same operations
            t10[j] = wJ1 * J0;
            t11[j] = wJ1 * J1;
            tb0[j] = wJ0 * r;
            tb1[j] = wJ1 * r;
            tc[j] = wgt * r * r;
        }
#pragma unroll 1
        for (int r = 0; r < 7; ++r)
        {
            double u00 = H00, u10 = H10, u11 = H11, ub0 = b0, ub1 = b1, uc = cost;
#pragma unroll
            for (int j = 0; j < 7; ++j)
            {
                u00 += t00[j];
                u10 += t10[j];
                u11 += t11[j];
                u00 += 1e-6; // H += 1e-6 * I (:809)
                u10 += 0.0;
                u11 += 1e-6;
                ub0 += tb0[j];
                ub1 += tb1[j];
                uc += tc[j];
            }
            H00 = __shfl(u00, gbase | r);
            H10 = __shfl(u10, gbase | r);
            H11 = __shfl(u11, gbase | r);
            b0 = __shfl(ub0, gbase | r);
            b1 = __shfl(ub1, gbase | r);
            cost = __shfl(uc, gbase | r);
        }
    }
    if (live && row == 0)
    {
        double s0, s1;
        ldlt2_solve(H00, H10, H11, b0, b1, s0, s1);
        const double delta0 = -s0, delta1 = -s1;
        d0 += delta0;
        d1 += delta1;
        const double rms = sqrt(cost / 98);
        const bool is_outlier = (rms > A.huber * 2.0) || (it + 1 < 2);
        const bool finished = sqrt(delta0 * delta0 + delta1 * delta1) < A.tol || it == A.max_iter - 1;
        A.disp[2 * k] = d0;
        A.disp[2 * k + 1] = d1;
        if (finished)
        {
            A.valid[k] = is_outlier ? 0 : 1;
            A.score[k] = rms;
            A.iters[k] = it + 1;
        }
        survives = !finished;
    }
    return survives;
}

__global__ __launch_bounds__(256, 2) void gn2_iter_rows_kernel(Gn2Args A, int it)
{
    const int n_in = A.counts[it];
    const int32_t *__restrict__ lin = A.list[it & 1];
    int32_t *__restrict__ lout = A.list[(it + 1) & 1];
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    const int groups_per_block = blockDim.x >> 3;
    for (int base = blockIdx.x * groups_per_block; base < n_in; base += gridDim.x * groups_per_block)
    {
        const int idx = base + (threadIdx.x >> 3);
        const bool live = idx < n_in; // uniform in the group
        const int64_t k = live ? lin[idx] : 0;
        double d0 = A.disp[2 * k], d1 = A.disp[2 * k + 1];
        const bool survives = gn2_rows_iteration(A, k, live, it, row, gbase, d0, d1);
        const unsigned long long m = __ballot(survives);
        int wbase = 0;
        if (lane == 0 && m)
            wbase = atomicAdd(&A.counts[it + 1], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (survives)
            lout[wbase + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)k;
    }
}

// The same iterations as ONE launch (round 4): every 8-lane group walks its own sequence of items (group g: g, g + G, ...)
// and runs each to its convergence, its state in registers; no lists, no counters, no atomics (a launch per iteration had
// been ~20 launches of ~18 us for a few thousand items, each appending its survivors to a list through an atomic).
__global__ __launch_bounds__(256, 2) void gn2_rows_persistent_kernel(Gn2Args A)
{
    const int lane = threadIdx.x & 63, row = lane & 7, gbase = lane & ~7;
    const int64_t n = A.n, n_groups = (int64_t)gridDim.x * 32;
    int64_t my_next = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3);
    bool have = false;
    int64_t k = 0;
    int it = 0;
    double d0 = 0, d1 = 0;
    for (;;)
    {
        if (!have && my_next < n)
        {
            k = my_next;
            my_next += n_groups;
            d0 = A.disp[2 * k];
            d1 = A.disp[2 * k + 1];
            it = 0;
            have = true;
        }
        if (!__ballot(have))
            break;
        const bool goes_on = gn2_rows_iteration(A, have ? k : 0, have, it, row, gbase, d0, d1);
        const bool on_g = __shfl((int)goes_on, gbase) != 0; // the group's verdict (formed on its lane 0)
        d0 = __shfl(d0, gbase);
        d1 = __shfl(d1, gbase);
        if (have && !on_g)
            have = false;
        ++it;
    }
}

// ------------------------------------------------------------------------------------------
// Finalisation geometry: the 16 numbers write_finalized_stereo_edge_pairs_to_file prints per final pair
// (src/Stereo_Matches.cpp:1656-1699; src/utility.cpp:95-119).  One thread per pair; the calibration inverses are
// formed once on the host (Eigen's cofactor inverse, restated), every product / cross / normalize in Eigen's
// fixed-size order.
struct FinalCalib
{
    double Kli[9], Kri[9], R21[9], T21[3];
};

__device__ inline void mv3(const double *m, const double *v, double *o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o[i] = (m[i * 3] * v[0] + m[i * 3 + 1] * v[1]) + m[i * 3 + 2] * v[2];
}
__device__ inline void mtv3(const double *m, const double *v, double *o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o[i] = (m[i] * v[0] + m[3 + i] * v[1]) + m[6 + i] * v[2];
}
__device__ inline void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline void normalize3(double *v)
{
    const double z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (z > 0)
    {
        const double n = sqrt(z);
        v[0] /= n;
        v[1] /= n;
        v[2] /= n;
    }
}

__global__ void finalize_pairs_kernel(FinalCalib C, const ebvo_edge *__restrict__ L, const ebvo_edge *__restrict__ R, DevCount nd,
                                      double *__restrict__ out)
{
    const int n = (int)devcount(nd);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
    {
        const ebvo_edge l = L[k], r = R[k];
        const double el[3] = {l.x, l.y, 1.0}, er[3] = {r.x, r.y, 1.0};
        double g1[3], g2[3], Rg1[3];
        mv3(C.Kli, el, g1);
        mv3(C.Kri, er, g2);
        mv3(C.R21, g1, Rg1);
        const double numerator = C.T21[0] - C.T21[2] * g2[0]; // e1.dot(T) - e3.dot(T) * e1.dot(ray2)
        const double denominator = Rg1[2] * g2[0] - Rg1[0];
        const double rho1 = numerator / denominator;
        double sl, cl, sr, cr;
        ebvo_sincos(l.theta, &sl, &cl);
        ebvo_sincos(r.theta, &sr, &cr);
        const double t1r[3] = {cl, sl, 0.0}, t2r[3] = {cr, sr, 0.0};
        double t1[3], t2[3], n1[3], c2[3], n2[3], T[3], p1[3], p2[3];
        mv3(C.Kli, t1r, t1);
        mv3(C.Kri, t2r, t2);
        cross3(t1, g1, n1);
        cross3(t2, g2, c2);
        mtv3(C.R21, c2, n2);
        cross3(n1, n2, T);
        normalize3(T);
#pragma unroll
        for (int i = 0; i < 3; ++i)
        {
            p1[i] = T[i] - T[2] * g1[i];
            p2[i] = T[i] - T[2] * g2[i];
        }
        normalize3(p1);
        normalize3(p2);
        double *o = out + (size_t)k * 16;
        o[0] = l.x; o[1] = l.y; o[2] = l.theta;
        o[3] = r.x; o[4] = r.y; o[5] = r.theta;
        o[6] = rho1 * g1[0]; o[7] = rho1 * g1[1]; o[8] = rho1 * g1[2];
        o[9] = T[0]; o[10] = T[1]; o[11] = T[2];
        o[12] = p1[0]; o[13] = p1[1];
        o[14] = p2[0]; o[15] = p2[1];
    }
}

} // namespace


// cv::undistort of one resident image: d_src (pitch) -> d_dst (dpitch); d_xs: scratch of w doubles
int refine_undistort_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_src, int pitch, int h, int w, const double K[4],
                             const double *dist, int n_dist, double *d_xs, uint8_t *d_dst, int dpitch)
{
    UndistortArgs A;
    A.fx = K[0];
    A.fy = K[1];
    A.u0 = K[2];
    A.v0 = K[3];
    A.k1 = n_dist > 0 ? dist[0] : 0.0;
    A.k2 = n_dist > 1 ? dist[1] : 0.0;
    A.p1 = n_dist > 2 ? dist[2] : 0.0;
    A.p2 = n_dist > 3 ? dist[3] : 0.0;
    A.k3 = n_dist > 4 ? dist[4] : 0.0;
    A.h = h;
    A.w = w;
    int ss0 = (1 << 12) / (w > 1 ? w : 1);
    ss0 = ss0 < 1 ? 1 : (ss0 > h ? h : ss0);
    A.ss0 = ss0;
    ProfScope ps(ctx, s, K_SOBEL);
    hipLaunchKernelGGL(undistort_row_kernel, dim3(1), dim3(64), 0, s.stream, A, d_xs);
    hipLaunchKernelGGL(undistort_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s.stream, d_src, pitch, A,
                       (const double *)d_xs, d_dst, dpitch);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int refine_sobel_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch, float *d_gx, float *d_gy,
                         void *d_gxy)
{
    ProfScope ps(ctx, s, K_SOBEL);
    hipLaunchKernelGGL(sobel_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s.stream, d_img, h, w, pitch, d_gx,
                       d_gy, (float2 *)d_gxy);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int refine_gn_stereo_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgL, const uint8_t *d_imgR, const void *d_gxy,
                             int h, int w, const ebvo_edge *d_L, int nL, const double *d_lines,
                             const int32_t *d_pair_left, const double *d_cand_xy, const ebvo_edge *d_R,
                             const int32_t *d_col_idx, const uint8_t *d_keep, int64_t n_pairs, int max_iter,
                             double tol, double huber, double *d_alpha, double *d_score, double *d_conf,
                             uint8_t *d_valid, int32_t *d_iters, double *d_refined_xy, const int32_t *d_n_pairs)
{
    if (n_pairs <= 0)
        return EBVO_OK;
    if (n_pairs > 0x7fffffffll)
        return EBVO_ERR_CAPACITY;
    if (w < 2 || h < 1)
        return EBVO_ERR_ARG;
    int rc;
    const size_t np = (size_t)n_pairs;
    const size_t nl = (size_t)nL;
    if (nL <= 0)
        return EBVO_ERR_ARG;
    const size_t npx = (size_t)h * w;
    if ((rc = ebvo_grow(ctx, s, s.gn_state, sizeof(double) * 4 * nl + sizeof(float) * 98 * nl)) ||
        (rc = ebvo_grow(ctx, s, s.gn_lists, sizeof(int32_t) * (2 * np + (size_t)max_iter + 2))) ||
        (rc = ebvo_grow(ctx, s, s.gn_pack, (sizeof(uint4) + 2 * sizeof(uint32_t)) * npx)))
        return rc;
    GnArgs A{};
    A.nL = nL;
    A.imgL = d_imgL;
    A.imgR = d_imgR;
    A.gxy = (const float2 *)d_gxy;
    uint4 *recR = (uint4 *)s.gn_pack.p;
    uint32_t *pix4R = (uint32_t *)(recR + npx), *pix4L = pix4R + npx;
    A.recR = recR;
    A.pix4R = pix4R;
    A.pix4L = pix4L;
    A.h = h;
    A.w = w;
    A.L = d_L;
    A.lines = d_lines;
    A.pair_left = d_pair_left;
    A.cand_xy = d_cand_xy;
    A.R = d_R;
    A.col_idx = d_col_idx;
    A.keep = d_keep;
    A.n_pairs = n_pairs;
    A.n_pairs_dev = d_n_pairs;
    A.max_iter = max_iter;
    A.tol = tol;
    A.huber = huber;
    A.alpha = d_alpha;
    A.score = d_score;
    A.conf = d_conf;
    A.refined_xy = d_refined_xy;
    A.valid = d_valid;
    A.iters = d_iters;
    A.mean_l = (double *)s.gn_state.p;
    A.sc = A.mean_l + 2 * nl;
    A.left_rec = (float *)(A.sc + 2 * nl);
    A.list[0] = (int32_t *)s.gn_lists.p;
    A.list[1] = A.list[0] + np;
    A.counts = A.list[1] + np;
    ProfScope ps(ctx, s, K_GN_REFINE);
    EBVO_HIP(ctx, hipMemsetAsync(A.counts, 0, sizeof(int32_t) * ((size_t)max_iter + 2), s.stream));
    const unsigned blocks = (unsigned)((n_pairs + 255) / 256 < 4096 ? (n_pairs + 255) / 256 : 4096);
    {
        const dim3 pg((w + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL(gn_pack_kernel, pg, dim3(256), 0, s.stream, d_imgL, h, w, w, pix4L, (uint4 *)nullptr);
        hipLaunchKernelGGL(gn_pack_kernel, pg, dim3(256), 0, s.stream, d_imgR, h, w, w, (uint32_t *)nullptr, recR);
    }
    hipLaunchKernelGGL(gn_left_kernel, dim3((unsigned)((nL + 31) / 32 < 8192 ? (nL + 31) / 32 : 8192)), dim3(256), 0,
                       s.stream, A);
    hipLaunchKernelGGL(gn_init_kernel, dim3(blocks), dim3(256), 0, s.stream, A);
    const bool no_rows = ctx->gn_no_rows != 0;                                  // developer keys (ebvo_debug_set 4 / 5)
    A.rows_below = ctx->gn_rows_below > 0 ? ctx->gn_rows_below : GN_ROWS_BELOW; // tools/gpu_gn_sweep.sh sweeps it
    // Iterations: the one-thread-per-pair layout, a launch per iteration, while more than rows_below pairs are active (each
    // of these launches returns at once when it finds fewer); then ONE launch of the eight-lanes-per-pair layout runs every
    // remaining pair to its convergence.  A problem that cannot hold more than rows_below pairs gets that launch only.
    if (no_rows)
        for (int it = 0; it < max_iter; ++it)
            hipLaunchKernelGGL(gn_iter_kernel, dim3(blocks), dim3(256), 0, s.stream, A, it, 0);
    else
    {
        if (n_pairs > A.rows_below)
            for (int it = 0; it < max_iter; ++it)
                hipLaunchKernelGGL(gn_iter_kernel, dim3(blocks), dim3(256), 0, s.stream, A, it, 1);
        const int64_t most = n_pairs < A.rows_below ? n_pairs : A.rows_below; // pairs the persistent launch can be handed
        // at most gn_persist_blocks workgroups (4 waves each; 256 = one wave per SIMD): with fewer groups than pairs a group
        // takes several pairs one after the other, which evens out the 1 .. max_iter iterations the pairs need.  With the
        // groups walking their own sequences (no shared cursor) more groups balance better: tools/gpu_gn_ab.py,
        // refinement stage at 256 / 512 / 1024 / 8192 blocks: 0.477 / 0.405 / 0.391 / 0.390 ms (EuRoC size), 1.10 / 0.99 / 1.00 / 1.02
        // (KITTI size); with the cursor 256 had been the optimum
        const int64_t bcap = ctx->gn_persist_blocks > 0 ? ctx->gn_persist_blocks : 1024;
        const unsigned pblocks = (unsigned)((most + 31) / 32 < bcap ? (most + 31) / 32 : bcap);
        if (ctx->gn_per_iteration_rows) // developer key: the row layout as a launch per iteration (the form before the persistent kernel)
            for (int it = 0; it < max_iter; ++it)
                hipLaunchKernelGGL(gn_iter_rows_kernel, dim3(pblocks), dim3(256), 0, s.stream, A, it, n_pairs > A.rows_below ? 2 : 0);
        else if (ctx->gn_persist_waves == 3)
            hipLaunchKernelGGL((gn_rows_persistent_kernel<3>), dim3(pblocks), dim3(256), 0, s.stream, A);
        else
            hipLaunchKernelGGL((gn_rows_persistent_kernel<2>), dim3(pblocks), dim3(256), 0, s.stream, A);
    }
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int refine_gn_temporal_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgK, const uint8_t *d_imgC, const void *d_gxy,
                               int h, int w, const ebvo_edge *d_kf, const ebvo_edge *d_cf, const double *d_init, int64_t n,
                               int max_iter, double tol, double huber, double *d_disp, double *d_score, uint8_t *d_valid,
                               int32_t *d_iters, int64_t n_first, const uint8_t *d_imgK2, const uint8_t *d_imgC2,
                               const void *d_gxy2)
{
    if (n <= 0)
        return EBVO_OK;
    if (n > 0x7fffffffll)
        return EBVO_ERR_CAPACITY;
    if (w < 2 || h < 1)
        return EBVO_ERR_ARG;
    int rc;
    const size_t np = (size_t)n;
    const size_t npx = (size_t)h * w;
    const bool two = d_imgC2 && d_imgC2 != d_imgC;
    if ((rc = ebvo_grow(ctx, s, s.gn_state, sizeof(double) * 6 * np + sizeof(float) * 98 * np)) ||
        (rc = ebvo_grow(ctx, s, s.gn_lists, sizeof(int32_t) * (2 * np + (size_t)max_iter + 2))) ||
        (rc = ebvo_grow(ctx, s, s.gn_pack, sizeof(uint4) * npx * (two ? 2 : 1))))
        return rc;
    Gn2Args A{};
    A.imgK = d_imgK;
    A.imgC = d_imgC;
    A.gxy = (const float2 *)d_gxy;
    A.n_first = (n_first >= 0 && d_imgK2) ? n_first : n; // items n_first .. n - 1 use the second image set
    A.imgK2 = d_imgK2 ? d_imgK2 : d_imgK;
    A.imgC2 = d_imgC2 ? d_imgC2 : d_imgC;
    A.gxy2 = d_gxy2 ? (const float2 *)d_gxy2 : (const float2 *)d_gxy;
    A.h = h;
    A.w = w;
    A.kf = d_kf;
    A.cf = d_cf;
    A.init = d_init;
    A.n = n;
    A.max_iter = max_iter;
    A.tol = tol;
    A.huber = huber;
    A.disp = d_disp;
    A.score = d_score;
    A.valid = d_valid;
    A.iters = d_iters;
    A.mean_l = (double *)s.gn_state.p;
    A.sc = A.mean_l + 2 * np;
    A.lrec = (float *)(A.sc + 4 * np);
    uint4 *rec = (uint4 *)s.gn_pack.p;
    A.recC = rec;
    A.recC2 = two ? rec + npx : rec;
    A.list[0] = (int32_t *)s.gn_lists.p;
    A.list[1] = A.list[0] + np;
    A.counts = A.list[1] + np;
    ProfScope ps(ctx, s, K_GN_REFINE);
    EBVO_HIP(ctx, hipMemsetAsync(A.counts, 0, sizeof(int32_t) * ((size_t)max_iter + 2), s.stream));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    {
        // the current-frame images as packed corner records (intensity + Sobel gradients): one load per sample point
        const dim3 pg((w + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL(gn_pack_kernel, pg, dim3(256), 0, s.stream, d_imgC, h, w, w, (uint32_t *)nullptr, rec);
        if (two)
            hipLaunchKernelGGL(gn_pack_kernel, pg, dim3(256), 0, s.stream, d_imgC2, h, w, w, (uint32_t *)nullptr, rec + npx);
    }
    hipLaunchKernelGGL(gn2_init_kernel, dim3((unsigned)((n + 31) / 32 < 8192 ? (n + 31) / 32 : 8192)), dim3(256), 0, s.stream, A);
    const bool rows = n <= GN_ROWS_MAX_PAIRS && !ctx->gn_no_rows; // small batch: eight lanes per item
    const unsigned rblocks = (unsigned)((n + 31) / 32 < 8192 ? (n + 31) / 32 : 8192);
    if (rows && !ctx->gn_per_iteration_rows) // (developer key 7: the row layout as a launch per iteration, the form before)
        hipLaunchKernelGGL(gn2_rows_persistent_kernel, dim3(rblocks < 1024 ? rblocks : 1024), dim3(256), 0, s.stream, A);
    else
        for (int it = 0; it < max_iter; ++it)
            if (rows)
                hipLaunchKernelGGL(gn2_iter_rows_kernel, dim3(rblocks), dim3(256), 0, s.stream, A, it);
            else
                hipLaunchKernelGGL(gn2_iter_kernel, dim3(blocks), dim3(256), 0, s.stream, A, it);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

static double cof3_host(const double *m, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}

// Matrix3d::inverse() (Eigen/src/LU/InverseImpl.h, cofactor form), row-major
static void inverse3_host(const double *m, double *inv)
{
    const double c0 = cof3_host(m, 0, 0), c1 = cof3_host(m, 1, 0), c2 = cof3_host(m, 2, 0);
    const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const double invdet = 1.0 / det;
    inv[0] = c0 * invdet;
    inv[1] = c1 * invdet;
    inv[2] = c2 * invdet;
    for (int i = 1; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            inv[i * 3 + j] = cof3_host(m, j, i) * invdet;
}

int refine_finalize_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const double *K_left, const double *K_right, const double *R21,
                                  const double *T21, const ebvo_edge *d_L, const ebvo_edge *d_R, int n, double *d_out,
                                  const int32_t *d_n)
{
    if (n <= 0)
        return EBVO_OK;
    FinalCalib C;
    inverse3_host(K_left, C.Kli);
    inverse3_host(K_right, C.Kri);
    for (int i = 0; i < 9; ++i)
        C.R21[i] = R21[i];
    for (int i = 0; i < 3; ++i)
        C.T21[i] = T21[i];
    ProfScope ps(ctx, s, K_MISC);
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(finalize_pairs_kernel, dim3(blocks), dim3(256), 0, s.stream, C, d_L, d_R, DevCount{n, d_n}, d_out);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

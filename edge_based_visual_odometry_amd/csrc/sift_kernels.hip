// sift_kernels.hip -- fixed-scale SIFT descriptors at the +-8 px points of every edge and the descriptor-distance filter.
//
// Replaces, in the reference (OpenCV's cv::SIFT is third-party and absent from the reference tree:
// PARITY UNPINNED, restated from the published OpenCV 4.x source, see oracle/ebvo_oracle.c: orc_sift_*):
//   Stereo_Matches::augment_Edge_Data            src/Stereo_Matches.cpp:655-689   (cv::SIFT::compute per left edge)
//   Stereo_Matches::apply_SIFT_filtering         :691-787                          (compute per candidate + min L2 distance)
//   finalize_stereo_edge_mates (descriptors)     :1627-1635
//
// A cv::KeyPoint(pt, size 1, angle) has octave 0 and layer 0, so cv::SIFT::compute builds one octave and reads the
// descriptor from its first level: GaussianBlur(float(image), sigma = sqrt(1.6^2 - 0.5^2), 13 taps, reflect-101).  The
// reference rebuilds that pyramid once per EDGE (a full-image blur per call: SURVEY.md calls it the real CPU bottleneck);
// here the level is formed once per image (sift_blur_*), and every keypoint is one thread:
//   - 11 x 11 window of central differences around the rounded point, rotated into the keypoint frame (hist_width 1.5),
//   - tri-linear accumulation into the (4 + 2) x (4 + 2) x (8 + 2) histogram IN SAMPLE ORDER (the scalar loop of
//     calcSIFTDescriptor: float additions are not associative, so the order is part of the result).  The histogram is
//     private to the thread and lives in LDS: hist[bin][lane].  Only the 4 x 4 x (8 + 2) cells that reach the descriptor
//     are stored (the border rows / columns of OpenCV's array are dropped at the end): 160 x 64 floats = 40 KB per
//     64-thread workgroup, four workgroups per CU.  Lane l only ever touches column l: whatever bins the lanes address,
//     the 64 accesses of a wave fall on 64 different banks;
//   - circular orientation wrap, 0.2 clipping, x 512, round to 0 .. 255.
// The alternative orders (LDS atomics, per-cell ownership) either change the bits from run to run or multiply the work.
//
// Compiled with -ffp-contract=off (float multiply and add stay separate, as in the oracle).
#include "ebvo_internal.h"
#include "ebvo_math.h"

namespace
{

struct Taps13
{
    float k[13];
};

__device__ inline int reflect101(int p, int n)
{
    if (n == 1)
        return 0;
    while (p < 0 || p >= n)
        p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// row pass: taps in ascending order (cv::RowFilter), float
__global__ __launch_bounds__(256) void sift_blur_rows_kernel(const uint8_t *__restrict__ img, int h, int w, int pitch, Taps13 T,
                                                             float *__restrict__ tmp)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    const uint8_t *row = img + (size_t)y * pitch;
    float s = (float)row[reflect101(x - 6, w)] * T.k[0];
#pragma unroll
    for (int t = 1; t < 13; ++t)
        s += (float)row[reflect101(x - 6 + t, w)] * T.k[t];
    tmp[(size_t)y * w + x] = s;
}

// column pass: centre first, then the symmetric pairs (cv::SymmColumnFilter)
__global__ __launch_bounds__(256) void sift_blur_cols_kernel(const float *__restrict__ tmp, int h, int w, Taps13 T,
                                                             float *__restrict__ base)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    float s = T.k[6] * tmp[(size_t)y * w + x];
#pragma unroll
    for (int t = 1; t <= 6; ++t)
        s += T.k[6 + t] * (tmp[(size_t)reflect101(y + t, h) * w + x] + tmp[(size_t)reflect101(y - t, h) * w + x]);
    base[(size_t)y * w + x] = s;
}

constexpr int SD = 4, SN = 8;                                  // SIFT_DESCR_WIDTH, SIFT_DESCR_HIST_BINS
constexpr int SHIST = SD * SD * (SN + 2);                      // 160: the cells that reach the descriptor (see the vote)
constexpr float FLT_EPS = 1.1920928955078125e-07f;

__device__ inline int cv_round_f(float v) { return (int)rintf(v); }

// ebvo_expf / ebvo_fast_atan2_deg (ebvo_math.h) with their case distinctions as selects: the same operations on the path
// the original takes, no branches -- the three dependent chains of a sample (orientation, magnitude, weight) then sit in
// one basic block and overlap (one wave per SIMD: nothing else hides their latency)
__device__ inline float sift_expf_sel(float x)
{
    const float fk = x * 1.44269504088896341f;
    const int k = (int)(fk + (fk >= 0.0f ? 0.5f : -0.5f));
    const float dk = (float)k;
    const float r = (x - dk * 0.693359375f) - dk * -2.12194440e-4f;
    float p = 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    float v = __builtin_ldexpf(p, k);
    v = x < -87.0f ? 0.0f : v;
    v = x > 88.0f ? 3.4028234663852886e38f * 2.0f : v;
    return x != x ? x : v;
}

__device__ inline float sift_fast_atan2_sel(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const bool xs = ax >= ay;
    const float c = (xs ? ay : ax) / ((xs ? ax : ay) + (float)2.2204460492503131e-16);
    const float c2 = c * c;
    const float t = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = xs ? t : 90.f - t;
    a = x < 0 ? 180.f - a : a;
    a = y < 0 ? 360.f - a : a;
    return a;
}

// calcSIFTDescriptor (modules/features2d/src/sift.simd.hpp), one thread per keypoint (edge e, side sd)
// `list` (with its length on the device, *n_list): the descriptors of those edges only, each written at its edge's own place.
__device__ __forceinline__ void sift_desc_body(float *__restrict__ hist, const float *__restrict__ base, int rows, int cols,
                                               const ebvo_edge *__restrict__ edges, int n_edges, float *__restrict__ desc_f,
                                               uint8_t *__restrict__ desc_u8, const int32_t *__restrict__ list,
                                               const int32_t *__restrict__ n_list, int vb, int vg)
{
    const int lane = threadIdx.x;
    if (list)
        n_edges = min(*n_list, n_edges);
    const int n_kp = n_edges * 2;
    for (int base_kp = vb * 64; base_kp < n_kp; base_kp += vg * 64)
    {
        const int kp_in = base_kp + lane;
        if (kp_in >= n_kp)
            continue; // no barrier below: every lane owns its own histogram column
        const int e = list ? list[kp_in >> 1] : kp_in >> 1, sd = kp_in & 1;
        const int kp = 2 * e + sd; // where the descriptor goes
        const ebvo_edge ed = edges[e];
        double sn, cs;
        ebvo_sincos(ed.theta, &sn, &cs);
        // get_Orthogonal_Shifted_Points(edge, 8), src/utility.cpp:128-139; cv::KeyPoint takes a Point2f
        const double pxd = sd ? ed.x + 8 * (-sn) : ed.x + 8 * (sn);
        const double pyd = sd ? ed.y + 8 * (cs) : ed.y + 8 * (-cs);
        const float ptx = (float)pxd, pty = (float)pyd;
        const float kp_angle = (float)(180 / 3.14159265358979323846 * ed.theta);
        float ori = 360.f - kp_angle;
        if (fabsf(ori - 360.f) < FLT_EPS)
            ori = 0.f;
        const int px = cv_round_f(ptx), py = cv_round_f(pty);
        const float arg = ori * (float)(3.14159265358979323846 / 180);
        double sd_, cd_;
        ebvo_sincos((double)arg, &sd_, &cd_);
        float cos_t = (float)cd_, sin_t = (float)sd_;
        const float bins_per_rad = SN / 360.f;
        const float exp_scale = -1.f / (SD * SD * 0.5f);
        const float hist_width = 3.0f * 0.5f;
        int radius = cv_round_f(hist_width * 1.4142135623730951f * (SD + 1) * 0.5f);
        const int diag = (int)sqrt(((double)cols) * cols + ((double)rows) * rows);
        radius = radius < diag ? radius : diag;
        cos_t /= hist_width;
        sin_t /= hist_width;
        float *H = hist + lane;
        for (int b = 0; b < SHIST; ++b)
            H[b * 64] = 0.f;
        // one sample that lies in the rotated window: its gradient (dx, dy) and rotated position -> the tri-linear vote
        auto vote_sample = [&](float dx, float dy, float c_rot, float r_rot, float rbin, float cbin) {
            const float wexp = (c_rot * c_rot + r_rot * r_rot) * exp_scale;
            const float Ori = sift_fast_atan2_sel(dy, dx);
            const float Mag = sqrtf(dx * dx + dy * dy);
            const float W = sift_expf_sel(wexp);
            float obin = (Ori - ori) * bins_per_rad;
            const float mag = Mag * W;
            const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
            int o0 = (int)floorf(obin);
            rbin -= r0;
            cbin -= c0;
            obin -= o0;
            if (o0 < 0)
                o0 += SN;
            if (o0 >= SN)
                o0 -= SN;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
            const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
            const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            // The eight cells of the tri-linear vote, in the order of calcSIFTDescriptor.  Two things shape the
            // indexing.  (1) The border rows and columns of OpenCV's (d + 2) x (d + 2) histogram only collect what
            // falls outside the descriptor and are dropped at the end: they are not stored here (every kept cell
            // still receives its own addends in sample order), which leaves 4 x 4 x (8 + 2) floats per keypoint,
            // 40 KB per workgroup instead of 90.  (2) The reference passes orientations of -180 .. 180 degrees
            // where OpenCV expects 0 .. 360, so o0 can stay negative (down to -4) after the single "+= n": in
            // OpenCV's FLAT array such a vote lands in the upper bins of the previous cell (column - 1, or the
            // last column of the row above).  That aliasing is part of the reference's result and is reproduced
            // by the carry below; a vote in front of the array (undefined behaviour in OpenCV) is dropped, as
            // in the restatement (oracle/ebvo_oracle.c: calc_sift_descriptor).
#define EBVO_SIFT_VOTE(dr, dc, dob, val)                                                                      \
    {                                                                                                         \
int rr_ = r0 + dr, cc_ = c0 + dc, oo_ = o0 + dob; /* interior coordinates: -1 .. SD */                \
if (oo_ < 0)                                                                                          \
{                                                                                                     \
    oo_ += SN + 2;                                                                                    \
    cc_ -= 1;                                                                                         \
    if (cc_ < -1)                                                                                     \
    {                                                                                                 \
        cc_ = SD;                                                                                     \
        rr_ -= 1;                                                                                     \
    }                                                                                                 \
}                                                                                                     \
if ((unsigned)rr_ < (unsigned)SD && (unsigned)cc_ < (unsigned)SD)                                     \
    H[((rr_ * SD + cc_) * (SN + 2) + oo_) * 64] += (val);                                             \
    }
            {
                // all eight votes without branches: cells by the rules of the macro above as selects; a vote outside the
                // 4 x 4 interior adds +0 to a bin none of the eight can be (cell 0, three bins past the sample's first:
                // its votes use two adjacent bins; bins are sums of non-negative terms, x + 0 == x).  The eight cells
                // are distinct, so they are read together, updated and written together.
                const float add[8] = {v_rco000, v_rco001, v_rco010, v_rco011, v_rco100, v_rco101, v_rco110, v_rco111};
                const int first = o0 < 0 ? o0 + SN + 2 : o0;
                const int scratch = (first + 3 < SN + 2 ? first + 3 : first + 3 - (SN + 2)) * 64;
                int idx[8];
                float val[8], cur[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const int dr = q >> 2, dc = (q >> 1) & 1, dob = q & 1;
                    int rr = r0 + dr, cq = c0 + dc, oo = o0 + dob;
                    const bool neg = oo < 0;
                    oo = neg ? oo + SN + 2 : oo;
                    cq = neg ? cq - 1 : cq;
                    const bool wrap = cq < -1;
                    cq = wrap ? SD : cq;
                    rr = wrap ? rr - 1 : rr;
                    const bool valid = (unsigned)rr < (unsigned)SD && (unsigned)cq < (unsigned)SD;
                    idx[q] = valid ? ((rr * SD + cq) * (SN + 2) + oo) * 64 : scratch;
                    val[q] = valid ? add[q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    cur[q] = H[idx[q]];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    H[idx[q]] = cur[q] + val[q];
            }
#undef EBVO_SIFT_VOTE
        };
        // A wave whose keypoints all have their 13 x 13 neighbourhood inside the image (all but the waves at the border) needs
        // no image tests, and fetches the four pixels of a sample one sample ahead, whether the sample will count or not:
        // their latency runs under the previous sample's arithmetic.
        const bool inside = radius == 5 && px >= 6 && px + 6 <= cols - 1 && py >= 6 && py + 6 <= rows - 1;
        if (__all(inside))
        {
            const float *ctr = base + (size_t)(py - 5) * cols + (px - 5); // the pixel of sample (i, j) = (-5, -5)
            float n0 = ctr[1], n1 = ctr[-1], n2 = ctr[-cols], n3 = ctr[cols];
#pragma unroll 1
            for (int i = -5; i <= 5; ++i)
#pragma unroll 1
                for (int j = -5; j <= 5; ++j)
                {
                    const float dx = n0 - n1, dy = n2 - n3;
                    const bool last = i == 5 && j == 5;
                    ctr += last ? 0 : (j == 5 ? cols - 10 : 1); // the next sample's pixel
                    n0 = ctr[1];
                    n1 = ctr[-1];
                    n2 = ctr[-cols];
                    n3 = ctr[cols];
                    const float c_rot = j * cos_t - i * sin_t;
                    const float r_rot = j * sin_t + i * cos_t;
                    const float rbin = r_rot + SD / 2 - 0.5f;
                    const float cbin = c_rot + SD / 2 - 0.5f;
                    if (rbin > -1 && rbin < SD && cbin > -1 && cbin < SD)
                        vote_sample(dx, dy, c_rot, r_rot, rbin, cbin);
                }
        }
        else
            for (int i = -radius; i <= radius; ++i)
                for (int j = -radius; j <= radius; ++j)
                {
                    const float c_rot = j * cos_t - i * sin_t;
                    const float r_rot = j * sin_t + i * cos_t;
                    const float rbin = r_rot + SD / 2 - 0.5f;
                    const float cbin = c_rot + SD / 2 - 0.5f;
                    const int r = py + i, c = px + j;
                    if (rbin > -1 && rbin < SD && cbin > -1 && cbin < SD && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)
                        vote_sample(base[(size_t)r * cols + c + 1] - base[(size_t)r * cols + c - 1],
                                    base[(size_t)(r - 1) * cols + c] - base[(size_t)(r + 1) * cols + c], c_rot, r_rot, rbin, cbin);
                }
        // the orientation histograms are circular; the 4 x 4 x 8 interior is the raw descriptor (kept in place)
        float nrm2 = 0;
        for (int i = 0; i < SD; ++i)
            for (int j = 0; j < SD; ++j)
            {
                const int idx = (i * SD + j) * (SN + 2);
                H[idx * 64] += H[(idx + SN) * 64];
                H[(idx + 1) * 64] += H[(idx + SN + 1) * 64];
            }
        // (the eight bins of a cell are read together; the sums still take their terms one by one, in bin order)
        for (int cell = 0; cell < SD * SD; ++cell)
        {
            float v[SN];
#pragma unroll
            for (int k = 0; k < SN; ++k)
                v[k] = H[(cell * (SN + 2) + k) * 64];
#pragma unroll
            for (int k = 0; k < SN; ++k)
                nrm2 += v[k] * v[k];
        }
        const float thr = sqrtf(nrm2) * 0.2f;
        nrm2 = 0;
        for (int cell = 0; cell < SD * SD; ++cell)
        {
            float v[SN];
#pragma unroll
            for (int k = 0; k < SN; ++k)
                v[k] = H[(cell * (SN + 2) + k) * 64];
#pragma unroll
            for (int k = 0; k < SN; ++k)
            {
                const float val = v[k] < thr ? v[k] : thr;
                H[(cell * (SN + 2) + k) * 64] = val;
                nrm2 += val * val;
            }
        }
        const float sq = sqrtf(nrm2);
        nrm2 = 512.f / (sq > FLT_EPS ? sq : FLT_EPS);
        for (int cell = 0; cell < SD * SD; ++cell)
        {
            // a cell's eight values leave as one 8-byte (u8) / two 16-byte (float) stores: a store per value was 128 (+ 128)
            // scattered store instructions per keypoint
            int q[SN];
#pragma unroll
            for (int k = 0; k < SN; ++k)
            {
                const int t = cv_round_f(H[(cell * (SN + 2) + k) * 64] * nrm2);
                q[k] = t < 0 ? 0 : (t > 255 ? 255 : t);
            }
            const size_t o = (size_t)kp * 128 + cell * SN;
            if (desc_f)
            {
                float4 *d = reinterpret_cast<float4 *>(desc_f + o);
                d[0] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
                d[1] = make_float4((float)q[4], (float)q[5], (float)q[6], (float)q[7]);
            }
            if (desc_u8)
            {
                uint2 w;
                w.x = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
                w.y = (unsigned)q[4] | ((unsigned)q[5] << 8) | ((unsigned)q[6] << 16) | ((unsigned)q[7] << 24);
                *reinterpret_cast<uint2 *>(desc_u8 + o) = w;
            }
        }
    }
}

__global__ __launch_bounds__(64) void sift_desc_kernel(const float *__restrict__ base, int rows, int cols,
                                                       const ebvo_edge *__restrict__ edges, int n_edges,
                                                       float *__restrict__ desc_f, uint8_t *__restrict__ desc_u8,
                                                       const int32_t *__restrict__ list, const int32_t *__restrict__ n_list)
{
    __shared__ float hist[SHIST * 64];
    sift_desc_body(hist, base, rows, cols, edges, n_edges, desc_f, desc_u8, list, n_list, blockIdx.x, gridDim.x);
}

// the listed descriptors of BOTH images of a pair in one launch (blockIdx.y = image): a launch per image ran ~1.2 rounds of
// the 1,024 waves the device holds (one per SIMD: the histograms fill the LDS), i.e. two rounds with the second a fifth full
struct SiftPairArgs
{
    const float *base[2];
    const ebvo_edge *edges[2];
    const int32_t *list[2], *n_list[2];
    uint8_t *desc_u8[2];
    int n_max[2];
};
__global__ __launch_bounds__(64) void sift_desc_pair_kernel(SiftPairArgs A, int rows, int cols)
{
    __shared__ float hist[SHIST * 64];
    const int im = blockIdx.y;
    sift_desc_body(hist, A.base[im], rows, cols, A.edges[im], A.n_max[im], nullptr, A.desc_u8[im], A.list[im], A.n_list[im],
                   blockIdx.x, gridDim.x);
}

// float descriptors (host-buffer call) -> bytes
__global__ void sift_to_u8_kernel(const float *__restrict__ f, int64_t n, uint8_t *__restrict__ u)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        u[k] = (uint8_t)f[k];
}

// apply_SIFT_filtering's score (src/Stereo_Matches.cpp:736-740): min of the four L2 distances between the two descriptors
// of the left edge and the two of the candidate.  Descriptor entries are integers 0 .. 255: the sums of squares are exact
// in any order, the distance is sqrt of an exact integer.  Sixteen lanes per pair, eight bytes per lane and descriptor.
__global__ __launch_bounds__(256) void sift_dist_kernel(const uint8_t *__restrict__ left, const uint8_t *__restrict__ cand,
                                                        const int32_t *__restrict__ pair_left,
                                                        const int32_t *__restrict__ cand_index /* NULL: pair k -> cand k */,
                                                        int64_t n_pairs, double thr, double *__restrict__ dist,
                                                        uint8_t *__restrict__ ok)
{
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = (int)(t & 15);
    for (int64_t k = t >> 4; k < ((n_pairs + groups - 1) / groups) * groups; k += groups)
    {
        const bool valid = k < n_pairs;
        unsigned s[4] = {0, 0, 0, 0};
        if (valid)
        {
            const size_t li = (size_t)pair_left[k], ri = cand_index ? (size_t)cand_index[k] : (size_t)k;
            unsigned long long a[2], b[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
            {
                a[q] = *reinterpret_cast<const unsigned long long *>(left + (li * 2 + q) * 128 + g * 8);
                b[q] = *reinterpret_cast<const unsigned long long *>(cand + (ri * 2 + q) * 128 + g * 8);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) // :736-739 order: (L1,R1), (L2,R1), (L1,R2), (L2,R2)
            {
                const unsigned long long x = a[tt & 1], y = b[tt >> 1];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const int d = (int)((x >> (8 * q)) & 0xff) - (int)((y >> (8 * q)) & 0xff);
                    s[tt] += (unsigned)(d * d);
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            for (int m = 1; m < 16; m <<= 1)
                s[tt] += __shfl_xor(s[tt], m);
        if (valid && g == 0)
        {
            double best = sqrt((double)s[0]);
#pragma unroll
            for (int tt = 1; tt < 4; ++tt)
            {
                const double d = sqrt((double)s[tt]);
                if (d < best)
                    best = d;
            }
            if (dist)
                dist[k] = best;
            if (ok)
                ok[k] = best < thr ? 1 : 0; // :752
        }
    }
}

__global__ void and_flags_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int64_t n, uint8_t *__restrict__ o)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        o[k] = (a[k] && b[k]) ? 1 : 0;
}

Taps13 host_taps()
{
    // createInitialImage: sig_diff = sqrtf(max(sigma^2 - 0.5^2, 0.01f)) in float; getGaussianKernel(13, sig_diff, CV_32F)
    const float sigma = 1.6f;
    const float v = sigma * sigma - 0.5f * 0.5f;
    const float sd = sqrtf(v > 0.01f ? v : 0.01f);
    const double sigmaX = (double)sd, scale2X = -0.5 / (sigmaX * sigmaX);
    double kd[13], sum = 0;
    for (int i = 0; i < 13; ++i)
    {
        const double x = i - (13 - 1) * 0.5;
        kd[i] = exp(scale2X * x * x);
        sum += kd[i];
    }
    sum = 1. / sum;
    Taps13 T;
    for (int i = 0; i < 13; ++i)
        T.k[i] = (float)(kd[i] * sum);
    return T;
}

inline unsigned grid1d(int64_t items, int per_block, int max_blocks)
{
    int64_t b = (items + per_block - 1) / per_block;
    b = b < 1 ? 1 : (b > max_blocks ? max_blocks : b);
    return (unsigned)b;
}

} // namespace

// gpyr[0] of cv::SIFT::compute for a resident image: d_tmp, d_base are h x w floats
int sift_base_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch, float *d_tmp, float *d_base)
{
    static const Taps13 T = host_taps();
    ProfScope ps(ctx, s, K_SIFT);
    const dim3 g((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(sift_blur_rows_kernel, g, dim3(256), 0, s.stream, d_img, h, w, pitch, T, d_tmp);
    hipLaunchKernelGGL(sift_blur_cols_kernel, g, dim3(256), 0, s.stream, (const float *)d_tmp, h, w, T, d_base);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_descriptors_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base, int h, int w, const ebvo_edge *d_edges, int n,
                             float *d_desc_f, uint8_t *d_desc_u8)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_SIFT);
    hipLaunchKernelGGL(sift_desc_kernel, dim3(grid1d((int64_t)n * 2, 64, 1 << 20)), dim3(64), 0, s.stream, d_base, h, w, d_edges,
                       n, d_desc_f, d_desc_u8, (const int32_t *)nullptr, (const int32_t *)nullptr);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_descriptors_listed_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base, int h, int w, const ebvo_edge *d_edges,
                                    const int32_t *d_list, const int32_t *d_n, int n_max, uint8_t *d_desc_u8)
{
    if (n_max <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_SIFT);
    // one keypoint per thread, one wave per SIMD (the histograms fill the LDS): a grid of what the device holds, strided
    hipLaunchKernelGGL(sift_desc_kernel, dim3(grid1d((int64_t)n_max * 2, 64, 4096)), dim3(64), 0, s.stream, d_base, h, w, d_edges,
                       n_max, (float *)nullptr, d_desc_u8, d_list, d_n);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_descriptors_listed_pair_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_base0, const float *d_base1, int h, int w,
                                         const ebvo_edge *d_edges0, const ebvo_edge *d_edges1, const int32_t *d_list0,
                                         const int32_t *d_list1, const int32_t *d_n, int n_max0, int n_max1, uint8_t *d_desc0,
                                         uint8_t *d_desc1)
{
    const int nm = n_max0 > n_max1 ? n_max0 : n_max1;
    if (nm <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_SIFT);
    SiftPairArgs A;
    A.base[0] = d_base0, A.base[1] = d_base1;
    A.edges[0] = d_edges0, A.edges[1] = d_edges1;
    A.list[0] = d_list0, A.list[1] = d_list1;
    A.n_list[0] = d_n, A.n_list[1] = d_n + 1;
    A.desc_u8[0] = d_desc0, A.desc_u8[1] = d_desc1;
    A.n_max[0] = n_max0, A.n_max[1] = n_max1;
    hipLaunchKernelGGL(sift_desc_pair_kernel, dim3(grid1d((int64_t)nm * 2, 64, 4096), 2), dim3(64), 0, s.stream, A, h, w);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

namespace
{
// flags[j] = 1 for every right edge some pair names (benign race: every writer stores 1)
__global__ __launch_bounds__(256) void sift_mark_right_kernel(const int32_t *__restrict__ col_idx, int64_t n_pairs, uint8_t *__restrict__ flags)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n_pairs; k += (int64_t)gridDim.x * 256)
        flags[col_idx[k]] = 1;
}

// blockIdx.y = 0: left edges whose CSR row is not empty; 1: flagged right edges.  One atomic per block and list (the order of a
// list does not matter: every descriptor is written at its edge's own place).
__global__ __launch_bounds__(256) void sift_used_lists_kernel(const int32_t *__restrict__ row_ptr, int nL, const uint8_t *__restrict__ flags,
                                                              int nR, int32_t *__restrict__ counts, int32_t *__restrict__ list_left,
                                                              int32_t *__restrict__ list_right)
{
    __shared__ int s_w[4], s_base;
    const int side = blockIdx.y, n = side ? nR : nL;
    int32_t *list = side ? list_right : list_left;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256)
    {
        const int i = i0 + threadIdx.x;
        const bool used = i < n && (side ? flags[i] != 0 : row_ptr[i + 1] > row_ptr[i]);
        const unsigned long long m = __ballot(used);
        if (lane == 0)
            s_w[wid] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            s_base = tot ? atomicAdd(&counts[side], tot) : 0;
        }
        __syncthreads();
        if (used)
        {
            int pre = 0;
            for (int k = 0; k < wid; ++k)
                pre += s_w[k];
            list[s_base + pre + __popcll(m & ((1ull << lane) - 1ull))] = i;
        }
        __syncthreads();
    }
}
} // namespace

int sift_used_edges_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const int32_t *d_col_idx, int64_t n_pairs,
                            int nR, int32_t *d_counts, int32_t *d_list_left, int32_t *d_list_right, uint8_t *d_flags)
{
    ProfScope ps(ctx, s, K_SIFT);
    EBVO_HIP(ctx, hipMemsetAsync(d_counts, 0, 2 * sizeof(int32_t), s.stream));
    if (nR > 0)
        EBVO_HIP(ctx, hipMemsetAsync(d_flags, 0, (size_t)nR, s.stream));
    if (n_pairs > 0)
        hipLaunchKernelGGL(sift_mark_right_kernel, dim3(grid1d(n_pairs, 256, 1024)), dim3(256), 0, s.stream, d_col_idx, n_pairs, d_flags);
    const int nmax = nL > nR ? nL : nR;
    if (nmax > 0)
        hipLaunchKernelGGL(sift_used_lists_kernel, dim3(grid1d(nmax, 256, 256), 2), dim3(256), 0, s.stream, d_row_ptr, nL,
                           (const uint8_t *)d_flags, nR, d_counts, d_list_left, d_list_right);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_to_u8_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_f, int64_t n, uint8_t *d_u8)
{
    if (n <= 0)
        return EBVO_OK;
    hipLaunchKernelGGL(sift_to_u8_kernel, dim3(grid1d(n, 256, 4096)), dim3(256), 0, s.stream, d_f, n, d_u8);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

// descriptor pairs (256 bytes each) of selected edges: dst[k] = src[index[k]], sixteen lanes per entry
__global__ __launch_bounds__(256) void sift_gather_kernel(const uint8_t *__restrict__ src, const int32_t *__restrict__ index, int n,
                                                          uint8_t *__restrict__ dst)
{
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = (int)(t & 15);
    for (int64_t k = t >> 4; k < n; k += groups)
        reinterpret_cast<uint4 *>(dst + (size_t)k * 256)[g] = reinterpret_cast<const uint4 *>(src + (size_t)index[k] * 256)[g];
}

int sift_gather_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_src, const int32_t *d_index, int n, uint8_t *d_dst)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_SIFT);
    hipLaunchKernelGGL(sift_gather_kernel, dim3(grid1d((int64_t)n * 16, 256, 4096)), dim3(256), 0, s.stream, d_src, d_index, n, d_dst);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_distances_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_left, const uint8_t *d_cand, const int32_t *d_pair_left,
                           const int32_t *d_cand_index, int64_t n_pairs, double thr, double *d_dist, uint8_t *d_ok)
{
    if (n_pairs <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_SIFT);
    hipLaunchKernelGGL(sift_dist_kernel, dim3(grid1d(n_pairs * 16, 256, 4096)), dim3(256), 0, s.stream, d_left, d_cand,
                       d_pair_left, d_cand_index, n_pairs, thr, d_dist, d_ok);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int sift_and_flags_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_a, const uint8_t *d_b, int64_t n, uint8_t *d_out)
{
    if (n <= 0)
        return EBVO_OK;
    hipLaunchKernelGGL(and_flags_kernel, dim3(grid1d(n, 256, 4096)), dim3(256), 0, s.stream, d_a, d_b, n, d_out);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

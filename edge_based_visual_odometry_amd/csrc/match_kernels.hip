// match_kernels.hip -- epipolar candidate search, geometric filters and NCC patch scoring on gfx950.
//
// Replaces, in the reference:
//   Stereo_Matches::CalculateEpipolarLine                      src/Stereo_Matches.cpp:10-20
//   extract_Epipolar_Edge_Indices / apply_Epipolar_Line_...    src/Stereo_Matches.cpp:91-109, :381-419
//   apply_Disparity_Filtering / apply_orientation_filter       src/Stereo_Matches.cpp:534-553, :863-915
//   Utility::get_edge_patches / get_patch_similarity           src/utility.cpp:182-212, :163-180
//   Bilinear_Interpolation<double>                             include/utility.h:81-104
//   Stereo_Matches::apply_NCC_Filtering                        src/Stereo_Matches.cpp:555-616
//   Temporal_Matches::apply_NCC_filtering_quads (scoring)      src/Temporal_Matches.cpp:426-468
//
// Candidate search.  The reference scans all N_R right edges for each of the N_L left edges
// (O(N_L * N_R) fp64 predicates).  Here the right edges are summarised by bounding boxes over
// consecutive index ranges (8 edges per chunk).  A block owns 64 consecutive left edges: it
// selects, in index order, the chunks whose box can intersect the union of its lefts' search
// regions ({epipolar band} ∩ {disparity square}), stages their edges in LDS, and then every lane
// walks only the staged chunks that can intersect ITS region and evaluates the reference's exact
// predicates on their edges -- all from LDS, in ascending right index.  The boxes are a conservative
// pre-filter only: the result (set AND order) is identical to the brute force for any input order;
// TOED's raster order just makes the boxes tight.  The exact predicates avoid the fp64 division and
// square root except within 2^-50 of a threshold (see pair_passes).
//
// NCC.  A 49-term reduction is a left-to-right row sum in each lane followed by a 3-step xor butterfly over 8 lanes --
// exactly the canonical order of the CPU path (oracle/ebvo_oracle.c: reduce49).  Element-wise patch arithmetic is float
// (CV_32F), reductions are double.  Host-buffer calls: 16 lanes per (left edge, candidate) pair, lanes 0-6 the rows of
// the "plus" patch, lanes 8-14 the rows of the "minus" patch (ncc_pairs_kernel).  Resident pipeline: every right edge's
// patches are sampled and normalised once into a bank (right_bank_kernel); a wave owns four left edges, stages their
// patches in LDS and scores their CSR pairs eight lanes per pair (ncc_tile_kernel).
//
// Compiled with -ffp-contract=off: no FMA contraction anywhere; the explicit fma of dot7 is exact-product accumulation.
#include <atomic>
#include <cstdlib>

#include "ebvo_internal.h"
#include "ebvo_math.h"

namespace
{

constexpr int CHUNK = 8;         // edges per chunk box = lanes per left edge in the candidate walk
constexpr int GROUP = 64;        // chunks per group box (512 edges)
constexpr int BATCH = 64;        // chunks staged in LDS at a time (512 edges, 12 KB)
constexpr int TILE = 64;         // left edges per block of the candidate search
constexpr int STAGE = 64;        // candidates per left edge kept by the counting pass (rows beyond it are refilled)
constexpr double BOX_SLACK = 1e-6;

struct Box
{
    double x0, x1, y0, y1;
};

// A size that is either known on the host (dev == nullptr) or lives in device memory: the device-resident
// pipeline never reads a count back before the last kernel of a pair, so every kernel takes its sizes this way
// and runs grid-stride over whatever the count turns out to be.
struct DevN
{
    int host;
    const int32_t *dev;
};
__device__ inline int devn(const DevN &d) { return d.dev ? *d.dev : d.host; }

// Number of candidate pairs: host value, or row_ptr[nL] with nL itself on the device; clamped to the capacity
// of the pair-indexed buffers.
struct DevPairs
{
    int64_t host;
    const int32_t *row_ptr;
    DevN nL;
    int64_t cap;
};
__device__ inline int64_t devpairs(const DevPairs &d)
{
    const int64_t n = d.row_ptr ? (int64_t)d.row_ptr[devn(d.nL)] : d.host;
    return n < d.cap ? n : d.cap;
}

struct CandParams
{
    double epi_thr, max_disp, orient_thr;
    int mask;
    DevN nL, nR;
    int64_t cap;        // capacity of col_idx (FILL)
    int32_t *stage;     // [nL][STAGE] first STAGE candidates of every left edge, written by the counting pass
    int32_t *tile_flag; // [tiles] 1: some row of the tile has more than STAGE candidates -> the fill pass redoes the tile
};

// ------------------------------------------------------------------------------------------
// (the small per-pair kernels below have their bodies as device functions of a VIRTUAL block index / grid size: the resident
// pipeline runs four of them -- lines, boxes, sincos, row pairs -- as block ranges of ONE launch, match_prep_kernel)
__device__ inline void lines_body(const double *__restrict__ F, const ebvo_edge *__restrict__ e, int n, double *__restrict__ lines,
                                  int vb, int vg)
{
    for (int k = vb * blockDim.x + threadIdx.x; k < n; k += vg * blockDim.x)
    {
        const double x = e[k].x, y = e[k].y;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            lines[(size_t)k * 3 + r] = (F[r * 3 + 0] * x + F[r * 3 + 1] * y) + F[r * 3 + 2];
    }
}

__global__ void lines_kernel(const double *__restrict__ F, const ebvo_edge *__restrict__ e, DevN nd,
                             double *__restrict__ lines)
{
    lines_body(F, e, devn(nd), lines, blockIdx.x, gridDim.x);
}

__device__ inline double wave_min(double v)
{
    for (int d = 32; d > 0; d >>= 1)
        v = fmin(v, __shfl_xor(v, d));
    return v;
}
__device__ inline double wave_max(double v)
{
    for (int d = 32; d > 0; d >>= 1)
        v = fmax(v, __shfl_xor(v, d));
    return v;
}

// Bounding boxes of the index ranges: one thread per chunk (CHUNK edges), one wave per group (64 chunks), so the group
// box is a wave reduction of the chunk boxes it has just produced.
__device__ inline void boxes_body(const ebvo_edge *__restrict__ R, int nR, Box *__restrict__ cb, Box *__restrict__ gb,
                                  int32_t *__restrict__ tile_flag, int ntiles, int vb, int vg)
{
    static_assert(GROUP == 64, "one wave per group");
    // the tile flags of the counting pass that follows (round 4: zeroed here instead of by a launch of their own)
    for (int t = vb * blockDim.x + threadIdx.x; t < ntiles; t += vg * blockDim.x)
        tile_flag[t] = 0;
    const int nchunks = (nR + CHUNK - 1) / CHUNK, ngroups = (nchunks + GROUP - 1) / GROUP;
    const int lane = threadIdx.x & 63;
    const double inf = __builtin_inf();
    for (int g = vb * 4 + (threadIdx.x >> 6); g < ngroups; g += vg * 4)
    {
        const int c = g * GROUP + lane;
        Box b;
        b.x0 = inf; b.x1 = -inf; b.y0 = inf; b.y1 = -inf;
        if (c < nchunks)
        {
            const int k0 = c * CHUNK, k1 = min(nR, k0 + CHUNK);
            b.x0 = b.x1 = R[k0].x;
            b.y0 = b.y1 = R[k0].y;
            for (int k = k0 + 1; k < k1; ++k)
            {
                const double x = R[k].x, y = R[k].y;
                b.x0 = fmin(b.x0, x);
                b.x1 = fmax(b.x1, x);
                b.y0 = fmin(b.y0, y);
                b.y1 = fmax(b.y1, y);
            }
            cb[c] = b;
        }
        Box u;
        u.x0 = wave_min(b.x0); u.x1 = wave_max(b.x1); u.y0 = wave_min(b.y0); u.y1 = wave_max(b.y1);
        if (lane == 0)
            gb[g] = u;
    }
}

__global__ __launch_bounds__(256) void boxes_kernel(const ebvo_edge *__restrict__ R, DevN nRd, Box *__restrict__ cb,
                                                    Box *__restrict__ gb, int32_t *__restrict__ tile_flag, int ntiles)
{
    boxes_body(R, devn(nRd), cb, gb, tile_flag, ntiles, blockIdx.x, gridDim.x);
}

// Can any point of the box satisfy the enabled epipolar / disparity predicates?  Conservative.
__device__ inline bool box_may_match(const Box &bx, double xl, double yl, double ah, double bh, double ch,
                                     double D, double band, int mask)
{
    double x0 = bx.x0, x1 = bx.x1, y0 = bx.y0, y1 = bx.y1;
    if (mask & EBVO_STAGE_DISPARITY)
    {
        x0 = fmax(x0, xl - D);
        x1 = fmin(x1, xl + D);
        y0 = fmax(y0, yl - D);
        y1 = fmin(y1, yl + D);
        if (x0 > x1 || y0 > y1)
            return false;
    }
    if (mask & EBVO_STAGE_EPIPOLAR)
    {
        const double gx0 = ah * x0, gx1 = ah * x1, gy0 = bh * y0, gy1 = bh * y1;
        const double gmin = fmin(gx0, gx1) + fmin(gy0, gy1) + ch;
        const double gmax = fmax(gx0, gx1) + fmax(gy0, gy1) + ch;
        if (gmin > band || gmax < -band)
            return false;
    }
    return true;
}

// Per-left-edge constants of the predicates.
struct LeftCtx
{
    double lx, ly, lth;
    double a, b, c, nrm;    // epipolar line and sqrt(a*a + b*b) (src/Stereo_Matches.cpp:99)
    double ah, bh, ch;      // normalised line, for the conservative box tests only
    double t_lo, t_hi;      // epi_thr * nrm * (1 -+ 2^-50)
    double s_lo, s_hi;      // max_disp^2 * (1 -+ 2^-50)
};

// The reference's predicates, bit-exact.
//   epipolar (:99-101):  fl(|a x + b y + c| / nrm) < thr.   With t = fl(thr*nrm): |num| < t(1-2^-50) implies the
//     rounded quotient is < thr, |num| > t(1+2^-50) implies it is >= thr (division is monotone and correctly rounded,
//     all roundings involved are <= 2^-53 relative); only inside that sliver is the division evaluated.
//   disparity (:545-546): fl(sqrt(fl(dx*dx + dy*dy))) <= D, same argument on s = fl(dx*dx + dy*dy) against D^2.
//   orientation (:887-901): as written.
// Straight-line form: the three quantities are always evaluated and compared against both margins; only a lane inside
// a 2^-50 sliver (practically never) takes the branch with the division / square root.  One lane = one pair in the
// candidate walk, so an early exit saves nothing unless all 64 lanes take it, while every nested exit costs
// exec-mask bookkeeping.
__device__ inline bool pair_passes(const LeftCtx &l, double rx, double ry, double rth, const CandParams &P)
{
    bool e_fast = true, e_maybe = true, d_fast = true, d_maybe = true, o_ok = true;
    double num = 0.0, s = 0.0;
    if (P.mask & EBVO_STAGE_EPIPOLAR)
    {
        num = fabs(l.a * rx + l.b * ry + l.c);
        e_fast = num < l.t_lo;
        e_maybe = num <= l.t_hi; // false for NaN
    }
    if (P.mask & EBVO_STAGE_DISPARITY)
    {
        const double dx = l.lx - rx, dy = l.ly - ry;
        s = dx * dx + dy * dy;
        d_fast = s < l.s_lo;
        d_maybe = s <= l.s_hi;
    }
    if (P.mask & EBVO_STAGE_ORIENTATION)
    {
        double od = fabs((l.lth - rth) * 0x1.ca5dc1a63c1f8p+5 /* 180.0 / M_PI */);
        if (od > 180.0)
            od = 360.0 - od;
        o_ok = od < P.orient_thr || fabs(od - 180.0) < P.orient_thr;
    }
    bool ok = o_ok && e_maybe && d_maybe;
    if (ok && !(e_fast && d_fast))
    {
        asm volatile("" : "+v"(num), "+v"(s)); // keeps the division and the square root inside the rare branch
        if (!e_fast)
            ok = num / l.nrm < P.epi_thr;
        if (ok && !d_fast)
            ok = sqrt(s) <= P.max_disp;
    }
    return ok;
}

// Conservative bounding box of one left edge's search region {band} ∩ {disparity square}.
__device__ inline Box region_box(const LeftCtx &l, double D, double band, int mask)
{
    const double inf = __builtin_inf();
    Box r;
    r.x0 = -inf; r.x1 = inf; r.y0 = -inf; r.y1 = inf;
    if (mask & EBVO_STAGE_DISPARITY)
    {
        r.x0 = l.lx - D; r.x1 = l.lx + D; r.y0 = l.ly - D; r.y1 = l.ly + D;
    }
    if ((mask & EBVO_STAGE_EPIPOLAR) && (mask & EBVO_STAGE_DISPARITY) && l.ah == l.ah && l.bh == l.bh && l.ch == l.ch)
    {
        // |ah x + bh y + ch| <= band inside the square: bound y from the x-range, then x from the y-range
        const double m = 1e-9; // below this slope the other coordinate is unconstrained
        if (fabs(l.bh) > m)
        {
            const double ya = -(l.ah * r.x0 + l.ch) / l.bh, yb = -(l.ah * r.x1 + l.ch) / l.bh;
            const double e = band / fabs(l.bh) + BOX_SLACK;
            r.y0 = fmax(r.y0, fmin(ya, yb) - e);
            r.y1 = fmin(r.y1, fmax(ya, yb) + e);
        }
        if (fabs(l.ah) > m)
        {
            const double xa = -(l.bh * r.y0 + l.ch) / l.ah, xb = -(l.bh * r.y1 + l.ch) / l.ah;
            const double e = band / fabs(l.ah) + BOX_SLACK;
            r.x0 = fmax(r.x0, fmin(xa, xb) - e);
            r.x1 = fmin(r.x1, fmax(xa, xb) + e);
        }
    }
    return r;
}

// Candidate search.  One block = one tile of TILE (64) consecutive left edges; the right edges are visited through
// two levels of index-range bounding boxes (chunks of 16 edges, groups of 64 chunks):
//   1. wave 0 builds the 64 left-edge contexts (LDS) and the union of their search regions;
//   2. all 256 threads select the groups, then the chunks, whose boxes meet the union (ordered LDS compaction, so
//      chunks stay in ascending index), and stage the selected chunks' edges in LDS, 64 chunks at a time;
//   3. walk: sixteen lanes per left edge (four rounds of sixteen left edges).  The sixteen lanes test the staged
//      chunk boxes against their edge's own region (four ballots -> a 64-bit chunk mask), then take the marked
//      chunks in ascending order, ONE pair test per lane per chunk; a ballot gives the 16-bit hit mask, hence every
//      hit's rank in the row.  ~2 M threads instead of one per left edge: the walk used to be a serial chain of
//      ~120 dependent LDS reads + tests per lane with two waves per SIMD to hide it behind.
// The counting pass also keeps the first STAGE candidates of every row; after the scan of the counts a copy kernel
// completes every row that fits, and the FILL pass only redoes tiles that hold a longer row.
template <bool FILL>
__global__ __launch_bounds__(256) void candidates_kernel(const ebvo_edge *__restrict__ L,
                                                         const ebvo_edge *__restrict__ R,
                                                         const double *__restrict__ lines,
                                                         const Box *__restrict__ cb, const Box *__restrict__ gb,
                                                         CandParams P, int32_t *__restrict__ cnt,
                                                         const int32_t *__restrict__ row_ptr,
                                                         int32_t *__restrict__ col_idx,
                                                         unsigned long long *__restrict__ total_part)
{
    __shared__ double s_x[BATCH * CHUNK], s_y[BATCH * CHUNK], s_th[BATCH * CHUNK];
    __shared__ Box s_box[BATCH];
    __shared__ LeftCtx s_left[TILE];
    __shared__ int s_round[256];   // chunks selected in the current round, ascending
    __shared__ int s_groups[256];  // groups selected in the current group round, ascending
    __shared__ int s_wcnt[4];
    __shared__ double s_red[4];
    __shared__ unsigned long long s_tot[4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // CHUNK lanes per left edge: group of the block, lane in group, bit offset of the group in a ballot
    constexpr int LPE = CHUNK, GPB = 256 / LPE, ROUNDS = TILE / GPB;
    constexpr unsigned long long GMASK = (1ull << LPE) - 1ull;
    static_assert(TILE % GPB == 0 && BATCH % LPE == 0 && BATCH <= 64, "tile rounds / marked-chunk mask");
    const int gid = tid / LPE, e = tid % LPE, gshift = lane & ~(LPE - 1);
    const int nL = devn(P.nL), nR = devn(P.nR);
    const int nchunks = (nR + CHUNK - 1) / CHUNK, ngroups = (nchunks + GROUP - 1) / GROUP;
    const double d2 = P.max_disp * P.max_disp;
    const double D = P.max_disp + BOX_SLACK, band = P.epi_thr + BOX_SLACK;
    const double inf = __builtin_inf();
    unsigned long long blk_total = 0; // thread 0: candidates counted by this block

    // ordered (index-preserving) compaction of one value per selected thread into an LDS list
    auto compact = [&](bool sel, int value, int *list) -> int {
        const unsigned long long m = __ballot(sel);
        if (lane == 0)
            s_wcnt[wid] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            if (k < wid) pre += s_wcnt[k];
            tot += s_wcnt[k];
        }
        if (sel)
            list[pre + __popcll(m & ((1ull << lane) - 1ull))] = value;
        __syncthreads();
        return tot;
    };

    if (!FILL && blockIdx.x == 0 && tid == 0)
        cnt[nL] = 0; // the scan covers nL + 1 counts: row_ptr[nL] = total
    if (FILL)
    {
        // rows with at most STAGE candidates are completed from the staging area (round 4: here, as the prologue of the fill
        // pass, instead of in a launch of their own; the tiles below only redo rows that are longer -- disjoint writes)
        const int e16 = tid & 15;
        const int rows = (gridDim.x * 256) >> 4;
        for (int i = (blockIdx.x * 256 + tid) >> 4; i < nL; i += rows)
        {
            const int64_t o = row_ptr[i];
            const int n = row_ptr[i + 1] - row_ptr[i];
            if (n > STAGE)
                continue;
            for (int k = e16; k < n; k += 16)
                if (o + k < P.cap)
                    col_idx[o + k] = P.stage[(size_t)i * STAGE + k];
        }
    }
    // grid-stride over the tiles so the launch does not depend on nL
    for (int tile = blockIdx.x; tile * TILE < nL; tile += gridDim.x)
    {
        if (FILL && !P.tile_flag[tile]) // every row of this tile was completed from the staging area
            continue;
        // ---- 1. left contexts and the union of the tile's search regions (wave 0)
        if (wid == 0)
        {
            const int i = tile * TILE + lane;
            const bool live = i < nL;
            const int il = live ? i : 0;
            LeftCtx l;
            l.lx = L[il].x; l.ly = L[il].y; l.lth = L[il].theta;
            l.a = lines[(size_t)il * 3]; l.b = lines[(size_t)il * 3 + 1]; l.c = lines[(size_t)il * 3 + 2];
            l.nrm = sqrt((l.a * l.a) + (l.b * l.b));
            l.ah = l.a / l.nrm; l.bh = l.b / l.nrm; l.ch = l.c / l.nrm;
            const double t = P.epi_thr * l.nrm;
            l.t_lo = t * (1.0 - 0x1p-50); l.t_hi = t * (1.0 + 0x1p-50);
            l.s_lo = d2 * (1.0 - 0x1p-50); l.s_hi = d2 * (1.0 + 0x1p-50);
            Box rb = region_box(l, D, band, P.mask);
            if (!live) { rb.x0 = inf; rb.x1 = -inf; rb.y0 = inf; rb.y1 = -inf; }
            s_left[lane] = l;
            const double a0 = wave_min(rb.x0), a1 = wave_max(rb.x1), b0 = wave_min(rb.y0), b1 = wave_max(rb.y1);
            if (lane == 0) { s_red[0] = a0; s_red[1] = a1; s_red[2] = b0; s_red[3] = b1; }
        }
        __syncthreads();
        Box U;
        U.x0 = s_red[0]; U.x1 = s_red[1]; U.y0 = s_red[2]; U.y1 = s_red[3];
        auto meets_union = [&](const Box &bx) {
            return !(bx.x0 > U.x1 || bx.x1 < U.x0 || bx.y0 > U.y1 || bx.y1 < U.y0);
        };

        // the left edges of this group of CHUNK lanes (one per walk round): hit counts and output cursors
        int n[ROUNDS];
        int64_t o[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
        {
            n[r] = 0;
            o[r] = 0;
        }
        if (FILL)
        {
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)
            {
                const int i = tile * TILE + r * GPB + gid;
                o[r] = (int64_t)row_ptr[i < nL ? i : 0];
            }
        }
        for (int g0 = 0; g0 < ngroups; g0 += 256)
        {
            const int g = g0 + tid;
            const int ngsel = compact(g < ngroups && meets_union(gb[g < ngroups ? g : 0]), g, s_groups);
            for (int q = 0; q < ngsel; q += 4)
            {
                // ---- 2. chunks of up to four selected groups: one chunk per thread
                const int gq = q + (tid >> 6);
                const int c = (gq < ngsel) ? s_groups[gq] * GROUP + (tid & 63) : nchunks;
                const int nsel = compact(c < nchunks && meets_union(cb[c < nchunks ? c : 0]), c, s_round);
                for (int b0 = 0; b0 < nsel; b0 += BATCH)
                {
                    const int mb = min(BATCH, nsel - b0);
                    for (int idx = tid; idx < mb * CHUNK; idx += 256)
                    {
                        const int k = s_round[b0 + idx / CHUNK] * CHUNK + (idx % CHUNK);
                        double x = __builtin_nan(""), y = x, th = x; // NaN fails every predicate
                        if (k < nR)
                        {
                            x = R[k].x; y = R[k].y; th = R[k].theta;
                        }
                        s_x[idx] = x; s_y[idx] = y; s_th[idx] = th;
                    }
                    if (tid < mb)
                        s_box[tid] = cb[s_round[b0 + tid]];
                    __syncthreads();
                    // ---- 3. walk
#pragma unroll
                    for (int r = 0; r < ROUNDS; ++r)
                    {
                        const int li = r * GPB + gid;
                        const int i = tile * TILE + li;
                        const bool live = i < nL; // uniform in the group
                        const LeftCtx l = s_left[li];
                        // which staged chunks can meet this edge's region: lane e looks at boxes e, e + LPE, e + 2 LPE, ...
                        unsigned long long pm = 0;
#pragma unroll
                        for (int k = 0; k < BATCH / LPE; ++k)
                        {
                            const int j = k * LPE + e;
                            const bool may = live && j < mb &&
                                             box_may_match(s_box[j < mb ? j : 0], l.lx, l.ly, l.ah, l.bh, l.ch, D, band,
                                                           P.mask);
                            const unsigned long long bal = __ballot(may);
                            pm |= ((bal >> gshift) & GMASK) << (LPE * k);
                        }
                        // marked chunks in ascending order, one pair test per lane
                        while (__any(pm != 0))
                        {
                            const bool act = pm != 0;
                            const int j = act ? __ffsll((long long)pm) - 1 : 0;
                            pm &= pm - 1; // 0 stays 0
                            const int idx = j * CHUNK + e;
                            const bool ok = act && pair_passes(l, s_x[idx], s_y[idx], s_th[idx], P);
                            const unsigned hits = (unsigned)((__ballot(ok) >> gshift) & GMASK);
                            if (ok)
                            {
                                const int rank = __popc(hits & ((1u << e) - 1u));
                                const int k = s_round[b0 + j] * CHUNK + e;
                                if (FILL)
                                {
                                    if (o[r] + rank < P.cap)
                                        col_idx[o[r] + rank] = k;
                                }
                                else if (n[r] + rank < STAGE)
                                    P.stage[(size_t)i * STAGE + n[r] + rank] = k;
                            }
                            const int nh = __popc(hits);
                            n[r] += nh;
                            o[r] += nh;
                        }
                    }
                    __syncthreads();
                }
            }
        }
        if (!FILL)
        {
            unsigned long long tsum = 0;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r)
            {
                const int i = tile * TILE + r * GPB + gid;
                if (e == 0 && i < nL)
                {
                    cnt[i] = n[r];
                    if (n[r] > STAGE)
                        P.tile_flag[tile] = 1; // benign race: every writer stores 1
                    tsum += (unsigned long long)n[r];
                }
            }
            // 64-bit total (guards the int32 CSR offsets and the buffer capacity): per-block partial, no atomics
            for (int d = 32; d > 0; d >>= 1)
                tsum += __shfl_down(tsum, d);
            if (lane == 0)
                s_tot[wid] = tsum;
            __syncthreads();
            if (tid == 0)
                blk_total += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        }
        __syncthreads();
    }
    if (!FILL && tid == 0)
        total_part[blockIdx.x] = blk_total;
}

// sum of the per-block candidate totals (host-buffer path; the pipeline sums them in pair_result_kernel)
__global__ void total_sum_kernel(const unsigned long long *__restrict__ part, int n, unsigned long long *__restrict__ total)
{
    unsigned long long v = 0;
    for (int k = threadIdx.x; k < n; k += 64)
        v += part[k];
    for (int d = 32; d > 0; d >>= 1)
        v += __shfl_down(v, d);
    if (threadIdx.x == 0)
        *total = v;
}

// rows with at most STAGE candidates are completed from the staging area (the fill pass only redoes the rest)
__global__ void candidates_copy_kernel(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ stage, DevN nLd,
                                       int64_t cap, int32_t *__restrict__ col_idx)
{
    const int nL = devn(nLd);
    const int e = threadIdx.x & 15; // sixteen lanes per row: coalesced within the row
    const int rows = (gridDim.x * blockDim.x) >> 4;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < nL; i += rows)
    {
        const int64_t o = row_ptr[i];
        const int n = row_ptr[i + 1] - row_ptr[i];
        if (n > STAGE)
            continue;
        for (int k = e; k < n; k += 16)
            if (o + k < cap)
                col_idx[o + k] = stage[(size_t)i * STAGE + k];
    }
}

// ------------------------------------------------------------------------------------------
// one launch zeroes every small counter array a stage needs (instead of one memset node per array)
struct ClearList
{
    int32_t *p[EBVO_CLEAR_MAX];
    int n[EBVO_CLEAR_MAX];
};
__global__ void clear_kernel(ClearList C)
{
    int32_t *__restrict__ p = C.p[blockIdx.y];
    const int n = C.n[blockIdx.y];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        p[k] = 0;
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of int32, up to four independent arrays per launch, two kernels whatever the length:
//   scan_reduce_kernel  sum of every 4096-element tile
//   scan_apply_kernel   each block adds up the sums of the tiles before it (a few hundred at most), then scans its
//                       own tile from that offset
// Lengths may live on the device (DevN); the grid is sized by the host-side bound and idle blocks leave at once.
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;
constexpr int SCAN_BATCH = 4;
constexpr int SCAN_DIRECT_TILES = 16; // up to 65,536 elements the scan is ONE launch (scan_apply_kernel, ScanBatch::direct)

struct ScanBatch
{
    const int32_t *in[SCAN_BATCH];
    int32_t *out[SCAN_BATCH];
    DevN n[SCAN_BATCH];
    int n_add;      // scan n + n_add elements; the extra element counts as zero whatever the memory holds: out[n] = total
    int32_t *total[SCAN_BATCH]; // optional (n_add == 1): the total is stored here as well
    int32_t *sums;  // [SCAN_BATCH][tiles]
    int tiles;
    int direct;     // few tiles: every block adds up the elements before its tile itself (one launch instead of two)
};

__device__ inline int32_t block_sum_256(int32_t v, int32_t *wsum)
{
    for (int d = 32; d > 0; d >>= 1)
        v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    return wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(ScanBatch S)
{
    __shared__ int32_t wsum[SCAN_BLOCK / 64];
    const int a = blockIdx.y;
    const int n_in = devn(S.n[a]), n = n_in + S.n_add;
    const int base = blockIdx.x * SCAN_TILE;
    if (base >= n)
        return;
    const int32_t *__restrict__ in = S.in[a];
    int32_t v = 0;
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t)
    {
        const int k = base + t * SCAN_BLOCK + threadIdx.x; // coalesced; the order inside a tile does not matter here
        v += (k < n_in) ? in[k] : 0;
    }
    const int32_t tot = block_sum_256(v, wsum);
    if (threadIdx.x == 0)
        S.sums[(size_t)a * S.tiles + blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(ScanBatch S)
{
    __shared__ int32_t wsum[SCAN_BLOCK / 64], wpre[SCAN_BLOCK / 64];
    const int a = blockIdx.y;
    const int n_in = devn(S.n[a]), n = n_in + S.n_add;
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    if (blockIdx.x * SCAN_TILE >= n)
        return;
    const int32_t *__restrict__ in = S.in[a];
    int32_t *__restrict__ out = S.out[a];
    // offset of this tile = sum of the tiles before it: from the reduce kernel's per-tile sums, or -- up to SCAN_DIRECT_TILES
    // tiles, i.e. at most 60 sixteen-byte loads per thread -- straight from the elements (the launch it saves costs more)
    int32_t o = 0;
    if (S.direct)
    {
        const int before = min((int)blockIdx.x * SCAN_TILE, n_in);
        if ((reinterpret_cast<uintptr_t>(in) & 15) == 0)
        {
            const int4 *p4 = reinterpret_cast<const int4 *>(in);
            for (int k = threadIdx.x; k < (before >> 2); k += SCAN_BLOCK)
            {
                const int4 q = p4[k];
                o += (q.x + q.y) + (q.z + q.w);
            }
            for (int k = (before & ~3) + threadIdx.x; k < before; k += SCAN_BLOCK)
                o += in[k];
        }
        else
            for (int k = threadIdx.x; k < before; k += SCAN_BLOCK)
                o += in[k];
    }
    else
        for (int k = threadIdx.x; k < (int)blockIdx.x; k += SCAN_BLOCK)
            o += S.sums[(size_t)a * S.tiles + k];
    const int32_t tile_off = block_sum_256(o, wpre);
    int32_t v[SCAN_ITEMS];
    int32_t s = 0;
    if (base + SCAN_ITEMS <= n_in && (reinterpret_cast<uintptr_t>(in + base) & 15) == 0)
    {
        const int4 *p = reinterpret_cast<const int4 *>(in + base);
#pragma unroll
        for (int t = 0; t < SCAN_ITEMS / 4; ++t)
        {
            const int4 q = p[t];
            v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
        }
    }
    else
    {
#pragma unroll
        for (int t = 0; t < SCAN_ITEMS; ++t)
            v[t] = (base + t < n_in) ? in[base + t] : 0;
    }
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t)
        s += v[t];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int32_t incl = s;
    for (int d = 1; d < 64; d <<= 1)
    {
        const int32_t t = __shfl_up(incl, d);
        if (lane >= d)
            incl += t;
    }
    if (lane == 63)
        wsum[wid] = incl;
    __syncthreads();
    int32_t pre = 0;
#pragma unroll
    for (int k = 0; k < SCAN_BLOCK / 64; ++k)
        if (k < wid)
            pre += wsum[k];
    int32_t run = tile_off + pre + incl - s;
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t)
    {
        if (base + t < n)
            out[base + t] = run;
        if (S.n_add == 1 && base + t == n_in && S.total[a])
            *S.total[a] = run; // the total, for callers that keep it apart from the offsets
        run += v[t];
    }
}

__global__ void gather_edges_kernel(const ebvo_edge *__restrict__ R, const int32_t *__restrict__ idx, int64_t n,
                                    ebvo_edge *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n)
        out[k] = R[idx[k]];
}

// ------------------------------------------------------------------------------------------
// Bilinear_Interpolation<double> on the u8 image (include/utility.h:81-104): NaN when a corner
// is outside the image or a coordinate is an exact integer (0/0).
// The reference divides each weight by (x2 - x1) resp. (y2 - y1).  For a non-integer coordinate
// those are exactly +1.0 and -1.0, and x / 1.0 == x, x / -1.0 == -x bit for bit; for an integer
// coordinate they are 0 and the quotient 0/0 is NaN.  The four fp64 divisions are therefore
// replaced by a select -- same bits, ~140 fewer instructions per sample.
// Straight-line form (no early return): the four byte loads of every sample are issued unconditionally from a
// clamped address, so the 28 loads of a patch row overlap; the NaN cases are applied by a final select.
__device__ inline double bilinear_nan(const uint8_t *__restrict__ img, int rows, int cols, int pitch, double x,
                                      double y)
{
    const double x1 = floor(x), x2 = ceil(x);
    const double yc = ceil(y), yf = floor(y);
    // include/utility.h:95-99 (a corner outside the image); NaN coordinates fail every comparison the same way
    const bool inside = (x1 >= 0) && (yf >= 0) && (x2 < cols) && (yc < rows);
    const int c1 = inside ? (int)x1 : 0;
    const int r1 = inside ? (int)yc * pitch : 0, r2 = inside ? (int)yf * pitch : 0;
    // the two corners of a row are adjacent bytes (x2 = x1 + 1 unless x is an integer, and then the result is NaN
    // whatever is loaded): one 2-byte load per row instead of two byte loads -- the sampling is address-divergent, so
    // the texture path charges per load instruction.  The image buffer has readable bytes past its last pixel.
    unsigned short p1, p2;
    __builtin_memcpy(&p1, img + r1 + c1, 2);
    __builtin_memcpy(&p2, img + r2 + c1, 2);
    const double I11 = (double)(p1 & 0xff);
    const double I21 = (double)(p1 >> 8);
    const double I12 = (double)(p2 & 0xff);
    const double I22 = (double)(p2 >> 8);
    const double wxa = x2 - x;    // (Q21.x - P.x) / (Q21.x - Q11.x), denominator exactly 1
    const double wxb = x - x1;    // (P.x - Q11.x) / (Q21.x - Q11.x)
    const double wya = -(yf - y); // (Q12.y - P.y) / (Q12.y - Q11.y), denominator exactly -1
    const double wyb = -(y - yc); // (P.y - Q11.y) / (Q12.y - Q11.y)
    const double f1 = wxa * I11 + wxb * I21;
    const double f2 = wxa * I12 + wxb * I22;
    const double v = wya * f1 + wyb * f2;
    // integer coordinate: the reference divides 0 by 0 (include/utility.h:101-103) -> NaN
    const bool ok = inside && (x2 != x1) && (yf != yc);
    return ok ? v : __builtin_nan("");
}

// 8-lane butterfly: ((s0+s1)+(s2+s3)) + ((s4+s5)+(s6+s7)) on every lane of the group.  The partner's value arrives
// through DPP (quad_perm for the xor-1 and xor-2 steps; row_half_mirror for the last one: lane i of an 8-lane group reads
// lane 7 - i, which after two steps holds the sum of the OTHER quad -- IEEE addition is commutative, so A + B and B + A are
// the same bits): register-to-register moves on the vector pipe instead of ds_bpermute round trips through the LDS.
template <int CTRL>
__device__ inline double dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(phi, plo);
}

__device__ inline double butterfly8(double s)
{
    s += dpp_f64<0xB1>(s);  // quad_perm:[1,0,3,2]
    s += dpp_f64<0x4E>(s);  // quad_perm:[2,3,0,1]
    s += dpp_f64<0x141>(s); // row_half_mirror
    return s;
}

// sin / cos of every edge orientation, one thread per edge (dense: the double-double routine costs
// ~700 instructions per wave, so it is evaluated once per edge here instead of once per 16-lane
// group inside the sampling kernels).  src/utility.cpp:84-87,151 call std::sin / std::cos.
// Up to two edge lists (left / right image of a pair) per launch: blockIdx.y selects the list.
struct PatchBatch
{
    const uint8_t *img[2];
    const ebvo_edge *edges[2];
    DevN n[2];
    double2 *sc[2];
    float *raw[2], *norm[2];
    uint8_t *flag[2];
};

__device__ inline void sincos_body(const ebvo_edge *__restrict__ e, int n, double2 *__restrict__ sc, int vb, int vg)
{
    for (int k = vb * blockDim.x + threadIdx.x; k < n; k += vg * blockDim.x)
    {
        double sn, cs;
        ebvo_sincos(e[k].theta, &sn, &cs);
        double2 v;
        v.x = sn;
        v.y = cs;
        sc[k] = v;
    }
}

__global__ void sincos_batch_kernel(PatchBatch B)
{
    sincos_body(B.edges[blockIdx.y], devn(B.n[blockIdx.y]), B.sc[blockIdx.y], blockIdx.x, gridDim.x);
}

__global__ void sincos_edges_kernel(const ebvo_edge *__restrict__ e, DevN nd, double2 *__restrict__ sc)
{
    const int n = devn(nd);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
    {
        double sn, cs;
        ebvo_sincos(e[k].theta, &sn, &cs);
        double2 v;
        v.x = sn;
        v.y = cs;
        sc[k] = v;
    }
}

// The same interpolation for a sample that is known to lie inside the image with both of its corner rows / columns (see
// patch_inside): no corner tests, no address clamps, a 32-bit offset from the (wave-uniform) image base, one 24-bit
// multiply-add for the row offset.  Bit-identical to bilinear_nan wherever both apply: the loaded bytes and every
// floating-point operation are the same; only the integer-coordinate NaN rule remains to be selected.
//
// Round 4: the weights come from v_fract_f64 instead of floor / ceil and four subtractions.  For a positive coordinate x that
// is not an integer, x - floor(x) is exact (a multiple of ulp(x) below 1), and ceil(x) - x = 1 - (x - floor(x)) is exact
// too (a multiple of ulp(x) in (0, 1]), so
//     wxb = x - x1 = fract(x),   wxa = x2 - x = 1 - fract(x),   wya = -(yf - y) = fract(y),   wyb = -(y - yc) = 1 - fract(y)
// are the reference's weights bit for bit (include/utility.h:101-103) at three instructions per axis instead of five, and
// the integer-coordinate rule (x2 == x1 or yf == yc: 0/0 -> NaN) is fract == 0.  `ok` reports that rule; the callers select
// NaN per sample, or (sample_row) once per row when some lane met it.
__device__ inline double bilinear_inside_raw(const uint8_t *__restrict__ img, int pitch, double x, double y, bool &ok)
{
    const double wxb = __builtin_amdgcn_fract(x), wya = __builtin_amdgcn_fract(y);
    const double wxa = 1.0 - wxb, wyb = 1.0 - wya;
    const unsigned of = __umul24((unsigned)(int)y, (unsigned)pitch) + (unsigned)(int)x; // row floor(y): (int) truncates, x, y > 0
    const unsigned oc = of + (unsigned)pitch; // row ceil(y) (= floor(y) + 1 unless y is an integer: NaN then, whatever is read)
    unsigned short p1, p2;
    __builtin_memcpy(&p1, img + oc, 2);
    __builtin_memcpy(&p2, img + of, 2);
    const double I11 = (double)(p1 & 0xff);
    const double I21 = (double)(p1 >> 8);
    const double I12 = (double)(p2 & 0xff);
    const double I22 = (double)(p2 >> 8);
    const double f1 = wxa * I11 + wxb * I21;
    const double f2 = wxa * I12 + wxb * I22;
    ok = (wxb != 0.0) && (wya != 0.0); // include/utility.h:101-103: 0/0 for an integer coordinate
    return wya * f1 + wyb * f2;
}

__device__ inline double bilinear_inside(const uint8_t *__restrict__ img, int pitch, double x, double y)
{
    bool ok;
    const double v = bilinear_inside_raw(img, pitch, x, y, ok);
    return ok ? v : __builtin_nan("");
}

// Row-pair image: pix2[y * pitch + x] = img(y, x) | img(y + 1, x) << 8.  The four corners of a sample are then FOUR
// CONSECUTIVE BYTES at offset 2 * (floor(y) * pitch + floor(x)): one 4-byte load per sample instead of two 2-byte loads.
// The sampling is address-divergent (every lane its own cache line): the texture addresser retires about one lane
// address per cycle and CU, which is what bounds the sampling kernels (GRBM_TA_BUSY, profiles/), so halving the load
// count halves that bound.  Same bytes, same arithmetic as bilinear_inside.
__device__ inline double bilinear_inside2_raw(const uint16_t *__restrict__ pix2, int pitch, double x, double y, bool &ok)
{
    const double wxb = __builtin_amdgcn_fract(x), wya = __builtin_amdgcn_fract(y); // see bilinear_inside_raw
    const double wxa = 1.0 - wxb, wyb = 1.0 - wya;
    const unsigned of = __umul24((unsigned)(int)y, (unsigned)pitch) + (unsigned)(int)x;
    unsigned q;
    __builtin_memcpy(&q, reinterpret_cast<const uint8_t *>(pix2) + 2u * of, 4);
    const double I12 = (double)(q & 0xffu);         // (floor y, x1)
    const double I11 = (double)((q >> 8) & 0xffu);  // (ceil y,  x1)
    const double I22 = (double)((q >> 16) & 0xffu); // (floor y, x2)
    const double I21 = (double)(q >> 24);           // (ceil y,  x2)
    const double f1 = wxa * I11 + wxb * I21;
    const double f2 = wxa * I12 + wxb * I22;
    ok = (wxb != 0.0) && (wya != 0.0);
    return wya * f1 + wyb * f2;
}

__device__ inline double bilinear_inside2(const uint16_t *__restrict__ pix2, int pitch, double x, double y)
{
    bool ok;
    const double v = bilinear_inside2_raw(pix2, pitch, x, y, ok);
    return ok ? v : __builtin_nan("");
}

__device__ inline void row_pairs_body(const uint8_t *__restrict__ img, uint16_t *__restrict__ out, int h, int w, int vb, int vg)
{
    const int n = h * w;
    // four pixels per thread; the image buffers have readable padding behind the last pixel (ebvo_capi.hip: img_base)
    for (int o = (vb * blockDim.x + threadIdx.x) * 4; o < n; o += vg * blockDim.x * 4)
    {
        unsigned a, b = 0;
        __builtin_memcpy(&a, img + o, 4);
        if (o + w + 3 < n)
            __builtin_memcpy(&b, img + o + w, 4);
        else
        {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (o + w + t < n)
                    b |= (unsigned)img[o + w + t] << (8 * t);
        }
        uint2 r;
        r.x = (a & 0xffu) | ((b & 0xffu) << 8) | ((a & 0xff00u) << 8) | ((b & 0xff00u) << 16);
        r.y = ((a >> 16) & 0xffu) | (((b >> 16) & 0xffu) << 8) | ((a >> 24) << 16) | ((b >> 24) << 24);
        if (o + 3 < n)
            __builtin_memcpy(out + o, &r, 8);
        else
            for (int t = 0; t < 4 && o + t < n; ++t)
                out[o + t] = (uint16_t)(t < 2 ? (r.x >> (16 * t)) : (r.y >> (16 * (t - 2))));
    }
}

__global__ __launch_bounds__(256) void row_pairs_kernel(const uint8_t *__restrict__ img0, const uint8_t *__restrict__ img1,
                                                        uint16_t *__restrict__ out0, uint16_t *__restrict__ out1, int h,
                                                        int w)
{
    row_pairs_body(blockIdx.y ? img1 : img0, blockIdx.y ? out1 : out0, h, w, blockIdx.x, gridDim.x);
}

// The four small preparations of the matching half of a resident pair as block ranges of ONE launch (round 4; each was a
// launch of ~5 us behind a ~1 us body): epipolar lines of the left edges, chunk / group boxes of the right edges (+ the tile
// flags of the counting pass), sin / cos of both edge lists, the row-pair images of both NCC images.  Nothing in one range
// depends on another.
struct PrepArgs
{
    const double *F;
    const ebvo_edge *L, *R;
    DevN nL, nR;
    double *lines;
    Box *cb, *gb;
    int32_t *tile_flag;
    int ntiles;
    double2 *scL, *scR;
    const uint8_t *img0, *img1;
    uint16_t *pix0, *pix1;
    int h, w;
    int b_lines, b_boxes, b_sincos, b_rows; // blocks per range (sincos and rows: per image)
};

__global__ __launch_bounds__(256) void match_prep_kernel(PrepArgs A)
{
    int b = blockIdx.x;
    if (b < A.b_lines)
    {
        lines_body(A.F, A.L, devn(A.nL), A.lines, b, A.b_lines);
        return;
    }
    b -= A.b_lines;
    if (b < A.b_boxes)
    {
        boxes_body(A.R, devn(A.nR), A.cb, A.gb, A.tile_flag, A.ntiles, b, A.b_boxes);
        return;
    }
    b -= A.b_boxes;
    if (b < 2 * A.b_sincos)
    {
        const int im = b >= A.b_sincos;
        sincos_body(im ? A.R : A.L, devn(im ? A.nR : A.nL), im ? A.scR : A.scL, b - im * A.b_sincos, A.b_sincos);
        return;
    }
    b -= 2 * A.b_sincos;
    const int im = b >= A.b_rows;
    row_pairs_body(im ? A.img1 : A.img0, im ? A.pix1 : A.pix0, A.h, A.w, b - im * A.b_rows, A.b_rows);
}

// True if every sample of both patches of an edge at (ex, ey) has its four corners inside the image: a sample lies within
// 5 + 3 sqrt(2) = 9.243 px of the edge in each coordinate (|sin|, |cos| <= 1), and its corners within one more pixel on
// the ceil side.  The small slack covers the rounding of the coordinate arithmetic.  (A kept TOED edge has
// 10 < x < W - 10: nearly always true, but not on the last quarter pixel, hence a test and not an assumption.)
__device__ inline bool patch_inside(int h, int w, double ex, double ey)
{
    return ex >= 9.3 && ey >= 9.3 && ex <= (double)w - 10.3 && ey <= (double)h - 10.3;
}

struct Row7
{
    float v[7];
};

// the general interpolation (corner tests, NaN outside the image) for the rare wave that holds an edge next to the image
// border: out of line, so that its registers and code do not weigh on the common path
__device__ __noinline__ Row7 sample_row_border(const uint8_t *__restrict__ img, int h, int w, int pitch, double cx, double cy,
                                               double sn, double cs, int i)
{
    Row7 r;
#pragma unroll
    for (int c = 0; c < 7; ++c)
    {
        const int j = c - 3;
        const double x = cs * (i)-sn * (j) + cx;
        const double y = sn * (i) + cs * (j) + cy;
        r.v[c] = (float)bilinear_nan(img, h, w, pitch, x, y);
    }
    return r;
}

// One lane's row (7 samples) of one side of an edge's patch pair.
// src/utility.cpp:82-93 (centres), :141-161 (grid), :206-209 (to float).
// `active` lanes sample; the fast interpolation is taken when it applies to every sampling lane of the wave.
template <bool PIX2 = false>
__device__ inline void sample_row(const uint8_t *__restrict__ img, int h, int w, int pitch, double ex, double ey,
                                  double sn, double cs, int side, int row, float p[7], bool active = true,
                                  const uint16_t *__restrict__ pix2 = nullptr /* PIX2: the row-pair image of img */)
{
    const double cx = side ? ex + 5 * (-sn) : ex + 5 * (sn);
    const double cy = side ? ey + 5 * (cs) : ey + 5 * (-cs);
    const int i = row - 3;
    if (__all(!active || patch_inside(h, w, ex, ey)))
    {
        // the NaN rule of an integer coordinate (measure zero) is applied once per row, and only when some lane met it: no
        // per-sample selects on the common path
        bool all_ok = true;
        if (active)
        {
#pragma unroll
            for (int c = 0; c < 7; ++c)
            {
                const int j = c - 3;
                const double x = cs * (i)-sn * (j) + cx;
                const double y = sn * (i) + cs * (j) + cy;
                bool ok;
                p[c] = (float)(PIX2 ? bilinear_inside2_raw(pix2, pitch, x, y, ok) : bilinear_inside_raw(img, pitch, x, y, ok));
                all_ok = all_ok && ok;
            }
        }
        if (__any(!all_ok))
        {
            if (active)
            {
#pragma unroll
                for (int c = 0; c < 7; ++c)
                {
                    const int j = c - 3;
                    const double x = cs * (i)-sn * (j) + cx;
                    const double y = sn * (i) + cs * (j) + cy;
                    if (__builtin_amdgcn_fract(x) == 0.0 || __builtin_amdgcn_fract(y) == 0.0)
                        p[c] = __builtin_nanf("");
                }
            }
        }
        return;
    }
    if (active)
    {
        const Row7 r = sample_row_border(img, h, w, pitch, cx, cy, sn, cs, i);
#pragma unroll
        for (int c = 0; c < 7; ++c)
            p[c] = r.v[c];
    }
}

// mean-centre, sum of squares and normalisation of a patch spread over an 8-lane group
// (src/utility.cpp:165-168, :174-178).  Returns the sentinel flag (ss < 1e-10).
__device__ inline bool normalise_rows(bool active, const float p[7], float nrm[7])
{
    double rs = 0.0;
    if (active)
    {
        rs = (double)p[0];
#pragma unroll
        for (int c = 1; c < 7; ++c)
            rs += (double)p[c];
    }
    const double mean = butterfly8(rs) / 49.0;
    const float m = (float)mean;
    float d[7];
    double qs = 0.0;
    if (active)
    {
#pragma unroll
        for (int c = 0; c < 7; ++c)
        {
            d[c] = p[c] - m;
            const float q = d[c] * d[c];
            qs = (c == 0) ? (double)q : qs + (double)q;
        }
    }
    const double ss = butterfly8(qs);
    const float inv = (float)(1.0 / sqrt(ss));
#pragma unroll
    for (int c = 0; c < 7; ++c)
        nrm[c] = active ? d[c] * inv : 0.0f;
    return ss < 1e-10;
}

// One lane's seven terms of a 49-term dot product, accumulated left to right in double.  The product of two floats is
// exact in double (24 + 24 significand bits), so fma(a, b, s) rounds exactly what s + a * b rounds: the explicit fma
// below returns the bits of the reference's separate multiply and add at half the instructions.
__device__ inline double dot7(const float a[7], const float b[7])
{
    double s = (double)a[0] * (double)b[0];
#pragma unroll
    for (int c = 1; c < 7; ++c)
        s = __builtin_fma((double)a[c], (double)b[c], s);
    return s;
}

__device__ inline double dot_rows(bool active, const float a[7], const float b[7])
{
    return butterfly8(active ? dot7(a, b) : 0.0);
}

__device__ inline double max4(double a, double b, double c, double d)
{
    double m = a; // std::max({a,b,c,d}): first maximum under operator<
    if (m < b) m = b;
    if (m < c) m = c;
    if (m < d) m = d;
    return m;
}

// Patches of n edges: raw floats (n x 2 x 49), optionally the normalised patches and sentinel flags.
// 16 lanes per edge; every 16-lane group strides over the edges with a wave-uniform trip count.
__global__ __launch_bounds__(256) void patches_kernel(PatchBatch B, int h, int w, int pitch)
{
    const uint8_t *__restrict__ img = B.img[blockIdx.y];
    const ebvo_edge *__restrict__ edges = B.edges[blockIdx.y];
    const double2 *__restrict__ sc = B.sc[blockIdx.y];
    float *__restrict__ raw = B.raw[blockIdx.y], *__restrict__ norm = B.norm[blockIdx.y];
    uint8_t *__restrict__ flag = B.flag[blockIdx.y];
    const int n = devn(B.n[blockIdx.y]);
    const int groups = (gridDim.x * blockDim.x) >> 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t & 15, side = g >> 3, row = g & 7;
    const int iters = (n + groups - 1) / groups;
    for (int it = 0; it < iters; ++it)
    {
        const int e = it * groups + (t >> 4);
        const bool valid = e < n; // uniform per 16-lane group
        const bool active = valid && row < 7;
        float p[7], nr[7];
#pragma unroll
        for (int c = 0; c < 7; ++c)
            p[c] = 0.0f;
        {
            const int ec = active ? e : 0; // inactive lanes read edge 0 (always present when the kernel has work) and discard it
            sample_row(img, h, w, pitch, edges[ec].x, edges[ec].y, sc[ec].x, sc[ec].y, side, row, p, active);
        }
        const bool sent = normalise_rows(active, p, nr);
        if (active)
        {
            const size_t o = (size_t)e * 98 + side * 49 + row * 7;
            if (raw)
            {
#pragma unroll
                for (int c = 0; c < 7; ++c)
                    raw[o + c] = p[c];
            }
            if (norm)
            {
#pragma unroll
                for (int c = 0; c < 7; ++c)
                    norm[o + c] = nr[c];
            }
            if (flag && row == 0)
                flag[(size_t)e * 2 + side] = sent ? 1 : 0;
        }
    }
}

// NCC of (left edge i, candidate k) pairs; 16 lanes per pair.  src/Stereo_Matches.cpp:585-608.
__global__ __launch_bounds__(256) void ncc_pairs_kernel(const uint8_t *__restrict__ imgR, int h, int w,
                                                        int pitch, const ebvo_edge *__restrict__ Rc,
                                                        const double2 *__restrict__ sc,
                                                        const int32_t *__restrict__ pair_left, DevCount npd,
                                                        const float *__restrict__ left_norm,
                                                        const uint8_t *__restrict__ left_flag, double thr,
                                                        double *__restrict__ sims, double *__restrict__ best,
                                                        uint8_t *__restrict__ keep, int32_t *__restrict__ match_cnt)
{
    const int64_t n_pairs = devcount(npd);
    if ((int64_t)blockIdx.x * 16 >= n_pairs) // a block scores 16 pairs; the grid may be sized for an upper bound
        return;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = t >> 4;
    const int g = (int)(t & 15), side = g >> 3, row = g & 7;
    const bool valid = k < n_pairs;
    const bool active = valid && row < 7;
    float p[7], rn[7];
#pragma unroll
    for (int c = 0; c < 7; ++c)
        p[c] = 0.0f;
    int li = 0;
    if (valid)
        li = pair_left[k];
    {
        const int64_t kc = active ? k : 0;
        sample_row(imgR, h, w, pitch, Rc[kc].x, Rc[kc].y, sc[kc].x, sc[kc].y, side, row, p, active);
    }
    const bool rsent = normalise_rows(active, p, rn);
    // this lane's row of the left plus / minus normalised patches
    float lp[7], lm[7];
#pragma unroll
    for (int c = 0; c < 7; ++c)
        lp[c] = lm[c] = 0.0f;
    bool lsent_p = false, lsent_m = false;
    if (active)
    {
        const float *ln = left_norm + (size_t)li * 98 + row * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c)
        {
            lp[c] = ln[c];
            lm[c] = ln[49 + c];
        }
    }
    if (valid)
    {
        lsent_p = left_flag[(size_t)li * 2] != 0;
        lsent_m = left_flag[(size_t)li * 2 + 1] != 0;
    }
    // side 0 lanes hold R+: (L+ . R+) = pp, (L- . R+) = np;  side 1 lanes hold R-: (L+ . R-) = pn, (L- . R-) = nn
    const double d_lp = dot_rows(active, lp, rn);
    const double d_lm = dot_rows(active, lm, rn);
    const double o_lp = __shfl_xor(d_lp, 8), o_lm = __shfl_xor(d_lm, 8);
    const int rs_i = rsent ? 1 : 0;
    const int rs_o = __shfl_xor(rs_i, 8);
    bool is_match = false;
    if (valid && g == 0)
    {
        const bool rsent_p = rs_i != 0, rsent_m = rs_o != 0;
        const double pp = (lsent_p || rsent_p) ? -1.0 : d_lp; // src/utility.cpp:170-172
        const double np = (lsent_m || rsent_p) ? -1.0 : d_lm;
        const double pn = (lsent_p || rsent_m) ? -1.0 : o_lp;
        const double nn = (lsent_m || rsent_m) ? -1.0 : o_lm;
        const double b = max4(pp, nn, pn, np); // src/Stereo_Matches.cpp:596
        if (sims)
        {
            sims[k * 4 + 0] = pp;
            sims[k * 4 + 1] = nn;
            sims[k * 4 + 2] = pn;
            sims[k * 4 + 3] = np;
        }
        if (best)
            best[k] = b;
        is_match = b > thr; // :597
        if (keep)
            keep[k] = is_match ? 1 : 0;
    }
    (void)match_cnt;
    (void)is_match;
}

// pair -> left row index (CSR expansion), one thread per left edge
__global__ void expand_rows_kernel(const int32_t *__restrict__ row_ptr, DevN nLd, int32_t *__restrict__ pair_left,
                                   int64_t cap)
{
    const int nL = devn(nLd);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1] && k < cap; ++k)
            pair_left[k] = i;
}

// ---- NCC of the resident pipeline: left patches staged in LDS, right patches from a padded bank ----------------------
// Every right TOED edge is a candidate of ~5 left edges spread over two or three interpolated rows, so its two patches are
// sampled and normalised ONCE into a bank; every left edge's patches are needed by its own CSR row only, so they never
// leave the workgroup that samples them.
//   right bank : per edge 2 sides x 7 rows x 8 floats = 448 B.  A lane owns one 32-byte row: two aligned 16-byte
//                accesses instead of seven dwords (the gather is address-divergent: the texture path charges per load
//                instruction and per 128-byte line touched).  Slot [7] of every row carries the side's sentinel flag
//                (sum of squares < 1e-10, src/utility.cpp:170).
//   ncc_tile   : a WAVE owns NW consecutive left edges = a contiguous range of CSR pairs.  Phase 1 samples and
//                normalises their patches into LDS (same layout as a bank entry, 7 rows), phase 2 walks the pairs of the
//                tile, eight lanes per pair (lane r = row r of all four patches): four 16-byte loads of the right rows,
//                four ds_read_b128 of the left rows, four 49-term dots in the canonical order (dot7 per row, then the
//                tree of butterfly8 -- round 4: reduced transposed, every lane ending up with ONE of the four dots, and with
//                the tile's loads issued early; see the kernel).  Arithmetic identical to ncc_pairs_kernel / the oracle:
//                the normalised rows are the same floats, the reductions the same additions.
constexpr int BANK_SIDE = 56;   // floats per side of a bank entry: 7 rows x 8 floats (224 B, 16-byte aligned rows)
constexpr int BANK_EDGE = 112;  // floats per edge (448 B)
constexpr int NCC_NW = 4;       // left edges per wave (one sampling round: four 16-lane groups)
constexpr int NCC_WPE = 4;      // waves per SIMD the tile kernel is compiled for (the prefetched loads need the registers)

__global__ __launch_bounds__(256) void right_bank_kernel(const uint8_t *__restrict__ img,
                                                         const uint16_t *__restrict__ pix2, int h, int w, int pitch,
                                                         const ebvo_edge *__restrict__ edges,
                                                         const double2 *__restrict__ sc, DevN nd,
                                                         float *__restrict__ bank)
{
    const int n = devn(nd);
    const int groups = (gridDim.x * blockDim.x) >> 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = t & 15, side = g >> 3, row = g & 7;
    const int iters = (n + groups - 1) / groups;
    for (int it = 0; it < iters; ++it)
    {
        const int e = it * groups + (t >> 4);
        const bool active = e < n && row < 7;
        float p[7], nr[7];
#pragma unroll
        for (int c = 0; c < 7; ++c)
            p[c] = 0.0f;
        {
            const int ec = active ? e : 0; // inactive lanes read edge 0 (always present when the kernel has work) and discard it
            sample_row<true>(img, h, w, pitch, edges[ec].x, edges[ec].y, sc[ec].x, sc[ec].y, side, row, p, active, pix2);
        }
        const bool sent = normalise_rows(active, p, nr);
        if (active)
        {
            float4 *dst = reinterpret_cast<float4 *>(bank + (size_t)e * BANK_EDGE + side * BANK_SIDE + row * 8);
            dst[0] = make_float4(nr[0], nr[1], nr[2], nr[3]);
            dst[1] = make_float4(nr[4], nr[5], nr[6], sent ? 1.0f : 0.0f);
        }
    }
}

// wave-synchronous exchange through LDS: the writes of this wave are complete and visible to its other lanes
__device__ inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One WAVE owns NW consecutive left edges (a contiguous range of CSR pairs): no workgroup barrier anywhere, so waves in
// their sampling phase and waves in their scoring phase overlap freely on a SIMD.
template <int NW, int WPE>
__global__ __launch_bounds__(256, WPE) void ncc_tile_kernel(const uint8_t *__restrict__ imgL,
                                                            const uint16_t *__restrict__ pix2L, int h, int w, int pitch,
                                                       const ebvo_edge *__restrict__ L, const double2 *__restrict__ scL,
                                                       DevN nLd, const int32_t *__restrict__ row_ptr,
                                                       const int32_t *__restrict__ col_idx,
                                                       const float *__restrict__ rbank, int64_t cap, double thr,
                                                       double *__restrict__ sims, double *__restrict__ best,
                                                       uint8_t *__restrict__ keep, int32_t *__restrict__ match_part)
{
    static_assert(NW == 4, "one sampling round: four 16-lane groups, one per left edge of the tile");
    __shared__ __attribute__((aligned(16))) float s_left_all[4][NW * 2 * 7 * 8];
    __shared__ int s_mc;
    const int nL = devn(nLd);
    const int ntiles = (nL + NW - 1) / NW;
    // XCD-aware tile order: workgroups b and b + 8 share an XCD (and its L2).  XCD x walks the x-th contiguous eighth of the
    // tiles (whatever the edge count turns out to be: the grid is sized by capacity), so the slice of the right bank its
    // pairs touch stays in ITS L2 and all eight XCDs get equal shares.  Requires gridDim.x % 8 == 0 (the host rounds).
    const int per_xcd = (ntiles + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int t_begin = xcd * per_xcd, t_end = t_begin + per_xcd < ntiles ? t_begin + per_xcd : ntiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane & 15, side = g >> 3, row = g & 7, r8 = lane & 7;
    float *s_left = s_left_all[wave];
    if (threadIdx.x == 0)
        s_mc = 0;
    int mc = 0; // kept pairs seen by this lane
    // A tile is a chain of DEPENDENT loads -- row starts -> column indices -> bank rows, edge -> pixels -- around ~1.3 us of
    // arithmetic, and the kernel's registers leave no room for another kernel's waves beside its own: with every load issued
    // where its address became known the waves spent half their time parked (SQ_WAIT_ANY 51 %) and the kernel cost the pair
    // rate 60 us for 41 us of VALU time (tools/gpu_prefix_chain.py).  So the loads are issued EARLY:
    //   * the head of the NEXT tile (row starts, the lane's edge and its sin / cos) while this tile is scored;
    //   * the tile's first 64 column indices, one per lane, as soon as its row starts are known, i.e. before the sampling
    //     (a pair's index then comes from a lane, ds_bpermute; longer tiles read the rest from memory);
    //   * the bank rows of the NEXT eight pairs before the current eight are scored.
    struct Head
    {
        int rp_l;      // lane t <= NW: row_ptr[e0 + min(t, rows)], clamped to the capacity of the pair buffers
        double2 xy, sc; // the lane's left edge (16-lane group el = lane >> 4) and sin / cos of its orientation
    };
    auto load_head = [&](int tile) {
        Head hd;
        const int e0 = tile * NW;
        const int rows = nL - e0 < NW ? nL - e0 : NW;
        hd.rp_l = 0;
        if (lane <= NW)
        {
            const int64_t v = row_ptr[e0 + (lane < rows ? lane : rows)];
            hd.rp_l = (int)(v < cap ? v : cap);
        }
        const int el = lane >> 4;
        const int ec = (el < rows && row < 7) ? e0 + el : 0; // inactive lanes read edge 0 (always present when the kernel has work)
        hd.xy = *reinterpret_cast<const double2 *>(&L[ec].x);
        hd.sc = scL[ec];
        return hd;
    };
    int tile = t_begin + slot * 4 + wave;
    Head cur{};
    if (tile < t_end)
        cur = load_head(tile);
    while (tile < t_end)
    {
        const int e0 = tile * NW;
        const int rows = nL - e0 < NW ? nL - e0 : NW;
        // row starts as wave-uniform scalars (v_readlane; __shfl would be a ds_bpermute through the LDS pipe each)
        int rps[NW + 1];
#pragma unroll
        for (int t = 0; t <= NW; ++t)
            rps[t] = __builtin_amdgcn_readlane(cur.rp_l, t);
        const int k0 = rps[0], k1 = rps[NW]; // lanes beyond `rows` hold rp[rows]
        // the tile's first 128 column indices, two per lane (an average tile has 18 pairs; one with more than 128 reads them
        // where it needs them)
        int ci_l = 0, ci_h = 0;
        if (k0 + lane < k1)
            ci_l = col_idx[k0 + lane];
        if (k0 + 64 + lane < k1)
            ci_h = col_idx[k0 + 64 + lane];
        // phase 1: the tile's left patches (src/Stereo_Matches.cpp:578), normalised as get_patch_similarity does
        {
            const int el = lane >> 4;
            const bool active = el < rows && row < 7;
            float p[7], nr[7];
#pragma unroll
            for (int c = 0; c < 7; ++c)
                p[c] = 0.0f;
            sample_row<true>(imgL, h, w, pitch, cur.xy.x, cur.xy.y, cur.sc.x, cur.sc.y, side, row, p, active, pix2L);
            const bool sent = normalise_rows(active, p, nr);
            if (active)
            {
                float4 *dst = reinterpret_cast<float4 *>(&s_left[((el * 2 + side) * 7 + row) * 8]);
                dst[0] = make_float4(nr[0], nr[1], nr[2], nr[3]);
                dst[1] = make_float4(nr[4], nr[5], nr[6], sent ? 1.0f : 0.0f);
            }
        }
        const int next_tile = tile + slots * 4;
        Head nxt{};
        if (next_tile < t_end)
            nxt = load_head(next_tile);
        wave_lds_sync();
        // phase 2: EIGHT lanes per pair (src/Stereo_Matches.cpp:585-608): lane r holds row r of R+, R-, L+, L-.
        // The four 49-term dots are reduced TRANSPOSED: lane r ends up owning ONE of them, L(u) . R(v) with u = bit 1 of r ^ bit 2,
        // v = bit 0 ^ bit 2 (0 = the + patch, 1 = the - patch), and names its rows accordingly: la / lb = its own / the other
        // left patch, ra / rb likewise.  Of its four row dots it keeps K = la.ra throughout, keeps S = lb.ra for one step, and
        // hands T1 = la.rb, T2 = lb.rb to its xor-1 partner (whose v is the opposite: they are ITS K and S terms):
        //     K += T1', S += T2' (xor 1);   K += S' (xor 2: the partner's u is the opposite);   K += K' (lane 7 - r: same u, v)
        // Every dot is still ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)) of the same seven-term row sums -- IEEE addition is
        // commutative, so which lane of a pair does the adding does not matter -- at 12 cross-lane instructions instead of 36.
        // Lanes 0..3 of a group then hold L+.R+, L+.R-, L-.R+, L-.R-.
        int r8l = r8;
        asm volatile("" : "+v"(r8l)); // the offsets below are rebuilt per tile: kept live across the sampling phase they spill
        const int own_u = ((r8l >> 1) ^ (r8l >> 2)) & 1, own_v = (r8l ^ (r8l >> 2)) & 1;
        const int row7 = r8l < 7 ? r8l : 6; // the eighth lane reads row 6 again (in bounds) and contributes zeros
        const int off_a = own_v * BANK_SIDE + row7 * 8, off_b = (own_v ^ 1) * BANK_SIDE + row7 * 8;
        // bank rows of the eight pairs starting at kb: lanes past the end read the last pair again and store nothing.  The eight
        // pairs lie in one 64-pair chunk (kb - k0 is a multiple of 8), so which of the two index registers holds them is
        // wave-uniform; `preloaded` is false only for a tile of more than 128 pairs.
        auto bank_rows = [&](auto preloaded, int kb, float4 &a0, float4 &a1, float4 &b0, float4 &b1) {
            const int k = kb + (lane >> 3);
            const int kc = k < k1 ? k : k1 - 1;
            int ri;
            if (decltype(preloaded)::value)
            {
                const int kbc = kb < k1 ? kb : k1 - 1;
                ri = __shfl(kbc - k0 < 64 ? ci_l : ci_h, (kc - k0) & 63);
            }
            else
                ri = col_idx[kc];
            const float *base = rbank + (size_t)ri * BANK_EDGE;
            const float4 *ra4 = reinterpret_cast<const float4 *>(base + off_a);
            const float4 *rb4 = reinterpret_cast<const float4 *>(base + off_b);
            a0 = ra4[0];
            a1 = ra4[1];
            b0 = rb4[0];
            b1 = rb4[1];
        };
        auto score = [&](int kb, const float4 &a0, const float4 &a1, const float4 &b0, const float4 &b1) {
            const int k = kb + (lane >> 3);
            const bool valid = k < k1;
            const int kc = valid ? k : k1 - 1;
            // local row of pair k: the number of row starts rp[1 .. rows - 1] at or below k
            int el = 0;
#pragma unroll
            for (int t = 1; t < NW; ++t)
                el += (t < rows && kc >= rps[t]) ? 1 : 0;
            const float4 *la4 = reinterpret_cast<const float4 *>(&s_left[((el * 2 + own_u) * 7 + row7) * 8]);
            const float4 *lb4 = reinterpret_cast<const float4 *>(&s_left[((el * 2 + (own_u ^ 1)) * 7 + row7) * 8]);
            const float4 p0 = la4[0], p1 = la4[1], m0 = lb4[0], m1 = lb4[1];
            const float ra[7] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z};
            const float rb[7] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z};
            const float la[7] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z};
            const float lb[7] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z};
            // four dots of this lane's rows, column by column (each accumulator sees its terms left to right, as dot7)
            double K, S, T1, T2;
            {
                const double lad = (double)la[0], lbd = (double)lb[0], rad = (double)ra[0], rbd = (double)rb[0];
                K = lad * rad;
                S = lbd * rad;
                T1 = lad * rbd;
                T2 = lbd * rbd;
            }
#pragma unroll
            for (int c = 1; c < 7; ++c)
            {
                const double lad = (double)la[c], lbd = (double)lb[c], rad = (double)ra[c], rbd = (double)rb[c];
                K = __builtin_fma(lad, rad, K);
                S = __builtin_fma(lbd, rad, S);
                T1 = __builtin_fma(lad, rbd, T1);
                T2 = __builtin_fma(lbd, rbd, T2);
            }
            if (r8l == 7)
                K = S = T1 = T2 = 0.0;
            K += dpp_f64<0xB1>(T1);  // quad_perm:[1,0,3,2]
            S += dpp_f64<0xB1>(T2);
            K += dpp_f64<0x4E>(S);   // quad_perm:[2,3,0,1]
            K += dpp_f64<0x141>(K);  // row_half_mirror
            // src/utility.cpp:170-172: a flat patch on either side scores -1 (the flag is the eighth float of every row)
            const double mine = (p1.w != 0.0f || a1.w != 0.0f) ? -1.0 : K;
            const double pn = dpp_f64<0x55>(mine), npv = dpp_f64<0xAA>(mine), nn = dpp_f64<0xFF>(mine); // lanes 1, 2, 3 of the quad
            if (valid && r8l == 0)
            {
                const double pp = mine;
                const double b = max4(pp, nn, pn, npv); // src/Stereo_Matches.cpp:596
                if (sims) // the four scores are an option of the resident path (EBVO_PAIR_NO_SIMS): the reference keeps only their
                {         // maximum (refine_final_scores, src/Stereo_Matches.cpp:600)
                    double4 *sp = reinterpret_cast<double4 *>(sims + (size_t)k * 4);
                    *sp = make_double4(pp, nn, pn, npv);
                }
                best[k] = b;
                const bool m = b > thr; // :597
                keep[k] = m ? 1 : 0;
                mc += m ? 1 : 0;
            }
        };
        // two register sets of bank rows, used alternately (a loop that renames "next" to "current" makes the compiler copy the
        // rows -- and wait for a load it has just issued).  The loads of the following eight pairs are issued UNCONDITIONALLY
        // (past the end: the last pair again, a cache hit): behind a branch the compiler's wait for the current rows becomes a
        // wait for every load in flight, the prefetched ones included.
        auto score_tile = [&](auto preloaded) {
            float4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
            bank_rows(preloaded, k0, xa0, xa1, xb0, xb1);
            for (int kb = k0; kb < k1; kb += 16)
            {
                bank_rows(preloaded, kb + 8, ya0, ya1, yb0, yb1);
                score(kb, xa0, xa1, xb0, xb1);
                bank_rows(preloaded, kb + 16, xa0, xa1, xb0, xb1);
                if (kb + 8 < k1) // wave-uniform
                    score(kb + 8, ya0, ya1, yb0, yb1);
            }
        };
        if (k1 - k0 > 128)
            score_tile(std::false_type{});
        else if (k0 < k1)
            score_tile(std::true_type{});
        wave_lds_sync(); // the next tile's patches overwrite s_left
        tile = next_tile;
        cur = nxt;
    }
    for (int d = 32; d > 0; d >>= 1)
        mc += __shfl_down(mc, d);
    __syncthreads();
    if (lane == 0 && mc)
        atomicAdd(&s_mc, mc);
    __syncthreads();
    if (threadIdx.x == 0)
        match_part[blockIdx.x] = s_mc;
}


// ---- temporal quads: candidate search and NCC on stored patches (Temporal_Matches, configs[2]) -----------------------
// apply_spatial_grid_filtering_quads + apply_orientation_filtering_quads (src/Temporal_Matches.cpp:335-414) decide, per
// keyframe stereo mate, which current-frame mates are candidates: the mate's LEFT edge lies in a grid cell (15 px,
// include/definitions.h:45) within +-ceil(30 / 15) cells of the keyframe mate's left edge, its RIGHT edge likewise
// relative to the keyframe mate's right edge (SpatialGrid::getCandidatesWithinRadius, include/Dataset.h:92-113: whole
// cells, no distance test), and both orientation differences pass the 10-degree test.
// The reference's two SpatialGrids become one CSR grid over the LEFT cells (mates of a cell in ascending index, the order
// in which the reference inserts them); a thread owns one keyframe mate and walks the (2 sr + 1)^2 neighbour cells in
// the reference's order (dy outer, dx inner): the candidates of a row come out in exactly the order of
// `left_candidates` (src/Temporal_Matches.cpp:352, :357), the right-grid membership is the cell test on the mate's right
// edge.  ~25 cells x ~40 mates per query instead of every mate of the five cell rows (the mates are in raster order:
// index ranges are narrow in y but span the image in x).
struct MateCells
{
    short lx, ly, rx, ry; // -30000: the edge is outside the grid and is never returned by a query
};

__device__ inline short cell_of(double v, int cell, int n_cells)
{
    const int c = (int)v / cell; // static_cast<int>(location.x) / cell_size
    return (c >= 0 && c < n_cells) ? (short)c : (short)-30000;
}

// cells of every current-frame mate + the population of the left cells
__global__ __launch_bounds__(256) void mate_cells_kernel(const ebvo_edge *__restrict__ L, const ebvo_edge *__restrict__ R, int n,
                                                         int cell, int gw, int gh, MateCells *__restrict__ out,
                                                         int32_t *__restrict__ cell_cnt)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    {
        MateCells m;
        m.lx = cell_of(L[j].x, cell, gw);
        m.ly = cell_of(L[j].y, cell, gh);
        m.rx = cell_of(R[j].x, cell, gw);
        m.ry = cell_of(R[j].y, cell, gh);
        // a mate is in the left grid only if BOTH left cells are inside (src/Temporal_Matches.cpp:31-34), same on the right
        if (m.lx < 0 || m.ly < 0)
            m.lx = m.ly = -30000;
        if (m.rx < 0 || m.ry < 0)
            m.rx = m.ry = -30000;
        out[j] = m;
        if (m.lx >= 0)
            atomicAdd(&cell_cnt[m.ly * gw + m.lx], 1);
    }
}

// exclusive scan of the cell populations (one block; a grid has a few thousand cells) -> cell_start[0 .. n_cells]
__global__ __launch_bounds__(256) void cell_scan_kernel(const int32_t *__restrict__ cnt, int n_cells, int32_t *__restrict__ start)
{
    __shared__ int32_t part[256];
    const int per = (n_cells + 255) / 256;
    const int beg = min(n_cells, (int)threadIdx.x * per), end = min(n_cells, beg + per);
    int32_t sum = 0;
    for (int c = beg; c < end; ++c)
        sum += cnt[c];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int32_t run = 0;
        for (int t = 0; t < 256; ++t)
        {
            const int32_t v = part[t];
            part[t] = run;
            run += v;
        }
        start[n_cells] = run;
    }
    __syncthreads();
    int32_t run = part[threadIdx.x];
    for (int c = beg; c < end; ++c)
    {
        start[c] = run;
        run += cnt[c];
    }
}

// mates into their cell's segment, in arrival order (sorted next)
__global__ __launch_bounds__(256) void cell_scatter_kernel(const MateCells *__restrict__ cells, int n, int gw,
                                                           const int32_t *__restrict__ start, int32_t *__restrict__ fill,
                                                           int32_t *__restrict__ list)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    {
        const MateCells m = cells[j];
        if (m.lx < 0)
            continue;
        const int c = m.ly * gw + m.lx;
        list[start[c] + atomicAdd(&fill[c], 1)] = j;
    }
}

// every cell's segment into ascending mate index (the reference's insertion order): rank sort, one wave per cell
__global__ __launch_bounds__(256) void cell_sort_kernel(const int32_t *__restrict__ start, int n_cells,
                                                        const int32_t *__restrict__ list, int32_t *__restrict__ sorted)
{
    const int lane = threadIdx.x & 63;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n_cells; c += gridDim.x * 4)
    {
        const int a = start[c], b = start[c + 1];
        for (int e = a + lane; e < b; e += 64)
        {
            const int32_t v = list[e];
            int rank = 0;
            for (int k = a; k < b; ++k)
                rank += list[k] < v ? 1 : 0;
            sorted[a + rank] = v; // the indices of a cell are distinct
        }
    }
}

__device__ inline bool orient_close(double a, double b, double thr)
{
    double od = fabs((a - b) * 0x1.ca5dc1a63c1f8p+5 /* rad_to_deg: theta * (180.0 / M_PI) */);
    if (od > 180.0)
        od = 360.0 - od;
    return od < thr || fabs(od - 180.0) < thr;
}

// sixteen lanes per keyframe mate: lane e tests every sixteenth mate of a cell's segment, the survivors of a step are
// ranked by a ballot so that the row keeps the order of the segment (one thread per mate walked ~900 dependent
// look-ups in a row: 350 us per pass at EuRoC size)
template <bool FILL>
__global__ __launch_bounds__(256) void temporal_candidates_kernel(const ebvo_edge *__restrict__ kfL,
                                                                  const ebvo_edge *__restrict__ kfR, int n_kf,
                                                                  const ebvo_edge *__restrict__ cfL,
                                                                  const ebvo_edge *__restrict__ cfR,
                                                                  const MateCells *__restrict__ cells,
                                                                  const int32_t *__restrict__ cell_start,
                                                                  const int32_t *__restrict__ cell_list, int cell, int sr, int gw,
                                                                  int gh, double orient_thr, int32_t *__restrict__ cnt,
                                                                  const int32_t *__restrict__ row_ptr,
                                                                  int32_t *__restrict__ col_idx, int64_t cap)
{
    const int lane = threadIdx.x & 63, e = lane & 15, gshift = lane & 48;
    const int groups = (gridDim.x * blockDim.x) >> 4;
    // the four groups of a wave advance together (the ballots below are wave-wide): a group past the end idles
    for (int w0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4; w0 < n_kf; w0 += groups)
    {
        const int i0 = w0 + (lane >> 4);
        const bool live = i0 < n_kf;
        const int i = live ? i0 : 0;
        const ebvo_edge kl = kfL[i], kr = kfR[i];
        // the query cells are not clipped (include/Dataset.h:95-96); neighbour cells outside the grid hold nothing
        const int qlx = (int)kl.x / cell, qly = (int)kl.y / cell, qrx = (int)kr.x / cell, qry = (int)kr.y / cell;
        int c = 0;
        int64_t o = (FILL && live) ? row_ptr[i] : 0;
        for (int dy = -sr; dy <= sr; ++dy)
            for (int dx = -sr; dx <= sr; ++dx)
            {
                const int ny = qly + dy, nx = qlx + dx;
                const bool in_grid = live && ny >= 0 && ny < gh && nx >= 0 && nx < gw;
                const int a = in_grid ? cell_start[ny * gw + nx] : 0, b = in_grid ? cell_start[ny * gw + nx + 1] : 0;
                for (int k0 = a; __any(k0 < b); k0 += 16)
                {
                    const int k = k0 + e;
                    bool ok = false;
                    int j = 0;
                    if (k < b)
                    {
                        j = cell_list[k];
                        const MateCells m = cells[j];
                        // right_set.count(cf_idx): the mate's right edge is in a neighbour cell of the right query
                        ok = abs(m.rx - qrx) <= sr && abs(m.ry - qry) <= sr && orient_close(kl.theta, cfL[j].theta, orient_thr) &&
                             orient_close(kr.theta, cfR[j].theta, orient_thr);
                    }
                    const unsigned hits = (unsigned)((__ballot(ok) >> gshift) & 0xffffull);
                    if (FILL && ok)
                    {
                        const int64_t pos = o + __popc(hits & ((1u << e) - 1u));
                        if (pos < cap)
                            col_idx[pos] = j;
                    }
                    const int nh = __popc(hits);
                    c += nh;
                    o += nh;
                }
            }
        if (!FILL && live && e == 0)
            cnt[i] = c;
    }
}

// apply_NCC_filtering_quads (src/Temporal_Matches.cpp:416-469) on mates' stored patches, by index: banks of normalised
// patches [mate][2][49] + sentinel flags [mate][2] (patches_kernel).  Sixteen lanes per quad: lanes 0-7 the left-image
// patches, lanes 8-15 the right-image ones; lane r holds row r of the four patches involved.
__global__ __launch_bounds__(256) void ncc_quads_indexed_kernel(const float *__restrict__ kfLn, const uint8_t *__restrict__ kfLf,
                                                                const float *__restrict__ kfRn, const uint8_t *__restrict__ kfRf,
                                                                const float *__restrict__ cfLn, const uint8_t *__restrict__ cfLf,
                                                                const float *__restrict__ cfRn, const uint8_t *__restrict__ cfRf,
                                                                const int32_t *__restrict__ quad_kf,
                                                                const int32_t *__restrict__ quad_cf, DevCount nqd, double thr,
                                                                double *__restrict__ sim_left, double *__restrict__ sim_right,
                                                                uint8_t *__restrict__ keep)
{
    const int64_t n_quads = devcount(nqd);
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = (int)(t & 15), img = g >> 3, row = g & 7;
    const int64_t iters = (n_quads + groups - 1) / groups;
    for (int64_t it = 0; it < iters; ++it)
    {
        const int64_t k = it * groups + (t >> 4);
        const bool valid = k < n_quads;
        const bool active = valid && row < 7;
        float a1[7], a2[7], b1[7], b2[7];
#pragma unroll
        for (int c = 0; c < 7; ++c)
            a1[c] = a2[c] = b1[c] = b2[c] = 0.0f;
        size_t ik = 0, ic = 0;
        if (valid)
        {
            ik = (size_t)quad_kf[k];
            ic = (size_t)quad_cf[k];
        }
        const float *kn = img ? kfRn : kfLn, *cn = img ? cfRn : cfLn;
        if (active)
        {
#pragma unroll
            for (int c = 0; c < 7; ++c)
            {
                a1[c] = kn[ik * 98 + row * 7 + c];      // keyframe .first
                a2[c] = kn[ik * 98 + 49 + row * 7 + c]; // keyframe .second
                b1[c] = cn[ic * 98 + row * 7 + c];
                b2[c] = cn[ic * 98 + 49 + row * 7 + c];
            }
        }
        const double d11 = dot_rows(active, a1, b1), d12 = dot_rows(active, a1, b2), d21 = dot_rows(active, a2, b1),
                     d22 = dot_rows(active, a2, b2);
        double sim = 0.0;
        if (valid && row == 0)
        {
            const uint8_t *kf = img ? kfRf : kfLf, *cf = img ? cfRf : cfLf;
            const bool ka = kf[ik * 2] != 0, kb = kf[ik * 2 + 1] != 0, ca = cf[ic * 2] != 0, cb = cf[ic * 2 + 1] != 0;
            // :441-450 order: (first, first), (first, second), (second, first), (second, second); -1 for a flat patch
            sim = max4((ka || ca) ? -1.0 : d11, (ka || cb) ? -1.0 : d12, (kb || ca) ? -1.0 : d21, (kb || cb) ? -1.0 : d22);
        }
        const double sim_r = __shfl_xor(sim, 8);
        if (valid && g == 0)
        {
            sim_left[k] = sim;
            sim_right[k] = sim_r;
            keep[k] = (sim > thr && sim_r > thr) ? 1 : 0; // :452
        }
    }
}

__global__ void count_flags_kernel(const uint8_t *__restrict__ f, DevCount nd, unsigned long long *__restrict__ out)
{
    const int64_t n = devcount(nd);
    unsigned long long c = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        c += f[k] ? 1 : 0;
    for (int d = 32; d > 0; d >>= 1)
        c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(out, c); // integer sum: order-independent
}

// apply_orientation_filter (src/Stereo_Matches.cpp:863-915) as a flag per listed pair: sixteen lanes per left edge
__global__ __launch_bounds__(256) void orient_flags_kernel(const ebvo_edge *__restrict__ L, int nL, const ebvo_edge *__restrict__ R,
                                                           const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx,
                                                           double thr, uint8_t *__restrict__ ok)
{
    const int e = threadIdx.x & 15;
    const int rows = (gridDim.x * blockDim.x) >> 4;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < nL; i += rows)
    {
        const double lth = L[i].theta;
        for (int k = row_ptr[i] + e; k < row_ptr[i + 1]; k += 16)
        {
            double od = fabs((lth - R[col_idx[k]].theta) * 0x1.ca5dc1a63c1f8p+5 /* 180.0 / M_PI */);
            if (od > 180.0)
                od = 360.0 - od;
            ok[k] = (od < thr || fabs(od - 180.0) < thr) ? 1 : 0; // :897
        }
    }
}

// Last kernel of a device-resident pair: gathers every count the host wants into one record (PairResult,
// ebvo_internal.h).
__global__ void pair_result_kernel(const int32_t *__restrict__ cntL, const int32_t *__restrict__ cntR,
                                   const unsigned long long *__restrict__ total_part, int n_total_part,
                                   const int32_t *__restrict__ match_part, int n_part, int64_t cap, int cand_cap,
                                   PairResult *__restrict__ out)
{
    // one block of 1024 threads: the per-block candidate totals and match counts are summed here
    __shared__ unsigned long long s_t[16];
    __shared__ int s_m[16];
    unsigned long long tot = 0;
    int m = 0;
    for (int k = threadIdx.x; k < n_total_part; k += blockDim.x)
        tot += total_part[k];
    for (int k = threadIdx.x; k < n_part; k += blockDim.x)
        m += match_part[k];
    for (int d = 32; d > 0; d >>= 1)
    {
        tot += __shfl_down(tot, d);
        m += __shfl_down(m, d);
    }
    if ((threadIdx.x & 63) == 0)
    {
        s_t[threadIdx.x >> 6] = tot;
        s_m[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        tot = 0;
        m = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k)
        {
            tot += s_t[k];
            m += s_m[k];
        }
    }
    if (threadIdx.x || blockIdx.x)
        return;
    PairResult r;
    r.n_total_left = cntL[0];
    r.n_left = cntL[1];
    r.n_total_right = cntR[0];
    r.n_right = cntR[1];
    r.n_pairs = (int64_t)tot;
    r.n_matches = m;
    r.overflow = (r.n_pairs > cap || r.n_pairs > 0x7fffffffll) ? 1 : 0;
    if (cand_cap > 0 && (cntL[4] > cand_cap || cntR[4] > cand_cap))
        r.overflow |= 2;
    r.pad = 0;
    *out = r;
}

// NCC of stored patch pairs (src/utility.cpp:163-180); 16 lanes per pair: side 0 = A, side 1 = B.
__global__ __launch_bounds__(256) void ncc_stored_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                         int n, double *__restrict__ sim)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 4, g = t & 15, side = g >> 3, row = g & 7;
    const bool valid = k < n;
    const bool active = valid && row < 7;
    float p[7], nr[7], other[7];
#pragma unroll
    for (int c = 0; c < 7; ++c)
        p[c] = 0.0f;
    if (active)
    {
        const float *src = (side ? B : A) + (size_t)k * 49 + row * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c)
            p[c] = src[c];
    }
    const bool sent = normalise_rows(active, p, nr);
#pragma unroll
    for (int c = 0; c < 7; ++c)
        other[c] = __shfl_xor(nr[c], 8);
    const int s_i = sent ? 1 : 0, s_o = __shfl_xor(s_i, 8);
    const double d = dot_rows(active, nr, other); // side 0: A row . B row
    if (valid && g == 0)
        sim[k] = (s_i || s_o) ? -1.0 : d;
}

// ------------------------------------------------------------------------------------------
// FP64 vector-ALU peak: 16 independent chains of (mul, add) or fma per thread.
template <bool FMA>
__global__ __launch_bounds__(256) void fp64_peak_kernel(double *out, int iters, double m, double a)
{
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
        v[t] = (double)(threadIdx.x + t) * 1e-3;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int t = 0; t < 16; ++t)
        {
            if (FMA)
                v[t] = __builtin_fma(v[t], m, a);
            else
                v[t] = v[t] * m + a; // contraction is off: v_mul_f64 + v_add_f64
        }
    }
    double s = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t)
        s += v[t];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// exclusive scans of n (+ n_add) int32 each, nb arrays (<= SCAN_BATCH) in one pair of launches; the lengths may live on
// the device, cap_n bounds them on the host
int device_exclusive_scan_batch(ebvo_ctx *ctx, Slot &s, ScanBatch S, int nb, int cap_n)
{
    if (nb < 1 || nb > SCAN_BATCH || cap_n < 0)
        return EBVO_ERR_ARG;
    S.tiles = (cap_n + SCAN_TILE - 1) / SCAN_TILE;
    if (S.tiles < 1)
        S.tiles = 1;
    int rc = ebvo_grow(ctx, s, s.scan_tmp, sizeof(int32_t) * (size_t)S.tiles * SCAN_BATCH);
    if (rc)
        return rc;
    S.sums = (int32_t *)s.scan_tmp.p;
    S.direct = S.tiles <= SCAN_DIRECT_TILES ? 1 : 0;
    if (S.tiles > 1 && !S.direct)
        hipLaunchKernelGGL(scan_reduce_kernel, dim3(S.tiles - 1, nb), dim3(SCAN_BLOCK), 0, s.stream, S);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(S.tiles, nb), dim3(SCAN_BLOCK), 0, s.stream, S);
    return EBVO_OK;
}

int device_exclusive_scan(ebvo_ctx *ctx, Slot &s, const int32_t *in, int32_t *out, DevN n, int n_add, int cap_n,
                          int32_t *total = nullptr)
{
    ScanBatch S{};
    S.in[0] = in;
    S.out[0] = out;
    S.n[0] = n;
    S.n_add = n_add;
    S.total[0] = total;
    return device_exclusive_scan_batch(ctx, s, S, 1, cap_n);
}

inline unsigned blocks_for(int64_t items, int per_block, int max_blocks)
{
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

} // namespace

int ebvo_clear_enqueue(ebvo_ctx *ctx, Slot &s, int32_t *const ptrs[], const int counts[], int n)
{
    if (n < 1 || n > EBVO_CLEAR_MAX)
        return EBVO_ERR_ARG;
    ClearList C{};
    int most = 1;
    for (int k = 0; k < n; ++k)
    {
        C.p[k] = ptrs[k];
        C.n[k] = counts[k];
        most = counts[k] > most ? counts[k] : most;
    }
    hipLaunchKernelGGL(clear_kernel, dim3(blocks_for(most, 256, 64), n), dim3(256), 0, s.stream, C);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int ebvo_device_scan(ebvo_ctx *ctx, Slot &s, const int32_t *in, int32_t *out, int n_host, const int32_t *n_dev, int n_add,
                     int cap_n, int32_t *d_total)
{
    ProfScope ps(ctx, s, K_SCAN);
    int rc = device_exclusive_scan(ctx, s, in, out, DevN{n_host, n_dev}, n_add, cap_n, d_total);
    if (rc)
        return rc;
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_lines_enqueue(ebvo_ctx *ctx, Slot &s, const double *d_F, const ebvo_edge *d_edges, int n, const int32_t *d_n,
                        int cap_n, double *d_lines)
{
    if (!d_n && n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_LINES);
    hipLaunchKernelGGL(lines_kernel, dim3(blocks_for(d_n ? cap_n : n, 256, 1024)), dim3(256), 0, s.stream, d_F, d_edges,
                       DevN{n, d_n}, d_lines);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

static CandParams cand_params(Slot &s, int nL, const int32_t *d_nL, int nR, const int32_t *d_nR, double epi_thr,
                              double max_disp, double orient_thr_deg, int stage_mask, int64_t cap)
{
    CandParams P;
    P.epi_thr = epi_thr;
    P.max_disp = max_disp;
    P.orient_thr = orient_thr_deg;
    P.mask = stage_mask;
    P.nL = DevN{nL, d_nL};
    P.nR = DevN{nR, d_nR};
    P.cap = cap;
    P.stage = (int32_t *)s.cand_stage.p;
    P.tile_flag = (int32_t *)s.cand_tileflag.p;
    return P;
}

int match_candidates_fill_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const int32_t *d_nL,
                                  const ebvo_edge *d_R, int nR, const int32_t *d_nR, int cap_edges,
                                  const double *d_lines, double epi_thr, double max_disp, double orient_thr_deg,
                                  int stage_mask)
{
    const CandParams P = cand_params(s, nL, d_nL, nR, d_nR, epi_thr, max_disp, orient_thr_deg, stage_mask, s.cap_pairs);
    ProfScope ps(ctx, s, K_CAND_FILL);
    hipLaunchKernelGGL(candidates_kernel<true>, dim3(blocks_for(d_nL ? cap_edges : nL, TILE, 4096 / (ctx->small_div > 0 ? ctx->small_div : 4))), dim3(256), 0,
                       s.stream, d_L, d_R, d_lines, (const Box *)s.boxes_chunk.p, (const Box *)s.boxes_group.p, P,
                       (int32_t *)nullptr, (const int32_t *)s.row_ptr.p, (int32_t *)s.col_idx.p,
                       (unsigned long long *)nullptr);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

// the buffers of the candidate search for capL left and capR right edges; *ntiles_out, *capgroups_out: what the launches need
static int candidates_buffers(ebvo_ctx *ctx, Slot &s, int capL, int capR, size_t *ntiles_out, int *capgroups_out)
{
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.row_ptr, sizeof(int32_t) * ((size_t)capL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.cand_cnt, sizeof(int32_t) * ((size_t)capL + 1))))
        return rc;
    const int capchunks = (capR + CHUNK - 1) / CHUNK + 1, capgroups = (capchunks + GROUP - 1) / GROUP + 1;
    if ((rc = ebvo_grow(ctx, s, s.boxes_chunk, sizeof(Box) * (size_t)capchunks)))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.boxes_group, sizeof(Box) * (size_t)capgroups)))
        return rc;
    const size_t ntiles = ((size_t)capL + TILE - 1) / TILE + 1;
    if ((rc = ebvo_grow(ctx, s, s.cand_stage, sizeof(int32_t) * STAGE * ((size_t)capL + 1))))
        return rc;
    if ((rc = ebvo_grow(ctx, s, s.cand_tileflag, sizeof(int32_t) * ntiles)))
        return rc;
    *ntiles_out = ntiles;
    *capgroups_out = capgroups;
    return EBVO_OK;
}

// lines + boxes (+ tile flags) + sincos + row pairs of a resident pair in one launch (match_prep_kernel); the later
// match_candidates_enqueue / match_ncc_resident_enqueue calls are told that their preparations have run
int match_prep_enqueue(ebvo_ctx *ctx, Slot &s, int h, int w, int cap_edges)
{
    int rc;
    size_t ntiles = 0;
    int capgroups = 0;
    if ((rc = candidates_buffers(ctx, s, cap_edges, cap_edges, &ntiles, &capgroups)) ||
        (rc = ebvo_grow(ctx, s, s.sincos, sizeof(double2) * 2 * (size_t)cap_edges)))
        return rc;
    PrepArgs A{};
    A.F = s.d_F;
    A.L = s.im[0].edges;
    A.R = s.im[1].edges;
    A.nL = DevN{0, s.im[0].counts + 1};
    A.nR = DevN{0, s.im[1].counts + 1};
    A.lines = (double *)s.lines.p;
    A.cb = (Box *)s.boxes_chunk.p;
    A.gb = (Box *)s.boxes_group.p;
    A.tile_flag = (int32_t *)s.cand_tileflag.p;
    A.ntiles = (int)ntiles;
    A.scL = (double2 *)s.sincos.p;
    A.scR = A.scL + cap_edges;
    A.img0 = ncc_img(s, 0);
    A.img1 = ncc_img(s, 1);
    A.pix0 = s.im[0].pix2;
    A.pix1 = s.im[1].pix2;
    A.h = h;
    A.w = w;
    A.b_lines = (int)blocks_for(cap_edges, 256, 1024);
    A.b_boxes = (int)blocks_for(capgroups, 4, 1024);
    A.b_sincos = (int)blocks_for(cap_edges, 256, 512);
    A.b_rows = (int)blocks_for(((int64_t)h * w + 3) / 4, 256, 512);
    ProfScope ps(ctx, s, K_BOXES);
    hipLaunchKernelGGL(match_prep_kernel, dim3(A.b_lines + A.b_boxes + 2 * A.b_sincos + 2 * A.b_rows), dim3(256), 0, s.stream, A);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_candidates_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const int32_t *d_nL,
                             const ebvo_edge *d_R, int nR, const int32_t *d_nR, int cap_edges, const double *d_lines,
                             double epi_thr, double max_disp, double orient_thr_deg, int stage_mask, bool fill, bool prep_done)
{
    const int capL = d_nL ? cap_edges : nL, capR = d_nR ? cap_edges : nR;
    int rc;
    size_t ntiles = 0;
    int capgroups = 0;
    if ((rc = candidates_buffers(ctx, s, capL, capR, &ntiles, &capgroups)))
        return rc;
    int32_t *cnt = (int32_t *)s.cand_cnt.p;
    if (!prep_done)
    {
        // chunk / group boxes; the kernel also zeroes the tile flags (cnt[nL], the scan's trailing zero, and the per-block
        // totals are written by the counting kernel)
        ProfScope ps(ctx, s, K_BOXES);
        hipLaunchKernelGGL(boxes_kernel, dim3(blocks_for(capgroups, 4, 1024)), dim3(256), 0, s.stream, d_R, DevN{nR, d_nR},
                           (Box *)s.boxes_chunk.p, (Box *)s.boxes_group.p, (int32_t *)s.cand_tileflag.p, (int)ntiles);
    }
    const CandParams P = cand_params(s, nL, d_nL, nR, d_nR, epi_thr, max_disp, orient_thr_deg, stage_mask, s.cap_pairs);
    {
        ProfScope ps(ctx, s, K_CAND_COUNT);
        // at most 1,024 blocks (what the device keeps resident, four waves of 116 registers per SIMD), each walking ~2 tiles at
        // KITTI size: a capacity-sized grid of 4,096 (half of its blocks without a tile) cost 1.4 % of the pair rate
        // (tools/gpu_ab_keys.py, key 19: 3410 / 3459 / 3456 / 3455 pairs/s for 4096 / 512 / 768 / 1024 blocks; one pair in
        // flight: 437 / 463 / 445 / 440 us)
        const int nblk = blocks_for(capL, TILE, ctx->cand_blocks > 0 ? ctx->cand_blocks : 1024);
        s.n_total_part = nblk;
        hipLaunchKernelGGL(candidates_kernel<false>, dim3(nblk), dim3(256), 0, s.stream, d_L, d_R, d_lines,
                           (const Box *)s.boxes_chunk.p, (const Box *)s.boxes_group.p, P, cnt, (const int32_t *)nullptr,
                           (int32_t *)nullptr, s.d_total + 1);
        if (!fill) // host-buffer path reads the total back; the pipeline sums the parts in pair_result_kernel
            hipLaunchKernelGGL(total_sum_kernel, dim3(1), dim3(64), 0, s.stream,
                               (const unsigned long long *)(s.d_total + 1), nblk, s.d_total);
    }
    if (ctx->stop_stage == 9)
        return EBVO_OK;
    {
        ProfScope ps(ctx, s, K_SCAN);
        if ((rc = device_exclusive_scan(ctx, s, cnt, (int32_t *)s.row_ptr.p, DevN{nL, d_nL}, 1, capL + 1)))
            return rc;
    }
    EBVO_HIP(ctx, hipGetLastError());
    if (ctx->stop_stage == 10)
        return EBVO_OK;
    if (fill)
        return match_candidates_fill_enqueue(ctx, s, d_L, nL, d_nL, d_R, nR, d_nR, cap_edges, d_lines, epi_thr, max_disp,
                                             orient_thr_deg, stage_mask);
    return EBVO_OK;
}

int match_patches_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_img, int h, int w, int pitch,
                          const ebvo_edge *d_edges, int n, const int32_t *d_n, int cap_n, float *d_raw, float *d_norm,
                          uint8_t *d_flag)
{
    if (!d_n && n <= 0)
        return EBVO_OK;
    const int cap = d_n ? cap_n : n;
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.sincos, sizeof(double2) * (size_t)cap)))
        return rc;
    PatchBatch B{};
    B.img[0] = d_img;
    B.edges[0] = d_edges;
    B.n[0] = DevN{n, d_n};
    B.sc[0] = (double2 *)s.sincos.p;
    B.raw[0] = d_raw;
    B.norm[0] = d_norm;
    B.flag[0] = d_flag;
    ProfScope ps(ctx, s, K_PATCHES);
    hipLaunchKernelGGL(sincos_batch_kernel, dim3(blocks_for(cap, 256, 1024), 1), dim3(256), 0, s.stream, B);
    hipLaunchKernelGGL(patches_kernel, dim3(blocks_for((int64_t)cap * 16, 256, 4096), 1), dim3(256), 0, s.stream, B, h, w,
                       pitch);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_ncc_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_imgR, int h, int w, int pitchR,
                            const ebvo_edge *d_Rc, const int32_t *d_row_ptr, int nL, int64_t n_pairs,
                            const float *d_left_norm, const uint8_t *d_left_flag, double thr, double *d_sims,
                            double *d_best, uint8_t *d_keep, int32_t *d_pair_left_scratch, void *d_sincos_scratch,
                            const int32_t *d_n_pairs)
{
    if (n_pairs <= 0 || nL <= 0)
        return EBVO_OK;
    int rc;
    if (!d_pair_left_scratch && (rc = ebvo_grow(ctx, s, s.pair_left, sizeof(int32_t) * (size_t)n_pairs)))
        return rc;
    if (!d_sincos_scratch && (rc = ebvo_grow(ctx, s, s.sincos, sizeof(double2) * (size_t)n_pairs)))
        return rc;
    int32_t *pair_left = d_pair_left_scratch ? d_pair_left_scratch : (int32_t *)s.pair_left.p;
    double2 *sc = d_sincos_scratch ? (double2 *)d_sincos_scratch : (double2 *)s.sincos.p;
    {
        ProfScope ps(ctx, s, K_MISC);
        hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks_for(nL, 256, 512)), dim3(256), 0, s.stream, d_row_ptr,
                           DevN{nL, nullptr}, pair_left, n_pairs);
        hipLaunchKernelGGL(sincos_edges_kernel, dim3(blocks_for(n_pairs, 256, 1024)), dim3(256), 0, s.stream, d_Rc,
                           DevN{(int)n_pairs, d_n_pairs}, sc);
    }
    {
        ProfScope ps(ctx, s, K_NCC_PAIRS);
        const int64_t threads = n_pairs * 16;
        hipLaunchKernelGGL(ncc_pairs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s.stream, d_imgR, h,
                           w, pitchR, d_Rc, (const double2 *)sc, (const int32_t *)pair_left, DevCount{n_pairs, d_n_pairs},
                           d_left_norm, d_left_flag, thr, d_sims, d_best, d_keep, (int32_t *)nullptr);
    }
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

// NCC of the CSR pairs in s.row_ptr / s.col_idx of the resident pair: sin/cos of both edge lists, the right bank, the
// tile kernel.  s.patches_norm_r holds the right bank (BANK_EDGE floats per edge).
int match_ncc_resident_enqueue(ebvo_ctx *ctx, Slot &s, int h, int w, int cap_edges, double thr, int left, bool want_sims,
                               bool prep_done)
{
    int rc;
    if ((rc = ebvo_grow(ctx, s, s.sincos, sizeof(double2) * 2 * (size_t)cap_edges)))
        return rc;
    const int iL = left ? 1 : 0, iR = 1 - iL; // workspace of the left / of the right image
    PatchBatch B{};
    for (int k = 0; k < 2; ++k)
    {
        B.img[k] = ncc_img(s, k);
        B.edges[k] = s.im[k].edges;
        B.n[k] = DevN{0, s.im[k].counts + 1};
        B.sc[k] = (double2 *)s.sincos.p + (size_t)k * cap_edges;
    }
    const DevN nLd{0, s.im[iL].counts + 1}, nRd{0, s.im[iR].counts + 1};
    {
        ProfScope ps(ctx, s, K_PATCHES);
        if (!prep_done) // (the resident pair's chain has run both in match_prep_kernel)
        {
            hipLaunchKernelGGL(row_pairs_kernel, dim3(blocks_for(((int64_t)h * w + 3) / 4, 256, 512), 2), dim3(256), 0, s.stream,
                               ncc_img(s, 0), ncc_img(s, 1), s.im[0].pix2, s.im[1].pix2, h, w);
            hipLaunchKernelGGL(sincos_batch_kernel, dim3(blocks_for(cap_edges, 256, 512), 2), dim3(256), 0, s.stream, B);
        }
        for (int rep = 0; rep < ((ctx->repeat_mask & 4) ? 2 : 1); ++rep)
            hipLaunchKernelGGL(right_bank_kernel, dim3(blocks_for((int64_t)cap_edges * 16, 256, 1024)), dim3(256), 0, s.stream,
                               ncc_img(s, iR), (const uint16_t *)s.im[iR].pix2, h, w, w, (const ebvo_edge *)s.im[iR].edges,
                               (const double2 *)B.sc[iR], nRd, (float *)s.patches_norm_r.p);
    }
    if (ctx->stop_stage == 12)
        return EBVO_OK;
    {
        ProfScope ps(ctx, s, K_NCC_PAIRS);
        int nblk = (int)blocks_for(cap_edges, NCC_NW * 4, EBVO_MATCH_PARTS);
        // no more blocks than the device keeps resident: a wave then walks ~8 tiles with the next tile's loads in flight instead
        // of two, and no block waits for a slot (developer key 17: a grid of that many blocks)
        {
            static std::atomic<int> resident[16];
            int r = resident[ctx->device & 15].load(std::memory_order_relaxed);
            if (r == 0)
            {
                int per_cu = 0, cus = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ncc_tile_kernel<NCC_NW, NCC_WPE>, 256, 0);
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
                r = (per_cu > 0 ? per_cu : NCC_WPE) * (cus > 0 ? cus : 256);
                resident[ctx->device & 15].store(r, std::memory_order_relaxed);
            }
            if (ctx->ncc_blocks > 0)
                r = ctx->ncc_blocks;
            nblk = nblk < r ? nblk : r;
        }
        nblk = nblk < 8 ? 8 : (nblk & ~7); // the XCD-aware order needs a multiple of 8 (tiles are walked grid-stride)
        nblk = nblk < EBVO_MATCH_PARTS ? nblk : EBVO_MATCH_PARTS;
        s.n_match_part = nblk;
        // NW = 4 left edges per wave, at most 5 waves per SIMD: measured best of {4, 8, 16} x {4, 5, 6, 8} (a bigger tile
        // serialises more sampling rounds in one wave; a higher occupancy target spills)
        for (int rep = 0; rep < ((ctx->repeat_mask & 8) ? 2 : 1); ++rep)
            hipLaunchKernelGGL((ncc_tile_kernel<NCC_NW, NCC_WPE>), dim3(nblk), dim3(256), 0, s.stream, ncc_img(s, iL),
                               (const uint16_t *)s.im[iL].pix2, h, w, w, (const ebvo_edge *)s.im[iL].edges,
                               (const double2 *)B.sc[iL], nLd, (const int32_t *)s.row_ptr.p, (const int32_t *)s.col_idx.p,
                               (const float *)s.patches_norm_r.p, s.cap_pairs, thr, want_sims ? (double *)s.sims.p : nullptr,
                               (double *)s.best.p, (uint8_t *)s.keep.p, s.d_matches);
    }
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_orient_flags_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_L, int nL, const ebvo_edge *d_R, const int32_t *d_row_ptr,
                               const int32_t *d_col_idx, int64_t n_pairs, double orient_thr_deg, uint8_t *d_ok)
{
    if (nL <= 0 || n_pairs <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_CAND_FILL);
    hipLaunchKernelGGL(orient_flags_kernel, dim3(blocks_for((int64_t)nL * 16, 256, 4096)), dim3(256), 0, s.stream, d_L, nL, d_R,
                       d_row_ptr, d_col_idx, orient_thr_deg, d_ok);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

size_t match_right_bank_bytes(int cap_edges) { return sizeof(float) * BANK_EDGE * (size_t)cap_edges; }


// temporal candidate search: the CSR grid of the current-frame mates (cells, cell_start, cell_list inside one buffer), then
// count (FILL = false: cnt[n_kf]) or fill (row_ptr given)
struct TemporalGrid
{
    MateCells *cells;
    int32_t *cnt, *fill, *start, *list, *sorted;
};
static TemporalGrid temporal_grid(void *base, int n_cf, int n_cells)
{
    TemporalGrid g;
    char *p = (char *)base;
    g.cells = (MateCells *)p;
    p += (sizeof(MateCells) * (size_t)n_cf + 63) & ~(size_t)63;
    g.cnt = (int32_t *)p;
    g.fill = g.cnt + n_cells;
    g.start = g.fill + n_cells;
    g.list = g.start + n_cells + 1;
    g.sorted = g.list + n_cf;
    return g;
}
size_t match_temporal_grid_bytes(int n_cf, int n_cells)
{
    return ((sizeof(MateCells) * (size_t)n_cf + 63) & ~(size_t)63) + sizeof(int32_t) * (3 * (size_t)n_cells + 1 + 2 * (size_t)n_cf) + 64;
}

int match_temporal_cells_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cfL, const ebvo_edge *d_cfR, int n_cf, int cell, int gw,
                                 int gh, void *d_grid)
{
    if (n_cf <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_CAND_COUNT);
    const int n_cells = gw * gh;
    const TemporalGrid g = temporal_grid(d_grid, n_cf, n_cells);
    EBVO_HIP(ctx, hipMemsetAsync(g.cnt, 0, sizeof(int32_t) * 2 * (size_t)n_cells, s.stream)); // cnt and fill
    const unsigned nb = blocks_for(n_cf, 256, 2048);
    hipLaunchKernelGGL(mate_cells_kernel, dim3(nb), dim3(256), 0, s.stream, d_cfL, d_cfR, n_cf, cell, gw, gh, g.cells, g.cnt);
    hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(256), 0, s.stream, g.cnt, n_cells, g.start);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(nb), dim3(256), 0, s.stream, g.cells, n_cf, gw, g.start, g.fill, g.list);
    hipLaunchKernelGGL(cell_sort_kernel, dim3(blocks_for(n_cells, 4, 2048)), dim3(256), 0, s.stream, g.start, n_cells, g.list,
                       g.sorted);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_temporal_candidates_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfL, const ebvo_edge *d_kfR, int n_kf,
                                      const ebvo_edge *d_cfL, const ebvo_edge *d_cfR, const void *d_grid, int n_cf, int cell,
                                      int sr, int gw, int gh, double orient_thr, int32_t *d_cnt, const int32_t *d_row_ptr,
                                      int32_t *d_col_idx, int64_t cap)
{
    if (n_kf <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, d_row_ptr ? K_CAND_FILL : K_CAND_COUNT);
    const TemporalGrid g = temporal_grid(const_cast<void *>(d_grid), n_cf, gw * gh);
    const unsigned nb = blocks_for(n_kf, 16, 8192); // 16 mates per block of 256 threads
    if (d_row_ptr)
        hipLaunchKernelGGL(temporal_candidates_kernel<true>, dim3(nb), dim3(256), 0, s.stream, d_kfL, d_kfR, n_kf, d_cfL, d_cfR,
                           g.cells, g.start, g.sorted, cell, sr, gw, gh, orient_thr, d_cnt, d_row_ptr, d_col_idx, cap);
    else
        hipLaunchKernelGGL(temporal_candidates_kernel<false>, dim3(nb), dim3(256), 0, s.stream, d_kfL, d_kfR, n_kf, d_cfL, d_cfR,
                           g.cells, g.start, g.sorted, cell, sr, gw, gh, orient_thr, d_cnt, d_row_ptr, d_col_idx, cap);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_count_flags_enqueue(ebvo_ctx *ctx, Slot &s, const uint8_t *d_flags, int64_t n, unsigned long long *d_out,
                              const int32_t *d_n)
{
    EBVO_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s.stream));
    if (n > 0)
        hipLaunchKernelGGL(count_flags_kernel, dim3(blocks_for(n, 256, 1024)), dim3(256), 0, s.stream, d_flags, DevCount{n, d_n},
                           d_out);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_ncc_quads_indexed_enqueue(ebvo_ctx *ctx, Slot &s, const float *kfLn, const uint8_t *kfLf, const float *kfRn,
                                    const uint8_t *kfRf, const float *cfLn, const uint8_t *cfLf, const float *cfRn,
                                    const uint8_t *cfRf, const int32_t *d_quad_kf, const int32_t *d_quad_cf, int64_t n_quads,
                                    double thr, double *d_sim_left, double *d_sim_right, uint8_t *d_keep, const int32_t *d_n_quads)
{
    if (n_quads <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_NCC_STORED);
    hipLaunchKernelGGL(ncc_quads_indexed_kernel, dim3(blocks_for(n_quads * 16, 256, 8192)), dim3(256), 0, s.stream, kfLn, kfLf, kfRn,
                       kfRf, cfLn, cfLf, cfRn, cfRf, d_quad_kf, d_quad_cf, DevCount{n_quads, d_n_quads}, thr, d_sim_left, d_sim_right,
                       d_keep);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_pair_result_enqueue(ebvo_ctx *ctx, Slot &s, int cand_cap)
{
    hipLaunchKernelGGL(pair_result_kernel, dim3(1), dim3(1024), 0, s.stream, (const int32_t *)s.im[0].counts,
                       (const int32_t *)s.im[1].counts, (const unsigned long long *)(s.d_total + 1), s.n_total_part,
                       (const int32_t *)s.d_matches, s.n_match_part, s.cap_pairs, cand_cap, s.d_result_host);
    EBVO_HIP(ctx, hipGetLastError());
    // (the record is written straight into the slot's page-locked h_result: it is visible to the host when the event behind the
    // chain has fired; round 3 copied it with one more node)
    return EBVO_OK;
}

int match_expand_rows_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, int64_t n_pairs, int32_t *d_pair_left)
{
    if (nL <= 0 || n_pairs <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks_for(nL, 256, 512)), dim3(256), 0, s.stream, d_row_ptr,
                       DevN{nL, nullptr}, d_pair_left, n_pairs);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int match_ncc_stored_enqueue(ebvo_ctx *ctx, Slot &s, const float *d_A, const float *d_B, int n, double *d_sim)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_NCC_STORED);
    const int64_t threads = (int64_t)n * 16;
    hipLaunchKernelGGL(ncc_stored_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s.stream, d_A, d_B, n,
                       d_sim);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int misc_fp64_peak(ebvo_ctx *ctx, Slot &s, int iters, double *tf_muladd, double *tf_fma)
{
    const int blocks = 256 * 8, threads = 256, inner = 4096;
    double *d_out = nullptr;
    EBVO_HIP(ctx, hipMalloc(&d_out, sizeof(double) * (size_t)blocks * threads));
    hipEvent_t e0, e1;
    EBVO_HIP(ctx, hipEventCreate(&e0));
    EBVO_HIP(ctx, hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
    {
        for (int it = -1; it < iters; ++it)
        {
            if (it == 0)
                EBVO_HIP(ctx, hipEventRecord(e0, s.stream));
            if (mode == 0)
                hipLaunchKernelGGL(fp64_peak_kernel<false>, dim3(blocks), dim3(threads), 0, s.stream, d_out, inner,
                                   1.0000001, 1e-9);
            else
                hipLaunchKernelGGL(fp64_peak_kernel<true>, dim3(blocks), dim3(threads), 0, s.stream, d_out, inner,
                                   1.0000001, 1e-9);
        }
        EBVO_HIP(ctx, hipEventRecord(e1, s.stream));
        EBVO_HIP(ctx, hipEventSynchronize(e1));
        float ms = 0;
        EBVO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        const double flops = 2.0 * 16.0 * (double)inner * (double)blocks * threads * iters;
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (mode == 0 && tf_muladd) *tf_muladd = tf;
        if (mode == 1 && tf_fma) *tf_fma = tf;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d_out);
    return EBVO_OK;
}

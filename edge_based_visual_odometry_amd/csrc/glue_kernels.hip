// glue_kernels.hip -- row-wise stage glue of Stereo_Matches::get_Stereo_Edge_Pairs on CSR candidate lists
// (SURVEY.md 8(f) rank 3), gfx950.  These are small closed-form row operations; they exist on the device so that a
// candidate list does not have to visit the host between the NCC pass, the refinement and the second NCC pass.
//
//   bnb_kernel        apply_Best_Nearly_Best_Test        src/Stereo_Matches.cpp:789-862
//   keep_best_kernel  apply_Lowe_Ratio_Test (as written: keeps only the best)   :916-964
//   cluster_kernel    EdgeClusterer::performClustering per row           src/EdgeClusterer.cpp:119-302
//   shift_kernel      shift_Edge_to_Epipolar_Line for every candidate (consolidate_redundant_edge_hypothesis with
//                     b_do_epipolar_shift, :976-996; Utility::getNormal/TangentialDistance2EpipolarLine,
//                     src/utility.cpp:46-74)
// A selection is returned as new_count[nL] + order[n_pairs]: order[row_ptr[i] + k] is the pair index of the k-th
// survivor of row i.  One thread per row (rows hold a handful of candidates).
#include <hip/hip_runtime.h>

#include "ebvo_internal.h"
#include "ebvo_math.h"
#include "ebvo_sort.h"

namespace
{

// rows of more than 16 candidates: libstdc++'s introsort, restated (ebvo_sort.h); out of line, its explicit recursion stack
// stays out of the common path's registers
__device__ __noinline__ void sort_long_row(int32_t *ord, int n, const double *scores, int higher)
{
    ebvo_sort_cmp c;
    c.score = scores;
    c.descending = higher;
    ebvo_std_sort(ord, n, &c);
}

// the keep count of a sorted row: candidates whose ratio against the BEST (not the previous one, :829) reaches thr
template <class ScoreAt>
__device__ inline int bnb_keep_count(int n, double thr, int higher, ScoreAt score_at)
{
    int keep = 1;
    const double best = score_at(0);
    for (int j = 0; j < n - 1; ++j)
    {
        const double next = score_at(j + 1);
        if (best == 0)
            break;
        const double ratio = higher ? next / best : best / next;
        if (ratio >= thr)
            ++keep;
        else
            break;
    }
    return keep;
}

constexpr int BNB_LONG_CAP = 256; // longest row the cooperative path stages in LDS
constexpr int BNB_QUEUE = 512;    // long rows one block can queue (the rest take the serial path in global memory)

__global__ __launch_bounds__(256) void bnb_kernel(const int32_t *__restrict__ row_ptr, int nL, const double *__restrict__ scores,
                                                  double thr, int higher, int32_t *__restrict__ new_count,
                                                  int32_t *__restrict__ order)
{
    // bit 1 of `higher`: the temporal variant (Temporal_Matches::apply_best_nearly_best_filtering_quads,
    // src/Temporal_Matches.cpp:517-570) rebuilds every row of two or more candidates in SORTED order, whether or not
    // something was dropped; the stereo test leaves an unpruned row untouched (src/Stereo_Matches.cpp:840)
    const bool always_sorted = (higher & 2) != 0;
    higher &= 1;
    // Phase 1, one thread per row: rows of up to 16 candidates (all but a handful) are sorted in LDS -- the scores are
    // fetched once, all loads in flight together; sorting in place in global memory made every step of the insertion
    // sort two dependent round trips (order[j - 1], then its score).  std::sort(indices, comp) (:809-813) runs ONE
    // insertion sort up to 16 entries, which leaves equal scores in their original order.
    __shared__ double s_sc[16][256];
    __shared__ int32_t s_ix[16][256];
    __shared__ int32_t s_queue[BNB_QUEUE];
    __shared__ int s_nq;
    const int t = threadIdx.x;
    if (t == 0)
        s_nq = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
    {
        const int b = row_ptr[i], n = row_ptr[i + 1] - b;
        int32_t *ord = order + b;
        if (n >= 2 && n <= 16)
        {
            for (int k = 0; k < n; ++k)
            {
                s_sc[k][t] = scores[b + k];
                s_ix[k][t] = k;
            }
            for (int k = 1; k < n; ++k)
            {
                const double sv = s_sc[k][t];
                const int32_t v = s_ix[k][t];
                int j = k;
                while (j > 0 && (higher ? sv > s_sc[j - 1][t] : sv < s_sc[j - 1][t]))
                {
                    s_sc[j][t] = s_sc[j - 1][t];
                    s_ix[j][t] = s_ix[j - 1][t];
                    --j;
                }
                s_sc[j][t] = sv;
                s_ix[j][t] = v;
            }
            const int keep = bnb_keep_count(n, thr, higher, [&](int j) { return s_sc[j][t]; });
            const bool sorted_out = keep < n || always_sorted; // nothing dropped: the reference leaves the row untouched (:840)
            new_count[i] = keep < n ? keep : n;
            for (int k = 0; k < n; ++k)
                ord[k] = b + (sorted_out ? s_ix[k][t] : k);
            continue;
        }
        if (n < 2)
        {
            if (n == 1)
                ord[0] = b;
            new_count[i] = n;
            continue;
        }
        // a long row: queued for phase 2; when the queue is full (or the row longer than its staging area), here and now
        const int q = n <= BNB_LONG_CAP ? atomicAdd(&s_nq, 1) : BNB_QUEUE;
        if (q < BNB_QUEUE)
        {
            s_queue[q] = i;
            continue;
        }
        for (int k = 0; k < n; ++k)
            ord[k] = b + k;
        sort_long_row(ord, n, scores, higher);
        const int keep = bnb_keep_count(n, thr, higher, [&](int j) { return scores[ord[j]]; });
        new_count[i] = keep < n ? keep : n;
        if (keep == n && !always_sorted)
            for (int k = 0; k < n; ++k)
                ord[k] = b + k;
    }
    __syncthreads();
    // Phase 2, one WAVE per long row (one thread sorting 30-60 entries in global memory set the kernel's duration: 187 us
    // whatever phase 1 did).  Longer rows go through libstdc++'s introsort, where ties land wherever the partitioning
    // puts them (reproduced move for move by sort_long_row) -- but without equal (or NaN) scores in the row every correct
    // sort returns the same permutation, so the wave ranks the entries in parallel and only a row WITH ties is sorted by
    // lane 0 alone, in LDS.
    __shared__ double s_ls[4][BNB_LONG_CAP];
    __shared__ int32_t s_lo[4][BNB_LONG_CAP];
    const int lane = t & 63, wv = t >> 6;
    const int nq = s_nq < BNB_QUEUE ? s_nq : BNB_QUEUE;
    for (int q = wv; q < nq; q += 4)
    {
        const int i = s_queue[q];
        const int b = row_ptr[i], n = row_ptr[i + 1] - b;
        double *ls = s_ls[wv];
        int32_t *lo = s_lo[wv];
        for (int k = lane; k < n; k += 64)
            ls[k] = scores[b + k];
        __builtin_amdgcn_wave_barrier();
        bool tie = false;
        for (int k = lane; k < n; k += 64)
        {
            const double sk = ls[k];
            int rank = 0;
            tie = tie || sk != sk;
            for (int m = 0; m < n; ++m)
            {
                const double sm = ls[m];
                rank += higher ? sm > sk : sm < sk;
                tie = tie || (m != k && sm == sk);
            }
            if (!tie)
                lo[rank] = k;
        }
        const bool any_tie = __any(tie);
        __builtin_amdgcn_wave_barrier();
        if (any_tie)
        {
            for (int k = lane; k < n; k += 64)
                lo[k] = k;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0)
                sort_long_row(lo, n, ls, higher);
            __builtin_amdgcn_wave_barrier();
        }
        int keep = 0;
        if (lane == 0)
            keep = bnb_keep_count(n, thr, higher, [&](int j) { return ls[lo[j]]; });
        keep = __shfl(keep, 0);
        const bool sorted_out = keep < n || always_sorted;
        if (lane == 0)
            new_count[i] = keep < n ? keep : n;
        for (int k = lane; k < n; k += 64)
            order[b + k] = b + (sorted_out ? lo[k] : k);
        __builtin_amdgcn_wave_barrier(); // the staging area is reused by this wave's next row
    }
}

__global__ void keep_best_kernel(const int32_t *__restrict__ row_ptr, int nL, const double *__restrict__ scores,
                                 int32_t *__restrict__ new_count, int32_t *__restrict__ order)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
    {
        const int b = row_ptr[i], n = row_ptr[i + 1] - b;
        new_count[i] = n ? 1 : 0;
        if (!n)
            continue;
        int best = 0;
        double mx = -1.0;
        for (int j = 0; j < n; ++j)
            if (scores[b + j] > mx)
            {
                mx = scores[b + j];
                best = j;
            }
        order[b] = b + best;
    }
}

// Utility::getTangentialDistance2EpipolarLine, src/utility.cpp:63-74; tan(theta) as sin / cos of the shared correctly
// rounded pair (the oracle's portable mode does the same)
__device__ inline double tangential_dist(double a1, double b1, double c1, double x, double y, double theta, double &xi,
                                         double &yi)
{
    double sn, cs;
    ebvo_sincos(theta, &sn, &cs);
    const double a_e = sn / cs, b_e = -1;
    const double c_e = -(a_e * x - y);
    xi = (b1 * c_e - b_e * c1) / (a1 * b_e - a_e * b1);
    yi = (c1 * a_e - c_e * a1) / (a1 * b_e - a_e * b1);
    return sqrt((xi - x) * (xi - x) + (yi - y) * (yi - y));
}

__global__ void shift_kernel(const ebvo_edge *__restrict__ cand, const double *__restrict__ lines,
                             const int32_t *__restrict__ pair_left, DevCount nd, ebvo_edge *__restrict__ out)
{
    const int64_t n = devcount(nd);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    {
        const int i = pair_left[k];
        const double a1 = lines[(size_t)i * 3], b1 = lines[(size_t)i * 3 + 1], c1 = lines[(size_t)i * 3 + 2];
        ebvo_edge e = cand[k];
        const double x = e.x, y = e.y, th = e.theta;
        e.index = 0;
        e.pad = 0;
        const double ex = x - a1 * (a1 * x + b1 * y + c1) / (a1 * a1 + b1 * b1); // src/utility.cpp:51-52, pow(., 2) = x * x
        const double ey = y - b1 * (a1 * x + b1 * y + c1) / (a1 * a1 + b1 * b1);
        if (sqrt((x - ex) * (x - ex) + (y - ey) * (y - ey)) < 0.4) // LOCATION_PERTURBATION
        {
            e.x = ex;
            e.y = ey;
        }
        else
        {
            double xi, yi;
            if (tangential_dist(a1, b1, c1, x, y, th, xi, yi) < 3) // EPIP_TANGENCY_DISPL_THRESH
            {
                e.x = xi;
                e.y = yi;
            }
            else
            {
                double sn, cs, theta = th;
                ebvo_sincos(theta, &sn, &cs);
                const double p = a1 * cs + b1 * sn, dp = -a1 * sn + b1 * cs; // :60-61
                if (p > 0 && dp < 0)
                    theta -= 0.174533; // ORIENT_PERTURBATION
                else if (p < 0 && dp < 0)
                    theta -= 0.174533;
                else if (p > 0 && dp > 0)
                    theta += 0.174533;
                else if (p < 0 && dp > 0)
                    theta += 0.174533;
                if (tangential_dist(a1, b1, c1, x, y, theta, xi, yi) < 3)
                {
                    e.x = xi;
                    e.y = yi;
                    e.theta = theta;
                }
            }
        }
        out[k] = e;
    }
}

// EdgeClusterer::performClustering for one row per thread (src/EdgeClusterer.cpp:119-302, driven by
// consolidate_redundant_edge_hypothesis :1006-1034).  Merging decisions use sqrt / compare only and equal the
// restatement exactly; the Gaussian weights use csrc/ebvo_math.h's exp, the routine the oracle's portable mode calls, so the
// centres are bit-identical to it (glibc's exp differs by 1 ulp on a fraction of inputs: <= 1e-12 px on a centre).
struct RowPoint
{
    double x, y, theta;
};

template <class EdgeT> // ebvo_edge (global memory) or RowPoint (the row staged in LDS)
__device__ inline void gaussian_average(const EdgeT *__restrict__ E, const int32_t *lab, int n, int label, double &gx,
                                        double &gy, double &gt)
{
    double sx = 0, sy = 0;
    int count = 0;
    for (int i = 0; i < n; ++i)
        if (lab[i] == label)
        {
            sx += E[i].x;
            sy += E[i].y;
            ++count;
        }
    if (!count)
    {
        gx = gy = gt = 0.0;
        return;
    }
    const double cx = sx / count, cy = sy / count;
    double tot = 0.0;
    for (int i = 0; i < n; ++i)
        if (lab[i] == label)
        {
            const double dx = E[i].x - cx, dy = E[i].y - cy;
            tot += sqrt(dx * dx + dy * dy);
        }
    const double mean = tot / count;
    double wx = 0, wy = 0, wt = 0, w = 0;
    for (int i = 0; i < n; ++i)
        if (lab[i] == label)
        {
            const double dx = E[i].x - cx, dy = E[i].y - cy;
            const double d = sqrt(dx * dx + dy * dy);
            const double q = (d - mean) / 2.0; // CLUSTER_ORIENT_GAUSS_SIGMA
            const double g = ebvo_exp(-0.5 * (q * q)); // the shared routine: same bits as the oracle's portable mode
            wx += g * E[i].x;
            wy += g * E[i].y;
            wt += g * E[i].theta;
            w += g;
        }
    gx = wx / w;
    gy = wy / w;
    gt = wt / w;
}

// Sixteen lanes per row.  The merging schedule itself is sequential (every point in turn, restart after a merge), but
// the search for the nearest foreign point, the cluster sizes and the relabelling are loops over the row, and the
// Gaussian averages are independent per cluster: lanes split the candidates (j = lane, lane + 16, ...) and the cluster
// labels.  Every floating-point sum is still accumulated in index order by ONE lane, so the results equal the serial
// form; with one thread per row the few rows with 20-40 candidates (O(n^3) work) set the kernel's duration.
// Rows of more than 64 candidates take the serial path on lane 0 (labels as a 64-bit presence mask otherwise).
__device__ inline void cluster_row_serial(const ebvo_edge *E, int32_t *lab, int n, int by_orientation, double orient_thr,
                                          ebvo_edge *centres, int32_t *new_count)
{
    bool merged = true;
    while (merged)
    {
        merged = false;
        for (int i = 0; i < n && !merged; ++i)
        {
            double min_dist = 1.7976931348623157e308;
            int nearest = -1;
            for (int j = 0; j < n; ++j)
                if (lab[i] != lab[j])
                {
                    const double dx = E[i].x - E[j].x, dy = E[i].y - E[j].y;
                    const double dist = sqrt(dx * dx + dy * dy);
                    if (dist < min_dist && dist < 1 && (!by_orientation || fabs(E[i].theta - E[j].theta) < orient_thr))
                    {
                        min_dist = dist;
                        nearest = j;
                    }
                }
            if (nearest != -1)
            {
                const int old_label = lab[nearest], new_label = lab[i];
                int so = 0, sn = 0;
                for (int k = 0; k < n; ++k)
                {
                    so += lab[k] == old_label;
                    sn += lab[k] == new_label;
                }
                if (so + sn <= 10) // MAX_CLUSTER_SIZE
                {
                    for (int k = 0; k < n; ++k)
                        if (lab[k] == old_label)
                            lab[k] = new_label;
                    merged = true;
                }
            }
        }
    }
    int C = 0;
    for (int l = 0; l < n; ++l)
    {
        bool present = false;
        for (int i = 0; i < n && !present; ++i)
            present = lab[i] == l;
        if (!present)
            continue;
        double gx, gy, gt;
        gaussian_average(E, lab, n, l, gx, gy, gt);
        ebvo_edge c;
        c.x = gx;
        c.y = gy;
        c.theta = gt;
        c.index = 0;
        c.pad = 0;
        centres[C] = c;
        for (int i = 0; i < n; ++i)
            if (lab[i] == l)
                lab[i] = -1 - C;
        ++C;
    }
    for (int i = 0; i < n; ++i)
        lab[i] = -1 - lab[i];
    *new_count = C;
}

__global__ __launch_bounds__(256) void cluster_kernel(const ebvo_edge *__restrict__ cand, const int32_t *__restrict__ row_ptr,
                                                      int nL, int by_orientation, int skip_single,
                                                      int32_t *__restrict__ new_count, ebvo_edge *__restrict__ centres,
                                                      int32_t *__restrict__ cluster_of)
{
    // labels of the parallel path live in LDS (64 per group): lanes of a group read what other lanes wrote in the step
    // before, which LDS orders within a wave
    __shared__ int32_t s_lab[16][64];
    // ... and so does the row itself (x, y, theta of up to 64 candidates per group): the merging loops read every point
    // of the row again and again, from global memory that was a chain of L1 / L2 round trips
    __shared__ RowPoint s_pt[16][64];
    const double orient_thr = 20.0 * 0x1.921fb54442d18p+1 / 180.0; // deg_to_rad(CLUSTER_ORIENT_THRESH): 20 * M_PI / 180
    const int e = threadIdx.x & 15;                                 // lane in the 16-lane group
    const int groups = (gridDim.x * blockDim.x) >> 4;
    // every lane of a wave runs the same number of row iterations (shuffles below need the whole wave converged)
    const int iters = (nL + groups - 1) / groups;
    for (int it = 0; it < iters; ++it)
    {
        const int r = it * groups + ((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
        const bool row_ok = r < nL;
        const int b = row_ok ? row_ptr[r] : 0, n = row_ok ? row_ptr[r + 1] - b : 0;
        const ebvo_edge *Eg = cand + b;
        const RowPoint *E = s_pt[threadIdx.x >> 4];
        int32_t *glab = cluster_of + b;
        int32_t *lab = s_lab[threadIdx.x >> 4];
        if (row_ok && e == 0)
            new_count[r] = n;
        const bool simple = n == 0 || (n == 1 && skip_single);
        if (row_ok && simple && n == 1 && e == 0)
        {
            centres[b] = Eg[0];
            glab[0] = 0;
        }
        const bool serial = !simple && n > 64;
        if (serial && e == 0)
        {
            for (int i = 0; i < n; ++i)
                glab[i] = i;
            cluster_row_serial(Eg, glab, n, by_orientation, orient_thr, centres + b, &new_count[r]);
        }
        const bool par = !simple && !serial; // uniform within the group
        if (par)
            for (int i = e; i < n; i += 16)
            {
                lab[i] = i;
                const ebvo_edge g = Eg[i];
                RowPoint q;
                q.x = g.x;
                q.y = g.y;
                q.theta = g.theta;
                s_pt[threadIdx.x >> 4][i] = q;
            }
        __builtin_amdgcn_wave_barrier();
        // ---- merging: wave-uniform loop, each group on its own row.  The reference restarts its scan at point 0 after
        // every merge and takes the first point (ascending) whose nearest foreign point can be merged.  A point that was
        // examined and could not merge stays that way until ITS OWN cluster changes: the set of points foreign to it is
        // the same, so is its nearest one, and the two cluster sizes of the size test only grow.  So only the members of the
        // cluster a merge has just formed need another look (`dirty`, a bit per point, the same in every lane of the group),
        // and the next point to examine is the lowest dirty one -- the same merges in the same order as the restarting
        // scan, without walking the settled head of the row again after each of them.
        unsigned long long dirty = par ? (n >= 64 ? ~0ull : (1ull << n) - 1ull) : 0ull;
        while (__any(dirty != 0ull))
        {
            const bool searching = dirty != 0ull;
            const int i = searching ? __ffsll((long long)dirty) - 1 : 0;
            double best = 1.7976931348623157e308;
            int nearest = -1, li = 0;
            if (searching)
            {
                li = lab[i];
                const double xi = E[i].x, yi = E[i].y, ti = E[i].theta;
                for (int j = e; j < n; j += 16)
                    if (lab[j] != li)
                    {
                        const double dx = xi - E[j].x, dy = yi - E[j].y;
                        const double dist = sqrt(dx * dx + dy * dy);
                        if (dist < best && dist < 1 && (!by_orientation || fabs(ti - E[j].theta) < orient_thr))
                        {
                            best = dist;
                            nearest = j;
                        }
                    }
            }
            // minimum over the group; ties -> the lowest j, as the serial scan finds it
#pragma unroll
            for (int d = 8; d > 0; d >>= 1)
            {
                const double ob = __shfl_xor(best, d);
                const int on = __shfl_xor(nearest, d);
                const bool take = on >= 0 && (nearest < 0 || ob < best || (ob == best && on < nearest));
                best = take ? ob : best;
                nearest = take ? on : nearest;
            }
            int so = 0, sn = 0, old_label = -1;
            if (searching && nearest >= 0)
            {
                old_label = lab[nearest];
                for (int k = e; k < n; k += 16)
                {
                    so += lab[k] == old_label;
                    sn += lab[k] == li;
                }
            }
#pragma unroll
            for (int d = 8; d > 0; d >>= 1)
            {
                so += __shfl_xor(so, d);
                sn += __shfl_xor(sn, d);
            }
            const bool merge = searching && nearest >= 0 && so + sn <= 10; // MAX_CLUSTER_SIZE
            unsigned lo = 0, hi = 0; // the members of the merged cluster among this lane's points
            if (merge)
                for (int k = e; k < n; k += 16)
                {
                    const int lk = lab[k];
                    if (lk == old_label)
                        lab[k] = li;
                    if (lk == old_label || lk == li)
                    {
                        lo |= k < 32 ? 1u << k : 0u;
                        hi |= k >= 32 ? 1u << (k - 32) : 0u;
                    }
                }
#pragma unroll
            for (int d = 8; d > 0; d >>= 1)
            {
                lo |= (unsigned)__shfl_xor((int)lo, d);
                hi |= (unsigned)__shfl_xor((int)hi, d);
            }
            if (merge)
                dirty |= ((unsigned long long)hi << 32) | lo;
            else if (searching)
                dirty &= ~(1ull << i);
            __builtin_amdgcn_wave_barrier();
        }
        // ---- one Gaussian-weighted average per cluster, clusters in ascending label order; labels renumbered
        unsigned long long present = 0;
        if (par)
            for (int i = 0; i < n; ++i) // n <= 64, uniform in the group
                present |= 1ull << lab[i];
        if (par)
        {
            for (int l = e; l < n; l += 16)
                if ((present >> l) & 1ull)
                {
                    double gx, gy, gt;
                    gaussian_average(E, lab, n, l, gx, gy, gt);
                    ebvo_edge c;
                    c.x = gx;
                    c.y = gy;
                    c.theta = gt;
                    c.index = 0;
                    c.pad = 0;
                    centres[b + __popcll(present & ((1ull << l) - 1ull))] = c;
                }
            if (e == 0)
                new_count[r] = __popcll(present);
        }
        __builtin_amdgcn_wave_barrier();
        if (par) // renumbered labels 0 .. C-1 in ascending label order
            for (int i = e; i < n; i += 16)
                glab[i] = __popcll(present & ((1ull << lab[i]) - 1ull));
        __builtin_amdgcn_wave_barrier(); // the LDS labels are reused by the next row of this group
    }
}

// ---- CSR plumbing of the device-resident chain (ebvo_stereo_finalize) ------------------------------------------------
// rows from per-pair flags: new_count[i] = flagged pairs of row i, order[row_ptr[i] + k] = pair index of the k-th one
__global__ void rows_from_flags_kernel(const int32_t *__restrict__ row_ptr, int nL, const uint8_t *__restrict__ flags,
                                       int32_t *__restrict__ new_count, int32_t *__restrict__ order)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
    {
        int c = 0;
        for (int k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
            if (flags[k])
                order[row_ptr[i] + c++] = k;
        new_count[i] = c;
    }
}

// per-pair data of a selection (new_count, order over the rows of rp_in) packed into the rows of rp_out.
// order == NULL: the first new_count[i] slots of row i (cluster centres).  emap != NULL: edges are E_src[emap[p]].
__global__ void gather_rows_kernel(const int32_t *__restrict__ rp_in, const int32_t *__restrict__ cnt,
                                   const int32_t *__restrict__ order, const int32_t *__restrict__ rp_out, int nL,
                                   const ebvo_edge *__restrict__ E_src, const int32_t *__restrict__ emap,
                                   ebvo_edge *__restrict__ E_dst, const double *__restrict__ D_src,
                                   double *__restrict__ D_dst)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
        for (int k = 0; k < cnt[i]; ++k)
        {
            const int p = order ? order[rp_in[i] + k] : rp_in[i] + k, q = rp_out[i] + k;
            if (E_dst)
            {
                ebvo_edge e = E_src[emap ? emap[p] : p];
                e.index = 0;
                e.pad = 0;
                E_dst[q] = e;
            }
            if (D_dst)
                D_dst[q] = D_src[p];
        }
}

__global__ void edges_to_xy_kernel(const ebvo_edge *__restrict__ e, DevCount nd, double *__restrict__ xy)
{
    const int64_t n = devcount(nd);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    {
        xy[2 * k] = e[k].x;
        xy[2 * k + 1] = e[k].y;
    }
}

__global__ void xy_to_edges_kernel(const double *__restrict__ xy, DevCount nd, ebvo_edge *__restrict__ e)
{
    const int64_t n = devcount(nd);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    {
        e[k].x = xy[2 * k];
        e[k].y = xy[2 * k + 1];
    }
}

// the rows that kept one candidate become the final pairs: left index, left edge, right centre, score
__global__ void final_pairs_kernel(const int32_t *__restrict__ rp_in, const int32_t *__restrict__ cnt,
                                   const int32_t *__restrict__ order, const int32_t *__restrict__ rp_out, int nL,
                                   const ebvo_edge *__restrict__ L, const ebvo_edge *__restrict__ cand,
                                   const double *__restrict__ score, int32_t *__restrict__ left_index,
                                   ebvo_edge *__restrict__ left_edge, ebvo_edge *__restrict__ right_edge,
                                   double *__restrict__ final_score)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
        if (cnt[i] > 0)
        {
            const int p = order[rp_in[i]], q = rp_out[i];
            left_index[q] = i;
            left_edge[q] = L[i];
            right_edge[q] = cand[p];
            final_score[q] = score[p];
        }
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }

} // namespace

// ---- temporal quads after the NCC filter (Temporal_Matches::get_Temporal_Edge_Pairs_from_Quads, :196-215) ---------
namespace
{
// source index of every surviving pair of a row selection: idx[rp_out[i] + k] = order[rp_in[i] + k], k < cnt[i]
__global__ void row_index_kernel(const int32_t *__restrict__ rp_in, const int32_t *__restrict__ cnt,
                                 const int32_t *__restrict__ order, const int32_t *__restrict__ rp_out, int nL,
                                 int32_t *__restrict__ idx)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
        for (int k = 0; k < cnt[i]; ++k)
            idx[rp_out[i] + k] = order[rp_in[i] + k];
}

struct GatherArgs
{
    const int32_t *i_src[4];
    int32_t *i_dst[4];
    const double *d_src[6];
    double *d_dst[6];
    const uint8_t *b_src[2];
    uint8_t *b_dst[2];
    const ebvo_edge *e_src[2];
    ebvo_edge *e_dst[2];
    int ni, nd, nb, ne;
};
__global__ void gather_kernel(GatherArgs G, const int32_t *__restrict__ idx, int64_t n)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    {
        const int32_t k = idx[j];
        for (int a = 0; a < G.ni; ++a)
            G.i_dst[a][j] = G.i_src[a][k];
        for (int a = 0; a < G.nd; ++a)
            G.d_dst[a][j] = G.d_src[a][k];
        for (int a = 0; a < G.nb; ++a)
            G.b_dst[a][j] = G.b_src[a][k];
        for (int a = 0; a < G.ne; ++a)
            G.e_dst[a][j] = G.e_src[a][k];
    }
}

// apply_photometric_refinement_quads, the per-quad inputs of one camera (:600-603): keyframe edge, current-frame edge,
// initial displacement = keyframe location - current-frame location
__global__ void quad_refine_inputs_kernel(const ebvo_edge *__restrict__ kfE, const int32_t *__restrict__ quad_kf,
                                          const ebvo_edge *__restrict__ cfE, const int32_t *__restrict__ quad_cf, int64_t n,
                                          ebvo_edge *__restrict__ kf_out, ebvo_edge *__restrict__ cf_out,
                                          double *__restrict__ init)
{
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
    {
        const ebvo_edge k = kfE[quad_kf[q]], c = cfE[quad_cf[q]];
        kf_out[q] = k;
        cf_out[q] = c;
        init[2 * q] = k.x - c.x;
        init[2 * q + 1] = k.y - c.y;
    }
}

// (:620-631): the cluster centres of a quad move to keyframe location - refined displacement where the refinement of that
// camera is valid; the orientation stays the current-frame mate's; the quad is valid if both cameras are
__global__ void quad_apply_refine_kernel(const ebvo_edge *__restrict__ kfL, const ebvo_edge *__restrict__ cfL,
                                         const double *__restrict__ dispL, const uint8_t *__restrict__ validL,
                                         const ebvo_edge *__restrict__ kfR, const ebvo_edge *__restrict__ cfR,
                                         const double *__restrict__ dispR, const uint8_t *__restrict__ validR, int64_t n,
                                         ebvo_edge *__restrict__ cenL, ebvo_edge *__restrict__ cenR,
                                         uint8_t *__restrict__ valid)
{
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
    {
        ebvo_edge l = cfL[q], r = cfR[q];
        if (validL[q] == 1)
        {
            l.x = kfL[q].x - dispL[2 * q];
            l.y = kfL[q].y - dispL[2 * q + 1];
        }
        if (validR[q] == 1)
        {
            r.x = kfR[q].x - dispR[2 * q];
            r.y = kfR[q].y - dispR[2 * q + 1];
        }
        cenL[q] = l;
        cenR[q] = r;
        valid[q] = (validL[q] == 1 && validR[q] == 1) ? 1 : 0;
    }
}

// apply_temporal_edge_clustering_quads after EdgeClusterer::performClustering (:660-722), one thread per keyframe mate.
// Cluster c of the row: its members in index order; for every member the FIRST candidate of the row at the smallest
// distance from it (the member itself unless an earlier candidate has the same location) supplies the right edge and the
// record the merged quad is copied from (`best_idx` = the last member's); the right centre is that edge, or the plain
// mean of the members' right edges (location and orientation).  Rows of fewer than two candidates stay as they are.
__global__ void quad_cluster_post_kernel(const int32_t *__restrict__ rp_in, int nL, const int32_t *__restrict__ new_count,
                                         const int32_t *__restrict__ cluster_of, const ebvo_edge *__restrict__ centres,
                                         const ebvo_edge *__restrict__ cenL, const ebvo_edge *__restrict__ cenR,
                                         const int32_t *__restrict__ rp_out, ebvo_edge *__restrict__ outL,
                                         ebvo_edge *__restrict__ outR, int32_t *__restrict__ src)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nL; i += gridDim.x * blockDim.x)
    {
        const int b = rp_in[i], n = rp_in[i + 1] - b, o = rp_out[i];
        if (n < 2)
        {
            if (n == 1)
            {
                outL[o] = cenL[b];
                outR[o] = cenR[b];
                src[o] = b;
            }
            continue;
        }
        for (int c = 0; c < new_count[i]; ++c)
        {
            double sx = 0, sy = 0, st = 0;
            int members = 0, best = -1;
            ebvo_edge only;
            for (int m = 0; m < n; ++m)
            {
                if (cluster_of[b + m] != c)
                    continue;
                int closest = -1;
                double dmin = 1.7976931348623157e308; // std::numeric_limits<double>::max()
                for (int k = 0; k < n; ++k)
                {
                    const double dx = cenL[b + m].x - cenL[b + k].x, dy = cenL[b + m].y - cenL[b + k].y;
                    const double d = sqrt(dx * dx + dy * dy); // cv::norm(Point2d)
                    if (d < dmin)
                    {
                        dmin = d;
                        closest = k;
                    }
                }
                if (closest < 0)
                    continue; // a NaN location: no candidate is "closer" (the member contributes nothing, :684)
                const ebvo_edge r = cenR[b + closest];
                sx += r.x;
                sy += r.y;
                st += r.theta;
                only = r;
                ++members;
                best = closest;
            }
            ebvo_edge right = only;
            if (members > 1)
            {
                right.x = sx / members;
                right.y = sy / members;
                right.theta = st / members;
            }
            // (a cluster none of whose members found a closest candidate is skipped by the reference; its slot is marked)
            outL[o + c] = centres[b + c];
            if (best >= 0)
                outR[o + c] = right;
            src[o + c] = best >= 0 ? b + best : -1;
        }
    }
}
} // namespace

int glue_row_index_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                           const int32_t *d_rp_out, int nL, int32_t *d_idx)
{
    if (nL <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(row_index_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_rp_in, d_cnt, d_order, d_rp_out, nL, d_idx);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_gather_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_idx, int64_t n, const GlueGather &g)
{
    if (n <= 0)
        return EBVO_OK;
    GatherArgs G = {};
    for (int a = 0; a < 4 && g.i_src[a]; ++a, ++G.ni)
        G.i_src[a] = g.i_src[a], G.i_dst[a] = g.i_dst[a];
    for (int a = 0; a < 6 && g.d_src[a]; ++a, ++G.nd)
        G.d_src[a] = g.d_src[a], G.d_dst[a] = g.d_dst[a];
    for (int a = 0; a < 2 && g.b_src[a]; ++a, ++G.nb)
        G.b_src[a] = g.b_src[a], G.b_dst[a] = g.b_dst[a];
    for (int a = 0; a < 2 && g.e_src[a]; ++a, ++G.ne)
        G.e_src[a] = g.e_src[a], G.e_dst[a] = g.e_dst[a];
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, G, d_idx, n);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_quad_refine_inputs_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfE, const int32_t *d_quad_kf,
                                    const ebvo_edge *d_cfE, const int32_t *d_quad_cf, int64_t n, ebvo_edge *d_kf_out,
                                    ebvo_edge *d_cf_out, double *d_init)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(quad_refine_inputs_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, d_kfE, d_quad_kf, d_cfE, d_quad_cf, n,
                       d_kf_out, d_cf_out, d_init);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_quad_apply_refine_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_kfL, const ebvo_edge *d_cfL, const double *d_dispL,
                                   const uint8_t *d_validL, const ebvo_edge *d_kfR, const ebvo_edge *d_cfR,
                                   const double *d_dispR, const uint8_t *d_validR, int64_t n, ebvo_edge *d_cenL,
                                   ebvo_edge *d_cenR, uint8_t *d_valid)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(quad_apply_refine_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, d_kfL, d_cfL, d_dispL, d_validL, d_kfR,
                       d_cfR, d_dispR, d_validR, n, d_cenL, d_cenR, d_valid);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_quad_cluster_post_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, int nL, const int32_t *d_new_count,
                                   const int32_t *d_cluster_of, const ebvo_edge *d_centres, const ebvo_edge *d_cenL,
                                   const ebvo_edge *d_cenR, const int32_t *d_rp_out, ebvo_edge *d_outL, ebvo_edge *d_outR,
                                   int32_t *d_src)
{
    if (nL <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(quad_cluster_post_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_rp_in, nL, d_new_count,
                       d_cluster_of, d_centres, d_cenL, d_cenR, d_rp_out, d_outL, d_outR, d_src);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_bnb_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const double *d_scores, double thr,
                     int higher_is_better, int32_t *d_new_count, int32_t *d_order)
{
    if (nL <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(bnb_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_row_ptr, nL, d_scores, thr, higher_is_better,
                       d_new_count, d_order);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_keep_best_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const double *d_scores,
                           int32_t *d_new_count, int32_t *d_order)
{
    if (nL <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(keep_best_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_row_ptr, nL, d_scores, d_new_count,
                       d_order);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_shift_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cand, const double *d_lines, const int32_t *d_pair_left,
                       int64_t n, ebvo_edge *d_out, const int32_t *d_n)
{
    if (n <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(shift_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, d_cand, d_lines, d_pair_left, DevCount{n, d_n},
                       d_out);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_cluster_enqueue(ebvo_ctx *ctx, Slot &s, const ebvo_edge *d_cand, const int32_t *d_row_ptr, int nL,
                         int by_orientation, int skip_single, int32_t *d_new_count, ebvo_edge *d_centres,
                         int32_t *d_cluster_of)
{
    if (nL <= 0)
        return EBVO_OK;
    ProfScope ps(ctx, s, K_MISC);
    hipLaunchKernelGGL(cluster_kernel, dim3(grid_for((int64_t)nL * 16)), dim3(256), 0, s.stream, d_cand, d_row_ptr, nL, by_orientation,
                       skip_single, d_new_count, d_centres, d_cluster_of);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_rows_from_flags_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_row_ptr, int nL, const uint8_t *d_flags,
                                 int32_t *d_new_count, int32_t *d_order)
{
    if (nL <= 0)
        return EBVO_OK;
    hipLaunchKernelGGL(rows_from_flags_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_row_ptr, nL, d_flags,
                       d_new_count, d_order);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_gather_rows_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                             const int32_t *d_rp_out, int nL, const ebvo_edge *d_E_src, const int32_t *d_emap,
                             ebvo_edge *d_E_dst, const double *d_D_src, double *d_D_dst)
{
    if (nL <= 0)
        return EBVO_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_rp_in, d_cnt, d_order, d_rp_out, nL,
                       d_E_src, d_emap, d_E_dst, d_D_src, d_D_dst);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_xy_enqueue(ebvo_ctx *ctx, Slot &s, ebvo_edge *d_edges, double *d_xy, int64_t n, bool to_edges, const int32_t *d_n)
{
    if (n <= 0)
        return EBVO_OK;
    if (to_edges)
        hipLaunchKernelGGL(xy_to_edges_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, (const double *)d_xy, DevCount{n, d_n},
                           d_edges);
    else
        hipLaunchKernelGGL(edges_to_xy_kernel, dim3(grid_for(n)), dim3(256), 0, s.stream, (const ebvo_edge *)d_edges,
                           DevCount{n, d_n}, d_xy);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

int glue_final_pairs_enqueue(ebvo_ctx *ctx, Slot &s, const int32_t *d_rp_in, const int32_t *d_cnt, const int32_t *d_order,
                             const int32_t *d_rp_out, int nL, const ebvo_edge *d_L, const ebvo_edge *d_cand,
                             const double *d_score, int32_t *d_left_index, ebvo_edge *d_left_edge, ebvo_edge *d_right_edge,
                             double *d_final_score)
{
    if (nL <= 0)
        return EBVO_OK;
    hipLaunchKernelGGL(final_pairs_kernel, dim3(grid_for(nL)), dim3(256), 0, s.stream, d_rp_in, d_cnt, d_order, d_rp_out, nL,
                       d_L, d_cand, d_score, d_left_index, d_left_edge, d_right_edge, d_final_score);
    EBVO_HIP(ctx, hipGetLastError());
    return EBVO_OK;
}

// adapters.hpp -- C++17 host-side mirror of the reference's interface for the hot path, over the
// C ABI of ebvo_hip.h.  Header-only; depends on the STL and ebvo_hip.h only, so it compiles with or
// without OpenCV / Eigen.  integration/*.cpp shows the reference-side translation units built on it.
//
//   ebvo::ThirdOrderEdgeDetectionHIP<EdgeT>  -- same public surface as ThirdOrderEdgeDetectionCPU
//                                               (include/toed/cpu_toed.hpp:70-115 of the reference)
//   ebvo::StereoMatcherHIP<EdgeT>            -- the bodies of apply_Epipolar_Line_Distance_Filtering,
//                                               apply_Disparity_Filtering, apply_orientation_filter and
//                                               apply_NCC_Filtering (include/Stereo_Matches.h:61-68) on
//                                               plain vectors; the reference-side glue copies to/from
//                                               Stereo_Edge_Pairs (include/Dataset.h:180-289)
//   ebvo::undistort / sift_descriptors / sift_min_distances -- cv::undistort (src/Pipeline.cpp:78-79) and the cv::SIFT
//                                               calls of augment_Edge_Data / apply_SIFT_filtering (src/Stereo_Matches.cpp:655-787)
//   ebvo::patch_similarity / ncc_quads       -- Utility::get_patch_similarity (src/utility.cpp:163-180),
//                                               Temporal_Matches::apply_NCC_filtering_quads scoring and the
//                                               MatlabNCCComputer::computeNCC-shaped entry
//
// EdgeT is the reference's `struct Edge` (or anything with .location.x, .location.y, .orientation,
// .index): TOED fills exactly those fields and leaves the rest as the default constructor set them
// (b_isEmpty = true, frame_source = -1), like src/toed/cpu_toed.cpp:557-563.
//
// Errors: the reference prints and continues; these adapters print the library's message to stderr and
// leave outputs empty, and additionally expose last_status for callers that want to check.
#ifndef EBVO_ADAPTERS_HPP
#define EBVO_ADAPTERS_HPP

#include <array>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../ebvo_hip.h"

namespace ebvo
{

// One shared device context per (device, size): the reference creates one TOED object per Pipeline and
// the matchers have no state of their own, so they borrow the TOED object's context.
class Context
{
  public:
    Context(int max_h, int max_w, int device = 0) : h_(max_h), w_(max_w)
    {
        status_ = ebvo_ctx_create(device, max_h, max_w, &ctx_);
        if (status_ != EBVO_OK)
            std::fprintf(stderr, "\033[1;31m[ERROR] ebvo_ctx_create: %s\033[0m\n", ebvo_strerror(status_));
    }
    ~Context() { ebvo_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    ebvo_ctx *get() const { return ctx_; }
    int status() const { return status_; }
    int max_h() const { return h_; }
    int max_w() const { return w_; }
    typedef std::shared_ptr<Context> Ptr;

    // Resident stage-wise path: what the last two get_Third_Order_Edges calls produced is still on the device (one result
    // per image workspace of the library, used alternately: main_VO detects left, then right, src/Pipeline.cpp:93-97) and
    // in page-locked memory (`edges`).  A later stage that is handed a vector with exactly these edges names the device copy
    // by its tag instead of uploading the vector.
    struct ResidentEdges
    {
        uint64_t tag = 0; // 0: nothing
        const ebvo_edge *edges = nullptr;
        int n = 0;
    };
    ResidentEdges resident[2];
    int next_workspace = 0;

  private:
    ebvo_ctx *ctx_ = nullptr;
    int status_ = EBVO_OK, h_, w_;
};

inline bool report(const Context &c, int rc, const char *where)
{
    if (rc == EBVO_OK)
        return true;
    std::fprintf(stderr, "\033[1;31m[ERROR] %s: %s (%s)\033[0m\n", where, ebvo_strerror(rc),
                 c.get() ? ebvo_last_error(c.get()) : "");
    return false;
}

template <class EdgeT>
inline ebvo_edge to_abi(const EdgeT &e)
{
    ebvo_edge o;
    o.x = e.location.x;
    o.y = e.location.y;
    o.theta = e.orientation;
    o.index = e.index;
    o.pad = 0;
    return o;
}

// read-only view of an array the library or a result object owns
template <class T>
struct Span
{
    const T *p = nullptr;
    size_t n = 0;
    Span() = default;
    Span(const T *p_, size_t n_) : p(p_), n(n_) {}
    Span(const std::vector<T> &v) : p(v.data()), n(v.size()) {}
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const T &operator[](size_t k) const { return p[k]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
};

inline bool same_bits(double a, double b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// the workspace whose resident edge list is EXACTLY `edges` (every x, y, orientation bit for bit, every index), or -1
template <class EdgeT>
inline int resident_workspace(const Context &c, const std::vector<EdgeT> &edges)
{
    for (int ws = 0; ws < 2; ++ws)
    {
        const Context::ResidentEdges &r = c.resident[ws];
        if (!r.tag || (size_t)r.n != edges.size())
            continue;
        size_t k = 0;
        for (; k < edges.size(); ++k)
        {
            const ebvo_edge &a = r.edges[k];
            const EdgeT &e = edges[k];
            if (!(same_bits(a.x, e.location.x) && same_bits(a.y, e.location.y) && same_bits(a.theta, e.orientation) &&
                  a.index == (int32_t)e.index))
                break;
        }
        if (k == edges.size())
            return ws;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
template <class EdgeT>
class ThirdOrderEdgeDetectionHIP
{
  public:
    // public data members of ThirdOrderEdgeDetectionCPU (include/toed/cpu_toed.hpp:90-114)
    double *subpix_edge_pts_final = nullptr; // N x 4: x, y, orientation, sub-pixel gradient magnitude
    int edge_pt_list_idx = 0;
    int num_of_edge_data = 4;
    int omp_threads = 1;
    double time_conv = 0, time_nms = 0;
    std::vector<EdgeT> toed_edges;
    int Total_Num_Of_TOED = 0;
    int last_status = EBVO_OK;
    typedef std::shared_ptr<ThirdOrderEdgeDetectionHIP> Ptr;

    ThirdOrderEdgeDetectionHIP(int H, int W, int device = 0)
        : img_height(H), img_width(W), ctx_(std::make_shared<Context>(H, W, device))
    {
        all4_.resize(4); // subpix_edge_pts_final of an image without a single maximum
        subpix_edge_pts_final = all4_.data();
        last_status = ctx_->status();
    }

    // get_Third_Order_Edges(cv::Mat): any Mat-like with .data (uchar*), .rows, .cols, .step
    template <class MatT>
    void get_Third_Order_Edges(const MatT &img)
    {
        get_Third_Order_Edges(reinterpret_cast<const uint8_t *>(img.data), img.rows, img.cols, (ptrdiff_t)img.step);
    }

    void get_Third_Order_Edges(const uint8_t *data, int rows, int cols, ptrdiff_t step)
    {
        toed_edges.clear(); // preprocessing(), src/toed/cpu_toed.cpp:86
        Total_Num_Of_TOED = 0;
        edge_pt_list_idx = 0;
        if (rows != img_height || cols != img_width)
        { // the reference object is sized once from the left image (include/Pipeline.h:89-101)
            std::fprintf(stderr, "\033[1;31m[ERROR] TOED: image %dx%d does not match %dx%d\033[0m\n", cols, rows,
                         img_width, img_height);
            last_status = EBVO_ERR_ARG;
            return;
        }
        // the result stays on the device (workspace `ws`) and arrives in page-locked memory: toed_edges is built straight
        // from it, subpix_edge_pts_final points at it (valid until the next call but one, like the reference's member buffer
        // is valid until the next call)
        const int ws = ctx_->next_workspace;
        ctx_->next_workspace ^= 1;
        ctx_->resident[ws] = Context::ResidentEdges();
        ebvo_toed_view v;
        last_status = ebvo_toed_resident(ctx_->get(), ws, data, rows, cols, step, 1, &v);
        if (!report(*ctx_, last_status, "ebvo_toed_resident"))
            return;
        time_conv = v.t_conv;
        time_nms = v.t_nms;
        const int n_kept = v.n_kept, n_total = v.n_total;
        toed_edges.reserve((size_t)n_kept);
        EdgeT e{}; // default-constructed: b_isEmpty = true, frame_source = -1
        for (int k = 0; k < n_kept; ++k)
        {
            e.location.x = v.edges[k].x;
            e.location.y = v.edges[k].y;
            e.orientation = v.edges[k].theta;
            e.index = v.edges[k].index;
            toed_edges.push_back(e);
        }
        subpix_edge_pts_final = n_total ? const_cast<double *>(v.all4) : all4_.data();
        ctx_->resident[ws].tag = v.tag;
        ctx_->resident[ws].edges = v.edges;
        ctx_->resident[ws].n = n_kept;
        Total_Num_Of_TOED = n_total;
        edge_pt_list_idx = n_total;
    }

    const Context::Ptr &context() const { return ctx_; }

  private:
    int img_height, img_width;
    Context::Ptr ctx_;
    std::vector<double> all4_;
};

// ---------------------------------------------------------------------------------------------
// Candidate lists as the reference keeps them per focused edge: indices into the candidate edge
// vector, ascending (EdgeCluster::contributing_edges_toed_indices[0], src/Stereo_Matches.cpp:413).
struct CandidateLists
{
    std::vector<int32_t> row_ptr; // nL + 1
    std::vector<int32_t> col_idx; // right TOED index per pair
    size_t rows() const { return row_ptr.empty() ? 0 : row_ptr.size() - 1; }
};

struct NccScores
{
    std::vector<double> pp_nn_pn_np; // 4 per pair (src/Stereo_Matches.cpp:592-595)
    std::vector<double> best;        // final_SIM_score (:596)
    std::vector<uint8_t> keep;       // best > NCC_THRESH (:597)
    std::vector<float> left_patches; // nL x 2 x 49 (left_edge_patches, :578)
};

template <class EdgeT>
class StereoMatcherHIP
{
  public:
    explicit StereoMatcherHIP(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
    int last_status = EBVO_OK;

    // Stereo_Matches::CalculateEpipolarLine with a row-major 3x3 F (src/Stereo_Matches.cpp:10-20).
    static std::vector<std::array<double, 3>> CalculateEpipolarLine(const double F[9], const std::vector<EdgeT> &edges)
    {
        std::vector<std::array<double, 3>> lines(edges.size());
        static thread_local std::vector<ebvo_edge> abi; // reused: a fresh 4 MB vector per frame costs more than the lines
        abi.resize(edges.size());
        for (size_t k = 0; k < edges.size(); ++k)
            abi[k] = to_abi(edges[k]);
        ebvo_epipolar_lines(F, abi.data(), (int)abi.size(), lines.empty() ? nullptr : lines[0].data());
        return lines;
    }

    // stage_mask = EBVO_STAGE_EPIPOLAR            -> apply_Epipolar_Line_Distance_Filtering (:381-419)
    //              ... | EBVO_STAGE_DISPARITY     -> + apply_Disparity_Filtering (:534-553)
    //              ... | EBVO_STAGE_ORIENTATION   -> + apply_orientation_filter (:863-915)
    // The three filters are independent predicates on (left, right) and each keeps the list order, so
    // applying them fused equals applying them one after the other.
    CandidateLists candidates(const std::vector<EdgeT> &left, const std::vector<EdgeT> &right,
                              const std::vector<std::array<double, 3>> &lines, int stage_mask = EBVO_STAGE_ALL,
                              double epi_thr = EBVO_EPIPOLAR_LINE_DIST_THRESH, double max_disp = EBVO_MAX_DISPARITY,
                              double orient_thr_deg = EBVO_ORIENT_THRESH_DEG)
    {
        CandidateLists out;
        std::vector<ebvo_edge> L(left.size()), R(right.size());
        for (size_t k = 0; k < left.size(); ++k)
            L[k] = to_abi(left[k]);
        for (size_t k = 0; k < right.size(); ++k)
            R[k] = to_abi(right[k]);
        out.row_ptr.assign(left.size() + 1, 0);
        int64_t n = 0;
        const double *ln = lines.empty() ? nullptr : lines[0].data();
        // one search when the list fits a first guess of 24 candidates per edge (the reported size is exact otherwise)
        int64_t cap = stage_mask == EBVO_STAGE_EPIPOLAR ? 0 : (int64_t)(24 * left.size() + 1024);
        for (int attempt = 0; attempt < 2; ++attempt)
        {
            out.col_idx.resize((size_t)cap);
            last_status = ebvo_epi_candidates(ctx_->get(), L.data(), (int)L.size(), R.data(), (int)R.size(), ln, epi_thr, max_disp,
                                              orient_thr_deg, stage_mask, out.row_ptr.data(), cap ? out.col_idx.data() : nullptr,
                                              cap, &n);
            if (last_status != EBVO_ERR_CAPACITY && !(cap == 0 && last_status == EBVO_OK && n > 0))
                break;
            cap = n;
        }
        out.col_idx.resize(last_status == EBVO_OK ? (size_t)n : 0);
        report(*ctx_, last_status, "ebvo_epi_candidates");
        return out;
    }

    // The three geometric stages of get_Stereo_Edge_Pairs (:1374, :1387, :1399) from ONE device search: the list under the
    // epipolar and disparity predicates + orient_ok[k] = pair k also passes apply_orientation_filter.  A stage-wise caller
    // fills matching_edge_clusters from the list at the first stage (the epipolar-only list, hundreds of candidates per
    // edge, is never formed), has nothing to do at the second and keeps the flagged pairs at the third.
    struct StagedCandidates
    {
        CandidateLists lists;
        std::vector<uint8_t> orient_ok;
    };
    StagedCandidates candidates_staged(const std::vector<EdgeT> &left, const std::vector<EdgeT> &right,
                                       const std::vector<std::array<double, 3>> &lines,
                                       double epi_thr = EBVO_EPIPOLAR_LINE_DIST_THRESH, double max_disp = EBVO_MAX_DISPARITY,
                                       double orient_thr_deg = EBVO_ORIENT_THRESH_DEG)
    {
        StagedCandidates out;
        std::vector<ebvo_edge> L(left.size()), R(right.size());
        for (size_t k = 0; k < left.size(); ++k)
            L[k] = to_abi(left[k]);
        for (size_t k = 0; k < right.size(); ++k)
            R[k] = to_abi(right[k]);
        out.lists.row_ptr.assign(left.size() + 1, 0);
        int64_t n = 0, cap = (int64_t)(24 * left.size() + 1024);
        const double *ln = lines.empty() ? nullptr : lines[0].data();
        for (int attempt = 0; attempt < 2; ++attempt)
        {
            out.lists.col_idx.resize((size_t)cap);
            out.orient_ok.resize((size_t)cap);
            last_status = ebvo_epi_candidates_staged(ctx_->get(), L.data(), (int)L.size(), R.data(), (int)R.size(), ln, epi_thr,
                                                     max_disp, orient_thr_deg, out.lists.row_ptr.data(), out.lists.col_idx.data(),
                                                     out.orient_ok.data(), cap, &n);
            if (last_status != EBVO_ERR_CAPACITY)
                break;
            cap = n;
        }
        out.lists.col_idx.resize(last_status == EBVO_OK ? (size_t)n : 0);
        out.orient_ok.resize(out.lists.col_idx.size());
        report(*ctx_, last_status, "ebvo_epi_candidates_staged");
        return out;
    }

    // apply_NCC_Filtering (:555-616): candidates are explicit edges (TOED edges in the first pass,
    // cluster centres in the second); images are the RAW left / right images (:562-563).
    NccScores ncc(const uint8_t *imgL, const uint8_t *imgR, int rows, int cols, ptrdiff_t stepL, ptrdiff_t stepR,
                  const std::vector<EdgeT> &left, const std::vector<int32_t> &row_ptr,
                  const std::vector<EdgeT> &candidate_per_pair, double thr = EBVO_NCC_THRESH)
    {
        NccScores s;
        std::vector<ebvo_edge> L(left.size()), Rc(candidate_per_pair.size());
        for (size_t k = 0; k < left.size(); ++k)
            L[k] = to_abi(left[k]);
        for (size_t k = 0; k < Rc.size(); ++k)
            Rc[k] = to_abi(candidate_per_pair[k]);
        s.pp_nn_pn_np.resize(Rc.size() * 4);
        s.best.resize(Rc.size());
        s.keep.resize(Rc.size());
        s.left_patches.resize(left.size() * 98);
        last_status = ebvo_ncc_pairs(ctx_->get(), imgL, imgR, rows, cols, stepL, stepR, L.data(), (int)L.size(),
                                     Rc.data(), row_ptr.data(), thr, s.left_patches.data(), s.pp_nn_pn_np.data(),
                                     s.best.data(), s.keep.data());
        report(*ctx_, last_status, "ebvo_ncc_pairs");
        return s;
    }

    // ---- the same stages for a caller that hands back the vectors get_Third_Order_Edges gave it (main_VO as it is) ----
    // If `left` and `right` are exactly the edge lists still resident on the device (Context::resident), the stage runs on
    // the device copies and its results are views of page-locked memory of the context -- no edge list is uploaded, no result
    // vector is allocated; they stay valid until the next call of the same stage.  Otherwise (edges edited, other edges, a
    // call in between that used the library's slot 0) the host-buffer path above runs and the views point into `own`.
    // Same bits either way (tests/test_cpp_adapter.py).
    struct StagedView
    {
        Span<int32_t> row_ptr, col_idx;             // the lists after the epipolar and disparity stages
        Span<uint8_t> orient_ok;                    // per listed pair: passes the orientation stage
        Span<int32_t> row_ptr_final, col_idx_final; // the lists after the orientation stage (the flagged pairs, in order)
        bool resident = false;
        StagedCandidates own;
        CandidateLists own_final;
        StagedView() = default;
        StagedView(StagedView &&) = default;
        StagedView &operator=(StagedView &&) = default;
        StagedView(const StagedView &) = delete;
        size_t rows() const { return row_ptr.empty() ? 0 : row_ptr.size() - 1; }
    };
    StagedView candidates_staged_view(const std::vector<EdgeT> &left, const std::vector<EdgeT> &right,
                                      const std::vector<std::array<double, 3>> &lines,
                                      double epi_thr = EBVO_EPIPOLAR_LINE_DIST_THRESH, double max_disp = EBVO_MAX_DISPARITY,
                                      double orient_thr_deg = EBVO_ORIENT_THRESH_DEG)
    {
        StagedView out;
        const int wl = resident_workspace(*ctx_, left), wr = resident_workspace(*ctx_, right);
        if (wl >= 0 && wr >= 0 && wl != wr && lines.size() == left.size())
        {
            ebvo_candidates_view v;
            last_status = ebvo_epi_candidates_resident(ctx_->get(), ctx_->resident[wl].tag, ctx_->resident[wr].tag,
                                                       lines.empty() ? nullptr : lines[0].data(), epi_thr, max_disp, orient_thr_deg,
                                                       EBVO_STAGE_EPIPOLAR | EBVO_STAGE_DISPARITY, 1, &v);
            if (last_status == EBVO_OK)
            {
                out.resident = true;
                out.row_ptr = Span<int32_t>(v.row_ptr, left.size() + 1);
                out.col_idx = Span<int32_t>(v.col_idx, (size_t)v.n_pairs);
                out.orient_ok = Span<uint8_t>(v.orient_ok, (size_t)v.n_pairs);
                out.row_ptr_final = Span<int32_t>(v.row_ptr_final, left.size() + 1);
                out.col_idx_final = Span<int32_t>(v.col_idx_final, (size_t)v.n_final);
                return out;
            }
            if (last_status != EBVO_ERR_STATE) // a stale tag is not an error: the host-buffer path takes over
            {
                report(*ctx_, last_status, "ebvo_epi_candidates_resident");
                return out;
            }
            ctx_->resident[0] = ctx_->resident[1] = Context::ResidentEdges();
        }
        out.own = candidates_staged(left, right, lines, epi_thr, max_disp, orient_thr_deg);
        out.row_ptr = out.own.lists.row_ptr;
        out.col_idx = out.own.lists.col_idx;
        out.orient_ok = out.own.orient_ok;
        // the flagged pairs of every row, in order (branch-free: a slot is overwritten unless its pair is flagged)
        out.own_final.row_ptr.assign(out.row_ptr.size(), 0);
        out.own_final.col_idx.resize(out.col_idx.size());
        size_t n = 0;
        for (size_t i = 0; i + 1 < out.row_ptr.size(); ++i)
        {
            for (int32_t k = out.row_ptr[i]; k < out.row_ptr[i + 1]; ++k)
            {
                out.own_final.col_idx[n] = out.col_idx[(size_t)k];
                n += out.orient_ok[(size_t)k];
            }
            out.own_final.row_ptr[i + 1] = (int32_t)n;
        }
        out.own_final.col_idx.resize(n);
        out.row_ptr_final = out.own_final.row_ptr;
        out.col_idx_final = out.own_final.col_idx;
        return out;
    }

    // apply_NCC_Filtering in its first pass (:1427): candidate k of the lists is right[col_idx[k]] (a TOED edge,
    // contributing_edges_toed_indices[0], :413).  The four scores per pair only if asked for.
    struct NccView
    {
        Span<float> left_patches;     // nL x 2 x 49 (left_edge_patches, :578)
        Span<double> pp_nn_pn_np;     // 4 per pair, empty unless want_sims
        Span<double> best;            // final_SIM_score (:596)
        Span<uint8_t> keep;           // best > NCC_THRESH (:597)
        bool resident = false;
        NccScores own;
        NccView() = default;
        NccView(NccView &&) = default;
        NccView &operator=(NccView &&) = default;
        NccView(const NccView &) = delete;
    };
    NccView ncc_indexed(const uint8_t *imgL, const uint8_t *imgR, int rows, int cols, ptrdiff_t stepL, ptrdiff_t stepR,
                        const std::vector<EdgeT> &left, const std::vector<EdgeT> &right, Span<int32_t> row_ptr,
                        Span<int32_t> col_idx, double thr = EBVO_NCC_THRESH, bool want_sims = false)
    {
        NccView out;
        if (row_ptr.size() != left.size() + 1 || (left.size() && (size_t)row_ptr[left.size()] != col_idx.size()))
        {
            last_status = EBVO_ERR_ARG;
            report(*ctx_, last_status, "ncc_indexed: row_ptr / col_idx do not describe one list per left edge");
            return out;
        }
        const int wl = resident_workspace(*ctx_, left), wr = resident_workspace(*ctx_, right);
        if (wl >= 0 && wr >= 0 && wl != wr)
        {
            ebvo_ncc_view v;
            last_status = ebvo_ncc_pairs_resident(ctx_->get(), ctx_->resident[wl].tag, ctx_->resident[wr].tag, imgL, imgR, rows, cols,
                                                  stepL, stepR, row_ptr.data(), col_idx.data(), thr,
                                                  EBVO_NCC_WANT_LEFT_PATCHES | (want_sims ? EBVO_NCC_WANT_SIMS : 0), &v);
            if (last_status == EBVO_OK)
            {
                out.resident = true;
                out.left_patches = Span<float>(v.left_patches, (size_t)v.n_left * 98);
                out.pp_nn_pn_np = Span<double>(v.sims, v.sims ? (size_t)v.n_pairs * 4 : 0);
                out.best = Span<double>(v.best, (size_t)v.n_pairs);
                out.keep = Span<uint8_t>(v.keep, (size_t)v.n_pairs);
                return out;
            }
            if (last_status != EBVO_ERR_STATE)
            {
                report(*ctx_, last_status, "ebvo_ncc_pairs_resident");
                return out;
            }
            ctx_->resident[0] = ctx_->resident[1] = Context::ResidentEdges();
        }
        std::vector<EdgeT> cand(col_idx.size());
        for (size_t k = 0; k < cand.size(); ++k)
        {
            if ((size_t)col_idx[k] >= right.size())
            {
                last_status = EBVO_ERR_ARG;
                report(*ctx_, last_status, "ncc_indexed: candidate index outside the right edge list");
                return out;
            }
            cand[k] = right[(size_t)col_idx[k]];
        }
        std::vector<int32_t> rp(row_ptr.begin(), row_ptr.end());
        out.own = ncc(imgL, imgR, rows, cols, stepL, stepR, left, rp, cand, thr);
        out.left_patches = out.own.left_patches;
        if (want_sims)
            out.pp_nn_pn_np = out.own.pp_nn_pn_np;
        out.best = out.own.best;
        out.keep = out.own.keep;
        return out;
    }

    // refine_edge_disparity (:1290-1358): photometric Gauss-Newton of every (left edge, candidate centre) pair along
    // the epipolar line.  Images are the undistorted left / right images; pass them swapped for is_left = false.
    struct Refined
    {
        std::vector<double> alpha, score, confidence, xy; // xy: n_pairs x 2 updated centre locations (:1349-1351)
        std::vector<uint8_t> validity;                    // refine_validities; 2 = left undefined by the reference
        std::vector<int32_t> iters;
    };
    Refined refine(const uint8_t *imgL, const uint8_t *imgR, int rows, int cols, ptrdiff_t stepL, ptrdiff_t stepR,
                   const std::vector<EdgeT> &left, const std::vector<std::array<double, 3>> &lines,
                   const std::vector<int32_t> &row_ptr, const std::vector<double> &candidate_xy)
    {
        Refined r;
        const size_t n = candidate_xy.size() / 2;
        std::vector<ebvo_edge> L(left.size());
        for (size_t k = 0; k < left.size(); ++k)
            L[k] = to_abi(left[k]);
        r.alpha.resize(n);
        r.score.resize(n);
        r.confidence.resize(n);
        r.xy.resize(2 * n);
        r.validity.resize(n);
        r.iters.resize(n);
        ebvo_gn_params p;
        ebvo_gn_default_params(&p);
        last_status = ebvo_gn_refine_stereo(ctx_->get(), imgL, imgR, rows, cols, stepL, stepR, L.data(), (int)L.size(),
                                            lines.empty() ? nullptr : lines[0].data(), row_ptr.data(), candidate_xy.data(),
                                            &p, r.alpha.data(), r.score.data(), r.confidence.data(), r.validity.data(),
                                            r.iters.data(), r.xy.data());
        report(*ctx_, last_status, "ebvo_gn_refine_stereo");
        return r;
    }

    // ---- stage glue on flattened candidate lists (row_ptr + one entry per candidate) ------------------------------
    struct Selection
    {
        std::vector<int32_t> new_count; // survivors per row
        std::vector<int32_t> order;     // order[row_ptr[i] + k] = index of the k-th survivor of row i
    };
    // apply_Best_Nearly_Best_Test (:789-862): is_NCC -> scores = refine_final_scores, else refine_confidences
    Selection bnb_test(const std::vector<int32_t> &row_ptr, const std::vector<double> &scores, double ratio, bool is_NCC)
    {
        Selection sel;
        sel.new_count.assign(row_ptr.size() - 1, 0);
        sel.order.assign(scores.size(), -1);
        last_status = ebvo_bnb_test(ctx_->get(), row_ptr.data(), (int)row_ptr.size() - 1, scores.data(), ratio, is_NCC ? 1 : 0,
                                    sel.new_count.data(), sel.order.data());
        report(*ctx_, last_status, "ebvo_bnb_test");
        return sel;
    }
    // apply_Lowe_Ratio_Test as written (:916-964): the best candidate of every row
    Selection keep_best(const std::vector<int32_t> &row_ptr, const std::vector<double> &scores)
    {
        Selection sel;
        sel.new_count.assign(row_ptr.size() - 1, 0);
        sel.order.assign(scores.size(), -1);
        last_status = ebvo_keep_best(ctx_->get(), row_ptr.data(), (int)row_ptr.size() - 1, scores.data(), sel.new_count.data(),
                                     sel.order.data());
        report(*ctx_, last_status, "ebvo_keep_best");
        return sel;
    }
    // consolidate_redundant_edge_hypothesis, shift pass (:976-996): shift_Edge_to_Epipolar_Line of every candidate
    std::vector<ebvo_edge> epipolar_shift(const std::vector<EdgeT> &candidates, const std::vector<std::array<double, 3>> &lines,
                                          const std::vector<int32_t> &row_ptr)
    {
        std::vector<ebvo_edge> in(candidates.size()), out(candidates.size());
        for (size_t k = 0; k < in.size(); ++k)
            in[k] = to_abi(candidates[k]);
        last_status = ebvo_epipolar_shift(ctx_->get(), in.data(), lines.empty() ? nullptr : lines[0].data(), row_ptr.data(),
                                          (int)row_ptr.size() - 1, out.data());
        report(*ctx_, last_status, "ebvo_epipolar_shift");
        return out;
    }
    // consolidate_redundant_edge_hypothesis, clustering pass (:1006-1034): EdgeClusterer per row
    struct Clusters
    {
        std::vector<int32_t> new_count, cluster_of; // clusters per row; cluster index of every input candidate
        std::vector<ebvo_edge> centres;             // centres[row_ptr[i] + c], c < new_count[i]
    };
    Clusters cluster_rows(const std::vector<ebvo_edge> &candidates, const std::vector<int32_t> &row_ptr, bool by_orientation,
                          bool skip_single)
    {
        Clusters c;
        c.new_count.assign(row_ptr.size() - 1, 0);
        c.cluster_of.assign(candidates.size(), -1);
        c.centres.resize(candidates.size());
        last_status = ebvo_cluster_rows(ctx_->get(), candidates.data(), row_ptr.data(), (int)row_ptr.size() - 1,
                                        by_orientation ? 1 : 0, skip_single ? 1 : 0, c.new_count.data(), c.centres.data(),
                                        c.cluster_of.data());
        report(*ctx_, last_status, "ebvo_cluster_rows");
        return c;
    }

    // Temporal_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton (src/Temporal_Matches.cpp:735-851) over n
    // (keyframe edge, current-frame edge, initial disparity) items of ONE camera, as apply_photometric_refinement_quads
    // issues them for the left and for the right image of every candidate quad (:605-610).
    struct RefinedTemporal
    {
        std::vector<double> disp, score; // disp: n x 2 refined_disparity; the new location is kf.location - disp
        std::vector<uint8_t> validity;
        std::vector<int32_t> iters;
    };
    RefinedTemporal refine_temporal(const uint8_t *imgKF, const uint8_t *imgCF, int rows, int cols, ptrdiff_t stepKF,
                                    ptrdiff_t stepCF, const std::vector<EdgeT> &kf, const std::vector<EdgeT> &cf,
                                    const std::vector<double> &init_disp)
    {
        RefinedTemporal r;
        const size_t n = kf.size();
        std::vector<ebvo_edge> K(n), C(n);
        for (size_t k = 0; k < n; ++k)
        {
            K[k] = to_abi(kf[k]);
            C[k] = to_abi(cf[k]);
        }
        r.disp.resize(2 * n);
        r.score.resize(n);
        r.validity.resize(n);
        r.iters.resize(n);
        ebvo_gn_params p;
        ebvo_gn_default_params(&p);
        last_status = ebvo_gn_refine_temporal(ctx_->get(), imgKF, imgCF, rows, cols, stepKF, stepCF, K.data(), C.data(),
                                              init_disp.data(), (int)n, &p, r.disp.data(), r.score.data(),
                                              r.validity.data(), r.iters.data());
        report(*ctx_, last_status, "ebvo_gn_refine_temporal");
        return r;
    }

    // Stereo_Matches::get_Stereo_Edge_Pairs (src/Stereo_Matches.cpp:1360-1540) for one frame in ONE pass over the device:
    // both TOED runs, the three geometric filters, [SIFT filter,] NCC, both Best-Nearly-Best tests, epipolar shift,
    // photometric refinement, shift + clustering, second NCC pass, best candidate per row -- images up, final pairs down,
    // every intermediate list stays in HBM (ebvo_stereo_upload_slot / submit / wait / finalize / fetch_final).  This is the
    // path a frame loop takes when it does not inspect the intermediate `matching_edge_clusters` between the stages; the
    // stage-wise calls above remain for a caller that does.
    struct FinalPairs
    {
        ebvo_stereo_counts stage1{};   // TOED edges, candidate pairs after the geometric filters, NCC matches
        ebvo_finalize_counts stages{}; // survivors of the later stages
        std::vector<ebvo_edge> left_edges;  // all kept TOED edges of the left image (stereo_frame->left_edges)
        std::vector<int32_t> left_index;    // per final pair: index into left_edges
        std::vector<ebvo_edge> right;       // per final pair: the matched right edge (a refined cluster centre)
        std::vector<double> ncc_score;      // per final pair
        std::vector<double> out16;          // per final pair: the 16 numbers of the output row (if calib was given)
    };
    FinalPairs stereo_edge_pairs(const uint8_t *imgL, const uint8_t *imgR, int rows, int cols, ptrdiff_t stepL, ptrdiff_t stepR,
                                 const double F21[9], const ebvo_stereo_calib *calib, bool use_sift = true, int slot = 0,
                                 const ebvo_stereo_params *stereo = nullptr, const ebvo_finalize_params *fin = nullptr)
    {
        FinalPairs r;
        ebvo_stereo_params sp;
        if (stereo)
            sp = *stereo;
        else
        {
            ebvo_stereo_default_params(&sp);
            std::memcpy(sp.F21, F21, sizeof sp.F21);
        }
        ebvo_finalize_params fp;
        if (fin)
            fp = *fin;
        else
        {
            ebvo_finalize_default_params(&fp);
            fp.use_sift = use_sift ? 1 : 0;
        }
        ebvo_ctx *c = ctx_->get();
        if (!report(*ctx_, last_status = ebvo_stereo_upload_slot(c, slot, imgL, imgR, rows, cols, stepL, stepR),
                    "ebvo_stereo_upload_slot") ||
            !report(*ctx_, last_status = ebvo_stereo_submit(c, slot, &sp), "ebvo_stereo_submit") ||
            !report(*ctx_, last_status = ebvo_stereo_wait(c, slot, &r.stage1), "ebvo_stereo_wait") ||
            !report(*ctx_, last_status = ebvo_stereo_finalize(c, slot, &fp, calib, &r.stages), "ebvo_stereo_finalize"))
            return r;
        const size_t n = (size_t)r.stages.n_final;
        r.left_edges.resize((size_t)r.stage1.n_left);
        r.left_index.resize(n);
        r.right.resize(n);
        r.ncc_score.resize(n);
        if (calib)
            r.out16.resize(16 * n);
        if (report(*ctx_, last_status = ebvo_stereo_fetch_slot(c, slot, r.left_edges.data(), nullptr, nullptr, nullptr, nullptr,
                                                               nullptr, nullptr, nullptr),
                   "ebvo_stereo_fetch_slot"))
            report(*ctx_, last_status = ebvo_stereo_fetch_final(c, slot, r.left_index.data(), r.right.data(),
                                                                r.ncc_score.data(), calib ? r.out16.data() : nullptr),
                   "ebvo_stereo_fetch_final");
        return r;
    }

    // The same in three enqueue-only steps for a frame loop that keeps several frames in flight (one slot each,
    // ebvo_stereo_set_slots): _begin uploads the pair and submits TOED + candidates + NCC; _chain reads their counts and
    // enqueues every later stage (no host synchronisation between the stages: their totals stay on the device); _end waits
    // for the chain and copies the final pairs back.  While the chain of frame k runs, the caller begins frame k + 1 and
    // ends frame k - 1: the chains of different slots overlap on the device (bench.py: dropin_final_pairs_per_s).
    bool stereo_edge_pairs_begin(const uint8_t *imgL, const uint8_t *imgR, int rows, int cols, ptrdiff_t stepL, ptrdiff_t stepR,
                                 const double F21[9], int slot, const ebvo_stereo_params *stereo = nullptr)
    {
        ebvo_stereo_params sp;
        if (stereo)
            sp = *stereo;
        else
        {
            ebvo_stereo_default_params(&sp);
            std::memcpy(sp.F21, F21, sizeof sp.F21);
        }
        ebvo_ctx *c = ctx_->get();
        return report(*ctx_, last_status = ebvo_stereo_upload_slot(c, slot, imgL, imgR, rows, cols, stepL, stepR),
                      "ebvo_stereo_upload_slot") &&
               report(*ctx_, last_status = ebvo_stereo_submit(c, slot, &sp), "ebvo_stereo_submit");
    }
    bool stereo_edge_pairs_chain(int slot, const ebvo_stereo_calib *calib, bool use_sift = true, ebvo_stereo_counts *stage1 = nullptr,
                                 const ebvo_finalize_params *fin = nullptr)
    {
        ebvo_finalize_params fp;
        if (fin)
            fp = *fin;
        else
        {
            ebvo_finalize_default_params(&fp);
            fp.use_sift = use_sift ? 1 : 0;
        }
        ebvo_stereo_counts tmp{};
        ebvo_ctx *c = ctx_->get();
        return report(*ctx_, last_status = ebvo_stereo_wait(c, slot, stage1 ? stage1 : &tmp), "ebvo_stereo_wait") &&
               report(*ctx_, last_status = ebvo_stereo_finalize_submit(c, slot, &fp, calib), "ebvo_stereo_finalize_submit");
    }
    FinalPairs stereo_edge_pairs_end(int slot, bool with_rows, int n_left_edges = -1)
    {
        FinalPairs r;
        ebvo_ctx *c = ctx_->get();
        if (!report(*ctx_, last_status = ebvo_stereo_finalize_wait(c, slot, &r.stages), "ebvo_stereo_finalize_wait"))
            return r;
        const size_t n = (size_t)r.stages.n_final;
        r.left_index.resize(n);
        r.right.resize(n);
        r.ncc_score.resize(n);
        if (with_rows)
            r.out16.resize(16 * n);
        if (n_left_edges >= 0)
        {
            r.left_edges.resize((size_t)n_left_edges);
            if (!report(*ctx_, last_status = ebvo_stereo_fetch_slot(c, slot, r.left_edges.data(), nullptr, nullptr, nullptr, nullptr,
                                                                    nullptr, nullptr, nullptr),
                        "ebvo_stereo_fetch_slot"))
                return r;
        }
        report(*ctx_, last_status = ebvo_stereo_fetch_final(c, slot, r.left_index.data(), r.right.data(), r.ncc_score.data(),
                                                            with_rows ? r.out16.data() : nullptr),
               "ebvo_stereo_fetch_final");
        return r;
    }

  private:
    Context::Ptr ctx_;
};

// Temporal_Matches::get_Temporal_Edge_Pairs_from_Quads (src/Temporal_Matches.cpp:168-218) on the FINAL stereo mates of
// resident pairs: the keyframe's mates stay on the device (set_keyframe after StereoMatcherHIP::stereo_edge_pairs on that
// slot), every later frame's mates are matched against them.  stages = 0 stops after the NCC filter (:192), 1 runs the
// SIFT filter, both Best-Nearly-Best tests, the photometric refinement and the edge clustering as well.
class TemporalMatcherHIP
{
  public:
    explicit TemporalMatcherHIP(Context::Ptr ctx) : ctx_(std::move(ctx)) {}
    int last_status = EBVO_OK;

    bool set_keyframe(int slot = 0) // keyframe = frame 0 in the reference (src/Pipeline.cpp:133-138)
    {
        return report(*ctx_, last_status = ebvo_temporal_set_keyframe(ctx_->get(), slot), "ebvo_temporal_set_keyframe");
    }

    struct Quads
    {
        ebvo_temporal_counts counts{};
        // per keyframe mate i: the quads row_ptr[i] .. row_ptr[i + 1] that left the chain
        std::vector<int32_t> row_ptr, cf_index;
        std::vector<ebvo_edge> left, right; // cluster centres (CF_left / CF_right center_edge)
        std::vector<double> ncc_left, sift_left, refine_score_left, refine_score_right;
        std::vector<uint8_t> refine_validity;
    };
    Quads match(int slot = 0, int stages = 1, const ebvo_temporal_params *params = nullptr)
    {
        Quads q;
        ebvo_temporal_params p;
        if (params)
            p = *params;
        else
        {
            ebvo_temporal_default_params(&p);
            p.stages = stages;
        }
        if (!report(*ctx_, last_status = ebvo_temporal_match(ctx_->get(), slot, &p, &q.counts), "ebvo_temporal_match") || !p.stages)
            return q;
        const size_t n = (size_t)q.counts.n_final;
        q.row_ptr.resize((size_t)q.counts.n_kf + 1);
        q.cf_index.resize(n);
        q.left.resize(n);
        q.right.resize(n);
        q.ncc_left.resize(n);
        q.sift_left.resize(n);
        q.refine_score_left.resize(n);
        q.refine_score_right.resize(n);
        q.refine_validity.resize(n);
        report(*ctx_,
               last_status = ebvo_temporal_fetch_final(ctx_->get(), slot, q.row_ptr.data(), q.cf_index.data(), q.left.data(),
                                                       q.right.data(), q.ncc_left.data(), q.sift_left.data(),
                                                       q.refine_score_left.data(), q.refine_score_right.data(),
                                                       q.refine_validity.data()),
               "ebvo_temporal_fetch_final");
        return q;
    }

  private:
    Context::Ptr ctx_;
};

// cv::undistort(src, dst, K, dist) of src/Pipeline.cpp:78-79 on a CV_8UC1 image (K = fx fy cx cy, dist = k1 k2 p1 p2 [k3])
inline std::vector<uint8_t> undistort(const Context &c, const uint8_t *img, int rows, int cols, ptrdiff_t step, const double K[4],
                                      const std::vector<double> &dist)
{
    std::vector<uint8_t> out((size_t)rows * cols);
    const int rc = ebvo_undistort(c.get(), img, rows, cols, step, K, dist.data(), (int)dist.size(), out.data(), cols);
    if (!report(c, rc, "ebvo_undistort"))
        out.clear();
    return out;
}

// The two SIFT descriptors of every edge (cv::SIFT::create()->compute at the +-8 px points; src/Stereo_Matches.cpp:669-677,
// :720-727, :1628-1634): n x 2 x 128 floats, descriptor of shifted_points.first then of .second.
template <class EdgeT>
inline std::vector<float> sift_descriptors(const Context &c, const uint8_t *img, int rows, int cols, ptrdiff_t step,
                                           const std::vector<EdgeT> &edges)
{
    std::vector<ebvo_edge> e(edges.size());
    for (size_t k = 0; k < edges.size(); ++k)
        e[k] = to_abi(edges[k]);
    std::vector<float> out(256 * edges.size());
    const int rc = ebvo_sift_descriptors(c.get(), img, rows, cols, step, e.data(), (int)e.size(), out.data());
    if (!report(c, rc, "ebvo_sift_descriptors"))
        out.clear();
    return out;
}

// apply_SIFT_filtering's score per candidate pair (:736-740): cand_desc holds one descriptor pair per pair of the CSR lists
inline std::vector<double> sift_min_distances(const Context &c, const std::vector<float> &left_desc,
                                              const std::vector<float> &cand_desc, const std::vector<int32_t> &row_ptr)
{
    std::vector<double> d(cand_desc.size() / 256);
    const int rc = ebvo_sift_min_distances(c.get(), left_desc.data(), (int)(left_desc.size() / 256), cand_desc.data(),
                                           row_ptr.data(), d.data());
    if (!report(c, rc, "ebvo_sift_min_distances"))
        d.clear();
    return d;
}


// Utility::get_patch_similarity on two 7x7 CV_32F patches (src/utility.cpp:163-180); also the shape
// of MatlabNCCComputer::computeNCC(patch1, patch2) -> double (include/MatlabNCCComputer.h:41): NaN
// on failure, like the MATLAB wrapper (src/MatlabNCCComputer.cpp:58-90).
inline double patch_similarity(const Context &c, const float *patch_one, const float *patch_two)
{
    double s = 0;
    const int rc = ebvo_ncc_patches(c.get(), patch_one, patch_two, 1, &s);
    if (!report(c, rc, "ebvo_ncc_patches"))
        return __builtin_nan("");
    return s;
}

// The numeric body of Stereo_Matches::write_finalized_stereo_edge_pairs_to_file (src/Stereo_Matches.cpp:1656-1699):
// 16 numbers per final (left edge, right edge) pair, computed on the device.
template <class EdgeT>
inline std::vector<double> finalize_pairs(const Context &c, const ebvo_stereo_calib &calib, const std::vector<EdgeT> &left,
                                          const std::vector<EdgeT> &right)
{
    std::vector<ebvo_edge> L(left.size()), R(right.size());
    for (size_t k = 0; k < left.size(); ++k)
        L[k] = to_abi(left[k]);
    for (size_t k = 0; k < right.size(); ++k)
        R[k] = to_abi(right[k]);
    std::vector<double> out(16 * left.size());
    const int rc = ebvo_finalize_pairs(c.get(), &calib, L.data(), R.data(), (int)L.size(), out.data());
    if (!report(c, rc, "ebvo_finalize_pairs"))
        out.clear();
    return out;
}

// ... and its text: the header line and one row of 16 space-separated numbers per pair, through the same iostream
// insertions as the reference (:1660, :1690-1695), hence the same default formatting (6 significant digits).
inline bool write_finalized_stereo_edge_pairs(const std::string &filename, const std::vector<double> &out16)
{
    std::ofstream outfile(filename);
    if (!outfile)
        return false;
    outfile << "left_edge_location, left_edge_orientation, right_edge_location, right_edge_orientation, "
               "left_edge_3D_point, left_edge_tangent"
            << std::endl;
    for (size_t k = 0; k + 16 <= out16.size(); k += 16)
    {
        const double *o = &out16[k];
        outfile << o[0] << " " << o[1] << " " << o[2] << " " << o[3] << " " << o[4] << " " << o[5] << " " << o[6] << " "
                << o[7] << " " << o[8] << " " << o[9] << " " << o[10] << " " << o[11] << " " << o[12] << " " << o[13] << " "
                << o[14] << " " << o[15] << std::endl;
    }
    return (bool)outfile;
}

} // namespace ebvo
#endif

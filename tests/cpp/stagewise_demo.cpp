// stagewise_demo.cpp -- main_VO's stage-after-stage sequence (src/Pipeline.cpp:24-29, :93-97; src/Stereo_Matches.cpp:1374-1427)
// through include/ebvo/adapters.hpp, every stage handed the std::vectors the previous one returned.  Checks that the
// adapters recognise those vectors as the edge lists still resident on the device, that the resident path returns exactly
// what the host-buffer path returns, and that every way of losing residency (edited edges, a call in between that used the
// library's workspace, images of another pair) falls back to the host-buffer path with the same results.
// usage: stagewise_demo <left.raw> <right.raw> <h> <w> <out.bin>
// out.bin: int32 nL, nR, n_listed, n_pairs; L edges, R edges, staged row_ptr, col_idx, orient_ok; NCC row_ptr, col_idx,
//          best (f64), keep (u8), left patches (f32 nL x 98), sims (4 f64 per pair)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "ebvo/adapters.hpp"

struct Point2d
{
    double x, y;
};
struct Edge
{
    Point2d location{-1.0, -1.0};
    double orientation = -100;
    bool b_isEmpty = true;
    int frame_source = -1;
    int index = 0;
};
struct Mat
{
    unsigned char *data;
    int rows, cols;
    size_t step;
};
typedef ebvo::StereoMatcherHIP<Edge> Matcher;

static std::vector<unsigned char> slurp(const char *path, size_t n)
{
    std::vector<unsigned char> b(n);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(b.data(), 1, n, f) != n)
    {
        std::fprintf(stderr, "cannot read %s\n", path);
        std::exit(2);
    }
    std::fclose(f);
    return b;
}

template <class T>
static bool same(ebvo::Span<T> a, const std::vector<T> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), sizeof(T) * a.size()) == 0);
}
template <class T>
static bool same(ebvo::Span<T> a, ebvo::Span<T> b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), sizeof(T) * a.size()) == 0);
}

// the orientation stage on the host (what the binding's apply_orientation_filter does to its lists)
static void keep_flagged(const Matcher::StagedView &st, size_t nL, std::vector<int32_t> &row_ptr, std::vector<int32_t> &col)
{
    row_ptr.assign(nL + 1, 0);
    col.clear();
    for (size_t i = 0; i < nL; ++i)
    {
        for (int32_t k = st.row_ptr[i]; k < st.row_ptr[i + 1]; ++k)
            if (st.orient_ok[(size_t)k])
                col.push_back(st.col_idx[(size_t)k]);
        row_ptr[i + 1] = (int32_t)col.size();
    }
}

#define CHECK(cond, code)                                                          \
    do                                                                             \
    {                                                                              \
        if (!(cond))                                                               \
        {                                                                          \
            std::fprintf(stderr, "stagewise_demo: check failed: %s\n", #cond);     \
            return code;                                                           \
        }                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 6)
        return 2;
    const int h = std::atoi(argv[3]), w = std::atoi(argv[4]);
    auto bl = slurp(argv[1], (size_t)h * w), br = slurp(argv[2], (size_t)h * w);
    Mat left{bl.data(), h, w, (size_t)w}, right{br.data(), h, w, (size_t)w};
    ebvo::ThirdOrderEdgeDetectionHIP<Edge>::Ptr TOED(new ebvo::ThirdOrderEdgeDetectionHIP<Edge>(h, w));
    CHECK(TOED->last_status == EBVO_OK, 3);
    const double f = 718.856, t = 0.54;
    const double F[9] = {0, 0, 0, 0, 0, -t / f, 0, t / f, 0};
    Matcher matcher(TOED->context());
    const ptrdiff_t sl = (ptrdiff_t)left.step, sr = (ptrdiff_t)right.step;

    // ---- frame 1: the unchanged sequence ------------------------------------------------------------------------------
    TOED->get_Third_Order_Edges(left);
    std::vector<Edge> left_edges = TOED->toed_edges;
    std::vector<double> all4_left(TOED->subpix_edge_pts_final, TOED->subpix_edge_pts_final + 4 * (size_t)TOED->Total_Num_Of_TOED);
    TOED->get_Third_Order_Edges(right);
    std::vector<Edge> right_edges = TOED->toed_edges;
    CHECK(!left_edges.empty() && left_edges[0].b_isEmpty && left_edges[0].frame_source == -1, 4);
    auto lines = Matcher::CalculateEpipolarLine(F, left_edges);
    auto st = matcher.candidates_staged_view(left_edges, right_edges, lines);
    CHECK(matcher.last_status == EBVO_OK && st.resident, 5);
    std::vector<int32_t> row_ptr, col;
    keep_flagged(st, left_edges.size(), row_ptr, col);
    CHECK(same(st.row_ptr_final, row_ptr) && same(st.col_idx_final, col), 24); // the device formed the same list
    // keep what the views show: the page-locked memory behind them is reused by the next call of the same stage
    const std::vector<int32_t> st_rp(st.row_ptr.begin(), st.row_ptr.end()), st_ci(st.col_idx.begin(), st.col_idx.end());
    const std::vector<uint8_t> st_ok(st.orient_ok.begin(), st.orient_ok.end());
    auto s = matcher.ncc_indexed(left.data, right.data, h, w, sl, sr, left_edges, right_edges, row_ptr, col, EBVO_NCC_THRESH, true);
    CHECK(matcher.last_status == EBVO_OK && s.resident, 6);
    const std::vector<double> s_best(s.best.begin(), s.best.end()), s_sims(s.pp_nn_pn_np.begin(), s.pp_nn_pn_np.end());
    const std::vector<uint8_t> s_keep(s.keep.begin(), s.keep.end());
    const std::vector<float> s_lp(s.left_patches.begin(), s.left_patches.end());
    CHECK(s_sims.size() == 4 * col.size() && s_lp.size() == 98 * left_edges.size(), 7);
    {
        // without the four scores: same best / keep
        auto s2 = matcher.ncc_indexed(left.data, right.data, h, w, sl, sr, left_edges, right_edges, row_ptr, col);
        CHECK(s2.resident && s2.pp_nn_pn_np.empty() && same(s2.best, s_best) && same(s2.keep, s_keep) && same(s2.left_patches, s_lp), 8);
    }

    // ---- the host-buffer path on the same vectors: identical results ----------------------------------------------------
    {
        auto ex = matcher.candidates_staged(left_edges, right_edges, lines);
        CHECK(matcher.last_status == EBVO_OK, 9);
        CHECK(ex.lists.row_ptr == st_rp && ex.lists.col_idx == st_ci && ex.orient_ok == st_ok, 10);
        std::vector<Edge> cand(col.size());
        for (size_t k = 0; k < cand.size(); ++k)
            cand[k] = right_edges[(size_t)col[k]];
        ebvo::NccScores e = matcher.ncc(left.data, right.data, h, w, sl, sr, left_edges, row_ptr, cand);
        CHECK(e.best == s_best && e.keep == s_keep && e.pp_nn_pn_np == s_sims && e.left_patches == s_lp, 11);
        // host-buffer calls that run no detector leave the resident lists alone ...
        auto st1 = matcher.candidates_staged_view(left_edges, right_edges, lines);
        CHECK(matcher.last_status == EBVO_OK && st1.resident && same(st1.col_idx, st_ci), 22);
        // ... a detector call behind the adapter's back overwrites them: the adapter still believes the edges are resident,
        // the library refuses the stale tags, the host-buffer path runs
        {
            int nk = 0, nt = 0;
            std::vector<ebvo_edge> tmp((size_t)h * w);
            CHECK(ebvo_toed(TOED->context()->get(), left.data, h, w, sl, tmp.data(), (int)tmp.size(), &nk, &nt, nullptr, 0, nullptr,
                            nullptr) == EBVO_OK,
                  23);
        }
        auto st2 = matcher.candidates_staged_view(left_edges, right_edges, lines);
        CHECK(matcher.last_status == EBVO_OK && !st2.resident, 12);
        CHECK(same(st2.row_ptr, st_rp) && same(st2.col_idx, st_ci) && same(st2.orient_ok, st_ok), 13);
        CHECK(same(st2.row_ptr_final, row_ptr) && same(st2.col_idx_final, col), 25);
        auto s3 = matcher.ncc_indexed(left.data, right.data, h, w, sl, sr, left_edges, right_edges, row_ptr, col, EBVO_NCC_THRESH, true);
        CHECK(matcher.last_status == EBVO_OK && !s3.resident, 14);
        CHECK(same(s3.best, s_best) && same(s3.keep, s_keep) && same(s3.pp_nn_pn_np, s_sims) && same(s3.left_patches, s_lp), 15);
    }

    // ---- frame 2 = the same images again (right first this time): resident again, left in the other workspace ------------
    TOED->get_Third_Order_Edges(right);
    std::vector<Edge> right_b = TOED->toed_edges;
    TOED->get_Third_Order_Edges(left);
    std::vector<Edge> left_b = TOED->toed_edges;
    CHECK((size_t)TOED->Total_Num_Of_TOED * 4 == all4_left.size() &&
              std::memcmp(TOED->subpix_edge_pts_final, all4_left.data(), sizeof(double) * all4_left.size()) == 0,
          16);
    {
        auto st4 = matcher.candidates_staged_view(left_b, right_b, lines);
        CHECK(st4.resident && same(st4.row_ptr, st_rp) && same(st4.col_idx, st_ci) && same(st4.orient_ok, st_ok), 17);
        auto s4 = matcher.ncc_indexed(left.data, right.data, h, w, sl, sr, left_b, right_b, row_ptr, col, EBVO_NCC_THRESH, true);
        CHECK(s4.resident && same(s4.best, s_best) && same(s4.keep, s_keep) && same(s4.pp_nn_pn_np, s_sims) &&
                  same(s4.left_patches, s_lp),
              18);
        // an edited edge list is not the resident one: the stage runs on what it was given
        std::vector<Edge> edited = left_b;
        edited[edited.size() / 2].orientation += 0.25;
        auto lines_e = Matcher::CalculateEpipolarLine(F, edited);
        auto st5 = matcher.candidates_staged_view(edited, right_b, lines_e);
        auto ex5 = matcher.candidates_staged(edited, right_b, lines_e);
        CHECK(!st5.resident && same(st5.row_ptr, ex5.lists.row_ptr) && same(st5.col_idx, ex5.lists.col_idx) &&
                  same(st5.orient_ok, ex5.orient_ok),
              19);
        CHECK(!(ex5.orient_ok == st_ok), 20); // ... and the edit shows in the result
        // a malformed list is refused, not launched
        std::vector<int32_t> bad = col;
        bad[bad.size() / 3] = (int32_t)right_b.size();
        auto s6 = matcher.ncc_indexed(left.data, right.data, h, w, sl, sr, left_b, right_b, row_ptr, bad);
        CHECK(matcher.last_status == EBVO_ERR_ARG && s6.best.empty(), 21);
    }

    // ---- get_Stereo_Edge_Pairs in one pass, two frames in flight: begin / chain / end equal the one-call form ---------------
    {
        ebvo_stereo_calib calib = {{f, 0, 607.1928, 0, f, 185.2157, 0, 0, 1}, {f, 0, 607.1928, 0, f, 185.2157, 0, 0, 1},
                                   {1, 0, 0, 0, 1, 0, 0, 0, 1}, {t, 0, 0}};
        CHECK(ebvo_stereo_set_slots(TOED->context()->get(), 2) == EBVO_OK, 30);
        auto ref0 = matcher.stereo_edge_pairs(left.data, right.data, h, w, sl, sr, F, &calib, true, 0);
        auto ref1 = matcher.stereo_edge_pairs(right.data, left.data, h, w, sr, sl, F, &calib, true, 1); // another pair: the images swapped
        CHECK(matcher.last_status == EBVO_OK && ref0.stages.n_final > 100, 31);
        CHECK(matcher.stereo_edge_pairs_begin(left.data, right.data, h, w, sl, sr, F, 0) &&
                  matcher.stereo_edge_pairs_begin(right.data, left.data, h, w, sr, sl, F, 1),
              32);
        ebvo_stereo_counts c0{}, c1{};
        CHECK(matcher.stereo_edge_pairs_chain(0, &calib, true, &c0) && matcher.stereo_edge_pairs_chain(1, &calib, true, &c1), 33);
        auto got1 = matcher.stereo_edge_pairs_end(1, true, c1.n_left);
        auto got0 = matcher.stereo_edge_pairs_end(0, true, c0.n_left);
        CHECK(matcher.last_status == EBVO_OK, 34);
        for (auto pr : {std::make_pair(&got0, &ref0), std::make_pair(&got1, &ref1)})
        {
            const auto &g = *pr.first, &r = *pr.second;
            CHECK(std::memcmp(&g.stages, &r.stages, sizeof g.stages) == 0 && g.left_index == r.left_index && g.ncc_score == r.ncc_score &&
                      g.out16.size() == r.out16.size() &&
                      std::memcmp(g.out16.data(), r.out16.data(), sizeof(double) * g.out16.size()) == 0 &&
                      g.right.size() == r.right.size() &&
                      std::memcmp(g.right.data(), r.right.data(), sizeof(ebvo_edge) * g.right.size()) == 0 &&
                      g.left_edges.size() == r.left_edges.size() &&
                      std::memcmp(g.left_edges.data(), r.left_edges.data(), sizeof(ebvo_edge) * g.left_edges.size()) == 0,
                  35);
        }
    }

    FILE *o = std::fopen(argv[5], "wb");
    int32_t hdr[4] = {(int32_t)left_edges.size(), (int32_t)right_edges.size(), (int32_t)st_ci.size(), (int32_t)col.size()};
    std::fwrite(hdr, sizeof hdr, 1, o);
    for (const auto *v : {&left_edges, &right_edges})
        for (const Edge &e : *v)
        {
            ebvo_edge a = ebvo::to_abi(e);
            std::fwrite(&a, sizeof a, 1, o);
        }
    std::fwrite(st_rp.data(), sizeof(int32_t), st_rp.size(), o);
    std::fwrite(st_ci.data(), sizeof(int32_t), st_ci.size(), o);
    std::fwrite(st_ok.data(), 1, st_ok.size(), o);
    std::fwrite(row_ptr.data(), sizeof(int32_t), row_ptr.size(), o);
    std::fwrite(col.data(), sizeof(int32_t), col.size(), o);
    std::fwrite(s_best.data(), sizeof(double), s_best.size(), o);
    std::fwrite(s_keep.data(), 1, s_keep.size(), o);
    std::fwrite(s_lp.data(), sizeof(float), s_lp.size(), o);
    std::fwrite(s_sims.data(), sizeof(double), s_sims.size(), o);
    std::fwrite(all4_left.data(), sizeof(double), all4_left.size(), o);
    std::fclose(o);
    std::printf("stagewise_demo ok: %zu + %zu edges, %zu listed, %zu pairs\n", left_edges.size(), right_edges.size(), st_ci.size(),
                col.size());
    return 0;
}

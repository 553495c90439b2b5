// adapter_demo.cpp -- exercises include/ebvo/adapters.hpp the way Pipeline::ProcessEdges and
// Stereo_Matches::get_Stereo_Edge_Pairs would (src/Pipeline.cpp:24-29, src/Stereo_Matches.cpp:1374-1427),
// with plain local types standing where cv::Mat / struct Edge stand in the reference tree.
// usage: adapter_demo <left.raw> <right.raw> <h> <w> <out.bin>
// out.bin: int32 nL, nR, totalL, totalR, npairs; then L edges (x,y,theta,index as 3 f64 + i32 + pad),
//          R edges, row_ptr, col_idx, sims (4 f64 per pair), keep (u8 per pair)
#include <cstdio>
#include <cstdlib>
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

int main(int argc, char **argv)
{
    if (argc != 6)
        return 2;
    const int h = std::atoi(argv[3]), w = std::atoi(argv[4]);
    auto bl = slurp(argv[1], (size_t)h * w), br = slurp(argv[2], (size_t)h * w);
    Mat left{bl.data(), h, w, (size_t)w}, right{br.data(), h, w, (size_t)w};

    ebvo::ThirdOrderEdgeDetectionHIP<Edge>::Ptr TOED(new ebvo::ThirdOrderEdgeDetectionHIP<Edge>(h, w));
    if (TOED->last_status != EBVO_OK)
        return 3;
    // Pipeline::ProcessEdges(left) / (right): results are copied out by value
    TOED->get_Third_Order_Edges(left);
    std::vector<Edge> left_edges = TOED->toed_edges;
    const int totalL = TOED->Total_Num_Of_TOED;
    TOED->get_Third_Order_Edges(right);
    std::vector<Edge> right_edges = TOED->toed_edges;
    const int totalR = TOED->Total_Num_Of_TOED;
    if (left_edges.empty() || left_edges[0].b_isEmpty != true || left_edges[0].frame_source != -1)
        return 4;

    // rectified KITTI-like geometry: l = F x with horizontal epipolar lines
    const double f = 718.856, t = 0.54;
    const double F[9] = {0, 0, 0, 0, 0, -t / f, 0, t / f, 0};
    ebvo::StereoMatcherHIP<Edge> matcher(TOED->context());
    auto lines = ebvo::StereoMatcherHIP<Edge>::CalculateEpipolarLine(F, left_edges);
    ebvo::CandidateLists c = matcher.candidates(left_edges, right_edges, lines);
    {
        // the staged call: flagged pairs of the (epipolar + disparity) list == the list of all three stages
        auto st = matcher.candidates_staged(left_edges, right_edges, lines);
        std::vector<int32_t> flagged;
        for (size_t k = 0; k < st.lists.col_idx.size(); ++k)
            if (st.orient_ok[k])
                flagged.push_back(st.lists.col_idx[k]);
        if (matcher.last_status != EBVO_OK || flagged != c.col_idx || st.lists.col_idx.size() <= c.col_idx.size())
            return 12;
    }
    std::vector<Edge> cand(c.col_idx.size());
    for (size_t k = 0; k < cand.size(); ++k)
        cand[k] = right_edges[(size_t)c.col_idx[k]];
    ebvo::NccScores s = matcher.ncc(left.data, right.data, h, w, (ptrdiff_t)left.step, (ptrdiff_t)right.step, left_edges,
                                    c.row_ptr, cand);
    if (matcher.last_status != EBVO_OK)
        return 5;
    const double self = ebvo::patch_similarity(*TOED->context(), s.left_patches.data(), s.left_patches.data());
    // refine_edge_disparity: every candidate pair, candidate centre = the right TOED edge
    std::vector<double> cand_xy(2 * cand.size());
    for (size_t k = 0; k < cand.size(); ++k)
    {
        cand_xy[2 * k] = cand[k].location.x;
        cand_xy[2 * k + 1] = cand[k].location.y;
    }
    auto refined = matcher.refine(left.data, right.data, h, w, (ptrdiff_t)left.step, (ptrdiff_t)right.step, left_edges, lines,
                                  c.row_ptr, cand_xy);
    if (matcher.last_status != EBVO_OK)
        return 6;

    // stage glue on the candidate lists: BNB on the NCC scores, epipolar shift, clustering, best per row
    auto bnb = matcher.bnb_test(c.row_ptr, s.best, EBVO_BNB_NCC, true);
    auto shifted = matcher.epipolar_shift(cand, lines, c.row_ptr);
    auto clusters = matcher.cluster_rows(shifted, c.row_ptr, false, true);
    auto bestsel = matcher.keep_best(c.row_ptr, s.best);
    if (matcher.last_status != EBVO_OK)
        return 8;
    // finalisation: every kept NCC match as a final pair -> the reference's output file
    std::vector<Edge> fl, fr;
    for (size_t i = 0; i + 1 < c.row_ptr.size(); ++i)
        for (int32_t k = c.row_ptr[i]; k < c.row_ptr[i + 1]; ++k)
            if (s.keep[(size_t)k])
            {
                fl.push_back(left_edges[i]);
                fr.push_back(cand[(size_t)k]);
            }
    ebvo_stereo_calib calib = {{f, 0, 607.1928, 0, f, 185.2157, 0, 0, 1}, {f, 0, 607.1928, 0, f, 185.2157, 0, 0, 1},
                               {1, 0, 0, 0, 1, 0, 0, 0, 1}, {t, 0, 0}};
    std::vector<double> fin = ebvo::finalize_pairs(*TOED->context(), calib, fl, fr);
    if (fin.size() != 16 * fl.size() || !ebvo::write_finalized_stereo_edge_pairs(std::string(argv[5]) + ".txt", fin))
        return 7;

    FILE *o = std::fopen(argv[5], "wb");
    int32_t hdr[5] = {(int32_t)left_edges.size(), (int32_t)right_edges.size(), totalL, totalR, (int32_t)cand.size()};
    std::fwrite(hdr, sizeof hdr, 1, o);
    for (const auto *v : {&left_edges, &right_edges})
        for (const Edge &e : *v)
        {
            ebvo_edge a = ebvo::to_abi(e);
            std::fwrite(&a, sizeof a, 1, o);
        }
    std::fwrite(c.row_ptr.data(), sizeof(int32_t), c.row_ptr.size(), o);
    std::fwrite(c.col_idx.data(), sizeof(int32_t), c.col_idx.size(), o);
    std::fwrite(s.pp_nn_pn_np.data(), sizeof(double), s.pp_nn_pn_np.size(), o);
    std::fwrite(s.keep.data(), 1, s.keep.size(), o);
    std::fwrite(&self, sizeof self, 1, o);
    std::fwrite(refined.alpha.data(), sizeof(double), refined.alpha.size(), o);
    std::fwrite(refined.score.data(), sizeof(double), refined.score.size(), o);
    std::fwrite(refined.xy.data(), sizeof(double), refined.xy.size(), o);
    std::fwrite(refined.validity.data(), 1, refined.validity.size(), o);
    std::fwrite(bnb.new_count.data(), sizeof(int32_t), bnb.new_count.size(), o);
    std::fwrite(bnb.order.data(), sizeof(int32_t), bnb.order.size(), o);
    std::fwrite(shifted.data(), sizeof(ebvo_edge), shifted.size(), o);
    std::fwrite(clusters.new_count.data(), sizeof(int32_t), clusters.new_count.size(), o);
    std::fwrite(clusters.cluster_of.data(), sizeof(int32_t), clusters.cluster_of.size(), o);
    std::fwrite(bestsel.new_count.data(), sizeof(int32_t), bestsel.new_count.size(), o);
    // get_Stereo_Edge_Pairs in one pass over the device (SIFT stages included): final pairs + output rows
    auto fp = matcher.stereo_edge_pairs(left.data, right.data, h, w, (ptrdiff_t)left.step, (ptrdiff_t)right.step, F, &calib, true);
    if (matcher.last_status != EBVO_OK || fp.left_edges.size() != left_edges.size())
        return 9;
    int32_t fh[7] = {fp.stages.n_sift, fp.stages.n_ncc, fp.stages.n_bnb, fp.stages.n_clusters, fp.stages.n_ncc2,
                     fp.stages.n_final, (int32_t)fp.stage1.n_pairs};
    std::fwrite(fh, sizeof fh, 1, o);
    std::fwrite(fp.left_index.data(), sizeof(int32_t), fp.left_index.size(), o);
    std::fwrite(fp.right.data(), sizeof(ebvo_edge), fp.right.size(), o);
    std::fwrite(fp.ncc_score.data(), sizeof(double), fp.ncc_score.size(), o);
    std::fwrite(fp.out16.data(), sizeof(double), fp.out16.size(), o);
    // temporal quads: this frame is the keyframe; the next frame is the same scene moved by two pixels
    ebvo::TemporalMatcherHIP temporal(TOED->context());
    if (!temporal.set_keyframe(0))
        return 10;
    std::vector<unsigned char> l2(bl.size()), r2(br.size());
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            l2[(size_t)y * w + x] = bl[(size_t)y * w + (x + w - 2) % w];
            r2[(size_t)y * w + x] = br[(size_t)y * w + (x + w - 2) % w];
        }
    auto fp2 = matcher.stereo_edge_pairs(l2.data(), r2.data(), h, w, w, w, F, &calib, true);
    auto tq = temporal.match(0, 1);
    if (matcher.last_status != EBVO_OK || temporal.last_status != EBVO_OK)
        return 11;
    int64_t th[4] = {tq.counts.n_kf, tq.counts.n_kept, tq.counts.n_bnb_sift, tq.counts.n_final};
    std::fwrite(th, sizeof th, 1, o);
    std::fwrite(tq.row_ptr.data(), sizeof(int32_t), tq.row_ptr.size(), o);
    std::fwrite(tq.cf_index.data(), sizeof(int32_t), tq.cf_index.size(), o);
    std::fwrite(tq.left.data(), sizeof(ebvo_edge), tq.left.size(), o);
    std::fwrite(tq.right.data(), sizeof(ebvo_edge), tq.right.size(), o);
    std::fwrite(tq.refine_validity.data(), 1, tq.refine_validity.size(), o);
    std::fclose(o);
    std::printf("adapter_demo ok: %zu + %zu edges, %zu pairs\n", left_edges.size(), right_edges.size(), cand.size());
    return 0;
}

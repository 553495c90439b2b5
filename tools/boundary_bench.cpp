// boundary_bench.cpp -- wall time of the stage-wise drop-in sequence as main_VO runs it through the adapters
// (src/Pipeline.cpp:24-29, :93-97; src/Stereo_Matches.cpp:1374-1427): ProcessEdges(left), ProcessEdges(right),
// CalculateEpipolarLine, the three geometric candidate stages (one device search: candidates_staged_view + the host-side
// drop of the unflagged pairs), apply_NCC_Filtering with the left patches -- one call after the other, every stage handed
// the std::vectors the previous one returned.  The adapters recognise those vectors as the edge lists still resident on the
// device (bit-for-bit comparison) and run the stage there; results are read from page-locked memory of the context.  Plain local types stand where cv::Mat / struct Edge stand in the reference tree.
// usage: boundary_bench <left.raw> <right.raw> <h> <w> <iterations>      prints one JSON object
#include <chrono>
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

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc != 6)
        return 2;
    const int h = std::atoi(argv[3]), w = std::atoi(argv[4]), iters = std::atoi(argv[5]);
    auto bl = slurp(argv[1], (size_t)h * w), br = slurp(argv[2], (size_t)h * w);
    Mat left{bl.data(), h, w, (size_t)w}, right{br.data(), h, w, (size_t)w};
    ebvo::ThirdOrderEdgeDetectionHIP<Edge>::Ptr TOED(new ebvo::ThirdOrderEdgeDetectionHIP<Edge>(h, w));
    if (TOED->last_status != EBVO_OK)
        return 3;
    const double f = 718.856, t = 0.54; // rectified KITTI-like geometry
    const double F[9] = {0, 0, 0, 0, 0, -t / f, 0, t / f, 0};
    ebvo::StereoMatcherHIP<Edge> matcher(TOED->context());
    double split[5] = {0, 0, 0, 0, 0};
    size_t n_left = 0, n_listed = 0, n_pairs = 0, n_kept = 0;
    int resident_calls = 0;
    double toed_split[4] = {0, 0, 0, 0}; // detector call (left), copy of its result, detector call (right), copy
    std::vector<int32_t> row_ptr, col; // the caller's lists, reused from frame to frame
    for (int it = 0; it <= iters; ++it) // the first turn is untimed: it sizes the library's buffers
    {
        double tk[6];
        tk[0] = now();
        TOED->get_Third_Order_Edges(left);
        const double t_a = now();
        std::vector<Edge> left_edges = TOED->toed_edges; // Pipeline::ProcessEdges copies the result out (src/Pipeline.cpp:28)
        const double t_b = now();
        TOED->get_Third_Order_Edges(right);
        const double t_c = now();
        std::vector<Edge> right_edges = TOED->toed_edges;
        tk[1] = now();
        if (it)
        {
            toed_split[0] += t_a - tk[0];
            toed_split[1] += t_b - t_a;
            toed_split[2] += t_c - t_b;
            toed_split[3] += tk[1] - t_c;
        }
        auto lines = ebvo::StereoMatcherHIP<Edge>::CalculateEpipolarLine(F, left_edges);
        tk[2] = now();
        // the vectors handed back are the ones the detector produced: the search runs on the device copies
        auto st = matcher.candidates_staged_view(left_edges, right_edges, lines);
        if (matcher.last_status != EBVO_OK)
            return 4;
        tk[3] = now();
        // the host step of the binding: after apply_orientation_filter the caller's lists are the flagged pairs of every
        // row (the staged search returns that list as well, formed on the device); the NCC stage names each survivor by its
        // right TOED index
        row_ptr.assign(st.row_ptr_final.begin(), st.row_ptr_final.end());
        col.assign(st.col_idx_final.begin(), st.col_idx_final.end());
        tk[4] = now();
        auto s = matcher.ncc_indexed(left.data, right.data, h, w, (ptrdiff_t)left.step, (ptrdiff_t)right.step, left_edges,
                                     right_edges, row_ptr, col);
        if (matcher.last_status != EBVO_OK)
            return 5;
        tk[5] = now();
        if (it)
        {
            for (int q = 0; q < 5; ++q)
                split[q] += tk[q + 1] - tk[q];
            resident_calls += (st.resident ? 1 : 0) + (s.resident ? 1 : 0);
        }
        n_left = left_edges.size();
        n_listed = st.col_idx.size();
        n_pairs = col.size();
        n_kept = 0;
        for (uint8_t kf : s.keep)
            n_kept += kf;
        if (s.left_patches.size() != 98 * n_left || s.best.size() != n_pairs)
            return 6;
    }
    double total = 0;
    for (double &v : split)
    {
        v /= iters;
        total += v;
    }
    std::printf("{\"pairs_per_s\": %.3f, \"ms\": {\"toed_both_images\": %.3f, \"epipolar_lines\": %.3f, \"candidates_staged\": %.3f, "
                "\"host_row_filter\": %.3f, \"ncc_with_left_patches\": %.3f}, \"left_edges\": %zu, \"listed_pairs\": %zu, "
                "\"candidate_pairs\": %zu, \"ncc_matches\": %zu, \"resident_stage_calls\": %d, \"stage_calls\": %d, \"toed_ms\": {\"get_Third_Order_Edges_left\": %.3f, "
                "\"copy_left\": %.3f, \"get_Third_Order_Edges_right\": %.3f, \"copy_right\": %.3f}}\n",
                1.0 / total, split[0] * 1e3, split[1] * 1e3, split[2] * 1e3, split[3] * 1e3, split[4] * 1e3, n_left, n_listed,
                n_pairs, n_kept, resident_calls, 2 * iters, toed_split[0] / iters * 1e3, toed_split[1] / iters * 1e3,
                toed_split[2] / iters * 1e3, toed_split[3] / iters * 1e3);
    return 0;
}

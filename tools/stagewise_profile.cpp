// stagewise_profile.cpp -- where the time of the resident stage-wise calls goes (developer tool; run on the GPU box).
// usage: stagewise_profile <left.raw> <right.raw> <h> <w> <iterations>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ebvo_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::vector<unsigned char> slurp(const char *path, size_t n)
{
    std::vector<unsigned char> b(n);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(b.data(), 1, n, f) != n)
        std::exit(2);
    std::fclose(f);
    return b;
}
struct Edge
{
    double x = -1, y = -1, o = -100;
    bool e = true;
    int fs = -1, index = 0;
};

int main(int argc, char **argv)
{
    if (argc != 6)
        return 2;
    const int h = std::atoi(argv[3]), w = std::atoi(argv[4]), iters = std::atoi(argv[5]);
    auto bl = slurp(argv[1], (size_t)h * w), br = slurp(argv[2], (size_t)h * w);
    ebvo_ctx *ctx = nullptr;
    if (ebvo_ctx_create(0, h, w, &ctx))
        return 3;
    ebvo_set_toed_mode(ctx, EBVO_TOED_HYBRID);
    const double f = 718.856, t = 0.54;
    const double F[9] = {0, 0, 0, 0, 0, -t / f, 0, t / f, 0};
    double acc[12] = {0};
    const char *names[12] = {"toed_resident(left, all4)", "toed_resident(right, all4)", "edges -> vector<Edge> x2", "vector copies x2",
                             "lines (library)", "candidates_resident", "final lists -> caller vectors", "ncc_resident (patches)", "toed_resident no all4 x2",
                             "ncc_resident (no patches)", "extra first toed_resident (SP_PRE)", "-"};
    std::vector<double> lines;
    std::vector<int32_t> rp, col;
    for (int it = 0; it <= iters; ++it)
    {
        double tk[16];
        int q = 0;
        ebvo_toed_view vl, vr;
        if (std::getenv("SP_PRE"))
        {
            ebvo_toed_view v0;
            double a = now();
            ebvo_toed_resident(ctx, 0, bl.data(), h, w, w, 1, &v0);
            acc[10] += now() - a;
        }
        tk[q++] = now();
        if (ebvo_toed_resident(ctx, 0, bl.data(), h, w, w, 1, &vl))
            return 4;
        tk[q++] = now();
        if (ebvo_toed_resident(ctx, 1, br.data(), h, w, w, 1, &vr))
            return 4;
        tk[q++] = now();
        std::vector<Edge> L((size_t)vl.n_kept), R((size_t)vr.n_kept);
        for (int k = 0; k < vl.n_kept; ++k)
        {
            L[k].x = vl.edges[k].x, L[k].y = vl.edges[k].y, L[k].o = vl.edges[k].theta, L[k].index = vl.edges[k].index;
        }
        for (int k = 0; k < vr.n_kept; ++k)
        {
            R[k].x = vr.edges[k].x, R[k].y = vr.edges[k].y, R[k].o = vr.edges[k].theta, R[k].index = vr.edges[k].index;
        }
        tk[q++] = now();
        std::vector<Edge> L2 = L, R2 = R;
        tk[q++] = now();
        lines.resize(3 * (size_t)vl.n_kept);
        ebvo_epipolar_lines(F, vl.edges, vl.n_kept, lines.data());
        tk[q++] = now();
        ebvo_candidates_view cv;
        if (ebvo_epi_candidates_resident(ctx, vl.tag, vr.tag, lines.data(), 0.5, 25.0, 10.0, 3, 1, &cv))
            return 5;
        tk[q++] = now();
        rp.assign(cv.row_ptr_final, cv.row_ptr_final + vl.n_kept + 1);
        col.assign(cv.col_idx_final, cv.col_idx_final + cv.n_final);
        tk[q++] = now();
        ebvo_ncc_view nv;
        if (ebvo_ncc_pairs_resident(ctx, vl.tag, vr.tag, bl.data(), br.data(), h, w, w, w, rp.data(), col.data(), 0.6,
                                    EBVO_NCC_WANT_LEFT_PATCHES, &nv))
            return 6;
        tk[q++] = now();
        ebvo_toed_view v2;
        ebvo_toed_resident(ctx, 0, bl.data(), h, w, w, 0, &v2);
        const uint64_t tl = v2.tag;
        ebvo_toed_resident(ctx, 1, br.data(), h, w, w, 0, &v2);
        tk[q++] = now();
        if (ebvo_ncc_pairs_resident(ctx, tl, v2.tag, bl.data(), br.data(), h, w, w, w, rp.data(), col.data(), 0.6, 0, &nv))
            return 7;
        tk[q++] = now();
        if (it)
            for (int k = 0; k + 1 < q; ++k)
                acc[k] += tk[k + 1] - tk[k];
        if (L2.size() + R2.size() == 7)
            std::printf("x");
    }
    for (int k = 0; k < 11; ++k)
        std::printf("%-32s %8.3f ms\n", names[k], acc[k] / iters * 1e3);
    ebvo_ctx_destroy(ctx);
    return 0;
}

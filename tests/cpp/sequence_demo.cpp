// sequence_demo.cpp -- drives include/ebvo/sequence.hpp: PNG decode and the batched KITTI / EuRoC feeders.
//   sequence_demo decode <file.png>                 -> "w h fnv1a64" of the decoded gray image (no GPU needed)
//   sequence_demo kitti <dir> <slots>               -> one line per pair: index n_left n_right n_pairs n_matches
//   sequence_demo euroc <csv> <left> <right> <slots>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "ebvo/sequence.hpp"

static uint64_t fnv(const std::vector<uint8_t> &v)
{
    uint64_t h = 1469598103934665603ull;
    for (uint8_t b : v)
        h = (h ^ b) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !std::strcmp(argv[1], "decode"))
    {
        ebvo::GrayImage g;
        const std::string err = ebvo::read_png_gray(argv[2], g);
        if (!err.empty())
        {
            std::printf("error: %s\n", err.c_str());
            return 2;
        }
        std::printf("%d %d %016" PRIx64 "\n", g.width, g.height, fnv(g.pixels));
        return 0;
    }
    std::unique_ptr<ebvo::StereoSequence> seq;
    int slots = 3;
    bool chain = false;
    if (argc >= 4 && !std::strcmp(argv[1], "kitti-chain")) // the same walk with the stages after the first NCC pass
    {
        seq.reset(new ebvo::KittiSequence(argv[2]));
        slots = std::atoi(argv[3]);
        chain = true;
    }
    else if (argc >= 4 && !std::strcmp(argv[1], "kitti"))
    {
        seq.reset(new ebvo::KittiSequence(argv[2]));
        slots = std::atoi(argv[3]);
    }
    else if (argc >= 6 && !std::strcmp(argv[1], "euroc"))
    {
        seq.reset(new ebvo::EurocSequence(argv[2], argv[3], argv[4]));
        slots = std::atoi(argv[5]);
    }
    else
    {
        std::fprintf(stderr, "usage: see the head of sequence_demo.cpp\n");
        return 1;
    }
    // size the context from the first pair
    ebvo::StereoImages first;
    if (!seq->getNext(first))
        return 3;
    seq->reset();
    ebvo_ctx *ctx = nullptr;
    int rc = ebvo_ctx_create(0, first.left.height, first.left.width, &ctx);
    if (rc != EBVO_OK)
    {
        std::fprintf(stderr, "ebvo_ctx_create: %s\n", ebvo_strerror(rc));
        return 4;
    }
    ebvo_set_toed_mode(ctx, EBVO_TOED_HYBRID);
    ebvo_stereo_params p;
    ebvo_stereo_default_params(&p);
    // KITTI calibration (config/kitti.yaml:13-28): rectified, F21 = [0 0 0; 0 0 -b'; 0 b' 0] up to scale
    const double fx = 718.856, T = 0.54;
    const double F[9] = {0, 0, 0, 0, 0, -T / fx, 0, T / fx, 0};
    std::memcpy(p.F21, F, sizeof F);
    if (chain)
    {
        ebvo_finalize_params fp;
        ebvo_finalize_default_params(&fp);
        fp.use_sift = 1;
        ebvo::BatchedStereoFeeder feeder(ctx, slots);
        const size_t n = feeder.run_chain(*seq, p, fp, nullptr,
                                          [](const ebvo::StereoImages &f, int, const ebvo_stereo_counts &c, const ebvo_finalize_counts &fc) {
                                              std::printf("%zu %d %lld %d %d %d %d\n", f.index, c.n_left, (long long)c.n_matches, fc.n_sift,
                                                          fc.n_bnb, fc.n_clusters, fc.n_final);
                                          });
        std::printf("pairs %zu status %d\n", n, feeder.status());
    }
    else
    {
        ebvo::BatchedStereoFeeder feeder(ctx, slots);
        const size_t n = feeder.run(*seq, p, [](const ebvo::StereoImages &f, int, const ebvo_stereo_counts &c) {
            std::printf("%zu %d %d %lld %lld\n", f.index, c.n_left, c.n_right, (long long)c.n_pairs, (long long)c.n_matches);
        });
        std::printf("pairs %zu status %d\n", n, feeder.status());
    }
    ebvo_ctx_destroy(ctx);
    return 0;
}

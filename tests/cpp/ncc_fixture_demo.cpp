// Sends pairs of stored 7x7 float patches through ebvo::patch_similarity -- the adapter a reference-side caller of
// Utility::get_patch_similarity (src/utility.cpp:163-180) / MatlabNCCComputer::computeNCC (include/MatlabNCCComputer.h:41)
// would bind -- one call per pair, and through the batched C entry point ebvo_ncc_patches.
// usage: ncc_fixture_demo <in.bin> <out.bin>; in = int32 n, then n x (49 floats A, 49 floats B); out = n doubles (one call
// per pair) followed by n doubles (one batched call).
#include <cstdio>
#include <vector>

#include "ebvo/adapters.hpp"

int main(int argc, char **argv)
{
    if (argc != 3)
        return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f)
        return 2;
    int n = 0;
    if (std::fread(&n, sizeof n, 1, f) != 1 || n <= 0 || n > 1 << 20)
        return 2;
    std::vector<float> A(49 * (size_t)n), B(49 * (size_t)n);
    for (int k = 0; k < n; ++k)
        if (std::fread(&A[49 * (size_t)k], sizeof(float), 49, f) != 49 || std::fread(&B[49 * (size_t)k], sizeof(float), 49, f) != 49)
            return 2;
    std::fclose(f);

    ebvo::Context ctx(64, 64);
    if (ctx.status() != EBVO_OK)
        return 3;
    std::vector<double> single(n), batch(n);
    for (int k = 0; k < n; ++k)
        single[k] = ebvo::patch_similarity(ctx, &A[49 * (size_t)k], &B[49 * (size_t)k]);
    if (ebvo_ncc_patches(ctx.get(), A.data(), B.data(), n, batch.data()) != EBVO_OK)
        return 4;
    FILE *o = std::fopen(argv[2], "wb");
    if (!o)
        return 2;
    std::fwrite(single.data(), sizeof(double), n, o);
    std::fwrite(batch.data(), sizeof(double), n, o);
    std::fclose(o);
    return 0;
}

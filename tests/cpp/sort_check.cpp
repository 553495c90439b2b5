// sort_check.cpp -- csrc/ebvo_sort.h (libstdc++'s std::sort restated) against the real std::sort of this toolchain,
// on index arrays with heavily tied scores, both comparators of apply_Best_Nearly_Best_Test, lengths 0 .. 3000.
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "../../edge_based_visual_odometry_amd/csrc/ebvo_sort.h"

int main()
{
    std::mt19937 rng(12345);
    long checked = 0;
    for (int trial = 0; trial < 4000; ++trial)
    {
        const int n = trial < 400 ? trial % 40 : (int)(rng() % (trial % 7 == 0 ? 3000 : 200));
        const int levels = 1 + (int)(rng() % 12); // few distinct scores: many ties
        std::vector<double> score(n);
        for (double &s : score)
            s = (double)(rng() % levels) * 0.125;
        if (trial % 5 == 0) // already sorted / reversed inputs hit the depth limit more easily
            std::sort(score.begin(), score.end());
        if (trial % 10 == 0)
            std::reverse(score.begin(), score.end());
        for (int desc = 0; desc < 2; ++desc)
        {
            std::vector<size_t> ref(n);
            std::iota(ref.begin(), ref.end(), 0);
            if (desc)
                std::sort(ref.begin(), ref.end(), [&](size_t a, size_t b) { return score[a] > score[b]; });
            else
                std::sort(ref.begin(), ref.end(), [&](size_t a, size_t b) { return score[a] < score[b]; });
            std::vector<int32_t> got(n);
            std::iota(got.begin(), got.end(), 0);
            ebvo_sort_cmp c{score.data(), desc};
            ebvo_std_sort(got.data(), n, &c);
            for (int k = 0; k < n; ++k)
                if ((size_t)got[k] != ref[k])
                {
                    std::printf("MISMATCH trial %d n %d desc %d at %d: %d vs %zu\n", trial, n, desc, k, got[k], ref[k]);
                    return 1;
                }
            ++checked;
        }
    }
    // an adversarial input that drives quicksort to its depth limit (the heapsort branch): organ-pipe with ties
    for (int n : {100, 1000, 5000})
    {
        std::vector<double> score(n);
        for (int k = 0; k < n; ++k)
            score[k] = (double)(k < n / 2 ? k : n - k) * (k % 3 == 0 ? 1.0 : 0.5);
        std::vector<size_t> ref(n);
        std::iota(ref.begin(), ref.end(), 0);
        std::sort(ref.begin(), ref.end(), [&](size_t a, size_t b) { return score[a] > score[b]; });
        std::vector<int32_t> got(n);
        std::iota(got.begin(), got.end(), 0);
        ebvo_sort_cmp c{score.data(), 1};
        ebvo_std_sort(got.data(), n, &c);
        for (int k = 0; k < n; ++k)
            if ((size_t)got[k] != ref[k])
            {
                std::printf("MISMATCH organ pipe n %d at %d\n", n, k);
                return 1;
            }
        ++checked;
    }
    std::printf("ok %ld\n", checked);
    return 0;
}

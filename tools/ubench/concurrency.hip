// How many kernels of different streams run at the same time: N streams, one single-block kernel each that spins for ~200 us.
// If K kernels can be resident together the batch takes ceil(N / K) x 200 us.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/concurrency.hip -o tools/ubench/concurrency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long ticks, int *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
        ;
    if (out && threadIdx.x == 0)
        out[blockIdx.x] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int max_streams = 16;
    std::vector<hipStream_t> st(max_streams);
    for (auto &s : st)
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d = nullptr;
    CK(hipMalloc(&d, 4096 * sizeof(int)));
    const unsigned long long ticks = 20000; // 200 us
    for (int blocks : {1, 256, 2048})
        for (int n : {1, 2, 3, 4, 6, 8, 12, 16})
        {
            for (int rep = 0; rep < 2; ++rep) // the first pass warms up
            {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int k = 0; k < n; ++k)
                    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, st[k], ticks, d);
                CK(hipDeviceSynchronize());
                const double dt = now() - t0;
                if (rep)
                    printf("blocks %4d  streams %2d: %7.1f us  -> %.2f kernels at a time\n", blocks, n, dt * 1e6, n * 200.0 / (dt * 1e6 - 15.0));
            }
        }
    return 0;
}

// Host cost and device time of a 30-kernel dependent chain: direct launches vs one hipGraphLaunch (stream capture).
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/graph_launch.hip -o gpurun_out/graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void small_kernel(int *p, int n, int it)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = p[i] * 3 + it;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int n = 1 << 16, chain = 30, reps = 200, nstreams = 4;
    std::vector<hipStream_t> st(nstreams);
    std::vector<int *> buf(nstreams);
    for (int k = 0; k < nstreams; ++k)
    {
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        CK(hipMalloc(&buf[k], n * sizeof(int)));
        CK(hipMemset(buf[k], 0, n * sizeof(int)));
    }
    auto chain_direct = [&](int k) {
        for (int c = 0; c < chain; ++c)
            hipLaunchKernelGGL(small_kernel, dim3(n / 256), dim3(256), 0, st[k], buf[k], n, c);
    };
    std::vector<hipGraphExec_t> exec(nstreams);
    for (int k = 0; k < nstreams; ++k)
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal));
        chain_direct(k);
        CK(hipStreamEndCapture(st[k], &g));
        CK(hipGraphInstantiate(&exec[k], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int mode = 0; mode < 2; ++mode)
        for (int ns = 1; ns <= nstreams; ns *= 2)
        {
            for (int w = 0; w < 20; ++w)
                for (int k = 0; k < ns; ++k)
                    if (mode) CK(hipGraphLaunch(exec[k], st[k])); else chain_direct(k);
            CK(hipDeviceSynchronize());
            const double t0 = now();
            double host = 0;
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < ns; ++k)
                {
                    const double a = now();
                    if (mode) CK(hipGraphLaunch(exec[k], st[k])); else chain_direct(k);
                    host += now() - a;
                }
            CK(hipDeviceSynchronize());
            const double t1 = now();
            printf("%-6s streams=%d  host %.1f us per chain of %d, wall %.1f us per chain (%.2f us per kernel)\n", mode ? "graph" : "direct",
                   ns, host / (reps * ns) * 1e6, chain, (t1 - t0) / (reps * ns) * 1e6, (t1 - t0) / (reps * ns * chain) * 1e6);
        }
    return 0;
}

// Do scalar / LDS instructions of a wave take issue slots away from the fp64 VALU stream of the SIMD (gfx950)?
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_mix issue_mix.hip ; run on the GPU box.
// Every step is 8 independent v_mul_f64 per wave plus N companion instructions of one kind; W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(double *out, double b)
{
    __shared__ double lds[512];
    lds[threadIdx.x] = b;
    lds[threadIdx.x + 256] = b;
    __syncthreads();
    double v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
        v[t] = 1.0 + (threadIdx.x + t) * 1e-3;
    unsigned sacc = blockIdx.x;
    unsigned addr = (threadIdx.x & 15) * 16;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 sink = {0.0, 0.0};
    for (int it = 0; it < ITER; ++it)
    {
#pragma unroll
        for (int t = 0; t < 8; ++t)
        {
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[t]) : "v"(b));
            if (t < N)
            {
                if (KIND == 1)
                    asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)::"scc");
                if (KIND == 2)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(sink) : "v"(addr));
                if (KIND == 3)
                    asm volatile("s_nop 0");
                if (KIND == 4)
                    asm volatile("v_add_u32 %0, %0, 1" : "+v"(addr));
            }
        }
        if (KIND == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sink));
    }
    double s = sink.x + sacc + addr;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        s += v[t];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N>
static void run(const char *name, double *d, int W)
{
    const int blocks = 256 * W, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(threads), 0, 0, d, 0.9999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(threads), 0, 0, d, 0.9999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)W * ITER; // per SIMD
    const double cyc = 2.4e9 * (ms * 1e-3) / steps;
    printf("W=%d  %-28s %8.3f ms  %6.2f cycles per step of 8 v_mul_f64 + %d companions (32 = VALU-bound)\n", W, name, ms,
           cyc, N);
}

int main()
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    for (int W : {1, 2, 4, 8})
    {
        if (W == 1) { run<0, 0>("mul only", d, 1); run<1, 4>("+4 s_add", d, 1); run<2, 2>("+2 ds_read_b128", d, 1); run<3, 4>("+4 s_nop", d, 1); run<4, 4>("+4 v_add_u32", d, 1); }
        if (W == 2) { run<0, 0>("mul only", d, 2); run<1, 4>("+4 s_add", d, 2); run<2, 2>("+2 ds_read_b128", d, 2); run<3, 4>("+4 s_nop", d, 2); run<4, 4>("+4 v_add_u32", d, 2); }
        if (W == 4) { run<0, 0>("mul only", d, 4); run<1, 4>("+4 s_add", d, 4); run<2, 2>("+2 ds_read_b128", d, 4); run<3, 4>("+4 s_nop", d, 4); run<4, 4>("+4 v_add_u32", d, 4); }
        if (W == 8) { run<0, 0>("mul only", d, 8); run<1, 4>("+4 s_add", d, 8); run<2, 2>("+2 ds_read_b128", d, 8); run<3, 4>("+4 s_nop", d, 8); run<4, 4>("+4 v_add_u32", d, 8); }
    }
    return 0;
}

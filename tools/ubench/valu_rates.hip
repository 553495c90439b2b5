// Issue cost of the fp64 / conversion instructions the sampling and candidate kernels lean on (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
// Each kernel runs ITER iterations of 8 independent chains of one operation per lane; cost = SIMD-cycles per
// wave-instruction = (n_simd * clock * time) / (waves * ITER * 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int ia)
{
    double v[8];
    int iv[8];
    unsigned uv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
    {
        v[t] = a + (threadIdx.x + t) * 1e-3;
        iv[t] = ia + t;
        uv[t] = (unsigned)(ia + t);
    }
    for (int it = 0; it < ITER; ++it)
    {
#pragma unroll
        for (int t = 0; t < 8; ++t)
        {
            if (OP == 0) v[t] = v[t] * b;                                   // v_mul_f64
            if (OP == 1) v[t] = v[t] + b;                                   // v_add_f64
            if (OP == 2) v[t] = __builtin_fma(v[t], b, a);                  // v_fma_f64
            if (OP == 3) v[t] = __builtin_floor(v[t]) + b;                  // v_floor_f64 + add
            if (OP == 4) v[t] = __builtin_ceil(v[t]) + b;                   // v_ceil_f64 + add
            if (OP == 5) { iv[t] = (int)v[t]; v[t] = v[t] + b; asm volatile("" : "+v"(iv[t])); } // v_cvt_i32_f64 + add
            if (OP == 6) { v[t] = (double)iv[t]; asm volatile("" : "+v"(v[t])); iv[t] += 1; }   // v_cvt_f64_i32 + iadd
            if (OP == 7) { v[t] = (double)uv[t]; asm volatile("" : "+v"(v[t])); uv[t] += 1; }   // v_cvt_f64_u32 + iadd
            if (OP == 8) { float f = (float)v[t]; asm volatile("" : "+v"(f)); v[t] = v[t] + b; out[0] = f > 1e30f ? 1.0 : out[0]; } // v_cvt_f32_f64 + add
            if (OP == 9) v[t] = __builtin_fmin(v[t], b);                    // v_min_f64
            if (OP == 10) { iv[t] = iv[t] * ia + t; }                       // v_mul_lo / mad
            if (OP == 11) { iv[t] = iv[t] + ia; asm volatile("" : "+v"(iv[t])); } // v_add_u32
            if (OP == 12) v[t] = __builtin_sqrt(v[t]);                      // sqrt sequence
            if (OP == 13) v[t] = a / v[t];                                  // division sequence
            if (OP == 14) { bool c = v[t] < b; v[t] = c ? v[t] + b : v[t]; } // cmp + cndmask*2 + add
            if (OP == 15) { float f = (float)iv[t]; asm volatile("" : "+v"(f)); iv[t] += (int)f; } // f32 cvts
        }
    }
    double s = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        s += v[t] + iv[t] + uv[t];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, double *d, int extra_ops)
{
    const int blocks = 256 * 8, threads = 256; // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 0.9999999, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 0.9999999, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double winstr = waves * ITER * 8.0;
    const double clk = 2.4e9; // nominal; relative numbers are what matters
    const double cyc = 1024.0 * clk * (ms * 1e-3) / winstr;
    printf("%-34s %8.3f ms  %6.2f SIMD-cycles per loop step (incl. %d companion op)\n", name, ms, cyc, extra_ops);
}

int main()
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    run<0>("v_mul_f64", d, 0);
    run<1>("v_add_f64", d, 0);
    run<2>("v_fma_f64", d, 0);
    run<3>("v_floor_f64 + v_add_f64", d, 1);
    run<4>("v_ceil_f64 + v_add_f64", d, 1);
    run<5>("v_cvt_i32_f64 + v_add_f64", d, 1);
    run<6>("v_cvt_f64_i32 + v_add_u32", d, 1);
    run<7>("v_cvt_f64_u32 + v_add_u32", d, 1);
    run<8>("v_cvt_f32_f64 + v_add_f64 (+cmp)", d, 2);
    run<9>("v_min_f64", d, 0);
    run<10>("v_mad_u32 (mul_lo + add)", d, 0);
    run<11>("v_add_u32", d, 0);
    run<12>("sqrt(double) sequence", d, 0);
    run<13>("double division sequence", d, 0);
    run<14>("v_cmp_f64 + cndmask x2 + add", d, 0);
    run<15>("v_cvt_f32_i32 + v_cvt_i32_f32 + add", d, 0);
    hipFree(d);
    return 0;
}

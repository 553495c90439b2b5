// VGPR bank effects on fp64 VALU issue (gfx950): v_mul_f64 / v_add_f64 with both sources in the same bank pair, in
// different bank pairs, or one source in an SGPR.  8 waves per SIMD, 8 independent chains per wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o bank_conflict bank_conflict.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048
// chains live in v[10:11] .. v[38:39] step 4 (pairs starting at even multiples of 2: banks {2,3},{2,3}...)
#define STEP8(OP, SRC)                                                                                               \
    OP " v[10:11], v[10:11], " SRC "\n\t" OP " v[14:15], v[14:15], " SRC "\n\t" OP " v[18:19], v[18:19], " SRC "\n\t"  \
    OP " v[22:23], v[22:23], " SRC "\n\t" OP " v[26:27], v[26:27], " SRC "\n\t" OP " v[30:31], v[30:31], " SRC "\n\t"  \
    OP " v[34:35], v[34:35], " SRC "\n\t" OP " v[38:39], v[38:39], " SRC "\n\t"
#define CLOB "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23", "v26", "v27", "v30", "v31", "v34", "v35", "v38", "v39", "v42", "v43", "v44", "v45"

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double b)
{
    // v[42:43] = bank pair {2,3} (same as the chains), v[44:45] = bank pair {0,1}
    asm volatile("v_mov_b32 v42, %0\n\tv_mov_b32 v43, %1\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %1\n\t"
                 "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %1\n\t"
                 "v_mov_b32 v18, %0\n\tv_mov_b32 v19, %1\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %1\n\t"
                 "v_mov_b32 v26, %0\n\tv_mov_b32 v27, %1\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %1\n\t"
                 "v_mov_b32 v34, %0\n\tv_mov_b32 v35, %1\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %1\n\t"
                 :
                 : "v"(__double2loint(b)), "v"(__double2hiint(b))
                 : CLOB);
    for (int it = 0; it < ITER; ++it)
    {
        if (KIND == 0) asm volatile(STEP8("v_mul_f64", "v[42:43]") ::: CLOB);
        if (KIND == 1) asm volatile(STEP8("v_mul_f64", "v[44:45]") ::: CLOB);
        if (KIND == 2) asm volatile(STEP8("v_mul_f64", "%0") ::"s"(b) : CLOB);
        if (KIND == 3) asm volatile(STEP8("v_add_f64", "v[42:43]") ::: CLOB);
        if (KIND == 4) asm volatile(STEP8("v_add_f64", "v[44:45]") ::: CLOB);
        if (KIND == 5) asm volatile(STEP8("v_add_f64", "%0") ::"s"(b) : CLOB);
        if (KIND == 6) asm volatile(STEP8("v_fma_f64", "v[44:45], v[44:45]") ::: CLOB);
        if (KIND == 7) asm volatile(STEP8("v_fma_f64", "v[42:43], v[44:45]") ::: CLOB);
    }
    double r;
    asm volatile("v_mov_b32 %0, v10" : "=v"(((int *)&r)[0])::CLOB);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, double *d)
{
    const int W = 8, blocks = 256 * W, threads = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, d, 0.9999999);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, d, 0.9999999);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %5.2f cycles (at 2.4 GHz) per wave instruction\n", name, ms,
           2.4e9 * (ms * 1e-3) / ((double)W * ITER * 8));
}

int main()
{
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    run<0>("v_mul_f64 v, v, v   (same bank pair)", d);
    run<1>("v_mul_f64 v, v, v   (other bank pair)", d);
    run<2>("v_mul_f64 v, v, s", d);
    run<3>("v_add_f64 v, v, v   (same bank pair)", d);
    run<4>("v_add_f64 v, v, v   (other bank pair)", d);
    run<5>("v_add_f64 v, v, s", d);
    run<6>("v_fma_f64 v, v, o, o (other bank pair)", d);
    run<7>("v_fma_f64 v, v, same, other", d);
    return 0;
}

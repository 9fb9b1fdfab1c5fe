// ubench_intmul.hip -- integer-multiply issue rate on gfx950.
// The RNS hot path is 64-bit modular arithmetic built from 32-bit multiplies,
// so this is its second roofline next to HBM (SURVEY.md 7 "hard parts").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

#define ITERS 4096
#define CH 16

template <int KIND>
__global__ __launch_bounds__(256) void k(u64* out, u32 a0, u32 b0)
{
    u64 acc[CH];
    u32 x = a0 + threadIdx.x, y = b0 + blockIdx.x;
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = x * (c + 1) + y;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (KIND == 0) { // v_mad_u64_u32
                acc[c] = (u64) (u32) acc[c] * (u64) x + acc[c];
            } else if (KIND == 1) { // v_mul_lo_u32
                acc[c] = (u32) acc[c] * x + 1;
            } else if (KIND == 2) { // v_mul_hi_u32
                acc[c] = __umulhi((u32) acc[c], x) + y;
            } else if (KIND == 3) { // 64-bit add, operand varies per iteration
                acc[c] = acc[c] + acc[(c + 1) % CH];
            } else if (KIND == 7) { // 32-bit add
                acc[c] = (u32) acc[c] + (u32) acc[(c + 1) % CH];
            } else if (KIND == 8) { // compare + select (v_cmp_u64 + 2 cndmask)
                acc[c] = (acc[c] >= acc[(c + 1) % CH]) ? acc[c] - 7 : acc[(c + 2) % CH];
            } else if (KIND == 9) { // xor 32
                acc[c] = (u32) acc[c] ^ ((u32) acc[(c + 1) % CH] >> 1);
            } else if (KIND == 4) { // v_fma_f64
                double d = __longlong_as_double(acc[c]);
                d = fma(d, 1.0000001, 0.5);
                acc[c] = __double_as_longlong(d);
            } else if (KIND == 5) { // v_fma_f32
                float f = __uint_as_float((u32) acc[c]);
                f = fmaf(f, 1.0000001f, 0.5f);
                acc[c] = __float_as_uint(f);
            } else if (KIND == 10) { // v_add_f64
                double d = __longlong_as_double(acc[c]);
                d = d + __longlong_as_double(acc[(c + 1) % CH]);
                acc[c] = __double_as_longlong(d);
            } else if (KIND == 11) { // v_mul_f64
                double d = __longlong_as_double(acc[c]);
                d = d * 1.0000001;
                acc[c] = __double_as_longlong(d);
            } else if (KIND == 12) { // v_rndne_f64 (+ add so it is not idempotent-folded)
                double d = __longlong_as_double(acc[c]);
                d = __builtin_rint(d) + 0.5;
                acc[c] = __double_as_longlong(d);
            } else if (KIND == 13) { // v_cvt_f64_u32 + v_cvt_u32_f64
                double d = (double) (u32) acc[c];
                acc[c] = (u32) (d * 0.5);
            } else if (KIND == 14) { // v_lshl_add_u64
                acc[c] = (acc[c] << 1) + acc[(c + 1) % CH];
            } else if (KIND == 6) { // v_mul_u32_u24 (full-rate 24-bit)
                acc[c] = __umul24((u32) acc[c], x) + 1;
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, u64* d)
{
    const int blocks = 256 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double) blocks * 256 * ITERS * CH;
    printf("%-16s %8.3f ms  %8.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
}

int main()
{
    u64* d; hipMalloc(&d, 256 * 16 * 256 * 8);
    run<0>("v_mad_u64_u32", d);
    run<1>("v_mul_lo_u32", d);
    run<2>("v_mul_hi_u32", d);
    run<3>("add_u64", d);
    run<4>("v_fma_f64", d);
    run<5>("v_fma_f32", d);
    run<6>("v_mul_u32_u24", d);
    run<7>("add_u32", d);
    run<8>("cmp64+sel+sub", d);
    run<9>("xor+shift32", d);
    run<10>("v_add_f64", d);
    run<11>("v_mul_f64", d);
    run<12>("rndne_f64+add", d);
    run<13>("cvt f64<->u32+mul", d);
    run<14>("v_lshl_add_u64", d);
    return 0;
}

// mad_dep.hip -- issue rate of v_mad_u64_u32 / v_lshl_add_u64 / v_add_co chains as a function of the number of
// independent chains a wave interleaves (ILP) and of the waves per SIMD.  Answers: does a dependent
// v_mad_u64_u32 stall the wave, and how many waves hide it?
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mad_dep.hip -o tools/exp/mad_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP, int OP>
__global__ __launch_bounds__(256) void k(u64* out, u32 a, u32 b, int iters)
{
    u64 x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = threadIdx.x + i;
    u32 m = a + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                u64 cy;
                if (OP == 0) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(x[i]), "=s"(cy) : "v"(m), "v"(b));
                else if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % ILP]));
                else if (OP == 2) { u32 lo = (u32) x[i]; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(x[i]), "=s"(cy) : "v"(lo), "v"(b)); }
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= x[i];
    out[(size_t) blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ILP, int OP>
void run(const char* name, int waves_per_simd, u64* out)
{
    // 256 CUs; a 256-thread block = 4 waves = 1 per SIMD; blocks per CU = waves_per_simd
    const int blocks = 256 * waves_per_simd, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double) blocks * 256 * iters * 16 * ILP;
    // cycles per wave-instruction per SIMD at 2.4 GHz: time * f / (instructions per SIMD)
    const double inst_per_simd = (double) waves_per_simd * iters * 16 * ILP;
    printf("%-28s ILP %d  waves/SIMD %d : %7.2f T lane-ops/s  %5.2f cycles/instr/SIMD (at 2.4 GHz)\n", name, ILP, waves_per_simd,
           ops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / inst_per_simd);
}

int main()
{
    u64* out; CK(hipMalloc((void**) &out, (size_t) 256 * 8 * 256 * 8));
    for (int w : {1, 2, 4}) {
        run<1, 0>("mad64 (addend chain)", w, out); run<2, 0>("mad64 (addend chain)", w, out);
        run<4, 0>("mad64 (addend chain)", w, out); run<8, 0>("mad64 (addend chain)", w, out);
        run<1, 2>("mad64 (multiplier chain)", w, out); run<2, 2>("mad64 (multiplier chain)", w, out);
        run<4, 2>("mad64 (multiplier chain)", w, out);
        run<2, 1>("lshl_add_u64", w, out); run<4, 1>("lshl_add_u64", w, out);
    }
    return 0;
}

// mall_bw.hip -- read+write bandwidth of an in-place stream (x[i] += 1, 16 B per lane) as a function of the
// working set: below the 256 MiB Infinity Cache the lines written by one launch are read by the next one
// from the cache instead of HBM.  Question: is a two-pass transform whose intermediate stays below that
// size faster than the 5.25 TB/s read+write ceiling of an HBM stream (tools/exp/copy_bw.hip)?
// Also: producer/consumer pair (kernel A writes buffer T from S, kernel B reads T and writes D), chunked so
// that T is re-used while S and D stream -- the shape of column pass -> row pass.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mall_bw.hip -o tools/exp/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void inc(ulonglong2* p, size_t n)
{
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t) gridDim.x * 256;
    for (; i < n; i += stride) { ulonglong2 v = p[i]; v.x += 1; v.y += 1; p[i] = v; }
}
__global__ __launch_bounds__(256) void cp(const ulonglong2* __restrict__ s, ulonglong2* __restrict__ d, size_t n)
{
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t) gridDim.x * 256;
    for (; i < n; i += stride) { ulonglong2 v = s[i]; v.x += 1; d[i] = v; }
}

int main()
{
    const size_t total = (size_t) 4 << 30; // bytes per buffer
    ulonglong2 *S, *T, *D;
    CK(hipMalloc((void**) &S, total)); CK(hipMalloc((void**) &T, total)); CK(hipMalloc((void**) &D, total));
    CK(hipMemset(S, 1, total)); CK(hipMemset(T, 1, total)); CK(hipMemset(D, 1, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("in-place x += 1, repeated launches over the same working set\n");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 2048}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        const int reps = (int) ((size_t) 16384 / mb) + 4;
        const int grid = (int) ((n / 256 < 256 * 32) ? n / 256 : 256 * 32);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(inc, dim3(grid), dim3(256), 0, 0, S, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(inc, dim3(grid), dim3(256), 0, 0, S, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %5zu MiB: %8.1f GB/s read+write (%.1f us per launch)\n", mb, 2.0 * bytes * reps / (ms * 1e-3) / 1e9, ms * 1e3 / reps);
    }
    printf("S -> T -> D in chunks (T chunk re-used in place of a full-size intermediate); bytes counted: S read + D write\n");
    for (size_t mb : {16, 32, 64, 128, 256, 4096}) {
        const size_t chunk = mb << 20, n = chunk / 16, chunks = total / chunk;
        const int grid = (int) ((n / 256 < 256 * 32) ? n / 256 : 256 * 32);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (size_t c = 0; c < chunks; c++) {
                const ulonglong2* s = S + c * n;
                ulonglong2* t = (mb == 4096) ? T : T; // one chunk-sized intermediate, re-used
                ulonglong2* d = D + c * n;
                hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, s, (mb == 4096) ? T : t, n);
                hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, (const ulonglong2*) t, d, n);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  chunk %5zu MiB: %8.1f GB/s algorithmic (S read + D write; %.3f ms for 4 GiB)\n", mb, 2.0 * total / (ms * 1e-3) / 1e9, ms);
    }
    return 0;
}

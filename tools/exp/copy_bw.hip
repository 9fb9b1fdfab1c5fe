// copy_bw.hip -- what a read+write stream sustains on one MI355X for the access shapes the NTT
// passes use (tools/exp, timing only).  Every variant moves `bytes` in and `bytes` out, 256 threads
// per workgroup, 16 (8-byte) or 8 (16-byte) accesses per lane issued back to back, like the passes.
//   lin8   contiguous,             8 B per lane  (global_load/store_dwordx2)
//   lin16  contiguous,            16 B per lane  (dwordx4)
//   col8   column-pass shape:  256 rows x 16 columns, 128-byte segments 2 KiB apart, 8 B per lane
//   col16  column-pass shape:  256 rows x 32 columns, 256-byte segments 2 KiB apart, 16 B per lane
//   row8   row-pass shape: 16 rows of 256 contiguous, lane i0 takes i0 + 16 k, 8 B per lane
//   row16  same rows, lane j takes the pair 2 j + 32 k, 16 B per lane
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/copy_bw.hip -o tools/exp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void lin8(const u64* __restrict__ in, u64* __restrict__ out)
{
    const size_t base = (size_t) blockIdx.x * 4096 + threadIdx.x;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = in[base + 256 * k];
#pragma unroll
    for (int k = 0; k < 16; k++) out[base + 256 * k] = x[k] + 1;
}
__global__ __launch_bounds__(256) void lin16(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out)
{
    const size_t base = (size_t) blockIdx.x * 2048 + threadIdx.x;
    ulonglong2 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = in[base + 256 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k].x += 1; out[base + 256 * k] = x[k]; }
}
// limb = 65536 elements; tile (limb, column block)
__global__ __launch_bounds__(256) void col8(const u64* __restrict__ in, u64* __restrict__ out)
{
    const size_t limb = blockIdx.x >> 4, cb = blockIdx.x & 15;
    const int t = threadIdx.x, col = t & 15, r1 = t >> 4;
    const size_t base = limb * 65536 + cb * 16 + col;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = in[base + (size_t) (r1 + 16 * k) * 256];
#pragma unroll
    for (int k = 0; k < 16; k++) out[base + (size_t) (16 * r1 + k) * 256] = x[k] + 1;
}
__global__ __launch_bounds__(256) void col16(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out)
{
    // 256 rows x 32 columns per workgroup: 16 pairs per row, lane = (pair, row group of 16)
    const size_t limb = blockIdx.x >> 3, cb = blockIdx.x & 7;
    const int t = threadIdx.x, pr = t & 15, r1 = t >> 4;
    const size_t base = limb * 32768 + cb * 16 + pr; // in 16-byte units: row = 128 units
    ulonglong2 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = in[base + (size_t) (r1 + 16 * k) * 128];
#pragma unroll
    for (int k = 0; k < 16; k++) { x[k].x += 1; out[base + (size_t) (16 * r1 + k) * 128] = x[k]; }
}
__global__ __launch_bounds__(256) void row8(const u64* __restrict__ in, u64* __restrict__ out)
{
    const int t = threadIdx.x, row = t >> 4, i0 = t & 15;
    const size_t base = (size_t) blockIdx.x * 4096 + row * 256 + i0;
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = in[base + 16 * k];
#pragma unroll
    for (int k = 0; k < 16; k++) out[base + 16 * k] = x[k] + 1;
}
__global__ __launch_bounds__(256) void row16(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out)
{
    const int t = threadIdx.x, row = t >> 4, j = t & 15;
    const size_t base = (size_t) blockIdx.x * 2048 + row * 128 + j;
    ulonglong2 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = in[base + 16 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k].x += 1; out[base + 16 * k] = x[k]; }
}
// the same contiguous stream with non-temporal hints (`nt`): loads, stores, both; and a read-only stream
template <int MODE>
__global__ __launch_bounds__(256) void lin16nt(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out)
{
    const size_t base = (size_t) blockIdx.x * 2048 + threadIdx.x;
    ulonglong2 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (MODE & 1) {
            x[k].x = __builtin_nontemporal_load(&in[base + 256 * k].x);
            x[k].y = __builtin_nontemporal_load(&in[base + 256 * k].y);
        } else x[k] = in[base + 256 * k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x[k].x += 1;
        if (MODE & 2) {
            __builtin_nontemporal_store(x[k].x, &out[base + 256 * k].x);
            __builtin_nontemporal_store(x[k].y, &out[base + 256 * k].y);
        } else out[base + 256 * k] = x[k];
    }
}
__global__ __launch_bounds__(256) void rd16(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out)
{
    const size_t base = (size_t) blockIdx.x * 2048 + threadIdx.x;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { const ulonglong2 v = in[base + 256 * k]; s += v.x ^ v.y; }
    if (s == 0x1234567887654321ull) out[base] = make_ulonglong2(s, s);
}
// write-only / read-only streams of the column-pass shape (the decomposing pass reads from L2)
__global__ __launch_bounds__(256) void wr8(u64* __restrict__ out)
{
    const size_t limb = blockIdx.x >> 4, cb = blockIdx.x & 15;
    const int t = threadIdx.x, col = t & 15, r1 = t >> 4;
    const size_t base = limb * 65536 + cb * 16 + col;
#pragma unroll
    for (int k = 0; k < 16; k++) out[base + (size_t) (16 * r1 + k) * 256] = base + k;
}
__global__ __launch_bounds__(256) void wr16(ulonglong2* __restrict__ out)
{
    const size_t limb = blockIdx.x >> 3, cb = blockIdx.x & 7;
    const int t = threadIdx.x, pr = t & 15, r1 = t >> 4;
    const size_t base = limb * 32768 + cb * 16 + pr;
#pragma unroll
    for (int k = 0; k < 16; k++) out[base + (size_t) (16 * r1 + k) * 128] = make_ulonglong2(base, k);
}

int main(int argc, char** argv)
{
    const size_t limbs = argc > 1 ? atol(argv[1]) : 4352; // 512 KiB each
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const size_t bytes = limbs * 65536 * 8;
    u64 *in, *out;
    CK(hipMalloc((void**) &in, bytes));
    CK(hipMalloc((void**) &out, bytes));
    CK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double factor) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-14s %6zu limbs (%7.1f MiB each way): %8.3f ms  %7.1f GB/s\n", name, limbs, bytes / 1048576.0, ms,
               factor * bytes / (ms * 1e-3) / 1e9);
    };
    const unsigned g8 = (unsigned) (limbs * 16), g16c = (unsigned) (limbs * 8);
    run("lin8", [&] { hipLaunchKernelGGL(lin8, dim3(g8), dim3(256), 0, 0, in, out); }, 2);
    run("lin16", [&] { hipLaunchKernelGGL(lin16, dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("col8", [&] { hipLaunchKernelGGL(col8, dim3(g8), dim3(256), 0, 0, in, out); }, 2);
    run("col16", [&] { hipLaunchKernelGGL(col16, dim3(g16c), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("row8", [&] { hipLaunchKernelGGL(row8, dim3(g8), dim3(256), 0, 0, in, out); }, 2);
    run("row16", [&] { hipLaunchKernelGGL(row16, dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("lin16 nt-load", [&] { hipLaunchKernelGGL((lin16nt<1>), dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("lin16 nt-store", [&] { hipLaunchKernelGGL((lin16nt<2>), dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("lin16 nt-both", [&] { hipLaunchKernelGGL((lin16nt<3>), dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 2);
    run("rd16", [&] { hipLaunchKernelGGL(rd16, dim3(g8), dim3(256), 0, 0, (const ulonglong2*) in, (ulonglong2*) out); }, 1);
    run("wr8", [&] { hipLaunchKernelGGL(wr8, dim3(g8), dim3(256), 0, 0, out); }, 1);
    run("wr16", [&] { hipLaunchKernelGGL(wr16, dim3(g16c), dim3(256), 0, 0, (ulonglong2*) out); }, 1);
    return 0;
}

// What a dependent chain of small kernels costs on this part, and what one more dependent load inside each costs.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/launch_chain.hip -o tools/exp/launch_chain && tools/exp/launch_chain
// Prints us per kernel for chains of 14 launches (HIP events around 50 repetitions of the chain):
//   empty kernels with a 600-byte argument block (what NttArgs is), 32 workgroups of 256 threads;
//   kernels that walk HOPS dependent 8-byte loads (a pointer chase through a table in HBM) before one streaming
//   load + store of 32 KiB per workgroup -- the shape of a column / row pass of 8 limbs at N = 2^14.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { unsigned long long pad[70]; const unsigned long long* table; const unsigned long long* in; unsigned long long* out; int hops; };
__global__ void k_empty(Args a) { if (a.hops < 0) a.out[0] = 1; }
__global__ __launch_bounds__(256) void k_chase(Args a)
{
    unsigned long long idx = blockIdx.x;
    for (int h = 0; h < a.hops; h++) idx = a.table[idx]; // wave-uniform: scalar loads, each depends on the last
    const unsigned long long* src = a.in + idx * 4096;
    unsigned long long* dst = a.out + (unsigned long long) blockIdx.x * 4096;
    unsigned long long v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = src[threadIdx.x + 256 * k];
#pragma unroll
    for (int k = 0; k < 16; k++) dst[threadIdx.x + 256 * k] = v[k] * 3 + 1;
}
int main()
{
    const int wgs = 32, chain = 14, reps = 50;
    unsigned long long *table, *in, *out;
    hipMalloc(&table, 4096 * 8); hipMalloc(&in, (size_t) wgs * 4096 * 8); hipMalloc(&out, (size_t) wgs * 4096 * 8);
    std::vector<unsigned long long> t(4096);
    for (int i = 0; i < 4096; i++) t[i] = (i * 7 + 3) % wgs;
    hipMemcpy(table, t.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipMemset(in, 1, (size_t) wgs * 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Args a{}; a.table = table; a.in = in; a.out = out;
    for (int mode = -1; mode <= 4; mode++) {
        a.hops = mode < 0 ? 0 : mode;
        auto run = [&] { for (int c = 0; c < chain; c++) { if (mode < 0) hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, a); else hipLaunchKernelGGL(k_chase, dim3(wgs), dim3(256), 0, 0, a); } };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (mode < 0) printf("empty kernel, 600-byte arguments:            %.2f us per kernel\n", ms * 1e3 / reps / chain);
        else printf("%d dependent scalar loads + 32 KiB per workgroup: %.2f us per kernel\n", mode, ms * 1e3 / reps / chain);
    }
    return 0;
}

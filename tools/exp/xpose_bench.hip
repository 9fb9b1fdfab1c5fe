// xpose_bench.hip -- the 16 x 16 exchange between the register rounds of the row pass (lane i of a 16-lane
// group holds elements i + 16 k, afterwards 16 i + k), done two ways on a full chip, 64-bit elements:
//   lds   wave-private LDS: 16 ds_write_b64, fence, 8 ds_read_b128 (what ntt.hip does)
//   dpp   no memory: four butterfly steps of cross-lane moves inside the 16-lane DPP row (the
//         "__shfl butterfly" the task statement names); gfx950 has quad_perm / row_ror but no xor-by-4
//         cross-lane move and no DPP on 64-bit moves, so a step costs ~10 VALU per register pair
// Both variants run the same FP64 filler work (ROUNDS x 16 fma) around the exchange so that the
// comparison is made where the kernels live: with the vector ALU as the busy unit.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/xpose_bench.hip -o tools/exp/xpose_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int phys(int e) { return e ^ (((e >> 5) & 7) << 1); }

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const u64 b = (u64) __double_as_longlong(v);
    const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) b, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long) (((u64) hi << 32) | lo));
}
// value of lane ^ D inside the 16-lane row
template <int D>
__device__ __forceinline__ double xor_lane(double v, int lane)
{
    if constexpr (D == 8) return dpp_mov<0x128>(v);          // row_ror:8
    else if constexpr (D == 1) return dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    else {                                                   // xor 4: ror 4 for lanes with bit 2 set, ror 12 otherwise
        const double a = dpp_mov<0x124>(v), b = dpp_mov<0x12C>(v);
        return (lane & 4) ? a : b;
    }
}
template <int D>
__device__ __forceinline__ void step(double (&x)[16], int lane)
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j & D) continue;
        const bool up = lane & D;
        const double send = up ? x[j] : x[j | D];
        const double recv = xor_lane<D>(send, lane);
        if (up) x[j] = recv; else x[j | D] = recv;
    }
}

template <int MODE, int ROUNDS>
__global__ __launch_bounds__(256) void k(double* out, double seed)
{
    __shared__ __attribute__((aligned(16))) u64 lds[4096];
    const int t = threadIdx.x, row = t >> 4, i0 = t & 15;
    double x[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) x[k2] = seed + t * 16 + k2;
    for (int r = 0; r < ROUNDS; r++) {
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) x[k2] = __builtin_fma(x[k2], 1.0000001, 0.5);
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) lds[phys(row * 256 + i0 + 16 * k2)] = (u64) __double_as_longlong(x[k2]);
            wave_fence();
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) {
                ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[phys(row * 256 + 16 * i0 + 2 * k2)]);
                x[2 * k2] = __longlong_as_double((long long) v.x);
                x[2 * k2 + 1] = __longlong_as_double((long long) v.y);
            }
            wave_fence();
        } else if (MODE == 1) {
            step<8>(x, i0); step<4>(x, i0); step<2>(x, i0); step<1>(x, i0);
        }
    }
    double s = 0;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) s += x[k2];
    out[(size_t) blockIdx.x * 256 + t] = s;
}

int main()
{
    const int blocks = 256 * 16, reps = 5;
    double* out;
    CK(hipMalloc((void**) &out, (size_t) blocks * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms per launch\n", name, ms / reps);
        return ms / reps;
    };
    constexpr int R = 256;
    const float base = run("filler only (16 fma/round)", [&] { hipLaunchKernelGGL((k<2, R>), dim3(blocks), dim3(256), 0, 0, out, 1.0); });
    const float l = run("filler + LDS exchange", [&] { hipLaunchKernelGGL((k<0, R>), dim3(blocks), dim3(256), 0, 0, out, 1.0); });
    const float d = run("filler + DPP butterfly", [&] { hipLaunchKernelGGL((k<1, R>), dim3(blocks), dim3(256), 0, 0, out, 1.0); });
    const double xch = (double) blocks * 4 * R; // wave-level exchanges per launch
    printf("per 16x16x64-lane exchange: LDS %.1f ns, DPP %.1f ns of chip time (%d waves resident per SIMD)\n",
           (l - base) * 1e6 / xch * 1024, (d - base) * 1e6 / xch * 1024, 4);
    return 0;
}

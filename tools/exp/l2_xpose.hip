// l2_xpose.hip -- can the intermediate of a two-pass transform stay in the L2 of one XCD?
// Persistent kernel, 2 workgroups per CU; workgroup b belongs to team b % 8 (the XCD it is dispatched to).  A team
// works on its own limbs (512 KiB each, N = 2^16) two at a time: 32 workgroups run the "column pass" of batch s
// (read the limb in 128-byte segments 2 KiB apart from HBM, write a team-private scratch slot), the other 32 the
// "row pass" of batch s - 1 (read the scratch slot contiguously, write the result to HBM).  Two scratch slots per
// team (2 x 2 limbs = 2 MiB of the 4 MiB L2), handed over through counters.  Counted bytes: input read + output
// written (2 W per limb, the accounting of the NTT roofline).  Modes: fence scope agent (what the memory model
// asks for across XCDs) or workgroup + L1-bypassing scratch loads (enough inside one XCD).
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/l2_xpose.hip -o l2_xpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define LIMB 65536
#define TEAMS 8
#define TEAM_WGS 64
#define BATCH 2

struct Ctl { unsigned a_done[2]; unsigned b_done[2]; unsigned err; unsigned pad[11]; };

template <int MODE>
__device__ __forceinline__ bool wait_for(unsigned* p, unsigned target, unsigned* err)
{
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { atomicExch(err, 1u); return false; }
    }
    return true;
}

// MODE 0: agent-scope release / acquire.  MODE 1: workgroup-scope fences, scratch read with sc0 (L1 bypass).
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void xpose(const u64* __restrict__ in, u64* __restrict__ out, u64* __restrict__ scratch,
                                             Ctl* ctl, int limbs_per_team)
{
    const int team = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int t = threadIdx.x;
    Ctl* c = ctl + team;
    u64* sc = scratch + (u64) team * 2 * BATCH * LIMB;
    const int batches = limbs_per_team / BATCH;
    const bool colside = rank < 32;
    const int sub = rank & 31, lb = sub >> 4, tile = sub & 15; // limb of the batch, tile of the limb
    __shared__ int ok;
    for (int s = 0; s < batches + 1; s++) {
        if (colside) {
            if (s >= batches) break;
            const int par = s & 1;
            // slot `par` must have been read by the row side of batch s - 2
            if (s >= 2) {
                if (t == 0) ok = wait_for<MODE>(&c->b_done[par], 32u * (unsigned) (s / 2), &c->err);
                __syncthreads();
                if (!ok) return;
            }
            const u64 limb = (u64) team + 8ull * ((u64) s * BATCH + lb);
            const u64* p = in + limb * LIMB;
            u64* q = sc + ((u64) par * BATCH + lb) * LIMB;
            u64 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int e = ((t >> 4) + 16 * k) * 256 + tile * 16 + (t & 15);
                v[k] = NT ? __builtin_nontemporal_load(&p[e]) : p[e];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int e = ((t >> 4) + 16 * k) * 256 + tile * 16 + (t & 15);
                q[e] = v[k] + 1;
            }
            if (MODE == 0) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(&c->a_done[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (s == 0) continue;
            const int sb = s - 1, par = sb & 1;
            if (t == 0) ok = wait_for<MODE>(&c->a_done[par], 32u * (unsigned) (sb / 2 + 1), &c->err);
            __syncthreads();
            if (!ok) return;
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const u64 limb = (u64) team + 8ull * ((u64) sb * BATCH + lb);
            const u64* q = sc + ((u64) par * BATCH + lb) * LIMB + tile * 4096;
            u64* o = out + limb * LIMB + tile * 4096;
            u64 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (MODE == 1) v[k] = __hip_atomic_load(&q[t + 256 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v[k] = q[t + 256 * k];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (NT) __builtin_nontemporal_store(v[k] + 1, &o[t + 256 * k]);
                else o[t + 256 * k] = v[k] + 1;
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(&c->b_done[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the same two passes as two launches through a full-size intermediate
__global__ __launch_bounds__(256) void colpass(const u64* __restrict__ in, u64* __restrict__ out)
{
    const u64 limb = blockIdx.y; const int tile = blockIdx.x, t = threadIdx.x;
    const u64* p = in + limb * LIMB; u64* q = out + limb * LIMB;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = p[((t >> 4) + 16 * k) * 256 + tile * 16 + (t & 15)];
#pragma unroll
    for (int k = 0; k < 16; k++) q[((t >> 4) + 16 * k) * 256 + tile * 16 + (t & 15)] = v[k] + 1;
}
__global__ __launch_bounds__(256) void rowpass(const u64* __restrict__ in, u64* __restrict__ out)
{
    const u64 limb = blockIdx.y; const int tile = blockIdx.x, t = threadIdx.x;
    const u64* p = in + limb * LIMB + tile * 4096; u64* q = out + limb * LIMB + tile * 4096;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = p[t + 256 * k];
#pragma unroll
    for (int k = 0; k < 16; k++) q[t + 256 * k] = v[k] + 1;
}

int main()
{
    const int limbs = 4096; // 2 GiB each way
    const size_t bytes = (size_t) limbs * LIMB * 8;
    u64 *S, *T, *D, *scr; Ctl* ctl;
    CK(hipMalloc((void**) &S, bytes)); CK(hipMalloc((void**) &T, bytes)); CK(hipMalloc((void**) &D, bytes));
    CK(hipMalloc((void**) &scr, (size_t) TEAMS * 2 * BATCH * LIMB * 8)); CK(hipMalloc((void**) &ctl, sizeof(Ctl) * TEAMS));
    CK(hipMemset(S, 0, bytes)); CK(hipMemset(D, 0xff, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(colpass, dim3(16, limbs), dim3(256), 0, 0, S, T);
        hipLaunchKernelGGL(rowpass, dim3(16, limbs), dim3(256), 0, 0, T, D);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("two launches through HBM:        %8.1f GB/s algorithmic (%.3f ms)\n", 2.0 * bytes / (ms * 1e-3) / 1e9, ms);
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(ctl, 0, sizeof(Ctl) * TEAMS));
            CK(hipMemset(D, 0xff, bytes));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((xpose<0, false>), dim3(TEAMS * TEAM_WGS), dim3(256), 0, 0, S, D, scr, ctl, limbs / TEAMS);
            if (mode == 1) hipLaunchKernelGGL((xpose<0, true>), dim3(TEAMS * TEAM_WGS), dim3(256), 0, 0, S, D, scr, ctl, limbs / TEAMS);
            if (mode == 2) hipLaunchKernelGGL((xpose<1, false>), dim3(TEAMS * TEAM_WGS), dim3(256), 0, 0, S, D, scr, ctl, limbs / TEAMS);
            if (mode == 3) hipLaunchKernelGGL((xpose<1, true>), dim3(TEAMS * TEAM_WGS), dim3(256), 0, 0, S, D, scr, ctl, limbs / TEAMS);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        Ctl h[TEAMS]; CK(hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost));
        unsigned err = 0; for (int i = 0; i < TEAMS; i++) err |= h[i].err;
        // every output word must be 2 (0 + 1 + 1)
        static u64 probe[4096];
        size_t bad = 0;
        for (int i = 0; i < 64; i++) {
            const size_t off = ((size_t) i * 9973 % limbs) * LIMB + (size_t) (i * 37 % 16) * 4096;
            CK(hipMemcpy(probe, D + off, sizeof(probe), hipMemcpyDeviceToHost));
            for (int j = 0; j < 4096; j++) bad += probe[j] != 2;
        }
        printf("persistent, scratch in L2, %s%s: %8.1f GB/s algorithmic (%.3f ms)  spin-timeout=%u wrong=%zu\n",
               mode < 2 ? "agent fences      " : "wg fences + sc0 ld", (mode & 1) ? ", nt streams" : "            ", 2.0 * bytes / (ms * 1e-3) / 1e9, ms, err, bad);
    }
    return 0;
}

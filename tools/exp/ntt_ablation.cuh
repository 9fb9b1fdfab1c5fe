// ntt_ablation.cuh -- knock-out versions of the hooks of heongpu_amd/csrc/ntt.hip, for tools/exp/ntt_exp.hip ONLY
// (results are deliberately wrong): NTT_EXP_MODE 1 = memory traffic only (butterflies skipped), 2 = arithmetic only,
// 3 = data traffic only (no butterflies, no twiddle loads).
#pragma once
#ifndef NTT_EXP_MODE
#define NTT_EXP_MODE 0
#endif
__device__ __forceinline__ u64 gld(const u64* p)
{
#if NTT_EXP_MODE == 2
    return (u64) (size_t) p * 0x9E3779B97F4A7C15ull >> 4;
#else
    return *p;
#endif
}
__device__ __forceinline__ void gst(u64* p, u64 v)
{
#if NTT_EXP_MODE == 2
    if (v == 0x123456789abcdefull) *p = v;
#else
    *p = v;
#endif
}
#if NTT_EXP_MODE == 1 || NTT_EXP_MODE == 3
#define NTT_ABLATE_BFLY(x, y, w) do { x ^= (w).x; y ^= (w).y; return; } while (0)
#define NTT_ABLATE_FPBFLY(x, y, w) do { x += __longlong_as_double((long long) (w).x); y -= __longlong_as_double((long long) (w).y); return; } while (0)
#else
#define NTT_ABLATE_BFLY(x, y, w)
#define NTT_ABLATE_FPBFLY(x, y, w)
#endif
#if NTT_EXP_MODE == 3 || NTT_EXP_MODE == 4 // 4: everything but the twiddle loads (values from registers: wrong results, same arithmetic)
#define NTT_ABLATE_TW(load, root0, s) make_ulonglong2(0x4030000000000000ull + (root0) + (s), 0x3cb0000000000000ull)
#else
#define NTT_ABLATE_TW(load, root0, s) (load)
#endif

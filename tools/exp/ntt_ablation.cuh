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

// mode 5 (timing only): the FP64 twiddle tables read as if they held plain doubles (8 bytes per twiddle, consecutive
// lanes consecutive), the companion recomputed -- what a compact table for the FP64 moduli would buy
#if NTT_EXP_MODE == 5
struct ConstTw;
__device__ __forceinline__ ulonglong2 ntt_exp_tw8(const ulonglong2* t, unsigned i, const FC& c)
{
    const double w = reinterpret_cast<const double*>(t)[i];
    const double wi = w * c.qi;
    return make_ulonglong2((unsigned long long) __double_as_longlong(w), (unsigned long long) __double_as_longlong(wi));
}
template <typename T> __device__ __forceinline__ ulonglong2 ntt_exp_tw8(T t, unsigned i, const FC&) { return t[i]; }
#define NTT_FP_TW(t, i, c) ntt_exp_tw8((t), (unsigned) (i), (c))
#else
#define NTT_FP_TW(t, i, c) ((t)[i])
#endif

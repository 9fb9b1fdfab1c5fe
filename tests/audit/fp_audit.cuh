// fp_audit.cuh -- TEST INFRASTRUCTURE, never part of heongpu_amd/lib/libhegpu.so.
//
// Included by heongpu_amd/csrc/fpmod.cuh (inside namespace hegpu) when the build names it with
// -DHEGPU_FP_AUDIT_HEADER='"...fp_audit.cuh"' (tests/audit/Makefile).  It fills the audit hooks of fpmod.cuh: the
// arithmetic itself stays the product's own code; what is added is, at EVERY executed operation,
//   * fp_mul:    y, w, k, t integers below 2^53, and t == y w - k q compared in exact 128-bit integer arithmetic;
//   * fp_reduce: x, r integers below 2^53, r == x (mod q), 2 |r| <= q (1 + 2^-40);
//   * fp_canon:  0 <= r < q;   fp_from_u64: v < 2^52;   fp_to_u64: an integer in [0, 2^52);
//   * every butterfly output / running sum handed to FP_AUDIT_VAL: an integer below 2^53;
// and the largest |v| / q seen per (site, stage, metric) in a device table, one per translation unit, read through
// hegpu_fp_audit_read_<tu>() (tests/test_gpu_fp_audit.py compares it with the bounds the product's comments claim).
#pragma once

#define FP_AUDIT_SITES (FPS_KINDS * 8)
#define FP_AUDIT_VIOLATIONS 8
// violation kinds
#define FPV_NONINTEGRAL 0
#define FPV_RANGE 1     // |v| >= 2^53 (or NaN)
#define FPV_MUL_INEXACT 2
#define FPV_REDUCE 3
#define FPV_CANON 4
#define FPV_FROM_U64 5
#define FPV_TO_U64 6

static __device__ unsigned long long g_fp_audit_max[FP_AUDIT_SITES * FP_STAGES * FPM_COUNT];
static __device__ unsigned long long g_fp_audit_viol[FP_AUDIT_VIOLATIONS];
static __device__ unsigned long long g_fp_audit_calls[4]; // fp_mul, fp_reduce, values, conversions
// first violation: kind + 1, site, stage, three values as bits
static __device__ unsigned long long g_fp_audit_first[6];

#define FP_AUDIT_FC_FIELDS \
    int site;              \
    mutable int stage;
#define FP_AUDIT_INIT(c, s) \
    do {                    \
        (c).site = (s);     \
        (c).stage = 0;      \
    } while (0)
#define FP_STAGE(c, s) ((c).stage = (s))
#define FP_AUDIT_MUL(y, wx, wy, k, t, c) fp_audit_mul((y), (wx), (k), (t), (c).q, (c).site, (c).stage)
#define FP_AUDIT_REDUCE(x, r, c) fp_audit_reduce((x), (r), (c).q, (c).site, (c).stage)
#define FP_AUDIT_CANON(r, c) fp_audit_canon((r), (c).q, (c).site, (c).stage)
#define FP_AUDIT_VAL(c, metric, v) fp_audit_val((v), (c).q, (c).site, (c).stage, (metric))
#define FP_AUDIT_FROM_U64(v) fp_audit_from_u64((v))
#define FP_AUDIT_TO_U64(r) fp_audit_to_u64((r))

__device__ __noinline__ void fp_audit_violation(int kind, int site, int stage, double a, double b, double c)
{
    atomicAdd(&g_fp_audit_viol[kind], 1ull);
    if (atomicCAS(&g_fp_audit_first[0], 0ull, (unsigned long long) (kind + 1)) == 0ull) {
        g_fp_audit_first[1] = (unsigned long long) site;
        g_fp_audit_first[2] = (unsigned long long) stage;
        g_fp_audit_first[3] = (unsigned long long) __double_as_longlong(a);
        g_fp_audit_first[4] = (unsigned long long) __double_as_longlong(b);
        g_fp_audit_first[5] = (unsigned long long) __double_as_longlong(c);
    }
}
// an integer of magnitude below 2^53?
__device__ __forceinline__ bool fp_audit_exact_int(double v)
{
    return __builtin_fabs(v) < 9007199254740992.0 && __builtin_rint(v) == v;
}
__device__ __forceinline__ void fp_audit_check(double v, int site, int stage)
{
    if (!(__builtin_fabs(v) < 9007199254740992.0)) fp_audit_violation(FPV_RANGE, site, stage, v, 0.0, 0.0);
    else if (__builtin_rint(v) != v) fp_audit_violation(FPV_NONINTEGRAL, site, stage, v, 0.0, 0.0);
}
// non-negative doubles order as their bit patterns
__device__ __forceinline__ void fp_audit_max(int site, int stage, int metric, double ratio)
{
    const unsigned idx = ((unsigned) site * FP_STAGES + ((unsigned) stage & (FP_STAGES - 1))) * FPM_COUNT + (unsigned) metric;
    const unsigned long long b = (unsigned long long) __double_as_longlong(ratio);
    if (b > g_fp_audit_max[idx]) atomicMax(&g_fp_audit_max[idx], b);
}
__device__ __forceinline__ void fp_audit_count(int which)
{
    // (sampled: one lane in 64 adds 64 -- a counter for the report, not a check)
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_fp_audit_calls[which], 64ull);
}
__device__ __noinline__ void fp_audit_val(double v, double q, int site, int stage, int metric)
{
    fp_audit_count(2);
    fp_audit_check(v, site, stage);
    const double a = __builtin_fabs(v);
    fp_audit_max(site, stage, metric, a / q);
    fp_audit_max(site, stage, FPM_ABS, a * 0x1p-53);
}
__device__ __noinline__ void fp_audit_mul(double y, double wx, double k, double t, double q, int site, int stage)
{
    fp_audit_count(0);
    if (!fp_audit_exact_int(y) || !fp_audit_exact_int(wx) || !fp_audit_exact_int(k) || !fp_audit_exact_int(t)) {
        fp_audit_check(y, site, stage);
        fp_audit_check(wx, site, stage);
        fp_audit_check(k, site, stage);
        fp_audit_check(t, site, stage);
        return;
    }
    const __int128 exact = (__int128) (long long) y * (long long) wx - (__int128) (long long) k * (long long) q;
    if (exact != (__int128) (long long) t) fp_audit_violation(FPV_MUL_INEXACT, site, stage, y, wx, t);
    fp_audit_max(site, stage, FPM_MUL_Y, __builtin_fabs(y) / q);
    fp_audit_max(site, stage, FPM_MUL_W, __builtin_fabs(wx) / q);
    fp_audit_max(site, stage, FPM_MUL_T, __builtin_fabs(t) / q);
    fp_audit_max(site, stage, FPM_ABS, __builtin_fabs(t) * 0x1p-53);
    fp_audit_max(site, stage, FPM_ABS, __builtin_fabs(k) * 0x1p-53);
}
__device__ __noinline__ void fp_audit_reduce(double x, double r, double q, int site, int stage)
{
    fp_audit_count(1);
    if (!fp_audit_exact_int(x) || !fp_audit_exact_int(r)) {
        fp_audit_check(x, site, stage);
        fp_audit_check(r, site, stage);
        return;
    }
    const long long xi = (long long) x, ri = (long long) r, qi = (long long) q;
    const bool congruent = ((xi - ri) % qi) == 0;
    const bool centred = 2.0 * __builtin_fabs(r) <= q * (1.0 + 0x1p-40);
    if (!congruent || !centred) fp_audit_violation(FPV_REDUCE, site, stage, x, r, q);
    fp_audit_max(site, stage, FPM_RED_IN, __builtin_fabs(x) / q);
    fp_audit_max(site, stage, FPM_ABS, __builtin_fabs(x) * 0x1p-53);
}
__device__ __noinline__ void fp_audit_canon(double r, double q, int site, int stage)
{
    if (!(r >= 0.0 && r < q && __builtin_rint(r) == r)) fp_audit_violation(FPV_CANON, site, stage, r, q, 0.0);
}
__device__ __noinline__ void fp_audit_from_u64(unsigned long long v)
{
    fp_audit_count(3);
    if (v >> 52) fp_audit_violation(FPV_FROM_U64, 0, 0, (double) v, 0.0, 0.0);
}
__device__ __noinline__ void fp_audit_to_u64(double r)
{
    fp_audit_count(3);
    if (!(r >= 0.0 && r < 4503599627370496.0 && __builtin_rint(r) == r)) fp_audit_violation(FPV_TO_U64, 0, 0, r, 0.0, 0.0);
}

// host side: copy the tables out (after a device synchronisation), optionally clearing them
#ifdef HEGPU_FP_TU
#define FP_AUDIT_CAT2(a, b) a##b
#define FP_AUDIT_CAT(a, b) FP_AUDIT_CAT2(a, b)
extern "C" __attribute__((visibility("default"))) int FP_AUDIT_CAT(hegpu_fp_audit_read_, HEGPU_FP_TU)(
    unsigned long long* maxes, unsigned long long* viol, unsigned long long* calls, unsigned long long* first, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (maxes && hipMemcpyFromSymbol(maxes, HIP_SYMBOL(g_fp_audit_max), sizeof(g_fp_audit_max)) != hipSuccess) return -2;
    if (viol && hipMemcpyFromSymbol(viol, HIP_SYMBOL(g_fp_audit_viol), sizeof(g_fp_audit_viol)) != hipSuccess) return -3;
    if (calls && hipMemcpyFromSymbol(calls, HIP_SYMBOL(g_fp_audit_calls), sizeof(g_fp_audit_calls)) != hipSuccess) return -4;
    if (first && hipMemcpyFromSymbol(first, HIP_SYMBOL(g_fp_audit_first), sizeof(g_fp_audit_first)) != hipSuccess) return -5;
    if (reset) {
        static const unsigned long long zeros[FP_AUDIT_SITES * FP_STAGES * FPM_COUNT] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fp_audit_max), zeros, sizeof(g_fp_audit_max)) != hipSuccess) return -6;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fp_audit_viol), zeros, sizeof(g_fp_audit_viol)) != hipSuccess) return -7;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fp_audit_calls), zeros, sizeof(g_fp_audit_calls)) != hipSuccess) return -8;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fp_audit_first), zeros, sizeof(g_fp_audit_first)) != hipSuccess) return -9;
    }
    return FP_AUDIT_SITES * FP_STAGES * FPM_COUNT;
}
#endif

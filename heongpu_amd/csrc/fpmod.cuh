// fpmod.cuh -- exact modular arithmetic on integers held in FP64 (device only).
// Shared by the forward NTT (ntt.hip) and the TFHE blind rotate (tfhe.hip).
#pragma once
#include "modarith.cuh"

namespace hegpu {

// Moduli below 2^50 (Mod::fp set by the plan builder) run the forward
// transform in double precision: every value is an integer held exactly in a
// double (|v| < 2^53), and a modular product costs 6 full-rate FP64
// instructions instead of ~18 integer ones (9 of them 32-bit multiplies):
//   h = RN(y*w), l = y*w - h (exact, FMA), k = rint(RN(y*w'))  with w' = RN(w/q),
//   t = (h - k*q) + l.
// |k - y*w/q| <= 1/2 + |y|*2^-52, so |h - k*q| < 2^52 is an integer and the FMA
// that forms it is exact, as is the final sum: t == y*w - k*q EXACTLY,
// |t| <= q*(1/2 + |y|*2^-52).  A butterfly then adds/subtracts t; magnitudes
// grow from b*q to at most (1.25*b + 0.5)*q per stage (q < 2^50), i.e. from
// b <= 1 to b <= 5.4 over the four stages of one register round, always
// below 2^53; each round ends with a centred reduction x - q*rint(x/q)
// (|x| <= q/2 afterwards).  Everything is exact integer arithmetic modulo q,
// so the canonical result is bit-identical to the integer path's.
// ---- audit hooks.  The product defines none of them (all expand to nothing).  The instrumented build of the test suite
// (tests/audit/Makefile -> tests/audit/lib/libhegpu_audit.so) names tests/audit/fp_audit.cuh here: the SAME arithmetic below,
// with every fp_mul result compared against exact 128-bit integer arithmetic, every value checked for integrality and
// |v| < 2^53, and the largest |v| / q seen per (site, stage) kept in a device table (tests/test_gpu_fp_audit.py reads it and
// compares it with the bounds the comments claim; tests/fp_model.py derives those bounds and searches them on the host).
// The macro must name a file, so a stray -D cannot produce a library that builds and behaves differently.
// sites: kind * 8 + sub (sub = log2 N - 12 where the body is specialised by degree, else 0); stage = index of the
// transform stage (0 .. log2 N - 1), the product / the sums after it
enum FpSiteKind {
    FPS_NONE = 0, FPS_FWD_COL = 1, FPS_FWD_COL_DECOMP = 2, FPS_FWD_ROW = 3, FPS_FWD_SINGLE = 4, FPS_KS_ROW = 5,
    FPS_KS_ROW_SPLIT = 6, FPS_INV = 7, FPS_TFHE_PREP = 8, FPS_TFHE_BR = 9, FPS_KINDS = 10
};
enum FpMetric { FPM_MUL_Y = 0, FPM_MUL_W = 1, FPM_MUL_T = 2, FPM_SUM = 3, FPM_RED_IN = 4, FPM_ABS = 5, FPM_COUNT = 6 };
#define FP_SITE(kind, sub) ((kind) * 8 + (sub))
#define FP_STAGE_PRODUCT 24 // digit x key / key x digit product of the fused key switch and the external product
#define FP_STAGE_SUMS 25    // the running sums of those products (+ digits since the last re-centring: 25, 26, 27)
#define FP_STAGE_INPUT 29   // load transforms (decomposition, mod-down half)
#define FP_STAGE_OUT 30     // what leaves the transform / the external product
#define FP_STAGES 32

#ifdef HEGPU_FP_AUDIT_HEADER
#include HEGPU_FP_AUDIT_HEADER
#else
#define FP_AUDIT_FC_FIELDS
#define FP_AUDIT_INIT(c, site)
#define FP_STAGE(c, s)
#define FP_AUDIT_MUL(y, wx, wy, k, t, c)
#define FP_AUDIT_REDUCE(x, r, c)
#define FP_AUDIT_CANON(r, c)
#define FP_AUDIT_VAL(c, metric, v)
#define FP_AUDIT_FROM_U64(v)
#define FP_AUDIT_TO_U64(r)
#endif

struct FC {
    double q;  // modulus
    double qi; // RN(1/q)
    FP_AUDIT_FC_FIELDS
};
__device__ __forceinline__ FC make_fc(u64 q, int site = FPS_NONE)
{
    FC c;
    c.q = (double) q;
    c.qi = 1.0 / c.q;
    FP_AUDIT_INIT(c, site);
    (void) site;
    return c;
}
__device__ __forceinline__ double as_f64(u64 v) { return __longlong_as_double((long long) v); }
__device__ __forceinline__ u64 as_bits(double v) { return (u64) __double_as_longlong(v); }
// v < 2^52: OR the integer into the mantissa of 2^52 and subtract 2^52 (one
// integer and one FP64 instruction instead of two conversions and an FMA)
__device__ __forceinline__ double fp_from_u64(u64 v)
{
    FP_AUDIT_FROM_U64(v);
    return as_f64(v | 0x4330000000000000ull) - 4503599627370496.0;
}
__device__ __forceinline__ double fp_from_u32(u32 v) { return fp_from_u64((u64) v); }
// r an integer in [0, 2^52)
__device__ __forceinline__ u64 fp_to_u64(double r)
{
    FP_AUDIT_TO_U64(r);
    return as_bits(r + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;
}
// centred residue, |result| <= q/2 (1 + 2^-40); exact for |x| < 2^53
__device__ __forceinline__ double fp_reduce(double x, const FC& c)
{
    const double r = __builtin_fma(-__builtin_rint(x * c.qi), c.q, x);
    FP_AUDIT_REDUCE(x, r, c);
    return r;
}
// canonical residue in [0, q)
__device__ __forceinline__ double fp_canon(double x, const FC& c)
{
    const double r = fp_reduce(x, c);
    const double o = r < 0.0 ? r + c.q : r;
    FP_AUDIT_CANON(o, c);
    return o;
}
// y*w - k*q, see above; w = (w, RN(w/q)) as doubles
__device__ __forceinline__ double fp_mul(double y, double wx, double wy, const FC& c)
{
    const double h = y * wx;
    const double l = __builtin_fma(y, wx, -h);
    const double k = __builtin_rint(y * wy);
    const double t = __builtin_fma(-k, c.q, h) + l;
    FP_AUDIT_MUL(y, wx, wy, k, t, c);
    return t;
}

} // namespace hegpu

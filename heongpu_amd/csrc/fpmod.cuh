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
struct FC {
    double q;  // modulus
    double qi; // RN(1/q)
};
__device__ __forceinline__ FC make_fc(u64 q)
{
    FC c;
    c.q = (double) q;
    c.qi = 1.0 / c.q;
    return c;
}
__device__ __forceinline__ double as_f64(u64 v) { return __longlong_as_double((long long) v); }
__device__ __forceinline__ u64 as_bits(double v) { return (u64) __double_as_longlong(v); }
// v < 2^52: OR the integer into the mantissa of 2^52 and subtract 2^52 (one
// integer and one FP64 instruction instead of two conversions and an FMA)
__device__ __forceinline__ double fp_from_u64(u64 v)
{
    return as_f64(v | 0x4330000000000000ull) - 4503599627370496.0;
}
__device__ __forceinline__ double fp_from_u32(u32 v) { return fp_from_u64((u64) v); }
// r an integer in [0, 2^52)
__device__ __forceinline__ u64 fp_to_u64(double r) { return as_bits(r + 4503599627370496.0) & 0xFFFFFFFFFFFFFull; }
// centred residue, |result| <= q/2 (1 + 2^-40); exact for |x| < 2^53
__device__ __forceinline__ double fp_reduce(double x, const FC& c)
{
    return __builtin_fma(-__builtin_rint(x * c.qi), c.q, x);
}
// canonical residue in [0, q)
__device__ __forceinline__ double fp_canon(double x, const FC& c)
{
    const double r = fp_reduce(x, c);
    return r < 0.0 ? r + c.q : r;
}
// y*w - k*q, see above; w = (w, RN(w/q)) as doubles
__device__ __forceinline__ double fp_mul(double y, double wx, double wy, const FC& c)
{
    const double h = y * wx;
    const double l = __builtin_fma(y, wx, -h);
    const double k = __builtin_rint(y * wy);
    return __builtin_fma(-k, c.q, h) + l;
}

} // namespace hegpu

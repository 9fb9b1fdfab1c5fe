// modarith.cuh -- 64-bit modular arithmetic for gfx950 (device + host).
//
// Replaces GPU-NTT's OPERATOR_GPU_64::{mult,add,sub,reduce_forced} (unvendored
// submodule; call sites e.g. reference src/lib/kernel/multiplication.cu:119-123,
// switchkey.cu:24,54).  CDNA4 has no 64x64 multiplier: everything is built
// from v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32, so the routines below are
// written to minimise 32-bit multiplies:
//   mul_shoup      10 mul32 (constant operand with precomputed companion)
//   mul_barrett    11 mul32 (two variable operands)
//   reduce128      18 mul32 (lazy 128-bit accumulator -> canonical residue)
// All results are canonical residues, i.e. bit-identical to the reference's
// Barrett `mult` for in-range inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hegpu {

// moduli of at most this many bits run on the exact FP64 arithmetic of fpmod.cuh (Mod::fp, set in context.cpp)
#define FP_MAX_MODULUS_BITS 50


typedef unsigned long long u64;
typedef unsigned int u32;

// Per-modulus record uploaded to the device (superset of GPU-NTT's
// Modulus64{value,bit,mu}).
struct Mod {
    u64 q;      // modulus value
    u64 mu;     // floor(2^(2*bit+1) / q)  (GPU-NTT Barrett constant)
    u64 r_hi;   // floor(2^128 / q) high word
    u64 r_lo;   // floor(2^128 / q) low word
    u64 r64;    // floor(2^64 / q)
    u64 qinv;   // q^-1 mod 2^64 (odd q; 0 otherwise): Montgomery reduction of lazy 128-bit sums (redc128)
    u32 bit;    // floor(log2 q) + 1
    u32 fp;     // 1: the plan's FORWARD twiddle tables of this modulus hold FP64 pairs (ntt.hip)
};

#if defined(__HIPCC__)
#define HG_HD __host__ __device__ __forceinline__
#else
#define HG_HD inline
#endif

// 64x64 -> 128 as four 32x32+64 multiply-adds (4 v_mad_u64_u32 on gfx950)
HG_HD void mul64wide(u64 a, u64 b, u64& hi, u64& lo)
{
    u32 a0 = (u32) a, a1 = (u32) (a >> 32);
    u32 b0 = (u32) b, b1 = (u32) (b >> 32);
    u64 p00 = (u64) a0 * b0;
    u64 p01 = (u64) a0 * b1 + (p00 >> 32);
    u64 p10 = (u64) a1 * b0 + (u32) p01;
    u64 p11 = (u64) a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = (p10 << 32) | (u32) p00;
    hi = p11;
}

HG_HD u64 mulhi64(u64 a, u64 b)
{
    u64 hi, lo;
    mul64wide(a, b, hi, lo);
    return hi;
}

// a + b mod q, inputs canonical
HG_HD u64 add_mod(u64 a, u64 b, u64 q)
{
    u64 s = a + b;
    return (s >= q) ? s - q : s;
}

// reference semantics of OPERATOR_GPU_64::sub: (a + q - b), one conditional
// subtraction.  sub(q, 0) == q is kept.
HG_HD u64 sub_mod(u64 a, u64 b, u64 q)
{
    u64 d = a + q - b;
    return (d >= q) ? d - q : d;
}

// (a * b) mod q for a*b < 2^(2*bit): shift-Barrett, 4+4+3 mul32.
HG_HD u64 mul_barrett(u64 a, u64 b, const Mod& m)
{
    u64 hi, lo;
    mul64wide(a, b, hi, lo);
    u32 s1 = m.bit - 2;
    u64 w = (hi << (64 - s1)) | (lo >> s1);   // z >> (bit-2), < 2^(bit+2)
    // (w*mu) >> (bit+3)  ==  ((w << (61-bit)) * mu) >> 64
    u64 qh = mulhi64(w << (61 - m.bit), m.mu);
    u64 r = lo - qh * m.q;
    return (r >= m.q) ? r - m.q : r;
}

// x mod q for any 64-bit x (OPERATOR_GPU_64::reduce_forced).
HG_HD u64 reduce64(u64 x, const Mod& m)
{
    u64 qh = mulhi64(x, m.r64);
    u64 r = x - qh * m.q;
    return (r >= m.q) ? r - m.q : r;
}

// (hi:lo) mod q for any 128-bit value; q < 2^62.  floor(2^128/q) ratio.
HG_HD u64 reduce128(u64 hi, u64 lo, const Mod& m)
{
    // bits [128,192) of (hi:lo) * (r_hi:r_lo), exact
    u64 c = mulhi64(lo, m.r_lo);
    u64 a_lo, a_hi, b_lo, b_hi;
    mul64wide(lo, m.r_hi, a_hi, a_lo);
    mul64wide(hi, m.r_lo, b_hi, b_lo);
    u64 s = c + a_lo;
    u64 carry = (s < c);
    u64 s2 = s + b_lo;
    carry += (s2 < s);
    u64 qh = hi * m.r_hi + a_hi + b_hi + carry;
    u64 r = lo - qh * m.q;
    return (r >= m.q) ? r - m.q : r;
}

// (hi:lo) * 2^-64 mod q for ANY 128-bit (hi:lo), odd q < 2^62: Montgomery reduction -- m = lo * q^-1 mod 2^64, then
// (hi:lo - m q) / 2^64 = hi - mulhi(m, q) exactly (the low words cancel), a value in (-q, hi]; made non-negative by one
// conditional +q (the wrapped difference plus q is exact because the true value lies in (-q, 0)) and reduced.
// 14 32-bit multiplies against the 18 of reduce128, and half its additions.  For sums whose constant factors carry
// the compensating 2^64 (the BFV base-conversion tables).  Round 3's form added q unconditionally and so needed
// hi + q < 2^64 -- false for sums of ~56 or more 122-bit terms; the callers' remaining contract is that the
// lazy sum itself fits 128 bits, which Context::build_host checks for every table it builds (lazy_sum_fits).
HG_HD u64 redc128(u64 hi, u64 lo, const Mod& m)
{
    const u64 k = lo * m.qinv;
    const u64 t = mulhi64(k, m.q); // < q
    const u64 d = hi - t;
    const u64 r = (hi < t) ? d + m.q : d; // in [0, hi]
    return reduce64(r, m);
}

// Shoup/Harvey multiply by a constant w with companion wp = floor(w*2^64/q):
// returns w*y - floor(wp*y/2^64)*q in [0, 2q) for ANY 64-bit y.
HG_HD u64 mul_shoup_lazy(u64 y, u64 w, u64 wp, u64 q)
{
    u64 qh = mulhi64(y, wp);
    return y * w - qh * q;
}

HG_HD u64 mul_shoup(u64 y, u64 w, u64 wp, u64 q)
{
    u64 r = mul_shoup_lazy(y, w, wp, q);
    return (r >= q) ? r - q : r;
}

// host-side constructors
inline Mod make_mod(u64 q)
{
    Mod m;
    m.q = q;
    m.bit = 64 - (u32) __builtin_clzll(q);
    m.mu = (u64) ((((unsigned __int128) 1) << (2 * m.bit + 1)) / q);
    unsigned __int128 ones = ~((unsigned __int128) 0);
    // floor(2^128/q) == floor((2^128-1)/q) unless q | 2^128 (q = 2^k)
    unsigned __int128 r = ones / q;
    if ((q & (q - 1)) == 0) r += 1;
    m.r_hi = (u64) (r >> 64);
    m.r_lo = (u64) r;
    m.r64 = (q == 1) ? 0 : (u64) ((((unsigned __int128) 1) << 64) / q);
    m.qinv = 0;
    if (q & 1) { // Newton: x <- x (2 - q x) doubles the number of correct low bits
        u64 x = q; // correct to 3 bits for odd q
        for (int i = 0; i < 6; i++) x *= 2 - q * x;
        m.qinv = x;
    }
    m.fp = 0;
    return m;
}
inline u64 shoup_companion(u64 w, u64 q)
{
    return (u64) ((((unsigned __int128) w) << 64) / q);
}

} // namespace hegpu

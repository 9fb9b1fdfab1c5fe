// rns.hip -- RNS element-wise kernels for gfx950 (see rns.hpp for the
// reference kernels each one replaces).
//
// All of these are streaming kernels bounded by HBM: every thread owns two
// adjacent coefficients and moves them with 16-byte loads/stores (dwordx4),
// a workgroup covers 512 contiguous coefficients of one limb, and the grid is
// (N/512, limb-ish, batch-ish) so that thousands of workgroups are in flight.
// Arithmetic that the reference does with repeated canonical add/mult is done
// lazily where it is exact (128-bit accumulation in the key-switch MAC), so
// outputs are the same canonical residues.
#include "rns.hpp"
#include <cstdlib>

namespace hegpu {

#define RNS_THREADS 256
#define RNS_PER_BLOCK 512

__device__ __forceinline__ ulonglong2 ld2(const u64* p) { return *reinterpret_cast<const ulonglong2*>(p); }
__device__ __forceinline__ void st2(u64* p, ulonglong2 v) { *reinterpret_cast<ulonglong2*>(p) = v; }
__device__ __forceinline__ u64 coeff0() { return ((u64) blockIdx.x * RNS_THREADS + threadIdx.x) * 2; }

static inline dim3 grid3(int n_power, int y, int z) { return dim3((1u << n_power) / RNS_PER_BLOCK, y, z); }

// ---------------------------------------------------------------- add/sub/neg
template <int OP>
__global__ __launch_bounds__(RNS_THREADS) void k_addition(const u64* __restrict__ a, const u64* __restrict__ b,
                                                          u64* __restrict__ out, const Mod* __restrict__ mods,
                                                          int n_power, int limbs)
{
    const u64 q = mods[blockIdx.y].q;
    // blockIdx.z runs over batch*parts; all of them have `limbs` limbs
    const u64 loc = coeff0() + ((u64) blockIdx.y << n_power) + (((u64) limbs * blockIdx.z) << n_power);
    ulonglong2 r;
    if (OP == 0) {
        ulonglong2 x = ld2(a + loc), y = ld2(b + loc);
        r.x = add_mod(x.x, y.x, q);
        r.y = add_mod(x.y, y.y, q);
    } else if (OP == 1) {
        ulonglong2 x = ld2(a + loc), y = ld2(b + loc);
        r.x = sub_mod(x.x, y.x, q);
        r.y = sub_mod(x.y, y.y, q);
    } else {
        ulonglong2 x = ld2(a + loc);
        r.x = sub_mod(0, x.x, q);
        r.y = sub_mod(0, x.y, q);
    }
    st2(out + loc, r);
}

hipError_t rns_addition(const u64* a, const u64* b, u64* out, const Mod* mods, int n_power, int limbs,
                        int parts, int batch, int op, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    dim3 g = grid3(n_power, limbs, parts * batch);
    if (op == 0) hipLaunchKernelGGL(k_addition<0>, g, dim3(RNS_THREADS), 0, st, a, b, out, mods, n_power, limbs);
    else if (op == 1) hipLaunchKernelGGL(k_addition<1>, g, dim3(RNS_THREADS), 0, st, a, b, out, mods, n_power, limbs);
    else hipLaunchKernelGGL(k_addition<2>, g, dim3(RNS_THREADS), 0, st, a, b, out, mods, n_power, limbs);
    return hipGetLastError();
}

__global__ __launch_bounds__(RNS_THREADS) void k_addition_strided(const u64* a, u64 sa, const u64* b, u64 sb,
                                                                  u64* out, u64 so, const Mod* __restrict__ mods,
                                                                  int n_power, int limbs, int parts)
{
    const u64 q = mods[blockIdx.y].q;
    const int z = blockIdx.z % parts, item = blockIdx.z / parts;
    const u64 loc = coeff0() + ((u64) blockIdx.y << n_power) + (((u64) limbs * z) << n_power);
    ulonglong2 x = ld2(a + sa * item + loc), y = ld2(b + sb * item + loc), r;
    r.x = add_mod(x.x, y.x, q);
    r.y = add_mod(x.y, y.y, q);
    st2(out + so * item + loc, r);
}

hipError_t rns_addition_strided(const u64* a, u64 sa, const u64* b, u64 sb, u64* out, u64 so, const Mod* mods,
                                int n_power, int limbs, int parts, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_addition_strided, grid3(n_power, limbs, parts * batch), dim3(RNS_THREADS), 0, st, a, sa, b,
                       sb, out, so, mods, n_power, limbs, parts);
    return hipGetLastError();
}

// ---------------------------------------------------------------- tensor product
__global__ __launch_bounds__(RNS_THREADS) void k_cross_multiplication(const u64* __restrict__ in1, u64 s1,
                                                                      const u64* __restrict__ in2, u64 s2,
                                                                      u64* __restrict__ out, u64 so,
                                                                      const Mod* __restrict__ mods, int n_power,
                                                                      int decomp_size)
{
    const Mod m = mods[blockIdx.y];
    const u64 loc = coeff0() + ((u64) blockIdx.y << n_power);
    const u64 part = (u64) decomp_size << n_power;
    const u64* p1 = in1 + s1 * blockIdx.z;
    const u64* p2 = in2 + s2 * blockIdx.z;
    u64* po = out + so * blockIdx.z;
    ulonglong2 a0 = ld2(p1 + loc), a1 = ld2(p1 + loc + part);
    ulonglong2 b0 = ld2(p2 + loc), b1 = ld2(p2 + loc + part);
    ulonglong2 o0, o1, o2;
    o0.x = mul_barrett(a0.x, b0.x, m);
    o0.y = mul_barrett(a0.y, b0.y, m);
    o2.x = mul_barrett(a1.x, b1.x, m);
    o2.y = mul_barrett(a1.y, b1.y, m);
    // a0*b1 + a1*b0: one 128-bit sum, one reduction (same canonical value)
    {
        u64 h1, l1, h2, l2;
        mul64wide(a0.x, b1.x, h1, l1);
        mul64wide(a1.x, b0.x, h2, l2);
        u64 lo = l1 + l2;
        u64 hi = h1 + h2 + (lo < l1);
        o1.x = reduce128(hi, lo, m);
        mul64wide(a0.y, b1.y, h1, l1);
        mul64wide(a1.y, b0.y, h2, l2);
        lo = l1 + l2;
        hi = h1 + h2 + (lo < l1);
        o1.y = reduce128(hi, lo, m);
    }
    st2(po + loc, o0);
    st2(po + loc + part, o1);
    st2(po + loc + 2 * part, o2);
}

hipError_t rns_cross_multiplication(const u64* in1, u64 s1, const u64* in2, u64 s2, u64* out, u64 so,
                                    const Mod* mods, int n_power, int decomp_size, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_cross_multiplication, grid3(n_power, decomp_size, batch), dim3(RNS_THREADS), 0, st, in1,
                       s1, in2, s2, out, so, mods, n_power, decomp_size);
    return hipGetLastError();
}

// ---------------------------------------------------------------- digit decomposition
__global__ __launch_bounds__(RNS_THREADS) void k_decompose(const u64* __restrict__ in, u64 in_stride,
                                                           u64* __restrict__ out, u64 out_stride,
                                                           const Mod* __restrict__ mods, int n_power, int nmods,
                                                           int split, int level)
{
    const u64 c = coeff0();
    const ulonglong2 x = ld2(in + in_stride * blockIdx.z + ((u64) blockIdx.y << n_power) + c);
    u64* po = out + out_stride * blockIdx.z + (((u64) nmods * blockIdx.y) << n_power) + c;
    for (int i = 0; i < nmods; i++) {
        const Mod m = mods[(i < split) ? i : i + level];
        ulonglong2 r;
        r.x = reduce64(x.x, m);
        r.y = reduce64(x.y, m);
        st2(po + ((u64) i << n_power), r);
    }
}

hipError_t rns_decompose(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods, int n_power,
                         int digits, int nmods, int split, int level, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_decompose, grid3(n_power, digits, batch), dim3(RNS_THREADS), 0, st, in, in_stride, out,
                       out_stride, mods, n_power, nmods, split, level);
    return hipGetLastError();
}

// ---------------------------------------------------------------- partial sums of a digit-split key switch
__global__ __launch_bounds__(RNS_THREADS) void k_sum_partials(const u64* __restrict__ buf, u64 buf_item_stride,
                                                              u64* __restrict__ out, u64 out_item_stride,
                                                              const Mod* __restrict__ mods, const int* __restrict__ mod_order,
                                                              int n_power, int digits, int rc, int splits)
{
    const u64 c = coeff0();
    const int part = blockIdx.y / rc, slot = blockIdx.y - part * rc;
    const u64 q = mods[mod_order ? mod_order[slot] : slot].q;
    const u64* pb = buf + buf_item_stride * blockIdx.z + ((u64) slot << n_power) + c;
    ulonglong2 v[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int d = ((s < splits ? s : 0) * digits) / splits + part;
        v[s] = ld2(pb + (((u64) d * rc) << n_power));
    }
    ulonglong2 r = v[0];
#pragma unroll
    for (int s = 1; s < 8; s++)
        if (s < splits) {
            r.x = add_mod(r.x, v[s].x, q);
            r.y = add_mod(r.y, v[s].y, q);
        }
    st2(out + out_item_stride * blockIdx.z + ((u64) (part * rc + slot) << n_power) + c, r);
}

hipError_t rns_sum_partials(const u64* buf, u64 buf_item_stride, u64* out, u64 out_item_stride, const Mod* mods,
                            const int* mod_order, int n_power, int digits, int rc, int splits, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (splits < 2 || splits > 8 || digits < 2 * splits) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sum_partials, grid3(n_power, 2 * rc, batch), dim3(RNS_THREADS), 0, st, buf, buf_item_stride, out,
                       out_item_stride, mods, mod_order, n_power, digits, rc, splits);
    return hipGetLastError();
}

// ---------------------------------------------------------------- key-switch inner product
__device__ __forceinline__ void acc_mad(u64& hi, u64& lo, u64 a, u64 b)
{
    u64 h, l;
    mul64wide(a, b, h, l);
    lo += l;
    hi += h + (lo < l);
}

// sum_j v[j] * row[j < valid ? j : 0] as a 128-bit integer, every operand below 2^61 (residues and table
// entries of moduli of at most 61 bits; v[j] == 0 beyond `valid`).  A 128-bit multiply-accumulate per term
// costs ~19 instructions, most of them carries and the shifted addends of the 4-multiply chain.  Here the four
// partial products of eight terms at a time are summed by weight instead -- a0*b1, a1*b0 < 2^61 and
// a1*b1 < 2^58 cannot overflow 64 bits in eight terms, so each is one v_mad_u64_u32 with a 64-bit addend; only
// a0*b0 needs a carry count -- and the columns are put together once per eight terms: 6 instructions per term.
// One term in five instructions: every partial product accumulates straight into its 64-bit column (v_mad_u64_u32
// takes the column as its addend), a0*b0 hands its carry to the count c00.  hipcc's own code for the same statements
// spends ~15 (a 64-bit add, a 64-bit compare and a select for the carry, moves that rebuild operands in VGPRs); the
// base-conversion kernels built on this are bound by nothing but their instruction count (vector ALU > 100 % busy by
// the 4-cycles-per-instruction measure).  b0 / b1: wave-uniform table entries (scalar registers).
__device__ __forceinline__ void col_mad_uniform(u64& s00, u64& s01, u64& s10, u64& s11, u32& c00, u32 a0, u32 a1, u32 b0,
                                                u32 b1)
{
    asm("v_mad_u64_u32 %0, vcc, %5, %7, %0\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %6, %8, %3"
        : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11), "+v"(c00)
        : "v"(a0), "v"(a1), "s"(b0), "s"(b1)
        : "vcc");
}
template <int M>
__device__ __forceinline__ void dot128(const u64 (&v)[M], const u64* __restrict__ row, int valid, u64& hi, u64& lo)
{
    hi = lo = 0;
#pragma unroll
    for (int j0 = 0; j0 < M; j0 += 8) {
        u64 s00 = 0, s01 = 0, s10 = 0, s11 = 0;
        u32 c00 = 0;
#pragma unroll
        for (int j = j0; j < j0 + 8 && j < M; j++) {
            const u64 b = row[j < valid ? j : 0];
            col_mad_uniform(s00, s01, s10, s11, c00, (u32) v[j], (u32) (v[j] >> 32), (u32) b, (u32) (b >> 32));
        }
        const u64 mid = s01 + s10;
        const u64 cm = mid < s01;
        const u64 l = s00 + (mid << 32);
        const u64 h = s11 + (mid >> 32) + (cm << 32) + c00 + (l < s00);
        lo += l;
        hi += h + (lo < l);
    }
}

// Workgroups are ordered batch-fastest (blockIdx.x = ciphertext): the `batch`
// workgroups that read the same key tile (same limb, same 512 coefficients)
// are dispatched back to back, so the evaluation key streams from HBM once
// per tile instead of once per ciphertext (it is reused out of L2 / MALL).
template <int ITEMS>
__global__ __launch_bounds__(RNS_THREADS) void k_keyswitch_mac(const u64* __restrict__ in, u64 in_stride,
                                                               const u64* __restrict__ key,
                                                               u64* __restrict__ out, u64 out_stride,
                                                               const Mod* __restrict__ mods, int n_power,
                                                               int digits, int nmods, int key_limbs, int split,
                                                               int level)
{
    const int item0 = blockIdx.x * ITEMS;
    const int y = blockIdx.z;
    const int kidx = (y < split) ? y : y + level;
    const Mod m = mods[kidx];
    const u64 c = ((u64) blockIdx.y * RNS_THREADS + threadIdx.x) * 2;
    const u64* pin = in + in_stride * item0 + ((u64) y << n_power) + c;
    const u64* pk = key + ((u64) kidx << n_power) + c;
    const u64 key_off1 = (u64) key_limbs << n_power;
    const u64 key_off2 = (u64) key_limbs << (n_power + 1);
    const u64 dig_off = (u64) nmods << n_power;
    u64 h[ITEMS][4], l[ITEMS][4];
#pragma unroll
    for (int t = 0; t < ITEMS; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) h[t][e] = l[t][e] = 0;
#pragma unroll 2
    for (int i = 0; i < digits; i++) {
        const ulonglong2 k0 = ld2(pk + key_off2 * i);
        const ulonglong2 k1 = ld2(pk + key_off2 * i + key_off1);
#pragma unroll
        for (int t = 0; t < ITEMS; t++) {
            const ulonglong2 d = ld2(pin + in_stride * t + dig_off * i);
            acc_mad(h[t][0], l[t][0], d.x, k0.x);
            acc_mad(h[t][1], l[t][1], d.y, k0.y);
            acc_mad(h[t][2], l[t][2], d.x, k1.x);
            acc_mad(h[t][3], l[t][3], d.y, k1.y);
        }
    }
#pragma unroll
    for (int t = 0; t < ITEMS; t++) {
        ulonglong2 r0, r1;
        r0.x = reduce128(h[t][0], l[t][0], m);
        r0.y = reduce128(h[t][1], l[t][1], m);
        r1.x = reduce128(h[t][2], l[t][2], m);
        r1.y = reduce128(h[t][3], l[t][3], m);
        u64* po = out + out_stride * (item0 + t) + ((u64) y << n_power) + c;
        st2(po, r0);
        st2(po + dig_off, r1);
    }
}

// The same inner product for up to KSM_KEYS evaluation keys at once (hoisted rotations: the NTT-domain digits of a
// ciphertext are shared by all Galois elements): every digit value is loaded once and multiplied into the
// accumulators of every key, so the digits -- 272 limbs per ciphertext at C4, the largest stream of the per-element
// part -- are read once per group of keys instead of once per key.  Result of key e at out + e * out_key_stride.
#define KSM_KEYS 4
struct KsmKeys { const u64* k[KSM_KEYS]; };
// One coefficient per thread.  The products are summed by partial-product weight, eight digits at a time (see
// dot128: operands below 2^61, only a0*b0 needs a carry count) -- 6 instead of ~19 instructions per term, which is
// what makes four keys per digit load affordable.
struct ColSum {
    u64 s00, s01, s10, s11;
    u32 c00;
};
__device__ __forceinline__ void col_clear(ColSum& s) { s.s00 = s.s01 = s.s10 = s.s11 = 0; s.c00 = 0; }
__device__ __forceinline__ void col_mad(ColSum& s, u64 a, u64 b)
{
    const u32 a0 = (u32) a, a1 = (u32) (a >> 32), b0 = (u32) b, b1 = (u32) (b >> 32);
    const u64 p = (u64) a0 * b0;
    s.s00 += p;
    s.c00 += s.s00 < p;
    s.s01 += (u64) a0 * b1;
    s.s10 += (u64) a1 * b0;
    s.s11 += (u64) a1 * b1;
}
// the same term in five instructions (col_mad_uniform with both operands in vector registers).  On its own it made
// k_keyswitch_mac_keys slower -- the compiler stopped requesting the key values of a sum ahead of the multiplies (99
// registers instead of 162) -- so that kernel requests them itself, one (key, part) ahead.
__device__ __forceinline__ void col_mad_v(ColSum& s, u64 a, u64 b)
{
    asm("v_mad_u64_u32 %0, vcc, %5, %7, %0\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %6, %8, %3"
        : "+v"(s.s00), "+v"(s.s01), "+v"(s.s10), "+v"(s.s11), "+v"(s.c00)
        : "v"((u32) a), "v"((u32) (a >> 32)), "v"((u32) b), "v"((u32) (b >> 32))
        : "vcc");
}
__device__ __forceinline__ void col_fold(const ColSum& s, u64& hi, u64& lo)
{
    const u64 mid = s.s01 + s.s10;
    const u64 cm = mid < s.s01;
    const u64 l = s.s00 + (mid << 32);
    const u64 h = s.s11 + (mid >> 32) + (cm << 32) + s.c00 + (l < s.s00);
    lo += l;
    hi += h + (lo < l);
}
// Key-stationary: a workgroup owns the key tiles of E keys for one limb and 256 / E coefficients -- all `digits`
// digits, both parts: digits x 4 KiB of LDS whatever E is -- and walks the ciphertexts (E at a time, one per
// group of 256 / E lanes).  Per ciphertext and coefficient a thread requests its `digits` digit values at once
// (independent loads: nothing else keeps this kernel from running at memory speed at two waves per SIMD) and
// multiplies them into the accumulators of the E keys out of LDS.
template <int E>
__global__ __launch_bounds__(RNS_THREADS) void k_keyswitch_mac_keys(const u64* __restrict__ in, u64 in_stride, KsmKeys keys,
                                                                    u64* __restrict__ out, u64 out_stride,
                                                                    u64 out_key_stride, const Mod* __restrict__ mods,
                                                                    int n_power, int digits, int nmods, int key_limbs,
                                                                    int split, int level, int items, int items_per_wg)
{
    constexpr int TILE = RNS_THREADS / E;
    extern __shared__ __attribute__((aligned(16))) u64 kl[]; // [E][2][digits][TILE]
    const int t = threadIdx.x, coef = t % TILE, lane = t / TILE;
    const int y = blockIdx.y;
    const int kidx = (y < split) ? y : y + level;
    const Mod m = mods[kidx];
    const u64 c0 = (u64) blockIdx.x * TILE;
    const u64 key_off1 = (u64) key_limbs << n_power;
    const u64 key_off2 = (u64) key_limbs << (n_power + 1);
    const u64 dig_off = (u64) nmods << n_power;
    // fill: (key, part, digit) rows of TILE coefficients; eight rows' worth of loads in flight per thread (a
    // load -> LDS store chain per element would cost one memory latency each)
    {
        const int rows = E * 2 * digits;              // a multiple of 2 E
        const int row0 = t / TILE, cf = t % TILE;     // this thread's first row; it takes rows row0 + k * E
        for (int r0 = row0; r0 < rows; r0 += 8 * E) {
            u64 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = (r0 + k * E < rows) ? r0 + k * E : row0;
                const int i = r % digits, ep = r / digits;
                v[k] = keys.k[ep >> 1][((u64) kidx << n_power) + c0 + cf + key_off2 * i + key_off1 * (ep & 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (r0 + k * E < rows) kl[(size_t) (r0 + k * E) * TILE + cf] = v[k];
        }
    }
    __syncthreads();
    const int item_end = min(items, (int) (blockIdx.z + 1) * items_per_wg);
    const int item_first = blockIdx.z * items_per_wg + lane;
    u64 dn[16]; // the digits of the NEXT ciphertext of this lane are requested before the products of this one
    auto fetch = [&](int item) {
        const u64* pin = in + in_stride * (item < item_end ? item : item_first) + ((u64) y << n_power) + c0 + coef;
#pragma unroll
        for (int i = 0; i < 16; i++) dn[i] = pin[dig_off * (i < digits ? i : digits - 1)];
    };
    if (item_first < item_end) fetch(item_first);
    for (int item = item_first; item < item_end; item += E) {
        u64 d[16];
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = (i < digits) ? dn[i] : 0;
        fetch(item + E);
        // (run-time loops over key and part: unrolled, the compiler requests all 2 E x 16 key values from LDS up
        // front -- 256 registers, one wave per SIMD and scratch)
        // the 16 key values of (key e, part) come out of LDS one (key, part) ahead of their products: part 0 in ka,
        // part 1 in kb
        auto keys_of = [&](u64 (&kv)[16], int ep) {
            const u64* kp = kl + (size_t) (ep * digits) * TILE + coef;
#pragma unroll
            for (int i = 0; i < 16; i++) kv[i] = kp[(size_t) (i < digits ? i : 0) * TILE];
        };
        auto sum_out = [&](const u64 (&kv)[16], int e, int part) {
            u64 hi = 0, lo = 0;
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += 8) {
                ColSum cs;
                col_clear(cs);
#pragma unroll
                for (int ii = 0; ii < 8; ii++) col_mad_v(cs, d[i0 + ii], kv[i0 + ii]);
                col_fold(cs, hi, lo);
            }
            out[out_key_stride * e + out_stride * item + ((u64) y << n_power) + c0 + coef + dig_off * part] =
                reduce128(hi, lo, m);
        };
        u64 ka[16], kb[16];
        keys_of(ka, 0);
#pragma unroll 1
        for (int e = 0; e < E; e++) {
            keys_of(kb, 2 * e + 1);
            sum_out(ka, e, 0);
            keys_of(ka, (e + 1 < E) ? 2 * e + 2 : 0);
            sum_out(kb, e, 1);
        }
    }
}

hipError_t rns_keyswitch_mac_keys(const u64* in, u64 in_stride, const u64* const* keys, int key_count, u64* out,
                                  u64 out_stride, u64 out_key_stride, const Mod* mods, int n_power, int digits, int nmods,
                                  int key_limbs, int split, int level, int batch, hipStream_t st)
{
    if (batch <= 0 || key_count <= 0) return hipSuccess;
    if (digits > 16 || key_count > KSM_KEYS) return hipErrorInvalidValue; // digits x 4 KiB of LDS, 16 digit registers
    KsmKeys kk;
    for (int e = 0; e < KSM_KEYS; e++) kk.k[e] = keys[e < key_count ? e : 0];
    const int E = key_count >= 4 ? 4 : key_count >= 2 ? 2 : 1; // three keys: a pair, then a single one
    // enough workgroups to fill the chip: the ciphertexts of a (tile, limb) are split over blockIdx.z if needed
    const unsigned tiles = (1u << n_power) / (RNS_THREADS / E);
    int zsplit = 1;
    while (zsplit * 2 * E <= batch && (unsigned long) tiles * nmods * zsplit < 2048) zsplit *= 2;
    const int per_wg = (batch + zsplit - 1) / zsplit;
    const size_t lds = (size_t) digits * 4096;
#define LAUNCH(EE, KOFF)                                                                                                   \
    do {                                                                                                                   \
        KsmKeys k2 = kk;                                                                                                   \
        for (int e = 0; e < KSM_KEYS; e++) k2.k[e] = kk.k[(KOFF) + e < KSM_KEYS ? (KOFF) + e : KSM_KEYS - 1];              \
        hipLaunchKernelGGL(k_keyswitch_mac_keys<EE>, dim3((1u << n_power) / (RNS_THREADS / (EE)), nmods, (batch + per_wg - 1) / per_wg), \
                           dim3(RNS_THREADS), lds, st, in, in_stride, k2, out + (u64) (KOFF) * out_key_stride, out_stride,     \
                           out_key_stride, mods, n_power, digits, nmods, key_limbs, split, level, batch, per_wg);          \
    } while (0)
    switch (key_count) {
        case 1: LAUNCH(1, 0); break;
        case 2: LAUNCH(2, 0); break;
        case 3: LAUNCH(2, 0); LAUNCH(1, 2); break;
        default: LAUNCH(4, 0); break;
    }
#undef LAUNCH
    return hipGetLastError();
}

hipError_t rns_keyswitch_mac(const u64* in, u64 in_stride, const u64* key, u64* out, u64 out_stride,
                             const Mod* mods, int n_power, int digits, int nmods, int key_limbs, int split,
                             int level, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    if (digits > 64) return hipErrorInvalidValue; // 128-bit accumulator bound
    // up to four ciphertexts per workgroup share every key load when the batch allows it
    if (batch % 4 == 0) {
        dim3 g(batch / 4, (1u << n_power) / RNS_PER_BLOCK, nmods);
        hipLaunchKernelGGL(k_keyswitch_mac<4>, g, dim3(RNS_THREADS), 0, st, in, in_stride, key, out, out_stride,
                           mods, n_power, digits, nmods, key_limbs, split, level);
    } else if (batch % 2 == 0) {
        dim3 g(batch / 2, (1u << n_power) / RNS_PER_BLOCK, nmods);
        hipLaunchKernelGGL(k_keyswitch_mac<2>, g, dim3(RNS_THREADS), 0, st, in, in_stride, key, out, out_stride,
                           mods, n_power, digits, nmods, key_limbs, split, level);
    } else {
        dim3 g(batch, (1u << n_power) / RNS_PER_BLOCK, nmods);
        hipLaunchKernelGGL(k_keyswitch_mac<1>, g, dim3(RNS_THREADS), 0, st, in, in_stride, key, out, out_stride,
                           mods, n_power, digits, nmods, key_limbs, split, level);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- mod-down (BFV, single kernel)
__global__ __launch_bounds__(RNS_THREADS) void k_divide_round_lastq(
    const u64* __restrict__ in, u64 in_stride, const u64* ct, u64 ct_stride, u64* out, u64 out_stride,
    const Mod* __restrict__ mods, const u64* __restrict__ half, const u64* __restrict__ half_mod,
    const u64* __restrict__ last_q_modinv, int n_power, int D, int switchkey)
{
    const int y = blockIdx.y;
    const int z = blockIdx.z & 1, b = blockIdx.z >> 1;
    const Mod m = mods[y];
    const u64 qP = mods[D].q;
    const u64 c = coeff0();
    const u64* pin = in + in_stride * b + (((u64) (D + 1) << n_power) * z) + c;
    ulonglong2 last = ld2(pin + ((u64) D << n_power));
    ulonglong2 x = ld2(pin + ((u64) y << n_power));
    const u64 h = half[0], hm = half_mod[y], inv = last_q_modinv[y];
    const u64 loc = (((u64) D << n_power) * z) + ((u64) y << n_power) + c;
    ulonglong2 cin = make_ulonglong2(0, 0);
    if (!(switchkey && z != 0)) cin = ld2(ct + ct_stride * b + loc);
    ulonglong2 r;
    {
        u64 l = add_mod(last.x, h, qP);
        l = reduce64(l, m);
        l = sub_mod(l, hm, m.q);
        u64 v = sub_mod(x.x, l, m.q);
        v = mul_barrett(v, inv, m);
        r.x = add_mod(cin.x, v, m.q);
    }
    {
        u64 l = add_mod(last.y, h, qP);
        l = reduce64(l, m);
        l = sub_mod(l, hm, m.q);
        u64 v = sub_mod(x.y, l, m.q);
        v = mul_barrett(v, inv, m);
        r.y = add_mod(cin.y, v, m.q);
    }
    st2(out + out_stride * b + loc, r);
}

hipError_t rns_divide_round_lastq(const u64* in, u64 in_stride, const u64* ct, u64 ct_stride, u64* out,
                                  u64 out_stride, const Mod* mods, const u64* half, const u64* half_mod,
                                  const u64* last_q_modinv, int n_power, int decomp, int switchkey, int batch,
                                  hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_divide_round_lastq, grid3(n_power, decomp, 2 * batch), dim3(RNS_THREADS), 0, st, in,
                       in_stride, ct, ct_stride, out, out_stride, mods, half, half_mod, last_q_modinv, n_power,
                       decomp, switchkey);
    return hipGetLastError();
}

// ---------------------------------------------------------------- mod-down stage one (CKKS / rescale)
__global__ __launch_bounds__(RNS_THREADS) void k_moddown_stage_one(
    const u64* __restrict__ in, u64 in_stride, u64* __restrict__ out, u64 out_stride,
    const Mod* __restrict__ mods, const u64* __restrict__ half, const u64* __restrict__ half_mod, int n_power,
    int first_decomp, int C)
{
    const int y = blockIdx.y; // cipher part
    const u64 c = coeff0();
    const u64 qP = mods[first_decomp].q;
    ulonglong2 last =
        ld2(in + in_stride * blockIdx.z + ((u64) C << n_power) + (((u64) (C + 1) << n_power) * y) + c);
    const u64 h = half[0];
    last.x = add_mod(last.x, h, qP);
    last.y = add_mod(last.y, h, qP);
    u64* po = out + out_stride * blockIdx.z + (((u64) C << n_power) * y) + c;
    for (int i = 0; i < C; i++) {
        const Mod m = mods[i];
        const u64 hm = half_mod[i];
        ulonglong2 r;
        r.x = sub_mod(reduce64(last.x, m), hm, m.q);
        r.y = sub_mod(reduce64(last.y, m), hm, m.q);
        st2(po + ((u64) i << n_power), r);
    }
}

hipError_t rns_moddown_stage_one(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                                 const u64* half, const u64* half_mod, int n_power, int first_decomp,
                                 int cur_decomp, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_moddown_stage_one, grid3(n_power, 2, batch), dim3(RNS_THREADS), 0, st, in, in_stride, out,
                       out_stride, mods, half, half_mod, n_power, first_decomp, cur_decomp);
    return hipGetLastError();
}

// ---------------------------------------------------------------- mod-down stage two / rescale
__global__ __launch_bounds__(RNS_THREADS) void k_moddown_stage_two(
    const u64* __restrict__ in_last, u64 last_stride, const u64* in, u64 in_stride, int in_limbs, const u64* ct,
    u64 ct_stride, u64* out, u64 out_stride, const Mod* __restrict__ mods,
    const u64* __restrict__ last_q_modinv, int n_power, int C, int with_ct)
{
    const int y = blockIdx.y;
    const int z = blockIdx.z & 1, b = blockIdx.z >> 1;
    const Mod m = mods[y];
    const u64 c = coeff0();
    const u64 loc = ((u64) y << n_power) + (((u64) C << n_power) * z) + c;
    ulonglong2 last = ld2(in_last + last_stride * b + loc);
    ulonglong2 x = ld2(in + in_stride * b + ((u64) y << n_power) + (((u64) in_limbs << n_power) * z) + c);
    const u64 inv = last_q_modinv[y];
    ulonglong2 r;
    r.x = mul_barrett(sub_mod(x.x, last.x, m.q), inv, m);
    r.y = mul_barrett(sub_mod(x.y, last.y, m.q), inv, m);
    if (with_ct == 1 || (with_ct == 2 && z == 0)) {
        ulonglong2 cin = ld2(ct + ct_stride * b + loc);
        r.x = add_mod(cin.x, r.x, m.q);
        r.y = add_mod(cin.y, r.y, m.q);
    } else if (with_ct == 2) {
        r.x = add_mod(0, r.x, m.q);
        r.y = add_mod(0, r.y, m.q);
    }
    st2(out + out_stride * b + loc, r);
}

hipError_t rns_moddown_stage_two(const u64* in_last, u64 last_stride, const u64* in, u64 in_stride, int in_limbs,
                                 const u64* ct, u64 ct_stride, u64* out, u64 out_stride, const Mod* mods,
                                 const u64* last_q_modinv, int n_power, int cur_decomp, int with_ct, int batch,
                                 hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_moddown_stage_two, grid3(n_power, cur_decomp, 2 * batch), dim3(RNS_THREADS), 0, st,
                       in_last, last_stride, in, in_stride, in_limbs, ct, ct_stride, out, out_stride, mods,
                       last_q_modinv, n_power, cur_decomp, with_ct);
    return hipGetLastError();
}

__device__ __forceinline__ void acc128_rns(u64& hi, u64& lo, u64 a, u64 b)
{
    u64 h, l;
    mul64wide(a, b, h, l);
    lo += l;
    hi += h + (lo < l);
}

// ---------------------------------------------------------------- method II: digit -> Q~ base conversion
// IEEE single-precision overflow estimate exactly as the reference computes it:
// r = sum_i (float)y_i / (float)q_i in digit order, roundf.
// Digit g = the `cnt` primes from s0 on, raised to every modulus of Q~ by the fast base conversion with the
// fp32 overflow estimate (switchkey.cu:816-870).  MAXC: cnt padded (the digit's residues stay in registers; the
// padded ones are zero).  The products with the conversion matrix are summed as 128-bit integers and reduced
// once per target modulus -- the digit residue is NOT reduced into the target modulus first (the reference does:
// reduce_forced, then mult): below 2^61 it is a valid factor of the lazy sum, and the canonical result is the same.
template <int MAXC>
__global__ __launch_bounds__(RNS_THREADS) void k_base_conversion_DtoQtilde(
    const u64* __restrict__ in, u64 in_stride, u64* __restrict__ out, u64 out_stride, const Mod* __restrict__ mods,
    const u64* __restrict__ matrix, const u64* __restrict__ mi_inv, const u64* __restrict__ prod,
    const int* __restrict__ I_j_, const int* __restrict__ I_location_, int n_power, int rc, int l, int level)
{
    const u32 idx = blockIdx.x * RNS_THREADS + threadIdx.x;
    const int g = blockIdx.y;
    const int cnt = I_j_[g], s0 = I_location_[g];
    const u64* pin = in + in_stride * blockIdx.z + idx + ((u64) s0 << n_power);
    u64* po = out + out_stride * blockIdx.z + idx + (((u64) g * rc) << n_power);
    u64 partial[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; i++) partial[i] = pin[(u64) (i < cnt ? i : 0) << n_power]; // all loads first
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
        const int ii = i < cnt ? i : 0;
        const Mod m = mods[s0 + ii];
        const u64 v = mul_barrett(partial[i], mi_inv[s0 + ii], m);
        partial[i] = i < cnt ? v : 0;
        const float q = __fdiv_rn((float) v, (float) m.q);
        r = i < cnt ? __fadd_rn(r, q) : r; // same order and operations as the reference's loop
    }
    const u64 r_ = (u64) roundf(r);
    // Round 3: (sum_i y_i M_ik - r prod_k) mod q_k as ONE lazy sum with the term r (q_k - prod_k) and one Montgomery
    // reduction (tables m2_matrix_mg / m2_negprod_mg carry the 2^64): the same canonical residue as the reference's
    // reduce, multiply, subtract (switchkey.cu:846-868) for 47 instead of 75 instructions per target modulus
#pragma unroll 1
    for (int i = 0; i < rc; i++) {
        const Mod m = mods[(i < l) ? i : i + level];
        u64 hi, lo;
        dot128(partial, matrix + (u64) i * cnt + (u64) s0 * rc, cnt, hi, lo);
        acc128_rns(hi, lo, r_, prod[i + g * rc]);
        po[(u64) i << n_power] = redc128(hi, lo, m);
    }
}

hipError_t rns_base_conversion_DtoQtilde(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                                         const u64* matrix, const u64* mi_inv, const u64* prod, const int* I_j,
                                         const int* I_location, int n_power, int d, int rc, int l, int level,
                                         int max_cnt, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    if (max_cnt < 1 || max_cnt > 32) return hipErrorInvalidValue;
    dim3 g((1u << n_power) / RNS_THREADS, d, batch);
#define LAUNCH(M)                                                                                                  \
    hipLaunchKernelGGL(k_base_conversion_DtoQtilde<M>, g, dim3(RNS_THREADS), 0, st, in, in_stride, out, out_stride, \
                       mods, matrix, mi_inv, prod, I_j, I_location, n_power, rc, l, level)
    if (max_cnt <= 2) LAUNCH(2);
    else if (max_cnt <= 4) LAUNCH(4);
    else if (max_cnt <= 8) LAUNCH(8);
    else if (max_cnt <= 16) LAUNCH(16);
    else LAUNCH(32);
#undef LAUNCH
    return hipGetLastError();
}

// Mod-down by the P_size special primes, one at a time, last first (the loop of switchkey.cu:497-534 / 1239-1276 /
// 1650-1683): step i takes lh_i = (special prime P_size-1-i's limb + half_i), and every remaining limb (special
// or not) becomes (x - (lh_i mod q - half_mod_i)) * P_i^-1.
// All Q_cur limbs of one coefficient (and part) per thread.  The reference's kernel (one thread per limb) redoes
// the mod-down chain AMONG the special primes -- P (P - 1) / 2 reductions that do not depend on the limb -- for
// every one of the Q_cur limbs; here it runs once (r[k] = special prime P_size - 1 - k, static register indices)
// and leaves the P_size values lh_i the limbs need.
// galois_elt != 0: the result goes through the coefficient permutation out[(i g) mod N] = +-v
// (divide_round_lastq_permute_*_kernel with P_size > 1, switchkey.cu:1621-1813).
template <int PMAX>
__global__ __launch_bounds__(RNS_THREADS) void k_moddown_extended(
    const u64* __restrict__ in, u64 in_stride, const u64* ct, u64 ct_stride, u64* out, u64 out_stride,
    const Mod* __restrict__ mods, const u64* __restrict__ half, const u64* __restrict__ half_mod,
    const u64* __restrict__ last_q_modinv, int n_power, int Qp_cur, int Q_cur, int first_Qp, int first_Q,
    int P_size, int with_ct, int galois_elt)
{
    const u32 idx = blockIdx.x * RNS_THREADS + threadIdx.x;
    const int z = blockIdx.z & 1, b = blockIdx.z >> 1;
    const u64* pin = in + in_stride * b + (((u64) Qp_cur << n_power) * z) + idx;
    u64 r[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; k++) r[k] = pin[(u64) (Q_cur + (k < P_size ? P_size - 1 - k : 0)) << n_power];
    int loc[PMAX];
    {
        int location_ = 0;
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            loc[i] = location_;
            location_ += first_Qp - 1 - i;
        }
    }
#pragma unroll
    for (int i = 0; i < PMAX; i++) {
        if (i < P_size) {
            r[i] = add_mod(r[i], half[i], mods[first_Qp - 1 - i].q); // lh_i
#pragma unroll
            for (int k = i + 1; k < PMAX; k++) {
                if (k < P_size) {
                    const int j = P_size - 1 - k;
                    const Mod mj = mods[first_Q + j];
                    u64 t1 = reduce64(r[i], mj);
                    t1 = sub_mod(t1, half_mod[loc[i] + first_Q + j], mj.q);
                    t1 = sub_mod(r[k], t1, mj.q);
                    r[k] = mul_barrett(t1, last_q_modinv[loc[i] + first_Q + j], mj);
                }
            }
        }
    }
    const u64 part = ((u64) Q_cur << n_power) * z;
    const bool add = with_ct == 1 || (with_ct == 2 && z == 0);
    const u64* pc = ct + (add ? ct_stride * b + part + idx : 0);
    const u32 raw = idx * (u32) galois_elt;
    const bool neg = galois_elt && ((raw >> n_power) & 1);
    u64* po = out + out_stride * b + part + (galois_elt ? (raw & ((1u << n_power) - 1)) : idx);
    for (int y0 = 0; y0 < Q_cur; y0 += 4) {
        u64 x[4], c4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { // loads first
            const int y = (y0 + u < Q_cur) ? y0 + u : Q_cur - 1;
            x[u] = pin[(u64) y << n_power];
            c4[u] = add ? pc[(u64) y << n_power] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int y = (y0 + u < Q_cur) ? y0 + u : Q_cur - 1;
            const Mod m = mods[y];
            u64 v = x[u];
#pragma unroll
            for (int i = 0; i < PMAX; i++) {
                if (i < P_size) {
                    u64 t1 = reduce64(r[i], m);
                    t1 = sub_mod(t1, half_mod[loc[i] + y], m.q);
                    t1 = sub_mod(v, t1, m.q);
                    v = mul_barrett(t1, last_q_modinv[loc[i] + y], m);
                }
            }
            v = add ? add_mod(c4[u], v, m.q) : v;
            x[u] = neg ? m.q - v : v; // no zero test on the negation: reference switchkey.cu:1694,1711
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (y0 + u < Q_cur) po[(u64) (y0 + u) << n_power] = x[u];
    }
}

// First half of the same mod-down for the NTT-domain form (context.cpp: m2_md_*): from the special limbs of one
// coefficient (coefficient domain) the value u_y = sum_i lh_i G_i[y] - C[y] for every limb y of Q; its transform
// is what the forward pass's epilogue subtracts from the accumulated limb before multiplying by W0[y].
// in: [2][Qp_cur][N] per item (only the P_size special slots are read), out: [2][Q_cur][N].
template <int PMAX>
__global__ __launch_bounds__(RNS_THREADS) void k_moddown_multi_stage_one(
    const u64* __restrict__ in, u64 in_stride, u64* __restrict__ out, u64 out_stride, const Mod* __restrict__ mods,
    const u64* __restrict__ half, const u64* __restrict__ half_mod, const u64* __restrict__ last_q_modinv,
    const u64* __restrict__ G, const u64* __restrict__ C, int n_power, int Qp_cur, int Q_cur, int first_Qp,
    int first_Q, int P_size)
{
    const u32 idx = blockIdx.x * RNS_THREADS + threadIdx.x;
    const int z = blockIdx.z & 1, b = blockIdx.z >> 1;
    const u64* pin = in + in_stride * b + (((u64) Qp_cur << n_power) * z) + idx;
    u64 r[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; k++) r[k] = pin[(u64) (Q_cur + (k < P_size ? P_size - 1 - k : 0)) << n_power];
    int loc[PMAX];
    {
        int location_ = 0;
#pragma unroll
        for (int i = 0; i < PMAX; i++) {
            loc[i] = location_;
            location_ += first_Qp - 1 - i;
        }
    }
#pragma unroll
    for (int i = 0; i < PMAX; i++) {
        if (i < P_size) {
            r[i] = add_mod(r[i], half[i], mods[first_Qp - 1 - i].q); // lh_i
#pragma unroll
            for (int k = i + 1; k < PMAX; k++) {
                if (k < P_size) {
                    const int j = P_size - 1 - k;
                    const Mod mj = mods[first_Q + j];
                    u64 t1 = reduce64(r[i], mj);
                    t1 = sub_mod(t1, half_mod[loc[i] + first_Q + j], mj.q);
                    t1 = sub_mod(r[k], t1, mj.q);
                    r[k] = mul_barrett(t1, last_q_modinv[loc[i] + first_Q + j], mj);
                }
            }
        } else {
            r[i] = 0;
        }
    }
    u64* po = out + out_stride * b + (((u64) Q_cur << n_power) * z) + idx;
#pragma unroll 1
    for (int y = 0; y < Q_cur; y++) {
        const Mod m = mods[y];
        u64 hi, lo;
        dot128(r, G + (u64) y * P_size, P_size, hi, lo); // lh_i, G_i < 2^61: un-reduced factors of the lazy sum
        po[(u64) y << n_power] = sub_mod(reduce128(hi, lo, m), C[y], m.q);
    }
}

hipError_t rns_moddown_multi_stage_one(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                                       const u64* half, const u64* half_mod, const u64* last_q_modinv, const u64* G,
                                       const u64* C, int n_power, int Qp_cur, int Q_cur, int first_Qp, int first_Q,
                                       int P_size, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (P_size > 15 || P_size < 2) return hipErrorInvalidValue;
    dim3 g((1u << n_power) / RNS_THREADS, 1, 2 * batch);
#define LAUNCH(M)                                                                                                  \
    hipLaunchKernelGGL(k_moddown_multi_stage_one<M>, g, dim3(RNS_THREADS), 0, st, in, in_stride, out, out_stride, mods, \
                       half, half_mod, last_q_modinv, G, C, n_power, Qp_cur, Q_cur, first_Qp, first_Q, P_size)
    if (P_size <= 2) LAUNCH(2);
    else if (P_size <= 4) LAUNCH(4);
    else if (P_size <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    return hipGetLastError();
}

hipError_t rns_moddown_extended(const u64* in, u64 in_stride, const u64* ct, u64 ct_stride, u64* out,
                                u64 out_stride, const Mod* mods, const u64* half, const u64* half_mod,
                                const u64* last_q_modinv, int n_power, int Qp_cur, int Q_cur, int first_Qp,
                                int first_Q, int P_size, int with_ct, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    if (P_size > 15 || P_size < 1) return hipErrorInvalidValue;
    dim3 g((1u << n_power) / RNS_THREADS, 1, 2 * batch);
#define LAUNCH(M)                                                                                                    \
    hipLaunchKernelGGL(k_moddown_extended<M>, g, dim3(RNS_THREADS), 0, st, in, in_stride, ct, ct_stride, out, out_stride, \
                       mods, half, half_mod, last_q_modinv, n_power, Qp_cur, Q_cur, first_Qp, first_Q, P_size, with_ct,    \
                       galois_elt)
#define MODDOWN_EXT_DISPATCH \
    if (P_size <= 2) LAUNCH(2); \
    else if (P_size <= 4) LAUNCH(4); \
    else if (P_size <= 8) LAUNCH(8); \
    else LAUNCH(16)
    const int galois_elt = 0;
    MODDOWN_EXT_DISPATCH;
    return hipGetLastError();
}

// ---------------------------------------------------------------- mod-down + Galois permutation
// One coefficient per thread: the destination index i*g mod N scatters.
#define MDP_PER 4
__global__ __launch_bounds__(RNS_THREADS) void k_moddown_permute(
    const u64* __restrict__ in, u64 in_stride, const u64* __restrict__ in2, u64 in2_stride, u64* __restrict__ out,
    u64 out_stride, const Mod* __restrict__ mods, const u64* __restrict__ half, const u64* __restrict__ half_mod,
    const u64* __restrict__ last_q_modinv, int galois_elt, int n_power, int Qp_cur, int Q_cur, int first_Qp,
    int first_Q, int P_size)
{
    // MDP_PER coefficients per thread, RNS_THREADS apart (a workgroup per 256 coefficients is bound by the rate
    // workgroups can be started at: 229 k of them at C3)
    const int y = blockIdx.y;
    const int z = blockIdx.z & 1, b = blockIdx.z >> 1;
    const Mod m = mods[y];
    const u32 idx0 = blockIdx.x * (RNS_THREADS * MDP_PER) + threadIdx.x;
    const u64* pin0 = in + in_stride * b + (((u64) Qp_cur << n_power) * z);
    u64 xs[MDP_PER], ls[MDP_PER], cs[MDP_PER];
#pragma unroll
    for (int r = 0; r < MDP_PER; r++) {
        const u32 idx = idx0 + RNS_THREADS * r;
        xs[r] = pin0[((u64) y << n_power) + idx];
        ls[r] = pin0[((u64) Q_cur << n_power) + idx];
        cs[r] = (z == 0) ? in2[in2_stride * b + ((u64) y << n_power) + idx] : 0;
    }
#pragma unroll
    for (int r = 0; r < MDP_PER; r++) {
        const u32 idx = idx0 + RNS_THREADS * r;
        u64 x = xs[r];
        {
            u64 l = add_mod(ls[r], half[0], mods[first_Qp - 1].q);
            l = reduce64(l, m);
            l = sub_mod(l, half_mod[y], m.q);
            l = sub_mod(x, l, m.q);
            x = mul_barrett(l, last_q_modinv[y], m);
        }
        if (z == 0) x = add_mod(cs[r], x, m.q);
        const u32 raw = idx * (u32) galois_elt;
        const u32 dst = raw & ((1u << n_power) - 1);
        if ((raw >> n_power) & 1) x = m.q - x; // no zero test: reference switchkey.cu:1694,1711
        out[out_stride * b + (((u64) Q_cur << n_power) * z) + ((u64) y << n_power) + dst] = x;
    }
}

hipError_t rns_moddown_permute(const u64* in, u64 in_stride, const u64* in2, u64 in2_stride, u64* out,
                               u64 out_stride, const Mod* mods, const u64* half, const u64* half_mod,
                               const u64* last_q_modinv, int galois_elt, int n_power, int Qp_cur, int Q_cur,
                               int first_Qp, int first_Q, int P_size, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    dim3 g((1u << n_power) / (RNS_THREADS * MDP_PER), Q_cur, 2 * batch);
    if (P_size > 15) return hipErrorInvalidValue;
    if (P_size == 1)
        hipLaunchKernelGGL(k_moddown_permute, g, dim3(RNS_THREADS), 0, st, in, in_stride, in2, in2_stride, out,
                           out_stride, mods, half, half_mod, last_q_modinv, galois_elt, n_power, Qp_cur, Q_cur,
                           first_Qp, first_Q, P_size);
    else {
        // several special primes: all limbs of a coefficient per thread (k_moddown_extended), the sum with the
        // other ciphertext part (part 0 only) and the permutation on the way out
        const u64* ct = in2;
        const u64 ct_stride = in2_stride;
        const int with_ct = 2;
        dim3 g((1u << n_power) / RNS_THREADS, 1, 2 * batch);
        MODDOWN_EXT_DISPATCH;
    }
    return hipGetLastError();
}
#undef LAUNCH
#undef MODDOWN_EXT_DISPATCH

// ---------------------------------------------------------------- strided limb copy
__global__ __launch_bounds__(RNS_THREADS) void k_copy_limbs(const u64* __restrict__ in, u64 in_part_stride,
                                                            u64 in_stride, u64* __restrict__ out,
                                                            u64 out_part_stride, u64 out_stride, int n_power,
                                                            int parts)
{
    const int z = blockIdx.z % parts, b = blockIdx.z / parts;
    const u64 c = coeff0() + ((u64) blockIdx.y << n_power);
    st2(out + out_stride * b + out_part_stride * z + c, ld2(in + in_stride * b + in_part_stride * z + c));
}

hipError_t rns_copy_limbs(const u64* in, u64 in_part_stride, u64 in_stride, u64* out, u64 out_part_stride,
                          u64 out_stride, int n_power, int limbs, int parts, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_copy_limbs, grid3(n_power, limbs, parts * batch), dim3(RNS_THREADS), 0, st, in,
                       in_part_stride, in_stride, out, out_part_stride, out_stride, n_power, parts);
    return hipGetLastError();
}

// ---------------------------------------------------------------- Galois automorphism in the NTT domain
// b(X) = a(X^g): slot j of the transform holds the value at psi^(2 br(j) + 1) (reference switchkey.cu:1461-1476),
// and b(psi^e) = a(psi^(e g mod 2N)), so the automorphism is the slot gather out[j] = in[j'] with
// 2 br(j') + 1 = (2 br(j) + 1) g mod 2N -- no sign, no arithmetic, the same map for every limb.  (The reference
// permutes in the coefficient domain, fused into the mod-down, switchkey.cu:1621-1813, and transforms after
// that; the canonical residues are the same.)  Coalesced 16-byte writes, 8-byte gathers that stay inside
// one limb (512 KiB at N = 2^16: L2-resident while the workgroups of that limb run).
__global__ __launch_bounds__(RNS_THREADS) void k_permute_ntt(const u64* __restrict__ in, u64 in_stride,
                                                             u64* __restrict__ out, u64 out_stride, int n_power,
                                                             u32 galois_elt)
{
    const u32 j = (u32) coeff0();
    const u32 mask = (2u << n_power) - 1u;
    const u32 e0 = (2u * (__brev(j) >> (32 - n_power)) + 1u) * galois_elt & mask;
    const u32 e1 = (2u * (__brev(j + 1) >> (32 - n_power)) + 1u) * galois_elt & mask;
    const u32 s0 = __brev((e0 - 1u) >> 1) >> (32 - n_power), s1 = __brev((e1 - 1u) >> 1) >> (32 - n_power);
    const u64* pi = in + in_stride * blockIdx.z + ((u64) blockIdx.y << n_power);
    ulonglong2 v;
    v.x = pi[s0];
    v.y = pi[s1];
    st2(out + out_stride * blockIdx.z + ((u64) blockIdx.y << n_power) + j, v);
}

hipError_t rns_permute_ntt(const u64* in, u64 in_stride, u64* out, u64 out_stride, int galois_elt, int n_power,
                           int limbs, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_permute_ntt, grid3(n_power, limbs, batch), dim3(RNS_THREADS), 0, st, in, in_stride, out,
                       out_stride, n_power, (u32) galois_elt);
    return hipGetLastError();
}

__global__ __launch_bounds__(RNS_THREADS) void k_copy_diag(const u64* __restrict__ in, u64 in_stride,
                                                           u64* __restrict__ out, u64 out_stride, int n_power, int rc)
{
    const u64 c = coeff0();
    const u64 d = blockIdx.y;
    st2(out + out_stride * blockIdx.z + ((d * (rc + 1)) << n_power) + c,
        ld2(in + in_stride * blockIdx.z + (d << n_power) + c));
}

hipError_t rns_copy_diag(const u64* in, u64 in_stride, u64* out, u64 out_stride, int n_power, int limbs, int rc,
                         int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    hipLaunchKernelGGL(k_copy_diag, grid3(n_power, limbs, batch), dim3(RNS_THREADS), 0, st, in, in_stride, out,
                       out_stride, n_power, rc);
    return hipGetLastError();
}

// hi:lo += a * b (a, b < 2^61: the running sum of a few dozen such products stays below 2^128)
__device__ __forceinline__ void acc128(u64& hi, u64& lo, u64 a, u64 b)
{
    u64 h, l;
    mul64wide(a, b, h, l);
    lo += l;
    hi += h + (lo < l);
}

// ---------------------------------------------------------------- BFV BEHZ kernels
// One thread per coefficient; the per-coefficient vectors (ibase / Bsk
// residues) must stay in registers, so the kernels are instantiated for a
// padded base size MAXB with every loop fully unrolled and guarded by the
// (wave-uniform) real size.  Runtime-indexed arrays would live in scratch.
// 64 = the reference's MAX_BSK_SIZE (src/include/heongpu/kernel/defines.h:26); a 128-bit accumulator
// takes 64 products of two 61-bit values (each < 2^122) without overflow.
#define BEHZ_MAX 64

// Branch-free bodies: entries beyond the real base size are computed on a clamped (valid) index and
// zeroed with a wave-uniform select, the inner products run over all MAXB slots (a zero operand adds
// nothing).  Guarding every slot with `if (i < ib)` instead made the compiler carry the whole register
// array through a chain of conditional blocks -- 236 registers and one wave per SIMD at MAXB = 16 -- so
// MAXB is kept close to the real size (behz_slots) and the few wasted products are accepted.
// SPLIT (launches that leave the chip mostly empty: one ciphertext pair at N = 2^16 is 1024 workgroups of 10 k
// instructions per thread): the four wavefronts of a workgroup share 64 coefficients; every wavefront repeats the
// short per-coefficient prologue and takes every fourth row of the base conversion -- four times the threads, a
// quarter of the rows each, the row tables still wave-uniform.
template <int MAXB, bool SPLIT>
__global__ __launch_bounds__(RNS_THREADS) void k_fast_convertion(const u64* __restrict__ in1, u64 s1,
                                                                 const u64* __restrict__ in2, u64 s2,
                                                                 u64* __restrict__ out1, u64 so, BehzDev b,
                                                                 int n_power)
{
    const u32 idx = SPLIT ? blockIdx.x * 64 + (threadIdx.x & 63) : blockIdx.x * RNS_THREADS + threadIdx.x;
    const int row0 = SPLIT ? (int) (threadIdx.x >> 6) : 0, row_step = SPLIT ? 4 : 1;
    const int idy = blockIdx.y;
    const int ib = b.ibase_size, ob = b.obase_size;
    const u64* input = ((idy >> 1) == 0) ? (in1 + s1 * blockIdx.z) : (in2 + s2 * blockIdx.z);
    const u64 location = idx + ((u64) ((idy & 1) * ib) << n_power);
    u64 temp[MAXB];
    u64* po = out1 + so * blockIdx.z + idx + ((u64) (idy * (ob + ib)) << n_power);
#pragma unroll
    for (int i = 0; i < MAXB; i++) {
        const int ii = i < ib ? i : ib - 1;
        const Mod mi = b.ibase[ii];
        const u64 v = input[location + ((u64) ii << n_power)];
        if (!SPLIT || (i & 3) == row0) po[(u64) ii << n_power] = v; // slots beyond ib rewrite limb ib-1 with its own value
        const u64 t = mul_barrett(v, b.mtilde_inv_punct[ii], mi); // x * m_tilde * (q/q_i)^-1, one product
        temp[i] = i < ib ? t : 0;
    }
    // m_tilde channel: m_tilde = 2^32, so reduction, product and sum modulo it are plain 32-bit
    // arithmetic (the same canonical values as the Barrett routines of the reference, multiplication.cu:44-60)
    u32 acc_mt32 = 0;
#pragma unroll
    for (int j = 0; j < MAXB; j++) acc_mt32 += (u32) temp[j] * (u32) b.base_change_matrix_m_tilde[j < ib ? j : 0];
    const u64 mt = b.m_tilde.q;
    u64 r_mt = (u64) (u32) (acc_mt32 * (u32) b.inv_prod_q_mod_m_tilde);
    r_mt = mt - r_mt;
    // Row i of the conversion into Bsk with its trailing factors folded into the constants (BehzDev::fc_matrix):
    //   out_i = ((sum_j temp_j M_ij) + t3_i prod_q_i) * inv_m_tilde_i = sum_j temp_j M'_ij + t3_i c1_i   (mod Bsk_i),
    // t3_i = r_mt, or r_mt - m_tilde (as Bsk_i - m_tilde + r_mt) when r_mt is in the upper half: one lazy 128-bit
    // sum and ONE reduction per row instead of a reduction and two Barrett products (the stored value is the
    // canonical residue of the same integer either way, multiplication.cu:66-90).
    const bool mt_neg = r_mt >= (mt >> 1);
#pragma unroll 1
    for (int i = row0; i < ob; i += row_step) {
        const Mod mo = b.obase[i];
        const u64* __restrict__ row = b.fc_matrix + i * ib;
        u64 hi, lo;
        dot128(temp, row, ib, hi, lo);
        const u64 t3 = mt_neg ? mo.q - mt + r_mt : r_mt; // < 2^61: a valid factor of the lazy sum
        acc128(hi, lo, t3, b.fc_c1[i]);
        po[(u64) (i + ib) << n_power] = redc128(hi, lo, mo); // the tables carry the 2^64
    }
}

// padded size for a base of m moduli: steps of 2 up to 16, of 4 up to 32, of 8 up to 64
static int behz_slots(int m)
{
    if (m <= 16) return (m + 1) & ~1;
    if (m <= 32) return (m + 3) & ~3;
    return (m + 7) & ~7;
}
#define BEHZ_DISPATCH(m)                                                                     \
    switch (behz_slots(m)) {                                                                 \
        case 2: LAUNCH(2); break;   case 4: LAUNCH(4); break;   case 6: LAUNCH(6); break;    \
        case 8: LAUNCH(8); break;   case 10: LAUNCH(10); break; case 12: LAUNCH(12); break;  \
        case 14: LAUNCH(14); break; case 16: LAUNCH(16); break; case 20: LAUNCH(20); break;  \
        case 24: LAUNCH(24); break; case 28: LAUNCH(28); break; case 32: LAUNCH(32); break;  \
        case 40: LAUNCH(40); break; case 48: LAUNCH(48); break; case 56: LAUNCH(56); break;  \
        default: LAUNCH(64); break;                                                          \
    }

// The split forms when the plain launch has fewer than 320 workgroups (measured, one ciphertext pair, plain / split:
// N = 2^14 fast_floor 20.2 / 15.3 us, fast_convertion 14.7 / 13.6; N = 2^15 55 / 58, 35 / 39; N = 2^16 288 / 395,
// 169 / 279 -- from 384 workgroups on the kernels are bound by their instruction count, which the split raises);
// the context option behz_split = 0 / 1 forces the choice (BehzDev::split; the tests run both forms on one context).
static bool behz_split(const BehzDev& b, int n_power, int polys, int batch)
{
    if (b.split >= 0) return b.split != 0;
    return ((long) (1u << n_power) / RNS_THREADS) * polys * batch < 320;
}

hipError_t rns_fast_convertion(const u64* in1, u64 s1, const u64* in2, u64 s2, u64* out, u64 so,
                               const BehzDev& b, int n_power, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    if (b.ibase_size > BEHZ_MAX || b.obase_size > BEHZ_MAX || b.ibase_size < 1) return hipErrorInvalidValue;
    if (behz_split(b, n_power, 4, batch)) {
        dim3 g((1u << n_power) / 64, 4, batch);
#define LAUNCH(M) hipLaunchKernelGGL((k_fast_convertion<M, true>), g, dim3(RNS_THREADS), 0, st, in1, s1, in2, s2, out, so, b, n_power)
        BEHZ_DISPATCH(b.ibase_size)
#undef LAUNCH
        return hipGetLastError();
    }
    dim3 g((1u << n_power) / RNS_THREADS, 4, batch);
#define LAUNCH(M) hipLaunchKernelGGL((k_fast_convertion<M, false>), g, dim3(RNS_THREADS), 0, st, in1, s1, in2, s2, out, so, b, n_power)
    BEHZ_DISPATCH(b.ibase_size)
#undef LAUNCH
    return hipGetLastError();
}

// SPLIT: as in k_fast_convertion; the rows of the first conversion computed by the four wavefronts meet in LDS
// ([row][coefficient], one barrier) before every wavefront runs its quarter of the second one.
template <int MAXB, bool SPLIT>
__global__ __launch_bounds__(RNS_THREADS) void k_fast_floor(const u64* __restrict__ in, u64 si,
                                                            u64* __restrict__ out1, u64 so, BehzDev b, int n_power)
{
    const u32 idx = SPLIT ? blockIdx.x * 64 + (threadIdx.x & 63) : blockIdx.x * RNS_THREADS + threadIdx.x;
    const int row0 = SPLIT ? (int) (threadIdx.x >> 6) : 0, row_step = SPLIT ? 4 : 1;
    __shared__ u64 meet[SPLIT ? (MAXB + 1) * 64 : 1];
    const int idy = blockIdx.y;
    const int ib = b.ibase_size, ob = b.obase_size;
    const u64* pq = in + si * blockIdx.z + idx + ((u64) (idy * (ib + ob)) << n_power);
    const u64* pB = pq + ((u64) ib << n_power);
    u64 reg_q[MAXB], temp3[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; i++) {
        const int ii = i < ib ? i : ib - 1;
        const u64 v = mul_barrett(pq[(u64) ii << n_power], b.t_inv_punct[ii], b.ibase[ii]); // x * t * (q/q_i)^-1
        reg_q[i] = i < ib ? v : 0;
    }
    // rows 0 .. ob-2: the moduli of B (-> temp3), row ob-1: m_sk.  (The forms are written out: sharing
    // the row computation through a lambda cost 30 registers.)  A row with its trailing factors folded into the
    // constants (BehzDev::ff_matrix, ff_tc):
    //   v_i = (x_Bsk_i t - sum_j reg_q_j M_ij) c_i = x_Bsk_i (t c_i) + sum_j reg_q_j (-M_ij c_i)   (mod Bsk_i)
    // -- one lazy 128-bit sum and one reduction instead of a reduction and two Barrett products; the canonical residue
    // of the same integer as the reference's chain (multiplication.cu:160-205).
    u64 reg_Bsk_last = 0;
    if constexpr (SPLIT) {
        const int lane = threadIdx.x & 63;
#pragma unroll 1
        for (int i = row0; i < ob; i += row_step) {
            const Mod mo = b.obase[i];
            u64 hi, lo;
            dot128(reg_q, b.ff_matrix + i * ib, ib, hi, lo);
            acc128(hi, lo, pB[(u64) i << n_power], b.ff_tc[i]);
            meet[i * 64 + lane] = redc128(hi, lo, mo);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXB; k++) temp3[k] = (k < ob - 1) ? meet[k * 64 + lane] : 0;
        reg_Bsk_last = meet[(ob - 1) * 64 + lane];
    } else {
        // a run-time loop over the rows keeps code size and register pressure down; the row's slot of temp3 is picked
        // with wave-uniform selects (a fully unrolled form for small bases was measured slower in round 3 and is gone)
#pragma unroll
        for (int i = 0; i < MAXB; i++) temp3[i] = 0;
#pragma unroll 1
        for (int i = 0; i < ob; i++) {
            const Mod mo = b.obase[i];
            u64 hi, lo;
            dot128(reg_q, b.ff_matrix + i * ib, ib, hi, lo);
            acc128(hi, lo, pB[(u64) i << n_power], b.ff_tc[i]);
            const bool last = i == ob - 1;
            const u64 v = redc128(hi, lo, mo);
            if (last) reg_Bsk_last = v;
#pragma unroll
            for (int k = 0; k < MAXB; k++) temp3[k] = (!last && k == i) ? v : temp3[k];
        }
    }
    const Mod msk = b.obase[ob - 1];
    u64 hi, lo;
    dot128(temp3, b.ff_msk_matrix, ob - 1, hi, lo);
    u64 t4sk = redc128(hi, lo, msk);
    u64 alpha_sk = sub_mod(msk.q, reg_Bsk_last, msk.q);
    alpha_sk = add_mod(alpha_sk, t4sk, msk.q);
    alpha_sk = mul_barrett(alpha_sk, b.inv_prod_B_mod_m_sk, msk);
    const bool neg = alpha_sk > (msk.q >> 1);
    u64* po = out1 + so * blockIdx.z + idx + ((u64) (idy * ib) << n_power);
#pragma unroll 1
    for (int i = row0; i < ib; i += row_step) {
        const Mod mi = b.ibase[i];
        const u64* __restrict__ row = b.ff_q_matrix + i * (ob - 1);
        u64 h2, l2;
        dot128(temp3, row, ob - 1, h2, l2); // un-reduced: 64 terms below 2^122 fit 128 bits
        // + the m_sk correction as one more term of the same lazy sum (multiplication.cu:243-262): alpha_sk is in
        // the upper half: (m_sk - alpha_sk) * prod_B, else alpha_sk * (q_i - prod_B) -- both factors below 2^61, no
        // reduction of alpha_sk into q_i first
        acc128(h2, l2, neg ? msk.q - alpha_sk : alpha_sk, neg ? b.ff_prod_B[i] : b.ff_neg_prod_B[i]);
        po[(u64) i << n_power] = redc128(h2, l2, mi);
    }
}

hipError_t rns_fast_floor(const u64* in, u64 si, u64* out, u64 so, const BehzDev& b, int n_power, int batch,
                          hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    if (b.ibase_size > BEHZ_MAX || b.obase_size > BEHZ_MAX || b.ibase_size < 1 || b.obase_size < 2)
        return hipErrorInvalidValue;
    const int m = b.ibase_size > b.obase_size - 1 ? b.ibase_size : b.obase_size - 1;
    if (behz_split(b, n_power, 3, batch)) {
        dim3 g((1u << n_power) / 64, 3, batch);
#define LAUNCH(M) hipLaunchKernelGGL((k_fast_floor<M, true>), g, dim3(RNS_THREADS), 0, st, in, si, out, so, b, n_power)
        BEHZ_DISPATCH(m)
#undef LAUNCH
        return hipGetLastError();
    }
    dim3 g((1u << n_power) / RNS_THREADS, 3, batch);
    // the 28-slot instance comes out of the register allocator at 256 registers (one wave per SIMD), the 32-slot one
    // at 250 (two): bases of 25..28 take the larger one (surplus slots hold zeros)
    const int m_plain = (behz_slots(m) == 28) ? 29 : m;
#define LAUNCH(M) hipLaunchKernelGGL((k_fast_floor<M, false>), g, dim3(RNS_THREADS), 0, st, in, si, out, so, b, n_power)
    BEHZ_DISPATCH(m_plain)
#undef LAUNCH
    return hipGetLastError();
}

} // namespace hegpu

// ntt.hip -- batched negacyclic NTT / INTT over Z_q[X]/(X^N+1) for gfx950.
//
// Replaces gpuntt::GPU_NTT{,_Inplace}, GPU_INTT{,_Inplace},
// GPU_NTT_Modulus_Ordered_Inplace and GPU_NTT_Poly_Ordered_Inplace of the
// reference's unvendored thirdparty/GPU-NTT (call sites: reference
// src/lib/host/ckks/operator.cu:919,956,996,1011,1197; bfv/operator.cu:393,410).
// Definition: natural-order coefficients in, bit-reversed
// evaluation order out; stage with m groups uses root table[m + group],
// table[r] = psi^bitreverse(r); inverse = Gentleman-Sande + N^-1.
//
// MI355X design (NOT a translation of GPU-NTT's CUDA kernels):
//  * two passes per transform, N = 2^S1 x 256: a "column" pass does the S1
//    large-stride stages on 4096-element tiles [2^S1 rows][4096>>S1 cols],
//    a "row" pass does the 8 contiguous stages on 16 rows of 256.  Every
//    global access is a 128-byte-or-wider coalesced segment; each pass moves
//    the limb exactly once (2W bytes per pass).
//  * 256 threads x 16 coefficients per workgroup: radix-16 butterflies run
//    entirely in VGPRs (4 stages per LDS exchange), LDS (34-36 KiB per
//    workgroup, padded against bank conflicts) is only the transpose medium.
//  * Shoup/Harvey lazy butterflies with a 3-multiply quotient estimate and a
//    multiply-accumulate chain against -q: 9 32-bit multiplies per butterfly,
//    values kept in [0,8q) (forward) / [0,4q) (inverse) with a single exact
//    correction at the end, so results are canonical residues --
//    bit-identical to any exact NTT (q < 2^61 keeps 8q inside 64 bits).
//  * wave-uniform twiddles (column pass, first round) are fetched through
//    the scalar cache; the rest are 16-byte (w, w') pairs read as dwordx4.
#include "ntt.hpp"
#define HEGPU_FP_TU ntt // (names this file's table reader in the instrumented test build, see fpmod.cuh)
#include "fpmod.cuh"

namespace hegpu {

#define NTT_THREADS 256

// Measurement builds (tools/exp/ntt_exp.hip) swap the global accessors and the butterflies for knock-outs through a
// header of their own, tools/exp/ntt_ablation.cuh.  The product has no numeric switch for that: the macro must name
// that file, so a stray -D cannot turn this library into one that builds and computes garbage.
#ifdef NTT_ABLATION_HEADER
#include NTT_ABLATION_HEADER
#else
__device__ __forceinline__ u64 gld(const u64* p) { return *p; }
__device__ __forceinline__ void gst(u64* p, u64 v) { *p = v; }
#define NTT_ABLATE_BFLY(x, y, w)
#define NTT_ABLATE_FPBFLY(x, y, w)
#define NTT_ABLATE_TW(load, root0, s) (load)
#define NTT_FP_TW(t, i, c) ((t)[i])
#endif

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return (x >= m) ? x - m : x; }

// A twiddle table seen through the CONSTANT address space.  hipcc turns a load with a wave-uniform address into a scalar
// load only when it can prove that nothing in the kernel has written the memory before it; in a persistent kernel (a loop
// with global stores in it) it cannot, and the "uniform" twiddles come back as vector loads whose s_waitcnt vmcnt(0) also
// waits for every prefetch in flight.  The tables are written once, by the context, before any launch: constant it is.
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
struct ConstTw {
    const u64x2_t __attribute__((address_space(4)))* p;
    __device__ __forceinline__ ulonglong2 operator[](u32 i) const
    {
        const u64x2_t v = p[i];
        return make_ulonglong2(v.x, v.y);
    }
};
__device__ __forceinline__ ConstTw const_tw(const ulonglong2* t)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return ConstTw{(const u64x2_t __attribute__((address_space(4)))*) reinterpret_cast<const u64x2_t*>(t)};
#pragma clang diagnostic pop
}

// scalar-cache reads of launch-constant tables from inside a persistent loop (see ConstTw)
__device__ __forceinline__ int ld_const_i32(const int* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return *(const int __attribute__((address_space(4)))*) p;
#pragma clang diagnostic pop
}
__device__ __forceinline__ Mod ld_const_mod(const Mod* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const u64 __attribute__((address_space(4)))* w = (const u64 __attribute__((address_space(4)))*) reinterpret_cast<const u64*>(p);
#pragma clang diagnostic pop
    static_assert(sizeof(Mod) == 56, "seven words");
    Mod m;
    m.q = w[0];
    m.mu = w[1];
    m.r_hi = w[2];
    m.r_lo = w[3];
    m.r64 = w[4];
    m.qinv = w[5];
    const u64 bf = w[6];
    m.bit = (u32) bf;
    m.fp = (u32) (bf >> 32);
    return m;
}

// ------------------------------------------------------------------ FP64 path (arithmetic: fpmod.cuh)
__device__ __forceinline__ void fp_ct_bfly(double& x, double& y, ulonglong2 w, const FC& c)
{
    NTT_ABLATE_FPBFLY(x, y, w);
    const double t = fp_mul(y, as_f64(w.x), as_f64(w.y), c);
    y = x - t;
    x = x + t;
    FP_AUDIT_VAL(c, FPM_SUM, x);
    FP_AUDIT_VAL(c, FPM_SUM, y);
}
// Where the centred reductions of the FP64 forward transform go (round 3).  A stage takes |x| <= b q to at most
// (1.25 b + 0.5) q (fpmod.cuh, q < 2^50), and everything stays exact while |x| < 2^53 = 8 * 2^50: from b = 1/2 that is
// SIX stages (1.125, 1.91, 2.88, 4.10, 5.63, 7.54), from an un-reduced input (b <= 1.05: a canonical residue, or a digit
// that is a residue of a prime at most 1/64 larger) five.  Round 2 reduced after every register round of four stages and
// the input of a decomposing launch as well -- five times per 16 stages; three suffice.  The schedule is computed at
// compile time from the bound: `before` bit s = reduce before local stage s, `at_end` = the hand-over bound would be
// exceeded.  Contract between the passes: the column stages hand over centred residues (|x| <= q / 2) -- the row stages
// of the fused key switch recompute their twiddle companions (growth 1.375 b + 0.5: five stages per reduction, not
// six), so a looser hand-over would cost that kernel the reduction the column pass saved.  What the schedule buys:
// the input of a decomposing launch is not reduced (its digit is a residue of a prime of the same size), and the one
// reduction inside the column stages sits after the fifth stage instead of the fourth.
struct FpSched { unsigned before; bool at_end; };
#define FP_STAGE_GROW(b) (1.25 * (b) + 0.5)
#define FP_BOUND_LIMIT 7.9
#define FP_HANDOVER 0.51
#define FP_UNREDUCED_IN 1.05
constexpr FpSched fp_sched(int stages, double b_in, double b_out_max)
{
    unsigned m = 0;
    double b = b_in;
    for (int s = 0; s < stages; s++) {
        if (FP_STAGE_GROW(b) > FP_BOUND_LIMIT) {
            m |= 1u << s;
            b = 0.5;
        }
        b = FP_STAGE_GROW(b);
    }
    return FpSched{m, b > b_out_max};
}
// LOGR CT stages; reductions where the schedule says (`before` bit s: before local stage s; at_end)
template <int LOGR, typename TW>
__device__ __forceinline__ void fp_ct_radix(double (&x)[1 << LOGR], TW tw, u32 root0,
                                            const FC& c, unsigned before, bool at_end)
{
#pragma unroll
    for (int s = 0; s < LOGR; s++) {
        FP_STAGE(c, 31 - __builtin_clz(root0) + s); // (audit build: global stage index; a reduction counts to the stage it precedes)
        if ((before >> s) & 1u) {
#pragma unroll
            for (int k = 0; k < (1 << LOGR); k++) x[k] = fp_reduce(x[k], c);
        }
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = NTT_ABLATE_TW(NTT_FP_TW(tw, (root0 << s) + b, c), root0, s);
#pragma unroll
            for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
    FP_STAGE(c, 31 - __builtin_clz(root0) + LOGR);
    if (at_end) {
#pragma unroll
        for (int k = 0; k < (1 << LOGR); k++) x[k] = fp_reduce(x[k], c);
    }
}
// the column stages of a transform with S1 of them (round A: S1 - 4, round B: 4) and the first four row stages
template <int S1> struct FpColSched {
    static constexpr FpSched s = fp_sched(S1, FP_UNREDUCED_IN, FP_HANDOVER);
    static constexpr unsigned a_before = s.before & ((1u << (S1 - 4)) - 1u);
    static constexpr unsigned b_before = (s.before >> (S1 - 4)) & 15u;
    static constexpr bool b_at_end = s.at_end;
};
// (the row stages: four stages, a reduction, four stages, the canonical reduction -- from |x| <= q / 2 each round
// stays below 4.11 q)
// last four stages of the row pass (re-laid table), result canonical in [0,q)
__device__ __forceinline__ void fp_ct_radix16_tb(double (&x)[16], const ulonglong2* __restrict__ tb, const FC& c)
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        if (s > 0) FP_STAGE(c, (c).stage + 1); // (the caller's first row round left the index of this round's first stage)
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = NTT_ABLATE_TW(NTT_FP_TW(tb, ((1 << s) - 1 + b) * 16, c), (u32) b, s);
#pragma unroll
            for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
    FP_STAGE(c, (c).stage + 1);
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = fp_canon(x[k], c);
}

// the same from the plain-double table (NttArgs::twB8): companion recomputed as w * RN(1/q).  Bound: the values come
// in centred (|x| <= q/2 (1 + 2^-40), the reduction that ends the first row round); with a recomputed companion a stage
// takes b q to at most (1.375 b + 0.5) q (fpmod.cuh / ks_row_mac_fp: the companion's relative error 1.5 * 2^-52 times
// |y w / q| <= b 2^50 adds 0.375 b to the quotient error), so four stages reach 5.22 q < 7.9 q <= 2^53; fp_canon takes that.
__device__ __forceinline__ void fp_ct_radix16_tb8(double (&x)[16], const double* __restrict__ tb, const FC& c)
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        // (a fence per stage: left free, the scheduler requests all fifteen twiddles up front and the 1024-thread single
        // pass -- 128 registers -- spills six of them)
        if (s > 0) __builtin_amdgcn_sched_barrier(0);
        if (s > 0) FP_STAGE(c, (c).stage + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const double wd = tb[((1 << s) - 1 + b) * 16];
            const ulonglong2 w = make_ulonglong2(as_bits(wd), as_bits(wd * c.qi));
#pragma unroll
            for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
    FP_STAGE(c, (c).stage + 1);
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = fp_canon(x[k], c);
}

// Per-modulus constants of the lazy butterflies.
struct QC {
    u64 q;   // modulus
    u64 q4;  // 4q  (lazy range bound, 8q < 2^64 because q < 2^61)
    u32 nq0, nq1; // -q mod 2^64, split
};
__device__ __forceinline__ QC make_qc(u64 q)
{
    QC c;
    c.q = q;
    c.q4 = 4 * q;
    u64 nq = 0 - q;
    c.nq0 = (u32) nq;
    c.nq1 = (u32) (nq >> 32);
    return c;
}

// Shoup multiplication by the constant w.x with companion w.y = floor(w.x*2^64/q).
// The quotient estimate drops only the lo*lo partial product of mulhi64 (3
// multiplies instead of 4), so it can be short by at most 1:
// result = y*w mod q + {0,1,2}q, inside [0,4q) for ANY 64-bit y.  All nine
// 32-bit multiplies are v_mad_u64_u32 / v_mul_lo_u32 (v_mul_hi_u32 issues at
// roughly half their rate on gfx950 -- profiles/r1a_first/ubench_intmul.txt),
// and y*w - qh*q is one multiply-accumulate chain against -q.
__device__ __forceinline__ u64 shoup_lazy(u64 y, ulonglong2 w, const QC& c)
{
    const u32 y0 = (u32) y, y1 = (u32) (y >> 32);
    const u32 w0 = (u32) w.x, w1 = (u32) (w.x >> 32), p0 = (u32) w.y, p1 = (u32) (w.y >> 32);
    const u64 A = (u64) y0 * p1;
    const u64 B = (u64) y1 * p0 + (u32) A;
    const u64 qh = (u64) y1 * p1 + (A >> 32) + (B >> 32);
    const u32 h0 = (u32) qh, h1 = (u32) (qh >> 32);
    u64 acc = (u64) y0 * w0;
    acc += (u64) h0 * c.nq0;
    const u32 hi = y0 * w1 + y1 * w0 + h0 * c.nq1 + h1 * c.nq0;
    return acc + ((u64) hi << 32);
}

// exact product: canonical residue of y*w for any 64-bit y
__device__ __forceinline__ u64 shoup_full(u64 y, ulonglong2 w, const QC& c)
{
    u64 r = shoup_lazy(y, w, c);
    r = csub(r, 2 * c.q);
    return csub(r, c.q);
}

// Harvey-style CT butterfly.  LAZY == false: x,y in [0,8q) -> [0,8q) (one
// conditional subtraction per butterfly; needs only q < 2^61).  LAZY == true
// (q <= NttArgs::lazy_q_max): no correction at all -- every stage adds at most 4q to the
// bound, which stays below 2^64 over all log2 N stages; one exact reduction ends the transform.
template <bool LAZY>
__device__ __forceinline__ void ct_bfly(u64& x, u64& y, ulonglong2 w, const QC& c)
{
    NTT_ABLATE_BFLY(x, y, w);
    u64 u = LAZY ? x : csub(x, c.q4);
    u64 t = shoup_lazy(y, w, c);
    x = u + t;
    y = u + c.q4 - t;
}

// GS butterfly, x,y in [0,4q) -> [0,4q)
__device__ __forceinline__ void gs_bfly(u64& x, u64& y, ulonglong2 w, const QC& c)
{
    NTT_ABLATE_BFLY(x, y, w);
    u64 s = x + y;
    u64 d = x + c.q4 - y;
    x = csub(s, c.q4);
    y = shoup_lazy(d, w, c);
}

// LOGR Cooley-Tukey stages on 2^LOGR register-resident values.  Local stage
// s, block b uses root index (root0 << s) + b.
template <int LOGR, bool LAZY, typename TW>
__device__ __forceinline__ void ct_radix(u64 (&x)[1 << LOGR], TW tw,
                                         u32 root0, const QC& c)
{
#pragma unroll
    for (int s = 0; s < LOGR; s++) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = NTT_ABLATE_TW(tw[(root0 << s) + b], root0, s);
#pragma unroll
            for (int j = 0; j < half; j++)
                ct_bfly<LAZY>(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// The last four CT stages of the row pass with the re-laid table: slot k of
// this thread is tb[k * 16] (tb already points at [mod][row][0][lane]).
template <bool LAZY>
__device__ __forceinline__ void ct_radix16_tb(u64 (&x)[16], const ulonglong2* __restrict__ tb, const QC& c)
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = tb[((1 << s) - 1 + b) * 16];
#pragma unroll
            for (int j = 0; j < half; j++)
                ct_bfly<LAZY>(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// LOGR Gentleman-Sande stages (reverse order of ct_radix).
template <int LOGR>
__device__ __forceinline__ void gs_radix(u64 (&x)[1 << LOGR], const ulonglong2* __restrict__ tw,
                                         u32 root0, const QC& c)
{
#pragma unroll
    for (int s = LOGR - 1; s >= 0; s--) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = NTT_ABLATE_TW(tw[(root0 << s) + b], root0, s);
#pragma unroll
            for (int j = 0; j < half; j++)
                gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// The first four GS stages of the inverse row pass with the re-laid table.
__device__ __forceinline__ void gs_radix16_tb(u64 (&x)[16], const ulonglong2* __restrict__ tb, const QC& c)
{
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        const int half = 8 >> s;
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = tb[((1 << s) - 1 + b) * 16];
#pragma unroll
            for (int j = 0; j < half; j++)
                gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// Same, but the very last stage (s == 0, global stage 0) folds N^-1 in and
// fully reduces: x' = (x+y)*ninv, y' = (x-y)*(w1*ninv).
template <int LOGR>
__device__ __forceinline__ void gs_radix_last(u64 (&x)[1 << LOGR], const ulonglong2* __restrict__ tw,
                                              u32 root0, ulonglong2 ninv, ulonglong2 w1ninv, const QC& c)
{
#pragma unroll
    for (int s = LOGR - 1; s >= 1; s--) {
        const int half = (1 << LOGR) >> (s + 1);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            ulonglong2 w = NTT_ABLATE_TW(tw[(root0 << s) + b], root0, s);
#pragma unroll
            for (int j = 0; j < half; j++)
                gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
    const int half = (1 << LOGR) >> 1;
#pragma unroll
    for (int j = 0; j < half; j++) {
        u64 s = x[j] + x[j + half];
        u64 d = x[j] + c.q4 - x[j + half];
        x[j] = shoup_full(s, ninv, c);
        x[j + half] = shoup_full(d, w1ninv, c);
    }
}

struct PolySel {
    u64 in_off, out_off; // element offsets of this polynomial
    int mod;             // modulus index into the plan
    int digit;           // RNS digit of a decomposing launch (else -1)
    int item, j;         // ciphertext of the batch, polynomial index inside it
};

// x / d for x < 2^16 with the host's magic = ceil(2^32 / d) (0: d == 1); scalar-ALU work
__device__ __forceinline__ int udiv16(int x, unsigned magic)
{
    return magic ? (int) __umulhi((unsigned) x, magic) : x;
}

__device__ __forceinline__ PolySel select_poly(const NttArgs& a, int poly)
{
    PolySel s;
    if (a.group_span) {
        // modulus-major walk: grid index = k * span + r  ->  r-th polynomial
        // (in item-major order) among those with modulus slot k
        const int k = udiv16(poly, a.mg_group_span), r = poly - k * a.group_span;
        if (a.polys_per_item) {
            const int per_item = udiv16(a.polys_per_item, a.mg_mod_count); // digits per item
            const int it = udiv16(r, a.mg_per_item), d = r - it * per_item;
            poly = it * a.polys_per_item + d * a.mod_count + k;
        } else {
            poly = r * a.mod_count + k;
        }
    }
    int item = 0, j = poly;
    if (a.polys_per_item) {
        item = udiv16(poly, a.mg_polys_per_item);
        j = poly - item * a.polys_per_item;
    }
    int k = j - udiv16(j, a.mg_mod_count) * a.mod_count;
    if (a.mod_order) k = a.mod_order[k];
    s.mod = a.mod_offset + k;
    u64 slot = a.poly_order ? (u64) a.poly_order[j] : (u64) j;
    s.digit = a.decomp_mods ? udiv16(j, a.mg_decomp_mods) : -1;
    u64 in_slot = a.decomp_mods ? (u64) s.digit * (a.decomp_in_mul ? a.decomp_in_mul : 1) + a.decomp_in_add : slot;
    s.in_off = (u64) item * a.in_item_stride + (in_slot << a.n_power);
    s.out_off = (u64) item * a.out_item_stride + (slot << a.n_power);
    s.item = item;
    s.j = j;
    return s;
}

// LDS tiles are exactly 4096 elements (32 KiB -> 5 workgroups per CU); bank
// conflicts are removed by XOR swizzles instead of padding.
// column tile: rows 16 apart (256 elements) would share banks in the 16-column
// tile -> flip the 128-byte half on odd 256-element blocks.
__device__ __forceinline__ int col_phys(int e) { return e ^ (((e >> 8) & 1) << 4); }
// row tile: a thread's 16 contiguous elements are read as 16-byte pairs at a
// 128-byte lane stride -> XOR the pair index (bits 3:1) with bits 7:5 so the
// 16 lanes of a b128 group hit 16 distinct 16-byte slots.
__device__ __forceinline__ int row_phys(int e) { return e ^ (((e >> 5) & 7) << 1); }

// The row-pass exchanges stay inside the 16 lanes that own one row, i.e. inside
// one wavefront, and a wave's LDS operations execute in order: only the
// compiler has to be kept from reordering them, no s_barrier is needed.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define COL_LDS_ELEMS 4096
#define ROW_LDS_ELEMS 4096

// ------------------------------------------------------------------ forward
// Moduli up to NttArgs::lazy_q_max take the correction-free butterflies in every stage (see ct_bfly): the bound of a
// value grows by at most 4q per stage, in + 4 log2(N) q < 2^64 (2^57 for N = 2^16 next to 61-bit primes; the 58/59-bit
// default chains at N <= 2^15).
__device__ __forceinline__ bool fwd_stages_lazy(const Mod& md, u64 lazy_q_max) { return md.q <= lazy_q_max; }
// The ROW stages alone can run correction-free for somewhat larger moduli: the column pass of such a
// modulus (conditional subtraction per butterfly) hands over values below 8q, eight correction-free
// stages add at most 4q each, and 40q < 2^64 holds up to q = floor(2^64 / 40) (2^58.67: the 58- and
// "59"-bit primes of the default chains, which sit just above 2^58).  The one exact reduction at the end
// (or the 128-bit inner product of ks_row_mac: 64 digits * 40q * q < 2^128) takes any such value.
#define NTT_ROW_LAZY_MAX_Q 0x0666666666666666ull
__device__ __forceinline__ bool row_stages_lazy(const Mod& md, u64 lazy_q_max) { return md.q <= lazy_q_max || md.q <= NTT_ROW_LAZY_MAX_Q; }

// Column pass: stages 0..S1-1 (row stride 256).  grid = (256/CT, batch).
// SREG: the 16 source coefficients of the thread are already in registers (`sreg`, in load order:
// group g, slot k at sreg[g * RA + k]) -- the multi-modulus column pass (ntt_fwd_col_multi) loads a
// digit tile once and runs this body for every target modulus; it also needs the tile free again
// before the next iteration writes it, hence the second barrier right after the exchange reads.
template <int S1, bool DECOMP, bool LAZY, bool SREG = false>
__device__ __forceinline__ void fwd_col_body(const NttArgs& a, const PolySel& ps, const Mod& md, u64* lds,
                                             const u64* sreg = nullptr)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    const int t = threadIdx.x;
    const QC qc = make_qc(md.q);
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const u64* __restrict__ src = a.in + ps.in_off + blockIdx.x * CT;
    u64* __restrict__ dst = a.out + ps.out_off + blockIdx.x * CT;

    // half_on (mod-down stage one as a load transform): exact residue, the butterflies' input
    // range does not cover v + q - half_mod for a 60-bit source modulus
    const u64 half_qP = a.half_on ? a.mods[a.half_src_mod].q : 0;
    const u64 half_hm = a.half_on ? a.half_mod[ps.mod] : 0;
    // all 16 loads in one block, then (uniform condition, one branch) the mod-down load transform: a branch per
    // element would serialise the loads -- one memory latency each
    u64 x[16];
    const int col = t % CT, r1 = t / CT;
    u64 v[16];
    if constexpr (NSA > 0) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
#pragma unroll
            for (int k = 0; k < RA; k++) v[g * RA + k] = SREG ? sreg[g * RA + k] : gld(&src[(u64) (rb + 16 * k) * 256 + c]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = SREG ? sreg[k] : gld(&src[(u64) k * 256 + col]);
    }
    if (DECOMP && a.half_on) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = sub_mod(reduce64(add_mod(v[k], a.half, half_qP), md), half_hm, md.q);
    } else if (DECOMP && !LAZY && a.mods[ps.digit].q > 8 * md.q) {
        // a digit of a much wider prime (61 bits next to 58): outside the [0, 8q) the correcting butterflies keep
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = reduce64(v[k], md);
    }
    if constexpr (NSA > 0) {
        // round A: G groups of radix RA, rows rbase + 16k
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
            u64 y[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = v[g * RA + k];
            // DECOMP: the digit (a residue of another prime of the plan) is NOT reduced
            // first: the correcting butterflies only need x < 8q (guarded above) and the
            // correction-free path x + 4 log2(N) q < 2^64 (lazy_q_max); congruence mod q is
            // kept and the row pass ends with an exact reduction.
            ct_radix<NSA, LAZY>(y, tw, 1u, qc);
#pragma unroll
            for (int k = 0; k < RA; k++) lds[col_phys((rb + 16 * k) * CT + c)] = y[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = lds[col_phys((16 * r1 + k) * CT + col)];
        if (SREG) __syncthreads();
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = v[k];
    }
    ct_radix<4, LAZY>(x, tw, (u32) (RA + r1), qc);
#pragma unroll
    for (int k = 0; k < 16; k++) gst(&dst[(u64) (16 * r1 + k) * 256 + col], x[k]);
}

// FP64 column pass (Mod::fp).  Output: centred residues as raw doubles -- the
// row pass of the same modulus consumes them as such.
template <int S1, bool DECOMP, bool WIDE, bool SREG = false>
__device__ __forceinline__ void fwd_col_body_fp(const NttArgs& a, const PolySel& ps, const Mod& md, u64* lds,
                                                ulonglong2* twl, const u64* sreg = nullptr)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    // (Inside the modulus loop of ntt_fwd_col_multi the compiler hoists the LDS / store offsets out of the loop:
    // ~50 address registers, 154 VGPRs and three waves per SIMD at S1 = 8 -- measured 3 % faster than
    // recomputing them per iteration at four waves.)
    const int t = threadIdx.x;
    const FC fc = make_fc(md.q, FP_SITE(DECOMP ? FPS_FWD_COL_DECOMP : FPS_FWD_COL, S1 - 4));
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const u64* __restrict__ src = a.in + ps.in_off + blockIdx.x * CT;
    u64* __restrict__ dst = a.out + ps.out_off + blockIdx.x * CT;

    // DECOMP: the input is a residue of the digit's own prime.  Up to 52 bits it
    // converts exactly and is reduced in floating point; a wider digit goes
    // through its halves: v = vh*2^32 + vl == vh*(2^32 mod q) + vl (mod q).
    // (WIDE is uniform per workgroup; the caller branches once.)
    double c32 = 0.0, c32i = 0.0;
    if constexpr (DECOMP && WIDE) {
        c32 = (double) reduce64(1ull << 32, md);
        c32i = c32 * fc.qi;
    }
    const u64 half_qP = (DECOMP && a.half_on) ? a.mods[a.half_src_mod].q : 0;
    const double half_hm = (DECOMP && a.half_on) ? fp_from_u64(a.half_mod[ps.mod]) : 0.0;
    // plain decomposition of a digit whose prime is at most 1/64 above this modulus: no input reduction (FpSched)
    // (not at S1 = 7: the second form of the load loops costs ntt_fwd_col_multi<7> sixteen registers and with them its
    // fourth wave per SIMD, which the integer limbs of the same kernel use -- tests/test_kernel_budgets.py)
    const bool in_small = S1 != 7 && DECOMP && !WIDE && !a.half_on && a.mods[ps.digit].q <= md.q + (md.q >> 6);
    typedef FpColSched<S1> CS;
    // SREG: `sreg` holds the thread's 16 source coefficients, the mod-down half already added: for a
    // source modulus of at most 52 bits as the bits of the converted double, for a wider one as u64
    // CNT source values -> reduced doubles.  The loads first, then (uniform conditions, one branch each around a
    // whole loop) the mod-down "+ half" / "- half mod q_j": a branch per element would split the loads into one
    // basic block -- one memory latency -- each.
    auto load_all = [&](auto& y, auto addr, int si0) {
        constexpr int CNT = sizeof(y) / sizeof(y[0]);
        FP_STAGE(fc, FP_STAGE_INPUT);
        if constexpr (SREG && !WIDE) {
            if (in_small) { // (uniform) the digit's own prime is at most 1/64 larger than this modulus: |y| <= 1.05 q
#pragma unroll
                for (int k = 0; k < CNT; k++) y[k] = as_f64(sreg[si0 + k]);
            } else {
#pragma unroll
                for (int k = 0; k < CNT; k++) y[k] = fp_reduce(as_f64(sreg[si0 + k]), fc);
            }
        } else {
            u64 v[CNT];
#pragma unroll
            for (int k = 0; k < CNT; k++) {
                v[k] = SREG ? sreg[si0 + k] : *addr(k);
                // keep the split conversion of a wide source inside the iteration (hoisted out of the modulus
                // loop it would pin 64 more registers per lane for all of it)
                if constexpr (SREG) asm volatile("" : "+v"(v[k]));
            }
            if (!SREG && DECOMP && a.half_on) {
#pragma unroll
                for (int k = 0; k < CNT; k++) v[k] = add_mod(v[k], a.half, half_qP);
            }
#pragma unroll
            for (int k = 0; k < CNT; k++) {
                if constexpr (DECOMP && WIDE) y[k] = fp_mul(fp_from_u32((u32) (v[k] >> 32)), c32, c32i, fc) + fp_from_u32((u32) v[k]);
                else if constexpr (DECOMP) y[k] = in_small ? fp_from_u64(v[k]) : fp_reduce(fp_from_u64(v[k]), fc);
                else y[k] = fp_from_u64(v[k]);
            }
        }
        if (DECOMP && a.half_on) {
#pragma unroll
            for (int k = 0; k < CNT; k++) y[k] = fp_reduce(y[k] - half_hm, fc);
        }
#pragma unroll
        for (int k = 0; k < CNT; k++) FP_AUDIT_VAL(fc, FPM_SUM, y[k]); // what enters the first stage
    };

    double x[16];
    const int col = t % CT, r1 = t / CT;
    if constexpr (NSA > 0) {
        // The second register round needs tw[((RA + r1) << s) + b], s < 4: 15 entries per lane, all inside
        // tw[RA .. 16 RA).  One coalesced 16-byte load per thread, issued with the coefficients and handed
        // over through LDS at the barrier the exchange needs anyway, replaces 15 dependent L2 round trips.
        // (indices stay below 16 RA <= 256 for every S1)
        const ulonglong2 mine = tw[t];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
            double y[RA];
            load_all(y, [&](int k) { return &src[(u64) (rb + 16 * k) * 256 + c]; }, g * RA);
            // (SREG = inside the modulus loop of ntt_fwd_col_multi, a loop with stores in it: the wave-uniform twiddles
            // of this round through the constant address space, or they come as vector loads -- see ConstTw)
            if constexpr (SREG) fp_ct_radix<NSA>(y, const_tw(tw), 1u, fc, CS::a_before, false);
            else fp_ct_radix<NSA>(y, tw, 1u, fc, CS::a_before, false);
#pragma unroll
            for (int k = 0; k < RA; k++) lds[col_phys((rb + 16 * k) * CT + c)] = as_bits(y[k]);
        }
        twl[t] = mine;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = as_f64(lds[col_phys((16 * r1 + k) * CT + col)]);
        if (SREG) __syncthreads();
        fp_ct_radix<4>(x, twl, (u32) (RA + r1), fc, CS::b_before, CS::b_at_end);
    } else {
        load_all(x, [&](int k) { return &src[(u64) k * 256 + col]; }, 0);
        fp_ct_radix<4>(x, tw, (u32) (RA + r1), fc, CS::b_before, CS::b_at_end);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        // Round 6: the multi-modulus kernel (SREG) only runs on launches of >= 2048 source tiles -- gigabytes of digits that
        // the row pass reads back from HBM long after: written with the non-temporal hint they do not push the twiddles and
        // the source tiles out of L2 (C4 step, same-box A/B: 8.11 -> 8.00 ms; profiles/r6_experiments/README.md)
        if constexpr (SREG) __builtin_nontemporal_store(as_bits(x[k]), &dst[(u64) (16 * r1 + k) * 256 + col]);
        else dst[(u64) (16 * r1 + k) * 256 + col] = as_bits(x[k]);
    }
}

template <int S1, bool DECOMP>
__global__ __launch_bounds__(NTT_THREADS) void ntt_fwd_col(NttArgs a)
{
    __shared__ u64 lds[(S1 > 4) ? COL_LDS_ELEMS : 1];
    __shared__ ulonglong2 twl[(S1 > 4) ? 256 : 1]; // second-round twiddles of the FP64 path
    int poly = blockIdx.y;
    if (DECOMP && a.only_int && a.int_slot_count > 0) {
        // compact grid over (item, digit, integer slot)
        const int cnt = a.int_slot_count, digits = a.polys_per_item / a.decomp_mods;
        const int per_item = digits * cnt;
        const int item = poly / per_item, r = poly - item * per_item;
        const int digit = r / cnt, w = r - digit * cnt;
        poly = item * a.polys_per_item + digit * a.decomp_mods + a.int_slots[w];
    }
    const PolySel ps = select_poly(a, poly);
    if (DECOMP && a.skip_identity && ps.mod == ps.digit) return;
    const Mod md = a.mods[ps.mod];
    if (DECOMP && a.only_int && md.fp) return; // done by ntt_fwd_col_multi
    if (DECOMP && a.copy_src) {
        // NttArgs::copy_src: this workgroup's column tile of limb (digit, slot), copied along
        constexpr int CT = 4096 >> S1;
        const int t = threadIdx.x, col = t % CT, r1 = t / CT;
        const u64 limb = ((u64) (ps.digit * a.copy_part_limbs + (ps.j - ps.digit * a.decomp_mods)) << a.n_power) +
                         blockIdx.x * CT;
        const u64* __restrict__ cs = a.copy_src + (u64) ps.item * a.copy_src_item_stride + limb;
        u64* __restrict__ cd = a.copy_dst + (u64) ps.item * a.copy_dst_item_stride + limb;
        u64 c[16];
#pragma unroll
        for (int k = 0; k < 16; k++) c[k] = cs[(u64) (16 * r1 + k) * 256 + col];
#pragma unroll
        for (int k = 0; k < 16; k++) cd[(u64) (16 * r1 + k) * 256 + col] = c[k];
    }
    if (md.fp) {
        if (DECOMP && a.mods[a.half_on ? a.half_src_mod : ps.digit].bit > 52)
            fwd_col_body_fp<S1, DECOMP, true>(a, ps, md, lds, twl);
        else fwd_col_body_fp<S1, DECOMP, false>(a, ps, md, lds, twl);
    } else if (fwd_stages_lazy(md, a.lazy_q_max)) fwd_col_body<S1, DECOMP, true>(a, ps, md, lds);
    else fwd_col_body<S1, DECOMP, false>(a, ps, md, lds);
}

// Store of the forward row pass: plain, or through the mod-down epilogue (NttEpilogue).  The thread's 16
// canonical results sit in the row tile `lds` at row_phys(row * 256 + i0 + 16 k); e0 = element offset of k = 0
// inside the limb.  The epilogue's operands (16 accumulator values, 16 of the added ciphertext) are requested
// eight at a time before anything is computed: one memory latency per half instead of one per element.
__device__ __forceinline__ void row_store_all(const NttArgs& a, const PolySel& ps, const Mod& md, u64* __restrict__ p,
                                              u64 e0, const u64* lds, int row, int i0)
{
    if (!a.epi.on) {
#pragma unroll
        for (int k = 0; k < 16; k++) gst(&p[e0 + 16 * k], lds[row_phys(row * 256 + i0 + 16 * k)]);
        return;
    }
    const NttEpilogue& ep = a.epi;
    const int part = udiv16(ps.j, ep.mg_limbs), limb = ps.j - part * ep.limbs;
    const u64* __restrict__ ks = ep.ks + ep.ks_item_stride * ps.item + ((u64) (part * ep.ks_part_limbs + limb) << a.n_power) + e0;
    const u64 off = ((u64) (part * ep.limbs + limb) << a.n_power);
    const bool with_ct = ep.ct && (!ep.ct_parts || part < ep.ct_parts);
    const u64* ct = with_ct ? ep.ct + ep.ct_item_stride * ps.item + off + e0 : nullptr; // may alias out: no __restrict__
    u64* out = ep.out + ep.out_item_stride * ps.item + off;
    const u64 inv = ep.inv[ps.mod];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        u64 kv[8], cv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) kv[k] = ks[16 * (8 * h + k)];
        if (with_ct) {
#pragma unroll
            for (int k = 0; k < 8; k++) cv[k] = ct[16 * (8 * h + k)];
        }
        // (uniform conditions around whole loops: a branch per element would split the loads and stores into
        // sixteen basic blocks again)
        u64 r[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            r[k] = mul_barrett(sub_mod(kv[k], lds[row_phys(row * 256 + i0 + 16 * (8 * h + k))], md.q), inv, md);
        if (with_ct) {
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = add_mod(cv[k], r[k], md.q);
        }
        if (ep.galois_inv) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 e = (u32) e0 + 16 * (8 * h + k);
                const u32 ex = ((2u * (__brev(e) >> (32 - a.n_power)) + 1u) * ep.galois_inv) & ((2u << a.n_power) - 1u);
                out[__brev((ex - 1u) >> 1) >> (32 - a.n_power)] = r[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) out[e0 + 16 * (8 * h + k)] = r[k];
        }
    }
}

template <bool LAZY>
__device__ __forceinline__ void fwd_row_body(const NttArgs& a, const PolySel& ps, const Mod& md, u64* lds)
{
    const int t = threadIdx.x;
    const QC qc = make_qc(md.q);
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    u64* __restrict__ p = a.out + ps.out_off + (u64) blockIdx.x * 4096;

    const int row = t >> 4, i0 = t & 15;
    const u32 crow = blockIdx.x * 16 + row; // global row index
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = gld(&p[row * 256 + i0 + 16 * k]);
    ct_radix<4, LAZY>(x, tw, (1u << s1) + crow, qc);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
        x[2 * k] = v.x;
        x[2 * k + 1] = v.y;
    }
    ct_radix16_tb<LAZY>(x, a.twB + ((u64) ps.mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0), qc);
    if (LAZY) {
        // x < 65q: one exact reduction (floor(2^64/q) quotient estimate)
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = reduce64(x[k], md);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = csub(csub(csub(x[k], qc.q4), 2 * qc.q), qc.q);
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) =
            make_ulonglong2(x[2 * k], x[2 * k + 1]);
    wave_lds_fence();
    row_store_all(a, ps, md, a.out + ps.out_off, (u64) blockIdx.x * 4096 + row * 256 + i0, lds, row, i0);
}

// FP64 row pass: raw doubles in (column pass output), canonical u64 out.
__device__ __forceinline__ void fwd_row_body_fp(const NttArgs& a, const PolySel& ps, const Mod& md, u64* lds)
{
    const int t = threadIdx.x;
    const FC fc = make_fc(md.q, FP_SITE(FPS_FWD_ROW, a.n_power - 12));
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    u64* __restrict__ p = a.out + ps.out_off + (u64) blockIdx.x * 4096;

    const int row = t >> 4, i0 = t & 15;
    const u32 crow = blockIdx.x * 16 + row;
    double x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = as_f64(p[row * 256 + i0 + 16 * k]);
    fp_ct_radix<4>(x, tw, (1u << s1) + crow, fc, 0u, true);
#pragma unroll
    for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = as_bits(x[k]);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
        x[2 * k] = as_f64(v.x);
        x[2 * k + 1] = as_f64(v.y);
    }
    fp_ct_radix16_tb8(x, a.twB8 + ((u64) ps.mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0), fc);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) =
            make_ulonglong2(fp_to_u64(x[2 * k]), fp_to_u64(x[2 * k + 1]));
    wave_lds_fence();
    row_store_all(a, ps, md, a.out + ps.out_off, (u64) blockIdx.x * 4096 + row * 256 + i0, lds, row, i0);
}

__global__ __launch_bounds__(NTT_THREADS) void ntt_fwd_row(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    const PolySel ps = select_poly(a, blockIdx.y);
    if (a.skip_identity && ps.mod == ps.digit) return;
    const Mod md = a.mods[ps.mod];
    if (md.fp) fwd_row_body_fp(a, ps, md, lds);
    else if (row_stages_lazy(md, a.lazy_q_max)) fwd_row_body<true>(a, ps, md, lds);
    else fwd_row_body<false>(a, ps, md, lds);
}

// ------------------------------------------------------------------ single pass, N <= 2^14
// A limb of N <= 2^14 coefficients (128 KiB) fits the 160 KiB of LDS: one workgroup of N / 16 threads runs
// all log2 N stages with the limb resident -- 2W of HBM traffic per limb instead of the 4W of the two passes.
// The structure is the two passes back to back: thread group g (256 threads) does the column stages of column
// tile g exactly as ntt_fwd_col does, but leaves its result in LDS, at the position the row stages of row tile
// T = row / 16 read it from: limb[T][row_phys((row % 16) * 256 + col)] -- the row pass's own tile layout, so
// the second half is fwd_row_body reading LDS instead of HBM.  The A -> B exchange of the column stages goes
// through the column tile's own positions (private to the group; a thread reads and writes the same 16
// positions in round B, so no barrier separates them).  Two __syncthreads per transform; every column-stage
// twiddle is wave-uniform (r1 = thread / CT with CT >= 64) and comes through the scalar cache.
template <int S1>
__device__ __forceinline__ int single_pos(int row, int col)
{
    return (row >> 4) * 4096 + row_phys((row & 15) * 256 + col);
}

// DECOMP (NttArgs::decomp_mods, opted in by the caller with single_decomp_ok): the source limb is a digit -- a
// residue of another prime of the plan, optionally through the mod-down "+ half" / "- half mod q" -- and is
// converted as the two-pass column bodies do (fwd_col_body / fwd_col_body_fp): all 16 loads of a thread first, then
// uniform conditions around whole loops.
template <int S1, bool FP, bool LAZY, bool DECOMP = false>
__device__ __forceinline__ void fwd_single_body(const NttArgs& a, const PolySel& ps, const Mod& md, u64* limb)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    const int t = threadIdx.x, g = t >> 8, tt = t & 255;
    const QC qc = make_qc(md.q);
    const FC fc = make_fc(md.q, FP_SITE(FPS_FWD_SINGLE, S1 - 4));
    const ulonglong2* __restrict__ tw = a.tw + ((u64) ps.mod << a.n_power);
    const u64* __restrict__ src = a.in + ps.in_off + g * CT;
    const int col = tt % CT, r1 = tt / CT;
    // DECOMP: the thread's 16 source coefficients in load order (group gi, slot k at [gi * RA + k]; NSA == 0: row k)
    u64 dv[DECOMP ? 16 : 1];
    double dy[(DECOMP && FP) ? 16 : 1];
    if constexpr (DECOMP) {
        if constexpr (NSA > 0) {
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                const int L = tt + 256 * gi;
                const int c = L % CT, rb = L / CT;
#pragma unroll
                for (int k = 0; k < RA; k++) dv[gi * RA + k] = src[(u64) (rb + 16 * k) * 256 + c];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) dv[k] = src[(u64) k * 256 + col];
        }
        const u64 half_qP = a.half_on ? a.mods[a.half_src_mod].q : 0;
        if constexpr (FP) {
            FP_STAGE(fc, FP_STAGE_INPUT);
            if (a.half_on) {
#pragma unroll
                for (int k = 0; k < 16; k++) dv[k] = add_mod(dv[k], a.half, half_qP);
            }
            if (a.mods[a.half_on ? a.half_src_mod : ps.digit].bit > 52) { // v = vh 2^32 + vl == vh (2^32 mod q) + vl
                const double c32 = (double) reduce64(1ull << 32, md), c32i = c32 * fc.qi;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    dy[k] = fp_mul(fp_from_u32((u32) (dv[k] >> 32)), c32, c32i, fc) + fp_from_u32((u32) dv[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) dy[k] = fp_reduce(fp_from_u64(dv[k]), fc);
            }
            if (a.half_on) {
                const double half_hm = fp_from_u64(a.half_mod[ps.mod]);
#pragma unroll
                for (int k = 0; k < 16; k++) dy[k] = fp_reduce(dy[k] - half_hm, fc);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) FP_AUDIT_VAL(fc, FPM_SUM, dy[k]);
        } else {
            if (a.half_on) {
                const u64 half_hm = a.half_mod[ps.mod];
#pragma unroll
                for (int k = 0; k < 16; k++) dv[k] = sub_mod(reduce64(add_mod(dv[k], a.half, half_qP), md), half_hm, md.q);
            } else if (!LAZY && a.mods[ps.digit].q > 8 * md.q) {
#pragma unroll
                for (int k = 0; k < 16; k++) dv[k] = reduce64(dv[k], md);
            }
        }
    }
    // ---- column stages 0 .. S1-1
    if constexpr (FP) {
        double x[16];
        if constexpr (NSA > 0) {
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                const int L = tt + 256 * gi;
                const int c = L % CT, rb = L / CT;
                double y[RA];
#pragma unroll
                for (int k = 0; k < RA; k++) {
                    if constexpr (DECOMP) y[k] = dy[gi * RA + k];
                    else y[k] = fp_from_u64(gld(&src[(u64) (rb + 16 * k) * 256 + c]));
                }
                fp_ct_radix<NSA>(y, tw, 1u, fc, FpColSched<S1>::a_before, false);
#pragma unroll
                for (int k = 0; k < RA; k++) limb[single_pos<S1>(rb + 16 * k, g * CT + c)] = as_bits(y[k]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = as_f64(limb[single_pos<S1>(16 * r1 + k, g * CT + col)]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if constexpr (DECOMP) x[k] = dy[k];
                else x[k] = fp_from_u64(src[(u64) k * 256 + col]);
            }
        }
        fp_ct_radix<4>(x, tw, (u32) (RA + r1), fc, FpColSched<S1>::b_before, FpColSched<S1>::b_at_end);
#pragma unroll
        for (int k = 0; k < 16; k++) limb[single_pos<S1>(16 * r1 + k, g * CT + col)] = as_bits(x[k]);
    } else {
        u64 x[16];
        if constexpr (NSA > 0) {
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                const int L = tt + 256 * gi;
                const int c = L % CT, rb = L / CT;
                u64 y[RA];
#pragma unroll
                for (int k = 0; k < RA; k++) {
                    if constexpr (DECOMP) y[k] = dv[gi * RA + k];
                    else y[k] = gld(&src[(u64) (rb + 16 * k) * 256 + c]);
                }
                ct_radix<NSA, LAZY>(y, tw, 1u, qc);
#pragma unroll
                for (int k = 0; k < RA; k++) limb[single_pos<S1>(rb + 16 * k, g * CT + c)] = y[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = limb[single_pos<S1>(16 * r1 + k, g * CT + col)];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if constexpr (DECOMP) x[k] = dv[k];
                else x[k] = src[(u64) k * 256 + col];
            }
        }
        ct_radix<4, LAZY>(x, tw, (u32) (RA + r1), qc);
#pragma unroll
        for (int k = 0; k < 16; k++) limb[single_pos<S1>(16 * r1 + k, g * CT + col)] = x[k];
    }
    __syncthreads();
    // ---- row stages S1 .. S1+7 on row tile g (16 rows of 256), wave-local exchanges as in fwd_row_body
    u64* lds = limb + g * 4096;
    const int row = tt >> 4, i0 = tt & 15;
    const u32 crow = g * 16 + row;
    const ulonglong2* __restrict__ tb = a.twB + ((u64) ps.mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0);
    u64 r[16];
    if constexpr (FP) {
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = as_f64(lds[row_phys(row * 256 + i0 + 16 * k)]);
        fp_ct_radix<4>(x, tw, (1u << S1) + crow, fc, 0u, true);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = as_bits(x[k]);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
            x[2 * k] = as_f64(v.x);
            x[2 * k + 1] = as_f64(v.y);
        }
        fp_ct_radix16_tb8(x, a.twB8 + ((u64) ps.mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0), fc);
#pragma unroll
        for (int k = 0; k < 16; k++) r[k] = fp_to_u64(x[k]);
    } else {
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = lds[row_phys(row * 256 + i0 + 16 * k)];
        ct_radix<4, LAZY>(x, tw, (1u << S1) + crow, qc);
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = x[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
            x[2 * k] = v.x;
            x[2 * k + 1] = v.y;
        }
        ct_radix16_tb<LAZY>(x, tb, qc);
        if (LAZY) {
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = reduce64(x[k], md);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = csub(csub(csub(x[k], qc.q4), 2 * qc.q), qc.q);
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) = make_ulonglong2(r[2 * k], r[2 * k + 1]);
    wave_lds_fence();
    row_store_all(a, ps, md, a.out + ps.out_off, (u64) g * 4096 + row * 256 + i0, lds, row, i0);
}

// grid = batch polynomials, N / 16 threads, N * 8 bytes of dynamic LDS
template <int S1, bool DECOMP = false>
__global__ __launch_bounds__(16 << S1, 4) void ntt_fwd_single(NttArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u64 limb[];
    const PolySel ps = select_poly(a, blockIdx.x);
    if (DECOMP && a.skip_identity && ps.mod == ps.digit) return;
    const Mod md = a.mods[ps.mod];
    if (md.fp) fwd_single_body<S1, true, false, DECOMP>(a, ps, md, limb);
    else if (fwd_stages_lazy(md, a.lazy_q_max)) fwd_single_body<S1, false, true, DECOMP>(a, ps, md, limb);
    else fwd_single_body<S1, false, false, DECOMP>(a, ps, md, limb);
}

// ------------------------------------------------------------------ fused row pass + key-switch MAC
__device__ __forceinline__ void acc128(u64& hi, u64& lo, u64 a, u64 b)
{
    u64 h, l;
    mul64wide(a, b, h, l);
    lo += l;
    hi += h + (lo < l);
}

template <bool LAZY>
__device__ __forceinline__ void ks_row_digit(u64 (&x)[16], const u64* __restrict__ p, u64* lds,
                                             const ulonglong2* twa, const ulonglong2* __restrict__ tb,
                                             const QC& qc, const Mod& md, int row, int i0)
{
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = gld(&p[row * 256 + i0 + 16 * k]);
    // first four stages: the row's 15 twiddles from LDS (slot (1 << s) - 1 + b at [slot][row])
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = twa[((1 << s) - 1 + b) * 16 + row];
#pragma unroll
            for (int j = 0; j < half; j++) ct_bfly<LAZY>(x[b * 2 * half + j], x[b * 2 * half + j + half], w, qc);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = x[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
        x[2 * k] = v.x;
        x[2 * k + 1] = v.y;
    }
    ct_radix16_tb<LAZY>(x, tb, qc);
    if (!LAZY) {
        // bring [0,8q) down to [0,q) so that 64 products stay below 2^128
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = csub(csub(csub(x[k], qc.q4), 2 * qc.q), qc.q);
    }
    // LAZY: the transform output stays un-reduced (any 64-bit value); the caller checked that
    // digits * 2^64 * q fits the 128-bit accumulator, the one exact reduction happens on the sum.
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) =
            make_ulonglong2(x[2 * k], x[2 * k + 1]);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = lds[row_phys(row * 256 + i0 + 16 * k)];
    wave_lds_fence();
}

// One-dimensional grid, group = (limb slot, tile).  The `items` workgroups of a group share the key tiles of that
// group (1 MiB at 16 digits) and nothing else is re-used, so a group should run on ONE XCD (workgroup b runs on XCD
// b % 8) -- one L2 instead of eight fetches every key tile -- AND every XCD should get the same number of
// workgroups.  Round 2 pinned group g to XCD g % 8, which is balanced for C4's 240 groups but not for the 4, 14 or 28
// groups of N <= 2^14 (round 3: two integer moduli on 2 of 8 XCDs).  Now the workgroups, in group-major order, are cut
// into eight equal contiguous chunks, one per XCD: a group straddles at most two XCDs, the load is even to within one
// workgroup.  grid = 8 * ceil(T / 8), T = slots * tiles * units.
// KIND: which limb slots the grid runs over -- 0 all (split launches, or the caller does not know the kinds: the
// kernels exit on the other kind's moduli), 1 / 2 the integer / the FP64 ones of KsMacArgs::int_slots (ascending).
struct KsIdx { int item, tile, slot; bool valid; int d0, d1; };
__host__ __device__ __forceinline__ unsigned ks_slots(const KsMacArgs& a, int kind)
{
    if (a.int_slot_count <= 0 || kind == 0) return (unsigned) a.rc;
    return kind == 1 ? (unsigned) a.int_slot_count : (unsigned) (a.rc - a.int_slot_count);
}
template <bool SPLIT, int KIND = 0>
__device__ __forceinline__ KsIdx ks_index(const KsMacArgs& a)
{
    const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    // with digit splits (item, split) takes the place of the item: splits of one item sit next to each other
    const unsigned splits = SPLIT ? (unsigned) a.splits : 1u;
    const unsigned units = (unsigned) a.items * splits;
    const unsigned tiles = 1u << (a.n_power - 12);
    const unsigned total = ks_slots(a, KIND) * tiles * units, chunk = (total + 7u) >> 3;
    const unsigned lin = xcd * chunk + j;
    const unsigned g = lin / units, unit = lin - g * units;
    const unsigned item = SPLIT ? unit / splits : unit, sp = SPLIT ? unit - item * splits : 0u;
    // groups in tile-major order: a chunk of the split kernel, whose integer slots cost about twice the FP64 ones,
    // then holds every slot in proportion (C4, one ciphertext: two tiles x all 17 slots per XCD)
    const unsigned nslots = ks_slots(a, KIND), gt = g / nslots;
    KsIdx r;
    r.valid = lin < total;
    r.item = (int) item;
    r.tile = (int) gt;
    int si = (int) (g - gt * nslots);
    if (KIND == 1 && a.int_slot_count > 0) {
        si = a.int_slots[r.valid ? si : 0];
    } else if (KIND == 2 && a.int_slot_count > 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) // the si-th slot that is not in the (ascending) list
            if (q < a.int_slot_count && a.int_slots[q] <= si) si++;
    }
    r.slot = si;
    r.d0 = SPLIT ? (int) (sp * (unsigned) a.digits / splits) : 0;
    r.d1 = SPLIT ? (int) ((sp + 1) * (unsigned) a.digits / splits) : a.digits;
    return r;
}

// (bodies shared by the kernels below; `twbuf`: KS_TW_LDS_BYTES of LDS for the digit-invariant twiddles)
#define KS_TW_LDS_BYTES ((15 * 256 + 15 * 16) * 8)
template <bool SPLIT>
__device__ __forceinline__ void ks_row_mac_int_body(const KsMacArgs& a, const KsIdx& ki, const Mod& md, int midx,
                                                    u64* lds, void* twbuf)
{
    const int t = threadIdx.x;
    const int item = ki.item, tile = ki.tile, slot = ki.slot;
    const QC qc = make_qc(md.q);
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) midx << a.n_power);
    const int row = t >> 4, i0 = t & 15;
    const u32 crow = tile * 16 + row;
    const ulonglong2* __restrict__ tb = a.twB + ((u64) midx * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0);
    const u64 n = (u64) 1 << a.n_power;
    const u64* __restrict__ pin = a.in + a.in_item_stride * item + ((u64) slot << a.n_power) + (u64) tile * 4096;
    const u64* __restrict__ pk = a.key + ((u64) midx << a.n_power) + (u64) tile * 4096 + row * 256 + i0;
    const u64 dig_off = (u64) a.rc << a.n_power;
    const u64 key_off1 = (u64) a.key_limbs << a.n_power, key_off2 = (u64) a.key_limbs << (a.n_power + 1);
    // the un-reduced output (any 64-bit value) times a key residue (< q) summed over the digits has to fit 128 bits
    const bool lazy = row_stages_lazy(md, a.lazy_q_max) && md.q <= ~0ull / (u64) a.digits;
    // digit-invariant twiddles of the first four stages, shared by the 16 lanes of a row (see
    // ks_row_mac_fp; the per-lane ones of the last four stages would need 61 KiB as pairs)
    ulonglong2* twa = reinterpret_cast<ulonglong2*>(twbuf); // [15 * 16]
    if (i0 < 15) {
        const int s = (i0 >= 7) ? 3 : (i0 >= 3) ? 2 : (i0 >= 1) ? 1 : 0;
        twa[i0 * 16 + row] = tw[((((u32) 1 << s1) + crow) << s) + (i0 - ((1 << s) - 1))];
    }
    wave_lds_fence();

    u64 h0[16], l0[16], h1[16], l1[16];
#pragma unroll
    for (int k = 0; k < 16; k++) h0[k] = l0[k] = h1[k] = l1[k] = 0;
    for (int i = ki.d0; i < ki.d1; i++) {
        u64 x[16];
        const u64* p = pin + dig_off * i;
        if (a.skip_identity && i == midx) {
            const u64* pi = a.ident + a.ident_item_stride * item + ((u64) i << a.n_power) + (u64) tile * 4096;
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = gld(&pi[row * 256 + i0 + 16 * k]);
        } else if (lazy) {
            ks_row_digit<true>(x, p, lds, twa, tb, qc, md, row, i0);
        } else {
            ks_row_digit<false>(x, p, lds, twa, tb, qc, md, row, i0);
        }
        const u64* k0 = pk + key_off2 * i;
        const u64* k1 = k0 + key_off1;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc128(h0[k], l0[k], x[k], k0[16 * k]);
            acc128(h1[k], l1[k], x[k], k1[16 * k]);
        }
    }
    // split launches: the partial sums over the first two digits of this workgroup's own range (KsMacArgs::splits)
    u64* po = (SPLIT ? const_cast<u64*>(pin) + dig_off * ki.d0 - (u64) tile * 4096
                     : a.out + a.out_item_stride * item + ((u64) slot << a.n_power)) +
              (u64) tile * 4096 + row * 256 + i0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        po[16 * k] = reduce128(h0[k], l0[k], md);
        po[dig_off + 16 * k] = reduce128(h1[k], l1[k], md);
    }
    (void) n;
}

template <bool SPLIT>
__global__ __launch_bounds__(NTT_THREADS, SPLIT ? 2 : 1) void ks_row_mac(KsMacArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    __shared__ ulonglong2 twa[15 * 16];
    const KsIdx ki = ks_index<SPLIT, SPLIT ? 0 : 1>(a);
    if (!ki.valid) return;
    const int midx = a.mod_order ? a.mod_order[ki.slot] : ki.slot;
    const Mod md = a.mods[midx];
    if (md.fp) return; // FP64 moduli are handled by ks_row_mac_fp
    ks_row_mac_int_body<SPLIT>(a, ki, md, midx, lds, twa);
}

// FP64 moduli (Mod::fp): same fused row pass + inner product, but the inner
// product is accumulated in FP64 as well: each digit*key product is reduced by
// fp_mul (|t| <= 2.46 q for the un-reduced digit, |digit| <= 5.22 q: bound at the product), the running sums are
// re-centred after every third digit (|acc| <= 7.88 q < 2^53) and made canonical once at the end.
// Two double accumulators per coefficient instead of two 128-bit integers
// halve the register footprint (3 waves per SIMD instead of 2).
template <bool SPLIT>
__device__ __forceinline__ void ks_row_mac_fp_body(const KsMacArgs& a, const KsIdx& ki, const Mod& md, int midx,
                                                   u64* lds, void* twbuf)
{
    const int t = threadIdx.x;
    const int item = ki.item, tile = ki.tile, slot = ki.slot;
    const FC fc = make_fc(md.q, FP_SITE(SPLIT ? FPS_KS_ROW_SPLIT : FPS_KS_ROW, a.n_power - 12));
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.tw + ((u64) midx << a.n_power);
    const int row = t >> 4, i0 = t & 15;
    const u32 crow = tile * 16 + row;
    const double* __restrict__ tb8 = a.twB8 + ((u64) midx * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0);
    const u64* __restrict__ pin = a.in + a.in_item_stride * item + ((u64) slot << a.n_power) + (u64) tile * 4096;
    const u64* __restrict__ pk = a.key + ((u64) midx << a.n_power) + (u64) tile * 4096 + row * 256 + i0;
    const u64 dig_off = (u64) a.rc << a.n_power;
    const u64 key_off1 = (u64) a.key_limbs << a.n_power, key_off2 = (u64) a.key_limbs << (a.n_power + 1);

    // The 30 twiddles a lane needs do not depend on the digit.  Fetched from the tables inside the
    // loop they cost ~25 exposed L2 latencies per digit (the table slice of a workgroup, 65 KiB, does
    // not fit the 32 KiB L1) and the SIMDs issued only 48 % of the time; 120 more registers are not
    // available.  So they are parked in LDS once per workgroup: the 15 of the last four stages are
    // private to the lane (slot k at [k][t]), the 15 of the first four are shared by the 16 lanes of
    // a row ([k][row], written by lane i0 == k).  Only w is kept; its companion RN(w/q) is
    // recomputed as w * RN(1/q) when the twiddle is used (one multiply per twiddle; a stage then takes |x| <= b q
    // to at most (1.375 b + 0.5) q: 0.5 -> 1.19, 2.13, 3.43, 5.22 q < 2^53 over the four stages of a round --
    // tests/fp_model.py derives and searches these bounds, tests/test_gpu_fp_audit.py measures them on the device).
    double* twl = reinterpret_cast<double*>(twbuf); // [15 * 256 + 15 * 16]
#pragma unroll
    for (int k = 0; k < 15; k++) twl[k * 256 + t] = tb8[k * 16];
    if (i0 < 15) {
        // slot i0 = (1 << s) - 1 + b of local stage s
        const int s = (i0 >= 7) ? 3 : (i0 >= 3) ? 2 : (i0 >= 1) ? 1 : 0;
        const int b = i0 - ((1 << s) - 1);
        twl[15 * 256 + i0 * 16 + row] = as_f64(tw[((((u32) 1 << s1) + crow) << s) + b].x);
    }
    wave_lds_fence(); // every value is read back by the wavefront that wrote it

    double a0[16], a1[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a0[k] = a1[k] = 0.0;
    int since = 0; // digits accumulated since the sums were last re-centred
    for (int i = ki.d0; i < ki.d1; i++) {
        double x[16];
        const u64* p = pin + dig_off * i;
        // The digit first, then the key tile of this digit: vmcnt counts in order, so the wait for the
        // coefficients leaves the 32 key loads in flight while the butterflies run (issued the other way
        // round, the first use of a coefficient would wait for the key as well).  The identity digit is the
        // NTT-domain limb (canonical u64) of the decomposed polynomial itself.
        const bool ident = a.skip_identity && i == midx;
        const u64* px = ident ? a.ident + a.ident_item_stride * item + ((u64) i << a.n_power) + (u64) tile * 4096 : p;
        u64 xr[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            // (large-launch form: the digits are read exactly once -- non-temporal; the key tile next to them is what the
            // ciphertexts of a group share through L2.  With the stores below: 8.16 -> 8.03 ms per C4 step, same-box A/B)
            if constexpr (!SPLIT) xr[k] = __builtin_nontemporal_load(&px[row * 256 + i0 + 16 * k]);
            else xr[k] = px[row * 256 + i0 + 16 * k];
        }
        const u64* k0 = pk + key_off2 * i;
        const u64* k1 = k0 + key_off1;
        u64 kv0[16], kv1[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            kv0[k] = k0[16 * k];
            kv1[k] = k1[16 * k];
        }
        if (ident) {
            FP_STAGE(fc, FP_STAGE_INPUT);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = fp_reduce(fp_from_u64(xr[k]), fc);
        } else {
            if constexpr (SPLIT) { // small launches (ks_row_mac_split shares its registers with the integer body): twiddles read where they are used
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = as_f64(xr[k]);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int half = 8 >> s;
                    FP_STAGE(fc, s1 + s);
#pragma unroll
                    for (int b = 0; b < (1 << s); b++) {
                        const double w = twl[15 * 256 + ((1 << s) - 1 + b) * 16 + row];
                        const ulonglong2 wp = make_ulonglong2(as_bits(w), as_bits(w * fc.qi));
#pragma unroll
                        for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], wp, fc);
                    }
                }
                FP_STAGE(fc, s1 + 4);
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = fp_reduce(x[k], fc);
#pragma unroll
                for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = as_bits(x[k]);
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
                    x[2 * k] = as_f64(v.x);
                    x[2 * k + 1] = as_f64(v.y);
                }
                // last four stages; their output goes into the product un-reduced (|x| <= 5.22 q, see the product)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int half = 8 >> s;
                    FP_STAGE(fc, s1 + 4 + s);
#pragma unroll
                    for (int b = 0; b < (1 << s); b++) {
                        const double wd = twl[((1 << s) - 1 + b) * 256 + t];
                        const ulonglong2 w = make_ulonglong2(as_bits(wd), as_bits(wd * fc.qi));
#pragma unroll
                        for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, fc);
                    }
                }
            } else {
                // Round 5: the twiddles of a stage are REQUESTED ONE STAGE AHEAD of the butterflies that use them (those of the
                // first two stages before the digit's coefficients are waited for) and pinned there with scheduling fences.
                // Left to itself the compiler read each twiddle right in front of its butterflies: ds_read, s_waitcnt
                // lgkmcnt(0), butterflies -- thirty exposed LDS round trips per digit at two waves per SIMD (the same find as in
                // the blind rotate, profiles/r5d_c5/README.md).  All thirty at once do not fit: 236 registers are taken (64
                // sums, 64 prefetched key values, the digit), a stage ahead costs at most 8 more doubles.  The indices are hidden
                // from the optimiser: the twiddles do not depend on the digit and would be hoisted out of the loop for good.
                int row_o = row, t_o = t;
                asm volatile("" : "+v"(row_o), "+v"(t_o));
                const double* twa = twl + 15 * 256 + row_o;
                const double* twb = twl + t_o;
                double wn[8], wc[8];
                wc[0] = twa[0];
                wn[0] = twa[16];
                wn[1] = twa[32];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = as_f64(xr[k]);
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int half = 8 >> s;
                    FP_STAGE(fc, s1 + s);
                    if (s > 0) {
#pragma unroll
                        for (int b = 0; b < (1 << s); b++) wc[b] = wn[b];
                    }
                    if (s < 3) { // next stage's twiddles (the first stage's successor was requested above)
                        if (s > 0) {
#pragma unroll
                            for (int b = 0; b < (2 << s); b++) wn[b] = twa[((2 << s) - 1 + b) * 16];
                        }
                    } else {
                        wn[0] = twb[0]; // first of the last four stages
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < (1 << s); b++) {
                        const double w = wc[b];
                        const ulonglong2 wp = make_ulonglong2(as_bits(w), as_bits(w * fc.qi));
#pragma unroll
                        for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], wp, fc);
                    }
                }
                FP_STAGE(fc, s1 + 4);
#pragma unroll
                for (int k = 0; k < 16; k++) x[k] = fp_reduce(x[k], fc);
#pragma unroll
                for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = as_bits(x[k]);
                wave_lds_fence();
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
                    x[2 * k] = as_f64(v.x);
                    x[2 * k + 1] = as_f64(v.y);
                }
                // last four stages; their output goes into the product un-reduced (|x| <= 5.22 q, see the product)
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int half = 8 >> s;
                    FP_STAGE(fc, s1 + 4 + s);
#pragma unroll
                    for (int b = 0; b < (1 << s); b++) wc[b] = wn[b];
                    if (s < 3) {
#pragma unroll
                        for (int b = 0; b < (2 << s); b++) wn[b] = twb[((2 << s) - 1 + b) * 256];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < (1 << s); b++) {
                        const double wd = wc[b];
                        const ulonglong2 w = make_ulonglong2(as_bits(wd), as_bits(wd * fc.qi));
#pragma unroll
                        for (int j = 0; j < half; j++) fp_ct_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, fc);
                    }
                }
            }
            // (round 5) NO centred reduction here: x goes into the product as the "twiddle" operand, whose magnitude only
            // enters the quotient error -- see the bound at the product below
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 8; k++)
                *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) =
                    make_ulonglong2(as_bits(x[2 * k]), as_bits(x[2 * k + 1]));
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = as_f64(lds[row_phys(row * 256 + i0 + 16 * k)]);
            wave_lds_fence();
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            // x plays the role of the "twiddle" (companion RN(x/q) ~ x * qi), the key (canonical, y < q < 2^50) is the
            // operand.  Round 5: x arrives UN-reduced from the last four stages, |x| <= 5.22 q (0.5 q after the reduction
            // in the middle, then b -> 1.375 b + 0.5 four times: the stages recompute their companions).  Exactness of
            // fp_mul(y, x, xi): xi = RN(x RN(1/q)) and the rounded product y xi are within |y x / q| * 1.5 * 2^-52 <=
            // 5.22 * 2^50 * 1.5 * 2^-52 = 1.96 of y x / q, so |k - y x / q| <= 2.46; k < 2^53 is an exact integer, h - k q
            // = (y x - k q) - l with |l| <= ulp(h) / 2 <= 2^49 is an integer below 2.46 q + 2^49 < 2^52 -- the FMA that
            // forms it and the sum with l are exact, t = y x - k q EXACTLY with |t| <= 2.46 q.  What the larger operand
            // costs is the accumulators' headroom: 0.5 q + 3 * 2.46 q = 7.88 q < 8 q <= 2^53, so they are re-centred after
            // every THIRD digit (round 4: x reduced to q / 2 first -- 48 instructions per digit and thread -- |t| <= 0.75 q,
            // re-centred every fourth: 24; now 0 + 32).
            const double xi = x[k] * fc.qi;
            FP_STAGE(fc, FP_STAGE_PRODUCT);
            a0[k] += fp_mul(fp_from_u64(kv0[k]), x[k], xi, fc);
            a1[k] += fp_mul(fp_from_u64(kv1[k]), x[k], xi, fc);
            FP_STAGE(fc, FP_STAGE_SUMS + since); // sums holding since + 1 products since they were last re-centred
            FP_AUDIT_VAL(fc, FPM_SUM, a0[k]);
            FP_AUDIT_VAL(fc, FPM_SUM, a1[k]);
        }
        if (++since == 3) {
            since = 0;
            FP_STAGE(fc, FP_STAGE_SUMS + 3);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                a0[k] = fp_reduce(a0[k], fc);
                a1[k] = fp_reduce(a1[k], fc);
            }
        }
    }
    u64* po = (SPLIT ? const_cast<u64*>(pin) + dig_off * ki.d0 - (u64) tile * 4096
                     : a.out + a.out_item_stride * item + ((u64) slot << a.n_power)) +
              (u64) tile * 4096 + row * 256 + i0;
    FP_STAGE(fc, FP_STAGE_OUT);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if constexpr (!SPLIT) {
            __builtin_nontemporal_store(fp_to_u64(fp_canon(a0[k], fc)), &po[16 * k]);
            __builtin_nontemporal_store(fp_to_u64(fp_canon(a1[k], fc)), &po[dig_off + 16 * k]);
        } else {
            po[16 * k] = fp_to_u64(fp_canon(a0[k], fc));
            po[dig_off + 16 * k] = fp_to_u64(fp_canon(a1[k], fc));
        }
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(NTT_THREADS) void ks_row_mac_fp(KsMacArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    __shared__ double twl[15 * 256 + 15 * 16];
    const KsIdx ki = ks_index<SPLIT, SPLIT ? 0 : 2>(a);
    if (!ki.valid) return;
    const int midx = a.mod_order ? a.mod_order[ki.slot] : ki.slot;
    const Mod md = a.mods[midx];
    if (!md.fp) return; // integer moduli are handled by ks_row_mac
    ks_row_mac_fp_body<SPLIT>(a, ki, md, midx, lds, twl);
}

// Split launches are small launches (fewer workgroups than the chip holds, or barely more): the integer and the
// FP64 moduli of a chain in ONE grid instead of two half-empty ones one after the other -- at N = 2^16, one
// ciphertext, four pieces the two kernels took 87 + 58 us (the two integer moduli of the chain alone 58: 128
// workgroups on 256 CUs).  Both bodies run two waves per SIMD (250 / 234 registers), so nothing is lost by sharing.
__global__ __launch_bounds__(NTT_THREADS, 2) void ks_row_mac_split(KsMacArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    __shared__ double twl[15 * 256 + 15 * 16];
    const KsIdx ki = ks_index<true>(a);
    if (!ki.valid) return;
    const int midx = a.mod_order ? a.mod_order[ki.slot] : ki.slot;
    const Mod md = a.mods[midx];
    if (md.fp) ks_row_mac_fp_body<true>(a, ki, md, midx, lds, twl);
    else ks_row_mac_int_body<true>(a, ki, md, midx, lds, twl);
}

hipError_t ks_row_mac_launch(const KsMacArgs& a, int items, hipStream_t st)
{
    if (a.digits > 64 || items <= 0) return hipErrorInvalidValue;
    if (a.splits > 1 && a.digits < 2 * a.splits) return hipErrorInvalidValue;
    KsMacArgs k = a;
    k.items = items;
    if (k.int_slot_count < 0 || k.int_slot_count > 8) k.int_slot_count = 0;
    const unsigned tiles = (1u << a.n_power) / 4096, units = (unsigned) items * (a.splits > 1 ? (unsigned) a.splits : 1u);
    auto grid_of = [&](int kind) { return dim3(8u * ((ks_slots(k, kind) * tiles * units + 7u) / 8u)); };
    if (a.splits > 1) {
        hipLaunchKernelGGL(ks_row_mac_split, grid_of(0), dim3(NTT_THREADS), 0, st, k);
    } else {
        // each kernel over the slots of its kind where the caller named them, else over all (the other kind exits)
        if (!a.no_fp && ks_slots(k, 2) > 0)
            hipLaunchKernelGGL(ks_row_mac_fp<false>, grid_of(2), dim3(NTT_THREADS), 0, st, k);
        if (!a.no_int && ks_slots(k, 1) > 0)
            hipLaunchKernelGGL(ks_row_mac<false>, grid_of(1), dim3(NTT_THREADS), 0, st, k);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ inverse
// Gentleman-Sande stages from t = 1 upward: row stages first (t = 1..128 on contiguous rows), then the
// column stages, n^-1 folded into the very last one.  Two arithmetic policies share the bodies:
//   ArInt  lazy Shoup butterflies, values in [0, 4q) between stages (any modulus below 2^61);
//   ArFp   exact FP64 (fpmod.cuh) for moduli below 2^50 (Mod::fp): x' = x + y, y' = fp_mul(x - y, w).  The
//          sums double per stage, so every second stage ends with a centred reduction: from |x| <= q to
//          <= 4q < 2^53 and back to q/2; a product of d = x - y (|d| <= 4q) stays exact because
//          |h - kq| <= 1.5 q + ulp(h)/2 < 2^52 is an integer.  11 FP64 instructions per butterfly against ~25
//          integer ones.  The inverse tables of such a modulus hold (double(w), RN(w/q)) pairs.
__device__ __forceinline__ void fp_gs_bfly(double& x, double& y, ulonglong2 w, const FC& c)
{
    const double s = x + y, d = x - y;
    x = s;
    y = fp_mul(d, as_f64(w.x), as_f64(w.y), c);
    FP_AUDIT_VAL(c, FPM_SUM, s);
}

struct ArInt {
    typedef u64 T;
    QC qc;
    __device__ __forceinline__ explicit ArInt(const Mod& md) : qc(make_qc(md.q)) {}
    __device__ __forceinline__ T from_canon(u64 v) const { return v; }
    __device__ __forceinline__ T from_bits(u64 v) const { return v; }
    __device__ __forceinline__ u64 to_bits(T v) const { return v; }
    template <int LOGR>
    __device__ __forceinline__ void radix(T (&x)[1 << LOGR], const ulonglong2* __restrict__ tw, u32 root0) const
    {
        gs_radix<LOGR>(x, tw, root0, qc);
    }
    __device__ __forceinline__ void radix16_tb(T (&x)[16], const ulonglong2* __restrict__ tb) const { gs_radix16_tb(x, tb, qc); }
    template <int LOGR>
    __device__ __forceinline__ void radix_last(T (&x)[1 << LOGR], const ulonglong2* __restrict__ tw, ulonglong2 ninv,
                                               ulonglong2 w1ninv, u64 (&out)[1 << LOGR]) const
    {
        gs_radix_last<LOGR>(x, tw, 1u, ninv, w1ninv, qc);
#pragma unroll
        for (int k = 0; k < (1 << LOGR); k++) out[k] = x[k];
    }
};

struct ArFp {
    typedef double T;
    FC fc;
    int s1; // log2 N - 8 (only the instrumented test build reads it: stage indices of its table)
    __device__ __forceinline__ explicit ArFp(const Mod& md, int n_power) : fc(make_fc(md.q, FP_SITE(FPS_INV, n_power - 12))), s1(n_power - 8) {}
    __device__ __forceinline__ T from_canon(u64 v) const { return fp_from_u64(v); }
    __device__ __forceinline__ T from_bits(u64 v) const { return as_f64(v); }
    __device__ __forceinline__ u64 to_bits(T v) const { return as_bits(v); }
    template <int N>
    __device__ __forceinline__ void reduce_all(T (&x)[N]) const
    {
#pragma unroll
        for (int k = 0; k < N; k++) x[k] = fp_reduce(x[k], fc);
    }
    // LOGR GS stages (local stage s = LOGR-1 .. 0, block b uses root (root0 << s) + b); inputs |x| <= q
    template <int LOGR>
    __device__ __forceinline__ void radix(T (&x)[1 << LOGR], const ulonglong2* __restrict__ tw, u32 root0) const
    {
        int done = 0;
#pragma unroll
        for (int s = LOGR - 1; s >= 0; s--) {
            const int half = (1 << LOGR) >> (s + 1);
            FP_STAGE(fc, 31 - __builtin_clz(root0) + s); // (the forward stage this one inverts; a reduction counts to the stage it follows)
#pragma unroll
            for (int b = 0; b < (1 << s); b++) {
                const ulonglong2 w = tw[(root0 << s) + b];
#pragma unroll
                for (int j = 0; j < half; j++) fp_gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, fc);
            }
            if ((++done & 1) == 0 || s == 0) reduce_all(x);
        }
    }
    __device__ __forceinline__ void radix16_tb(T (&x)[16], const ulonglong2* __restrict__ tb) const
    {
#pragma unroll
        for (int s = 3; s >= 0; s--) {
            const int half = 8 >> s;
            FP_STAGE(fc, s1 + 4 + s);
#pragma unroll
            for (int b = 0; b < (1 << s); b++) {
                const ulonglong2 w = tb[((1 << s) - 1 + b) * 16];
#pragma unroll
                for (int j = 0; j < half; j++) fp_gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, fc);
            }
            if (s == 2 || s == 0) reduce_all(x);
        }
    }
    // the last LOGR stages of the transform: n^-1 folded into the final one, canonical residues out
    template <int LOGR>
    __device__ __forceinline__ void radix_last(T (&x)[1 << LOGR], const ulonglong2* __restrict__ tw, ulonglong2 ninv,
                                               ulonglong2 w1ninv, u64 (&out)[1 << LOGR]) const
    {
        int done = 0;
#pragma unroll
        for (int s = LOGR - 1; s >= 1; s--) {
            const int half = (1 << LOGR) >> (s + 1);
            FP_STAGE(fc, s);
#pragma unroll
            for (int b = 0; b < (1 << s); b++) {
                const ulonglong2 w = tw[(1u << s) + b];
#pragma unroll
                for (int j = 0; j < half; j++) fp_gs_bfly(x[b * 2 * half + j], x[b * 2 * half + j + half], w, fc);
            }
            if ((++done & 1) == 0) reduce_all(x);
        }
        constexpr int half = (1 << LOGR) >> 1;
        FP_STAGE(fc, 0);
#pragma unroll
        for (int j = 0; j < half; j++) {
            const double sum = x[j] + x[j + half], d = x[j] - x[j + half];
            out[j] = fp_to_u64(fp_canon(fp_mul(sum, as_f64(ninv.x), as_f64(ninv.y), fc), fc));
            out[j + half] = fp_to_u64(fp_canon(fp_mul(d, as_f64(w1ninv.x), as_f64(w1ninv.y), fc), fc));
        }
    }
};

// Row stages of one 16-row tile.  src: 16 x 256 coefficients in global memory (canonical residues); `lds`: the
// tile's 4096-element exchange buffer.  TO_LDS: leave the result in `lds` (row_phys layout, the single pass)
// instead of storing it to dst.  crow0 = global index of the tile's first row.
template <typename AR, bool TO_LDS, bool TENSOR = false>
__device__ __forceinline__ void inv_row_part(const AR& ar, const NttArgs& a, int mod, int tt, u32 crow0,
                                             const u64* __restrict__ src, u64* __restrict__ dst, u64* lds,
                                             int tensor_part = 0)
{
    typedef typename AR::T T;
    const int s1 = a.n_power - 8;
    const ulonglong2* __restrict__ tw = a.itw + ((u64) mod << a.n_power);
    const int row = tt >> 4, i0 = tt & 15;
    const u32 crow = crow0 + row;
    if (TENSOR) {
        // the tile of the tensor product instead of a stored limb (NttArgs::tensor_in): four elements at a time, so
        // that at most sixteen loads are in flight per lane (the kernel runs at 128 registers)
        const Mod md = a.mods[mod];
        const u64 part_off = (u64) a.tensor_limbs << a.n_power;
        const u64* __restrict__ a0 = src, * __restrict__ a1 = src + part_off, * __restrict__ b0 = src + 2 * part_off,
                 * __restrict__ b1 = src + 3 * part_off;
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            u64 r[4];
            if (tensor_part == 1) { // (uniform per workgroup)
                u64 va0[4], va1[4], vb0[4], vb1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = row * 256 + i0 + 16 * (k0 + k);
                    va0[k] = a0[e]; va1[k] = a1[e]; vb0[k] = b0[e]; vb1[k] = b1[e];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) { // a0 b1 + a1 b0: one 128-bit sum, one reduction
                    u64 h1, l1, h2, l2;
                    mul64wide(va0[k], vb1[k], h1, l1);
                    mul64wide(va1[k], vb0[k], h2, l2);
                    const u64 lo = l1 + l2;
                    r[k] = reduce128(h1 + h2 + (lo < l1), lo, md);
                }
            } else {
                const u64* __restrict__ pa = tensor_part == 0 ? a0 : a1;
                const u64* __restrict__ pb = tensor_part == 0 ? b0 : b1;
                u64 va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = row * 256 + i0 + 16 * (k0 + k);
                    va[k] = pa[e]; vb[k] = pb[e];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = mul_barrett(va[k], vb[k], md);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) lds[row_phys(row * 256 + i0 + 16 * (k0 + k))] = r[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = gld(&src[row * 256 + i0 + 16 * k]);
    }
    wave_lds_fence();
    T x[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]);
        x[2 * k] = ar.from_canon(v.x);
        x[2 * k + 1] = ar.from_canon(v.y);
    }
    ar.radix16_tb(x, a.itwB + ((u64) mod * (15u << (a.n_power - 4))) + ((u64) crow * 15 * 16 + i0));
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<ulonglong2*>(&lds[row_phys(row * 256 + 16 * i0 + 2 * k)]) =
            make_ulonglong2(ar.to_bits(x[2 * k]), ar.to_bits(x[2 * k + 1]));
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = ar.from_bits(lds[row_phys(row * 256 + i0 + 16 * k)]);
    ar.template radix<4>(x, tw, (1u << s1) + crow);
    if constexpr (TO_LDS) {
#pragma unroll
        for (int k = 0; k < 16; k++) lds[row_phys(row * 256 + i0 + 16 * k)] = ar.to_bits(x[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) gst(&dst[row * 256 + i0 + 16 * k], ar.to_bits(x[k]));
    }
}

// Where the inverse transform's results go when NttArgs::iepi is on (see NttInvEpilogue): everything the
// epilogue needs for one polynomial, set up once per workgroup.
struct InvEpi {
    const u64* last; // P limb of the part (coefficient domain)
    const u64* ct;   // added limb or nullptr
    u64* out;
    Mod md;
    u64 half, qP, hm, inv;
    int galois, n_power;
    bool pre; // `last` already holds the value to subtract (several special primes, NttInvEpilogue::u)
};
__device__ __forceinline__ bool inv_epi_setup(const NttArgs& a, const PolySel& ps, InvEpi& e, bool& skip)
{
    skip = false;
    if (!a.iepi.on) return false;
    const NttInvEpilogue& ep = a.iepi;
    const int pc = ep.p_count ? ep.p_count : 1;
    const int part = udiv16(ps.j, ep.mg_slots), limb = ps.j - part * (ep.limbs + pc);
    if (limb >= ep.limbs) { skip = true; return true; } // the special limbs: transformed by their own launch
    const u64 item_in = a.out_item_stride * ps.item;
    const u64 off = (u64) (part * ep.limbs + limb) << a.n_power;
    e.pre = ep.u != nullptr;
    e.last = e.pre ? ep.u + ep.u_item_stride * ps.item + off
                   : a.out + item_in + ((u64) (part * (ep.limbs + 1) + ep.limbs) << a.n_power);
    e.ct = (ep.ct && part < ep.add_parts) ? ep.ct + ep.ct_item_stride * ps.item + off : nullptr;
    e.out = ep.out + ep.out_item_stride * ps.item + off;
    e.md = a.mods[ps.mod];
    e.half = ep.half;
    e.qP = a.mods[ep.p_mod].q;
    e.hm = e.pre ? 0 : ep.half_mod[limb];
    e.inv = ep.inv[limb];
    e.galois = ep.galois_elt;
    e.n_power = a.n_power;
    return true;
}
// CNT canonical results o[k] of elements e0 + pos(k) of the limb: plain store to p + pos(k), or the epilogue.  The
// epilogue's operands are requested for all CNT elements before anything is computed (uniform conditions around
// whole loops -- see row_store_all).
template <int CNT, bool EPI, typename POS>
__device__ __forceinline__ void inv_store(const InvEpi& epr, u64* __restrict__ p, u32 e0, POS pos, const u64 (&o)[CNT])
{
    if constexpr (!EPI) {
#pragma unroll
        for (int k = 0; k < CNT; k++) gst(&p[pos(k)], o[k]);
        return;
    }
    const InvEpi* ep = &epr;
    u64 lv[CNT], cv[CNT], r[CNT];
#pragma unroll
    for (int k = 0; k < CNT; k++) lv[k] = ep->last[e0 + pos(k)];
    if (ep->ct) {
#pragma unroll
        for (int k = 0; k < CNT; k++) cv[k] = ep->ct[e0 + pos(k)];
    }
    if (!ep->pre) { // (uniform condition around a whole loop)
#pragma unroll
        for (int k = 0; k < CNT; k++) {
            u64 l = add_mod(lv[k], ep->half, ep->qP);
            l = reduce64(l, ep->md);
            lv[k] = sub_mod(l, ep->hm, ep->md.q);
        }
    }
#pragma unroll
    for (int k = 0; k < CNT; k++) r[k] = mul_barrett(sub_mod(o[k], lv[k], ep->md.q), ep->inv, ep->md);
    if (ep->ct) {
#pragma unroll
        for (int k = 0; k < CNT; k++) r[k] = add_mod(cv[k], r[k], ep->md.q);
    }
    if (ep->galois) {
#pragma unroll
        for (int k = 0; k < CNT; k++) {
            const u32 raw = (e0 + (u32) pos(k)) * (u32) ep->galois;
            // no zero test on the negation: reference switchkey.cu:1694,1711
            ep->out[raw & ((1u << ep->n_power) - 1u)] = ((raw >> ep->n_power) & 1u) ? ep->md.q - r[k] : r[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < CNT; k++) ep->out[e0 + pos(k)] = r[k];
    }
}

// Column stages of one column tile (R rows x CT columns), results to global memory.  FROM_LDS: the row
// stages left their output in the LDS-resident limb (`buf` = the limb, positions single_pos); otherwise the
// tile is read from `p` (global, in place) and `buf` is the tile's 4096-element exchange buffer.
// KEEP: the results also stay in `keep` (slot gi * RA + k, the load order of the forward column stages)
template <int S1, typename AR, bool FROM_LDS, bool KEEP = false, bool EPI = false>
__device__ __forceinline__ void inv_col_part(const AR& ar, const NttArgs& a, int mod, int tt, int g, u64* __restrict__ p,
                                             u64* buf, u64 (&keep)[16], const InvEpi& ep = InvEpi(), u32 e0 = 0)
{
    typedef typename AR::T T;
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    const ulonglong2* __restrict__ tw = a.itw + ((u64) mod << a.n_power);
    const ulonglong2 ninv = a.ninv[mod], w1ninv = a.w1ninv[mod];
    const int col = tt % CT, r1 = tt / CT;
    auto pos = [&](int row, int c) -> int {
        if constexpr (FROM_LDS) return single_pos<S1>(row, g * CT + c);
        else return col_phys(row * CT + c);
    };
    T x[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
        x[k] = ar.from_bits(FROM_LDS ? buf[pos(16 * r1 + k, col)] : gld(&p[(u64) (16 * r1 + k) * 256 + col]));
    if constexpr (NSA > 0) {
        ar.template radix<4>(x, tw, (u32) (RA + r1));
#pragma unroll
        for (int k = 0; k < 16; k++) buf[pos(16 * r1 + k, col)] = ar.to_bits(x[k]);
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            const int L = tt + NTT_THREADS * gi;
            const int c = L % CT, rb = L / CT;
            T y[RA];
            u64 o[RA];
#pragma unroll
            for (int k = 0; k < RA; k++) y[k] = ar.from_bits(buf[pos(rb + 16 * k, c)]);
            ar.template radix_last<NSA>(y, tw, ninv, w1ninv, o);
            if constexpr (RA == 16) {
                // two halves: sixteen elements' worth of epilogue operands would not fit the register budget
                u64 oa[8], ob[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { oa[k] = o[k]; ob[k] = o[8 + k]; }
                inv_store<8, EPI>(ep, p, e0, [&](int k) { return (u32) ((rb + 16 * k) * 256 + c); }, oa);
                inv_store<8, EPI>(ep, p, e0, [&](int k) { return (u32) ((rb + 16 * (8 + k)) * 256 + c); }, ob);
            } else {
                inv_store<RA, EPI>(ep, p, e0, [&](int k) { return (u32) ((rb + 16 * k) * 256 + c); }, o);
            }
            if constexpr (KEEP) {
#pragma unroll
                for (int k = 0; k < RA; k++) keep[gi * RA + k] = o[k];
            }
        }
    } else {
        u64 o[16];
        ar.template radix_last<4>(x, tw, ninv, w1ninv, o);
        {
            u64 oa[8], ob[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { oa[k] = o[k]; ob[k] = o[8 + k]; }
            inv_store<8, EPI>(ep, p, e0, [&](int k) { return (u32) (k * 256 + col); }, oa);
            inv_store<8, EPI>(ep, p, e0, [&](int k) { return (u32) ((8 + k) * 256 + col); }, ob);
        }
        if constexpr (KEEP) {
#pragma unroll
            for (int k = 0; k < 16; k++) keep[k] = o[k];
        }
    }
}

// Row pass: reads a.in, writes a.out (FP64 moduli: centred residues as raw doubles, consumed by ntt_inv_col).
__global__ __launch_bounds__(NTT_THREADS) void ntt_inv_row(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    const PolySel ps = select_poly(a, blockIdx.y);
    if (a.iepi.on && ps.j - udiv16(ps.j, a.iepi.mg_slots) * (a.iepi.limbs + (a.iepi.p_count ? a.iepi.p_count : 1)) >= a.iepi.limbs)
        return; // special limbs: own launch
    const Mod md = a.mods[ps.mod];
    const u64* __restrict__ src = a.in + ps.in_off + (u64) blockIdx.x * 4096;
    u64* __restrict__ dst = a.out + ps.out_off + (u64) blockIdx.x * 4096;
    if (md.fp) inv_row_part<ArFp, false>(ArFp(md, a.n_power), a, ps.mod, threadIdx.x, blockIdx.x * 16, src, dst, lds);
    else inv_row_part<ArInt, false>(ArInt(md), a, ps.mod, threadIdx.x, blockIdx.x * 16, src, dst, lds);
}

// The same with the tensor product as the load transform (NttArgs::tensor_in); a kernel of its own so that the plain
// row pass keeps its register budget.
__device__ __forceinline__ const u64* tensor_src(const NttArgs& a, const PolySel& ps, int& part)
{
    part = udiv16(ps.j, a.mg_tensor_limbs);
    const int limb = ps.j - part * a.tensor_limbs;
    return a.tensor_in + (u64) ps.item * a.tensor_item_stride + ((u64) limb << a.n_power);
}
__global__ __launch_bounds__(NTT_THREADS) void ntt_inv_row_tensor(NttArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 lds[ROW_LDS_ELEMS];
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    int part;
    const u64* __restrict__ src = tensor_src(a, ps, part) + (u64) blockIdx.x * 4096;
    u64* __restrict__ dst = a.out + ps.out_off + (u64) blockIdx.x * 4096;
    if (md.fp) inv_row_part<ArFp, false, true>(ArFp(md, a.n_power), a, ps.mod, threadIdx.x, blockIdx.x * 16, src, dst, lds, part);
    else inv_row_part<ArInt, false, true>(ArInt(md), a, ps.mod, threadIdx.x, blockIdx.x * 16, src, dst, lds, part);
}

// Column pass last, in place on a.out.  grid = (256 / CT, batch).  EPI: with NttArgs::iepi (a kernel of its own:
// the epilogue's state would cost the plain transform a wave per SIMD).
template <int S1, bool EPI = false>
__global__ __launch_bounds__(NTT_THREADS) void ntt_inv_col(NttArgs a)
{
    constexpr int CT = 4096 >> S1;
    __shared__ u64 lds[(S1 > 4) ? COL_LDS_ELEMS : 1];
    const PolySel ps = select_poly(a, blockIdx.y);
    const Mod md = a.mods[ps.mod];
    if (a.only_int && md.fp) return; // a src_inv decomposing launch finishes these limbs itself
    u64* __restrict__ p = a.out + ps.out_off + blockIdx.x * CT;
    u64 unused[16];
    if constexpr (EPI) {
        InvEpi epi;
        bool skip;
        inv_epi_setup(a, ps, epi, skip);
        if (skip) return;
        if (md.fp) inv_col_part<S1, ArFp, false, false, true>(ArFp(md, a.n_power), a, ps.mod, threadIdx.x, 0, p, lds, unused, epi, blockIdx.x * CT);
        else inv_col_part<S1, ArInt, false, false, true>(ArInt(md), a, ps.mod, threadIdx.x, 0, p, lds, unused, epi, blockIdx.x * CT);
    } else {
        if (md.fp) inv_col_part<S1, ArFp, false>(ArFp(md, a.n_power), a, ps.mod, threadIdx.x, 0, p, lds, unused);
        else inv_col_part<S1, ArInt, false>(ArInt(md), a, ps.mod, threadIdx.x, 0, p, lds, unused);
    }
}

// Single pass for N <= 2^14 (see ntt_fwd_single): thread group g runs the row stages of row tile g into the
// LDS-resident limb, then the column stages of column tile g out of it.  grid = batch, N / 16 threads.
template <int S1, bool EPI, bool TENSOR, typename AR>
__device__ __forceinline__ void inv_single_body(const AR& ar, const NttArgs& a, const PolySel& ps, u64* limb)
{
    constexpr int CT = 4096 >> S1;
    const int t = threadIdx.x, g = t >> 8, tt = t & 255;
    InvEpi epi;
    if constexpr (EPI) {
        bool skip;
        inv_epi_setup(a, ps, epi, skip);
        if (skip) return; // (uniform per workgroup: before any barrier)
    }
    if constexpr (TENSOR) {
        int part;
        const u64* src = tensor_src(a, ps, part) + (u64) g * 4096;
        inv_row_part<AR, true, true>(ar, a, ps.mod, tt, g * 16, src, nullptr, limb + g * 4096, part);
    } else {
        inv_row_part<AR, true>(ar, a, ps.mod, tt, g * 16, a.in + ps.in_off + (u64) g * 4096, nullptr, limb + g * 4096);
    }
    __syncthreads();
    u64 unused[16];
    if constexpr (EPI) inv_col_part<S1, AR, true, false, true>(ar, a, ps.mod, tt, g, a.out + ps.out_off + g * CT, limb, unused, epi, g * CT);
    else inv_col_part<S1, AR, true>(ar, a, ps.mod, tt, g, a.out + ps.out_off + g * CT, limb, unused);
}

template <int S1, bool EPI = false, bool TENSOR = false>
__global__ __launch_bounds__(16 << S1, 4) void ntt_inv_single(NttArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u64 limb[];
    const PolySel ps = select_poly(a, blockIdx.x);
    const Mod md = a.mods[ps.mod];
    if (md.fp) inv_single_body<S1, EPI, TENSOR>(ArFp(md, a.n_power), a, ps, limb);
    else inv_single_body<S1, EPI, TENSOR>(ArInt(md), a, ps, limb);
}

// Decomposing column pass, one workgroup per SOURCE tile: the 16 coefficients a thread needs are
// loaded once and stay in registers while the workgroup walks the rc target moduli of that digit
// (reference: cipher_broadcast*_kernel + the first half of GPU_NTT for each of the rc copies,
// switchkey.cu:11-59 / ckks/operator.cu:932-960).  No global load sits on the critical path of an
// iteration, the source limb is read exactly once from HBM, and the stores of iteration j drain while
// iteration j + 1 computes.  With NttArgs::src_inv the workgroup first finishes the inverse transform
// of its source tile (see below).  Two barriers per iteration (tile written -> read -> free again); the
// second-round twiddles of the FP64 body are double-buffered by iteration parity, which the first
// barrier of the following iteration makes safe.  grid = (256 / CT, items * digits).
template <int S1>
__global__ __launch_bounds__(NTT_THREADS) void ntt_fwd_col_multi(NttArgs a)
{
    constexpr int R = 1 << S1;
    constexpr int CT = 4096 / R;
    constexpr int NSA = S1 - 4;
    constexpr int RA = 1 << NSA;
    constexpr int G = 16 / RA;
    __shared__ u64 lds[(S1 > 4) ? COL_LDS_ELEMS : 1];
    __shared__ ulonglong2 twl[(S1 > 4) ? 512 : 1];
    const int t = threadIdx.x;
    const int rc = a.decomp_mods;
    const int digits = udiv16(a.polys_per_item, a.mg_decomp_mods);
    const int item = udiv16(blockIdx.y, a.mg_per_item), digit = blockIdx.y - item * digits;
    const u64 in_slot = (u64) digit * (a.decomp_in_mul ? a.decomp_in_mul : 1) + a.decomp_in_add;
    const u64* __restrict__ src = a.in + (u64) item * a.in_item_stride + (in_slot << a.n_power) + blockIdx.x * CT;
    u64 sreg[16];
    const int smod = a.half_on ? a.half_src_mod : digit;
    if (a.src_inv && a.mods[smod].fp) {
        // The source limb is still half-way through its INVERSE transform (row stages done by ntt_inv_row):
        // the column tile its last stages produce is exactly the tile this workgroup decomposes, in the very
        // register layout (rows rb + 16 k of column c), so they run here -- ntt_inv_col's write of the
        // coefficient-domain limb and the re-read disappear.  The tile is still stored (in place): the
        // integer target moduli are transformed by the per-polynomial kernel, which reads it.  FP64 source
        // moduli only: with the integer inverse inlined as well the modulus loop below lost a wave per SIMD;
        // limbs of integer moduli get their column stages from ntt_inv_col (ntt_launch_inv_rows).
        const Mod im = a.mods[smod];
        inv_col_part<S1, ArFp, false, true>(ArFp(im, a.n_power), a, smod, t, 0, const_cast<u64*>(src), lds, sreg);
        __syncthreads(); // the exchange tile is free again
    } else if constexpr (NSA > 0) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int L = t + NTT_THREADS * g;
            const int c = L % CT, rb = L / CT;
#pragma unroll
            for (int k = 0; k < RA; k++) sreg[g * RA + k] = src[(u64) (rb + 16 * k) * 256 + c];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) sreg[k] = src[(u64) k * 256 + (t % CT)];
    }
    const Mod smd = a.mods[a.half_on ? a.half_src_mod : digit];
    if (a.half_on) {
#pragma unroll
        for (int k = 0; k < 16; k++) sreg[k] = add_mod(sreg[k], a.half, smd.q);
    }
    const bool wide = smd.bit > 52; // uniform per workgroup
    if (!wide) {
#pragma unroll
        for (int k = 0; k < 16; k++) sreg[k] = as_bits(fp_from_u64(sreg[k]));
    }
    int done = 0; // executed iterations (parity of the twiddle buffer)
    for (int k = 0; k < rc; k++) {
        PolySel ps;
        // (launch-constant tables read through the scalar cache: inside this loop plain loads of them are vector loads
        // with an s_waitcnt vmcnt(0) each -- round 5, see ConstTw)
        ps.mod = a.mod_offset + (a.mod_order ? ld_const_i32(a.mod_order + k) : k);
        ps.digit = digit;
        ps.item = item;
        ps.j = digit * rc + k;
        ps.in_off = 0;
        ps.out_off = (u64) item * a.out_item_stride + ((u64) ps.j << a.n_power);
        if (a.skip_identity && ps.mod == digit) continue;
        const Mod md = ld_const_mod(a.mods + ps.mod);
        if (!md.fp) continue; // integer target moduli: ntt_fwd_col<S1, true> with only_int
        ulonglong2* tl = twl + ((S1 > 4) ? 256 * (done & 1) : 0);
        done++;
        if (wide) fwd_col_body_fp<S1, true, true, true>(a, ps, md, lds, tl, sreg);
        else fwd_col_body_fp<S1, true, false, true>(a, ps, md, lds, tl, sreg);
    }
}

// ------------------------------------------------------------------ launch
// The multi-modulus column pass needs enough source tiles to fill the chip on its own (one workgroup
// per source tile instead of one per (source tile, target modulus)); below that the per-polynomial
// kernel runs.  NttArgs::col_multi: 0 never, 1 always, otherwise automatic.
template <int S1>
static bool use_col_multi(const NttArgs& a, int batch)
{
    constexpr int CT = 4096 >> S1;
    if (!a.decomp_mods || a.col_multi == 0 || batch % a.decomp_mods) return false;
    if (a.copy_src) return false; // the copy rides on the per-polynomial kernel only
    if (a.col_multi == 1) return true;
    return (long) (256 / CT) * (batch / a.decomp_mods) >= 2048;
}

// FP64 target moduli through the multi-modulus kernel, the integer ones (if the plan has any: the
// 60-bit q0 and P of the C4 chain) through the per-polynomial kernel, which skips the rest
template <int S1>
static void launch_col_multi(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    if (a.plan_has_fp)
        hipLaunchKernelGGL((ntt_fwd_col_multi<S1>), dim3(256 / CT, batch / a.decomp_mods), dim3(NTT_THREADS), 0, st, a);
    if ((a.plan_has_int || !a.plan_has_fp) && !(a.plan_has_fp && a.int_slot_count < 0)) {
        NttArgs c = a;
        c.group_span = 0;
        c.mg_group_span = 0;
        c.only_int = a.plan_has_fp;
        int polys = batch;
        if (c.only_int && a.int_slot_count > 0) polys = batch / a.decomp_mods * a.int_slot_count;
        else c.int_slot_count = 0;
        hipLaunchKernelGGL((ntt_fwd_col<S1, true>), dim3(256 / CT, polys), dim3(NTT_THREADS), 0, st, c);
    }
}

// One LDS-resident workgroup per limb against the two passes, by launch size (tools/single_pass_sweep.py, us per
// forward transform, single / two passes): N = 2^12 (four workgroups per CU): 7.3 / 10.0 at 9..144 limbs, 10.3 / 14.0
// at 288 -- always; N = 2^13 (two per CU): 10.2 / 10.6 at 36, 10.3 / 11.8 at 72, 11.5 / 16.7 at 144 -- from 64 limbs;
// N = 2^14 (one per CU, 14 stages in a row): 16.1 / 13.1 at 36, 17.0 / 19.7 at 72, 18.8 / 26.6 at 144, but 36.0 / 32.9
// at 288 (256 CUs: a second round for 32 limbs) and 56.0 / 57.6 at 576 -- from 48 limbs, when the last round of
// workgroups is not mostly empty.
static bool use_single_pass(const NttArgs& a, int batch)
{
    if (a.single_pass == 1) return true;
    if (a.single_pass == 0) return false;
    const int s1 = a.n_power - 8;
    if (s1 <= 4) return true;
    if (s1 == 5) return batch >= 64;
    if (batch < 48) return false;
    const long rounds = ((long) batch + 255) / 256;
    return (long) batch * 4 >= rounds * 256 * 3;
}

template <int S1>
static void launch_fwd_col_only(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    if (use_col_multi<S1>(a, batch)) launch_col_multi<S1>(a, batch, st);
    else if (a.decomp_mods)
        hipLaunchKernelGGL((ntt_fwd_col<S1, true>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((ntt_fwd_col<S1, false>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, a);
}

template <int S1>
static void launch_fwd(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    if constexpr (S1 <= 6) {
        if (!a.decomp_mods && use_single_pass(a, batch)) { // LDS-resident single pass (N <= 2^14)
            static const hipError_t attr = hipFuncSetAttribute((const void*) ntt_fwd_single<S1>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 << (S1 + 8));
            (void) attr;
            hipLaunchKernelGGL((ntt_fwd_single<S1>), dim3(batch), dim3(16 << S1), (size_t) 8 << (S1 + 8), st, a);
            return;
        }
        // a decomposing launch whose caller allows it (no operand of the epilogue may alias the source limbs: every
        // workgroup reads a whole source limb while others already store) and that is neither the multi-modulus
        // column pass nor carrying a copy
        if (a.decomp_mods && a.single_decomp_ok && !a.copy_src && !a.src_inv && !use_col_multi<S1>(a, batch) &&
            use_single_pass(a, batch)) {
            static const hipError_t attr_d = hipFuncSetAttribute((const void*) ntt_fwd_single<S1, true>,
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 8 << (S1 + 8));
            (void) attr_d;
            NttArgs c = a; // natural order: the targets of one source limb run back to back and share it in L2
            c.group_span = 0;
            c.mg_group_span = 0;
            hipLaunchKernelGGL((ntt_fwd_single<S1, true>), dim3(batch), dim3(16 << S1), (size_t) 8 << (S1 + 8), st, c);
            return;
        }
    }
    if (use_col_multi<S1>(a, batch)) {
        launch_col_multi<S1>(a, batch, st);
    } else if (a.decomp_mods) {
        NttArgs c = a; // natural order for the decomposing column pass (see ntt_launch_fwd_col)
        c.group_span = 0;
        c.mg_group_span = 0;
        hipLaunchKernelGGL((ntt_fwd_col<S1, true>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, c);
    } else
        hipLaunchKernelGGL((ntt_fwd_col<S1, false>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, a);
    NttArgs b = a;
    b.in = a.out;
    b.in_item_stride = a.out_item_stride;
    // the row pass is in place on `out`; it keeps decomp_mods only to know the digit
    hipLaunchKernelGGL(ntt_fwd_row, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, b);
}

template <int S1>
static void launch_inv(const NttArgs& a, int batch, hipStream_t st)
{
    constexpr int CT = 4096 >> S1;
    if constexpr (S1 <= 6) {
        if (use_single_pass(a, batch) && !a.poly_order) { // LDS-resident single pass (N <= 2^14)
            static const hipError_t attr = hipFuncSetAttribute((const void*) ntt_inv_single<S1, false>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 << (S1 + 8));
            static const hipError_t attr_e = hipFuncSetAttribute((const void*) ntt_inv_single<S1, true>,
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 8 << (S1 + 8));
            static const hipError_t attr_t = hipFuncSetAttribute((const void*) ntt_inv_single<S1, false, true>,
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 8 << (S1 + 8));
            (void) attr;
            (void) attr_e;
            (void) attr_t;
            if (a.tensor_in)
                hipLaunchKernelGGL((ntt_inv_single<S1, false, true>), dim3(batch), dim3(16 << S1), (size_t) 8 << (S1 + 8), st, a);
            else if (a.iepi.on)
                hipLaunchKernelGGL((ntt_inv_single<S1, true>), dim3(batch), dim3(16 << S1), (size_t) 8 << (S1 + 8), st, a);
            else
                hipLaunchKernelGGL((ntt_inv_single<S1, false>), dim3(batch), dim3(16 << S1), (size_t) 8 << (S1 + 8), st, a);
            return;
        }
    }
    if (a.tensor_in) hipLaunchKernelGGL(ntt_inv_row_tensor, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, a);
    else hipLaunchKernelGGL(ntt_inv_row, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, a);
    NttArgs b = a;
    b.in = a.out;
    b.in_item_stride = a.out_item_stride;
    if (a.iepi.on) hipLaunchKernelGGL((ntt_inv_col<S1, true>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, b);
    else hipLaunchKernelGGL((ntt_inv_col<S1, false>), dim3(256 / CT, batch), dim3(NTT_THREADS), 0, st, b);
}

static unsigned magic16(int d) { return d <= 1 ? 0u : (unsigned) ((0x100000000ull + (unsigned) d - 1) / (unsigned) d); }
// divisors of select_poly; every dividend is a grid index (< 65536) or polys_per_item (<= batch)
static void fill_magics(NttArgs& g)
{
    g.mg_group_span = magic16(g.group_span);
    g.mg_mod_count = magic16(g.mod_count);
    g.mg_per_item = magic16(g.mod_count ? g.polys_per_item / g.mod_count : 0);
    g.mg_polys_per_item = magic16(g.polys_per_item);
    g.mg_decomp_mods = magic16(g.decomp_mods);
    g.epi.mg_limbs = magic16(g.epi.limbs);
    g.iepi.mg_slots = magic16(g.iepi.limbs + (g.iepi.p_count ? g.iepi.p_count : 1));
    g.mg_tensor_limbs = magic16(g.tensor_limbs);
}

bool ntt_decomp_uses_multi(const NttArgs& a, int batch)
{
    if (!a.plan_has_fp) return false;
    switch (a.n_power - 8) {
        case 4: return use_col_multi<4>(a, batch);
        case 5: return use_col_multi<5>(a, batch);
        case 6: return use_col_multi<6>(a, batch);
        case 7: return use_col_multi<7>(a, batch);
        case 8: return use_col_multi<8>(a, batch);
    }
    return false;
}

hipError_t ntt_launch_inv_rows(const NttArgs& a, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (a.n_power < 12 || a.n_power > 16 || batch > 65535) return hipErrorInvalidValue;
    NttArgs g = a;
    g.group_span = 0;
    if (!a.poly_order && a.mod_count > 1 && batch % a.mod_count == 0 &&
        (!a.polys_per_item || a.polys_per_item % a.mod_count == 0))
        g.group_span = batch / a.mod_count;
    fill_magics(g);
    hipLaunchKernelGGL(ntt_inv_row, dim3((1u << a.n_power) / 4096, batch), dim3(NTT_THREADS), 0, st, g);
    if (a.plan_has_int) { // limbs of integer moduli: column stages here, the FP64 ones in the src_inv launch
        NttArgs b = g;
        b.in = g.out;
        b.in_item_stride = g.out_item_stride;
        b.only_int = 1;
        switch (a.n_power - 8) {
#define CASE(S) case S: hipLaunchKernelGGL((ntt_inv_col<S, false>), dim3(256 / (4096 >> S), batch), dim3(NTT_THREADS), 0, st, b); break;
            CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        }
    }
    return hipGetLastError();
}

hipError_t ntt_launch_fwd_col(const NttArgs& a, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (a.n_power < 12 || a.n_power > 16 || batch > 65535 || a.poly_order) return hipErrorInvalidValue;
    if (a.src_inv && !ntt_decomp_uses_multi(a, batch)) return hipErrorInvalidValue;
    NttArgs g = a;
    g.group_span = 0;
    // A decomposing launch walks the polynomials in their natural order (item, digit, modulus slot):
    // the rc consumers of one source limb are then dispatched back to back -- tile x of all of them on
    // XCD x % 8 -- and all but the first read it from that XCD's L2.  (The modulus-major walk of the
    // plain transform would put them batch / rc polynomials apart: rc reads of every source limb from
    // HBM, 8.6 GB instead of 0.5 GB per 64-ciphertext launch at C4.)  The column pass only touches the
    // first 256 twiddles of a modulus, 4 KiB, so it has no table locality to protect.
    if (!a.decomp_mods && a.mod_count > 1 && batch % a.mod_count == 0 &&
        (!a.polys_per_item || a.polys_per_item % a.mod_count == 0))
        g.group_span = batch / a.mod_count;
    fill_magics(g);
    switch (a.n_power - 8) {
        case 4: launch_fwd_col_only<4>(g, batch, st); break;
        case 5: launch_fwd_col_only<5>(g, batch, st); break;
        case 6: launch_fwd_col_only<6>(g, batch, st); break;
        case 7: launch_fwd_col_only<7>(g, batch, st); break;
        case 8: launch_fwd_col_only<8>(g, batch, st); break;
    }
    return hipGetLastError();
}

hipError_t ntt_launch(const NttArgs& a, int batch, bool inverse, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (a.n_power < 12 || a.n_power > 16) return hipErrorInvalidValue;
    if (a.decomp_mods && (inverse || a.poly_order || !a.polys_per_item)) return hipErrorInvalidValue;
    if (a.src_inv && (!a.decomp_mods || batch > 65535 || !ntt_decomp_uses_multi(a, batch))) return hipErrorInvalidValue;
    if (a.copy_src && (!a.decomp_mods || inverse)) return hipErrorInvalidValue;
    if (a.tensor_in && (!inverse || a.poly_order || a.iepi.on || a.polys_per_item != 3 * a.tensor_limbs || batch % a.polys_per_item))
        return hipErrorInvalidValue;
    if (a.iepi.on && (!inverse || a.poly_order || batch % a.polys_per_item ||
                      a.polys_per_item != 2 * (a.iepi.limbs + (a.iepi.p_count ? a.iepi.p_count : 1)) ||
                      ((a.iepi.p_count > 1) != (a.iepi.u != nullptr))))
        return hipErrorInvalidValue;
    if (batch > 65535) {
        // gridDim.y limit: split (poly_order / mod_order semantics need the
        // absolute polynomial index, so only plain batches are split)
        // split on item boundaries (or modulus-cycle boundaries for a flat batch)
        const int unit = a.polys_per_item ? a.polys_per_item : a.mod_count;
        if ((!a.polys_per_item && a.poly_order) || (batch % unit) || unit > 65535) return hipErrorInvalidValue;
        int done = 0;
        while (done < batch) {
            int chunk = batch - done;
            int maxc = (65535 / unit) * unit;
            if (chunk > maxc) chunk = maxc;
            NttArgs c = a;
            if (a.polys_per_item) {
                // every per-item pointer of the launch moves with the items of the piece
                const u64 items_done = (u64) (done / unit);
                c.in = a.in + items_done * a.in_item_stride;
                c.out = a.out + items_done * a.out_item_stride;
                if (a.epi.on) {
                    c.epi.ks = a.epi.ks + items_done * a.epi.ks_item_stride;
                    if (a.epi.ct) c.epi.ct = a.epi.ct + items_done * a.epi.ct_item_stride;
                    c.epi.out = a.epi.out + items_done * a.epi.out_item_stride;
                }
                if (a.iepi.on) {
                    if (a.iepi.ct) c.iepi.ct = a.iepi.ct + items_done * a.iepi.ct_item_stride;
                    c.iepi.out = a.iepi.out + items_done * a.iepi.out_item_stride;
                    if (a.iepi.u) c.iepi.u = a.iepi.u + items_done * a.iepi.u_item_stride;
                }
                if (a.copy_src) {
                    c.copy_src = a.copy_src + items_done * a.copy_src_item_stride;
                    c.copy_dst = a.copy_dst + items_done * a.copy_dst_item_stride;
                }
                if (a.tensor_in) c.tensor_in = a.tensor_in + items_done * a.tensor_item_stride;
            } else {
                c.in = a.in + ((u64) done << a.n_power);
                c.out = a.out + ((u64) done << a.n_power);
            }
            hipError_t e = ntt_launch(c, chunk, inverse, st);
            if (e != hipSuccess) return e;
            done += chunk;
        }
        return hipSuccess;
    }
    NttArgs g = a;
    g.group_span = 0;
    if (!a.poly_order && a.mod_count > 1 && batch % a.mod_count == 0 &&
        (!a.polys_per_item || a.polys_per_item % a.mod_count == 0))
        g.group_span = batch / a.mod_count;
    fill_magics(g);
    switch (a.n_power - 8) {
#define CASE(S)                                   \
    case S:                                       \
        if (inverse) launch_inv<S>(g, batch, st); \
        else launch_fwd<S>(g, batch, st);         \
        break;
        CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return hipGetLastError();
}

} // namespace hegpu

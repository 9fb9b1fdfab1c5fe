// tfhe.hip -- TFHE gate bootstrapping (config C5) for gfx950.
//
// Replaces the reference's tfhe_*_pre_comp_kernel, the 2 x n = 1024 launches of
// tfhe_bootstrapping_kernel_{unique,regular}_step{1,2}, tfhe_sample_extraction_kernel
// and tfhe_key_switching_kernel (reference src/lib/kernel/bootstrapping.cu:378-1436,
// host loop src/lib/host/tfhe/operator.cu:200-294) and SmallForwardNTT /
// SmallInverseNTT (src/lib/kernel/small_ntt.cu:10-126).
//
// MI355X design: ONE persistent workgroup per gate runs all n = 512 blind-rotate
// iterations.  The accumulator (2 x 1024 int32 = 8 KiB) never leaves LDS, where
// the reference round-trips a 64 KiB product buffer and the accumulator through
// HBM 512 times per gate.  A workgroup is 4 wavefronts; wavefront (y,z) owns
// the gadget digit z of accumulator polynomial y and runs its 1024-point
// negacyclic NTT alone: 16 coefficients per lane, radix-16 / radix-16 / radix-4
// rounds with wave-local LDS exchanges (no s_barrier inside a transform).  The
// external product is reduced across the four wavefronts through the same
// 32 KiB of LDS, wavefronts 0/1 run the two inverse transforms and update the
// accumulator.  All arithmetic is exact (60-bit NTT prime), so the int32 torus
// results are identical to the reference's.
#include "tfhe.hpp"
#include <cstdlib>
#define HEGPU_FP_TU tfhe // (names this file's table reader in the instrumented test build, see fpmod.cuh)
#include "fpmod.cuh"
#include "drbg.hpp"

namespace hegpu {

#define TF_N 1024
// from this many gates per call the key switching runs several gates per workgroup (sweep: profiles/r4_c5/ks_shapes.txt)
#define TFHE_KS_BATCH_MIN 48
#define TF_THREADS 256

__device__ __forceinline__ u64 tcsub(u64 x, u64 m) { return (x >= m) ? x - m : x; }

struct TQ {
    u64 q, q4;
    u32 nq0, nq1;
};

// 9-multiply lazy Shoup product, result in [0,4q) (see ntt.hip:shoup_lazy)
__device__ __forceinline__ u64 tshoup(u64 y, ulonglong2 w, const TQ& c)
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

__device__ __forceinline__ void t_ct(u64& x, u64& y, ulonglong2 w, const TQ& c)
{
    u64 u = tcsub(x, c.q4);
    u64 t = tshoup(y, w, c);
    x = u + t;
    y = u + c.q4 - t;
}

__device__ __forceinline__ void t_gs(u64& x, u64& y, ulonglong2 w, const TQ& c)
{
    u64 s = x + y;
    u64 d = x + c.q4 - y;
    x = tcsub(s, c.q4);
    y = tshoup(d, w, c);
}

// LDS staging index: one pad element per 16 so that the 16-contiguous-per-lane
// and the 64-element-block access patterns are both bank-conflict free.
#define TF_BUF (TF_N + TF_N / 16)
__device__ __forceinline__ int bi(int e) { return e + (e >> 4); }

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Forward 1024-point NTT by one wavefront.  In: x[k] = element lane + 64k
// (any value < 8q).  Out: x[k] = slot 16*lane + k, canonical.  `buf` = this
// wave's 1024-element LDS area.
__device__ __forceinline__ void wave_ntt1024(u64 (&x)[16], u64* buf, const ulonglong2* __restrict__ tw,
                                             const TQ& c, int lane)
{
    // stages 0-3 (strides 512..64): roots 1 | 2,3 | 4..7 | 8..15, wave-uniform
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = tw[(1 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++) t_ct(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) buf[bi(lane + 64 * k)] = x[k];
    wave_fence();
    // stages 4-7 inside 64-element blocks: lane = (block b, c0), elements 64b + c0 + 4m
    const int b = lane >> 2, c0 = lane & 3;
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[bi(64 * b + c0 + 4 * m)];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const ulonglong2 w = tw[((16 + b) << s) + bb];
#pragma unroll
            for (int j = 0; j < half; j++) t_ct(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) buf[bi(64 * b + c0 + 4 * m)] = x[m];
    wave_fence();
    // stages 8,9 on 16 contiguous elements per lane
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = buf[bi(16 * lane + k)];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const ulonglong2 w8 = tw[256 + 4 * lane + g];
        t_ct(x[4 * g + 0], x[4 * g + 2], w8, c);
        t_ct(x[4 * g + 1], x[4 * g + 3], w8, c);
        const ulonglong2 w9a = tw[512 + 8 * lane + 2 * g], w9b = tw[512 + 8 * lane + 2 * g + 1];
        t_ct(x[4 * g + 0], x[4 * g + 1], w9a, c);
        t_ct(x[4 * g + 2], x[4 * g + 3], w9b, c);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = tcsub(tcsub(tcsub(x[k], c.q4), 2 * c.q), c.q);
    wave_fence();
}

// Inverse 1024-point NTT by one wavefront.  In: x[k] = slot 16*lane + k
// (canonical).  Out: x[k] = coefficient lane + 64k, canonical, N^-1 applied.
__device__ __forceinline__ void wave_intt1024(u64 (&x)[16], u64* buf, const ulonglong2* __restrict__ itw,
                                              ulonglong2 ninv, ulonglong2 w1ninv, const TQ& c, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const ulonglong2 w9a = itw[512 + 8 * lane + 2 * g], w9b = itw[512 + 8 * lane + 2 * g + 1];
        t_gs(x[4 * g + 0], x[4 * g + 1], w9a, c);
        t_gs(x[4 * g + 2], x[4 * g + 3], w9b, c);
        const ulonglong2 w8 = itw[256 + 4 * lane + g];
        t_gs(x[4 * g + 0], x[4 * g + 2], w8, c);
        t_gs(x[4 * g + 1], x[4 * g + 3], w8, c);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) buf[bi(16 * lane + k)] = x[k];
    wave_fence();
    const int b = lane >> 2, c0 = lane & 3;
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[bi(64 * b + c0 + 4 * m)];
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        const int half = 8 >> s;
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const ulonglong2 w = itw[((16 + b) << s) + bb];
#pragma unroll
            for (int j = 0; j < half; j++) t_gs(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) buf[bi(64 * b + c0 + 4 * m)] = x[m];
    wave_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = buf[bi(lane + 64 * k)];
#pragma unroll
    for (int s = 3; s >= 1; s--) {
        const int half = 8 >> s;
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const ulonglong2 w = itw[(1 << s) + bb];
#pragma unroll
            for (int j = 0; j < half; j++) t_gs(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { // last stage with N^-1 folded in, exact
        u64 s = x[j] + x[j + 8];
        u64 d = x[j] + c.q4 - x[j + 8];
        u64 r0 = tshoup(s, ninv, c), r1 = tshoup(d, w1ninv, c);
        x[j] = tcsub(tcsub(r0, 2 * c.q), c.q);
        x[j + 8] = tcsub(tcsub(r1, 2 * c.q), c.q);
    }
    wave_fence();
}

// reference bootstrapping.cu:662-674
__device__ __forceinline__ int modswitch(int input, int modulus_log)
{
    const u64 range_log = 63 - modulus_log;
    const u64 half_range = 1ULL << (range_log - 1);
    const u64 r = (((u64) (u32) input) << 32) + half_range;
    return (int) (r >> range_log);
}

// Boot key re-layout: reference order [i][y][z][c][N] (slot s) -> [i][y][z][c][k][lane]
// with s = 16*lane + k, so that a wavefront reads its 16 slots per lane with
// fully coalesced 512-byte loads.
__global__ __launch_bounds__(256) void k_tfhe_prepare_bootkey(const u64* __restrict__ src, u64* __restrict__ dst,
                                                              u64 polys)
{
    const u64 p = blockIdx.x;
    if (p >= polys) return;
    for (int t = threadIdx.x; t < TF_N; t += 256) {
        const int k = t >> 6, lane = t & 63;
        dst[p * TF_N + t] = src[p * TF_N + 16 * lane + k];
    }
}

// Blind rotation + sample extraction for one gate per workgroup.
// in_a [shape][n], in_b [shape]; bk = prepared boot key; out_a [shape][N], out_b [shape].
__global__ __launch_bounds__(TF_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tfhe_blind_rotate(const int* __restrict__ in_a,
                                                                  const int* __restrict__ in_b,
                                                                  const u64* __restrict__ bk, int* __restrict__ out_a,
                                                                  int* __restrict__ out_b, TfheDev p, int encoded, int shape)
{
    if (bk[0] != 0) { // FP64-layout key: k_tfhe_blind_rotate_fp of the same call runs instead
        if (bk[0] != 1 && blockIdx.x == 0 && threadIdx.x == 0 && p.bad_key) *p.bad_key = 1; // neither layout: say so
        return;
    }
    bk += TFHE_PREP_HEADER;
    __shared__ int acc[2][TF_N];
    __shared__ __attribute__((aligned(16))) u64 buf[4][TF_BUF];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int y = wv >> 1, z = wv & 1;
    const int n = p.n;
    // gates in a grid-stride loop: the grid is capped (tfhe_blind_rotate), so a call whose key has the other layout pays
    // for at most TFHE_INT_GRID_MAX workgroups that read one word and leave, not for one per gate
    for (int g = blockIdx.x; g < shape; g += gridDim.x) {
    TQ c;
    c.q = p.mod.q;
    c.q4 = 4 * c.q;
    {
        u64 nq = 0 - c.q;
        c.nq0 = (u32) nq;
        c.nq1 = (u32) (nq >> 32);
    }
    const u64 threshold = c.q >> 1;

    // acc_0 = (0, X^(2N - b~) * mu)   (bootstrapping.cu:905-933)
    {
        const int bN = 2 * TF_N - modswitch(in_b[g], 10);
        for (int j = t; j < TF_N; j += TF_THREADS) {
            acc[0][j] = 0;
            acc[1][j] = (bN < TF_N) ? ((j < bN) ? -encoded : encoded) : ((j < bN - TF_N) ? encoded : -encoded);
        }
    }
    __syncthreads();

    const int shift = 32 - 10 * (z + 1);
    for (int i = 0; i < n; i++) {
        const int aN = modswitch(in_a[(u64) g * n + i], 10);
        // (X^a~ * acc_y - acc_y), gadget digit z, lifted to Z_q  (bootstrapping.cu:1063-1119)
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = lane + 64 * k;
            // X^aN * acc, coefficient j: acc[(j - aN) mod 2N] with the sign of the negacyclic wrap -- index
            // arithmetic and a select (branches per element would put every LDS read in its own basic block)
            const int idx = (j - aN) & (2 * TF_N - 1);
            const int v = acc[y][idx & (TF_N - 1)];
            const int r = (idx & TF_N) ? -v : v;
            const u32 diff = (u32) r - (u32) acc[y][j];
            const int d = (int) (((diff + (u32) p.offset) >> shift) & (u32) p.mask_mod) - p.half_bg;
            x[k] = (d < 0) ? (u64) (c.q + (long long) d) : (u64) d;
        }
        wave_ntt1024(x, buf[wv], p.tw, c, lane);
        // external product terms with BK_i[y][z][c], c = 0,1
        const u64* bkp = bk + ((((u64) i * 2 + y) * 2 + z) * 2) * TF_N + lane;
        u64 p0[16], p1[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            p0[k] = mul_barrett(x[k], bkp[64 * k], p.mod);
            p1[k] = mul_barrett(x[k], bkp[TF_N + 64 * k], p.mod);
        }
        __syncthreads(); // every wave is done with its NTT staging area
        // reduction over the four (y,z) waves through LDS
        if (wv >= 2) {
            u64* d0 = buf[2 * (wv - 2)];
            u64* d1 = buf[2 * (wv - 2) + 1];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                d0[bi(lane + 64 * k)] = p0[k];
                d1[bi(lane + 64 * k)] = p1[k];
            }
        }
        __syncthreads();
        if (wv < 2) {
            const u64* s0 = buf[2 * wv];
            const u64* s1 = buf[2 * wv + 1];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                p0[k] = add_mod(p0[k], s0[bi(lane + 64 * k)], c.q);
                p1[k] = add_mod(p1[k], s1[bi(lane + 64 * k)], c.q);
            }
        }
        __syncthreads();
        if (wv < 2) {
            // wave 0 keeps c=0 and hands c=1 to wave 1; wave 1 keeps c=1, hands c=0
            u64* give = buf[wv];
#pragma unroll
            for (int k = 0; k < 16; k++) give[bi(lane + 64 * k)] = (wv == 0) ? p1[k] : p0[k];
        }
        __syncthreads();
        if (wv < 2) {
            const u64* take = buf[1 - wv];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u64 mine = (wv == 0) ? p0[k] : p1[k];
                x[k] = add_mod(mine, take[bi(lane + 64 * k)], c.q);
            }
        }
        __syncthreads();
        if (wv < 2) {
            wave_intt1024(x, buf[wv], p.itw, p.ninv, p.w1ninv, c, lane);
            // centred lift to int32 and accumulate (bootstrapping.cu:1294-1311)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int post = (x[k] >= threshold) ? (int) (long long) (x[k] - c.q) : (int) (long long) x[k];
                const int j = lane + 64 * k;
                acc[wv][j] = (int) ((u32) acc[wv][j] + (u32) post);
            }
        }
        __syncthreads();
    }
    // sample extraction at index 0 (bootstrapping.cu:1314-1347), k = 1
    for (int j = t; j < TF_N; j += TF_THREADS)
        out_a[(u64) g * TF_N + j] = (j < 1) ? acc[0][j] : (int) (0u - (u32) acc[0][TF_N - j]);
    if (t == 0) out_b[g] = acc[1][0];
    __syncthreads(); // the accumulators are free for the next gate
    }
}

// ------------------------------------------------------------------ FP64 blind rotate
// The external product is an exact integer negacyclic convolution whose result
// is only needed modulo 2^32.  For a real boot key (torus32 coefficients v) it
// is computed modulo a 44-bit prime p' in FP64 (fpmod.cuh; ~3x fewer
// instructions per butterfly than with the reference's 60-bit prime) after
// splitting v = hi*2^16 + lo, 0 <= lo < 2^16, |hi| <= 2^15:
//   |conv(d, lo or hi)| <= (k+1)*l * N * Bg/2 * 2^16 = 4 * 1024 * 512 * 65536 = 2^37 < p'/2,
// so both halves are recovered exactly and  lo-part + (hi-part << 16) mod 2^32
// equals the reference's centred result mod 2^32.  Four output polynomials
// (c = 0,1 x {lo,hi}) -> one inverse transform per wavefront: no idle waves.
// The boot key is converted once (k_tfhe_prepare_bootkey_fp); a key whose
// coefficients do not fit int32 keeps the integer path.
__device__ __forceinline__ void f_ct(double& x, double& y, ulonglong2 w, const FC& c)
{
    const double t = fp_mul(y, as_f64(w.x), as_f64(w.y), c);
    y = x - t;
    x = x + t;
    FP_AUDIT_VAL(c, FPM_SUM, x);
    FP_AUDIT_VAL(c, FPM_SUM, y);
}
__device__ __forceinline__ void f_gs(double& x, double& y, ulonglong2 w, const FC& c)
{
    const double s = x + y, d = x - y;
    x = s;
    y = fp_mul(d, as_f64(w.x), as_f64(w.y), c);
    FP_AUDIT_VAL(c, FPM_SUM, s);
}

// Forward 1024-point NTT mod p' by one wavefront.  In: x[k] = element lane + 64k,
// |x| <= p'.  Out: x[k] = slot 16*lane + k, centred (|x| <= p'/2).  Magnitudes
// grow by at most 0.51 p' per stage (10 stages: < 7 p' < 2^47): no reduction
// before the end.
// `twl`: the same table for the lane-dependent twiddles (stages 4..9); the blind rotate passes an LDS copy
// (the 27 per-lane loads of every transform are loop-invariant and otherwise come from L2 630 times)
__device__ __forceinline__ void fwave_ntt1024(double (&x)[16], u64* buf, const ulonglong2* __restrict__ tw,
                                              const ulonglong2* twl, const FC& c, int lane)
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        FP_STAGE(c, s);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = tw[(1 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++) f_ct(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) buf[bi(lane + 64 * k)] = as_bits(x[k]);
    wave_fence();
    const int b = lane >> 2, c0 = lane & 3;
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = as_f64(buf[bi(64 * b + c0 + 4 * m)]);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        FP_STAGE(c, 4 + s);
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const ulonglong2 w = twl[((16 + b) << s) + bb];
#pragma unroll
            for (int j = 0; j < half; j++) f_ct(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) buf[bi(64 * b + c0 + 4 * m)] = as_bits(x[m]);
    wave_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = as_f64(buf[bi(16 * lane + k)]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const ulonglong2 w8 = twl[256 + 4 * lane + g];
        FP_STAGE(c, 8);
        f_ct(x[4 * g + 0], x[4 * g + 2], w8, c);
        f_ct(x[4 * g + 1], x[4 * g + 3], w8, c);
        const ulonglong2 w9a = twl[512 + 8 * lane + 2 * g], w9b = twl[512 + 8 * lane + 2 * g + 1];
        FP_STAGE(c, 9);
        f_ct(x[4 * g + 0], x[4 * g + 1], w9a, c);
        f_ct(x[4 * g + 2], x[4 * g + 3], w9b, c);
    }
    FP_STAGE(c, 10);
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = fp_reduce(x[k], c);
    wave_fence();
}

// low 32 bits (two's complement) of an integer-valued double, |v| < 2^51
__device__ __forceinline__ u32 f_low32(double v) { return (u32) as_bits(v + 6755399441055744.0); }

// Boot key conversion: one wavefront per polynomial.  src: reference layout,
// NTT domain mod q (slot order); dst: [poly][half][k / 2][lane][k & 1] doubles = forward
// NTT mod p' of the lo / hi halves of the int32 coefficients, centred.  Sets
// *bad when a coefficient does not fit int32.
__global__ __launch_bounds__(64) void k_tfhe_prepare_bootkey_fp(const u64* __restrict__ src, u64* __restrict__ dst,
                                                                u64 polys, TfheDev p, int* bad)
{
    __shared__ __attribute__((aligned(16))) u64 buf[TF_BUF];
    const u64 pi = blockIdx.x;
    if (pi >= polys) return;
    const int lane = threadIdx.x;
    TQ c;
    c.q = p.mod.q;
    c.q4 = 4 * c.q;
    {
        u64 nq = 0 - c.q;
        c.nq0 = (u32) nq;
        c.nq1 = (u32) (nq >> 32);
    }
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = src[pi * TF_N + 16 * lane + k];
    wave_intt1024(x, buf, p.itw, p.ninv, p.w1ninv, c, lane);
    const FC fc = make_fc(p.fprime, FP_SITE(FPS_TFHE_PREP, 0));
    double lo[16], hi[16];
    bool oob = false;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const long long v = (x[k] > (c.q >> 1)) ? (long long) (x[k] - c.q) : (long long) x[k];
        oob |= (v < -2147483648LL) || (v > 2147483647LL);
        const int vi = (int) v;
        lo[k] = (double) (vi & 0xffff);
        hi[k] = (double) (vi >> 16);
    }
    if (oob) atomicOr(bad, 1);
    fwave_ntt1024(lo, buf, p.ftw, p.ftw, fc, lane);
    fwave_ntt1024(hi, buf, p.ftw, p.ftw, fc, lane);
    // [k / 2][lane][k & 1]: a lane's slots 16 lane + k, two per 16-byte load, 64 lanes contiguous (1 KiB per load)
    u64* d = dst + pi * 2 * TF_N;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        d[(k >> 1) * 128 + lane * 2 + (k & 1)] = as_bits(lo[k]);
        d[TF_N + (k >> 1) * 128 + lane * 2 + (k & 1)] = as_bits(hi[k]);
    }
}

__global__ void k_tfhe_set_header(u64* hdr, u64 fmt) { hdr[0] = fmt; }

// ------------------------------------------------------------------ FP64 blind rotate, three workgroups per CU
// Round 4.  Counters of the kernel above at 8192 gates (profiles/r4a_c5/): 2.37 GHz (not power-limited -- there is
// no HBM stream next to the FP64 work), vector ALU 0.61 busy, waves waiting 45 % of their cycles, 1.9 waves per SIMD,
// and SQ_LDS_BANK_CONFLICT = 18 % of the CU cycles.  Two causes, two changes:
//  * the per-lane twiddles of stages 4..9 sat in LDS in table order ((w, w') pairs, 16 bytes): lane l reads
//    tw[256 + 4l + g] (64-byte lane stride: 4-way conflict) and tw[512 + 8l + e] (128-byte stride: 8-way).  Here they
//    are re-laid in the order of use -- [stage][slot][lane] -- so a wavefront reads 64 consecutive doubles.
//  * 74 KiB of LDS and 254 registers allow two workgroups per CU (two waves per SIMD).  One table serves both
//    directions (itw[2^L + o] = -tw[2^L + (2^L - 1 - o)]: the inverse reads the forward table mirrored and multiplies
//    y - x instead of x - y), entries are the plain double w with the companion formed as w * RN(1/p') (one multiply per
//    twiddle read; a representative may differ from the RN(w/p') one, the exact integer result cannot): 8 KiB instead
//    of 32; the key values are requested two polynomials at a time (64 registers instead of 128): 50 KiB and <= 168
//    registers = three workgroups per CU, the third wave of a SIMD filling the LDS-exchange and barrier waits.
#define TF3_TW 1008 // 240 (stages 4..7: [s][bb][b]) + 256 (stage 8: [g][lane]) + 512 (stage 9: [e][lane])
__device__ __forceinline__ int tf3_src(int e)
{
    if (e < 240) {
        const int r = (e >> 4) + 1, b = e & 15; // r = (1 << s) + bb
        const int s = 31 - __builtin_clz(r);
        return ((16 + b) << s) + (r - (1 << s));
    }
    if (e < 496) return 256 + 4 * ((e - 240) & 63) + ((e - 240) >> 6);
    return 512 + 8 * ((e - 496) & 63) + ((e - 496) >> 6);
}
__device__ __forceinline__ void f_ct_l(double& x, double& y, double w, const FC& c)
{
    const double t = fp_mul(y, w, w * c.qi, c);
    y = x - t;
    x = x + t;
    FP_AUDIT_VAL(c, FPM_SUM, x);
    FP_AUDIT_VAL(c, FPM_SUM, y);
}
// inverse butterfly with the FORWARD twiddle of the mirrored position: (x - y) * (-w) = (y - x) * w
__device__ __forceinline__ void f_gs_l(double& x, double& y, double w, const FC& c)
{
    const double s = x + y, d = y - x;
    x = s;
    y = fp_mul(d, w, w * c.qi, c);
    FP_AUDIT_VAL(c, FPM_SUM, s);
}
// as fwave_ntt1024, lane-dependent twiddles from the re-laid LDS table
__device__ __forceinline__ void fwave_ntt1024_l(double (&x)[16], u64* buf, const ulonglong2* __restrict__ tw,
                                                const double* twl, const FC& c, int lane)
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        FP_STAGE(c, s);
#pragma unroll
        for (int b = 0; b < (1 << s); b++) {
            const ulonglong2 w = tw[(1 << s) + b];
#pragma unroll
            for (int j = 0; j < half; j++) f_ct(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
    // Round 5: the lane-dependent twiddles are REQUESTED AHEAD of the exchange they follow (they depend on nothing but the
    // lane) and pinned there with scheduling fences.  Left to itself the compiler issued each twiddle read right in front
    // of its butterflies -- ds_read, s_waitcnt lgkmcnt(0), butterflies, fifteen times per round: one exposed LDS round trip
    // per twiddle (profiles/r5d_c5/README.md).
    const int b = lane >> 2, c0 = lane & 3;
    double wa[15];
#pragma unroll
    for (int e = 0; e < 15; e++) wa[e] = twl[16 * e + b];
#pragma unroll
    for (int k = 0; k < 16; k++) buf[bi(lane + 64 * k)] = as_bits(x[k]);
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = as_f64(buf[bi(64 * b + c0 + 4 * m)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int half = 8 >> s;
        FP_STAGE(c, 4 + s);
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const double w = wa[(1 << s) - 1 + bb];
#pragma unroll
            for (int j = 0; j < half; j++) f_ct_l(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
    double w8[4], w9[8];
#pragma unroll
    for (int g = 0; g < 4; g++) w8[g] = twl[240 + 64 * g + lane];
#pragma unroll
    for (int e = 0; e < 8; e++) w9[e] = twl[496 + 64 * e + lane];
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) buf[bi(64 * b + c0 + 4 * m)] = as_bits(x[m]);
    wave_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = as_f64(buf[bi(16 * lane + k)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        FP_STAGE(c, 8);
        f_ct_l(x[4 * g + 0], x[4 * g + 2], w8[g], c);
        f_ct_l(x[4 * g + 1], x[4 * g + 3], w8[g], c);
        FP_STAGE(c, 9);
        f_ct_l(x[4 * g + 0], x[4 * g + 1], w9[2 * g], c);
        f_ct_l(x[4 * g + 2], x[4 * g + 3], w9[2 * g + 1], c);
    }
    // No reduction at the end: the outputs (|x| < 7 p' < 2^47, integers) are the "twiddle" side of the external
    // product's fp_mul, whose result bound does not depend on that operand's magnitude -- with the companion
    // x * RN(1/p') (relative error <= 2^-52) and a key value |y| <= p'/2 < 2^43: |k - y x / p'| <= 1/2 + 2^46 * 2^-52
    // + 2^-7, so |y x - k p'| <= 0.53 p'; h = RN(y x) < 2^90 and h - k p' is an integer below 2^45, so the FMA that
    // forms it is exact, as is the sum with l = y x - h (an integer below 2^38).  Four such terms: 2.13 p', inside the
    // 2.2 p' the inverse transform takes.  (16 x 3 instructions less per transform.)
    wave_fence();
}
// as fwave_intt1024; the lane-dependent inverse twiddles are the forward table's mirrored entries with the sign moved
// into the butterfly (f_gs_l)
__device__ __forceinline__ void fwave_intt1024_l(double (&x)[16], u64* buf, const ulonglong2* __restrict__ itw,
                                                 const double* twl, ulonglong2 ninv, ulonglong2 w1ninv, const FC& c,
                                                 int lane)
{
    const int rl = 63 - lane;
    // itw[512 + 8 lane + e] = -tw[512 + 8 (63 - lane) + (7 - e)], itw[256 + 4 lane + g] = -tw[256 + 4 (63 - lane) + (3 - g)]
    // (twiddles requested ahead of their butterflies, as in fwave_ntt1024_l)
    double w8[4], w9[8];
#pragma unroll
    for (int e = 0; e < 8; e++) w9[e] = twl[496 + 64 * e + rl];
#pragma unroll
    for (int g = 0; g < 4; g++) w8[g] = twl[240 + 64 * g + rl];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        FP_STAGE(c, 10 + 9); // (instrumented build: the inverse stages are listed as 10 + s, next to the forward ones of the site)
        f_gs_l(x[4 * g + 0], x[4 * g + 1], w9[7 - 2 * g], c);
        f_gs_l(x[4 * g + 2], x[4 * g + 3], w9[6 - 2 * g], c);
        FP_STAGE(c, 10 + 8);
        f_gs_l(x[4 * g + 0], x[4 * g + 2], w8[3 - g], c);
        f_gs_l(x[4 * g + 1], x[4 * g + 3], w8[3 - g], c);
    }
    const int b = lane >> 2, c0 = lane & 3;
    double wa[15];
#pragma unroll
    for (int e = 0; e < 15; e++) wa[e] = twl[16 * e + (15 - b)];
#pragma unroll
    for (int k = 0; k < 16; k++) buf[bi(16 * lane + k)] = as_bits(x[k]);
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = as_f64(buf[bi(64 * b + c0 + 4 * m)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 3; s >= 0; s--) {
        const int half = 8 >> s;
        FP_STAGE(c, 10 + 4 + s);
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            // itw[((16 + b) << s) + bb] = -tw[((16 + 15 - b) << s) + ((1 << s) - 1 - bb)]
            const double w = wa[(1 << s) - 1 + ((1 << s) - 1 - bb)];
#pragma unroll
            for (int j = 0; j < half; j++) f_gs_l(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = fp_reduce(x[m], c);
    wave_fence();
#pragma unroll
    for (int m = 0; m < 16; m++) buf[bi(64 * b + c0 + 4 * m)] = as_bits(x[m]);
    wave_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = as_f64(buf[bi(lane + 64 * k)]);
#pragma unroll
    for (int s = 3; s >= 1; s--) {
        const int half = 8 >> s;
        FP_STAGE(c, 10 + s);
#pragma unroll
        for (int bb = 0; bb < (1 << s); bb++) {
            const ulonglong2 w = itw[(1 << s) + bb];
#pragma unroll
            for (int j = 0; j < half; j++) f_gs(x[bb * 2 * half + j], x[bb * 2 * half + j + half], w, c);
        }
    }
    FP_STAGE(c, 10);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double s = x[j] + x[j + 8], d = x[j] - x[j + 8];
        x[j] = fp_mul(s, as_f64(ninv.x), as_f64(ninv.y), c);
        x[j + 8] = fp_mul(d, as_f64(w1ninv.x), as_f64(w1ninv.y), c);
    }
    wave_fence();
}

__global__ __launch_bounds__(TF_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_tfhe_blind_rotate_fp(
    const int* __restrict__ in_a, const int* __restrict__ in_b, const u64* __restrict__ prepared, int* __restrict__ out_a,
    int* __restrict__ out_b, TfheDev p, int encoded)
{
    if (prepared[0] != 1) return; // not the FP64 layout: the integer kernel of the same call takes it (or flags it)
    const u64* __restrict__ bk = prepared + TFHE_PREP_HEADER;
    __shared__ int acc[2][TF_N];
    __shared__ __attribute__((aligned(16))) u64 buf[4][TF_BUF];
    __shared__ double twl[TF3_TW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int y = wv >> 1, z = wv & 1;
    for (int e = t; e < TF3_TW; e += TF_THREADS) twl[e] = as_f64(p.ftw[tf3_src(e)].x);
    const int g = blockIdx.x;
    const int n = p.n;
    const FC fc = make_fc(p.fprime, FP_SITE(FPS_TFHE_BR, 0));
    {
        const int bN = 2 * TF_N - modswitch(in_b[g], 10);
        for (int j = t; j < TF_N; j += TF_THREADS) {
            acc[0][j] = 0;
            acc[1][j] = (bN < TF_N) ? ((j < bN) ? -encoded : encoded) : ((j < bN - TF_N) ? encoded : -encoded);
        }
    }
    __syncthreads();

    const int shift = 32 - 10 * (z + 1);
    const int cc = wv >> 1, sh = (wv & 1) ? 16 : 0;
    for (int i = 0; i < n; i++) {
        // key polynomial r = (digit wavefront (wv + r) & 3, output wv): [(i, digit)][o = wv][k / 2][lane][k & 1]
        const ulonglong2* bkp = reinterpret_cast<const ulonglong2*>(bk + ((u64) i * 16 + wv) * TF_N) + lane;
        // the first key polynomial is requested ahead of the decomposition and arrives under the forward transform;
        // the other three are loaded where they are used -- with three waves per SIMD their latency is covered by
        // the other workgroups (measured: no prefetch at all is as fast at 8192 gates, a rolling two-polynomial
        // prefetch needs 168 + 21 spilled registers and is slower: profiles/r4_c5/README.md)
        double ka[16];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const ulonglong2 v = bkp[(u64) (((wv + 0) & 3) * 4) * (TF_N / 2) + k * 64];
            ka[2 * k] = as_f64(v.x);
            ka[2 * k + 1] = as_f64(v.y);
        }
        const int aN = modswitch(in_a[(u64) g * n + i], 10);
        double x[16];
        {
            // X^aN * acc - acc, coefficient j = lane + 64 k: acc[(j - aN) mod 2N] with the sign of the negacyclic wrap.
            // Round 5: ALL 32 accumulator reads are requested before anything is computed, and the wrap's sign is applied
            // as (v ^ m) - m with m = 0 / -1 from the index bit.  Round 4's select on a compare kept every element's
            // condition in VCC across its own LDS read, so the compiler emitted read -> s_waitcnt lgkmcnt(0) -> select
            // sixteen times in a row: sixteen LDS round trips per iteration and wavefront on the critical path.
            int rv[16], av[16];
#pragma unroll
            for (int k = 0; k < 16; k++) rv[k] = acc[y][(lane + 64 * k - aN) & (TF_N - 1)];
#pragma unroll
            for (int k = 0; k < 16; k++) av[k] = acc[y][lane + 64 * k];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int idx = (lane + 64 * k - aN) & (2 * TF_N - 1);
                const int m = -((idx >> 10) & 1);
                const u32 diff = (u32) ((rv[k] ^ m) - m) - (u32) av[k];
                const int d = (int) (((diff + (u32) p.offset) >> shift) & (u32) p.mask_mod) - p.half_bg;
                x[k] = (double) d;
            }
        }
        fwave_ntt1024_l(x, buf[wv], p.ftw, twl, fc, lane);
        // The transformed digit goes to the own staging area (free after the transform); after the barrier
        // every wavefront reads the other three and forms its output sum_w X_w * BK[w][own] in registers
        // (plain LDS reads -- no atomics, no read-back).  x (un-reduced, below 7 p') is the "twiddle" of the products:
        // companion RN(x/p') ~ x * RN(1/p'), one multiply per product.  |sum| <= 4 * 0.53 p' (fwave_ntt1024_l).
#pragma unroll
        for (int k = 0; k < 16; k++) buf[wv][k * 64 + lane] = as_bits(x[k]);
        __syncthreads();
        FP_STAGE(fc, FP_STAGE_PRODUCT);
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = fp_mul(ka[k], x[k], x[k] * fc.qi, fc);
#pragma unroll
        for (int r = 1; r < 4; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const ulonglong2 v = bkp[(u64) (((wv + r) & 3) * 4) * (TF_N / 2) + k * 64];
                ka[2 * k] = as_f64(v.x);
                ka[2 * k + 1] = as_f64(v.y);
            }
            // (the sixteen values of the other wavefront's digit requested together: one LDS round trip, not eight)
            const u64* ob = &buf[(wv + r) & 3][lane];
            double xo[16];
#pragma unroll
            for (int k = 0; k < 16; k++) xo[k] = as_f64(ob[k * 64]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                FP_STAGE(fc, FP_STAGE_PRODUCT);
                x[k] += fp_mul(ka[k], xo[k], xo[k] * fc.qi, fc);
                FP_STAGE(fc, FP_STAGE_SUMS + r - 1);
                FP_AUDIT_VAL(fc, FPM_SUM, x[k]);
            }
        }
        __syncthreads(); // all staging areas read: the inverse transform may use them as scratch
        fwave_intt1024_l(x, buf[wv], p.fitw, twl, p.fninv, p.fw1ninv, fc, lane);
        FP_STAGE(fc, FP_STAGE_OUT);
#pragma unroll
        for (int k = 0; k < 16; k++) FP_AUDIT_VAL(fc, FPM_SUM, x[k]); // the convolution's coefficients themselves: |x| <= 2^37
#pragma unroll
        for (int k = 0; k < 16; k++)
            atomicAdd(reinterpret_cast<u32*>(&acc[cc][lane + 64 * k]), f_low32(x[k]) << sh);
        __syncthreads();
    }
    for (int j = t; j < TF_N; j += TF_THREADS)
        out_a[(u64) g * TF_N + j] = (j < 1) ? acc[0][j] : (int) (0u - (u32) acc[0][TF_N - j]);
    if (t == 0) out_b[g] = acc[1][0];
}

// out = enc + m*(s1*in1 + s2*in2) on the 32-bit torus (bootstrapping.cu:378-660)
__global__ __launch_bounds__(256) void k_tfhe_gate_pre(int* __restrict__ out_a, int* __restrict__ out_b,
                                                       const int* __restrict__ a1, const int* __restrict__ b1,
                                                       const int* __restrict__ a2, const int* __restrict__ b2,
                                                       int encoded, int s1, int s2, int m, int n)
{
    const int g = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        u32 v = 0;
        const u32 x1 = (u32) a1[(u64) g * n + i];
        const u32 x2 = a2 ? (u32) a2[(u64) g * n + i] : 0u;
        v = (s1 > 0) ? v + x1 : v - x1;
        v = (s2 > 0) ? v + x2 : v - x2;
        out_a[(u64) g * n + i] = (int) ((u32) m * v);
    }
    if (threadIdx.x == 0) {
        u32 v = (u32) encoded;
        const u32 y1 = (u32) m * (u32) b1[g];
        const u32 y2 = b2 ? (u32) m * (u32) b2[g] : 0u;
        v = (s1 > 0) ? v + y1 : v - y1;
        v = (s2 > 0) ? v + y2 : v - y2;
        out_b[g] = (int) v;
    }
}

// tfhe_key_switching_kernel (bootstrapping.cu:1349-1436): one workgroup per
// gate, 256 threads x 2 output coefficients (n = 512); the digit of each input
// coefficient is wave-uniform, key rows are 2 KiB coalesced reads.
// SPLIT: gridDim.y workgroups share a gate, `chunk` input coefficients each, and add their partial sums into the
// zeroed output with integer atomics (sums on the 32-bit torus: any order gives the same bits).  The loop over
// the N k input coefficients is a chain of L2 round trips -- 1.3 ms per gate however few gates there are -- so a
// launch that does not fill the chip is cut into more workgroups.
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_tfhe_key_switching(const int* __restrict__ in_a,
                                                            const int* __restrict__ in_b, int* __restrict__ out_a,
                                                            int* __restrict__ out_b, const int* __restrict__ ks_a,
                                                            const int* __restrict__ ks_b, int bb, int len, int n,
                                                            int Nk, int chunk)
{
    const int g = blockIdx.x, t = threadIdx.x;
    const int mask = (1 << bb) - 1;
    const u32 precision_offset = 1u << (32 - (1 + bb * len));
    const int i_begin = SPLIT ? (int) blockIdx.y * chunk : 0;
    const int i_end = SPLIT ? ((i_begin + chunk < Nk) ? i_begin + chunk : Nk) : Nk;
    u32 acc0 = 0, acc1 = 0, accb = (t == 0 && i_begin == 0) ? (u32) in_b[g] : 0u;
    const int* pa = in_a + (u64) g * Nk;
    // All `len` key rows of a coefficient are requested before any of them is used, unconditionally: a zero
    // digit (no row in the key) reads row 0 of its digit position and masks the value, a digit position beyond
    // `len` likewise.  With the loads inside `if (d != 0)` every row was a basic block of its own -- 8192
    // dependent L2 round trips per gate, 0.37 us each.
    const int t2 = (t + 256 < n) ? t + 256 : t;
    const u32 m2 = (t + 256 < n) ? 0xffffffffu : 0u;
    for (int i = i_begin; i < i_end; i++) {
        const u32 a = (u32) pa[i] + precision_offset;
        u32 v0[8], v1[8], vb[8];
#pragma unroll
        for (int i2 = 0; i2 < 8; i2++) {
            const bool on = i2 < len;
            const int d = on ? (int) ((a >> ((32 - (i2 + 1) * bb) & 31)) & (u32) mask) : 0;
            const u32 m = d ? 0xffffffffu : 0u;
            const u64 row = ((u64) i * len + (on ? i2 : 0)) * mask + (d ? d - 1 : 0);
            const int* ka = ks_a + row * n;
            v0[i2] = (u32) ka[t] & m;
            v1[i2] = (u32) ka[t2] & m & m2;
            vb[i2] = (t == 0) ? ((u32) ks_b[row] & m) : 0u;
        }
#pragma unroll
        for (int i2 = 0; i2 < 8; i2++) {
            acc0 -= v0[i2];
            acc1 -= v1[i2];
            accb -= vb[i2];
        }
    }
    if (SPLIT) {
        atomicAdd(reinterpret_cast<u32*>(&out_a[(u64) g * n + t]), acc0);
        if (t + 256 < n) atomicAdd(reinterpret_cast<u32*>(&out_a[(u64) g * n + t + 256]), acc1);
        if (t == 0) atomicAdd(reinterpret_cast<u32*>(&out_b[g]), accb);
    } else {
        out_a[(u64) g * n + t] = (int) acc0;
        if (t + 256 < n) out_a[(u64) g * n + t + 256] = (int) acc1;
        if (t == 0) out_b[g] = (int) accb;
    }
}

// Many gates per call (C5: 8192): KS_GB gates per workgroup share the key rows.  One gate per workgroup reads
// N k * len rows of 2 KiB per gate -- 16 MiB, 137 GB of L2 traffic for 8192 gates, a quarter of it rows of zero
// digits that are masked away.  Here the 2^bb - 1 = 3 candidate rows of a digit position are loaded once and each of
// the KS_GB gates subtracts the one its digit names (the digit is wave-uniform: a scalar mask per gate and row):
// 6 KiB of key per position for 8 gates instead of 16.  Same sums on the 32-bit torus, so the same bits.  The b
// column of the key: lane (gate, position) of the first wavefronts keeps its own partial sum.  base_bit = 2 only.
// A workgroup is a chain of N k iterations of ~4.5 us whatever KS_GB is (4.6 ms), three are resident per CU (144
// registers; capped at 128 the loads are no longer batched: 7.9 ms), so KS_GB is chosen to finish in ONE round of
// 768 workgroups where 8, 12 or 16 gates per workgroup allow it (tools/tfhe_ks_shapes.py).
// SPLIT: as in k_tfhe_key_switching -- gridDim.y workgroups share a group of gates, `chunk` input coefficients each,
// partial sums added into the zeroed output with integer atomics (a call of 1024 gates, C5's share per GPU: 128 groups
// of 8 x 6 pieces fill the chip, where 1024 one-gate workgroups took 1.65 ms).
template <int KS_GB, bool SPLIT>
__global__ __launch_bounds__(256) void k_tfhe_key_switching_batched(const int* __restrict__ in_a,
                                                                    const int* __restrict__ in_b,
                                                                    int* __restrict__ out_a, int* __restrict__ out_b,
                                                                    const int* __restrict__ ks_a,
                                                                    const int* __restrict__ ks_b, int len, int n, int Nk,
                                                                    int shape, int chunk)
{
    constexpr int bb = 2, mask = 3;
    __shared__ u32 bsum[KS_GB];
    // The candidate rows of a digit position as an indexable register file (round 4): slot [digit][thread] holds the
    // thread's own two key values of row digit - 1 (slot 0: zeros), written by the thread that reads them (LDS operations
    // of a wave are ordered: no barrier), so a gate's row is ONE ds_read_b64 at a wave-uniform offset and two
    // subtractions.  Round 3 picked it with six selects on scalar conditions and two masks per gate and position:
    // 768 vector instructions per iteration and thread, vector ALU 0.55 busy, 6.9 ms per 8192 gates; now 3.98 ms.
    __shared__ __attribute__((aligned(16))) u64 cand[2][4][256];
    const int t = threadIdx.x;
    cand[0][0][t] = cand[1][0][t] = 0;
    const int g0 = blockIdx.x * KS_GB;
    const u32 precision_offset = 1u << (32 - (1 + bb * len));
    const int t2 = (t + 256 < n) ? t + 256 : t;
    const u32 m2 = (t + 256 < n) ? 0xffffffffu : 0u;
    u32 acc0[KS_GB], acc1[KS_GB];
#pragma unroll
    for (int gg = 0; gg < KS_GB; gg++) acc0[gg] = acc1[gg] = 0;
    if (t < KS_GB) bsum[t] = 0;
    // (gate, position) of this lane for the b column
    const int bg = (t >> 3) < KS_GB ? (t >> 3) : 0, bp = t & 7;
    const int bgate = (g0 + bg < shape) ? g0 + bg : shape - 1;
    const bool bon = t < 8 * KS_GB && bp < len;
    u32 accb = 0;
    const int* pa[KS_GB];
#pragma unroll
    for (int gg = 0; gg < KS_GB; gg++) pa[gg] = in_a + (u64) ((g0 + gg < shape) ? g0 + gg : shape - 1) * Nk;
    const int i_begin = SPLIT ? (int) blockIdx.y * chunk : 0;
    const int i_end = SPLIT ? ((i_begin + chunk < Nk) ? i_begin + chunk : Nk) : Nk;
    for (int i = i_begin; i < i_end; i++) {
        u32 a[KS_GB];
#pragma unroll
        for (int gg = 0; gg < KS_GB; gg++) a[gg] = (u32) pa[gg][i] + precision_offset;
        if (bon) {
            const u32 ab = (u32) in_a[(u64) bgate * Nk + i] + precision_offset;
            const int d = (int) ((ab >> ((32 - (bp + 1) * bb) & 31)) & (u32) mask);
            const u64 row = ((u64) i * len + bp) * mask + (d ? d - 1 : 0);
            accb -= d ? (u32) ks_b[row] : 0u;
        }
#pragma unroll
        for (int i2 = 0; i2 < 8; i2++) {
            if (i2 < len) { // (uniform)
                const int* kr = ks_a + ((u64) i * len + i2) * mask * n;
                u32 r0[3], r1[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    r0[r] = (u32) kr[(u64) r * n + t];
                    r1[r] = (u32) kr[(u64) r * n + t2] & m2;
                }
#pragma unroll
                for (int r = 0; r < 3; r++) cand[i2 & 1][1 + r][t] = (u64) r0[r] | ((u64) r1[r] << 32);
#pragma unroll
                for (int gg = 0; gg < KS_GB; gg++) {
                    const u32 d = (a[gg] >> ((32 - (i2 + 1) * bb) & 31)) & (u32) mask; // wave-uniform
                    const u64 v = cand[i2 & 1][d][t];
                    acc0[gg] -= (u32) v;
                    acc1[gg] -= (u32) (v >> 32);
                }
            }
        }
    }
    __syncthreads();
    if (bon) atomicAdd(&bsum[bg], accb);
    __syncthreads();
#pragma unroll
    for (int gg = 0; gg < KS_GB; gg++) {
        const int g = g0 + gg;
        if (g < shape) {
            if (SPLIT) {
                atomicAdd(reinterpret_cast<u32*>(&out_a[(u64) g * n + t]), acc0[gg]);
                if (t + 256 < n) atomicAdd(reinterpret_cast<u32*>(&out_a[(u64) g * n + t + 256]), acc1[gg]);
                if (t == 0) atomicAdd(reinterpret_cast<u32*>(&out_b[g]), (i_begin == 0 ? (u32) in_b[g] : 0u) + bsum[gg]);
            } else {
                out_a[(u64) g * n + t] = (int) acc0[gg];
                if (t + 256 < n) out_a[(u64) g * n + t + 256] = (int) acc1[gg];
                if (t == 0) out_b[g] = (int) ((u32) in_b[g] + bsum[gg]);
            }
        }
    }
}

// ------------------------------------------------------------------ front end: keys, encryption, decryption
__global__ __launch_bounds__(256) void k_tfhe_gen_bits(int* __restrict__ out, int count, DrbgKey seed, u64 stream)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = drbg_bit(seed, stream, (u64) i);
}

hipError_t tfhe_gen_secret(int* lwe_key, int* tlwe_key, int n, int kN, DrbgKey seed, u64 stream0, hipStream_t st)
{
    hipLaunchKernelGGL(k_tfhe_gen_bits, dim3((n + 255) / 256), dim3(256), 0, st, lwe_key, n, seed, stream0);
    hipLaunchKernelGGL(k_tfhe_gen_bits, dim3((kN + 255) / 256), dim3(256), 0, st, tlwe_key, kN, seed, stream0 + 1);
    return hipGetLastError();
}

// one workgroup per LWE sample
__global__ __launch_bounds__(256) void k_tfhe_lwe_encrypt(int* __restrict__ out_a, int* __restrict__ out_b,
                                                          const int* __restrict__ key, const int* __restrict__ msg,
                                                          int ks_mode, const int* __restrict__ tlwe_key, int base_bit,
                                                          int len, int n, double noise_c, DrbgKey seed, u64 stream_a,
                                                          u64 stream_e)
{
    __shared__ u32 red[256];
    const u64 s = blockIdx.x;
    const int t = threadIdx.x;
    u32 acc = 0;
    for (int j = t; j < n; j += 256) {
        const int a = drbg_torus_uniform(seed, stream_a, s * n + j);
        out_a[s * n + j] = a;
        acc += (u32) a * (u32) key[j];
    }
    red[t] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) {
        u32 m;
        if (ks_mode) {
            const int mask = (1 << base_bit) - 1;
            const u64 v = s % mask + 1, j = (s / mask) % len, i = s / ((u64) mask * len);
            m = (u32) tlwe_key[i] * (u32) v * (1u << (32 - (j + 1) * base_bit));
        } else {
            m = (u32) msg[s];
        }
        out_b[s] = (int) (red[0] + m + (u32) drbg_torus_gaussian(seed, stream_e, s, noise_c));
    }
}

hipError_t tfhe_lwe_encrypt(int* out_a, int* out_b, const int* key, const int* msg, int ks_mode, const int* tlwe_key,
                            int base_bit, int len, int n, u64 shape, double noise_c, DrbgKey seed, u64 stream_a,
                            u64 stream_e, hipStream_t st)
{
    if (shape == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tfhe_lwe_encrypt, dim3((unsigned) shape), dim3(256), 0, st, out_a, out_b, key, msg, ks_mode,
                       tlwe_key, base_bit, len, n, noise_c, seed, stream_a, stream_e);
    return hipGetLastError();
}

__device__ __forceinline__ TQ make_tq(u64 q)
{
    TQ c;
    c.q = q;
    c.q4 = 4 * q;
    const u64 nq = 0 - q;
    c.nq0 = (u32) nq;
    c.nq1 = (u32) (nq >> 32);
    return c;
}
__device__ __forceinline__ u64 lift32(int v, u64 q) { return v < 0 ? q - (u64) (-(long long) v) : (u64) v; }

// NTT of the lifted TLWE key, one wavefront; out in slot order
__global__ __launch_bounds__(64) void k_tfhe_key_ntt(const int* __restrict__ tlwe_key, u64* __restrict__ out, TfheDev p)
{
    __shared__ __attribute__((aligned(16))) u64 buf[TF_BUF];
    const int lane = threadIdx.x;
    const TQ c = make_tq(p.mod.q);
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = lift32(tlwe_key[lane + 64 * k], c.q);
    wave_ntt1024(x, buf, p.tw, c, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) out[16 * lane + k] = x[k];
}

// one wavefront per boot-key row (i, y, z), k = 1
__global__ __launch_bounds__(64) void k_tfhe_gen_bootkey(u64* __restrict__ boot_key, const int* __restrict__ lwe_key,
                                                         const u64* __restrict__ tlwe_ntt, TfheDev p, double noise_c,
                                                         DrbgKey seed, u64 stream_a, u64 stream_e)
{
    __shared__ __attribute__((aligned(16))) u64 buf[TF_BUF];
    const int lane = threadIdx.x;
    const u64 row = blockIdx.x; // (i*2 + y)*l + z
    const int z = (int) (row % p.bk_l), y = (int) ((row / p.bk_l) % 2);
    const u64 i = row / (2 * (u64) p.bk_l);
    const TQ c = make_tq(p.mod.q);
    const u32 mu = (u32) lwe_key[i] << (32 - (z + 1) * p.bk_bg_bit);
    int a[16], b[16];
    u64 x[16];
    // a uniform; a*S through the transform
#pragma unroll
    for (int k = 0; k < 16; k++) {
        a[k] = drbg_torus_uniform(seed, stream_a, row * TF_N + lane + 64 * k);
        x[k] = lift32(a[k], c.q);
    }
    wave_ntt1024(x, buf, p.tw, c, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = mul_barrett(x[k], tlwe_ntt[16 * lane + k], p.mod);
    wave_intt1024(x, buf, p.itw, p.ninv, p.w1ninv, c, lane);
    const u64 half = c.q >> 1;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const long long prod = x[k] > half ? -(long long) (c.q - x[k]) : (long long) x[k];
        b[k] = (int) ((u32) prod + (u32) drbg_torus_gaussian(seed, stream_e, row * TF_N + lane + 64 * k, noise_c));
    }
    if (lane == 0) { // coefficient 0 lives in x[0] of lane 0
        if (y == 0) a[0] = (int) ((u32) a[0] + mu);
        else b[0] = (int) ((u32) b[0] + mu);
    }
    u64* out = boot_key + row * 2 * TF_N;
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = lift32(a[k], c.q);
    wave_ntt1024(x, buf, p.tw, c, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) out[16 * lane + k] = x[k];
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = lift32(b[k], c.q);
    wave_ntt1024(x, buf, p.tw, c, lane);
#pragma unroll
    for (int k = 0; k < 16; k++) out[TF_N + 16 * lane + k] = x[k];
}

hipError_t tfhe_gen_bootkey(const TfheDev& p, u64* boot_key, const int* lwe_key, const int* tlwe_key, u64* tlwe_ntt,
                            double noise_c, DrbgKey seed, u64 stream_a, u64 stream_e, hipStream_t st)
{
    if (p.k != 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tfhe_key_ntt, dim3(1), dim3(64), 0, st, tlwe_key, tlwe_ntt, p);
    hipLaunchKernelGGL(k_tfhe_gen_bootkey, dim3((unsigned) (p.n * 2 * p.bk_l)), dim3(64), 0, st, boot_key, lwe_key,
                       tlwe_ntt, p, noise_c, seed, stream_a, stream_e);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_tfhe_lwe_phase(const int* __restrict__ a, const int* __restrict__ b,
                                                        const int* __restrict__ key, int* __restrict__ phase, int n)
{
    __shared__ u32 red[256];
    const u64 s = blockIdx.x;
    const int t = threadIdx.x;
    u32 acc = 0;
    for (int j = t; j < n; j += 256) acc += (u32) a[s * n + j] * (u32) key[j];
    red[t] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) phase[s] = (int) ((u32) b[s] - red[0]);
}

hipError_t tfhe_lwe_phase(const int* a, const int* b, const int* key, int* phase, int n, int shape, hipStream_t st)
{
    if (shape <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tfhe_lwe_phase, dim3(shape), dim3(256), 0, st, a, b, key, phase, n);
    return hipGetLastError();
}

// Synchronous (one-time): tries the FP64 layout, falls back to the integer
// layout when a key coefficient does not fit int32.
hipError_t tfhe_prepare_bootkey(const TfheDev& p, const u64* src, u64* dst, u64 polys, bool allow_fp, int* fmt_out,
                                hipStream_t st)
{
    hipError_t e;
    *fmt_out = 0;
    if (allow_fp) {
        if ((e = hipMemsetAsync(dst, 0, TFHE_PREP_HEADER * sizeof(u64), st)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_tfhe_prepare_bootkey_fp, dim3((unsigned) polys), dim3(64), 0, st, src,
                           dst + TFHE_PREP_HEADER, polys, p, reinterpret_cast<int*>(dst + 1));
        int bad = 0;
        if ((e = hipMemcpyAsync(&bad, dst + 1, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
        if (!bad) {
            hipLaunchKernelGGL(k_tfhe_set_header, dim3(1), dim3(1), 0, st, dst, (u64) 1);
            *fmt_out = 1;
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(k_tfhe_prepare_bootkey, dim3((unsigned) polys), dim3(256), 0, st, src, dst + TFHE_PREP_HEADER,
                       polys);
    hipLaunchKernelGGL(k_tfhe_set_header, dim3(1), dim3(1), 0, st, dst, (u64) 0);
    return hipGetLastError();
}

// The layout of the prepared key is its header word (1 = FP64, 0 = integer), read by the kernels themselves in stream
// order: BOTH are launched and the one whose layout is absent exits at once (an empty grid of `shape` workgroups, ~5 us).
// Round 4 picked the kernel on the host from a per-pointer cache filled by a synchronous read -- not ordered behind the
// caller's stream, stale after the buffer was overwritten or its address reused, and a mismatch was a silent return
// (ADVICE r4, medium).  A header that is neither layout sets the context's pinned flag; its next entry reports it.
#define TFHE_INT_GRID_MAX 2048
hipError_t tfhe_blind_rotate(const TfheDev& p, const int* in_a, const int* in_b, const u64* bk_prepared, int* out_a,
                             int* out_b, int encoded, int shape, hipStream_t st)
{
    if (shape <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tfhe_blind_rotate_fp, dim3(shape), dim3(TF_THREADS), 0, st, in_a, in_b, bk_prepared, out_a, out_b,
                       p, encoded);
    // (two workgroups per CU are resident: 512 on the chip; 2048 keeps the tail of a large batch balanced)
    hipLaunchKernelGGL(k_tfhe_blind_rotate, dim3(shape < TFHE_INT_GRID_MAX ? shape : TFHE_INT_GRID_MAX), dim3(TF_THREADS), 0, st,
                       in_a, in_b, bk_prepared, out_a, out_b, p, encoded, shape);
    return hipGetLastError();
}

hipError_t tfhe_gate_pre(int* out_a, int* out_b, const int* a1, const int* b1, const int* a2, const int* b2,
                         int encoded, int s1, int s2, int m, int n, int shape, hipStream_t st)
{
    hipLaunchKernelGGL(k_tfhe_gate_pre, dim3(shape), dim3(256), 0, st, out_a, out_b, a1, b1, a2, b2, encoded, s1, s2,
                       m, n);
    return hipGetLastError();
}

hipError_t tfhe_key_switching(const TfheDev& p, const int* in_a, const int* in_b, int* out_a, int* out_b,
                              const int* ks_a, const int* ks_b, int shape, int ks_batched, int ks_pieces, hipStream_t st)
{
    if (p.n > 512 || p.ks_length > 8) return hipErrorInvalidValue;
    if (shape <= 0) return hipSuccess;
    const int Nk = p.N * p.k;
    // ks_batched: -1 by launch size, 0 never, 1 always, 8 / 12 / 16 always with that many gates per workgroup;
    // ks_pieces: -1 by launch size, otherwise the number of workgroups the coefficient loop of a gate (group) is cut into
    const bool can_batch = p.ks_base_bit == 2 && p.n >= 256;
    const bool batch = can_batch && ks_batched != 0 && (ks_batched >= 1 || shape >= TFHE_KS_BATCH_MIN);
    if (batch) {
        // KS_GB gates per workgroup share the key rows.  A workgroup is a chain of iterations over the input coefficients
        // (~4 us each: the loads of an iteration depend on nothing, but its 24 row loads are all there is to overlap),
        // so throughput comes from MANY short chains: 16 gates per workgroup and the coefficient loop cut so that the launch
        // has ~8192 workgroups (sweep: profiles/r4_c5/ks_sweep2.txt -- 8192 gates: 1 piece 4.2 ms, 16 pieces 3.06 ms;
        // 1024 gates: 4.2 / 0.43 ms; more gates per workgroup or fewer pieces lose everywhere)
        const int gb = (ks_batched >= 8) ? ks_batched : 16;
        const int groups = (shape + gb - 1) / gb;
        int pieces = ks_pieces >= 1 ? ks_pieces : 8192 / groups;
        if (pieces < 1) pieces = 1;
        if (pieces > 64) pieces = 64;
        while (pieces > 1 && Nk / pieces < 16) pieces--;
        const int chunk = (Nk + pieces - 1) / pieces;
        pieces = (Nk + chunk - 1) / chunk;
        const dim3 grid((unsigned) groups, (unsigned) pieces);
        if (pieces > 1) {
            hipError_t e = hipMemsetAsync(out_a, 0, (size_t) shape * p.n * sizeof(int), st);
            if (e == hipSuccess) e = hipMemsetAsync(out_b, 0, (size_t) shape * sizeof(int), st);
            if (e != hipSuccess) return e;
        }
#define KS_LAUNCH(GB)                                                                                                      \
    do {                                                                                                                   \
        if (pieces > 1)                                                                                                    \
            hipLaunchKernelGGL((k_tfhe_key_switching_batched<GB, true>), grid, dim3(256), 0, st, in_a, in_b, out_a, out_b,  \
                               ks_a, ks_b, p.ks_length, p.n, Nk, shape, chunk);                                           \
        else                                                                                                               \
            hipLaunchKernelGGL((k_tfhe_key_switching_batched<GB, false>), grid, dim3(256), 0, st, in_a, in_b, out_a, out_b, \
                               ks_a, ks_b, p.ks_length, p.n, Nk, shape, chunk);                                           \
    } while (0)
        if (gb == 8) KS_LAUNCH(8);
        else if (gb == 12) KS_LAUNCH(12);
        else KS_LAUNCH(16);
#undef KS_LAUNCH
        return hipGetLastError();
    }
    // one gate per workgroup; fewer gates than resident workgroups (eight per CU): cut the coefficient loop, at least 16
    // coefficients a piece
    int pieces = 1;
    if (ks_pieces >= 1) pieces = ks_pieces > 64 ? 64 : ks_pieces;
    else
        while (pieces < 64 && (long) shape * pieces * 2 <= 2048 && Nk / (pieces * 2) >= 16) pieces *= 2;
    if (pieces > 1) {
        const int chunk = (Nk + pieces - 1) / pieces;
        pieces = (Nk + chunk - 1) / chunk;
        hipError_t e = hipMemsetAsync(out_a, 0, (size_t) shape * p.n * sizeof(int), st);
        if (e == hipSuccess) e = hipMemsetAsync(out_b, 0, (size_t) shape * sizeof(int), st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_tfhe_key_switching<true>, dim3(shape, pieces), dim3(256), 0, st, in_a, in_b, out_a, out_b, ks_a,
                           ks_b, p.ks_base_bit, p.ks_length, p.n, Nk, chunk);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_tfhe_key_switching<false>, dim3(shape), dim3(256), 0, st, in_a, in_b, out_a, out_b, ks_a, ks_b,
                       p.ks_base_bit, p.ks_length, p.n, Nk, Nk);
    return hipGetLastError();
}

} // namespace hegpu

// encode.hip -- CKKS encoder / decoder kernels for gfx950:
// the "special" FFT over the rotation group (replaces gpufft::GPU_Special_FFT of the
// unvendored thirdparty/GPU-FFT; algorithm as used by src/lib/host/ckks/encoder.cu:21-98:
// HEAAN's fftSpecial / fftSpecialInv), the double <-> RNS conversions and the CRT
// composition (src/lib/kernel/encoding.cu:143-392).  FP64 throughout, no FMA contraction
// (the library is built with -ffp-contract=off), so results are reproducible operation
// for operation by the CPU oracle.
#include "keygen.hpp"

namespace hegpu {

#define EN_THREADS 256

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub_(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }

// ---- the special FFT (round 6: on the tile scheme of ntt.hip instead of one global-memory launch per stage).
// slots <= 2^13 complex doubles (128 KiB) live in the LDS of ONE workgroup for all their stages: one launch, the
// message conversion (double -> complex on the way in for encode, complex -> double on the way out for decode) in its
// load / store.  Larger transforms (N = 2^15, 2^16) are cut as ntt.hip cuts a limb: the log2(slots) - 13 stages with
// strides >= 2^13 run in a register pass over global memory (every access a coalesced 16-byte-per-lane segment), the 13
// contiguous stages in LDS chunks of 2^13 -- two launches.  Two stages per exchange (radix-4 register rounds), one
// barrier per round.  Every butterfly performs exactly the operations of the stage-per-launch form it replaces
//   forward (decode):  u = a, v = b*w          -> (u + v, u - v)        (fftSpecial)
//   inverse (encode):  u = a + b, v = (a-b)*w  -> (u, v), the stride-1 stage scaled by fix  (fftSpecialInv)
// in the same order per element (no FMA contraction: -ffp-contract=off), so the result is bit-identical to the CPU
// oracle's loop (oracle/o_encode.c) whatever the grouping.
#define EN_LOG_CHUNK 13
#define EN_FFT_THREADS 1024

struct EnIo {
    const double* real_in; // encode: the message (nullptr: the transform reads `data`)
    int in_size, in_complex;
    double* real_out;      // decode: the message (nullptr: the transform writes `data`)
    int out_complex;
};
__device__ __forceinline__ cplx en_load(const cplx* __restrict__ v, const EnIo& io, int e)
{
    if (!io.real_in) return v[e];
    if (io.in_complex) return e < io.in_size ? reinterpret_cast<const cplx*>(io.real_in)[e] : cplx{0.0, 0.0};
    return {e < io.in_size ? io.real_in[e] : 0.0, 0.0};
}
__device__ __forceinline__ void en_store(cplx* __restrict__ v, const EnIo& io, int e, cplx z)
{
    if (!io.real_out) v[e] = z;
    else if (io.out_complex) reinterpret_cast<cplx*>(io.real_out)[e] = z;
    else io.real_out[e] = z.re;
}
template <bool INVERSE>
__device__ __forceinline__ void en_bfly(cplx& a, cplx& b, cplx w)
{
    if (INVERSE) {
        const cplx x = cadd(a, b), y = cmul(csub_(a, b), w);
        a = x;
        b = y;
    } else {
        const cplx bw = cmul(b, w);
        const cplx x = cadd(a, bw), y = csub_(a, bw);
        a = x;
        b = y;
    }
}
// two stages (half-lengths L and 2L) on the four elements i + k L, i = blk * 4L + j, j < L
template <bool INVERSE>
__device__ __forceinline__ void en_radix4(cplx (&e)[4], const cplx* __restrict__ roots, int L, int j)
{
    if (INVERSE) {
        en_bfly<true>(e[0], e[2], roots[2 * L + j]);
        en_bfly<true>(e[1], e[3], roots[2 * L + j + L]);
        const cplx w = roots[L + j];
        en_bfly<true>(e[0], e[1], w);
        en_bfly<true>(e[2], e[3], w);
    } else {
        const cplx w = roots[L + j];
        en_bfly<false>(e[0], e[1], w);
        en_bfly<false>(e[2], e[3], w);
        en_bfly<false>(e[0], e[2], roots[2 * L + j]);
        en_bfly<false>(e[1], e[3], roots[2 * L + j + L]);
    }
}

// the contiguous stages (half-lengths 1 .. 2^(logch-1)) of chunk blockIdx.x, resident in LDS
template <bool INVERSE>
__global__ __launch_bounds__(EN_FFT_THREADS) void k_en_fft_chunk(cplx* __restrict__ v, const cplx* __restrict__ roots,
                                                                 int logch, double fix, EnIo io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char en_lds_raw[];
    cplx* lds = reinterpret_cast<cplx*>(en_lds_raw);
    const int CH = 1 << logch, t = threadIdx.x, T = blockDim.x;
    const int base = blockIdx.x << logch;
    for (int e = t; e < CH; e += T) lds[e] = en_load(v, io, base + e);
    __syncthreads();
    const bool odd = logch & 1;
    if (INVERSE && odd) { // the single stage first: half-length CH / 2
        const int L = CH >> 1;
        for (int q = t; q < L; q += T) en_bfly<true>(lds[q], lds[q + L], roots[L + q]);
        __syncthreads();
    }
    // Radix-4 rounds.  The three twiddles of a butterfly group depend on (round, thread) only, not on the data: those of
    // round r + 1 are REQUESTED BEFORE the barrier that ends round r, so their L2 latency (the table is 16 bytes per
    // entry, far too large to park next to 128 KiB of data) runs under the exchange instead of in front of every round.
    // EN_FFT_ITERS groups per thread and round (CH / 4 / threads <= 2).
    const int rounds = logch >> 1;
    auto round_s = [&](int r) { return INVERSE ? 2 * (rounds - 1 - r) : 2 * r; }; // forward: (1,2), (4,8), ...; inverse: downwards to (2,1)
    constexpr int ITERS = 2;
    cplx tw[ITERS][3];
    auto request = [&](int r) {
        const int s = round_s(r), L = 1 << s;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int q = t + it * T;
            if (q < (CH >> 2)) {
                const int j = q & (L - 1);
                tw[it][0] = roots[L + j];
                tw[it][1] = roots[2 * L + j];
                tw[it][2] = roots[2 * L + j + L];
            }
        }
    };
    if (rounds > 0) request(0);
    for (int r = 0; r < rounds; r++) {
        const int s = round_s(r), L = 1 << s;
        cplx e[ITERS][4];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int q = t + it * T;
            if (q < (CH >> 2)) {
                const int j = q & (L - 1), i = ((q >> s) << (s + 2)) + j;
#pragma unroll
                for (int k = 0; k < 4; k++) e[it][k] = lds[i + k * L];
                if (INVERSE) {
                    en_bfly<true>(e[it][0], e[it][2], tw[it][1]);
                    en_bfly<true>(e[it][1], e[it][3], tw[it][2]);
                    en_bfly<true>(e[it][0], e[it][1], tw[it][0]);
                    en_bfly<true>(e[it][2], e[it][3], tw[it][0]);
                    if (L == 1) { // the stride-1 stage carries the scale (fftSpecialInv)
#pragma unroll
                        for (int k = 0; k < 4; k++) e[it][k] = {e[it][k].re * fix, e[it][k].im * fix};
                    }
                } else {
                    en_bfly<false>(e[it][0], e[it][1], tw[it][0]);
                    en_bfly<false>(e[it][2], e[it][3], tw[it][0]);
                    en_bfly<false>(e[it][0], e[it][2], tw[it][1]);
                    en_bfly<false>(e[it][1], e[it][3], tw[it][2]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) lds[i + k * L] = e[it][k];
            }
        }
        if (r + 1 < rounds) request(r + 1);
        __syncthreads();
    }
    if (!INVERSE && odd) {
        const int L = CH >> 1;
        for (int q = t; q < L; q += T) en_bfly<false>(lds[q], lds[q + L], roots[L + q]);
        __syncthreads();
    }
    for (int e = t; e < CH; e += T) en_store(v, io, base + e, lds[e]);
}

// the REM = 1 or 2 stages whose half-lengths are >= 2^logch, in registers over global memory
template <bool INVERSE, int REM>
__global__ __launch_bounds__(EN_THREADS) void k_en_fft_outer(cplx* __restrict__ v, const cplx* __restrict__ roots, int logch,
                                                             EnIo io)
{
    const int q = blockIdx.x * EN_THREADS + threadIdx.x;
    const int L = 1 << logch;
    if constexpr (REM == 2) {
        const int j = q & (L - 1), i = ((q >> logch) << (logch + 2)) + j;
        cplx e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = en_load(v, io, i + k * L);
        en_radix4<INVERSE>(e, roots, L, j);
#pragma unroll
        for (int k = 0; k < 4; k++) en_store(v, io, i + k * L, e[k]);
    } else {
        const int j = q & (L - 1), i = ((q >> logch) << (logch + 1)) + j;
        cplx a = en_load(v, io, i), b = en_load(v, io, i + L);
        en_bfly<INVERSE>(a, b, roots[L + j]);
        en_store(v, io, i, a);
        en_store(v, io, i + L, b);
    }
}

// `real_in` (encode) / `real_out` (decode): the message itself, converted in the first load / the last store of the
// transform (k_en_double_to_complex / k_en_complex_to_double of rounds 1-5, reference encoder.cu:120 / :505)
hipError_t en_special_fft(void* data, const void* roots, int log_slots, bool inverse, double fix, hipStream_t st,
                          const double* real_in, int in_size, int in_complex, double* real_out, int out_complex)
{
    if (log_slots < 2 || log_slots > EN_LOG_CHUNK + 2) return hipErrorInvalidValue;
    const int logch = log_slots < EN_LOG_CHUNK ? log_slots : EN_LOG_CHUNK, rem = log_slots - logch;
    const int chunks = 1 << rem, ch = 1 << logch;
    const int threads = (ch >> 2) < EN_FFT_THREADS ? (ch >> 2) : EN_FFT_THREADS;
    const size_t lds_bytes = (size_t) ch * sizeof(cplx);
    static const hipError_t a0 = hipFuncSetAttribute((const void*) k_en_fft_chunk<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int) (sizeof(cplx) << EN_LOG_CHUNK));
    static const hipError_t a1 = hipFuncSetAttribute((const void*) k_en_fft_chunk<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int) (sizeof(cplx) << EN_LOG_CHUNK));
    (void) a0;
    (void) a1;
    const EnIo plain{nullptr, 0, 0, nullptr, 0};
    const EnIo in_io{real_in, in_size, in_complex, nullptr, 0}, out_io{nullptr, 0, 0, real_out, out_complex};
    cplx* v = (cplx*) data;
    const cplx* r = (const cplx*) roots;
    const int outer_grid = ((1 << log_slots) >> (rem == 2 ? 2 : 1)) / EN_THREADS;
    if (!inverse) {
        // contiguous stages first, then the wide ones; the message leaves in the last store
        hipLaunchKernelGGL(k_en_fft_chunk<false>, dim3(chunks), dim3(threads), lds_bytes, st, v, r, logch, 1.0, rem ? plain : out_io);
        if (rem == 1) hipLaunchKernelGGL((k_en_fft_outer<false, 1>), dim3(outer_grid), dim3(EN_THREADS), 0, st, v, r, logch, out_io);
        if (rem == 2) hipLaunchKernelGGL((k_en_fft_outer<false, 2>), dim3(outer_grid), dim3(EN_THREADS), 0, st, v, r, logch, out_io);
    } else {
        if (rem == 1) hipLaunchKernelGGL((k_en_fft_outer<true, 1>), dim3(outer_grid), dim3(EN_THREADS), 0, st, v, r, logch, in_io);
        if (rem == 2) hipLaunchKernelGGL((k_en_fft_outer<true, 2>), dim3(outer_grid), dim3(EN_THREADS), 0, st, v, r, logch, in_io);
        hipLaunchKernelGGL(k_en_fft_chunk<true>, dim3(chunks), dim3(threads), lds_bytes, st, v, r, logch, fix, rem ? plain : in_io);
    }
    return hipGetLastError();
}

// round(x) -> residues; |x| < 2^128 (encode_kernel_ckks_conversion, encoding.cu:166-232)
__device__ __forceinline__ void en_store_rns(u64* __restrict__ plain, u64 at, double value, const Mod* __restrict__ mods,
                                             int limbs, int n_power)
{
    double c = round(value);
    const bool neg = signbit(c);
    c = fabs(c);
    const double two64 = 18446744073709551616.0;
    const u64 lo = (u64) fmod(c, two64), hi = (u64) (c / two64);
    for (int i = 0; i < limbs; i++) {
        const Mod m = mods[i];
        const u64 r = reduce128(hi, lo, m);
        plain[at + ((u64) i << n_power)] = neg ? sub_mod(m.q, r, m.q) : r;
    }
}

__global__ __launch_bounds__(EN_THREADS) void k_en_conversion(u64* __restrict__ plain, const cplx* __restrict__ msg,
                                                              const Mod* __restrict__ mods, int limbs,
                                                              const int* __restrict__ reverse_order, int n_power)
{
    // one thread per VALUE: coefficient v = the real part of slot v (v < N/2) or the imaginary part of slot v - N/2
    const int v = blockIdx.x * EN_THREADS + threadIdx.x;
    const int slots = 1 << (n_power - 1);
    const double* z = reinterpret_cast<const double*>(msg + reverse_order[v & (slots - 1)]);
    en_store_rns(plain, (u64) v, z[v >= slots ? 1 : 0], mods, limbs, n_power);
}

hipError_t en_conversion(u64* plain, const void* msg, const Mod* mods, int limbs, const int* reverse_order, int n_power,
                         hipStream_t st)
{
    hipLaunchKernelGGL(k_en_conversion, dim3((1u << n_power) / EN_THREADS), dim3(EN_THREADS), 0, st, plain,
                       (const cplx*) msg, mods, limbs, reverse_order, n_power);
    return hipGetLastError();
}

// encode_kernel_coeff_ckks_conversion (encoding.cu:106-137): coefficient idx = round(message[idx] * scale);
// encode_kernel_double_ckks_conversion (:43-77): every entry = round(value) (message == nullptr)
__global__ __launch_bounds__(EN_THREADS) void k_en_coeff_conversion(u64* __restrict__ plain,
                                                                    const double* __restrict__ message, int size,
                                                                    double scale_or_value, const Mod* __restrict__ mods,
                                                                    int limbs, int n_power)
{
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x;
    const double v = message ? (idx < size ? message[idx] : 0.0) * scale_or_value : scale_or_value;
    en_store_rns(plain, (u64) idx, v, mods, limbs, n_power);
}

hipError_t en_coeff_conversion(u64* plain, const double* message, int size, double scale_or_value, const Mod* mods,
                               int limbs, int n_power, hipStream_t st)
{
    hipLaunchKernelGGL(k_en_coeff_conversion, dim3((1u << n_power) / EN_THREADS), dim3(EN_THREADS), 0, st, plain, message,
                       size, scale_or_value, mods, limbs, n_power);
    return hipGetLastError();
}

// ---- CRT composition (encode_kernel_compose, encoding.cu:234-383; biginteger helpers
// util/bigintegerarith.cuh): little-endian 64-bit words, at most EN_MAX_WORDS of them
#define EN_MAX_WORDS 64

// one wavefront per workgroup: a thread is one long dependent chain -- l mul_barrett + l x l multiply-accumulate words --
// so the launch is spread over as many CUs as it has wavefronts (N = 2^14: 256 workgroups, one per CU)
#define EN_COMPOSE_THREADS 64
// LMAX: compile-time bound on the word count l, so that the accumulator lives in registers (every index below is static:
// the word loops are fully unrolled with `k < l` guards).  Rounds 1-5 kept acc[64] in scratch memory and ran two values
// per thread: 86 us at N = 2^14 -- more than the rest of a decode together.  The integers are the same whatever the
// order of evaluation (each partial sum is brought below M, its canonical value), and the final conversion adds the words
// in the reference's order.
template <int LMAX>
__device__ __forceinline__ double en_compose_one(const u64* __restrict__ plain, u64 at, const Mod* __restrict__ mods,
                                                 const u64* __restrict__ Mi_inv, const u64* __restrict__ Mi,
                                                 const u64* __restrict__ upper_half, const u64* __restrict__ M, int l,
                                                 double inv_scale, int n_power, u64* tl)
{
    // The l residues of the value are requested TOGETHER (unrolled, independent loads) and their products with Mi_inv parked
    // in the thread's own column of `tl` -- inside the word loop below each would cost a full memory latency per limb
    // (measured: 26 us of a 28 us kernel at N = 2^14).  A thread reads back only what it wrote itself: no barrier.
#pragma unroll
    for (int k = 0; k < LMAX; k++) {
        if (k < l) tl[k * EN_COMPOSE_THREADS + threadIdx.x] = mul_barrett(plain[at + ((u64) k << n_power)], Mi_inv[k], mods[k]);
    }
    // Round 6: the l terms are summed WITHOUT a comparison / subtraction after each (the reference brings every partial sum
    // below M: l compares + up to l subtractions of l words each, 40 % of the instructions at l = 9).  The sum is below
    // l M < 2^(64 l + 6): one more word (`top`).  Its quotient by M is floor(sum_i t_i / q_i) -- sum_i t_i M / q_i over M -- which
    // a double-precision sum of the l fractions gives to within one (absolute error below l 2^-51: wrong only when the true
    // sum sits that close to an integer, and then by exactly one); ONE multiply-subtract of ke M and one correction (add M
    // back if the difference went negative, subtract M once more if it is still >= M) leave the canonical value in [0, M) --
    // the same integer the reference's chain of reductions ends with.
    u64 acc[LMAX], top = 0;
#pragma unroll
    for (int k = 0; k < LMAX; k++) acc[k] = 0;
    double tf = 0.0;
    for (int i = 0; i < l; i++) {
        const u64 t = tl[i * EN_COMPOSE_THREADS + threadIdx.x];
        tf += (double) t * (1.0 / (double) mods[i].q);
        const u64* mi = Mi + (u64) i * l;
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < LMAX; k++) {
            if (k < l) {
                u64 hi, lo;
                mul64wide(mi[k], t, hi, lo);
                const u64 s1 = lo + carry;
                const u64 c1 = s1 < lo;
                const u64 s2 = acc[k] + s1;
                const u64 c2 = s2 < s1;
                acc[k] = s2;
                carry = hi + c1 + c2;
            }
        }
        top += carry;
    }
    {
        const u64 ke = (u64) tf; // <= l
        u64 borrow = 0, mcarry = 0;
#pragma unroll
        for (int k = 0; k < LMAX; k++) {
            if (k < l) {
                u64 hi, lo;
                mul64wide(M[k], ke, hi, lo);
                const u64 sub = lo + mcarry; // word k of ke M
                mcarry = hi + (sub < lo);
                const u64 d = acc[k] - sub;
                const u64 b1 = acc[k] < sub;
                const u64 d2 = d - borrow;
                const u64 b2 = d < borrow;
                acc[k] = d2;
                borrow = b1 | b2;
            }
        }
        top = top - mcarry - borrow; // 0, or all ones when ke was one too large
        bool geq = true, decided = false; // acc >= M ?  (most significant differing word decides)
#pragma unroll
        for (int k = LMAX - 1; k >= 0; k--) {
            if (k < l && !decided && acc[k] != M[k]) {
                geq = acc[k] > M[k];
                decided = true;
            }
        }
        const bool negative = top != 0;
        if (negative || geq) { // acc += M  or  acc -= M
            u64 c = 0;
#pragma unroll
            for (int k = 0; k < LMAX; k++) {
                if (k < l) {
                    const u64 m = M[k];
                    if (negative) {
                        const u64 s1 = acc[k] + m;
                        const u64 c1 = s1 < m;
                        const u64 s2 = s1 + c;
                        const u64 c2 = s2 < s1;
                        acc[k] = s2;
                        c = c1 | c2;
                    } else {
                        const u64 d = acc[k] - m;
                        const u64 b1 = acc[k] < m;
                        const u64 d2 = d - c;
                        const u64 b2 = d < c;
                        acc[k] = d2;
                        c = b1 | b2;
                    }
                }
            }
        }
    }
    bool upper = true, decided = false;
#pragma unroll
    for (int k = LMAX - 1; k >= 0; k--) {
        if (k < l && !decided && acc[k] != upper_half[k]) {
            upper = acc[k] > upper_half[k];
            decided = true;
        }
    }
    const double two64 = 18446744073709551616.0;
    double result = 0.0, w = inv_scale;
    if (upper) {
        // negative value: word-wise difference to M, exactly as the reference accumulates it
#pragma unroll
        for (int j = 0; j < LMAX; j++) {
            if (j < l) {
                const u64 m = M[j];
                if (acc[j] > m) {
                    const u64 diff = acc[j] - m;
                    result += diff ? (double) diff * w : 0.0;
                } else {
                    const u64 diff = m - acc[j];
                    result -= diff ? (double) diff * w : 0.0;
                }
                w *= two64;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < LMAX; j++) {
            if (j < l) {
                result += acc[j] ? (double) acc[j] * w : 0.0;
                w *= two64;
            }
        }
    }
    return result;
}

// one thread per VALUE (coefficient v of the plaintext: the real part of slot v for v < N/2, the imaginary part of slot
// v - N/2 otherwise): N threads instead of N/2 chains twice as long
// (one wavefront per workgroup, see EN_COMPOSE_THREADS)
template <int LMAX>
__global__ __launch_bounds__(EN_COMPOSE_THREADS) void k_en_compose(cplx* __restrict__ msg, const u64* __restrict__ plain,
                                                           const Mod* __restrict__ mods, const u64* __restrict__ Mi_inv,
                                                           const u64* __restrict__ Mi,
                                                           const u64* __restrict__ upper_half,
                                                           const u64* __restrict__ M, int l, double scale,
                                                           const int* __restrict__ reverse_order, int n_power)
{
    const int v = blockIdx.x * EN_COMPOSE_THREADS + threadIdx.x;
    const int slots = 1 << (n_power - 1), slot = v & (slots - 1);
    const double inv_scale = 1.0 / scale;
    __shared__ u64 tl[LMAX * EN_COMPOSE_THREADS];
    const double r = en_compose_one<LMAX>(plain, (u64) v, mods, Mi_inv, Mi, upper_half, M, l, inv_scale, n_power, tl);
    double* out = reinterpret_cast<double*>(msg + reverse_order[slot]);
    out[v >= slots ? 1 : 0] = r;
}

hipError_t en_compose(void* msg, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                      const u64* upper_half, const u64* M, int l, double scale, const int* reverse_order, int n_power,
                      hipStream_t st)
{
    if (l > EN_MAX_WORDS) return hipErrorInvalidValue;
    const dim3 grid((1u << n_power) / EN_COMPOSE_THREADS), block(EN_COMPOSE_THREADS);
#define EN_COMPOSE(LM) hipLaunchKernelGGL(k_en_compose<LM>, grid, block, 0, st, (cplx*) msg, plain, mods, Mi_inv, Mi, upper_half, M, l, scale, reverse_order, n_power)
    if (l <= 8) EN_COMPOSE(8);
    else if (l <= 16) EN_COMPOSE(16);
    else if (l <= 32) EN_COMPOSE(32);
    else EN_COMPOSE(EN_MAX_WORDS);
#undef EN_COMPOSE
    return hipGetLastError();
}

// decode_kernel_coeff_ckks_compose (encoding.cu:387-464): one real value per coefficient
template <int LMAX>
__global__ __launch_bounds__(EN_COMPOSE_THREADS) void k_en_coeff_compose(double* __restrict__ message,
                                                                 const u64* __restrict__ plain,
                                                                 const Mod* __restrict__ mods,
                                                                 const u64* __restrict__ Mi_inv,
                                                                 const u64* __restrict__ Mi,
                                                                 const u64* __restrict__ upper_half,
                                                                 const u64* __restrict__ M, int l, double scale,
                                                                 int n_power)
{
    const int idx = blockIdx.x * EN_COMPOSE_THREADS + threadIdx.x;
    __shared__ u64 tl[LMAX * EN_COMPOSE_THREADS];
    message[idx] = en_compose_one<LMAX>(plain, (u64) idx, mods, Mi_inv, Mi, upper_half, M, l, 1.0 / scale, n_power, tl);
}

hipError_t en_coeff_compose(double* message, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                            const u64* upper_half, const u64* M, int l, double scale, int n_power, hipStream_t st)
{
    if (l > EN_MAX_WORDS) return hipErrorInvalidValue;
    const dim3 grid((1u << n_power) / EN_COMPOSE_THREADS), block(EN_COMPOSE_THREADS);
#define EN_COMPOSE(LM) hipLaunchKernelGGL(k_en_coeff_compose<LM>, grid, block, 0, st, message, plain, mods, Mi_inv, Mi, upper_half, M, l, scale, n_power)
    if (l <= 8) EN_COMPOSE(8);
    else if (l <= 16) EN_COMPOSE(16);
    else if (l <= 32) EN_COMPOSE(32);
    else EN_COMPOSE(EN_MAX_WORDS);
#undef EN_COMPOSE
    return hipGetLastError();
}

} // namespace hegpu

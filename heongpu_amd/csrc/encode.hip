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

// message [size] -> (message, 0) padded with zeros to `slots`; complex input: (re, im) pairs
__global__ __launch_bounds__(EN_THREADS) void k_en_double_to_complex(const double* __restrict__ in, int size,
                                                                     cplx* __restrict__ out, int complex_in)
{
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x;
    if (complex_in) out[idx] = idx < size ? ((const cplx*) in)[idx] : cplx{0.0, 0.0};
    else out[idx] = {idx < size ? in[idx] : 0.0, 0.0};
}
__global__ __launch_bounds__(EN_THREADS) void k_en_complex_to_double(const cplx* __restrict__ in,
                                                                     double* __restrict__ out, int complex_out)
{
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x;
    if (complex_out) ((cplx*) out)[idx] = in[idx];
    else out[idx] = in[idx].re;
}

// One butterfly stage with block length len = 2*lenh; thread = one butterfly.
// forward (decode):  u = a, v = b*w        -> (u + v, u - v)          (fftSpecial)
// inverse (encode):  u = a + b, v = (a-b)*w -> (u, v), last stage scaled by fix (fftSpecialInv)
template <bool INVERSE>
__global__ __launch_bounds__(EN_THREADS) void k_en_fft_stage(cplx* __restrict__ v, const cplx* __restrict__ roots,
                                                             int lenh, double fix, int last)
{
    const int t = blockIdx.x * EN_THREADS + threadIdx.x; // butterfly index
    const int j = t & (lenh - 1), blk = t / lenh;
    const int i0 = blk * 2 * lenh + j, i1 = i0 + lenh;
    const cplx w = roots[lenh + j];
    const cplx a = v[i0], b = v[i1];
    cplx x, y;
    if (INVERSE) {
        x = cadd(a, b);
        y = cmul(csub_(a, b), w);
        if (last) {
            x = {x.re * fix, x.im * fix};
            y = {y.re * fix, y.im * fix};
        }
    } else {
        const cplx bw = cmul(b, w);
        x = cadd(a, bw);
        y = csub_(a, bw);
    }
    v[i0] = x;
    v[i1] = y;
}

hipError_t en_special_fft(void* data, const void* roots, int log_slots, bool inverse, double fix, hipStream_t st)
{
    const int slots = 1 << log_slots;
    const int grid = (slots / 2 + EN_THREADS - 1) / EN_THREADS;
    if (slots / 2 < EN_THREADS) return hipErrorInvalidValue;
    if (!inverse) {
        for (int lenh = 1; lenh < slots; lenh <<= 1)
            hipLaunchKernelGGL(k_en_fft_stage<false>, dim3(grid), dim3(EN_THREADS), 0, st, (cplx*) data,
                               (const cplx*) roots, lenh, 1.0, 0);
    } else {
        for (int lenh = slots >> 1; lenh >= 1; lenh >>= 1)
            hipLaunchKernelGGL(k_en_fft_stage<true>, dim3(grid), dim3(EN_THREADS), 0, st, (cplx*) data,
                               (const cplx*) roots, lenh, fix, lenh == 1);
    }
    return hipGetLastError();
}

hipError_t en_double_to_complex(const double* in, int size, void* out, int slots, bool complex_in, hipStream_t st)
{
    hipLaunchKernelGGL(k_en_double_to_complex, dim3(slots / EN_THREADS), dim3(EN_THREADS), 0, st, in, size, (cplx*) out,
                       (int) complex_in);
    return hipGetLastError();
}
hipError_t en_complex_to_double(const void* in, double* out, int slots, bool complex_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_en_complex_to_double, dim3(slots / EN_THREADS), dim3(EN_THREADS), 0, st, (const cplx*) in, out,
                       (int) complex_out);
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
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x; // slot
    const cplx z = msg[reverse_order[idx]];
    en_store_rns(plain, (u64) idx, z.re, mods, limbs, n_power);
    en_store_rns(plain, (u64) idx + (1u << (n_power - 1)), z.im, mods, limbs, n_power);
}

hipError_t en_conversion(u64* plain, const void* msg, const Mod* mods, int limbs, const int* reverse_order, int n_power,
                         hipStream_t st)
{
    hipLaunchKernelGGL(k_en_conversion, dim3((1u << (n_power - 1)) / EN_THREADS), dim3(EN_THREADS), 0, st, plain,
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

__device__ __forceinline__ bool big_geq(const u64* a, const u64* b, int n)
{
    for (int k = n - 1; k >= 0; k--) {
        if (a[k] != b[k]) return a[k] > b[k];
    }
    return true;
}

__device__ double en_compose_one(const u64* __restrict__ plain, u64 at, const Mod* __restrict__ mods,
                                 const u64* __restrict__ Mi_inv, const u64* __restrict__ Mi,
                                 const u64* __restrict__ upper_half, const u64* __restrict__ M, int l, double inv_scale,
                                 int n_power)
{
    u64 acc[EN_MAX_WORDS];
    for (int k = 0; k < l; k++) acc[k] = 0;
    for (int i = 0; i < l; i++) {
        const u64 t = mul_barrett(plain[at + ((u64) i << n_power)], Mi_inv[i], mods[i]);
        // acc += Mi[i] * t  (l words; the sum stays below 2*M < 2^(64 l))
        const u64* mi = Mi + (u64) i * l;
        u64 carry = 0;
        for (int k = 0; k < l; k++) {
            u64 hi, lo;
            mul64wide(mi[k], t, hi, lo);
            const u64 s1 = lo + carry;
            const u64 c1 = s1 < lo;
            const u64 s2 = acc[k] + s1;
            const u64 c2 = s2 < s1;
            acc[k] = s2;
            carry = hi + c1 + c2;
        }
        if (big_geq(acc, M, l)) {
            u64 borrow = 0;
            for (int k = 0; k < l; k++) {
                const u64 d = acc[k] - M[k];
                const u64 b1 = acc[k] < M[k];
                const u64 d2 = d - borrow;
                const u64 b2 = d < borrow;
                acc[k] = d2;
                borrow = b1 | b2;
            }
        }
    }
    const double two64 = 18446744073709551616.0;
    double result = 0.0, w = inv_scale;
    if (big_geq(acc, upper_half, l)) {
        // negative value: word-wise difference to M, exactly as the reference accumulates it
        for (int j = 0; j < l; j++, w *= two64) {
            if (acc[j] > M[j]) {
                const u64 diff = acc[j] - M[j];
                result += diff ? (double) diff * w : 0.0;
            } else {
                const u64 diff = M[j] - acc[j];
                result -= diff ? (double) diff * w : 0.0;
            }
        }
    } else {
        for (int j = 0; j < l; j++, w *= two64) result += acc[j] ? (double) acc[j] * w : 0.0;
    }
    return result;
}

__global__ __launch_bounds__(EN_THREADS) void k_en_compose(cplx* __restrict__ msg, const u64* __restrict__ plain,
                                                           const Mod* __restrict__ mods, const u64* __restrict__ Mi_inv,
                                                           const u64* __restrict__ Mi,
                                                           const u64* __restrict__ upper_half,
                                                           const u64* __restrict__ M, int l, double scale,
                                                           const int* __restrict__ reverse_order, int n_power)
{
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x;
    const double inv_scale = 1.0 / scale;
    cplx z;
    z.re = en_compose_one(plain, (u64) idx, mods, Mi_inv, Mi, upper_half, M, l, inv_scale, n_power);
    z.im = en_compose_one(plain, (u64) idx + (1u << (n_power - 1)), mods, Mi_inv, Mi, upper_half, M, l, inv_scale,
                          n_power);
    msg[reverse_order[idx]] = z;
}

hipError_t en_compose(void* msg, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                      const u64* upper_half, const u64* M, int l, double scale, const int* reverse_order, int n_power,
                      hipStream_t st)
{
    if (l > EN_MAX_WORDS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_en_compose, dim3((1u << (n_power - 1)) / EN_THREADS), dim3(EN_THREADS), 0, st, (cplx*) msg,
                       plain, mods, Mi_inv, Mi, upper_half, M, l, scale, reverse_order, n_power);
    return hipGetLastError();
}

// decode_kernel_coeff_ckks_compose (encoding.cu:387-464): one real value per coefficient
__global__ __launch_bounds__(EN_THREADS) void k_en_coeff_compose(double* __restrict__ message,
                                                                 const u64* __restrict__ plain,
                                                                 const Mod* __restrict__ mods,
                                                                 const u64* __restrict__ Mi_inv,
                                                                 const u64* __restrict__ Mi,
                                                                 const u64* __restrict__ upper_half,
                                                                 const u64* __restrict__ M, int l, double scale,
                                                                 int n_power)
{
    const int idx = blockIdx.x * EN_THREADS + threadIdx.x;
    message[idx] = en_compose_one(plain, (u64) idx, mods, Mi_inv, Mi, upper_half, M, l, 1.0 / scale, n_power);
}

hipError_t en_coeff_compose(double* message, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                            const u64* upper_half, const u64* M, int l, double scale, int n_power, hipStream_t st)
{
    if (l > EN_MAX_WORDS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_en_coeff_compose, dim3((1u << n_power) / EN_THREADS), dim3(EN_THREADS), 0, st, message, plain,
                       mods, Mi_inv, Mi, upper_half, M, l, scale, n_power);
    return hipGetLastError();
}

} // namespace hegpu

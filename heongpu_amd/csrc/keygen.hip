// keygen.hip -- key generation / encryption / decryption kernels for gfx950
// (SURVEY.md 8f next-1).  All of them are element-wise streams over
// [poly][limb][N] arrays (HBM-bound); the transforms between them are the NTT
// kernels of the hot path.
#include "keygen.hpp"

namespace hegpu {

#define KG_THREADS 256

__global__ __launch_bounds__(KG_THREADS) void k_kg_uniform(u64* __restrict__ out, const Mod* __restrict__ mods,
                                                           int n_power, int limbs, DrbgKey seed, u64 stream)
{
    const u64 n = (u64) blockIdx.x * KG_THREADS + threadIdx.x;
    const int limb = blockIdx.y, poly = blockIdx.z;
    const u64 e = (((u64) poly * limbs + limb) << n_power) + n;
    out[e] = drbg_uniform(seed, stream, e, mods[limb]);
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_gaussian(u64* __restrict__ out, const Mod* __restrict__ mods,
                                                            int n_power, int limbs, DrbgKey seed, u64 stream,
                                                            GaussCdt cdt)
{
    const u64 n = (u64) blockIdx.x * KG_THREADS + threadIdx.x;
    const int poly = blockIdx.y;
    const int v = drbg_gaussian(seed, stream, ((u64) poly << n_power) + n, cdt);
    for (int j = 0; j < limbs; j++) out[(((u64) poly * limbs + j) << n_power) + n] = lift_small(v, mods[j].q);
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_ternary(u64* __restrict__ out, const Mod* __restrict__ mods,
                                                           int n_power, int limbs, DrbgKey seed, u64 stream)
{
    const u64 n = (u64) blockIdx.x * KG_THREADS + threadIdx.x;
    const int poly = blockIdx.y;
    const int v = drbg_ternary(seed, stream, ((u64) poly << n_power) + n);
    for (int j = 0; j < limbs; j++) out[(((u64) poly * limbs + j) << n_power) + n] = lift_small(v, mods[j].q);
}

hipError_t kg_uniform(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                      hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_uniform, dim3((1u << n_power) / KG_THREADS, limbs, polys), dim3(KG_THREADS), 0, st, out,
                       mods, n_power, limbs, seed, stream);
    return hipGetLastError();
}
hipError_t kg_gaussian(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                       const GaussCdt& cdt, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_gaussian, dim3((1u << n_power) / KG_THREADS, polys), dim3(KG_THREADS), 0, st, out, mods,
                       n_power, limbs, seed, stream, cdt);
    return hipGetLastError();
}
hipError_t kg_ternary(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                      hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_ternary, dim3((1u << n_power) / KG_THREADS, polys), dim3(KG_THREADS), 0, st, out, mods,
                       n_power, limbs, seed, stream);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_secret_rns(const int* __restrict__ positions,
                                                              const int* __restrict__ values, int count,
                                                              u64* __restrict__ out, const Mod* __restrict__ mods,
                                                              int n_power, int limbs)
{
    const int i = blockIdx.x * KG_THREADS + threadIdx.x;
    if (i >= count) return;
    const int pos = positions[i], v = values[i];
    for (int j = 0; j < limbs; j++) out[((u64) j << n_power) + pos] = lift_small(v, mods[j].q);
}

hipError_t kg_secret_rns(const int* positions, const int* values, int count, u64* out, const Mod* mods, int n_power,
                         int limbs, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0, ((size_t) limbs << n_power) * sizeof(u64), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_kg_secret_rns, dim3((count + KG_THREADS - 1) / KG_THREADS), dim3(KG_THREADS), 0, st,
                       positions, values, count, out, mods, n_power, limbs);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_publickey(u64* __restrict__ pk, const u64* __restrict__ sk,
                                                             const u64* __restrict__ e, const u64* __restrict__ a,
                                                             const Mod* __restrict__ mods, int n_power, int limbs)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    const Mod m = mods[blockIdx.y];
    const u64 av = a[loc];
    u64 t = mul_barrett(sk[loc], av, m);
    t = add_mod(t, e[loc], m.q);
    pk[loc] = sub_mod(0, t, m.q);
    pk[loc + ((u64) limbs << n_power)] = av;
}

hipError_t kg_publickey(u64* pk, const u64* sk, const u64* e, const u64* a, const Mod* mods, int n_power, int limbs,
                        hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_publickey, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, pk, sk, e,
                       a, mods, n_power, limbs);
    return hipGetLastError();
}

__device__ __forceinline__ u32 bitrev(u32 v, int bits) { return __brev(v) >> (32 - bits); }

// slot permutation of an NTT-domain polynomial under X -> X^g (keygeneration.cu:742-755)
__device__ __forceinline__ u32 ntt_permutation(u32 index, u32 galois_elt, int n_power)
{
    const u32 n = 1u << n_power;
    const u32 reversed = bitrev(index + n, n_power + 1);
    const u32 raw = ((galois_elt * reversed) >> 1) & (n - 1);
    return bitrev(raw, n_power);
}

// One kernel for the six generators of the reference (keygeneration.cu relinkey_gen_kernel :145,
// relinkey_gen_II_kernel :584, galoiskey_gen_kernel :757, galoiskey_gen_II_kernel :807,
// switchkey_gen_kernel :896, switchkey_gen_II_kernel :941).  Digit i of the key is
//   ( -(s' * a_i + e_i) + [limb y belongs to digit i] * carried * P  ,  a_i )   over the Q' limbs,
// with (s', carried) = (s, s^2) relinearisation, (s o g^-1, s) Galois, (s_new, s_old) switch key.
// Method I: digits = Q, digit_width = 1, P = the one special prime; method II: digits of
// digit_width primes (Sk_pair = y / width; special limbs belong to no digit), P = product of the
// special primes, multiplied in one factor at a time like the reference.
__global__ __launch_bounds__(KG_THREADS) void k_kg_switchkey(u64* __restrict__ key, const u64* __restrict__ sk,
                                                             const u64* __restrict__ e, const u64* __restrict__ a,
                                                             const Mod* __restrict__ mods,
                                                             const u64* __restrict__ factor, int galois_elt,
                                                             const u64* __restrict__ old_sk, int n_power, int limbs,
                                                             int digits, int digit_width, int q_size, int p_size)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const int y = blockIdx.y;
    const Mod m = mods[y];
    const u64 s = sk[idx + ((u64) y << n_power)];
    const u64 sp = galois_elt ? sk[((u64) y << n_power) + ntt_permutation(idx, (u32) galois_elt, n_power)] : s;
    u64 carried = old_sk ? old_sk[idx + ((u64) y << n_power)] : (galois_elt ? s : mul_barrett(s, s, m));
    const int own = (y < q_size) ? y / digit_width : -1;
    if (own >= 0)
        for (int j = 0; j < p_size; j++) carried = mul_barrett(carried, factor[j * q_size + y], m);
    for (int i = 0; i < digits; i++) {
        const u64 src = idx + ((u64) y << n_power) + ((u64) (limbs * i) << n_power);
        const u64 av = a[src];
        u64 k0 = mul_barrett(sp, av, m);
        k0 = add_mod(k0, e[src], m.q);
        k0 = sub_mod(0, k0, m.q);
        if (i == own) k0 = add_mod(k0, carried, m.q);
        const u64 dst = idx + ((u64) y << n_power) + ((u64) (limbs * i) << (n_power + 1));
        key[dst] = k0;
        key[dst + ((u64) limbs << n_power)] = av;
    }
}

hipError_t kg_switchkey(u64* key, const u64* sk, const u64* e, const u64* a, const Mod* mods, const u64* factor,
                        int galois_elt, const u64* old_sk, int n_power, int limbs, int digits, int digit_width,
                        int q_size, int p_size, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_switchkey, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, key, sk, e,
                       a, mods, factor, galois_elt, old_sk, n_power, limbs, digits, digit_width, q_size, p_size);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_pk_u(const u64* __restrict__ pk, const u64* __restrict__ u,
                                                        u64* __restrict__ out, const Mod* __restrict__ mods,
                                                        int n_power, int limbs)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    const u64 z = ((u64) limbs << n_power) * blockIdx.z;
    out[loc + z] = mul_barrett(pk[loc + z], u[loc], mods[blockIdx.y]);
}

hipError_t kg_pk_u(const u64* pk, const u64* u, u64* out, const Mod* mods, int n_power, int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_pk_u, dim3((1u << n_power) / KG_THREADS, limbs, 2), dim3(KG_THREADS), 0, st, pk, u, out,
                       mods, n_power, limbs);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_message_add(u64* __restrict__ ct, const u64* __restrict__ plain,
                                                               const Mod* __restrict__ mods, int n_power)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    ct[loc] = add_mod(ct[loc], plain[loc], mods[blockIdx.y].q);
}

hipError_t kg_message_add(u64* ct, const u64* plain, const Mod* mods, int n_power, int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_message_add, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, ct, plain,
                       mods, n_power);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_sk_mul_ckks(const u64* __restrict__ ct, u64* __restrict__ plain,
                                                               const u64* __restrict__ sk,
                                                               const Mod* __restrict__ mods, int n_power, int limbs)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    const Mod m = mods[blockIdx.y];
    const u64 c1 = mul_barrett(ct[loc + ((u64) limbs << n_power)], sk[loc], m);
    plain[loc] = add_mod(c1, ct[loc], m.q);
}

hipError_t kg_sk_multiplication_ckks(const u64* ct, u64* plain, const u64* sk, const Mod* mods, int n_power,
                                     int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_sk_mul_ckks, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, ct, plain,
                       sk, mods, n_power, limbs);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_message_add(u64* __restrict__ ct, const u64* __restrict__ plain,
                                                                   const Mod* __restrict__ mods,
                                                                   const u64* __restrict__ coeff_div, u64 Q_mod_t,
                                                                   u64 upper_threshold, u64 t, int n_power)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const int y = blockIdx.y;
    const Mod m = mods[y];
    const u64 message = plain[idx];
    // 64-bit wrap-around and the detour through `int` are the reference's (encryption.cu:160-163)
    u64 fix = message * Q_mod_t;
    fix = fix + upper_threshold;
    fix = (u64) (long long) (int) (fix / t);
    u64 c0 = mul_barrett(message, coeff_div[y], m);
    c0 = add_mod(c0, fix, m.q);
    const u64 loc = idx + ((u64) y << n_power);
    ct[loc] = add_mod(ct[loc], c0, m.q);
}

hipError_t kg_bfv_message_add(u64* ct, const u64* plain, const Mod* mods, const u64* coeff_div, u64 Q_mod_t,
                              u64 upper_threshold, u64 t, int n_power, int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_message_add, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, ct,
                       plain, mods, coeff_div, Q_mod_t, upper_threshold, t, n_power);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_plain_addsub(const u64* __restrict__ ct,
                                                                    const u64* __restrict__ plain, u64* __restrict__ out,
                                                                    const Mod* __restrict__ mods,
                                                                    const u64* __restrict__ coeff_div, u64 Q_mod_t,
                                                                    u64 upper_threshold, u64 t, int n_power, int limbs,
                                                                    int sub)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    const Mod m = mods[y];
    const u64 loc = idx + ((u64) y << n_power) + (((u64) limbs * z) << n_power);
    u64 c = ct[loc];
    if (z == 0) {
        const u64 message = plain[idx];
        u64 fix = message * Q_mod_t;
        fix = fix + upper_threshold;
        fix = (u64) (long long) (int) (fix / t);
        u64 r = mul_barrett(message, coeff_div[y], m);
        r = add_mod(r, fix, m.q);
        c = sub ? sub_mod(c, r, m.q) : add_mod(r, c, m.q);
    }
    out[loc] = c;
}

hipError_t kg_bfv_plain_addsub(const u64* ct, const u64* plain, u64* out, const Mod* mods, const u64* coeff_div,
                               u64 Q_mod_t, u64 upper_threshold, u64 t, int n_power, int limbs, int sub,
                               hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_plain_addsub, dim3((1u << n_power) / KG_THREADS, limbs, 2), dim3(KG_THREADS), 0, st,
                       ct, plain, out, mods, coeff_div, Q_mod_t, upper_threshold, t, n_power, limbs, sub);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_threshold(const u64* __restrict__ plain, u64* __restrict__ out,
                                                                 const Mod* __restrict__ mods,
                                                                 const u64* __restrict__ inc, u64 upper_threshold,
                                                                 int n_power)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const int y = blockIdx.y;
    const u64 v = plain[idx];
    out[idx + ((u64) y << n_power)] = (v >= upper_threshold) ? add_mod(v, inc[y], mods[y].q) : v;
}

hipError_t kg_bfv_threshold(const u64* plain, u64* out, const Mod* mods, const u64* upper_half_increment,
                            u64 upper_threshold, int n_power, int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_threshold, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, plain,
                       out, mods, upper_half_increment, upper_threshold, n_power);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_sk_mul(const u64* __restrict__ in, const u64* __restrict__ sk,
                                                          u64* __restrict__ out, const Mod* __restrict__ mods,
                                                          int n_power)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    out[loc] = mul_barrett(in[loc], sk[loc], mods[blockIdx.y]);
}

hipError_t kg_sk_multiplication(const u64* in, const u64* sk, u64* out, const Mod* mods, int n_power, int limbs,
                                hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_sk_mul, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, in, sk, out,
                       mods, n_power);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_coeff_multadd(const u64* __restrict__ ct0, const u64* __restrict__ x,
                                                                 u64* __restrict__ out, u64 t,
                                                                 const Mod* __restrict__ mods, int n_power)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power);
    const Mod m = mods[blockIdx.y];
    out[loc] = mul_barrett(add_mod(ct0[loc], x[loc], m.q), reduce64(t, m), m);
}

hipError_t kg_coeff_multadd(const u64* ct0, const u64* x, u64* out, u64 t, const Mod* mods, int n_power, int limbs,
                            hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_coeff_multadd, dim3((1u << n_power) / KG_THREADS, limbs), dim3(KG_THREADS), 0, st, ct0, x,
                       out, t, mods, n_power);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_decryption(const u64* __restrict__ ct0,
                                                                  const u64* __restrict__ ct1,
                                                                  u64* __restrict__ plain,
                                                                  const Mod* __restrict__ mods, BfvDecryptDev d,
                                                                  int n_power, int limbs)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const u64 t = d.plain.q, g = d.gamma.q;
    u64 sum_t = 0, sum_g = 0;
    for (int i = 0; i < limbs; i++) {
        const Mod m = mods[i];
        const u64 loc = idx + ((u64) i << n_power);
        u64 mt = add_mod(ct0[loc], ct1[loc], m.q);
        const u64 g_i = reduce64(g, m);
        mt = mul_barrett(mt, t, m);
        mt = mul_barrett(mt, g_i, m);
        mt = mul_barrett(mt, d.Qi_inverse[i], m);
        u64 in_t = reduce64(mt, d.plain), in_g = reduce64(mt, d.gamma);
        in_t = mul_barrett(in_t, d.Qi_t[i], d.plain);
        in_g = mul_barrett(in_g, d.Qi_gamma[i], d.gamma);
        sum_t = add_mod(sum_t, in_t, t);
        sum_g = add_mod(sum_g, in_g, g);
    }
    sum_t = mul_barrett(sum_t, d.mulq_inv_t, d.plain);
    sum_g = mul_barrett(sum_g, d.mulq_inv_gamma, d.gamma);
    u64 result;
    if (sum_g > (g >> 1)) {
        const u64 g_t = reduce64(g, d.plain), sg_t = reduce64(sum_g, d.plain);
        result = sub_mod(g_t, sg_t, t);
        result = add_mod(sum_t, result, t);
        result = mul_barrett(result, d.inv_gamma, d.plain);
    } else {
        const u64 st = reduce64(sum_t, d.plain), sg_t = reduce64(sum_g, d.plain);
        result = sub_mod(st, sg_t, t);
        result = mul_barrett(result, d.inv_gamma, d.plain);
    }
    plain[idx] = result;
}

hipError_t kg_bfv_decryption(const u64* ct0, const u64* ct1s, u64* plain, const Mod* mods, const BfvDecryptDev& d,
                             int n_power, int limbs, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_decryption, dim3((1u << n_power) / KG_THREADS), dim3(KG_THREADS), 0, st, ct0, ct1s,
                       plain, mods, d, n_power, limbs);
    return hipGetLastError();
}

__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_encode(u64* __restrict__ out,
                                                              const long long* __restrict__ message,
                                                              const int* __restrict__ location, u64 t,
                                                              int message_size)
{
    const int idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const int loc = location[idx];
    u64 v = 0;
    if (idx < message_size) {
        long long m = message[idx];
        if (m < 0) m += (long long) t;
        v = (u64) m;
    }
    out[loc] = v;
}
__global__ __launch_bounds__(KG_THREADS) void k_kg_bfv_decode(u64* __restrict__ message, const u64* __restrict__ in,
                                                              const int* __restrict__ location)
{
    const int idx = blockIdx.x * KG_THREADS + threadIdx.x;
    message[idx] = in[location[idx]];
}

hipError_t kg_bfv_encode_scatter(u64* out, const long long* message, const int* location, u64 t, int message_size,
                                 int n_power, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_encode, dim3((1u << n_power) / KG_THREADS), dim3(KG_THREADS), 0, st, out, message,
                       location, t, message_size);
    return hipGetLastError();
}
hipError_t kg_bfv_decode_gather(u64* message, const u64* in, const int* location, int n_power, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_bfv_decode, dim3((1u << n_power) / KG_THREADS), dim3(KG_THREADS), 0, st, message, in,
                       location);
    return hipGetLastError();
}

// ---- CKKS ciphertext (+,-,*) one real constant, and multiplication / division by the imaginary unit
// addition_constant_plain_ckks_poly / substraction_constant_plain_ckks_poly (addition.cu:219-300): part 0
// gets +-round(value) mod q_j, the other parts are copied; cipher_constant_plain_multiplication_kernel
// (multiplication.cu:333-372): every part times round(value) mod q_j.  |value| < 2^128.
__global__ __launch_bounds__(KG_THREADS) void k_kg_ckks_constant(const u64* __restrict__ ct, double value,
                                                                 u64* __restrict__ out, const Mod* __restrict__ mods,
                                                                 int n_power, int op)
{
    const u64 loc = (u64) blockIdx.x * KG_THREADS + threadIdx.x + ((u64) blockIdx.y << n_power) +
                    (((u64) gridDim.y * blockIdx.z) << n_power);
    const u64 x = ct[loc];
    if (op != 2 && blockIdx.z != 0) { out[loc] = x; return; }
    const Mod m = mods[blockIdx.y];
    double c = round(value);
    const bool neg = signbit(c);
    c = fabs(c);
    const double two64 = 18446744073709551616.0;
    const u64 lo = (u64) fmod(c, two64), hi = (u64) (c / two64);
    u64 pt = reduce128(hi, lo, m);
    if (neg) pt = sub_mod(m.q, pt, m.q); // sub(q, 0) == q is the reference's
    out[loc] = op == 0 ? add_mod(x, pt, m.q) : op == 1 ? sub_mod(x, pt, m.q) : mul_barrett(x, pt, m);
}

hipError_t kg_ckks_constant(const u64* ct, double value, u64* out, const Mod* mods, int n_power, int limbs, int parts,
                            int op, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_ckks_constant, dim3((1u << n_power) / KG_THREADS, limbs, parts), dim3(KG_THREADS), 0, st,
                       ct, value, out, mods, n_power, op);
    return hipGetLastError();
}

// cipher_add_by_gaussian_integer_kernel / cipher_mult_by_gaussian_integer_kernel (multiplication.cu:497-570):
// the constant round(re) + round(im) * i in every slot.  In the NTT domain i is +psi^(N/2) on the first half
// of the positions and -psi^(N/2) on the second, so the slot constant is re +- im * psi^(N/2) mod q_j; op 0
// adds it to part 0 (the other parts are copied), op 1 multiplies every part by it.  The reference turns the
// rounded doubles into residues with NTL big integers (ckks/operator.cu:583-617) and accepts any magnitude; a
// double is mant * 2^e: below 2^128 the residue comes from its two 64-bit halves, beyond that from
// (mant mod q) * (2^e mod q) -- exact for every finite double.
__device__ __forceinline__ u64 residue_of_rounded(double value, const Mod& m)
{
    double c = round(value);
    const bool neg = signbit(c);
    c = fabs(c);
    const double two64 = 18446744073709551616.0;
    u64 r;
    if (c < two64 * two64) {
        const u64 lo = (u64) fmod(c, two64), hi = (u64) (c / two64);
        r = reduce128(hi, lo, m);
    } else {
        int e;
        const double fr = frexp(c, &e);              // c = fr * 2^e, 0.5 <= fr < 1
        const u64 mant = (u64) ldexp(fr, 53);        // the 53-bit integer mantissa, exact
        r = reduce64(mant, m);
        u64 p = reduce64(2, m), acc = reduce64(1, m); // 2^(e - 53) mod q by square and multiply
        for (int sh = e - 53; sh; sh >>= 1) {
            if (sh & 1) acc = reduce128(mulhi64(acc, p), acc * p, m);
            p = reduce128(mulhi64(p, p), p * p, m);
        }
        r = reduce128(mulhi64(r, acc), r * acc, m);
    }
    return (neg && r) ? m.q - r : r; // NTL: (x % q) made non-negative
}
__global__ __launch_bounds__(KG_THREADS) void k_kg_ckks_gaussian(const u64* __restrict__ ct, double re, double im,
                                                                 u64* __restrict__ out, const u64* __restrict__ psi_half,
                                                                 const Mod* __restrict__ mods, int n_power, int op)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const u64 loc = idx + ((u64) blockIdx.y << n_power) + (((u64) gridDim.y * blockIdx.z) << n_power);
    const u64 x = ct[loc];
    if (op == 0 && blockIdx.z != 0) { out[loc] = x; return; }
    const Mod m = mods[blockIdx.y];
    const u64 c_real = residue_of_rounded(re, m), c_imag = residue_of_rounded(im, m);
    const u64 const_imag = mul_barrett(c_imag, psi_half[blockIdx.y], m);
    const u64 k = (idx < (1u << (n_power - 1))) ? add_mod(c_real, const_imag, m.q) : sub_mod(c_real, const_imag, m.q);
    out[loc] = op == 0 ? add_mod(x, k, m.q) : mul_barrett(x, k, m);
}

hipError_t kg_ckks_gaussian(const u64* ct, double re, double im, u64* out, const u64* psi_half, const Mod* mods,
                            int n_power, int limbs, int parts, int op, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_ckks_gaussian, dim3((1u << n_power) / KG_THREADS, limbs, parts), dim3(KG_THREADS), 0, st, ct,
                       re, im, out, psi_half, mods, n_power, op);
    return hipGetLastError();
}

// cipher_mult_by_i_kernel / cipher_div_by_i_kernel (multiplication.cu:441-495): in the NTT domain the
// monomial X^(N/2) (= i in every slot) is +psi^(N/2) on the first half of the positions and -psi^(N/2) on
// the second; psi_half[j] = forward table entry 1 of modulus j
__global__ __launch_bounds__(KG_THREADS) void k_kg_ckks_mult_i(const u64* __restrict__ ct, u64* __restrict__ out,
                                                               const u64* __restrict__ psi_half,
                                                               const Mod* __restrict__ mods, int n_power, int divide)
{
    const u32 idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const u64 loc = idx + ((u64) blockIdx.y << n_power) + (((u64) gridDim.y * blockIdx.z) << n_power);
    const Mod m = mods[blockIdx.y];
    const u64 psi = psi_half[blockIdx.y];
    const bool first = idx < (1u << (n_power - 1));
    const u64 w = (first != (divide != 0)) ? psi : sub_mod(0, psi, m.q);
    out[loc] = mul_barrett(ct[loc], w, m);
}

hipError_t kg_ckks_mult_i(const u64* ct, u64* out, const u64* psi_half, const Mod* mods, int n_power, int limbs,
                          int parts, int divide, hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_ckks_mult_i, dim3((1u << n_power) / KG_THREADS, limbs, parts), dim3(KG_THREADS), 0, st, ct,
                       out, psi_half, mods, n_power, divide);
    return hipGetLastError();
}

// negacyclic_shift_poly_coeffmod_kernel (switchkey.cu:1433-1457): multiplication by X^shift in the coefficient
// domain; the wrapped coefficients are stored as q - x without a zero test, as the reference does
__global__ __launch_bounds__(KG_THREADS) void k_kg_negacyclic_shift(const u64* __restrict__ in, u64* __restrict__ out,
                                                                    const Mod* __restrict__ mods, int shift,
                                                                    int n_power)
{
    const int idx = blockIdx.x * KG_THREADS + threadIdx.x;
    const u64 base = ((u64) blockIdx.y << n_power) + (((u64) gridDim.y << n_power) * blockIdx.z);
    const int raw = idx + shift;
    u64 v = in[idx + base];
    if ((raw >> n_power) & 1) v = mods[blockIdx.y].q - v;
    out[(raw & ((1 << n_power) - 1)) + base] = v;
}

hipError_t kg_negacyclic_shift(const u64* in, u64* out, const Mod* mods, int shift, int n_power, int limbs, int parts,
                               hipStream_t st)
{
    hipLaunchKernelGGL(k_kg_negacyclic_shift, dim3((1u << n_power) / KG_THREADS, limbs, parts), dim3(KG_THREADS), 0, st,
                       in, out, mods, shift, n_power);
    return hipGetLastError();
}

} // namespace hegpu

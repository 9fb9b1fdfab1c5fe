// drbg.hpp -- the backend's own deterministic random bit generator and the
// three samplers of the key-generation / encryption path (host + device).
//
// The reference draws from rngongpu::RNG<AES> seeded with RAND_bytes
// (src/lib/util/random.cu:20-60): its streams can never be reproduced, only its
// distributions (random.cuh:52-708): uniform mod q_i per limb, a rounded
// Gaussian (sigma = 3.2, secstdparams.h:22) and a uniform ternary value shared
// by all limbs of a coefficient.  Here the bits come from the counter-based
// Philox4x32-10 generator (Salmon et al., "Parallel random numbers: as easy as
// 1, 2, 3", SC'11), written from the published round function: every output
// word is a pure function of (seed, stream, index), so the device kernels, the
// host code and the CPU oracle produce identical samples without sharing state.
#pragma once
#include "modarith.cuh"

namespace hegpu {

struct PhiloxOut {
    u32 w[4];
};

HG_HD PhiloxOut philox4x32_10(u32 k0, u32 k1, u32 c0, u32 c1, u32 c2, u32 c3)
{
    const u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; r++) {
        const u64 p0 = (u64) M0 * c0, p1 = (u64) M1 * c2;
        const u32 n0 = (u32) (p1 >> 32) ^ c1 ^ k0;
        const u32 n1 = (u32) p1;
        const u32 n2 = (u32) (p0 >> 32) ^ c3 ^ k1;
        const u32 n3 = (u32) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    PhiloxOut o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

// 128 bits for (seed, stream, index)
HG_HD PhiloxOut drbg_block(u64 seed, u64 stream, u64 index)
{
    return philox4x32_10((u32) seed, (u32) (seed >> 32), (u32) index, (u32) (index >> 32), (u32) stream,
                         (u32) (stream >> 32));
}

// Rounded Gaussian by CDT inversion: cdt[k] = floor(2^63 * P(|X| <= k)) for
// X = round(N(0, sigma^2)), k = 0..DRBG_GAUSS_MAX-1; magnitudes are clipped at
// DRBG_GAUSS_MAX (6 sigma = 19, the bound SEAL-style samplers use).
#define DRBG_GAUSS_MAX 19
struct GaussCdt {
    u64 t[DRBG_GAUSS_MAX];
};

HG_HD int drbg_gaussian(u64 seed, u64 stream, u64 index, const GaussCdt& cdt)
{
    const PhiloxOut o = drbg_block(seed, stream, index);
    const u64 r = (u64) o.w[0] | ((u64) o.w[1] << 32);
    const u64 u = r >> 1;
    int k = 0;
    for (int j = 0; j < DRBG_GAUSS_MAX; j++) k += (u >= cdt.t[j]) ? 1 : 0;
    return (r & 1) ? -k : k;
}

// uniform in {-1, 0, 1}
HG_HD int drbg_ternary(u64 seed, u64 stream, u64 index)
{
    const PhiloxOut o = drbg_block(seed, stream, index);
    return (int) (((u64) o.w[0] * 3) >> 32) - 1;
}

// uniform in [0, q): 128 random bits reduced mod q (bias < 2^-66)
HG_HD u64 drbg_uniform(u64 seed, u64 stream, u64 index, const Mod& m)
{
    const PhiloxOut o = drbg_block(seed, stream, index);
    const u64 lo = (u64) o.w[0] | ((u64) o.w[1] << 32), hi = (u64) o.w[2] | ((u64) o.w[3] << 32);
    return reduce128(hi, lo, m);
}

// Torus32 Gaussian noise for the TFHE front end (stddev alpha as a fraction of the torus; the
// reference draws curand / std::normal_distribution values, tfhe/encryptor.cu:55,
// keygenerator.cu).  A sum of 16 uniform 32-bit words (Irwin-Hall: mean 8*2^32, stddev
// 2^32*sqrt(16/12), support +-6.9 sigma) scaled by one FP64 multiply and rounded: integer
// arithmetic plus a single IEEE operation, hence identical on the device, the host and in the
// CPU oracle.  c = alpha * 2^32 / (2^32 * sqrt(4/3)) = alpha / 1.1547005383792517.
HG_HD int drbg_torus_gaussian(u64 seed, u64 stream, u64 index, double c)
{
    u64 sum = 0;
    for (int j = 0; j < 4; j++) {
        const PhiloxOut o = drbg_block(seed, stream, 4 * index + j);
        sum += (u64) o.w[0] + (u64) o.w[1] + (u64) o.w[2] + (u64) o.w[3];
    }
    const double g = (double) ((long long) sum - (8ll << 32));
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = rint(g * c);
#else
    const double r = __builtin_rint(g * c);
#endif
    return (int) (u32) (long long) r; // wraps on the 32-bit torus
}
// uniform torus32 value / uniform bit
HG_HD int drbg_torus_uniform(u64 seed, u64 stream, u64 index) { return (int) drbg_block(seed, stream, index).w[0]; }
HG_HD int drbg_bit(u64 seed, u64 stream, u64 index) { return (int) (drbg_block(seed, stream, index).w[0] & 1u); }

// signed small integer -> residue
HG_HD u64 lift_small(int v, u64 q) { return v < 0 ? q - (u64) (-v) : (u64) v; }

} // namespace hegpu

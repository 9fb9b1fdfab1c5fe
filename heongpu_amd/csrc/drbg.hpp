// drbg.hpp -- the backend's own deterministic random bit generator and the
// three samplers of the key-generation / encryption path (host + device).
//
// The reference draws from rngongpu::RNG<AES> (AES-CTR) seeded with RAND_bytes
// (src/lib/util/random.cu:20-60): its streams can never be reproduced, only its
// distributions (random.cuh:52-708): uniform mod q_i per limb, a rounded
// Gaussian (sigma = 3.2, secstdparams.h:22) and a uniform ternary value shared
// by all limbs of a coefficient.  Here the bits come from the ChaCha20 block
// function (Bernstein's original layout: 256-bit key, 64-bit block counter,
// 64-bit nonce; the same core as RFC 8439, whose section 2.3.2 vector pins this
// implementation) used as a counter-mode PRF: key = the generator's 256-bit
// seed, nonce = the stream id of the sampling call, counter = the sample index.
// Every output word is a pure function of (seed, stream, index), so the device
// kernels, the host code and the CPU oracle produce identical samples without
// sharing state -- and, unlike the Philox generator of round 1, knowing outputs
// (e.g. the public `a` polynomials) reveals nothing about the key or about the
// other streams (secret key, noise).  Seeds: 256 bits from the OS
// (hegpu_rng_create_from_entropy); the 64-bit-seed constructor exists for
// reproducible tests only.
#pragma once
#include "modarith.cuh"

namespace hegpu {

struct DrbgKey {
    u32 k[8];
};
struct DrbgOut {
    u32 w[4];
};

HG_HD u32 drbg_rotl(u32 v, int c) { return (v << c) | (v >> (32 - c)); }

#define DRBG_QR(a, b, c, d)                \
    a += b; d ^= a; d = drbg_rotl(d, 16);  \
    c += d; b ^= c; b = drbg_rotl(b, 12);  \
    a += b; d ^= a; d = drbg_rotl(d, 8);   \
    c += d; b ^= c; b = drbg_rotl(b, 7);

// first 128 bits of the ChaCha20 block (key, counter = index, nonce = stream)
HG_HD DrbgOut drbg_block(const DrbgKey& key, u64 stream, u64 index)
{
    const u32 c0 = 0x61707865u, c1 = 0x3320646eu, c2 = 0x79622d32u, c3 = 0x6b206574u;
    u32 x0 = c0, x1 = c1, x2 = c2, x3 = c3;
    u32 x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3];
    u32 x8 = key.k[4], x9 = key.k[5], x10 = key.k[6], x11 = key.k[7];
    u32 x12 = (u32) index, x13 = (u32) (index >> 32), x14 = (u32) stream, x15 = (u32) (stream >> 32);
    for (int r = 0; r < 10; r++) {
        DRBG_QR(x0, x4, x8, x12)
        DRBG_QR(x1, x5, x9, x13)
        DRBG_QR(x2, x6, x10, x14)
        DRBG_QR(x3, x7, x11, x15)
        DRBG_QR(x0, x5, x10, x15)
        DRBG_QR(x1, x6, x11, x12)
        DRBG_QR(x2, x7, x8, x13)
        DRBG_QR(x3, x4, x9, x14)
    }
    DrbgOut o;
    o.w[0] = x0 + c0; o.w[1] = x1 + c1; o.w[2] = x2 + c2; o.w[3] = x3 + c3;
    return o;
}
#undef DRBG_QR

// test-only expansion of a 64-bit seed: key words 0 and 1, the rest zero
HG_HD DrbgKey drbg_key_from_u64(u64 seed)
{
    DrbgKey k;
    k.k[0] = (u32) seed; k.k[1] = (u32) (seed >> 32);
    for (int i = 2; i < 8; i++) k.k[i] = 0;
    return k;
}

// Rounded Gaussian by CDT inversion: cdt[k] = floor(2^63 * P(|X| <= k)) for
// X = round(N(0, sigma^2)), k = 0..DRBG_GAUSS_MAX-1; magnitudes are clipped at
// DRBG_GAUSS_MAX (6 sigma = 19, the bound SEAL-style samplers use).
#define DRBG_GAUSS_MAX 19
struct GaussCdt {
    u64 t[DRBG_GAUSS_MAX];
};

HG_HD int drbg_gaussian(const DrbgKey& seed, u64 stream, u64 index, const GaussCdt& cdt)
{
    const DrbgOut o = drbg_block(seed, stream, index);
    const u64 r = (u64) o.w[0] | ((u64) o.w[1] << 32);
    const u64 u = r >> 1;
    int k = 0;
    for (int j = 0; j < DRBG_GAUSS_MAX; j++) k += (u >= cdt.t[j]) ? 1 : 0;
    return (r & 1) ? -k : k;
}

// uniform in {-1, 0, 1}
HG_HD int drbg_ternary(const DrbgKey& seed, u64 stream, u64 index)
{
    const DrbgOut o = drbg_block(seed, stream, index);
    return (int) (((u64) o.w[0] * 3) >> 32) - 1;
}

// uniform in [0, q): 128 random bits reduced mod q (bias < 2^-66)
HG_HD u64 drbg_uniform(const DrbgKey& seed, u64 stream, u64 index, const Mod& m)
{
    const DrbgOut o = drbg_block(seed, stream, index);
    const u64 lo = (u64) o.w[0] | ((u64) o.w[1] << 32), hi = (u64) o.w[2] | ((u64) o.w[3] << 32);
    return reduce128(hi, lo, m);
}

// Torus32 Gaussian noise for the TFHE front end (stddev alpha as a fraction of the torus; the
// reference draws curand / std::normal_distribution values, tfhe/encryptor.cu:55,
// keygenerator.cu).  A sum of 16 uniform 32-bit words (Irwin-Hall: mean 8*2^32, stddev
// 2^32*sqrt(16/12), support +-6.9 sigma) scaled by one FP64 multiply and rounded: integer
// arithmetic plus a single IEEE operation, hence identical on the device, the host and in the
// CPU oracle.  c = alpha * 2^32 / (2^32 * sqrt(4/3)) = alpha / 1.1547005383792517.
HG_HD int drbg_torus_gaussian(const DrbgKey& seed, u64 stream, u64 index, double c)
{
    u64 sum = 0;
    for (int j = 0; j < 4; j++) {
        const DrbgOut o = drbg_block(seed, stream, 4 * index + j);
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
HG_HD int drbg_torus_uniform(const DrbgKey& seed, u64 stream, u64 index) { return (int) drbg_block(seed, stream, index).w[0]; }
HG_HD int drbg_bit(const DrbgKey& seed, u64 stream, u64 index) { return (int) (drbg_block(seed, stream, index).w[0] & 1u); }

// signed small integer -> residue
HG_HD u64 lift_small(int v, u64 q) { return v < 0 ? q - (u64) (-v) : (u64) v; }

} // namespace hegpu

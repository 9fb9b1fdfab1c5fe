// tfhe.hpp -- TFHE gate-bootstrapping launchers (internal C++).
#pragma once
#include "drbg.hpp"
#include "modarith.cuh"

namespace hegpu {

// Fixed parameter set of the reference (src/lib/host/tfhe/context.cu:15-57).
struct TfheDev {
    Mod mod;                 // NTT prime 1152921504606877697
    const ulonglong2* tw;    // (psi^br(j), companion), N entries
    const ulonglong2* itw;   // (psi^-br(j), companion)
    ulonglong2 ninv, w1ninv; // N^-1 and itw[1]*N^-1 with companions
    int n, N, k, bk_l, bk_bg_bit;
    int offset, mask_mod, half_bg;
    int ks_base_bit, ks_length;
    // FP64 blind rotate (tfhe.hip): a 44-bit NTT prime of our own with its
    // 1024-entry tables as (double(w), RN(w/p')) pairs
    u64 fprime;
    const ulonglong2* ftw;
    const ulonglong2* fitw;
    ulonglong2 fninv, fw1ninv;
    // pinned host word the blind rotate sets when it is handed a buffer whose header is neither layout: the context
    // reports it at its next entry (cabi.cpp) -- a mismatch is never a silent return
    int* bad_key;
};

// Prepared boot key: TFHE_PREP_HEADER u64 of header (word 0: 1 = FP64 layout,
// 0 = integer layout; word 1: scratch flag) followed by the key data.
#define TFHE_PREP_HEADER 128

// ---- TFHE front end: keys, bit encryption, decryption (reference tfhe/keygenerator.cu,
// encryptor.cu, decryptor.cu); samplers in drbg.hpp
// binary LWE key [n] and TLWE key [k*N]
hipError_t tfhe_gen_secret(int* lwe_key, int* tlwe_key, int n, int kN, DrbgKey seed, u64 stream0, hipStream_t st);
// LWE encryptions under `key` [n]: a [shape][n], b [shape] = <a,key> + msg + noise.
// msg: torus32 messages [shape], or (ks_mode) the key-switch key pattern
// tlwe_key[i] * v * 2^(32 - (j+1)*base_bit) for shape index ((i*len + j)*(base-1) + v-1)
hipError_t tfhe_lwe_encrypt(int* out_a, int* out_b, const int* key, const int* msg, int ks_mode, const int* tlwe_key,
                            int base_bit, int len, int n, u64 shape, double noise_c, DrbgKey seed, u64 stream_a,
                            u64 stream_e, hipStream_t st);
// boot key in the reference layout [n][k+1][l][k+1][N] (NTT domain mod the 60-bit prime): row
// (i,y,z) = TLWE_S(0) + s_i * 2^(32 - (z+1)*bg_bit) on component y; tlwe_ntt: scratch [N]
hipError_t tfhe_gen_bootkey(const TfheDev& p, u64* boot_key, const int* lwe_key, const int* tlwe_key, u64* tlwe_ntt,
                            double noise_c, DrbgKey seed, u64 stream_a, u64 stream_e, hipStream_t st);
// phase[s] = b[s] - <a[s], key>
hipError_t tfhe_lwe_phase(const int* a, const int* b, const int* key, int* phase, int n, int shape, hipStream_t st);

// *fmt_out: the layout written (1 = FP64, 0 = integer)
hipError_t tfhe_prepare_bootkey(const TfheDev& p, const u64* src, u64* dst, u64 polys, bool allow_fp, int* fmt_out,
                                hipStream_t st);
// The layout is read ON THE DEVICE from the prepared key's header word, in stream order: both kernels are launched and
// the one whose layout is absent exits (an empty grid, ~5 us); a header that is neither sets *p.bad_key.
hipError_t tfhe_blind_rotate(const TfheDev& p, const int* in_a, const int* in_b, const u64* bk_prepared, int* out_a,
                             int* out_b, int encoded, int shape, hipStream_t st);
hipError_t tfhe_gate_pre(int* out_a, int* out_b, const int* a1, const int* b1, const int* a2, const int* b2,
                         int encoded, int s1, int s2, int m, int n, int shape, hipStream_t st);
hipError_t tfhe_key_switching(const TfheDev& p, const int* in_a, const int* in_b, int* out_a, int* out_b,
                              const int* ks_a, const int* ks_b, int shape, int ks_batched, int ks_pieces, hipStream_t st);

} // namespace hegpu

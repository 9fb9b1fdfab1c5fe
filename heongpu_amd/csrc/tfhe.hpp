// tfhe.hpp -- TFHE gate-bootstrapping launchers (internal C++).
#pragma once
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
};

// Prepared boot key: TFHE_PREP_HEADER u64 of header (word 0: 1 = FP64 layout,
// 0 = integer layout; word 1: scratch flag) followed by the key data.
#define TFHE_PREP_HEADER 128

hipError_t tfhe_prepare_bootkey(const TfheDev& p, const u64* src, u64* dst, u64 polys, bool allow_fp, hipStream_t st);
hipError_t tfhe_blind_rotate(const TfheDev& p, const int* in_a, const int* in_b, const u64* bk_prepared, int* out_a,
                             int* out_b, int encoded, int shape, hipStream_t st);
hipError_t tfhe_gate_pre(int* out_a, int* out_b, const int* a1, const int* b1, const int* a2, const int* b2,
                         int encoded, int s1, int s2, int m, int n, int shape, hipStream_t st);
hipError_t tfhe_key_switching(const TfheDev& p, const int* in_a, const int* in_b, int* out_a, int* out_b,
                              const int* ks_a, const int* ks_b, int shape, hipStream_t st);

} // namespace hegpu

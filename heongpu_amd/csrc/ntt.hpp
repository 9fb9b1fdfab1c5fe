// ntt.hpp -- NTT plan + launch interface (internal C++; the C ABI is in
// include/hegpu.h).
#pragma once
#include "modarith.cuh"

namespace hegpu {

// Arguments of one batched transform.  Mirrors what the reference hands to
// gpuntt:: (table/modulus pointers + ntt_rns_configuration + batch,mod_count
// [+ order]); see include/hegpu.h for the file:line citations.
struct NttArgs {
    const u64* in;            // batch polynomials (or base for poly_order)
    u64* out;                 // may equal in
    const Mod* mods;          // plan: per-modulus records
    const ulonglong2* tw;     // plan: forward (psi^br(j), shoup companion) [mod][N]
    const ulonglong2* itw;    // plan: inverse (psi^-br(j), companion)     [mod][N]
    const ulonglong2* ninv;   // plan: (N^-1, companion)                    [mod]
    const ulonglong2* w1ninv; // plan: (itw[1]*N^-1, companion)             [mod]
    const int* mod_order;     // optional: modulus index = mod_order[i % mod_count]
    const int* poly_order;    // optional: polynomial j of an item lives at slot poly_order[j]
    int n_power;
    int mod_count;
    int mod_offset;           // added to the modulus index (caller-offset tables)
    // Batched ciphertexts ("items"): polynomial p of the launch belongs to item
    // p / polys_per_item and is polynomial j = p % polys_per_item of it; item b
    // starts at in + b*in_item_stride / out + b*out_item_stride (elements).
    // polys_per_item == 0 means one item holding the whole batch.
    int polys_per_item;
    u64 in_item_stride;
    u64 out_item_stride;
    // Fused RNS digit decomposition (forward only): when decomp_mods > 0,
    // polynomial j of an item is digit d = j / decomp_mods re-reduced into its
    // modulus; it is READ from input slot d (not j) and every coefficient goes
    // through reduce64 first.  Replaces cipher_broadcast*_kernel + NTT
    // (reference switchkey.cu:11-59 followed by ckks/operator.cu:956).
    int decomp_mods;
};

hipError_t ntt_launch(const NttArgs& a, int batch, bool inverse, hipStream_t st);

} // namespace hegpu

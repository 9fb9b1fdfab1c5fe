// ntt.hpp -- NTT plan + launch interface (internal C++; the C ABI is in
// include/hegpu.h).
#pragma once
#include "modarith.cuh"

namespace hegpu {

// Arguments of one batched transform.  Mirrors what the reference hands to
// gpuntt:: (table/modulus pointers + ntt_rns_configuration + batch,mod_count
// [+ order]); see include/hegpu.h for the file:line citations.
// Optional epilogue of the forward row pass: instead of storing the transformed limb x the
// kernel writes (ks - x) * inv[modulus] + ct -- the second stage of the leveled mod-down
// (divide_round_lastq_leveled_stage_two_kernel, reference switchkey.cu:707-771) fused into the
// transform that produces its operand.  Polynomial j of an item is (part = j / limbs,
// limb = j % limbs); ks/ct/out are [part][*][N] per item.
struct NttEpilogue {
    const u64* ks;      // accumulated key-switch result, limb at ((part * ks_part_limbs + limb) << n_power)
    u64 ks_item_stride;
    int ks_part_limbs;
    const u64* ct;      // added term (NULL: none), limb at ((part * limbs + limb) << n_power); may alias out
    u64 ct_item_stride;
    int ct_parts;       // only parts below this get the added term (0: every part)
    // Galois automorphism b(X) = a(X^g) applied to the result, as a slot scatter: the value of slot j goes to
    // slot j' with 2 br(j') + 1 = (2 br(j) + 1) * galois_inv mod 2N, galois_inv = g^-1 mod 2N (0: none).  The
    // added term is read unpermuted (it is permuted with the sum).  out must not alias ct then.
    unsigned galois_inv;
    u64* out;           // same layout as ct
    u64 out_item_stride;
    const u64* inv;     // per-modulus P^-1
    int limbs;
    int on;
    unsigned mg_limbs; // set by ntt_launch: ceil(2^32 / limbs), see NttArgs::mg_*
};

// Optional epilogue of the INVERSE transform (BFV key switching, one special prime): the coefficient-domain limb x
// of modulus q_j the transform produces is not stored; instead
//     v = (x - ((p + half) mod P mod q_j - half_mod[j])) * inv[j]  (+ ct if part < add_parts)
// goes to out -- divide_round_lastq_kernel / divide_round_lastq_permute_bfv_kernel (reference
// switchkey.cu:400-478, 1621-1813) fused into the transform that produces their operand; with galois_elt != 0
// through the coefficient permutation out[(i g) mod N] = +-v.  An item is [2][limbs + 1][N] (polynomial
// j = part * (limbs + 1) + limb); slot `limbs` of a part is the P limb, which must ALREADY be in the coefficient
// domain (its own launch, before this one): the launch skips those two polynomials and reads p from them.
struct NttInvEpilogue {
    int on;
    int limbs;         // Q
    int add_parts;     // parts below this get ct added
    int p_mod;         // modulus index of P
    int galois_elt;    // 0: no permutation
    unsigned mg_slots; // set by ntt_launch: ceil(2^32 / (limbs + 1))
    u64 half;          // floor(P / 2)
    const u64* half_mod; // [limbs]
    const u64* inv;      // [limbs]  P^-1 mod q_j
    const u64* ct;     // [2][limbs][N] per item (may alias out when galois_elt == 0)
    u64 ct_item_stride;
    u64* out;          // [2][limbs][N] per item
    u64 out_item_stride;
    // Several special primes (key switching method II, divide_round_lastq_extended*_kernel, switchkey.cu:480-611):
    // an item is [2][limbs + p_count][N], the p_count special slots of a part are skipped, and the value subtracted
    // from x is not formed from the P limb but read from `u` ([2][limbs][N] per item, coefficient domain:
    // rns_moddown_multi_stage_one), v = (x - u) * inv[j] with inv = the product of the special primes' inverses.
    int p_count;       // 0 means 1
    const u64* u;      // nullptr: one special prime, the P limb is read from the item itself
    u64 u_item_stride;
};

struct NttArgs {
    const u64* in;            // batch polynomials (or base for poly_order)
    u64* out;                 // may equal in
    const Mod* mods;          // plan: per-modulus records
    const ulonglong2* tw;     // plan: forward (psi^br(j), shoup companion) [mod][N]
    const ulonglong2* itw;    // plan: inverse (psi^-br(j), companion)     [mod][N]
    // plan: the same twiddles re-laid for the last four (contiguous) stages of
    // the row pass: [mod][row c][slot k = 0..14][lane j = 0..15], slot k =
    // (1<<s)-1+b of local stage s, so that the 16 lanes of a row read 256
    // contiguous bytes per slot (the natural table would be a 128-byte lane
    // stride there).  Entry = tw[(((N/256 + c)*16 + j) << s) + b].
    const ulonglong2* twB;
    const ulonglong2* itwB;
    // plan: twB's layout as plain doubles for the FP64 moduli (zeros elsewhere).  The forward row stages of an FP64 limb
    // read 15 lane-dependent twiddles per thread -- as (w, RN(w/q)) pairs 480 bytes next to 256 bytes of coefficients; the
    // companion is one multiply (w * RN(1/q), as ks_row_mac_fp forms it), so the table need not carry it (round 5:
    // tools/exp/ntt_exp5 bounded the gain at 4-7 % of the transform before the table was built)
    const double* twB8;
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
    // With decomp_mods: skip polynomial (digit d, modulus index d).  For that
    // pair NTT_d(INTT_d(x) mod q_d) == x, so the caller copies the NTT-domain
    // limb into the slot instead (rns_copy_diag) and both passes exit early.
    int skip_identity;
    // Set by ntt_launch: when > 0 the grid walks polynomials modulus-major
    // (all polynomials of one modulus back to back) so that concurrently
    // running workgroups share one modulus' twiddle table in L2.
    int group_span; // polynomials per modulus class = batch / mod_count
    // Set by ntt_launch: ceil(2^32 / d) for the wave-uniform divisions of select_poly (dividends are
    // below 2^16 -- gridDim.y -- so mulhi(x, magic) == x / d exactly; 0 stands for d == 1).  The
    // compiler's own expansion of an integer division runs on the vector ALU even for uniform
    // operands: ~20 instructions each, 12 % of the FP64 column pass.
    unsigned mg_group_span, mg_per_item, mg_polys_per_item, mg_mod_count, mg_decomp_mods;
    NttEpilogue epi; // forward only, needs polys_per_item
    NttInvEpilogue iepi; // inverse only, needs polys_per_item == 2 * (limbs + 1)
    // Decomposing launches read digit d from input slot d * decomp_in_mul + decomp_in_add
    // (0 means 1 / 0).  half_on: the loaded residue v of modulus `half_src_mod` becomes
    // ((v + half) mod that modulus) mod q_j - half_mod[j] -- the first stage of the leveled
    // mod-down (divide_round_lastq_leveled_stage_one_kernel, switchkey.cu:678-705) as the load
    // transform of the NTT that follows it.
    int decomp_in_mul, decomp_in_add;
    int half_on, half_src_mod;
    u64 half;
    const u64* half_mod;
    // decomposing launches: 0 = one workgroup per (source tile, target modulus), 1 = one workgroup per
    // source tile walking all target moduli (ntt_fwd_col_multi), anything else = by launch size
    int col_multi;
    // N <= 2^14: one LDS-resident pass per transform.  1 always, 0 never (the two passes), otherwise by launch
    // size: one workgroup per limb needs a launch that fills the chip; a small one finishes sooner as two passes
    // of four times as many workgroups (C2, one ciphertext: inverse of 8 limbs 19.6 us against 6.0 + 5.5 us).
    int single_pass;
    int plan_has_fp, plan_has_int; // the plan holds FP64 (< 2^50) / integer-butterfly moduli
    // Integer butterflies: a modulus up to this bound runs ALL log2 N forward stages without a conditional
    // subtraction (set from the plan: (2^64 - 1 - 2 max q) / (4 log2 N), see context.cpp build_plan)
    u64 lazy_q_max;
    int only_int;                  // set by the launcher: the per-polynomial kernel skips FP64 moduli
    // Decomposing launch at N <= 2^14: the caller states that neither the output nor an operand of the epilogue
    // overlaps the source limbs, so the launch may run as ONE pass (ntt_fwd_single<S1, true>) by the single-pass rule
    int single_decomp_ok;
    // Decomposing launches whose FP64 targets go through ntt_fwd_col_multi: the target slots (index into the
    // decomp_mods moduli of a digit) that have INTEGER moduli, if the caller knows them -- the per-polynomial kernel
    // is then launched for exactly those (count > 0), or not at all (count < 0: none); 0 = unknown: it is launched
    // for every (digit, slot) and the workgroups of FP64 slots exit at once (15 of 17 at C4: 245 k empty
    // workgroups, 0.2 ms).
    int int_slot_count;
    int int_slots[8];
    // Decomposing launch through the multi-modulus kernel only (ntt_decomp_uses_multi): the source limbs are
    // the output of ntt_launch_inv_rows -- for FP64 source moduli: inverse row stages done, column stages
    // still to do -- and the kernel finishes their inverse transform itself (in place, the coefficient-domain
    // limbs are stored too).
    int src_inv;
    // Decomposing launches through the per-polynomial column kernel only (not ntt_decomp_uses_multi): the workgroup
    // of polynomial (digit d, target slot k) also copies its column tile of limb d * copy_part_limbs + k from
    // copy_src to copy_dst (items copy_*_item_stride apart) -- the copy of the kept limbs that the rescale's
    // epilogue reads (it overwrites them in place), without a launch of its own.  NULL: nothing.
    const u64* copy_src;
    u64* copy_dst;
    u64 copy_src_item_stride, copy_dst_item_stride;
    int copy_part_limbs;
    // Inverse launches only: the NTT-domain input of polynomial j = (part p, limb) of an item -- polys_per_item = 3 *
    // tensor_limbs -- is not read from `in` but formed on the load as the degree-2 tensor product of two transformed
    // ciphertexts held at tensor_in as [4][tensor_limbs][N] per item (a0, a1, b0, b1): p = 0: a0 b0, 1: a0 b1 + a1 b0,
    // 2: a1 b1 -- cross_multiplication (reference multiplication.cu:102-126) as the load transform of the inverse
    // transform that follows it in multiply_bfv (bfv/operator.cu:399-414): the [3][L][N] product is never stored.
    const u64* tensor_in; // nullptr: off
    u64 tensor_item_stride;
    int tensor_limbs;
    unsigned mg_tensor_limbs; // set by ntt_launch
};

hipError_t ntt_launch(const NttArgs& a, int batch, bool inverse, hipStream_t st);
// whether a decomposing launch of this shape takes the multi-modulus column kernel (so that src_inv may be used)
bool ntt_decomp_uses_multi(const NttArgs& a, int batch);
// first half of an inverse transform: row stages of every limb, column stages only for the limbs of integer
// moduli; the FP64 limbs are finished by the src_inv decomposing launch that follows
hipError_t ntt_launch_inv_rows(const NttArgs& a, int batch, hipStream_t st);
// forward column pass only (the row pass is done by ks_row_mac_launch)
hipError_t ntt_launch_fwd_col(const NttArgs& a, int batch, hipStream_t st);

// Fused "row pass + key-switch inner product" (method I/II): for every
// (ciphertext, target limb k, 4096-coefficient tile) one workgroup walks the
// digits, finishes each digit's forward NTT (the 8 contiguous stages) in
// registers/LDS and accumulates digit * key[digit][c][limb] in 128-bit lazy
// accumulators; only the two accumulated polynomials are written.  Replaces
// the row pass of GPU_NTT_Modulus_Ordered_Inplace + keyswitch_multiply_accumulate*
// (reference ckks/operator.cu:956-988): the [digits][rc][N] NTT output is
// never stored or re-read.
struct KsMacArgs {
    const u64* in;          // column-pass output, [item][digit][rc][N]
    u64 in_item_stride;
    const u64* key;         // [digit][2][key_limbs][N]
    u64* out;               // [item][2][rc][N]
    u64 out_item_stride;
    const Mod* mods;
    const ulonglong2* tw;
    const ulonglong2* twB;
    const double* twB8;     // as NttArgs::twB8 (the FP64 body parks plain doubles in LDS: half the bytes to fetch)
    const int* mod_order;   // modulus index of limb slot k (NULL: k)
    int n_power, digits, rc, key_limbs;
    u64 lazy_q_max;         // as NttArgs::lazy_q_max (the column pass that wrote `in` used the same bound)
    int skip_identity;      // digit d at modulus d is not transformed: its NTT-domain limb is read from `ident`
    const u64* ident;       // [item][digit][N] NTT-domain limbs (the polynomial that was decomposed)
    u64 ident_item_stride;
    int items;              // set by ks_row_mac_launch
    // Launches too small for one workgroup per (ciphertext, limb slot, tile) to fill the chip: `splits` workgroups
    // share such a unit, each takes the digits [s * digits / splits, (s + 1) * digits / splits) and leaves its two
    // partial sums (canonical residues) IN PLACE of the first two digits of its range in `in` (those regions are
    // read by this workgroup alone; digits >= 2 * splits); rns_sum_partials adds them into `out` afterwards.
    int splits;             // 0 / 1: none
    int no_fp, no_int;      // the plan has no FP64 / no integer-butterfly moduli: that kernel is not launched at all
    // The limb slots with integer moduli in ascending order, if the caller knows them (count > 0; as
    // NttArgs::int_slots): each kernel then runs on a compact grid over the slots of its own kind.
    int int_slot_count;
    int int_slots[8];
};
hipError_t ks_row_mac_launch(const KsMacArgs& a, int items, hipStream_t st);

} // namespace hegpu

// rns.hpp -- launchers for the RNS element-wise kernels (internal C++).
// Each launcher cites the reference kernel it replaces; layouts are the
// reference's limb-major planar layout: [part][limb][coeff] per ciphertext,
// ciphertexts of a batch `*_stride` elements apart.
#pragma once
#include "modarith.cuh"

namespace hegpu {

// reference src/lib/kernel/addition.cu:10-47
hipError_t rns_addition(const u64* a, const u64* b, u64* out, const Mod* mods, int n_power,
                        int limbs, int parts, int batch, int op /*0 add,1 sub,2 neg*/,
                        hipStream_t st);

// addition with per-ciphertext strides: out[b] = a[b] + b_[b] over [parts][limbs][N]
// (addition.cu:10-21 as used by ckks/operator.cu:1149)
hipError_t rns_addition_strided(const u64* a, u64 sa, const u64* b, u64 sb, u64* out, u64 so, const Mod* mods,
                                int n_power, int limbs, int parts, int batch, hipStream_t st);

// reference multiplication.cu:102-126
hipError_t rns_cross_multiplication(const u64* in1, u64 s1, const u64* in2, u64 s2, u64* out,
                                    u64 so, const Mod* mods, int n_power, int decomp_size,
                                    int batch, hipStream_t st);

// reference switchkey.cu:11-59, 1558-1590, 1592-1619 (digit decomposition):
// out[y][i][n] = in[y][n] mod q_{map(i)},  map(i) = i < split ? i : i+level.
hipError_t rns_decompose(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                         int n_power, int digits, int nmods, int split, int level, int batch,
                         hipStream_t st);

// reference switchkey.cu:61-285: out[c][y][n] = sum_i in[i][y][n]*key[i][c][kidx(y)][n].
// key strides use key_limbs (= Q' at depth 0); kidx(y) = (y < split) ? y : y + level
// (method I leveled: split = l, level = depth maps row l to the P limb; method II,
// switchkey.cu:287-398: rows >= l are the P limbs).
// the same for up to 4 keys sharing one read of the digits (hoisted rotations); result of key e at
// out + e * out_key_stride
hipError_t rns_keyswitch_mac_keys(const u64* in, u64 in_stride, const u64* const* keys, int key_count, u64* out,
                                  u64 out_stride, u64 out_key_stride, const Mod* mods, int n_power, int digits, int nmods,
                                  int key_limbs, int split, int level, int batch, hipStream_t st);
hipError_t rns_keyswitch_mac(const u64* in, u64 in_stride, const u64* key, u64* out, u64 out_stride,
                             const Mod* mods, int n_power, int digits, int nmods, int key_limbs,
                             int split, int level, int batch, hipStream_t st);

// reference switchkey.cu:872-927 / 985-1046 (method II digit -> Q~ fast base
// conversion with the float32 overflow estimate); out [d][rc][N]
hipError_t rns_base_conversion_DtoQtilde(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                                         const u64* matrix, const u64* mi_inv, const u64* prod, const int* I_j,
                                         const int* I_location, int n_power, int d, int rc, int l, int level,
                                         int max_cnt /* widest digit */, int batch, hipStream_t st);

// first half of the multi-prime mod-down in its NTT-domain form (context.cpp m2_md_*; ops.cpp ckks_moddown_multi)
hipError_t rns_moddown_multi_stage_one(const u64* in, u64 in_stride, u64* out, u64 out_stride, const Mod* mods,
                                       const u64* half, const u64* half_mod, const u64* last_q_modinv, const u64* G,
                                       const u64* C, int n_power, int Qp_cur, int Q_cur, int first_Qp, int first_Q,
                                       int P_size, int batch, hipStream_t st);
// reference switchkey.cu:480-611 / 1222-1282 (mod-down by P_size primes);
// with_ct: 0 none, 1 both parts, 2 part 0 only
hipError_t rns_moddown_extended(const u64* in, u64 in_stride, const u64* ct, u64 ct_stride, u64* out,
                                u64 out_stride, const Mod* mods, const u64* half, const u64* half_mod,
                                const u64* last_q_modinv, int n_power, int Qp_cur, int Q_cur, int first_Qp,
                                int first_Q, int P_size, int with_ct, int batch, hipStream_t st);

// reference switchkey.cu:400-478 (switchkey != 0: ct added to part 0 only)
hipError_t rns_divide_round_lastq(const u64* in, u64 in_stride, const u64* ct, u64 ct_stride,
                                  u64* out, u64 out_stride, const Mod* mods, const u64* half,
                                  const u64* half_mod, const u64* last_q_modinv, int n_power,
                                  int decomp, int switchkey, int batch, hipStream_t st);

// reference switchkey.cu:678-705
hipError_t rns_moddown_stage_one(const u64* in, u64 in_stride, u64* out, u64 out_stride,
                                 const Mod* mods, const u64* half, const u64* half_mod, int n_power,
                                 int first_decomp, int cur_decomp, int batch, hipStream_t st);

// reference switchkey.cu:707-771 (ct may alias out); with_ct: 0 none (rescale,
// switchkey.cu:792-815), 1 both parts, 2 part 0 only (switchkey variant)
hipError_t rns_moddown_stage_two(const u64* in_last, u64 last_stride, const u64* in, u64 in_stride,
                                 int in_limbs, const u64* ct, u64 ct_stride, u64* out,
                                 u64 out_stride, const Mod* mods, const u64* last_q_modinv,
                                 int n_power, int cur_decomp, int with_ct, int batch,
                                 hipStream_t st);

// reference switchkey.cu:1621-1813 (mod-down by P_size primes + Galois permute)
hipError_t rns_moddown_permute(const u64* in, u64 in_stride, const u64* in2, u64 in2_stride,
                               u64* out, u64 out_stride, const Mod* mods, const u64* half,
                               const u64* half_mod, const u64* last_q_modinv, int galois_elt,
                               int n_power, int Qp_cur, int Q_cur, int first_Qp, int first_Q,
                               int P_size, int batch, hipStream_t st);

// plain strided copy of `limbs` limbs x `parts` parts (switchkey.cu:776-790,
// bfv_duplicate's c0 copy)
// Galois automorphism b(X) = a(X^g) of `limbs` NTT-domain limbs per item (slot gather); out must not alias in
hipError_t rns_permute_ntt(const u64* in, u64 in_stride, u64* out, u64 out_stride, int galois_elt, int n_power,
                           int limbs, int batch, hipStream_t st);
hipError_t rns_copy_limbs(const u64* in, u64 in_part_stride, u64 in_stride, u64* out,
                          u64 out_part_stride, u64 out_stride, int n_power, int limbs, int parts,
                          int batch, hipStream_t st);

// out[b][d*(rc+1)][*] = in[b][d][*] for d < limbs: places NTT-domain limb d in
// the (digit d, modulus d) slot of a [l][rc][N] key-switch buffer
hipError_t rns_copy_diag(const u64* in, u64 in_stride, u64* out, u64 out_stride, int n_power, int limbs, int rc,
                         int batch, hipStream_t st);

struct BehzDev {
    const Mod* ibase;       // q_0..q_{Q-1}
    const Mod* obase;       // Bsk
    Mod m_tilde;
    Mod plain;
    u64 inv_prod_q_mod_m_tilde;
    u64 inv_prod_B_mod_m_sk;
    const u64* inv_m_tilde_mod_Bsk;
    const u64* prod_q_mod_Bsk;
    const u64* base_change_matrix_Bsk;
    const u64* base_change_matrix_m_tilde;
    const u64* inv_punctured_prod_mod_base_array;
    const u64* inv_prod_q_mod_Bsk;
    const u64* inv_punctured_prod_mod_B_array;
    const u64* base_change_matrix_q;
    const u64* base_change_matrix_msk;
    const u64* prod_B_mod_q;
    // merged constants (context.cpp): m_tilde * inv_punct_q[i], t * inv_punct_q[i] mod q_i;
    // inv_prod_q_mod_Bsk[i] * inv_punct_B[i] mod Bsk_i
    const u64* mtilde_inv_punct;
    const u64* t_inv_punct;
    const u64* invq_inv_punct_B;
    const u64* msk_mod_q; // m_sk mod q_i
    // rows with their trailing constant factors multiplied in (context.cpp), so that a row is ONE lazy 128-bit sum and
    // one reduction: fc_matrix[i][j] = base_change_matrix_Bsk[i][j] * inv_m_tilde_mod_Bsk[i], fc_c1[i] = prod_q_mod_Bsk[i]
    // * inv_m_tilde_mod_Bsk[i]  (mod Bsk_i);  ff_matrix[i][j] = -base_change_matrix_Bsk[i][j] * c_i, ff_tc[i] = t * c_i
    // with c_i = inv_prod_q_mod_Bsk[i] [* inv_punctured_prod_mod_B_array[i] for i < |B|]  (mod Bsk_i)
    // Every one of these also carries the factor 2^64 (mod its modulus) that the Montgomery reduction of the lazy
    // sum (redc128) divides out: ff_q_matrix = base_change_matrix_q, ff_msk_matrix = base_change_matrix_msk,
    // ff_prod_B / ff_neg_prod_B = prod_B_mod_q[i] / q_i - prod_B_mod_q[i], each times 2^64.
    const u64* fc_matrix;
    const u64* fc_c1;
    const u64* ff_matrix;
    const u64* ff_tc;
    const u64* ff_q_matrix;
    const u64* ff_msk_matrix;
    const u64* ff_prod_B;
    const u64* ff_neg_prod_B;
    int ibase_size, obase_size;
    int split; // rows of the base conversions over four wavefronts: 1 / 0 forced, -1 by launch size (option behz_split)
};

// Sum of the partial inner products of a digit-split ks_row_mac launch (KsMacArgs::splits): part p of split s sits in
// the digit buffer `buf` ([digit][rc][N] per item) at digit s * digits / splits + p; out[item][p][slot] = their sum
// modulo the modulus of the slot (mod_order as in the launch, NULL: slot k is modulus k).
hipError_t rns_sum_partials(const u64* buf, u64 buf_item_stride, u64* out, u64 out_item_stride, const Mod* mods,
                            const int* mod_order, int n_power, int digits, int rc, int splits, int batch,
                            hipStream_t st);

// reference multiplication.cu:10-100
hipError_t rns_fast_convertion(const u64* in1, u64 s1, const u64* in2, u64 s2, u64* out, u64 so,
                               const BehzDev& b, int n_power, int batch, hipStream_t st);
// reference multiplication.cu:128-272
hipError_t rns_fast_floor(const u64* in, u64 si, u64* out, u64 so, const BehzDev& b, int n_power,
                          int batch, hipStream_t st);

} // namespace hegpu

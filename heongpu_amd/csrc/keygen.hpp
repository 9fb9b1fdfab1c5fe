// keygen.hpp -- samplers and element-wise kernels of key generation,
// encryption and decryption; internal C++.
#pragma once
#include "drbg.hpp"

namespace hegpu {

// out[poly][limb][N] uniform mod q_limb                    (random.cuh: modular_uniform_*)
hipError_t kg_uniform(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                      hipStream_t st);
// out[poly][limb][N]: one rounded Gaussian per (poly, coefficient), lifted into every limb
hipError_t kg_gaussian(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                       const GaussCdt& cdt, hipStream_t st);
// same with a uniform ternary value
hipError_t kg_ternary(u64* out, const Mod* mods, int n_power, int limbs, int polys, DrbgKey seed, u64 stream,
                      hipStream_t st);
// secretkey_gen_kernel_v2 + secretkey_rns_kernel (keygeneration.cu:39-88): scatter `count`
// (position, +-1) pairs into an all-zero polynomial and lift it into `limbs` limbs
hipError_t kg_secret_rns(const int* positions, const int* values, int count, u64* out, const Mod* mods, int n_power,
                         int limbs, hipStream_t st);
// publickey_gen_kernel (keygeneration.cu:93-116): pk = [-(s*a + e), a], all NTT domain
hipError_t kg_publickey(u64* pk, const u64* sk, const u64* e, const u64* a, const Mod* mods, int n_power, int limbs,
                        hipStream_t st);
// relinkey_gen_kernel / galoiskey_gen_kernel (keygeneration.cu:145-185, 757-805), method I:
// key[d][0][j] = -(s'_j * a_dj + e_dj) + (d == j) * factor_j * t_j,  key[d][1][j] = a_dj with
// (s', t) = (s, s*s) for relinearisation, (sigma_g(s), s) for a Galois key (galois_elt != 0)
hipError_t kg_switchkey(u64* key, const u64* sk, const u64* e, const u64* a, const Mod* mods, const u64* factor,
                        int galois_elt, const u64* old_sk, int n_power, int limbs, int digits, int digit_width,
                        int q_size, int p_size, hipStream_t st);
// pk_u_kernel (encryption.cu:10-26): out[z][j] = pk[z][j] * u[j]
hipError_t kg_pk_u(const u64* pk, const u64* u, u64* out, const Mod* mods, int n_power, int limbs, hipStream_t st);
// cipher_message_add_kernel (encryption.cu:254-267): ct[0][j] += plain[j]
hipError_t kg_message_add(u64* ct, const u64* plain, const Mod* mods, int n_power, int limbs, hipStream_t st);
// sk_multiplication_ckks (decryption.cu:349-367): plain[j] = ct[0][j] + ct[1][j] * sk[j]
hipError_t kg_sk_multiplication_ckks(const u64* ct, u64* plain, const u64* sk, const Mod* mods, int n_power,
                                     int limbs, hipStream_t st);

// BFV: part 0 of a fresh encryption gets Delta*m + the rounding fix (tail of
// enc_div_lastq_bfv_kernel, encryption.cu:158-172); plain [N] mod t, ct [2][Q][N] coefficient domain
hipError_t kg_bfv_message_add(u64* ct, const u64* plain, const Mod* mods, const u64* coeff_div, u64 Q_mod_t,
                              u64 upper_threshold, u64 t, int n_power, int limbs, hipStream_t st);
// addition_plain_bfv_poly / substraction_plain_bfv_poly (addition.cu:50-176): out = ct +- (Delta*m + fix)
// on part 0, part 1 copied; sub != 0 subtracts
hipError_t kg_bfv_plain_addsub(const u64* ct, const u64* plain, u64* out, const Mod* mods, const u64* coeff_div,
                               u64 Q_mod_t, u64 upper_threshold, u64 t, int n_power, int limbs, int sub,
                               hipStream_t st);
// threshold_kernel (multiplication.cu:274-296): plain [N] mod t -> [limbs][N] centred lift into each q_i
hipError_t kg_bfv_threshold(const u64* plain, u64* out, const Mod* mods, const u64* upper_half_increment,
                            u64 upper_threshold, int n_power, int limbs, hipStream_t st);
// sk_multiplication (decryption.cu:10-23): out[j] = in[j] * sk[j]
hipError_t kg_sk_multiplication(const u64* in, const u64* sk, u64* out, const Mod* mods, int n_power, int limbs,
                                hipStream_t st);
// coeff_multadd (decryption.cu, noise budget path): out[j] = (ct0[j] + x[j]) * t mod q_j
hipError_t kg_coeff_multadd(const u64* ct0, const u64* x, u64* out, u64 t, const Mod* mods, int n_power, int limbs,
                            hipStream_t st);
struct BfvDecryptDev {
    Mod plain, gamma;
    const u64 *Qi_t, *Qi_gamma, *Qi_inverse;
    u64 mulq_inv_t, mulq_inv_gamma, inv_gamma;
};
// decryption_kernel (decryption.cu:44-120): plain [N] from c0 and c1*s, both [Q][N] coefficient domain
hipError_t kg_bfv_decryption(const u64* ct0, const u64* ct1s, u64* plain, const Mod* mods, const BfvDecryptDev& d,
                             int n_power, int limbs, hipStream_t st);

// encode_kernel_bfv / decode_kernel_bfv (encoding.cu:11-41): slot idx <-> position location[idx]
hipError_t kg_bfv_encode_scatter(u64* out, const long long* message, const int* location, u64 t, int message_size,
                                 int n_power, hipStream_t st);
hipError_t kg_bfv_decode_gather(u64* message, const u64* in, const int* location, int n_power, hipStream_t st);

// ---- CKKS encoder (encode.hip)
// special FFT over `1 << log_slots` complex doubles in place; roots: the rotation-group-ordered
// table; inverse: scaled by `fix`
// the special FFT over `data` (slots complex doubles); encode hands the message in (`real_in`: converted in the first
// load), decode takes it out (`real_out`: converted in the last store)
hipError_t en_special_fft(void* data, const void* roots, int log_slots, bool inverse, double fix, hipStream_t st,
                          const double* real_in = nullptr, int in_size = 0, int in_complex = 0, double* real_out = nullptr,
                          int out_complex = 0);
// message == nullptr: the constant round(scale_or_value) everywhere
hipError_t en_coeff_conversion(u64* plain, const double* message, int size, double scale_or_value, const Mod* mods,
                               int limbs, int n_power, hipStream_t st);
hipError_t en_coeff_compose(double* message, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                            const u64* upper_half, const u64* M, int l, double scale, int n_power, hipStream_t st);
hipError_t en_conversion(u64* plain, const void* msg, const Mod* mods, int limbs, const int* reverse_order, int n_power,
                         hipStream_t st);
hipError_t en_compose(void* msg, const u64* plain, const Mod* mods, const u64* Mi_inv, const u64* Mi,
                      const u64* upper_half, const u64* M, int l, double scale, const int* reverse_order, int n_power,
                      hipStream_t st);

// CKKS ciphertext with one constant: op 0 add, 1 subtract (part 0 only), 2 multiply (all parts)
hipError_t kg_ckks_constant(const u64* ct, double value, u64* out, const Mod* mods, int n_power, int limbs, int parts,
                            int op, hipStream_t st);
// cipher_add_by_gaussian_integer_kernel / cipher_mult_by_gaussian_integer_kernel (multiplication.cu:497-570)
hipError_t kg_ckks_gaussian(const u64* ct, double re, double im, u64* out, const u64* psi_half, const Mod* mods,
                            int n_power, int limbs, int parts, int op, hipStream_t st);
hipError_t kg_ckks_mult_i(const u64* ct, u64* out, const u64* psi_half, const Mod* mods, int n_power, int limbs,
                          int parts, int divide, hipStream_t st);

hipError_t kg_negacyclic_shift(const u64* in, u64* out, const Mod* mods, int shift, int n_power, int limbs, int parts,
                               hipStream_t st);

} // namespace hegpu

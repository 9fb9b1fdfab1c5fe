/* o_kernels.h -- internal prototypes of the restated kernels.
 * TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h). */
#ifndef O_KERNELS_H
#define O_KERNELS_H
#include "hegpu_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
void o_cipher_broadcast(const u64* input, u64* output, const omod_t* mods,
                        int n_power, int Q, int rns_mod_count);
void o_cipher_broadcast_leveled(const u64* input, u64* output,
                                const omod_t* mods, int first_rns_mod_count,
                                int current_rns_mod_count, int n_power,
                                int grid_y);
void o_keyswitch_mac(const u64* input, const u64* key, u64* output,
                     const omod_t* mods, int n_power, int Qt, int digits);
void o_keyswitch_mac_leveled(const u64* input, const u64* key, u64* output,
                             const omod_t* mods, int first_rns_mod_count,
                             int current_decomp_mod_count, int n_power);
void o_divide_round_lastq(const u64* input, const u64* ct, u64* output,
                          const omod_t* mods, const u64* half,
                          const u64* half_mod, const u64* last_q_modinv,
                          int n_power, int decomp_mod_count, int switchkey);
void o_divide_round_lastq_leveled_stage_one(
    const u64* input, u64* output, const omod_t* mods, const u64* half,
    const u64* half_mod, int n_power, int first_decomp_count,
    int current_decomp_count);
void o_divide_round_lastq_leveled_stage_two(
    const u64* input_last, const u64* input, const u64* ct, u64* output,
    const omod_t* mods, const u64* last_q_modinv, int n_power,
    int current_decomp_count, int switchkey);
void o_move_cipher_leveled(const u64* input, u64* output, int n_power,
                           int current_decomp_count);
void o_divide_round_lastq_rescale(const u64* input_last, const u64* input,
                                  u64* output, const omod_t* mods,
                                  const u64* last_q_modinv, int n_power,
                                  int current_decomp_count);
void o_ckks_duplicate(const u64* cipher, u64* output, const omod_t* mods,
                      int n_power, int first_rns_mod_count,
                      int current_rns_mod_count, int current_decomp_mod_count);
void o_bfv_duplicate(const u64* cipher, u64* output1, u64* output2,
                     const omod_t* mods, int n_power, int Q,
                     int rns_mod_count);
void o_divide_round_lastq_permute(const u64* input, const u64* input2,
                                  u64* output, const omod_t* mods,
                                  const u64* half, const u64* half_mod,
                                  const u64* last_q_modinv, int galois_elt,
                                  int n_power, int Q_prime_size, int Q_size,
                                  int first_Q_prime_size, int first_Q_size,
                                  int P_size);
void o_fast_convertion(const octx_t* c, const u64* in1, const u64* in2,
                       u64* out1);
void o_fast_floor(const octx_t* c, const u64* in_baseq_Bsk, u64* out1);
#ifdef __cplusplus
}
#endif
#endif

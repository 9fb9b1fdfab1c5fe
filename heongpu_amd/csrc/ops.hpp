// ops.hpp -- batched operator sequences on a Context (internal C++).
#pragma once
#include "context.hpp"
#include "keygen.hpp"

namespace hegpu {

enum { OP_CKKS_RELIN = 1, OP_CKKS_RESCALE = 2, OP_CKKS_GALOIS = 3, OP_BFV_MULTIPLY = 4, OP_BFV_RELIN = 5,
       OP_BFV_GALOIS = 6, OP_KEYGEN_SECRET = 7, OP_KEYGEN_PUBLIC = 8, OP_KEYGEN_SWITCH = 9, OP_CKKS_ENCRYPT = 10, OP_BFV_ENCRYPT = 11,
       OP_BFV_DECRYPT = 12, OP_BFV_DECODE = 13, OP_CKKS_ENCODE = 14,
       OP_CKKS_DECODE = 15, OP_BFV_MULTIPLY_PLAIN = 16, OP_CKKS_ROTATE_HOISTED = 17 };

size_t ops_workspace_elems(const Context& c, int op, int depth, int batch);

hipError_t op_ckks_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                            int depth, int batch, hipStream_t st);
// `phases`: which launches of the sequence run (all by default; hegpu_probe_ckks_relinearize times them one by one)
enum { RELIN_PHASE_INTT_C2 = 1, RELIN_PHASE_COLUMN = 2, RELIN_PHASE_ROW_MAC = 4, RELIN_PHASE_INTT_P = 8,
       RELIN_PHASE_MODDOWN = 16, RELIN_PHASE_ALL = 31 };
hipError_t op_ckks_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                               hipStream_t st, unsigned phases = RELIN_PHASE_ALL);
hipError_t op_ckks_rescale(const Context& c, u64* ct, u64 cs, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_ckks_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                int galois_elt, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                           int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                              hipStream_t st);
hipError_t op_bfv_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                               int galois_elt, int batch, u64* ws, hipStream_t st);

// fast_single_hoisting_rotation_ckks_method_I / _II (ckks/operator.cu:4674-5446): `count` rotations of one
// ciphertext with the decomposition and the digit NTT shared; keys / galois_elts are HOST arrays
hipError_t op_ckks_rotate_hoisted(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* const* keys,
                                  const int* galois_elts, int count, int depth, int batch, u64* ws, hipStream_t st,
                                  int group = 1 /* accumulators in ws: 1 (OP_CKKS_GALOIS) or 4 (OP_CKKS_ROTATE_HOISTED) */);

// key-switching method II (P_size > 1)
hipError_t op_ckks_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                                  hipStream_t st);
hipError_t op_ckks_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                   int galois_elt, int depth, int batch, u64* ws, hipStream_t st);
hipError_t op_bfv_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                                 hipStream_t st);
hipError_t op_bfv_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                  int galois_elt, int batch, u64* ws, hipStream_t st);

// ---- key generation / encryption / decryption, key-switch method I
// The generator state: every sampling call consumes one stream id of the DRBG (drbg.hpp).
struct Rng {
    DrbgKey seed{}; // 256-bit ChaCha20 key (drbg.hpp)
    u64 stream = 0;
};
// HEKeyGenerator::generate_secret_key_v2 (ckks/keygenerator.cu:85-160); sk [Q'][N], NTT domain
hipError_t op_gen_secret_key(const Context& c, Rng& r, int hamming_weight, u64* sk, u64* ws, hipStream_t st);
// generate_public_key (ckks/keygenerator.cu:167-240); pk [2][Q'][N]
hipError_t op_gen_public_key(const Context& c, Rng& r, const u64* sk, u64* pk, u64* ws, hipStream_t st);
// generate_relin_key_method_I (:242-324) / generate_galois_key_method_I (:415-560); key [Q][2][Q'][N];
// galois_elt == 0: relinearisation key; old_sk != nullptr (galois_elt == 0): generate_switch_key_method_I
// (:996-1095), the key under `sk` that carries old_sk
hipError_t op_gen_switch_key(const Context& c, Rng& r, const u64* sk, int galois_elt, const u64* old_sk, u64* key,
                             u64* ws, hipStream_t st);
// HEEncryptor<CKKS>::encrypt_ckks (ckks/encryptor.cu:36-110); plain [Q][N] NTT domain, ct [2][Q][N]
hipError_t op_ckks_encrypt(const Context& c, Rng& r, const u64* pk, const u64* plain, u64* ct, u64* ws,
                           hipStream_t st);
// HEDecryptor<CKKS>::decrypt_ckks (ckks/decryptor.cu:38-58); plain [l][N], l = Q - depth
hipError_t op_ckks_decrypt(const Context& c, const u64* ct, const u64* sk, int depth, u64* plain, hipStream_t st);
// HEEncryptor<BFV>::encrypt_bfv (bfv/encryptor.cu:39-108); plain [N] mod t, ct [2][Q][N] coefficient domain
hipError_t op_bfv_encrypt(const Context& c, Rng& r, const u64* pk, const u64* plain, u64* ct, u64* ws,
                          hipStream_t st);
// HEDecryptor<BFV>::decrypt_bfv (bfv/decryptor.cu:36-120), coefficient-domain ciphertext; plain [N]
hipError_t op_bfv_decrypt(const Context& c, const u64* ct, const u64* sk, u64* plain, u64* ws, hipStream_t st);
// first half of HEDecryptor<BFV>::noise_budget_calculation (bfv/decryptor.cu:170-225):
// out [Q][N] = t * (c0 + c1*s) mod q_j, coefficient domain (the caller composes and takes the norm)
hipError_t op_bfv_noise_rns(const Context& c, const u64* ct, const u64* sk, u64* out, hipStream_t st);
// HEEncoder<BFV>::encode_bfv / decode_bfv (bfv/encoder.cu:48-95, 213-249): message [size <= N]
// int64 (negative values wrap mod t) -> plain [N]; plain [N] -> message [N].  ws: N words (decode).
hipError_t op_bfv_encode(const Context& c, const long long* message, int message_size, u64* plain, hipStream_t st);
hipError_t op_bfv_decode(const Context& c, const u64* plain, u64* message, u64* ws, hipStream_t st);
// HEOperator<BFV>::multiply_plain_bfv, coefficient-domain ciphertext (bfv/operator.cu:432-503)
hipError_t op_bfv_plain_to_ntt(const Context& c, const u64* plain, u64* out, hipStream_t st);
hipError_t op_bfv_multiply_plain(const Context& c, const u64* ct, const u64* plain, u64* out, u64* ws, hipStream_t st);
// HEEncoder<CKKS>::encode_ckks / encode_ckks_coeff / decode_ckks / decode_ckks_coeff (ckks/encoder.cu:100-690).
// encode mode: 0 real slots, 1 complex slots ((re, im) pairs), 2 coefficients (<= N), 3 `scalar` in every slot;
// decode mode: 0 real parts [N/2], 1 complex slots [N/2 pairs], 2 coefficients [N].
// message: device doubles; plain [Q - depth][N] NTT domain
hipError_t op_ckks_encode(const Context& c, int mode, const double* message, int message_size, double scalar,
                          double scale, u64* plain, u64* ws, hipStream_t st);
hipError_t op_ckks_decode(const Context& c, int mode, const u64* plain, int depth, double scale, double* message,
                          u64* ws, hipStream_t st);

} // namespace hegpu

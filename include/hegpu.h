/*
 * hegpu.h -- C ABI of the MI355X-native RNS-polynomial backend (libhegpu.so).
 *
 * The reference (Alisah-Ozcan/HEonGPU) has no FFI seam: its boundary is the
 * C++ class API plus, one level down, the gpuntt:: free functions and the
 * __global__ kernels of src/lib/kernel.  This header is that lower seam as a
 * plain C ABI: raw device pointers, sizes and a hipStream_t (as void*); no
 * torch or C++ types.  Every entry point cites the reference interface it
 * replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - all data is uint64 residues in the reference's limb-major planar layout:
 *    element [part][limb][coeff] of a ciphertext at
 *    coeff + (limb << n_power) + part * (limbs << n_power);
 *  - batched calls take `batch` ciphertexts `*_stride` ELEMENTS apart (batch == 0: nothing is launched, the call
 *    succeeds; batch < 0: an error);
 *  - every call is asynchronous on `stream`; return value 0 = success,
 *    otherwise a hipError_t code (or HEGPU_E_* below); hegpu_last_error()
 *    returns a message for the calling thread;
 *  - device pointers and the stream must belong to the device the context was uploaded to; the call runs there
 *    whatever the calling thread's current device is (see hegpu_context_upload_device).
 */
#ifndef HEGPU_H
#define HEGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hegpu_context hegpu_context; /* opaque */
typedef void* hegpu_stream;                 /* hipStream_t */

enum { HEGPU_BFV = 1, HEGPU_CKKS = 2 };
enum { HEGPU_SEC_NONE = 0, HEGPU_SEC_128 = 128, HEGPU_SEC_192 = 192, HEGPU_SEC_256 = 256 };
enum { HEGPU_TABLES_QP = 0, HEGPU_TABLES_Q_BSK = 1, HEGPU_TABLES_PLAIN = 2 /* BFV plain modulus t, batching */ };
enum {
    HEGPU_E_INVALID = 10001, /* std::invalid_argument in the reference */
    HEGPU_E_LOGIC = 10002,   /* std::logic_error */
    HEGPU_E_RUNTIME = 10003, /* std::runtime_error */
    HEGPU_E_NODEVICE = 10004
};

const char* hegpu_last_error(void);
const char* hegpu_version(void);

/* ------------------------------------------------------------------ context
 * HEContextImpl<S>: set_poly_modulus_degree + set_coeff_modulus_bit_sizes +
 * [set_plain_modulus] + generate  (src/lib/host/ckks/context.cu:33-147,272-440;
 * src/lib/host/bfv/context.cu:33-147,396-705).  Creation is host-only (works
 * without a GPU); hegpu_context_upload() puts the tables on the current
 * device.  Errors mirror the reference's exceptions. */
int hegpu_context_create(int scheme, int poly_modulus_degree, const int* log_q_bits, int q_count,
                         const int* log_p_bits, int p_count, uint64_t plain_modulus, int sec_level,
                         hegpu_context** out);
/* set_coeff_modulus_default_values(P_modulus_size) (bfv/context.cu:267-374) */
int hegpu_context_create_default(int scheme, int poly_modulus_degree, int p_count, uint64_t plain_modulus,
                                 int sec_level, hegpu_context** out);
/* set_coeff_modulus_values(Q, P) (bfv/context.cu:149-265): explicit primes */
int hegpu_context_create_from_primes(int scheme, int poly_modulus_degree, const uint64_t* primes, int q_count,
                                     int p_count, uint64_t plain_modulus, hegpu_context** out);
/* the checks set_coeff_modulus_values runs before it accepts explicit primes (bfv/context.cu:149-220: widths through
 * coefficient_validator, total width against the security table; plus that every value admits a 2N-th root of unity);
 * 0, or the reference's exception as a code + hegpu_last_error */
int hegpu_validate_coeff_modulus_values(int poly_modulus_degree, const uint64_t* primes, int q_count, int p_count,
                                        int sec_level);
void hegpu_context_destroy(hegpu_context* ctx);
/* Run-time options of a context (the reference configures through objects as well: ExecutionOptions,
 * src/include/heongpu/util/storagemanager.cuh:34-97; MemoryPoolConfig, util/memorypool.cuh:38-54).  They select
 * between forms of the same computation -- every setting gives bit-identical results -- and exist for measurements
 * and for the parity tests, which run each form.  -1 = chosen by launch size (the default of the tri-state ones).
 *   "fused_row_mac"  -1/0/1  key switch: row pass of the digit NTT fused with the key inner product / the
 *                            reference's kernel sequence (NTT of all digits, then keyswitch_multiply_accumulate)
 *   "fused_moddown"  0/1     mod-down stages as load transform / epilogue of the transform between them (1)
 *   "col_multi"      -1/0/1  decomposing column pass: one workgroup per source tile walking the target moduli
 *   "single_pass"    -1/0/1  N <= 2^14: one LDS-resident pass per transform instead of two
 *   "ntt_galois"     0/1     CKKS rotations without leaving the NTT domain (1) / in the reference's order
 *   "galois_scatter" 0/1     the NTT-domain automorphism as the mod-down epilogue's store (1) / a kernel of its own
 *   "fuse_inverse"   0/1     the inverse transform feeding a decomposition ends inside it (1)
 *   "copy_along"     0/1     rescale: the copy of the kept limbs rides on the column pass (1)
 *   "digit_split"    -1/0/2/4  workgroups per unit of the fused key switch for small launches
 *   "fp_ntt"         0/1     moduli below 2^50 on the exact FP64 butterflies (1); ONLY before hegpu_context_upload
 *                            (HEGPU_E_LOGIC afterwards: it decides the layout of the twiddle tables)
 *   "behz_split"     -1/0/1  BFV base conversions with their rows over four wavefronts
 *   "fused_tensor"   0/1     BFV multiply: the tensor product as the load transform of the inverse transform (1)
 * Every option but "fp_ntt" may be changed at any time, but not concurrently with calls on the same context.  The
 * environment variables HEGPU_<NAME IN CAPITALS> seed the defaults once, when a context is created; no call path
 * reads the environment.  Unknown name / value out of range: HEGPU_E_INVALID. */
int hegpu_context_set_option(hegpu_context* ctx, const char* name, int value);
int hegpu_context_get_option(const hegpu_context* ctx, const char* name, int* value);
int hegpu_context_upload(hegpu_context* ctx);
/* ---- several GPUs in one process (batches of independent ciphertexts shard across the GPUs of a node,
 * only the evaluation keys are replicated).  One context per device: hegpu_context_clone copies the host-side
 * parameter set and options (not the device tables), hegpu_context_upload_device puts a context's tables on the
 * given device (hegpu_context_upload = the calling thread's current device).  Every entry point then runs on its
 * context's device whatever the caller's current device is (the caller's is restored on return), so a consumer drives
 * device d from any thread -- one OpenMP thread per device as the reference drives one stream per thread in
 * example/basic/9_multi_stream_usage_way1.cpp:27-66 -- with buffers and streams that belong to device d.
 * hegpu_broadcast_key copies keys[0] (on the device of ctxs[0]) to keys[i] on the device of ctxs[i], i = 1..n-1;
 * hegpu_broadcast_bytes is the same for any buffer (the TFHE boot and key-switch keys) with the devices given directly.
 * The copy into buffer i is ordered on streams[i] (NULL: the default streams), the source must be complete on
 * streams[0]'s timeline.  Path (hegpu_last_broadcast_path of the calling thread, or *path_out): peer access is
 * queried per edge (hipDeviceCanAccessPeer) and enabled (hipDeviceEnablePeerAccess); when every destination is
 * directly reachable from the source -- a fully connected xGMI node -- the copies are a FLAT fan-out, one hop, every
 * link of the source busy at once; otherwise a binomial TREE in <= 32 MiB chunks (hop r + 1 of a chunk starts when that
 * chunk has arrived).  HEGPU_BCAST_STAGED: at least one edge has no peer access and is staged through host memory by
 * the runtime; HEGPU_BCAST_SAME_DEVICE: every buffer lies on one device (a functional run on a 1-GPU box).
 * hegpu_context_device: -1 before upload. */
enum { HEGPU_BCAST_FLAT = 1, HEGPU_BCAST_TREE = 2, HEGPU_BCAST_STAGED = 0x100, HEGPU_BCAST_SAME_DEVICE = 0x200 };
int hegpu_context_clone(const hegpu_context* src, hegpu_context** out);
int hegpu_context_upload_device(hegpu_context* ctx, int device);
int hegpu_context_device(const hegpu_context* ctx);
int hegpu_broadcast_key(hegpu_context* const* ctxs, int n_ctx, uint64_t* const* keys, size_t elems,
                        const hegpu_stream* streams);
int hegpu_broadcast_bytes(const int* devices, int n, void* const* bufs, size_t bytes, const hegpu_stream* streams,
                          int* path_out);
int hegpu_last_broadcast_path(void);
/* integer properties: "n_power","Q_size","P_size","Q_prime_size","bsk_modulus" */
long hegpu_context_int(const hegpu_context* ctx, const char* name);
/* copy the named HOST table (reference member name without trailing '_',
 * e.g. "modulus","ntt_table","last_q_modinv","base_change_matrix_Bsk")
 * into out[cap]; returns the element count, <0 if unknown / too small */
long hegpu_context_get(const hegpu_context* ctx, const char* name, uint64_t* out, long cap);
/* device address of a named device table ("new_prime_locations", ...) */
const void* hegpu_context_device_ptr(const hegpu_context* ctx, const char* name);
/* keygeneration.cu:684-728 steps_to_galois_elt; 0 for |steps| >= N/2 (the reference prints an error and
 * returns 0 there); -1 only if the host code failed (no exception crosses this boundary) */
int hegpu_steps_to_galois_elt(int steps, int coeff_count, int group_order);

/* ------------------------------------------------------------------ NTT seam
 * gpuntt::GPU_NTT / GPU_NTT_Inplace / GPU_INTT / GPU_INTT_Inplace
 *   (call sites src/lib/host/bfv/operator.cu:393,410; ckks/operator.cu:919,1011)
 * gpuntt::GPU_NTT_Modulus_Ordered_Inplace  (ckks/operator.cu:956,1524)
 * gpuntt::GPU_NTT_Poly_Ordered_Inplace     (ckks/operator.cu:996,1197)
 * `batch` polynomials of N coefficients; polynomial i uses modulus
 * mod_offset + (mod_order ? mod_order[i % mod_count] : i % mod_count) of the
 * chosen table set and lives at slot (poly_order ? poly_order[i] : i).
 * mod_order / poly_order are DEVICE int arrays or NULL.  in == out allowed. */
int hegpu_ntt(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int inverse, int batch,
              int mod_count, int mod_offset, const int* mod_order, const int* poly_order, hegpu_stream stream);

/* The six gpuntt:: entry points by name, for a maintainer who binds call site by call site.  The
 * reference passes table / modulus / n^-1 pointers (offset by the caller for a sub-chain) and an
 * ntt_rns_configuration; here the context owns the tables, so `table_set` + `mod_offset` select what those
 * pointers selected, `inverse` is cfg.ntt_type == INVERSE (n^-1 of the same moduli is applied, as
 * cfg.mod_inverse does), cfg.stream is the last argument.  All return 0 or an error code.
 *   gpuntt::GPU_NTT(in, out, tables, moduli, cfg, batch, mod_count)              bfv/operator.cu:393
 *   gpuntt::GPU_NTT_Inplace(inout, tables, moduli, cfg, batch, mod_count)        ckks/operator.cu:1011
 *   gpuntt::GPU_INTT(in, out, itables, moduli, cfg, batch, mod_count)            ckks/operator.cu:1461
 *   gpuntt::GPU_INTT_Inplace(inout, itables, moduli, cfg, batch, mod_count)      bfv/operator.cu:410, ckks :919
 *   gpuntt::GPU_NTT_Modulus_Ordered_Inplace(inout, tables, moduli, cfg, batch, mod_count, order)   ckks :956,1524
 *   gpuntt::GPU_NTT_Poly_Ordered_Inplace(inout, tables, moduli, cfg, batch, mod_count, order)      ckks :996,1197
 * `order` is a DEVICE int array, as in the reference. */
int hegpu_GPU_NTT(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int mod_offset, int batch,
                  int mod_count, hegpu_stream stream);
int hegpu_GPU_NTT_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int mod_offset, int batch,
                          int mod_count, hegpu_stream stream);
int hegpu_GPU_INTT(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int mod_offset, int batch,
                   int mod_count, hegpu_stream stream);
int hegpu_GPU_INTT_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int mod_offset, int batch,
                           int mod_count, hegpu_stream stream);
int hegpu_GPU_NTT_Modulus_Ordered_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int inverse,
                                          int mod_offset, int batch, int mod_count, const int* order,
                                          hegpu_stream stream);
int hegpu_GPU_NTT_Poly_Ordered_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int inverse,
                                       int mod_offset, int batch, int mod_count, const int* order,
                                       hegpu_stream stream);

/* ------------------------------------------------------------------ kernels
 * 1:1 replacements of the reference __global__ kernels (grid dims become
 * arguments).  `table_set` selects the modulus array (Q' chain or q|Bsk). */
/* src/lib/kernel/addition.cu:10-47 ; op 0=addition 1=substraction 2=negation */
int hegpu_addition(hegpu_context* ctx, const uint64_t* in1, const uint64_t* in2, uint64_t* out, int limbs,
                   int parts, int batch, int op, hegpu_stream stream);
/* src/lib/kernel/multiplication.cu:102-126 */
int hegpu_cross_multiplication(hegpu_context* ctx, int table_set, const uint64_t* in1, uint64_t in1_stride,
                               const uint64_t* in2, uint64_t in2_stride, uint64_t* out, uint64_t out_stride,
                               int decomp_size, int batch, hegpu_stream stream);
/* src/lib/kernel/switchkey.cu:11-59,1558-1619: out[d][i][n] = in[d][n] mod
 * q_{i < split ? i : i+level} for d < digits, i < nmods */
int hegpu_cipher_broadcast(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                           uint64_t out_stride, int digits, int nmods, int split, int level, int batch,
                           hegpu_stream stream);
/* src/lib/kernel/switchkey.cu:61-398: out[c][y][n] = sum_i in[i][y][n] * key[i][c][kidx(y)][n],
 * kidx(y) = y < split ? y : y + level (non-leveled: split = nmods, level = 0;
 * method I leveled: split = l, level = depth; method II leveled: same) */
int hegpu_keyswitch_multiply_accumulate(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride,
                                        const uint64_t* key, uint64_t* out, uint64_t out_stride, int digits,
                                        int nmods, int key_limbs, int split, int level, int batch,
                                        hegpu_stream stream);
/* src/lib/kernel/switchkey.cu:872-927,985-1046 (method II digit -> Q~ base conversion, P_size > 1) */
int hegpu_base_conversion_DtoQtilde(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                                    uint64_t out_stride, int depth, int batch, hegpu_stream stream);
/* src/lib/kernel/switchkey.cu:400-478 */
int hegpu_divide_round_lastq(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                             uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int switchkey, int batch,
                             hegpu_stream stream);
/* src/lib/kernel/switchkey.cu:1621-1813 (scheme of ctx picks bfv/ckks form) */
int hegpu_divide_round_lastq_permute(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride,
                                     const uint64_t* in2, uint64_t in2_stride, uint64_t* out,
                                     uint64_t out_stride, int galois_elt, int depth, int batch,
                                     hegpu_stream stream);
/* ---- the leveled (CKKS) mod-down, rescale and multi-prime mod-down kernels, launch by launch.  l = Q - depth is the
 * reference's current_decomp_count; every layout is [2][limbs][N] per item.
 * divide_round_lastq_leveled_stage_one_kernel (switchkey.cu:678-705): the last limb of each part of `in`, + half,
 * reduced into every kept modulus, - half_mod.
 *   rescale == 0 (relinearize / rotate, ckks/operator.cu:1003): in [2][l+1][N] with the special-prime limb at slot l,
 *                out [2][l][N], tables half / half_mod;
 *   rescale != 0 (rescale, ckks/operator.cu:1205): in [2][l][N] with q_{l-1} at slot l-1, out [2][l-1][N], tables
 *                rescaled_half + depth / rescaled_half_mod + location(depth). */
int hegpu_divide_round_lastq_leveled_stage_one(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                                               uint64_t out_stride, int rescale, int depth, int batch,
                                               hegpu_stream stream);
/* divide_round_lastq_leveled_stage_two_kernel / _switchkey_kernel (switchkey.cu:707-771): out = (in - in_last) *
 * P^-1 + ct over [2][l][N]; in_last [2][l][N] (the NTT of stage one's output), in [2][l+1][N]; switchkey != 0 adds
 * ct to part 0 only.  ct may alias out. */
int hegpu_divide_round_lastq_leveled_stage_two(hegpu_context* ctx, const uint64_t* in_last, uint64_t last_stride,
                                               const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                                               uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int switchkey,
                                               int depth, int batch, hegpu_stream stream);
/* move_cipher_leveled_kernel (switchkey.cu:776-790, ckks/operator.cu:1219): the l-1 kept limbs of both parts,
 * [2][l][N] -> [2][l][N] (part stride l on both sides) */
int hegpu_move_cipher_leveled(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                              uint64_t out_stride, int depth, int batch, hegpu_stream stream);
/* divide_round_lastq_rescale_kernel (switchkey.cu:792-815, ckks/operator.cu:1225): out [2][l-1][N] =
 * (in - in_last) * q_{l-1}^-1; in_last [2][l-1][N], in [2][l][N]; tables rescaled_last_q_modinv + location(depth) */
int hegpu_divide_round_lastq_rescale(hegpu_context* ctx, const uint64_t* in_last, uint64_t last_stride,
                                     const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                                     int depth, int batch, hegpu_stream stream);
/* divide_round_lastq_extended_kernel (switchkey.cu:480-543: + ct on both parts, BFV method II relinearize),
 * _extended_switchkey_kernel (:545-611: + ct on part 0) and _extended_leveled_kernel (:1222-1282: no ct, CKKS
 * method II): mod-down by the P_size special primes, one after the other.  in [2][rc][N] (rc = Q' - depth,
 * coefficient domain), out / ct [2][l][N].  mode 0 leveled (ct ignored), 1 ct on both parts, 2 ct on part 0. */
int hegpu_divide_round_lastq_extended(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                                      uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int mode, int depth,
                                      int batch, hegpu_stream stream);
/* src/lib/kernel/multiplication.cu:10-100 / 128-272 (BFV BEHZ) */
int hegpu_fast_convertion(hegpu_context* ctx, const uint64_t* in1, uint64_t in1_stride, const uint64_t* in2,
                          uint64_t in2_stride, uint64_t* out, uint64_t out_stride, int batch,
                          hegpu_stream stream);
int hegpu_fast_floor(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                     uint64_t out_stride, int batch, hegpu_stream stream);

/* ------------------------------------------------------------------ operators
 * HEArithmeticOperator<S> sequences, batched over independent ciphertexts.
 * `ws` is a caller-provided device workspace of at least
 * hegpu_workspace_bytes(ctx, op, depth, batch) bytes (no allocation happens
 * inside, so the calls are hipGraph-capturable). */
enum {
    HEGPU_OP_CKKS_RELIN = 1,
    HEGPU_OP_CKKS_RESCALE = 2,
    HEGPU_OP_CKKS_GALOIS = 3,
    HEGPU_OP_BFV_MULTIPLY = 4,
    HEGPU_OP_BFV_RELIN = 5,
    HEGPU_OP_BFV_GALOIS = 6,
    HEGPU_OP_KEYGEN_SECRET = 7,
    HEGPU_OP_KEYGEN_PUBLIC = 8,
    HEGPU_OP_KEYGEN_SWITCH = 9, /* relinearisation and Galois keys */
    HEGPU_OP_CKKS_ENCRYPT = 10,
    HEGPU_OP_BFV_ENCRYPT = 11,
    HEGPU_OP_BFV_DECRYPT = 12,
    HEGPU_OP_BFV_DECODE = 13,
    HEGPU_OP_CKKS_ENCODE = 14,
    HEGPU_OP_CKKS_DECODE = 15, /* depth-dependent */
    HEGPU_OP_BFV_MULTIPLY_PLAIN = 16,
    HEGPU_OP_CKKS_ROTATE_HOISTED = 17 /* hegpu_ckks_rotate_hoisted with room for four accumulators */
};
size_t hegpu_workspace_bytes(const hegpu_context* ctx, int op, int depth, int batch);

/* HEOperator<CKKS>::multiply_ckks (src/lib/host/ckks/operator.cu:796-837):
 * ct [2][l][N] x [2][l][N] -> out [3][l][N], l = Q - depth */
int hegpu_ckks_multiply(hegpu_context* ctx, const uint64_t* ct1, uint64_t ct1_stride, const uint64_t* ct2,
                        uint64_t ct2_stride, uint64_t* out, uint64_t out_stride, int depth, int batch,
                        hegpu_stream stream);
/* relinearize_inplace (host/ckks/operator.cuh:1053-1094): key-switch method I
 * when P_size == 1 (relinearize_seal_method_inplace_ckks, ckks/operator.cu:899-1023,
 * relin_key [Q][2][Q'][N]), method II when P_size > 1
 * (relinearize_external_product_method2_inplace_ckks, ckks/operator.cu:1025-1154,
 * relin_key [d][2][Q'][N]).  ct [3][l][N] -> first two parts, in place. */
int hegpu_ckks_relinearize_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t ct_stride,
                                   const uint64_t* relin_key, int depth, int batch, void* ws, size_t ws_bytes,
                                   hegpu_stream stream);
/* rescale_inplace_ckks_leveled (ckks/operator.cu:1156-1244):
 * ct [2][l][N] -> [2][l-1][N] in place (caller then uses depth+1) */
int hegpu_ckks_rescale_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t ct_stride, int depth, int batch,
                               void* ws, size_t ws_bytes, hegpu_stream stream);
/* apply_galois_ckks_method_I (ckks/operator.cu:1422-1559); the result buffer must not contain the input: HEGPU_E_INVALID
 * if the address ranges of the two batches overlap (the epilogue scatters results while c0 is still being read) */
int hegpu_ckks_apply_galois(hegpu_context* ctx, const uint64_t* ct, uint64_t ct_stride, uint64_t* out,
                            uint64_t out_stride, const uint64_t* galois_key, int galois_elt, int depth,
                            int batch, void* ws, size_t ws_bytes, hegpu_stream stream);
/* fast_single_hoisting_rotation_ckks (host/ckks/operator.cuh:2133-2196; method I ckks/operator.cu:4674-4953,
 * method II :5092-5446): `count` Galois automorphisms of ONE ciphertext per batch item.  The INTT of the
 * ciphertext, the digit decomposition and the forward NTT of the digits do not depend on the element and run
 * once; per element: key inner product, INTT, mod-down + permutation, NTT.  Entry i is written to
 * out + i * 2 l N (per item: out_stride apart), l = Q - depth; galois_elts[i] == 0 copies the input (the
 * reference's result[0] / a zero shift).  galois_keys / galois_elts are HOST arrays of length count;
 * galois_keys[i] is the DEVICE key of element i (ignored for 0).  Every entry is bit-identical to
 * hegpu_ckks_apply_galois with the same element and key.  Workspace: HEGPU_OP_CKKS_GALOIS is enough; with
 * HEGPU_OP_CKKS_ROTATE_HOISTED (room for four accumulators) the inner products of four elements at a time share one
 * read of the digits. */
int hegpu_ckks_rotate_hoisted(hegpu_context* ctx, const uint64_t* ct, uint64_t ct_stride, uint64_t* out,
                              uint64_t out_stride, const uint64_t* const* galois_keys, const int* galois_elts,
                              int count, int depth, int batch, void* ws, size_t ws_bytes, hegpu_stream stream);
/* multiply_bfv (src/lib/host/bfv/operator.cu:336-430): coefficient domain,
 * ct [2][Q][N] x [2][Q][N] -> out [3][Q][N] */
int hegpu_bfv_multiply(hegpu_context* ctx, const uint64_t* ct1, uint64_t ct1_stride, const uint64_t* ct2,
                       uint64_t ct2_stride, uint64_t* out, uint64_t out_stride, int batch, void* ws,
                       size_t ws_bytes, hegpu_stream stream);
/* relinearize_seal_method_inplace (bfv/operator.cu:505-583) */
int hegpu_bfv_relinearize_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t ct_stride,
                                  const uint64_t* relin_key, int batch, void* ws, size_t ws_bytes,
                                  hegpu_stream stream);
/* apply_galois_method_I (bfv/operator.cu:771-864); the result buffer must not contain the input (as above) */
int hegpu_bfv_apply_galois(hegpu_context* ctx, const uint64_t* ct, uint64_t ct_stride, uint64_t* out,
                           uint64_t out_stride, const uint64_t* galois_key, int galois_elt, int batch, void* ws,
                           size_t ws_bytes, hegpu_stream stream);

/* ---- key generation, encryption, decryption (key-switching method I) ----
 * The random values are the backend's own: a ChaCha20 counter-mode PRF keyed by a 256-bit seed
 * (csrc/drbg.hpp; the reference's rngongpu::RNG<AES> is AES-CTR seeded from RAND_bytes and cannot be
 * reproduced, only its distributions: uniform mod q_i, rounded Gaussian sigma = 3.2 clipped at
 * 6 sigma, uniform ternary; src/include/heongpu/util/random.cuh:52-708).  Workspaces:
 * hegpu_workspace_bytes(ctx, HEGPU_OP_KEYGEN_* / HEGPU_OP_CKKS_ENCRYPT, 0, 1).  All keys NTT domain. */
typedef struct hegpu_rng hegpu_rng;
/* 256 bits from the operating system (getrandom): the constructor to use for real keys */
int hegpu_rng_create_from_entropy(hegpu_rng** out);
/* caller-provided 256-bit seed (e.g. from an HSM or a KDF) */
int hegpu_rng_create_seeded(const uint8_t seed[32], hegpu_rng** out);
/* REPRODUCIBLE TESTS ONLY -- a 64-bit seed is brute-forceable: key words 0,1 = seed, rest zero */
int hegpu_rng_create(uint64_t seed, hegpu_rng** out);
/* the PRF itself (host side): first 128 bits of the ChaCha20 block with counter = index, nonce =
 * stream; pinned by the RFC 8439 section 2.3.2 vector in tests/test_cabi.py */
int hegpu_drbg_block(const uint8_t key[32], uint64_t stream, uint64_t index, uint32_t out[4]);
void hegpu_rng_destroy(hegpu_rng* rng);
/* HEKeyGenerator::generate_secret_key_v2 (src/lib/host/ckks/keygenerator.cu:85-160):
 * ternary with exactly hamming_weight non-zeros; sk [Q'][N].  Synchronises the stream once. */
int hegpu_generate_secret_key(hegpu_context* ctx, hegpu_rng* rng, int hamming_weight, uint64_t* sk, void* ws,
                              size_t ws_bytes, hegpu_stream stream);
/* generate_public_key (keygenerator.cu:167-240, kernel/keygeneration.cu:93-116); pk [2][Q'][N] */
int hegpu_generate_public_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, uint64_t* pk, void* ws,
                              size_t ws_bytes, hegpu_stream stream);
/* generate_relin_key_method_I / _II (keygenerator.cu:242-414, keygeneration.cu:145-185, 584-629);
 * rk [d][2][Q'][N], d = Q (method I, P_size == 1) or the number of digits of the depth-0 partition
 * (method II: ceil(Q / P_size) for CKKS, ceil(Q / 2) for BFV) */
int hegpu_generate_relin_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, uint64_t* rk, void* ws,
                             size_t ws_bytes, hegpu_stream stream);
/* HEKeyGenerator::generate_switch_key (ckks/keygenerator.cu:996-1250, switchkey_gen_kernel /
 * switchkey_gen_II_kernel keygeneration.cu:896-989): key under new_sk that carries old_sk; layout of a
 * relinearisation key */
int hegpu_generate_switch_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* new_sk, const uint64_t* old_sk,
                              uint64_t* swk, void* ws, size_t ws_bytes, hegpu_stream stream);
/* generate_galois_key_method_I / _II, one element (keygenerator.cu:415-990, galoiskey_gen_kernel /
 * galoiskey_gen_II_kernel keygeneration.cu:742-858) */
int hegpu_generate_galois_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, int galois_elt, uint64_t* gk,
                              void* ws, size_t ws_bytes, hegpu_stream stream);
/* HEEncryptor<CKKS>::encrypt_ckks (src/lib/host/ckks/encryptor.cu:36-110); plain [Q][N], ct [2][Q][N] */
int hegpu_ckks_encrypt(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* pk, const uint64_t* plain, uint64_t* ct,
                       void* ws, size_t ws_bytes, hegpu_stream stream);
/* HEDecryptor<CKKS>::decrypt_ckks (src/lib/host/ckks/decryptor.cu:38-58); plain [Q-depth][N] */
int hegpu_ckks_decrypt(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, int depth, uint64_t* plain,
                       hegpu_stream stream);
/* HEEncryptor<BFV>::encrypt_bfv (src/lib/host/bfv/encryptor.cu:39-108, kernel/encryption.cu:91-179):
 * plain [N] residues mod t (what the batch encoder produces), ct [2][Q][N] coefficient domain */
int hegpu_bfv_encrypt(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* pk, const uint64_t* plain, uint64_t* ct,
                      void* ws, size_t ws_bytes, hegpu_stream stream);
/* HEDecryptor<BFV>::decrypt_bfv (src/lib/host/bfv/decryptor.cu:36-120, kernel/decryption.cu:10-120);
 * coefficient-domain 2-part ciphertext, plain [N] mod t.  Workspace HEGPU_OP_BFV_DECRYPT. */
int hegpu_bfv_decrypt(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, uint64_t* plain, void* ws,
                      size_t ws_bytes, hegpu_stream stream);

/* HEEncoder<BFV>::encode_bfv / decode_bfv (src/lib/host/bfv/encoder.cu:21-95, 213-249,
 * kernel/encoding.cu:11-41): batching over the slots of Z_t[X]/(X^N+1).  message: device int64
 * [message_size <= N] (negative values wrap mod t, missing slots are zero); plain [N] mod t. */
int hegpu_bfv_encode(hegpu_context* ctx, const int64_t* message, int message_size, uint64_t* plain,
                     hegpu_stream stream);
int hegpu_bfv_decode(hegpu_context* ctx, const uint64_t* plain, uint64_t* message, void* ws, size_t ws_bytes,
                     hegpu_stream stream);

/* HEEncoder<CKKS>::encode_ckks / decode_ckks for real vectors (src/lib/host/ckks/encoder.cu:21-160,
 * 449-513; kernel/encoding.cu:143-392).  The special FFT over the rotation group replaces
 * gpufft::GPU_Special_FFT (thirdparty/GPU-FFT, unvendored; HEAAN's fftSpecial/fftSpecialInv as the
 * root tables of encoder.cu:40-90 imply).  message: device doubles, at most N/2 slots (missing
 * slots are zero); plain [Q - depth][N], NTT domain.  FP64: results agree with the CPU oracle
 * bit for bit (same operation order, no FMA contraction); the reference's own rounding order
 * inside GPU-FFT is not known. */
int hegpu_ckks_encode(hegpu_context* ctx, const double* message, int message_size, double scale, uint64_t* plain,
                      void* ws, size_t ws_bytes, hegpu_stream stream);
int hegpu_ckks_decode(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message, void* ws,
                      size_t ws_bytes, hegpu_stream stream);
/* The other encodings of HEEncoder<CKKS> (ckks/encoder.cu):
 *  _complex: vector<Complex64> into / out of the slots (:294-352, :515-584); message = N/2 (re, im) pairs
 *  _coeff:   encoding::COEFFICIENT -- the message is the polynomial itself, round(m_i * scale), at most N
 *            values (:222-261, encode_kernel_coeff_ckks_conversion encoding.cu:106-137); decoding returns the
 *            N coefficients (:586-635, decode_kernel_coeff_ckks_compose encoding.cu:387-464)
 *  _scalar:  one double (or an int64 cast to double) in every slot: the constant polynomial
 *            round(value * scale), written directly in the NTT domain (:412-446,
 *            encode_kernel_double_ckks_conversion encoding.cu:43-77) */
int hegpu_ckks_encode_complex(hegpu_context* ctx, const double* message, int message_size, double scale,
                              uint64_t* plain, void* ws, size_t ws_bytes, hegpu_stream stream);
int hegpu_ckks_decode_complex(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message,
                              void* ws, size_t ws_bytes, hegpu_stream stream);
int hegpu_ckks_encode_coeff(hegpu_context* ctx, const double* message, int message_size, double scale, uint64_t* plain,
                            hegpu_stream stream);
int hegpu_ckks_decode_coeff(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message,
                            void* ws, size_t ws_bytes, hegpu_stream stream);
int hegpu_ckks_encode_scalar(hegpu_context* ctx, double value, double scale, uint64_t* plain, hegpu_stream stream);

/* HEDecryptor<BFV>::remainder_noise_budget, device part (src/lib/host/bfv/decryptor.cu:170-225):
 * out [Q][N] = t * (c0 + c1*s) mod q_j in the coefficient domain; the CRT composition and the
 * infinity norm are host work (include/heongpu/heongpu.hpp) */
int hegpu_bfv_noise_rns(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, uint64_t* out, hegpu_stream stream);

/* ---- ciphertext (x) plaintext
 * cipherplain_kernel (src/lib/kernel/multiplication.cu:298-311): out[z][j] = ct[z][j] * plain[j] for the
 * two parts, `limbs` limbs each, everything in the NTT domain (CKKS multiply_plain,
 * ckks/operator.cu multiply_plain_ckks).  CKKS add/sub of a plaintext is hegpu_addition on part 0. */
int hegpu_cipherplain_multiplication(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out,
                                     int limbs, hegpu_stream stream);
/* CKKS ciphertext with one real constant (HEArithmeticOperator::add_plain / sub_plain / multiply_plain with a
 * double, ckks/operator.cuh:312-390, :507-585, :812-925): value = constant * scale, rounded to the
 * nearest integer.  op 0: part 0 += value (addition_constant_plain_ckks_poly, addition.cu:219-258), 1: part 0
 * -= value (:260-300), 2: every part *= value (cipher_constant_plain_multiplication_kernel,
 * multiplication.cu:333-372).  ct, out: [parts][limbs][N], NTT domain; in place allowed. */
enum { HEGPU_CONST_ADD = 0, HEGPU_CONST_SUB = 1, HEGPU_CONST_MUL = 2 };
int hegpu_ckks_constant_op(hegpu_context* ctx, int op, const uint64_t* ct, double value, uint64_t* out, int limbs,
                           int parts, hegpu_stream stream);
/* add_constant_plain_ckks_v2 / multiply_const_plain_ckks_v2 (ckks/operator.cu:567-724, kernels
 * multiplication.cu:497-570): the Gaussian integer round(re) + round(im) i in every slot of an NTT-domain
 * ciphertext; op 0: added to part 0, op 1: every part multiplied by it.  re / im are the already scaled
 * values (the caller applies input.scale resp. the modulus factor exactly as the reference's host code does). */
int hegpu_ckks_gaussian_integer_op(hegpu_context* ctx, int op, const uint64_t* ct, double re, double im, uint64_t* out,
                                   int limbs, int parts, hegpu_stream stream);
/* HEArithmeticOperator::mult_i / div_i (cipher_mult_by_i_kernel / cipher_div_by_i_kernel,
 * multiplication.cu:441-495): every slot times +-i, no level consumed */
int hegpu_ckks_mult_i(hegpu_context* ctx, const uint64_t* ct, uint64_t* out, int limbs, int parts, int divide,
                      hegpu_stream stream);
/* addition_plain_bfv_poly / substraction_plain_bfv_poly (src/lib/kernel/addition.cu:50-176): part 0 gets
 * +-(floor(Q/t)*m + fix), part 1 is copied; plain [N] mod t, coefficient-domain ciphertext */
int hegpu_bfv_plain_addsub(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out, int sub,
                           hegpu_stream stream);
/* HEOperator<BFV>::multiply_plain_bfv (src/lib/host/bfv/operator.cu:432-503): threshold lift, NTT,
 * cipherplain product, INTT.  Workspace HEGPU_OP_BFV_MULTIPLY_PLAIN. */
/* HEArithmeticOperator<BFV>::transform_to_ntt(Plaintext) (bfv/operator.cu:1398-1431): plain [N] mod t -> [Q][N],
 * threshold lift (multiplication.cu:274-296) + forward NTT; such a plaintext multiplies an NTT-domain
 * ciphertext with hegpu_cipherplain_multiplication.  (Ciphertexts change domain with hegpu_ntt.) */
int hegpu_bfv_plain_to_ntt(hegpu_context* ctx, const uint64_t* plain, uint64_t* out, hegpu_stream stream);
/* HEArithmeticOperator<BFV>::multiply_power_of_X (negacyclic_shift_poly_coeffmod_kernel, switchkey.cu:1433-1457):
 * out = in * X^shift in the coefficient domain, [parts][limbs][N]; out must not alias in */
int hegpu_negacyclic_shift(hegpu_context* ctx, const uint64_t* in, uint64_t* out, int shift, int limbs, int parts,
                           hegpu_stream stream);
int hegpu_bfv_multiply_plain(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out, void* ws,
                             size_t ws_bytes, hegpu_stream stream);

/* ------------------------------------------------------------------ TFHE
 * Gate bootstrapping on the reference's fixed STD128 set
 * (src/lib/host/tfhe/context.cu:15-57: n=512, N=1024, k=1, l=2, Bg=2^10,
 * ks_base_bit=2, ks_length=8, NTT prime 1152921504606877697).
 * LWE ciphertexts: a [shape][n] + b [shape], int32 torus
 * (src/lib/host/tfhe/operator.cu:296-314).  Boot key: uint64 NTT domain,
 * reference layout [n][k+1][l][k+1][N] (bootstrapping.cu:1037-1041);
 * key-switch key a [N*k][ks_length][base-1][n], b [N*k][ks_length][base-1]
 * (bootstrapping.cu:1385-1412). */
typedef struct hegpu_tfhe_context hegpu_tfhe_context;
enum {
    HEGPU_GATE_NAND = 0, HEGPU_GATE_AND = 1, HEGPU_GATE_AND_FIRST_NOT = 2, HEGPU_GATE_NOR = 3,
    HEGPU_GATE_OR = 4, HEGPU_GATE_XNOR = 5, HEGPU_GATE_XOR = 6, HEGPU_GATE_NOT = 7
};
/* HEContextImpl<TFHE>::HEContextImpl (tfhe/context.cu:15-57); host only */
int hegpu_tfhe_context_create(hegpu_tfhe_context** out);
void hegpu_tfhe_context_destroy(hegpu_tfhe_context* ctx);
/* "fp" 0/1: re-encode a torus32 boot key for the FP64 blind rotate (1, read by hegpu_tfhe_prepare_bootkey);
 * "ks_batched" -1/0/1/8/12/16: key switching with that many gates per workgroup sharing the key rows: by launch
 * size (-1, default: 16 per workgroup from 48 gates per call), never, always (16), always with that many;
 * "ks_pieces" -1 / 1..64: workgroups the input-coefficient loop of a gate (or group of gates) is cut into (partial sums
 * added with integer atomics: the same bits), -1 = by launch size.
 * Defaults seeded once, at creation, from HEGPU_TFHE_FP / HEGPU_TFHE_KS_BATCHED / HEGPU_TFHE_KS_PIECES when these hold
 * whole decimal integers the setter accepts (anything else is ignored). */
int hegpu_tfhe_context_set_option(hegpu_tfhe_context* ctx, const char* name, int value);
/* "n","N","k","bk_l","bk_bg_bit","ks_base_bit","ks_length","offset","bootkey_elems",
 * "prepared_bootkey_elems","kskey_a_elems","kskey_b_elems" */
long hegpu_tfhe_context_int(const hegpu_tfhe_context* ctx, const char* name);
uint64_t hegpu_tfhe_prime(const hegpu_tfhe_context* ctx);
/* One-time conversion of a boot key (reference layout, NTT domain) into the layout the
 * blind rotate reads; `prepared` holds "prepared_bootkey_elems" uint64.  Synchronises the
 * stream once: a real key (torus32 coefficients) is re-encoded for the FP64 blind rotate,
 * any other key keeps the reference's 60-bit prime (results are identical either way). */
int hegpu_tfhe_prepare_bootkey(hegpu_tfhe_context* ctx, const uint64_t* boot_key, uint64_t* prepared,
                               hegpu_stream stream);
/* Layout of a prepared boot key: 1 = FP64, 0 = integer (its header word), -1 on ANY error.  A query only: the blind
 * rotate reads the header word on the device, in the order of the call's stream, so a buffer may be filled, copied
 * (hegpu_broadcast_bytes) or overwritten on that stream right before a gate call without any host synchronisation.
 * This call drains the device, then reads the word.  `refresh` is ignored (nothing is remembered since round 5).
 * A bootstrapping call that is handed a buffer whose header is neither layout writes no outputs and raises the context's
 * bad-key flag on the device: see hegpu_tfhe_status. */
int hegpu_tfhe_prepared_format(hegpu_tfhe_context* ctx, const uint64_t* prepared, int refresh);
/* tfhe_{nand,and,and_first_not,nor,or,xnor,xor}_pre_comp_kernel / tfhe_not_comp_kernel
 * (src/lib/kernel/bootstrapping.cu:378-660); in2_* ignored for NOT */
int hegpu_tfhe_gate_precompute(hegpu_tfhe_context* ctx, int gate, int32_t* out_a, int32_t* out_b,
                               const int32_t* in1_a, const int32_t* in1_b, const int32_t* in2_a,
                               const int32_t* in2_b, int shape, hegpu_stream stream);
/* HELogicOperator<TFHE>::bootstrapping (tfhe/operator.cu:200-270): blind rotate with
 * test vector mu = encode_to_torus32(1,8) + sample extraction; out_a [shape][k*N] */
int hegpu_tfhe_bootstrapping(hegpu_tfhe_context* ctx, const int32_t* in_a, const int32_t* in_b,
                             const uint64_t* prepared_boot_key, int32_t* out_a, int32_t* out_b, int shape,
                             hegpu_stream stream);
/* The error check after a launch that the reference does with HEONGPU_CUDA_CHECK(cudaGetLastError())
 * (src/include/heongpu/util/util.cuh:47-55), for the one error this backend can only find ON THE DEVICE: the layout of a
 * prepared boot key is read by the blind-rotate kernel itself, in stream order; a buffer that is no prepared key (header word
 * neither 0 nor 1) makes it write NO outputs and raise the context's bad-key flag.  hegpu_tfhe_status drains `stream`, then
 * returns HEGPU_E_INVALID if the flag is up -- and clears it -- else 0: bootstrapping / gate / mux call + status = the
 * error at the call that caused it.  The flag belongs to the context, not to a stream: with several streams on one
 * context, drain them all before asking.  A caller that never asks hears about it at the entry of its NEXT
 * hegpu_tfhe_bootstrapping / _gate / _mux (before anything of that call is queued); no other entry looks at the flag. */
int hegpu_tfhe_status(hegpu_tfhe_context* ctx, hegpu_stream stream);
/* HELogicOperator<TFHE>::key_switching (tfhe/operator.cu:272-294).  The output sample must not overlap the input
 * sample (HEGPU_E_INVALID): the forms that cut a gate's coefficient loop over several workgroups clear the outputs first. */
int hegpu_tfhe_key_switching(hegpu_tfhe_context* ctx, const int32_t* in_a, const int32_t* in_b, int32_t* out_a,
                             int32_t* out_b, const int32_t* ks_key_a, const int32_t* ks_key_b, int shape,
                             hegpu_stream stream);
/* HELogicOperator<TFHE>::{NAND,AND,NOR,OR,XNOR,XOR} (tfhe/operator.cuh:53-640):
 * precompute -> bootstrapping -> key switching.  ws: (n + k*N + 2) * shape int32. */
int hegpu_tfhe_gate(hegpu_tfhe_context* ctx, int gate, const int32_t* in1_a, const int32_t* in1_b,
                    const int32_t* in2_a, const int32_t* in2_b, int32_t* out_a, int32_t* out_b,
                    const uint64_t* prepared_boot_key, const int32_t* ks_key_a, const int32_t* ks_key_b, int shape,
                    void* ws, size_t ws_bytes, hegpu_stream stream);

/* ---- TFHE front end: keys, bit encryption, decryption, MUX
 * (src/lib/host/tfhe/keygenerator.cu, encryptor.cu, decryptor.cu, include/.../tfhe/operator.cuh:676-800).
 * Own DRBG as for the RLWE schemes; torus noise = scaled Irwin-Hall(16) (csrc/drbg.hpp) with the
 * reference's standard deviations (tfhe/context.cu:39-42).  Keys are binary. */
int hegpu_tfhe_generate_secret_key(hegpu_tfhe_context* ctx, hegpu_rng* rng, int32_t* lwe_key /* [n] */,
                                   int32_t* tlwe_key /* [k*N] */, hegpu_stream stream);
/* boot_key: reference layout [n][k+1][l][k+1][N] (NTT domain), then hegpu_tfhe_prepare_bootkey;
 * ks_a [N*k][ks_length][base-1][n], ks_b [N*k][ks_length][base-1]; ws: N uint64 */
int hegpu_tfhe_generate_bootstrapping_key(hegpu_tfhe_context* ctx, hegpu_rng* rng, const int32_t* lwe_key,
                                          const int32_t* tlwe_key, uint64_t* boot_key, int32_t* ks_a, int32_t* ks_b,
                                          void* ws, size_t ws_bytes, hegpu_stream stream);
/* LWE encryption of torus32 messages (a bit is +-1/8 = +-2^29): a [shape][n], b [shape] */
int hegpu_tfhe_encrypt(hegpu_tfhe_context* ctx, hegpu_rng* rng, const int32_t* lwe_key, const int32_t* messages,
                       int shape, int32_t* out_a, int32_t* out_b, hegpu_stream stream);
/* phase = b - <a, key>; the bit is phase > 0 */
int hegpu_tfhe_decrypt_phase(hegpu_tfhe_context* ctx, const int32_t* lwe_key, const int32_t* a, const int32_t* b,
                             int shape, int32_t* phase, hegpu_stream stream);
/* MUX(in1, in2, control) = OR(AND(control, in1), AND(NOT control, in2)) with two bootstraps and one
 * key switch (operator.cuh:688-800).  ws: (n + 1 + 2*(k*N + 1)) * shape int32. */
int hegpu_tfhe_mux(hegpu_tfhe_context* ctx, const int32_t* in1_a, const int32_t* in1_b, const int32_t* in2_a,
                   const int32_t* in2_b, const int32_t* c_a, const int32_t* c_b, int32_t* out_a, int32_t* out_b,
                   const uint64_t* prepared_boot_key, const int32_t* ks_a, const int32_t* ks_b, int shape, void* ws,
                   size_t ws_bytes, hegpu_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* HEGPU_H */

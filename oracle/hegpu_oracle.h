/*
 * hegpu_oracle.h -- CPU restatement of HEonGPU's RNS-polynomial hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under heongpu_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / timed CPU
 * baseline.
 *
 * PARITY UNPINNED at the limb level: the reference's arithmetic core
 * (github.com/Alisah-Ozcan/GPU-NTT, version unpinned, .gitmodules:4-6) is an
 * empty submodule in /root/reference, the reference is CUDA-only and cannot
 * be built or run in this pipeline, and its tests hold no golden vectors
 * (SURVEY.md 8c).  This restatement follows the in-tree call sites, table
 * generators and kernels line by line (each function cites file:line) and is
 * pinned by (a) deterministic parameter derivation checked against
 * tests/golden/ (python big-int restatement, oracle/pyref.py) and the
 * constants hard-coded in the reference (default moduli, TFHE psi), and
 * (b) semantic encrypt->op->decrypt round trips in tests/ (the shape of the
 * reference's own gtest suite).
 */
#ifndef HEGPU_ORACLE_H
#define HEGPU_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t u64;

/* GPU-NTT Modulus64 (unvendored): {value, bit, mu}. SURVEY.md 8a-a1 / 9. */
typedef struct { u64 value, bit, mu; } omod_t;

#define O_MAX_MOD 80
#define O_MAX_BSK 64 /* defines.h:26 MAX_BSK_SIZE */

enum { O_BFV = 1, O_CKKS = 2 };

/* ---- modular arithmetic (OPERATOR64 / OPERATOR_GPU_64) ---- */
omod_t o_mod(u64 q);
u64 o_add(u64 a, u64 b, const omod_t* m);
u64 o_sub(u64 a, u64 b, const omod_t* m);
u64 o_mult(u64 a, u64 b, const omod_t* m);
u64 o_reduce_forced(u64 a, const omod_t* m);
u64 o_exp(u64 base, u64 e, const omod_t* m);
u64 o_modinv(u64 a, const omod_t* m);
/* counts Barrett calls with a*b >= 2^(2*bit): behaviour there is unpinned */
extern u64 o_barrett_domain_violations;

/* ---- number theory / parameter derivation (util.cu:127-464) ---- */
int o_is_prime(u64 v);
int o_generate_primes(u64 n, const int* bit_sizes, int count, u64* out);
int o_generate_internal_primes(u64 n, int count, u64* out);
u64 o_min_primitive_root(u64 degree, u64 q);
void o_ntt_table(u64 psi, u64 q, int n_power, u64* out);
void o_intt_table(u64 psi, u64 q, int n_power, u64* out);
u64 o_n_inverse(u64 n, u64 q);
int o_default_modulus_128(u64 n, u64* out); /* defaultmodulus.cpp:12-80 */
int o_default_modulus(u64 n, int sec_level, u64* out); /* :12-175, levels 128 / 192 / 256 */
int o_max_logq(u64 n, int sec_level);                /* util/secstdparams.h:25-79 */
int o_steps_to_galois_elt(int steps, int n, int group_order);
u64 o_splitmix64(u64 x);
/* synthetic limb data: splitmix64(seed + limb*2^32 + idx) mod q */
void o_fill_poly(u64* out, u64 seed, int limb, u64 n, u64 q);

/* ---- NTT family (GPU-NTT entry points) ---- */
void o_ntt_limb(u64* a, const u64* table, const omod_t* m, int n_power);
void o_intt_limb(u64* a, const u64* itable, const omod_t* m, u64 n_inv,
                 int n_power);
void o_gpu_ntt(const u64* in, u64* out, const u64* tables, const omod_t* mods,
               int n_power, int batch, int mod_count);
void o_gpu_intt(const u64* in, u64* out, const u64* itables,
                const omod_t* mods, const u64* n_inv, int n_power, int batch,
                int mod_count);
void o_gpu_ntt_modulus_ordered(u64* data, const u64* tables,
                               const omod_t* mods, const u64* n_inv,
                               int inverse, int n_power, int batch,
                               int mod_count, const int* order);
void o_gpu_ntt_poly_ordered(u64* data, const u64* tables, const omod_t* mods,
                            const u64* n_inv, int inverse, int n_power,
                            int batch, int mod_count, const int* order);

/* ---- context (bfv|ckks/context.cu generate()) ---- */
typedef struct octx {
    int scheme, n_power;
    u64 n;
    int Q_size, P_size, Qp_size;
    omod_t mod[O_MAX_MOD];
    u64 psi[O_MAX_MOD];
    u64 n_inv[O_MAX_MOD];
    u64 *ntt_table, *intt_table; /* [Qp][N] */
    u64 *last_q_modinv, *half, *half_mod, *factor;
    int n_last_q_modinv, n_half, n_half_mod, n_factor;
    /* CKKS leveled rescale tables (ckks/context.cu:342-368) */
    u64 *rescaled_last_q_modinv, *rescaled_half_mod, *rescaled_half;
    int n_rescaled_tri, n_rescaled_half;
    /* CKKS operator ctor tables (ckks/operator.cu:24-56) */
    int *new_prime_locations, *new_input_locations;
    int n_prime_loc, n_input_loc;
    /* BFV BEHZ (bfv/context.cu:487-705, 939-1347) */
    omod_t plain_mod, m_tilde, gamma;
    int bsk_size;
    omod_t bsk[O_MAX_BSK];
    u64 bsk_psi[O_MAX_BSK];
    u64 *base_change_matrix_Bsk;         /* [bsk][Q] */
    u64 *inv_punctured_prod_mod_base;    /* [Q] */
    u64 *base_change_matrix_m_tilde;     /* [Q] */
    u64 inv_prod_q_mod_m_tilde;
    u64 *inv_m_tilde_mod_Bsk;            /* [bsk] */
    u64 *prod_q_mod_Bsk;                 /* [bsk] */
    u64 *inv_prod_q_mod_Bsk;             /* [bsk] */
    u64 *base_change_matrix_q;           /* [Q][bsk-1] */
    u64 *base_change_matrix_msk;         /* [bsk-1] */
    u64 *inv_punctured_prod_mod_B;       /* [bsk-1] */
    u64 inv_prod_B_mod_m_sk;
    u64 *prod_B_mod_q;                   /* [Q] */
    omod_t merge_mod[O_MAX_MOD + O_MAX_BSK];
    u64 merge_psi[O_MAX_MOD + O_MAX_BSK];
    u64 merge_n_inv[O_MAX_MOD + O_MAX_BSK];
    u64 *merge_ntt_table, *merge_intt_table; /* [Q+bsk][N] */
    /* key-switching method II tables (contextpool.cpp), NULL when P_size == 1 */
    struct o_m2* m2;
} octx_t;

/* KeySwitchParameterGenerator output for one depth (contextpool.cpp:161-438) */
typedef struct o_m2_level {
    int d;            /* digits */
    int rc;           /* current Q~ size = Q' - depth */
    int *I_j, *I_location;
    u64 *Mi_inv;      /* [sum I_j] */
    u64 *matrix;      /* per digit l: [rc][I_j[l]] at offset I_location[l]*rc */
    u64 *prod;        /* [d][rc] */
    int n_matrix;
} o_m2_level_t;
typedef struct o_m2 {
    int levels;       /* 1 for BFV, Q for CKKS */
    int m;            /* digit width: 2 for BFV, P_size for CKKS */
    o_m2_level_t* lv;
} o_m2_t;
void o_m2_build(octx_t* c);
void o_m2_free(octx_t* c);
void o_base_conversion_DtoQtilde(const octx_t* c, const u64* in, u64* out, int depth);
/* switchkey.cu:480-611, 1222-1282; rc / l = current Q' / Q sizes; mode 0 no ct, 1 ct on both parts, 2 ct on part 0 */
void o_divide_round_lastq_extended(const octx_t* c, const u64* input, const u64* ct, u64* output, int rc, int l,
                                   int mode);

/* primes given explicitly (Q then P); plain_modulus used for BFV only */
octx_t* o_ctx_create(int scheme, int n_power, const u64* primes, int Q_size,
                     int P_size, u64 plain_modulus);
void o_ctx_free(octx_t* c);
/* copy a named table into out (u64), returns element count or -1 */
long o_ctx_get(const octx_t* c, const char* name, u64* out, long cap);

/* ---- kernels restated (grid loops made explicit) ---- */
/* addition.cu:10-47 */
void o_addition(const u64* a, const u64* b, u64* out, const omod_t* mods,
                int n_power, int limbs, int parts);
void o_substraction(const u64* a, const u64* b, u64* out, const omod_t* mods,
                    int n_power, int limbs, int parts);
void o_negation(const u64* a, u64* out, const omod_t* mods, int n_power,
                int limbs, int parts);
/* multiplication.cu:102-126 */
void o_cross_multiplication(const u64* in1, const u64* in2, u64* out,
                            const omod_t* mods, int n_power, int decomp_size);

/* ---- operator sequences (host orchestration restated) ---- */
/* ckks/operator.cu:796-837 ; ct layout [part][l][N], l = Q - depth */
void o_ckks_multiply(const octx_t* c, const u64* ct1, const u64* ct2,
                     u64* out3, int depth);
/* ckks/operator.cu:899-1023 ; ct3 [3][l][N] in place -> first 2 parts */
void o_ckks_relinearize(const octx_t* c, u64* ct3, const u64* relin_key,
                        int depth);
/* ckks/operator.cu:1156-1244 ; ct [2][l][N] -> [2][l-1][N] in place */
void o_ckks_rescale(const octx_t* c, u64* ct, int depth);
/* ckks/operator.cu:1422-1559 */
void o_ckks_apply_galois(const octx_t* c, const u64* ct, u64* out,
                         const u64* galois_key, int galois_elt, int depth);
void o_ckks_rotate_hoisted(const octx_t* c, const u64* ct, u64* out, const u64* const* keys,
                           const int* galois_elts, int count, int depth); /* ckks/operator.cu:4674-5446 */
/* bfv/operator.cu:336-430 */
void o_bfv_multiply(const octx_t* c, const u64* ct1, const u64* ct2,
                    u64* out3);
/* bfv/operator.cu:505-583 */
void o_bfv_relinearize(const octx_t* c, u64* ct3, const u64* relin_key);
/* bfv/operator.cu:771-864 */
void o_bfv_apply_galois(const octx_t* c, const u64* ct, u64* out,
                        const u64* galois_key, int galois_elt);

/* key-switching method II (P_size > 1): bfv/operator.cu:585-672, 866-973;
 * ckks/operator.cu:1025-1154, 1561-1720.  key layout [d][2][Q'][N]. */
void o_bfv_relinearize_II(const octx_t* c, u64* ct3, const u64* relin_key);
void o_bfv_apply_galois_II(const octx_t* c, const u64* ct, u64* out, const u64* galois_key, int galois_elt);
void o_ckks_relinearize_II(const octx_t* c, u64* ct3, const u64* relin_key, int depth);
void o_ckks_apply_galois_II(const octx_t* c, const u64* ct, u64* out, const u64* galois_key, int galois_elt,
                            int depth);

/* CPU-baseline helper: batch of independent mul+relin (OpenMP over cts) */
int o_omp_threads(void);
/* pair b = inputs (first + b) % uniq; OpenMP over the pairs; returns the threads of the region */
int o_ckks_mul_relin_batch_tiled(const octx_t* c, const u64* ct1u, const u64* ct2u, int uniq, int first,
                                 u64* out3, const u64* relin_key, int depth, int batch);
void o_ckks_mul_relin_batch(const octx_t* c, const u64* ct1, const u64* ct2,
                            u64* out3, const u64* relin_key, int depth,
                            int batch);

/* ---- TFHE gate bootstrapping (config C5; o_tfhe.c) ---- */
typedef struct otfhe otfhe_t;
otfhe_t* o_tfhe_create(void);
void o_tfhe_free(otfhe_t* c);
u64 o_tfhe_prime(const otfhe_t* c);
int32_t o_tfhe_encode_to_torus32(uint32_t mu, uint32_t m_size);
void o_tfhe_gate_pre(int32_t* out_a, int32_t* out_b, const int32_t* a1, const int32_t* b1, const int32_t* a2,
                     const int32_t* b2, int32_t encoded, int s1, int s2, int m, int n, int shape);
void o_tfhe_bootstrapping(const otfhe_t* c, const int32_t* in_a, const int32_t* in_b, const u64* boot_key,
                          int32_t* out_a, int32_t* out_b, int32_t encoded, int shape);
void o_tfhe_to_ntt(const otfhe_t* c, const int32_t* poly, u64* out);
void o_tfhe_polymul(const otfhe_t* c, const int32_t* a, const int32_t* s, int32_t* out);
void o_tfhe_key_switching(const otfhe_t* c, const int32_t* in_a, const int32_t* in_b, int32_t* out_a,
                          int32_t* out_b, const int32_t* ks_a, const int32_t* ks_b, int shape);

#ifdef __cplusplus
}
#endif
#endif

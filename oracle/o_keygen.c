/*
 * o_keygen.c -- CPU restatement of key generation, CKKS encryption and
 * decryption.  TEST INFRASTRUCTURE ONLY (see
 * hegpu_oracle.h).
 *
 * The reference's random values cannot be reproduced (AES generator seeded
 * from RAND_bytes, src/lib/util/random.cu:20-60); what is restated here is
 * (a) the backend's published sampling rule -- the ChaCha20 block function as a
 * counter-mode PRF keyed by the 256-bit seed, counter = index, nonce = stream, the three
 * samplers of random.cuh:52-708 (uniform mod q_i from 128 bits, rounded
 * Gaussian sigma = 3.2 by CDT inversion clipped at 6 sigma, uniform ternary)
 * -- written independently of the product's csrc/drbg.hpp, and (b) the
 * reference's kernels and host sequences that consume those values, index for
 * index.  PARITY UNPINNED by the reference (no vectors exist); pinned by the
 * semantic round trips in tests/test_oracle_keygen.py.
 */
#include "hegpu_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ChaCha20 block function, Bernstein's layout (256-bit key, 64-bit counter = index, 64-bit nonce =
 * stream), written from the algorithm description in RFC 8439 section 2.1-2.3; the first four
 * output words are the sample's 128 random bits. */
typedef struct { uint32_t k[8]; } okey_t;
#define ROTL32(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
static void quarter(uint32_t* s, int a, int b, int c, int d)
{
    s[a] += s[b]; s[d] ^= s[a]; s[d] = ROTL32(s[d], 16);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = ROTL32(s[b], 12);
    s[a] += s[b]; s[d] ^= s[a]; s[d] = ROTL32(s[d], 8);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = ROTL32(s[b], 7);
}
static void block(okey_t seed, u64 stream, u64 index, uint32_t out[4])
{
    uint32_t init[16] = { 0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u };
    uint32_t s[16];
    for (int i = 0; i < 8; i++) init[4 + i] = seed.k[i];
    init[12] = (uint32_t) index; init[13] = (uint32_t) (index >> 32);
    init[14] = (uint32_t) stream; init[15] = (uint32_t) (stream >> 32);
    memcpy(s, init, sizeof(s));
    for (int round = 0; round < 20; round += 2) {
        quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15);
        quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14);
    }
    for (int i = 0; i < 4; i++) out[i] = s[i] + init[i];
}
/* exported for the known-answer test (tests/test_oracle_keygen.py) */
void o_drbg_block(const uint32_t key[8], u64 stream, u64 index, uint32_t out[4])
{
    okey_t k;
    memcpy(k.k, key, sizeof(k.k));
    block(k, stream, index, out);
}

#define GAUSS_MAX 19
static void gauss_cdt(u64 t[GAUSS_MAX])
{
    /* P(|round(N(0, sigma^2))| <= k) = erf((k + 1/2) / (sigma sqrt 2)), scaled to 63 bits */
    for (int k = 0; k < GAUSS_MAX; k++) t[k] = (u64) (erf(((double) k + 0.5) / (3.2 * 1.4142135623730951)) * 9223372036854775808.0);
}

static int sample_gaussian(okey_t seed, u64 stream, u64 index, const u64* t)
{
    uint32_t w[4];
    block(seed, stream, index, w);
    u64 r = (u64) w[0] | ((u64) w[1] << 32);
    u64 u = r >> 1;
    int k = 0;
    while (k < GAUSS_MAX && u >= t[k]) k++;
    return (r & 1) ? -k : k;
}

static int sample_ternary(okey_t seed, u64 stream, u64 index)
{
    uint32_t w[4];
    block(seed, stream, index, w);
    return (int) (((u64) w[0] * 3) >> 32) - 1;
}

static u64 sample_uniform(okey_t seed, u64 stream, u64 index, u64 q)
{
    uint32_t w[4];
    block(seed, stream, index, w);
    u128 v = ((u128) ((u64) w[2] | ((u64) w[3] << 32)) << 64) | ((u64) w[0] | ((u64) w[1] << 32));
    return (u64) (v % q);
}

static u64 lift(int v, u64 q) { return v < 0 ? q - (u64) (-v) : (u64) v; }

typedef struct { okey_t seed; u64 stream; } orng_t;

/* out[poly][limb][N] */
static void fill_uniform(const octx_t* c, orng_t* r, u64* out, int limbs, int polys)
{
    u64 stream = r->stream++;
    for (int p = 0; p < polys; p++)
        for (int j = 0; j < limbs; j++)
            for (u64 n = 0; n < c->n; n++) {
                u64 e = (((u64) p * limbs + j) << c->n_power) + n;
                out[e] = sample_uniform(r->seed, stream, e, c->mod[j].value);
            }
}
static void fill_gaussian(const octx_t* c, orng_t* r, u64* out, int limbs, int polys)
{
    u64 t[GAUSS_MAX];
    gauss_cdt(t);
    u64 stream = r->stream++;
    for (int p = 0; p < polys; p++)
        for (u64 n = 0; n < c->n; n++) {
            int v = sample_gaussian(r->seed, stream, ((u64) p << c->n_power) + n, t);
            for (int j = 0; j < limbs; j++) out[(((u64) p * limbs + j) << c->n_power) + n] = lift(v, c->mod[j].value);
        }
}
static void fill_ternary(const octx_t* c, orng_t* r, u64* out, int limbs, int polys)
{
    u64 stream = r->stream++;
    for (int p = 0; p < polys; p++)
        for (u64 n = 0; n < c->n; n++) {
            int v = sample_ternary(r->seed, stream, ((u64) p << c->n_power) + n);
            for (int j = 0; j < limbs; j++) out[(((u64) p * limbs + j) << c->n_power) + n] = lift(v, c->mod[j].value);
        }
}

/* HEKeyGenerator::generate_secret_key_v2 (ckks/keygenerator.cu:85-160) with the
 * backend's generator in place of mt19937; secretkey_rns_kernel
 * (keygeneration.cu:62-88); forward NTT over the Q' limbs. */
void o_gen_secret_key(const octx_t* c, orng_t* r, int hamming_weight, u64* sk)
{
    const int n = (int) c->n, Qp = c->Qp_size;
    int* index = (int*) malloc(sizeof(int) * n);
    int* coeff = (int*) calloc(n, sizeof(int));
    for (int i = 0; i < n; i++) index[i] = i;
    u64 stream = r->stream++;
    for (int i = 0; i < hamming_weight; i++) {
        uint32_t w[4];
        block(r->seed, stream, (u64) i, w);
        int j = i + (int) (((u64) w[0] * (u64) (n - i)) >> 32);
        int tmp = index[i]; index[i] = index[j]; index[j] = tmp;
        coeff[index[i]] = (w[1] & 1) ? 1 : -1;
    }
    for (int j = 0; j < Qp; j++)
        for (int i = 0; i < n; i++) sk[((u64) j << c->n_power) + i] = coeff[i] < 0 ? c->mod[j].value - 1 : (u64) coeff[i];
    o_gpu_ntt(sk, sk, c->ntt_table, c->mod, c->n_power, Qp, Qp);
    free(index);
    free(coeff);
}

/* generate_public_key (ckks/keygenerator.cu:167-240) + publickey_gen_kernel (keygeneration.cu:93-116) */
void o_gen_public_key(const octx_t* c, orng_t* r, const u64* sk, u64* pk)
{
    const int Qp = c->Qp_size;
    const u64 sz = (u64) Qp << c->n_power;
    u64* e = (u64*) malloc(2 * sz * sizeof(u64));
    u64* a = e + sz;
    fill_uniform(c, r, a, Qp, 1);
    fill_gaussian(c, r, e, Qp, 1);
    o_gpu_ntt(e, e, c->ntt_table, c->mod, c->n_power, Qp, Qp);
    for (int y = 0; y < Qp; y++)
        for (u64 i = 0; i < c->n; i++) {
            u64 loc = i + ((u64) y << c->n_power);
            u64 t = o_mult(sk[loc], a[loc], &c->mod[y]);
            t = o_add(t, e[loc], &c->mod[y]);
            pk[loc] = o_sub(0, t, &c->mod[y]);
            pk[loc + sz] = a[loc];
        }
    free(e);
}

static uint32_t bitrev(uint32_t v, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}
/* keygeneration.cu:742-755 */
static int permutation(int index, int galois_elt, int coeff_count, int n_power)
{
    int i = index + coeff_count;
    int reversed = (int) bitrev((uint32_t) i, n_power + 1);
    int index_raw = (int) (((uint32_t) galois_elt * (uint32_t) reversed) >> 1);
    index_raw &= coeff_count - 1;
    return (int) bitrev((uint32_t) index_raw, n_power);
}
static u64 modinv_2n(u64 g, u64 two_n)
{
    for (u64 x = 1; x < two_n; x += 2)
        if (((g * x) & (two_n - 1)) == 1) return x;
    return 0;
}

/* generate_relin_key_method_I (ckks/keygenerator.cu:242-324, relinkey_gen_kernel
 * keygeneration.cu:145-185) when galois_elt == 0, else generate_galois_key_method_I for
 * one element (keygenerator.cu:415-500, galoiskey_gen_kernel keygeneration.cu:757-805). */
static void gen_switch_key(const octx_t* c, orng_t* r, const u64* sk, int galois_elt, const u64* old_sk, u64* key);
void o_gen_switch_key(const octx_t* c, orng_t* r, const u64* sk, int galois_elt, u64* key)
{
    gen_switch_key(c, r, sk, galois_elt, NULL, key);
}
/* generate_switch_key_method_I (ckks/keygenerator.cu:996-1095, switchkey_gen_kernel
 * keygeneration.cu:896-939): same sampling order, the diagonal carries old_sk */
void o_gen_switch_key_new_old(const octx_t* c, orng_t* r, const u64* new_sk, const u64* old_sk, u64* key)
{
    gen_switch_key(c, r, new_sk, 0, old_sk, key);
}
static void gen_switch_key(const octx_t* c, orng_t* r, const u64* sk, int galois_elt, const u64* old_sk, u64* key)
{
    const int Q = c->Q_size, Qp = c->Qp_size, P = c->P_size, np = c->n_power;
    /* method I: Q digits, limb y carries the payload in digit y (i == block_y, keygeneration.cu:166).
     * method II (relinkey_gen_II_kernel :584-629, galoiskey_gen_II_kernel :807-858,
     * switchkey_gen_II_kernel :941-989): d = d_leveled[0] digits, Sk_pair from
     * sk_pair_counter (contextpool.cpp:48-66): the digit index repeated I_j times, INT32_MAX for
     * the special limbs; the payload is multiplied by every special prime in turn. */
    const int d = (P == 1) ? Q : c->m2->lv[0].d;
    int sk_pair[64];
    if (P == 1) {
        for (int y = 0; y < Qp; y++) sk_pair[y] = y; /* y == Q never equals i < Q */
    } else {
        int k = 0;
        for (int l = 0; l < d; l++)
            for (int t = 0; t < c->m2->lv[0].I_j[l]; t++) sk_pair[k++] = l;
        for (int t = 0; t < P; t++) sk_pair[k++] = 0x7fffffff;
    }
    const u64 sz = ((u64) d * Qp) << np;
    u64* e = (u64*) malloc(2 * sz * sizeof(u64));
    u64* a = e + sz;
    fill_uniform(c, r, a, Qp, d);
    fill_gaussian(c, r, e, Qp, d);
    o_gpu_ntt(e, e, c->ntt_table, c->mod, np, d * Qp, Qp);
    const int inv = galois_elt ? (int) modinv_2n((u64) galois_elt, 2 * c->n) : 0;
    for (int y = 0; y < Qp; y++)
        for (int idx = 0; idx < (int) c->n; idx++) {
            u64 s = sk[idx + ((u64) y << np)];
            u64 sp = galois_elt ? sk[((u64) y << np) + permutation(idx, inv, (int) c->n, np)] : s;
            for (int i = 0; i < d; i++) {
                u64 src = idx + ((u64) y << np) + ((u64) (Qp * i) << np);
                u64 k0 = o_mult(sp, a[src], &c->mod[y]);
                k0 = o_add(k0, e[src], &c->mod[y]);
                k0 = o_sub(0, k0, &c->mod[y]);
                if (i == sk_pair[y]) {
                    u64 t = old_sk ? old_sk[idx + ((u64) y << np)] : (galois_elt ? s : o_mult(s, s, &c->mod[y]));
                    for (int j = 0; j < P; j++) t = o_mult(t, c->factor[j * Q + y], &c->mod[y]);
                    k0 = o_add(k0, t, &c->mod[y]);
                }
                u64 dst = idx + ((u64) y << np) + ((u64) (Qp * i) << (np + 1));
                key[dst] = k0;
                key[dst + ((u64) Qp << np)] = a[src];
            }
        }
    free(e);
}

/* enc_div_lastq_ckks_kernel (encryption.cu:181-252) */
static void enc_div_lastq_ckks(const octx_t* c, const u64* pk, const u64* e, u64* ct)
{
    const int np = c->n_power, Qp = c->Qp_size, Q = c->Q_size, P = c->P_size;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < Q; y++)
            for (u64 idx = 0; idx < c->n; idx++) {
                u64 last_pk[15];
                for (int i = 0; i < P; i++) {
                    u64 loc = idx + ((u64) (Q + i) << np) + (((u64) Qp << np) * z);
                    last_pk[i] = o_add(pk[loc], e[loc], &c->mod[Q + i]);
                }
                u64 loc = idx + ((u64) y << np) + (((u64) Qp << np) * z);
                u64 input = o_add(pk[loc], e[loc], &c->mod[y]);
                int location = 0;
                for (int i = 0; i < P; i++) {
                    u64 lh = o_add(last_pk[P - 1 - i], c->half[i], &c->mod[Qp - 1 - i]);
                    for (int j = 0; j < P - 1 - i; j++) {
                        u64 t1 = o_reduce_forced(lh, &c->mod[Q + j]);
                        t1 = o_sub(t1, c->half_mod[location + Q + j], &c->mod[Q + j]);
                        t1 = o_sub(last_pk[j], t1, &c->mod[Q + j]);
                        last_pk[j] = o_mult(t1, c->last_q_modinv[location + Q + j], &c->mod[Q + j]);
                    }
                    u64 t1 = o_reduce_forced(lh, &c->mod[y]);
                    t1 = o_sub(t1, c->half_mod[location + y], &c->mod[y]);
                    t1 = o_sub(input, t1, &c->mod[y]);
                    input = o_mult(t1, c->last_q_modinv[location + y], &c->mod[y]);
                    location += Qp - 1 - i;
                }
                ct[idx + ((u64) y << np) + (((u64) Q << np) * z)] = input;
            }
}

/* HEEncryptor<CKKS>::encrypt_ckks (ckks/encryptor.cu:36-110) */
void o_ckks_encrypt(const octx_t* c, orng_t* r, const u64* pk, const u64* plain, u64* ct)
{
    const int np = c->n_power, Qp = c->Qp_size, Q = c->Q_size;
    const u64 sz = (u64) Qp << np;
    u64* u = (u64*) malloc(5 * sz * sizeof(u64));
    u64* e = u + sz;
    u64* pku = e + 2 * sz;
    fill_ternary(c, r, u, Qp, 1);
    fill_gaussian(c, r, e, Qp, 2);
    o_gpu_ntt(u, u, c->ntt_table, c->mod, np, Qp, Qp);
    for (int z = 0; z < 2; z++) /* pk_u_kernel, encryption.cu:10-26 */
        for (int y = 0; y < Qp; y++)
            for (u64 i = 0; i < c->n; i++) {
                u64 loc = i + ((u64) y << np);
                pku[loc + sz * z] = o_mult(pk[loc + sz * z], u[loc], &c->mod[y]);
            }
    o_gpu_intt(pku, pku, c->intt_table, c->mod, c->n_inv, np, 2 * Qp, Qp);
    enc_div_lastq_ckks(c, pku, e, ct);
    o_gpu_ntt(ct, ct, c->ntt_table, c->mod, np, 2 * Q, Q);
    for (int y = 0; y < Q; y++) /* cipher_message_add_kernel, encryption.cu:254-267 */
        for (u64 i = 0; i < c->n; i++) {
            u64 loc = i + ((u64) y << np);
            ct[loc] = o_add(ct[loc], plain[loc], &c->mod[y]);
        }
    free(u);
}

/* HEDecryptor<CKKS>::decrypt_ckks + sk_multiplication_ckks (ckks/decryptor.cu:38-58,
 * decryption.cu:349-367) */
void o_ckks_decrypt(const octx_t* c, const u64* ct, const u64* sk, int depth, u64* plain)
{
    const int np = c->n_power, l = c->Q_size - depth;
    for (int y = 0; y < l; y++)
        for (u64 i = 0; i < c->n; i++) {
            u64 loc = i + ((u64) y << np);
            u64 c1 = o_mult(ct[loc + ((u64) l << np)], sk[loc], &c->mod[y]);
            plain[loc] = o_add(c1, ct[loc], &c->mod[y]);
        }
}

/* ------------------------------------------------------------------ BFV encryption / decryption
 * Constants of bfv/context.cu:501-516, 605-620 (generate_Q_mod_t :939-950,
 * generate_coeff_div_plain_modulus :952-983 -- GMP there, 32-bit limbs here --,
 * generate_Qi_t / Qi_gamma / Qi_inverse / mulq_inv_t / mulq_inv_gamma / inv_gamma :1239-1343). */
typedef struct {
    u64 Q_mod_t, upper_threshold, mulq_inv_t, mulq_inv_gamma, inv_gamma;
    u64 coeff_div[O_MAX_MOD], Qi_t[O_MAX_MOD], Qi_gamma[O_MAX_MOD], Qi_inverse[O_MAX_MOD];
} bfv_consts_t;

static void bfv_consts(const octx_t* c, bfv_consts_t* k)
{
    const int Q = c->Q_size;
    const omod_t* t = &c->plain_mod;
    const omod_t* g = &c->gamma;
    k->Q_mod_t = 1;
    for (int i = 0; i < Q; i++) k->Q_mod_t = o_mult(k->Q_mod_t, c->mod[i].value % t->value, t);
    k->upper_threshold = (t->value + 1) >> 1;
    /* floor(prod q / t): little-endian 32-bit limbs */
    uint32_t big[2 * O_MAX_MOD + 2];
    int len = 1;
    big[0] = 1;
    for (int i = 0; i < Q; i++) {
        /* multiply by the 64-bit prime as two 32-bit halves */
        uint32_t lo = (uint32_t) c->mod[i].value, hi = (uint32_t) (c->mod[i].value >> 32);
        uint32_t tmp[2 * O_MAX_MOD + 4];
        memset(tmp, 0, sizeof(tmp));
        for (int a = 0; a < len; a++) {
            u64 carry = 0;
            u64 v = (u64) big[a] * lo + tmp[a];
            tmp[a] = (uint32_t) v; carry = v >> 32;
            v = (u64) big[a] * hi + tmp[a + 1] + carry;
            tmp[a + 1] = (uint32_t) v; carry = v >> 32;
            for (int b = a + 2; carry; b++) { v = (u64) tmp[b] + carry; tmp[b] = (uint32_t) v; carry = v >> 32; }
        }
        len += 2;
        while (len > 1 && tmp[len - 1] == 0) len--;
        memcpy(big, tmp, sizeof(uint32_t) * len);
    }
    { /* divide by t (t < 2^62): schoolbook, most significant limb first */
        u128 rem = 0;
        for (int a = len - 1; a >= 0; a--) {
            u128 cur = (rem << 32) | big[a];
            big[a] = (uint32_t) (cur / t->value);
            rem = cur % t->value;
        }
    }
    for (int i = 0; i < Q; i++) {
        u128 rem = 0;
        for (int a = len - 1; a >= 0; a--) rem = ((rem << 32) | big[a]) % c->mod[i].value;
        k->coeff_div[i] = (u64) rem;
    }
    for (int i = 0; i < Q; i++) {
        u64 a = 1, b = 1, d = 1;
        for (int j = 0; j < Q; j++) {
            if (i == j) continue;
            a = o_mult(a, c->mod[j].value % t->value, t);
            b = o_mult(b, c->mod[j].value % g->value, g);
            d = o_mult(d, o_modinv(c->mod[j].value % c->mod[i].value, &c->mod[i]), &c->mod[i]);
        }
        k->Qi_t[i] = a; k->Qi_gamma[i] = b; k->Qi_inverse[i] = d;
    }
    u64 mt = 1, mg = 1;
    for (int i = 0; i < Q; i++) {
        mt = o_mult(mt, o_modinv(c->mod[i].value % t->value, t), t);
        mg = o_mult(mg, o_modinv(c->mod[i].value % g->value, g), g);
    }
    k->mulq_inv_t = t->value - mt;
    k->mulq_inv_gamma = g->value - mg;
    k->inv_gamma = o_modinv(g->value % t->value, t);
}

/* HEEncryptor<BFV>::encrypt_bfv (bfv/encryptor.cu:39-108) with enc_div_lastq_bfv_kernel
 * (encryption.cu:91-179): the mod-down part equals the CKKS kernel's, part 0 then gets
 * Delta*m + fix.  plain [N] mod t; ct [2][Q][N] coefficient domain. */
void o_bfv_encrypt(const octx_t* c, orng_t* r, const u64* pk, const u64* plain, u64* ct)
{
    const int np = c->n_power, Qp = c->Qp_size, Q = c->Q_size;
    const u64 sz = (u64) Qp << np;
    bfv_consts_t k;
    bfv_consts(c, &k);
    u64* u = (u64*) malloc(5 * sz * sizeof(u64));
    u64* e = u + sz;
    u64* pku = e + 2 * sz;
    fill_ternary(c, r, u, Qp, 1);
    fill_gaussian(c, r, e, Qp, 2);
    o_gpu_ntt(u, u, c->ntt_table, c->mod, np, Qp, Qp);
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < Qp; y++)
            for (u64 i = 0; i < c->n; i++) {
                u64 loc = i + ((u64) y << np);
                pku[loc + sz * z] = o_mult(pk[loc + sz * z], u[loc], &c->mod[y]);
            }
    o_gpu_intt(pku, pku, c->intt_table, c->mod, c->n_inv, np, 2 * Qp, Qp);
    enc_div_lastq_ckks(c, pku, e, ct); /* identical arithmetic up to the message (encryption.cu:103-157) */
    for (int y = 0; y < Q; y++)
        for (u64 i = 0; i < c->n; i++) {
            u64 message = plain[i];
            u64 fix = message * k.Q_mod_t;
            fix = fix + k.upper_threshold;
            fix = (u64) (long long) (int) (fix / c->plain_mod.value); /* encryption.cu:163 `int(...)` */
            u64 c0 = o_mult(message, k.coeff_div[y], &c->mod[y]);
            c0 = o_add(c0, fix, &c->mod[y]);
            u64 loc = i + ((u64) y << np);
            ct[loc] = o_add(ct[loc], c0, &c->mod[y]);
        }
    free(u);
}

/* HEDecryptor<BFV>::decrypt_bfv (bfv/decryptor.cu:36-120), coefficient-domain input:
 * NTT(c1) * s -> INTT -> decryption_kernel (decryption.cu:44-120) */
void o_bfv_decrypt(const octx_t* c, const u64* ct, const u64* sk, u64* plain)
{
    const int np = c->n_power, Q = c->Q_size;
    const u64 sz = (u64) Q << np;
    bfv_consts_t k;
    bfv_consts(c, &k);
    u64* t1 = (u64*) malloc(sz * sizeof(u64));
    o_gpu_ntt(ct + sz, t1, c->ntt_table, c->mod, np, Q, Q);
    for (int y = 0; y < Q; y++)
        for (u64 i = 0; i < c->n; i++) {
            u64 loc = i + ((u64) y << np);
            t1[loc] = o_mult(t1[loc], sk[loc], &c->mod[y]);
        }
    o_gpu_intt(t1, t1, c->intt_table, c->mod, c->n_inv, np, Q, Q);
    const omod_t* t = &c->plain_mod;
    const omod_t* g = &c->gamma;
    for (u64 idx = 0; idx < c->n; idx++) {
        u64 sum_t = 0, sum_g = 0;
        for (int i = 0; i < Q; i++) {
            u64 loc = idx + ((u64) i << np);
            u64 mt = o_add(ct[loc], t1[loc], &c->mod[i]);
            u64 g_i = o_reduce_forced(g->value, &c->mod[i]);
            mt = o_mult(mt, t->value, &c->mod[i]);
            mt = o_mult(mt, g_i, &c->mod[i]);
            mt = o_mult(mt, k.Qi_inverse[i], &c->mod[i]);
            u64 in_t = o_reduce_forced(mt, t), in_g = o_reduce_forced(mt, g);
            in_t = o_mult(in_t, k.Qi_t[i], t);
            in_g = o_mult(in_g, k.Qi_gamma[i], g);
            sum_t = o_add(sum_t, in_t, t);
            sum_g = o_add(sum_g, in_g, g);
        }
        sum_t = o_mult(sum_t, k.mulq_inv_t, t);
        sum_g = o_mult(sum_g, k.mulq_inv_gamma, g);
        u64 result;
        if (sum_g > (g->value >> 1)) {
            u64 g_t = o_reduce_forced(g->value, t), sg = o_reduce_forced(sum_g, t);
            result = o_sub(g_t, sg, t);
            result = o_add(sum_t, result, t);
            result = o_mult(result, k.inv_gamma, t);
        } else {
            u64 st = o_reduce_forced(sum_t, t), sg = o_reduce_forced(sum_g, t);
            result = o_sub(st, sg, t);
            result = o_mult(result, k.inv_gamma, t);
        }
        plain[idx] = result;
    }
    free(t1);
}

/* ------------------------------------------------------------------ BFV batch encoder
 * HEEncoder<BFV> ctor (bfv/encoder.cu:21-46: slot i sits at bit-reversed index of
 * (3^i - 1)/2 resp. (2N - 3^i - 1)/2), encode_kernel_bfv / decode_kernel_bfv
 * (encoding.cu:11-41), NTT mod t with the minimal 2N-th root (bfv/context.cu:489-499). */
static void encoding_location(const octx_t* c, int* loc)
{
    const int n = (int) c->n, m = n << 1;
    int pos = 1;
    for (int i = 0; i < n / 2; i++) {
        loc[i] = (int) bitrev((uint32_t) ((pos - 1) >> 1), c->n_power);
        pos = (pos * 3) & (m - 1);
    }
    for (int i = n / 2; i < n; i++) {
        loc[i] = (int) bitrev((uint32_t) ((m - pos - 1) >> 1), c->n_power);
        pos = (pos * 3) & (m - 1);
    }
}

void o_bfv_encode(const octx_t* c, const int64_t* message, int message_size, u64* plain)
{
    const int n = (int) c->n;
    const u64 t = c->plain_mod.value;
    int* loc = (int*) malloc(sizeof(int) * n);
    u64* itab = (u64*) malloc(sizeof(u64) * n);
    encoding_location(c, loc);
    for (int i = 0; i < n; i++) {
        int64_t v = i < message_size ? message[i] : 0;
        if (v < 0) v += (int64_t) t;
        plain[loc[i]] = (u64) v;
    }
    u64 psi = o_min_primitive_root(2 * c->n, t);
    o_intt_table(psi, t, c->n_power, itab);
    o_intt_limb(plain, itab, &c->plain_mod, o_n_inverse(c->n, t), c->n_power);
    free(loc);
    free(itab);
}

void o_bfv_decode(const octx_t* c, const u64* plain, u64* message)
{
    const int n = (int) c->n;
    const u64 t = c->plain_mod.value;
    int* loc = (int*) malloc(sizeof(int) * n);
    u64* tab = (u64*) malloc(sizeof(u64) * n);
    u64* tmp = (u64*) malloc(sizeof(u64) * n);
    encoding_location(c, loc);
    memcpy(tmp, plain, sizeof(u64) * n);
    u64 psi = o_min_primitive_root(2 * c->n, t);
    o_ntt_table(psi, t, c->n_power, tab);
    o_ntt_limb(tmp, tab, &c->plain_mod, c->n_power);
    for (int i = 0; i < n; i++) message[i] = tmp[loc[i]];
    free(loc);
    free(tab);
    free(tmp);
}

/* ------------------------------------------------------------------ TFHE front end
 * Binary keys, the boot key TGSW rows in the reference layout [n][k+1][l][k+1][N] (NTT domain;
 * bootstrapping.cu:1037-1041), the key-switch key [N][ks_length][base-1][n]
 * (bootstrapping.cu:1385-1412), LWE bit encryption and the decryption phase
 * (tfhe/keygenerator.cu, encryptor.cu, decryptor.cu).  Noise: the backend's published rule, a
 * scaled Irwin-Hall(16) sum (one FP64 multiply + rint). */
static int32_t torus_gaussian(okey_t seed, u64 stream, u64 index, double c)
{
    u64 sum = 0;
    for (int j = 0; j < 4; j++) {
        uint32_t w[4];
        block(seed, stream, 4 * index + j, w);
        sum += (u64) w[0] + w[1] + w[2] + w[3];
    }
    double g = (double) ((int64_t) sum - ((int64_t) 8 << 32));
    return (int32_t) (uint32_t) (int64_t) rint(g * c);
}
static int32_t torus_uniform(okey_t seed, u64 stream, u64 index)
{
    uint32_t w[4];
    block(seed, stream, index, w);
    return (int32_t) w[0];
}

#define TF_n 512
#define TF_N 1024
#define TF_l 2
#define TF_bg 10
#define TF_ksl 8
#define TF_ksb 2
static const double IH = 1.1547005383792517;
static const double KS_STDEV = (1.0 / 32768.0) * 0.7978845608028654, BK_STDEV = 9e-9 * 0.7978845608028654;

void o_tfhe_gen_secret(orng_t* r, int32_t* lwe_key, int32_t* tlwe_key)
{
    u64 s0 = r->stream;
    r->stream += 2;
    for (int i = 0; i < TF_n; i++) { uint32_t w[4]; block(r->seed, s0, (u64) i, w); lwe_key[i] = (int32_t) (w[0] & 1u); }
    for (int i = 0; i < TF_N; i++) { uint32_t w[4]; block(r->seed, s0 + 1, (u64) i, w); tlwe_key[i] = (int32_t) (w[0] & 1u); }
}

static void lwe_encrypt(const int32_t* key, u64 s, uint32_t msg, int32_t* a, int32_t* b, double c, okey_t seed,
                        u64 stream_a, u64 stream_e)
{
    uint32_t acc = 0;
    for (int j = 0; j < TF_n; j++) {
        a[j] = torus_uniform(seed, stream_a, s * TF_n + j);
        acc += (uint32_t) a[j] * (uint32_t) key[j];
    }
    *b = (int32_t) (acc + msg + (uint32_t) torus_gaussian(seed, stream_e, s, c));
}

void o_tfhe_gen_bootkey(const otfhe_t* c, orng_t* r, const int32_t* lwe_key, const int32_t* tlwe_key, u64* boot_key,
                        int32_t* ks_a, int32_t* ks_b)
{
    u64 s0 = r->stream;
    r->stream += 4;
    const double cb = BK_STDEV / IH, ck = KS_STDEV / IH;
    int32_t a[TF_N], b[TF_N], prod[TF_N];
    for (u64 row = 0; row < (u64) TF_n * 2 * TF_l; row++) {
        const int z = (int) (row % TF_l), y = (int) ((row / TF_l) % 2);
        const u64 i = row / (2 * TF_l);
        for (int t = 0; t < TF_N; t++) a[t] = torus_uniform(r->seed, s0, row * TF_N + t);
        o_tfhe_polymul(c, a, tlwe_key, prod);
        for (int t = 0; t < TF_N; t++)
            b[t] = (int32_t) ((uint32_t) prod[t] + (uint32_t) torus_gaussian(r->seed, s0 + 1, row * TF_N + t, cb));
        const uint32_t mu = (uint32_t) lwe_key[i] << (32 - (z + 1) * TF_bg);
        if (y == 0) a[0] = (int32_t) ((uint32_t) a[0] + mu);
        else b[0] = (int32_t) ((uint32_t) b[0] + mu);
        o_tfhe_to_ntt(c, a, boot_key + row * 2 * TF_N);
        o_tfhe_to_ntt(c, b, boot_key + row * 2 * TF_N + TF_N);
    }
    const int mask = (1 << TF_ksb) - 1;
    const u64 rows = (u64) TF_N * TF_ksl * mask;
    for (u64 s = 0; s < rows; s++) {
        const u64 v = s % mask + 1, j = (s / mask) % TF_ksl, i = s / ((u64) mask * TF_ksl);
        const uint32_t m = (uint32_t) tlwe_key[i] * (uint32_t) v * (1u << (32 - (j + 1) * TF_ksb));
        lwe_encrypt(lwe_key, s, m, ks_a + s * TF_n, ks_b + s, ck, r->seed, s0 + 2, s0 + 3);
    }
}

void o_tfhe_encrypt(orng_t* r, const int32_t* lwe_key, const int32_t* messages, int shape, int32_t* a, int32_t* b)
{
    u64 s0 = r->stream;
    r->stream += 2;
    for (int s = 0; s < shape; s++)
        lwe_encrypt(lwe_key, (u64) s, (uint32_t) messages[s], a + (u64) s * TF_n, b + s, KS_STDEV / IH, r->seed, s0, s0 + 1);
}

void o_tfhe_phase(const int32_t* lwe_key, const int32_t* a, const int32_t* b, int shape, int32_t* phase)
{
    for (int s = 0; s < shape; s++) {
        uint32_t acc = 0;
        for (int j = 0; j < TF_n; j++) acc += (uint32_t) a[(u64) s * TF_n + j] * (uint32_t) lwe_key[j];
        phase[s] = (int32_t) ((uint32_t) b[s] - acc);
    }
}

/* ---- BFV plaintext lift and ciphertext (x) plaintext
 * threshold_kernel (multiplication.cu:274-296) with upper_halfincrement = q_j - t and
 * upper_threshold = (t + 1) >> 1 (bfv/context.cu:501-508), forward NTT:
 * HEOperator<BFV>::transform_to_ntt_bfv_plain (bfv/operator.cu:1398-1431); out [Q][N] */
void o_bfv_plain_to_ntt(const octx_t* c, const u64* plain, u64* out)
{
    const int Q = c->Q_size, np = c->n_power;
    const u64 t = c->plain_mod.value, thr = (t + 1) >> 1;
    for (int y = 0; y < Q; y++)
        for (u64 i = 0; i < c->n; i++) {
            const u64 v = plain[i];
            out[i + ((u64) y << np)] = (v >= thr) ? o_add(v, c->mod[y].value - t, &c->mod[y]) : v;
        }
    o_gpu_ntt(out, out, c->ntt_table, c->mod, np, Q, Q);
}

/* HEOperator<BFV>::multiply_plain_bfv, coefficient-domain ciphertext (bfv/operator.cu:432-503):
 * lift + NTT of the plaintext, NTT of both parts, cipherplain_kernel (multiplication.cu:298-311), INTT */
void o_bfv_multiply_plain(const octx_t* c, const u64* ct, const u64* plain, u64* out)
{
    const int Q = c->Q_size, np = c->n_power;
    u64* pl = (u64*) malloc(sizeof(u64) * ((u64) Q << np));
    o_bfv_plain_to_ntt(c, plain, pl);
    o_gpu_ntt(ct, out, c->ntt_table, c->mod, np, 2 * Q, Q);
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < Q; y++)
            for (u64 i = 0; i < c->n; i++) {
                const u64 a = i + ((u64) y << np), b = a + (((u64) Q << np) * z);
                out[b] = o_mult(out[b], pl[a], &c->mod[y]);
            }
    o_gpu_intt(out, out, c->intt_table, c->mod, c->n_inv, np, 2 * Q, Q);
    free(pl);
}

/* negacyclic_shift_poly_coeffmod_kernel (switchkey.cu:1433-1457); in, out [parts][limbs][N] */
void o_negacyclic_shift(const octx_t* c, const u64* in, u64* out, int shift, int limbs, int parts)
{
    const int np = c->n_power;
    const int mask = (1 << np) - 1;
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++)
            for (int idx = 0; idx < (int) c->n; idx++) {
                const u64 base = ((u64) y << np) + (((u64) limbs << np) * z);
                const int raw = idx + shift;
                u64 v = in[idx + base];
                if ((raw >> np) & 1) v = c->mod[y].value - v;
                out[(raw & mask) + base] = v;
            }
}

/*
 * o_context.c -- parameter / table derivation of the oracle.
 * TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h).  Follows
 *   ckks/context.cu:301-368, bfv/context.cu:423-705, 939-1347,
 *   util.cu:701-767, ckks/operator.cu:24-56.
 */
#include "hegpu_oracle.h"
#include <stdlib.h>
#include <string.h>

static u64* alloc64(size_t n) { return (u64*) calloc(n ? n : 1, sizeof(u64)); }

/* bfv/context.cu:76-93 modInverse via extended gcd, unsigned wraparound
 * arithmetic exactly as written (only ever called with m = 2^32). */
static u64 ext_gcd(u64 a, u64 b, u64* x, u64* y)
{
    if (a == 0) { *x = 0; *y = 1; return b; }
    u64 x1, y1;
    u64 g = ext_gcd(b % a, a, &x1, &y1);
    *x = y1 - (b / a) * x1;
    *y = x1;
    return g;
}
static u64 mod_inverse_gcd(u64 a, u64 m)
{
    u64 x, y;
    if (ext_gcd(a, m, &x, &y) != 1) return 0;
    return (x % m + m) % m;
}

static void ckks_tables(octx_t* c)
{
    int Q = c->Q_size, Qp = c->Qp_size, P = c->P_size;
    /* ckks/context.cu:342-368 rescale tables, triangular */
    int tri = 0;
    for (int j = 0; j < Q - 1; j++) tri += (Q - 1) - j;
    c->n_rescaled_tri = tri;
    c->n_rescaled_half = Q - 1;
    c->rescaled_last_q_modinv = alloc64(tri);
    c->rescaled_half_mod = alloc64(tri);
    c->rescaled_half = alloc64(Q);
    int k = 0;
    for (int j = 0; j < Q - 1; j++) {
        int inner = (Q - 1) - j;
        c->rescaled_half[j] = c->mod[inner].value >> 1;
        for (int i = 0; i < inner; i++) {
            u64 t = c->mod[inner].value % c->mod[i].value;
            c->rescaled_last_q_modinv[k] = o_modinv(t, &c->mod[i]);
            c->rescaled_half_mod[k] = c->rescaled_half[j] % c->mod[i].value;
            k++;
        }
    }
    /* ckks/operator.cu:24-39 new_prime_locations */
    int np = 0;
    for (int i = 0, cnt = Q; i < Q; i++, cnt--) np += cnt + P;
    c->new_prime_locations = (int*) calloc(np, sizeof(int));
    c->n_prime_loc = np;
    k = 0;
    for (int i = 0, cnt = Q; i < Q; i++, cnt--) {
        for (int j = 0; j < cnt; j++) c->new_prime_locations[k++] = j;
        for (int j = 0; j < P; j++) c->new_prime_locations[k++] = Q + j;
    }
    /* ckks/operator.cu:41-51 new_input_locations */
    c->n_input_loc = 2 * (Qp - 1);
    c->new_input_locations = (int*) calloc(c->n_input_loc + 1, sizeof(int));
    k = 0;
    for (int i = 0, cnt = Qp; i < Qp - 1; i++, cnt--) {
        int sum = cnt - 1;
        for (int j = 0; j < 2; j++) {
            c->new_input_locations[k++] = sum;
            sum += cnt;
        }
    }
}

static void bfv_tables(octx_t* c, u64 plain_modulus)
{
    int Q = c->Q_size, Qp = c->Qp_size;
    u64 n = c->n;
    c->plain_mod = o_mod(plain_modulus);
    c->m_tilde = o_mod(((u64) 1) << 32); /* bfv/context.cu:510 */

    /* bfv/context.cu:518-531 bsk size; gamma = extra largest prime */
    int total_bits = 0;
    for (int i = 0; i < Qp; i++) total_bits += (int) c->mod[i].bit;
    int bsk = Qp;
    if ((int) c->plain_mod.bit + total_bits + 32 >= 61 * Q + 61) bsk++;
    c->bsk_size = bsk;
    u64 ip[O_MAX_BSK + 2];
    o_generate_internal_primes(n, bsk + 1, ip);
    for (int i = 0; i < bsk; i++) c->bsk[i] = o_mod(ip[i]);
    c->gamma = o_mod(ip[bsk]);
    for (int i = 0; i < bsk; i++)
        c->bsk_psi[i] = o_min_primitive_root(2 * n, c->bsk[i].value);

    const omod_t* q = c->mod;
    const omod_t* B = c->bsk;
    const omod_t* msk = &c->bsk[bsk - 1];

    /* :990-1013 base_matrix_q_Bsk[k*Q+i] = prod_{j!=i} q_j mod Bsk_k */
    c->base_change_matrix_Bsk = alloc64((size_t) bsk * Q);
    for (int k = 0; k < bsk; k++)
        for (int i = 0; i < Q; i++) {
            u64 t = 1;
            for (int j = 0; j < Q; j++)
                if (i != j) t = o_mult(t, q[j].value, &B[k]);
            c->base_change_matrix_Bsk[k * Q + i] = t;
        }
    /* calculate_Mi_inv (util.cu:800-822) */
    c->inv_punctured_prod_mod_base = alloc64(Q);
    for (int i = 0; i < Q; i++) {
        u64 t = 1;
        for (int j = 0; j < Q; j++)
            if (i != j) t = o_mult(t, q[j].value % q[i].value, &q[i]);
        c->inv_punctured_prod_mod_base[i] = o_modinv(t, &q[i]);
    }
    /* :1015-1035 */
    c->base_change_matrix_m_tilde = alloc64(Q);
    for (int i = 0; i < Q; i++) {
        u64 t = 1;
        for (int j = 0; j < Q; j++)
            if (i != j)
                t = o_mult(t, q[j].value % c->m_tilde.value, &c->m_tilde);
        c->base_change_matrix_m_tilde[i] = t;
    }
    /* :1037-1053 */
    {
        u64 t = 1;
        for (int i = 0; i < Q; i++)
            t = o_mult(t, q[i].value % c->m_tilde.value, &c->m_tilde);
        c->inv_prod_q_mod_m_tilde = mod_inverse_gcd(t, c->m_tilde.value);
    }
    /* :1055-1068 */
    c->inv_m_tilde_mod_Bsk = alloc64(bsk);
    for (int i = 0; i < bsk; i++)
        c->inv_m_tilde_mod_Bsk[i] = o_modinv(c->m_tilde.value, &B[i]);
    /* :1070-1103 */
    c->prod_q_mod_Bsk = alloc64(bsk);
    c->inv_prod_q_mod_Bsk = alloc64(bsk);
    for (int i = 0; i < bsk; i++) {
        u64 t = 1;
        for (int j = 0; j < Q; j++) t = o_mult(t, q[j].value, &B[i]);
        c->prod_q_mod_Bsk[i] = t;
        c->inv_prod_q_mod_Bsk[i] = o_modinv(t, &B[i]);
    }
    /* :1105-1128 base_matrix_Bsk_q[k*(bsk-1)+i] = prod_{j!=i} B_j mod q_k */
    c->base_change_matrix_q = alloc64((size_t) Q * (bsk - 1));
    for (int k = 0; k < Q; k++)
        for (int i = 0; i < bsk - 1; i++) {
            u64 t = 1;
            for (int j = 0; j < bsk - 1; j++)
                if (i != j) t = o_mult(t, B[j].value % q[k].value, &q[k]);
            c->base_change_matrix_q[k * (bsk - 1) + i] = t;
        }
    /* :1130-1150 */
    c->base_change_matrix_msk = alloc64(bsk);
    for (int i = 0; i < bsk - 1; i++) {
        u64 t = 1;
        for (int j = 0; j < bsk - 1; j++)
            if (i != j) t = o_mult(t, B[j].value, msk);
        c->base_change_matrix_msk[i] = t;
    }
    /* :1152-1173 */
    c->inv_punctured_prod_mod_B = alloc64(bsk);
    for (int i = 0; i < bsk - 1; i++) {
        u64 t = 1;
        for (int j = 0; j < bsk - 1; j++)
            if (i != j) t = o_mult(t, B[j].value, &B[i]);
        c->inv_punctured_prod_mod_B[i] = o_modinv(t, &B[i]);
    }
    /* :1175-1190 */
    {
        u64 t = 1;
        for (int i = 0; i < bsk - 1; i++) t = o_mult(t, B[i].value, msk);
        c->inv_prod_B_mod_m_sk = o_modinv(t, msk);
    }
    /* :1192-1208 */
    c->prod_B_mod_q = alloc64(Q);
    for (int i = 0; i < Q; i++) {
        u64 t = 1;
        for (int j = 0; j < bsk - 1; j++)
            t = o_mult(t, B[j].value % q[i].value, &q[i]);
        c->prod_B_mod_q[i] = t;
    }
    /* :1210-1241 merged base [q_0..q_{Q-1}, Bsk...] + its NTT tables */
    int L = Q + bsk;
    for (int i = 0; i < Q; i++) {
        c->merge_mod[i] = q[i];
        c->merge_psi[i] = c->psi[i];
    }
    for (int i = 0; i < bsk; i++) {
        c->merge_mod[Q + i] = B[i];
        c->merge_psi[Q + i] = c->bsk_psi[i];
    }
    c->merge_ntt_table = alloc64((size_t) L * n);
    c->merge_intt_table = alloc64((size_t) L * n);
    for (int i = 0; i < L; i++) {
        o_ntt_table(c->merge_psi[i], c->merge_mod[i].value, c->n_power,
                    c->merge_ntt_table + (size_t) i * n);
        o_intt_table(c->merge_psi[i], c->merge_mod[i].value, c->n_power,
                     c->merge_intt_table + (size_t) i * n);
        c->merge_n_inv[i] = o_n_inverse(n, c->merge_mod[i].value);
    }
}

octx_t* o_ctx_create(int scheme, int n_power, const u64* primes, int Q_size,
                     int P_size, u64 plain_modulus)
{
    octx_t* c = (octx_t*) calloc(1, sizeof(octx_t));
    c->scheme = scheme;
    c->n_power = n_power;
    c->n = ((u64) 1) << n_power;
    c->Q_size = Q_size;
    c->P_size = P_size;
    c->Qp_size = Q_size + P_size;
    int Qp = c->Qp_size;
    u64 n = c->n;
    c->ntt_table = alloc64((size_t) Qp * n);
    c->intt_table = alloc64((size_t) Qp * n);
    for (int i = 0; i < Qp; i++) {
        c->mod[i] = o_mod(primes[i]);
        c->psi[i] = o_min_primitive_root(2 * n, primes[i]);
        o_ntt_table(c->psi[i], primes[i], n_power,
                    c->ntt_table + (size_t) i * n);
        o_intt_table(c->psi[i], primes[i], n_power,
                     c->intt_table + (size_t) i * n);
        c->n_inv[i] = o_n_inverse(n, primes[i]);
    }
    /* util.cu:701-767 */
    int tri = 0;
    for (int i = 0; i < P_size; i++) tri += (Qp - 1) - i;
    c->n_last_q_modinv = c->n_half_mod = tri;
    c->n_half = P_size;
    c->n_factor = P_size * Q_size;
    c->last_q_modinv = alloc64(tri);
    c->half_mod = alloc64(tri);
    c->half = alloc64(P_size);
    c->factor = alloc64(c->n_factor);
    int k = 0;
    for (int i = 0; i < P_size; i++) {
        c->half[i] = c->mod[Qp - 1 - i].value >> 1;
        for (int j = 0; j < (Qp - 1) - i; j++) {
            u64 t = c->mod[Qp - 1 - i].value % c->mod[j].value;
            c->last_q_modinv[k] = o_modinv(t, &c->mod[j]);
            c->half_mod[k] = c->half[i] % c->mod[j].value;
            k++;
        }
        for (int j = 0; j < Q_size; j++)
            c->factor[i * Q_size + j] =
                c->mod[Qp - 1 - i].value % c->mod[j].value;
    }
    if (scheme == O_CKKS) ckks_tables(c);
    if (scheme == O_BFV) bfv_tables(c, plain_modulus);
    if (P_size > 1) o_m2_build(c);
    return c;
}

void o_ctx_free(octx_t* c)
{
    if (!c) return;
    free(c->ntt_table); free(c->intt_table);
    free(c->last_q_modinv); free(c->half); free(c->half_mod); free(c->factor);
    free(c->rescaled_last_q_modinv); free(c->rescaled_half_mod);
    free(c->rescaled_half);
    free(c->new_prime_locations); free(c->new_input_locations);
    free(c->base_change_matrix_Bsk); free(c->inv_punctured_prod_mod_base);
    free(c->base_change_matrix_m_tilde); free(c->inv_m_tilde_mod_Bsk);
    free(c->prod_q_mod_Bsk); free(c->inv_prod_q_mod_Bsk);
    free(c->base_change_matrix_q); free(c->base_change_matrix_msk);
    free(c->inv_punctured_prod_mod_B); free(c->prod_B_mod_q);
    free(c->merge_ntt_table); free(c->merge_intt_table);
    o_m2_free(c);
    free(c);
}

static long put(const u64* src, long n, u64* out, long cap)
{
    if (!src) return -1;
    if (n > cap) return -2;
    memcpy(out, src, n * sizeof(u64));
    return n;
}
static long put_mods(const omod_t* m, long n, u64* out, long cap)
{
    if (n > cap) return -2;
    for (long i = 0; i < n; i++) out[i] = m[i].value;
    return n;
}
static long put_int(const int* src, long n, u64* out, long cap)
{
    if (!src) return -1;
    if (n > cap) return -2;
    for (long i = 0; i < n; i++) out[i] = (u64) (long) src[i];
    return n;
}

long o_ctx_get(const octx_t* c, const char* nm, u64* out, long cap)
{
    long Q = c->Q_size, Qp = c->Qp_size, B = c->bsk_size, n = (long) c->n;
#define IS(s) (strcmp(nm, s) == 0)
    if (IS("modulus")) return put_mods(c->mod, Qp, out, cap);
    if (IS("psi")) return put(c->psi, Qp, out, cap);
    if (IS("n_inverse")) return put(c->n_inv, Qp, out, cap);
    if (IS("ntt_table")) return put(c->ntt_table, Qp * n, out, cap);
    if (IS("intt_table")) return put(c->intt_table, Qp * n, out, cap);
    if (IS("last_q_modinv"))
        return put(c->last_q_modinv, c->n_last_q_modinv, out, cap);
    if (IS("half")) return put(c->half, c->n_half, out, cap);
    if (IS("half_mod")) return put(c->half_mod, c->n_half_mod, out, cap);
    if (IS("factor")) return put(c->factor, c->n_factor, out, cap);
    if (IS("rescaled_last_q_modinv"))
        return put(c->rescaled_last_q_modinv, c->n_rescaled_tri, out, cap);
    if (IS("rescaled_half_mod"))
        return put(c->rescaled_half_mod, c->n_rescaled_tri, out, cap);
    if (IS("rescaled_half"))
        return put(c->rescaled_half, c->n_rescaled_half, out, cap);
    if (IS("new_prime_locations"))
        return put_int(c->new_prime_locations, c->n_prime_loc, out, cap);
    if (IS("new_input_locations"))
        return put_int(c->new_input_locations, c->n_input_loc, out, cap);
    if (c->m2 && !strncmp(nm, "m2_", 3)) {
        /* method II tables, all depths concatenated (same order as the product) */
        long cnt = 0;
        for (int lv = 0; lv < c->m2->levels; lv++) {
            const o_m2_level_t* L = &c->m2->lv[lv];
            long nl = 0;
            for (int g = 0; g < L->d; g++) nl += L->I_j[g];
            long k = 0;
            const u64* src = NULL;
            if (IS("m2_I_j") || IS("m2_I_location")) {
                k = L->d;
                if (cnt + k > cap) return -2;
                for (long i = 0; i < k; i++)
                    out[cnt + i] = (u64) (IS("m2_I_j") ? L->I_j[i] : L->I_location[i]);
                cnt += k;
                continue;
            } else if (IS("m2_Mi_inv")) { k = nl; src = L->Mi_inv; }
            else if (IS("m2_matrix")) { k = L->n_matrix; src = L->matrix; }
            else if (IS("m2_prod")) { k = (long) L->d * L->rc; src = L->prod; }
            else return -1;
            if (cnt + k > cap) return -2;
            memcpy(out + cnt, src, k * sizeof(u64));
            cnt += k;
        }
        return cnt;
    }
    if (c->scheme != O_BFV) return -1;
    if (IS("base_Bsk")) return put_mods(c->bsk, B, out, cap);
    if (IS("base_Bsk_psi")) return put(c->bsk_psi, B, out, cap);
    if (IS("gamma")) return put(&c->gamma.value, 1, out, cap);
    if (IS("base_change_matrix_Bsk"))
        return put(c->base_change_matrix_Bsk, B * Q, out, cap);
    if (IS("inv_punctured_prod_mod_base_array"))
        return put(c->inv_punctured_prod_mod_base, Q, out, cap);
    if (IS("base_change_matrix_m_tilde"))
        return put(c->base_change_matrix_m_tilde, Q, out, cap);
    if (IS("inv_prod_q_mod_m_tilde"))
        return put(&c->inv_prod_q_mod_m_tilde, 1, out, cap);
    if (IS("inv_m_tilde_mod_Bsk"))
        return put(c->inv_m_tilde_mod_Bsk, B, out, cap);
    if (IS("prod_q_mod_Bsk")) return put(c->prod_q_mod_Bsk, B, out, cap);
    if (IS("inv_prod_q_mod_Bsk"))
        return put(c->inv_prod_q_mod_Bsk, B, out, cap);
    if (IS("base_change_matrix_q"))
        return put(c->base_change_matrix_q, Q * (B - 1), out, cap);
    if (IS("base_change_matrix_msk"))
        return put(c->base_change_matrix_msk, B - 1, out, cap);
    if (IS("inv_punctured_prod_mod_B_array"))
        return put(c->inv_punctured_prod_mod_B, B - 1, out, cap);
    if (IS("inv_prod_B_mod_m_sk"))
        return put(&c->inv_prod_B_mod_m_sk, 1, out, cap);
    if (IS("prod_B_mod_q")) return put(c->prod_B_mod_q, Q, out, cap);
    if (IS("q_Bsk_merge_modulus"))
        return put_mods(c->merge_mod, Q + B, out, cap);
    if (IS("q_Bsk_merge_ntt_tables"))
        return put(c->merge_ntt_table, (Q + B) * n, out, cap);
    if (IS("q_Bsk_merge_intt_tables"))
        return put(c->merge_intt_table, (Q + B) * n, out, cap);
    if (IS("q_Bsk_n_inverse")) return put(c->merge_n_inv, Q + B, out, cap);
#undef IS
    return -1;
}

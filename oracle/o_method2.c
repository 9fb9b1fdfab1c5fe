/*
 * o_method2.c -- key-switching method II (hybrid, P_size > 1) of the oracle.
 * TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h; PARITY UNPINNED).  Follows
 *   src/lib/kernel/contextpool.cpp:11-438 (digit partition + D->Q~ tables),
 *   src/lib/kernel/switchkey.cu:287-398,480-611,872-927,985-1046,1222-1282,
 *   src/lib/host/bfv/operator.cu:585-672,866-973, ckks/operator.cu:1025-1154,1561-1720.
 * The fast base conversion uses a FLOAT32 overflow estimate
 * (switchkey.cu:889-906 / 1003-1020): restated with IEEE single precision,
 * same summation order, no contraction (-ffp-contract=off).
 * Caveat: the BFV kernel multiplies un-reduced `partial` values
 * (switchkey.cu:911-916); where that leaves Barrett's exact domain the
 * reference's bit pattern depends on unvendored GPU-NTT internals -- this
 * restatement (and the HIP path) reduce first, i.e. agree as residues.
 */
#include "hegpu_oracle.h"
#include "o_kernels.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static u64* alloc64(size_t n) { return (u64*) calloc(n ? n : 1, sizeof(u64)); }

void o_m2_build(octx_t* c)
{
    const int Q = c->Q_size, Qp = c->Qp_size, P = c->P_size;
    o_m2_t* M = (o_m2_t*) calloc(1, sizeof(o_m2_t));
    M->m = (c->scheme == O_BFV) ? 2 : P;   /* contextpool.hpp:29, contextpool.cpp:106 */
    M->levels = (c->scheme == O_BFV) ? 1 : Q;
    M->lv = (o_m2_level_t*) calloc(M->levels, sizeof(o_m2_level_t));
    for (int lvl = 0; lvl < M->levels; lvl++) {
        o_m2_level_t* L = &M->lv[lvl];
        const int l = Q - lvl, rc = Qp - lvl;
        /* current base: q_0..q_{l-1}, P...  (contextpool.cpp:222-224 erase pattern) */
        omod_t base[O_MAX_MOD];
        for (int i = 0; i < l; i++) base[i] = c->mod[i];
        for (int i = 0; i < P; i++) base[l + i] = c->mod[Q + i];
        /* d_counter (contextpool.cpp:11-31) */
        int I_j[O_MAX_MOD], d = 0, rem = l;
        while (rem > 0) { I_j[d++] = (rem > M->m) ? M->m : rem; rem -= M->m; }
        L->d = d; L->rc = rc;
        L->I_j = (int*) calloc(d, sizeof(int));
        L->I_location = (int*) calloc(d, sizeof(int));
        int loc = 0, nm = 0;
        for (int i = 0; i < d; i++) { L->I_j[i] = I_j[i]; L->I_location[i] = loc; loc += I_j[i]; nm += I_j[i] * rc; }
        L->n_matrix = nm;
        L->Mi_inv = alloc64(l);
        L->matrix = alloc64(nm);
        L->prod = alloc64((size_t) d * rc);
        size_t mi = 0;
        for (int g = 0; g < d; g++) {
            const int s0 = L->I_location[g], cnt = L->I_j[g];
            for (int k = 0; k < rc; k++)         /* :161-191 / 193-236 */
                for (int i = 0; i < cnt; i++) {
                    u64 t = 1;
                    for (int j = 0; j < cnt; j++)
                        if (i != j) t = o_mult(t, base[s0 + j].value % base[k].value, &base[k]);
                    L->matrix[mi++] = t;
                }
            for (int i = 0; i < cnt; i++) {      /* :238-308 */
                u64 t = 1;
                for (int j = 0; j < cnt; j++)
                    if (i != j) t = o_mult(t, base[s0 + j].value % base[s0 + i].value, &base[s0 + i]);
                L->Mi_inv[s0 + i] = o_modinv(t, &base[s0 + i]);
            }
            for (int k = 0; k < rc; k++) {       /* :368-438 */
                u64 t = 1;
                for (int j = 0; j < cnt; j++) t = o_mult(t, base[s0 + j].value % base[k].value, &base[k]);
                L->prod[(size_t) g * rc + k] = t;
            }
        }
    }
    c->m2 = M;
}

void o_m2_free(octx_t* c)
{
    if (!c->m2) return;
    for (int i = 0; i < c->m2->levels; i++) {
        o_m2_level_t* L = &c->m2->lv[i];
        free(L->I_j); free(L->I_location); free(L->Mi_inv); free(L->matrix); free(L->prod);
    }
    free(c->m2->lv);
    free(c->m2);
    c->m2 = NULL;
}

/* switchkey.cu:872-927 (bfv) / 985-1046 (leveled): digit -> Q~ fast base
 * conversion with fp32 overflow estimate; out [d][rc][N] */
static void base_conversion_DtoQtilde(const octx_t* c, const o_m2_level_t* L, const u64* in, u64* out, int l,
                                      int level)
{
    const int np = c->n_power, rc = L->rc;
    const u64 n = c->n;
#pragma omp parallel for collapse(2) schedule(static)
    for (int g = 0; g < L->d; g++)
        for (u64 x = 0; x < n; x++) {
            const int cnt = L->I_j[g], s0 = L->I_location[g];
            u64 partial[20];
            float r = 0;
            for (int i = 0; i < cnt; i++) {
                u64 t = in[x + ((u64) (s0 + i) << np)];
                partial[i] = o_mult(t, L->Mi_inv[s0 + i], &c->mod[s0 + i]);
                float div = (float) partial[i];
                float mod = (float) c->mod[s0 + i].value;
                r += (div / mod);
            }
            r = roundf(r);
            u64 r_ = (u64) r;
            for (int i = 0; i < rc; i++) {
                const omod_t* m = &c->mod[(i < l) ? i : i + level];
                u64 temp = 0;
                for (int j = 0; j < cnt; j++) {
                    u64 mu = o_reduce_forced(partial[j], m);
                    mu = o_mult(mu, L->matrix[j + i * cnt + s0 * rc], m);
                    temp = o_add(temp, mu, m);
                }
                u64 r_mul = o_mult(r_, L->prod[i + g * rc], m);
                out[x + ((u64) i << np) + (((u64) g * rc) << np)] = o_sub(temp, r_mul, m);
            }
        }
}

/* exported form of the kernel above for the kernel-level parity test of
 * hegpu_base_conversion_DtoQtilde: in [l][N] coefficient domain, out [d][rc][N] */
void o_base_conversion_DtoQtilde(const octx_t* c, const u64* in, u64* out, int depth)
{
    base_conversion_DtoQtilde(c, &c->m2->lv[depth], in, out, c->Q_size - depth, depth);
}

/* switchkey.cu:287-398 keyswitch_multiply_accumulate_leveled_method_II_kernel
 * (also covers the non-leveled bfv call with level = 0) */
static void keyswitch_mac_II(const u64* input, const u64* key, u64* output, const omod_t* mods, int first_rns,
                             int l, int rc, int d, int level, int n_power)
{
    const u64 n = ((u64) 1) << n_power;
    const u64 ko1 = (u64) first_rns << n_power, ko2 = (u64) first_rns << (n_power + 1);
    for (int y = 0; y < rc; y++) {
        const int kidx = (y < l) ? y : y + level;
        const omod_t* m = &mods[kidx];
        for (u64 x = 0; x < n; x++) {
            u64 s0 = 0, s1 = 0;
            for (int i = 0; i < d; i++) {
                u64 in = input[x + ((u64) y << n_power) + (((u64) i * rc) << n_power)];
                u64 kb = x + ((u64) kidx << n_power) + ko2 * i;
                s0 = o_add(s0, o_mult(in, key[kb], m), m);
                s1 = o_add(s1, o_mult(in, key[kb + ko1], m), m);
            }
            output[x + ((u64) y << n_power)] = s0;
            output[x + ((u64) y << n_power) + ((u64) rc << n_power)] = s1;
        }
    }
}

/* switchkey.cu:480-545 (bfv: + ct) / 545-611 (switchkey: + ct on part 0) / 1222-1282 (leveled: no ct);
 * mode 0 = no ct, 1 = ct on both parts, 2 = ct on part 0 only */
void o_divide_round_lastq_extended(const octx_t* c, const u64* input, const u64* ct, u64* output, int rc, int l, int mode);
static void divide_round_lastq_extended(const octx_t* c, const u64* input, const u64* ct, u64* output, int rc, int l,
                                        int mode)
{
    o_divide_round_lastq_extended(c, input, ct, output, rc, l, mode);
}
void o_divide_round_lastq_extended(const octx_t* c, const u64* input, const u64* ct, u64* output, int rc, int l,
                                        int mode)
{
    const int np = c->n_power, P = c->P_size, fQp = c->Qp_size, fQ = c->Q_size;
    const u64 n = c->n;
    const omod_t* mods = c->mod;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < l; y++)
            for (u64 x = 0; x < n; x++) {
                u64 last_ct[15];
                for (int i = 0; i < P; i++)
                    last_ct[i] = input[x + ((u64) (l + i) << np) + (((u64) rc << np) * z)];
                u64 in = input[x + ((u64) y << np) + (((u64) rc << np) * z)];
                int loc = 0;
                for (int i = 0; i < P; i++) {
                    u64 lh = o_add(last_ct[P - 1 - i], c->half[i], &mods[fQp - 1 - i]);
                    for (int j = 0; j < (P - 1 - i); j++) {
                        const omod_t* mj = &mods[fQ + j];
                        u64 t1 = o_reduce_forced(lh, mj);
                        t1 = o_sub(t1, c->half_mod[loc + fQ + j], mj);
                        t1 = o_sub(last_ct[j], t1, mj);
                        last_ct[j] = o_mult(t1, c->last_q_modinv[loc + fQ + j], mj);
                    }
                    u64 t1 = o_reduce_forced(lh, &mods[y]);
                    t1 = o_sub(t1, c->half_mod[loc + y], &mods[y]);
                    t1 = o_sub(in, t1, &mods[y]);
                    in = o_mult(t1, c->last_q_modinv[loc + y], &mods[y]);
                    loc += (fQp - 1 - i);
                }
                u64 o = x + ((u64) y << np) + (((u64) l << np) * z);
                if (mode == 2) output[o] = o_add(z == 0 ? ct[o] : 0ULL, in, &mods[y]);
                else output[o] = mode ? o_add(ct[o], in, &mods[y]) : in;
            }
}

static int prime_loc_offset(const octx_t* c, int depth)
{
    int counter = c->Qp_size, location = 0;
    for (int i = 0; i < depth; i++) { location += counter; counter--; }
    return location;
}

/* bfv/operator.cu:585-672 */
void o_bfv_relinearize_II(const octx_t* c, u64* ct3, const u64* key)
{
    const int np = c->n_power, Q = c->Q_size, Qp = c->Qp_size;
    const u64 n = c->n;
    const o_m2_level_t* L = &c->m2->lv[0];
    u64* temp1 = alloc64((size_t) n * Q * Qp + 2 * n * Qp);
    u64* temp2 = temp1 + (size_t) n * Q * Qp;
    base_conversion_DtoQtilde(c, L, ct3 + ((u64) Q << (np + 1)), temp1, Q, 0);
    o_gpu_ntt(temp1, temp1, c->ntt_table, c->mod, np, L->d * Qp, Qp);
    keyswitch_mac_II(temp1, key, temp2, c->mod, Qp, Q, Qp, L->d, 0, np);
    o_gpu_intt(temp2, temp2, c->intt_table, c->mod, c->n_inv, np, 2 * Qp, Qp);
    divide_round_lastq_extended(c, temp2, ct3, ct3, Qp, Q, 1);
    free(temp1);
}

/* bfv/operator.cu:866-973 */
void o_bfv_apply_galois_II(const octx_t* c, const u64* ct, u64* out, const u64* key, int galois_elt)
{
    const int np = c->n_power, Q = c->Q_size, Qp = c->Qp_size;
    const u64 n = c->n;
    const o_m2_level_t* L = &c->m2->lv[0];
    u64* temp2 = alloc64((size_t) n * Q * Qp + 2 * n * Qp);
    u64* temp3 = temp2 + (size_t) n * Q * Qp;
    base_conversion_DtoQtilde(c, L, ct + (u64) Q * n, temp2, Q, 0);
    o_gpu_ntt(temp2, temp2, c->ntt_table, c->mod, np, L->d * Qp, Qp);
    keyswitch_mac_II(temp2, key, temp3, c->mod, Qp, Q, Qp, L->d, 0, np);
    o_gpu_intt(temp3, temp3, c->intt_table, c->mod, c->n_inv, np, 2 * Qp, Qp);
    o_divide_round_lastq_permute(temp3, ct, out, c->mod, c->half, c->half_mod, c->last_q_modinv, galois_elt, np, Qp,
                                 Q, Qp, Q, c->P_size);
    free(temp2);
}

/* ckks/operator.cu:1025-1154 */
void o_ckks_relinearize_II(const octx_t* c, u64* ct3, const u64* key, int depth)
{
    const int np = c->n_power, Q = c->Q_size, Qp = c->Qp_size;
    const u64 n = c->n;
    const int l = Q - depth, rc = Qp - depth;
    const o_m2_level_t* L = &c->m2->lv[depth];
    const int* order = c->new_prime_locations + prime_loc_offset(c, depth);
    u64* c2 = ct3 + ((u64) l << (np + 1));
    o_gpu_intt(c2, c2, c->intt_table, c->mod, c->n_inv, np, l, l);
    u64* temp1 = alloc64((size_t) n * Q * Qp + 2 * n * Qp);
    u64* temp2 = temp1 + (size_t) n * Q * Qp;
    base_conversion_DtoQtilde(c, L, c2, temp1, l, depth);
    o_gpu_ntt_modulus_ordered(temp1, c->ntt_table, c->mod, c->n_inv, 0, np, L->d * rc, rc, order);
    keyswitch_mac_II(temp1, key, temp2, c->mod, Qp, l, rc, L->d, depth, np);
    o_gpu_ntt_modulus_ordered(temp2, c->intt_table, c->mod, c->n_inv, 1, np, 2 * rc, rc, order);
    divide_round_lastq_extended(c, temp2, NULL, temp1, rc, l, 0);
    o_gpu_ntt(temp1, temp1, c->ntt_table, c->mod, np, 2 * l, l);
    o_addition(temp1, ct3, ct3, c->mod, np, l, 2);
    free(temp1);
}

/* ckks/operator.cu:1561-1720 */
void o_ckks_apply_galois_II(const octx_t* c, const u64* ct, u64* out, const u64* key, int galois_elt, int depth)
{
    const int np = c->n_power, Q = c->Q_size, Qp = c->Qp_size;
    const u64 n = c->n;
    const int l = Q - depth, rc = Qp - depth;
    const o_m2_level_t* L = &c->m2->lv[depth];
    const int* order = c->new_prime_locations + prime_loc_offset(c, depth);
    u64* temp0 = alloc64((size_t) 2 * n * Q + (size_t) n * Q * Qp + 2 * n * Qp);
    u64* temp3 = temp0 + (size_t) 2 * n * Q;
    u64* temp4 = temp3 + (size_t) n * Q * Qp;
    o_gpu_intt(ct, temp0, c->intt_table, c->mod, c->n_inv, np, 2 * l, l);
    base_conversion_DtoQtilde(c, L, temp0 + (u64) l * n, temp3, l, depth);
    o_gpu_ntt_modulus_ordered(temp3, c->ntt_table, c->mod, c->n_inv, 0, np, L->d * rc, rc, order);
    keyswitch_mac_II(temp3, key, temp4, c->mod, Qp, l, rc, L->d, depth, np);
    o_gpu_ntt_modulus_ordered(temp4, c->intt_table, c->mod, c->n_inv, 1, np, 2 * rc, rc, order);
    o_divide_round_lastq_permute(temp4, temp0, out, c->mod, c->half, c->half_mod, c->last_q_modinv, galois_elt, np,
                                 rc, l, Qp, Q, c->P_size);
    o_gpu_ntt(out, out, c->ntt_table, c->mod, np, 2 * l, l);
    free(temp0);
}

/*
 * o_tfhe.c -- CPU restatement of the reference's TFHE gate-bootstrapping path
 * (config C5).  TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h; PARITY UNPINNED:
 * the reference's tests hold no vectors; the in-block NTT is in-tree here:
 * small_ntt.cu:10-126).  Fixed parameter set tfhe/context.cu:15-57:
 * n=512, N=1024, k=1, l=2, Bg=2^10, ks_base_bit=2, ks_length=8,
 * prime 1152921504606877697, psi 1689264667710614.
 */
#include "hegpu_oracle.h"
#include <stdlib.h>
#include <string.h>

#define T_N 1024
#define T_NP 10

typedef struct otfhe {
    omod_t prime;
    u64 ntt[T_N], intt[T_N];
    u64 n_inverse;
    int n, N, k, bk_l, bk_bg_bit, bg, half_bg, mask_mod, offset;
    int ks_base_bit, ks_length;
} otfhe_t;

/* tfhe/context.cu:15-57 */
otfhe_t* o_tfhe_create(void)
{
    otfhe_t* c = (otfhe_t*) calloc(1, sizeof(otfhe_t));
    c->prime = o_mod(1152921504606877697ULL);
    u64 psi = 1689264667710614ULL;
    o_ntt_table(psi, c->prime.value, T_NP, c->ntt);
    o_ntt_table(o_modinv(psi, &c->prime), c->prime.value, T_NP, c->intt);
    c->n_inverse = o_modinv(1024, &c->prime);
    c->ks_base_bit = 2; c->ks_length = 8;
    c->n = 512; c->N = 1024; c->k = 1; c->bk_l = 2; c->bk_bg_bit = 10;
    c->bg = 1 << 10; c->half_bg = c->bg >> 1; c->mask_mod = c->bg - 1;
    long long sum = 0; /* compute_offset, context.cu:70-81 */
    for (int i = 1; i <= c->bk_l; i++) sum += 1LL << (32 - i * c->bk_bg_bit);
    c->offset = (int) (sum * c->half_bg);
    return c;
}
void o_tfhe_free(otfhe_t* c) { free(c); }
u64 o_tfhe_prime(const otfhe_t* c) { return c->prime.value; }

/* tfhe/operator.cu:317-323 */
int32_t o_tfhe_encode_to_torus32(uint32_t mu, uint32_t m_size)
{
    u64 interval = ((1ULL << 63) / m_size) * 2;
    u64 phase64 = mu * interval;
    return (int32_t) (phase64 >> 32);
}

/* bootstrapping.cu:378-660: all gate pre-computations are
 * out = enc + m*(s1*in1 + s2*in2) on the 32-bit torus (wrapping uint32). */
void o_tfhe_gate_pre(int32_t* out_a, int32_t* out_b, const int32_t* a1, const int32_t* b1, const int32_t* a2,
                     const int32_t* b2, int32_t encoded, int s1, int s2, int m, int n, int shape)
{
    for (int g = 0; g < shape; g++) {
        for (int i = 0; i < n; i++) {
            uint32_t v = 0;
            v = (s1 > 0) ? v + (uint32_t) a1[g * n + i] : v - (uint32_t) a1[g * n + i];
            v = (s2 > 0) ? v + (uint32_t) a2[g * n + i] : v - (uint32_t) a2[g * n + i];
            out_a[g * n + i] = (int32_t) ((uint32_t) m * v);
        }
        uint32_t v = (uint32_t) encoded;
        v = (s1 > 0) ? v + (uint32_t) m * (uint32_t) b1[g] : v - (uint32_t) m * (uint32_t) b1[g];
        v = (s2 > 0) ? v + (uint32_t) m * (uint32_t) b2[g] : v - (uint32_t) m * (uint32_t) b2[g];
        out_b[g] = (int32_t) v;
    }
}

/* bootstrapping.cu:662-674 torus_modulus_switch_log */
static int32_t modswitch(int32_t input, int modulus_log)
{
    u64 range_log = 63 - modulus_log;
    u64 half_range = 1ULL << (range_log - 1);
    u64 r = (((u64) (uint32_t) input) << 32) + half_range;
    return (int32_t) (r >> range_log);
}

/* X^a * p - p for a in [0, 2N] (bootstrapping.cu:944-982 / 1063-1101) */
static void rotate_diff(const int32_t* p, int32_t* out, int a, int N)
{
    for (int j = 0; j < N; j++) {
        int32_t r;
        if (a < N) r = (j < a) ? -(int32_t) ((uint32_t) p[N - a + j]) : p[j - a];
        else {
            int m = a - N;
            r = (j < m) ? p[N - m + j] : -(int32_t) ((uint32_t) p[j - m]);
        }
        out[j] = (int32_t) ((uint32_t) r - (uint32_t) p[j]);
    }
}

/* HELogicOperator<TFHE>::bootstrapping (tfhe/operator.cu:200-270) followed by
 * tfhe_sample_extraction_kernel(index 0).  boot_key: [n][k+1][l][k+1][N] u64
 * NTT domain; out_a [shape][k*N], out_b [shape]. */
void o_tfhe_bootstrapping(const otfhe_t* c, const int32_t* in_a, const int32_t* in_b, const u64* boot_key,
                          int32_t* out_a, int32_t* out_b, int32_t encoded, int shape)
{
    const int n = c->n, N = c->N, k = c->k, l = c->bk_l;
    const omod_t* q = &c->prime;
    const u64 threshold = q->value >> 1;
#pragma omp parallel for schedule(dynamic)
    for (int g = 0; g < shape; g++) {
        int32_t acc[2][T_N], diff[T_N];
        u64 prod[2][2][2][T_N]; /* [y][z][c][N] = temp_boot of one gate */
        u64 poly[T_N];
        /* iteration 0 builds acc_0 = (0, X^(2N-b~) * mu) on the fly (:905-933) */
        int bN = 2 * N - modswitch(in_b[g], T_NP);
        memset(acc, 0, sizeof(acc));
        for (int j = 0; j < N; j++) {
            if (bN < N) acc[k][j] = (j < bN) ? -encoded : encoded;
            else acc[k][j] = (j < bN - N) ? encoded : -encoded;
        }
        for (int i = 0; i < n; i++) {
            int aN = modswitch(in_a[g * n + i], T_NP);
            /* the unique step reads a~ as uint32, the regular one as int32:
             * identical for a~ in [0, 2N] */
            for (int y = 0; y <= k; y++) {
                rotate_diff(acc[y], diff, aN, N);
                for (int z = 0; z < l; z++) {
                    int shift = 32 - (c->bk_bg_bit * (z + 1));
                    for (int j = 0; j < N; j++) {
                        int32_t d = (int32_t) ((((uint32_t) diff[j] + (uint32_t) c->offset) >> shift) &
                                               (uint32_t) c->mask_mod) - c->half_bg;
                        poly[j] = (d < 0) ? (u64) (q->value + (long long) d) : (u64) d;
                    }
                    o_ntt_limb(poly, c->ntt, q, T_NP);
                    const u64* bk = boot_key + (u64) i * (k + 1) * ((u64) l * (k + 1) * N) +
                                    (u64) y * ((u64) l * (k + 1) * N) + (u64) z * ((u64) (k + 1) * N);
                    for (int cc = 0; cc <= k; cc++)
                        for (int j = 0; j < N; j++) prod[y][z][cc][j] = o_mult(poly[j], bk[cc * N + j], q);
                }
            }
            for (int cc = 0; cc <= k; cc++) { /* step 2 (:1142-1312) */
                for (int j = 0; j < N; j++) {
                    u64 s = 0;
                    for (int y = 0; y <= k; y++)
                        for (int z = 0; z < l; z++) s = o_add(s, prod[y][z][cc][j], q);
                    poly[j] = s;
                }
                o_intt_limb(poly, c->intt, q, c->n_inverse, T_NP);
                for (int j = 0; j < N; j++) {
                    int32_t post = (poly[j] >= threshold) ? (int32_t) (long long) (poly[j] - q->value)
                                                          : (int32_t) (long long) poly[j];
                    acc[cc][j] = (int32_t) ((uint32_t) acc[cc][j] + (uint32_t) post);
                }
            }
        }
        /* tfhe_sample_extraction_kernel, index 0 (:1314-1347) */
        for (int y = 0; y < k; y++)
            for (int j = 0; j < N; j++)
                out_a[(u64) g * k * N + y * N + j] = (j < 1) ? acc[y][j] : -(int32_t) ((uint32_t) acc[y][N - j]);
        out_b[g] = acc[k][0];
    }
}

/* tfhe_key_switching_kernel (bootstrapping.cu:1349-1436).
 * ks_key_a [N*k][ks_length][base-1][n], ks_key_b [N*k][ks_length][base-1] */
void o_tfhe_key_switching(const otfhe_t* c, const int32_t* in_a, const int32_t* in_b, int32_t* out_a,
                          int32_t* out_b, const int32_t* ks_a, const int32_t* ks_b, int shape)
{
    const int n = c->n, Nk = c->N * c->k, len = c->ks_length, bb = c->ks_base_bit;
    const int base = 1 << bb, mask = base - 1;
    const int precision_offset = 1 << (32 - (1 + bb * len));
#pragma omp parallel for schedule(dynamic)
    for (int g = 0; g < shape; g++) {
        uint32_t acc_a[512];
        memset(acc_a, 0, sizeof(acc_a));
        uint32_t acc_b = (uint32_t) in_b[g];
        for (int i = 0; i < Nk; i++) {
            int32_t a = in_a[(u64) g * Nk + i];
            for (int i2 = 0; i2 < len; i2++) {
                int d = (int) ((((uint32_t) a + (uint32_t) precision_offset) >> (32 - ((i2 + 1) * bb))) &
                               (uint32_t) mask);
                if (d != 0) {
                    u64 row = ((u64) i * len + i2) * mask + (d - 1);
                    const int32_t* ka = ks_a + row * n;
                    for (int j = 0; j < n; j++) acc_a[j] -= (uint32_t) ka[j];
                    acc_b -= (uint32_t) ks_b[row];
                }
            }
        }
        for (int j = 0; j < n; j++) out_a[(u64) g * n + j] = (int32_t) acc_a[j];
        out_b[g] = (int32_t) acc_b;
    }
}

/* ---- helpers for the semantic tests (key generation lives in tests/) ---- */
/* int32 torus polynomial -> NTT domain residues mod the TFHE prime */
void o_tfhe_to_ntt(const otfhe_t* c, const int32_t* poly, u64* out)
{
    for (int j = 0; j < T_N; j++)
        out[j] = (poly[j] < 0) ? (u64) (c->prime.value + (long long) poly[j]) : (u64) poly[j];
    o_ntt_limb(out, c->ntt, &c->prime, T_NP);
}

/* exact negacyclic product of an int32 torus polynomial with a small integer
 * polynomial (|s_j| <= 1), result wrapped to int32 */
void o_tfhe_polymul(const otfhe_t* c, const int32_t* a, const int32_t* s, int32_t* out)
{
    u64 x[T_N], y[T_N];
    o_tfhe_to_ntt(c, a, x);
    o_tfhe_to_ntt(c, s, y);
    for (int j = 0; j < T_N; j++) x[j] = o_mult(x[j], y[j], &c->prime);
    o_intt_limb(x, c->intt, &c->prime, c->n_inverse, T_NP);
    const u64 th = c->prime.value >> 1;
    for (int j = 0; j < T_N; j++)
        out[j] = (x[j] >= th) ? (int32_t) (long long) (x[j] - c->prime.value) : (int32_t) (long long) x[j];
}

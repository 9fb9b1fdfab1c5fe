/*
 * o_ntt.c -- negacyclic NTT family of the oracle (GPU-NTT entry points as
 * used at the reference's call sites).  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (GPU-NTT is unvendored); what is pinned in-tree:
 *  - tables: util.cu:398-451 (entry j = psi^(+-)bitrev(j))
 *  - butterfly / root indexing: small_ntt.cu:18-48 (CT, root m+group),
 *    small_ntt.cu:86-123 (GS, root m+group, then * n^-1)
 *  - output slot order: switchkey.cu:1461-1476 (slot j = value at
 *    psi^(2*bitrev(j)+1)).
 */
#include "hegpu_oracle.h"
#include <string.h>

/* CooleyTukeyUnit over all stages; natural order in, bit-reversed out */
void o_ntt_limb(u64* a, const u64* table, const omod_t* m, int n_power)
{
    u64 n = ((u64) 1) << n_power;
    u64 t = n;
    for (u64 mm = 1; mm < n; mm <<= 1) {
        t >>= 1;
        for (u64 i = 0; i < mm; i++) {
            u64 j1 = 2 * i * t;
            u64 w = table[mm + i];
            for (u64 j = j1; j < j1 + t; j++) {
                u64 u = a[j];
                u64 v = o_mult(a[j + t], w, m);
                a[j] = o_add(u, v, m);
                a[j + t] = o_sub(u, v, m);
            }
        }
    }
}

/* GentlemanSandeUnit over all stages, then * n^-1 */
void o_intt_limb(u64* a, const u64* itable, const omod_t* m, u64 n_inv,
                 int n_power)
{
    u64 n = ((u64) 1) << n_power;
    u64 t = 1;
    for (u64 mm = n >> 1; mm >= 1; mm >>= 1) {
        for (u64 i = 0; i < mm; i++) {
            u64 j1 = 2 * i * t;
            u64 w = itable[mm + i];
            for (u64 j = j1; j < j1 + t; j++) {
                u64 u = a[j];
                u64 v = a[j + t];
                a[j] = o_add(u, v, m);
                a[j + t] = o_mult(o_sub(u, v, m), w, m);
            }
        }
        t <<= 1;
    }
    for (u64 j = 0; j < n; j++) a[j] = o_mult(a[j], n_inv, m);
}

/* gpuntt::GPU_NTT / GPU_NTT_Inplace (call sites bfv/operator.cu:393,
 * ckks/operator.cu:1011): poly i uses modulus/table i % mod_count. */
void o_gpu_ntt(const u64* in, u64* out, const u64* tables, const omod_t* mods,
               int n_power, int batch, int mod_count)
{
    u64 n = ((u64) 1) << n_power;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < batch; i++) {
        int k = i % mod_count;
        if (out != in) memcpy(out + i * n, in + i * n, n * sizeof(u64));
        o_ntt_limb(out + i * n, tables + k * n, &mods[k], n_power);
    }
}

/* gpuntt::GPU_INTT / GPU_INTT_Inplace (bfv/operator.cu:410,
 * ckks/operator.cu:919); n^-1 = cfg.mod_inverse[i % mod_count]. */
void o_gpu_intt(const u64* in, u64* out, const u64* itables,
                const omod_t* mods, const u64* n_inv, int n_power, int batch,
                int mod_count)
{
    u64 n = ((u64) 1) << n_power;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < batch; i++) {
        int k = i % mod_count;
        if (out != in) memcpy(out + i * n, in + i * n, n * sizeof(u64));
        o_intt_limb(out + i * n, itables + k * n, &mods[k], n_inv[k],
                    n_power);
    }
}

/* gpuntt::GPU_NTT_Modulus_Ordered_Inplace (ckks/operator.cu:956,1524):
 * poly i uses modulus/table (and n^-1) index order[i % mod_count]. */
void o_gpu_ntt_modulus_ordered(u64* data, const u64* tables,
                               const omod_t* mods, const u64* n_inv,
                               int inverse, int n_power, int batch,
                               int mod_count, const int* order)
{
    u64 n = ((u64) 1) << n_power;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < batch; i++) {
        int k = order[i % mod_count];
        if (inverse)
            o_intt_limb(data + i * n, tables + k * n, &mods[k], n_inv[k],
                        n_power);
        else
            o_ntt_limb(data + i * n, tables + k * n, &mods[k], n_power);
    }
}

/* gpuntt::GPU_NTT_Poly_Ordered_Inplace (ckks/operator.cu:996,1197): poly i
 * lives at data + order[i]*N; modulus index i % mod_count relative to the
 * caller-offset tables/mods/n_inv pointers. */
void o_gpu_ntt_poly_ordered(u64* data, const u64* tables, const omod_t* mods,
                            const u64* n_inv, int inverse, int n_power,
                            int batch, int mod_count, const int* order)
{
    u64 n = ((u64) 1) << n_power;
    for (int i = 0; i < batch; i++) {
        int k = i % mod_count;
        u64* p = data + (u64) order[i] * n;
        if (inverse)
            o_intt_limb(p, tables + k * n, &mods[k], n_inv[k], n_power);
        else
            o_ntt_limb(p, tables + k * n, &mods[k], n_power);
    }
}

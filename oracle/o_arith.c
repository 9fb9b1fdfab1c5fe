/*
 * o_arith.c -- modular arithmetic + number theory of the oracle.
 * TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h).  PARITY UNPINNED for the
 * GPU-NTT primitives: their source is not in /root/reference; the Barrett
 * record below is the published GPU-NTT algorithm restated.
 */
#include "hegpu_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

u64 o_barrett_domain_violations = 0;

/* GPU-NTT Modulus<Data64>(q): bit = floor(log2 q)+1, mu = floor(2^(2bit+1)/q).
 * Constructed from a bare prime at reference util.cu:271 `Modulus64(q)`. */
omod_t o_mod(u64 q)
{
    omod_t m;
    m.value = q;
    m.bit = 64 - (u64) __builtin_clzll(q);
    m.mu = (u64) ((((u128) 1) << (2 * m.bit + 1)) / q);
    return m;
}

/* OPERATOR_GPU_64::add -- used e.g. addition.cu:20 */
u64 o_add(u64 a, u64 b, const omod_t* m)
{
    u64 s = a + b;
    return (s >= m->value) ? (s - m->value) : s;
}

/* OPERATOR_GPU_64::sub -- used e.g. addition.cu:33.  NOTE sub(q,0) == q
 * (non-canonical), relied on by multiplication.cu:185,234. */
u64 o_sub(u64 a, u64 b, const omod_t* m)
{
    u64 d = a + m->value;
    d = d - b;
    return (d >= m->value) ? (d - m->value) : d;
}

/* OPERATOR_GPU_64::mult -- Barrett, exact and canonical whenever
 * a*b < 2^(2*bit) (one conditional subtraction suffices there). */
u64 o_mult(u64 a, u64 b, const omod_t* m)
{
    u128 z = (u128) a * b;
    if (m->bit < 64 && (z >> (2 * m->bit)) != 0)
        __atomic_fetch_add(&o_barrett_domain_violations, 1, __ATOMIC_RELAXED);
    u128 w = z >> (m->bit - 2);
    w = (u128) ((u64) w) * m->mu;
    w = w >> (m->bit + 3);
    w = (u128) ((u64) w) * m->value;
    z = z - w;
    u64 r = (u64) z;
    return (r >= m->value) ? (r - m->value) : r;
}

/* OPERATOR_GPU_64::reduce_forced -- full reduction of any 64-bit value
 * (switchkey.cu:54, multiplication.cu:62). */
u64 o_reduce_forced(u64 a, const omod_t* m) { return a % m->value; }

/* OPERATOR64::exp (util.cu:149) */
u64 o_exp(u64 base, u64 e, const omod_t* m)
{
    u64 r = 1 % m->value;
    u64 b = base % m->value;
    while (e) {
        if (e & 1) r = o_mult(r, b, m);
        b = o_mult(b, b, m);
        e >>= 1;
    }
    return r;
}

/* OPERATOR64::modinv (util.cu:412,460; prime moduli only) */
u64 o_modinv(u64 a, const omod_t* m) { return o_exp(a, m->value - 2, m); }

/* util.cu:127-166 miller_rabin.  The reference draws random bases; the
 * verdict is deterministic, so fixed bases (exact for all 64-bit inputs). */
static int miller_rabin(u64 v)
{
    static const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    omod_t m = o_mod(v);
    u64 d = v - 1, r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    if (r == 0) return 0;
    for (int i = 0; i < 12; i++) {
        u64 a = bases[i] % v;
        if (a == 0) continue;
        u64 x = o_exp(a, d, &m);
        if (x == 1 || x == v - 1) continue;
        u64 count = 0;
        int ok = 0;
        while (count < r - 1) {
            x = o_mult(x, x, &m);
            count++;
            if (x == v - 1) { ok = 1; break; }
        }
        if (!ok) return 0;
    }
    return 1;
}

/* util.cu:168-217 is_prime */
int o_is_prime(u64 v)
{
    if (v < 3 || (v & 1) == 0) return 0;
    for (u64 p = 3; p < 1000; p += 2) {
        int isp = 1;
        for (u64 d = 3; d * d <= p; d += 2)
            if (p % d == 0) { isp = 0; break; }
        if (!isp) continue;
        if (v == p) return 1;
        if (v % p == 0) return 0;
    }
    return miller_rabin(v);
}

/* util.cu:219-242 generate_proper_primes: scan down from
 * floor((2^b-1)/factor)*factor+1 in steps of factor. Returns count found. */
static int proper_primes(u64 factor, int bit_size, int count, u64* dst)
{
    u64 value = ((((u64) 1) << bit_size) - 1) / factor * factor + 1;
    u64 lower = ((u64) 1) << (bit_size - 1);
    int found = 0;
    while (found < count && value > lower) {
        if (o_is_prime(value)) dst[found++] = value;
        value -= factor;
    }
    return found;
}

/* util.cu:244-276 generate_primes: per distinct bit size collect the needed
 * number of primes (descending), hand out smallest-first in request order. */
int o_generate_primes(u64 n, const int* bit_sizes, int count, u64* out)
{
    int need[64] = {0};
    u64* tab[64] = {0};
    int left[64] = {0};
    for (int i = 0; i < count; i++) {
        if (bit_sizes[i] > 61 || bit_sizes[i] < 30) return -1;
        need[bit_sizes[i]]++;
    }
    for (int b = 0; b < 64; b++) {
        if (!need[b]) continue;
        tab[b] = (u64*) malloc(sizeof(u64) * need[b]);
        if (proper_primes(2 * n, b, need[b], tab[b]) != need[b]) return -2;
        left[b] = need[b];
    }
    for (int i = 0; i < count; i++) {
        int b = bit_sizes[i];
        out[i] = tab[b][--left[b]]; /* .back(); pop_back() */
    }
    for (int b = 0; b < 64; b++) free(tab[b]);
    return 0;
}

/* util.cu:278-310: prime_count primes of MAX_MOD_BIT_COUNT=61 bits */
int o_generate_internal_primes(u64 n, int count, u64* out)
{
    int bits[O_MAX_BSK + 2];
    if (count > O_MAX_BSK + 2) return -1;
    for (int i = 0; i < count; i++) bits[i] = 61;
    return o_generate_primes(n, bits, count, out);
}

/* util.cu:312-380: minimal primitive degree-th root (degree = 2N).  The
 * reference starts from a random primitive root and takes the minimum over
 * all odd powers, i.e. over ALL primitive roots => deterministic. */
u64 o_min_primitive_root(u64 degree, u64 q)
{
    omod_t m = o_mod(q);
    u64 group = q - 1;
    if (group % degree) return 0;
    u64 quot = group / degree;
    u64 root = 0;
    for (u64 g = 2; g < q; g++) {
        u64 cand = o_exp(g, quot, &m);
        if (o_exp(cand, degree >> 1, &m) == q - 1) { root = cand; break; }
    }
    u64 gen_sq = o_mult(root, root, &m);
    u64 cur = root;
    for (u64 i = 0; i < degree; i += 2) {
        if (cur < root) root = cur;
        cur = o_mult(cur, gen_sq, &m);
    }
    return root;
}

static inline u64 bitrev(u64 x, int bits)
{
    u64 r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* util.cu:398-423 generate_ntt_table: entry j = psi^bitreverse(j) */
void o_ntt_table(u64 psi, u64 q, int n_power, u64* out)
{
    omod_t m = o_mod(q);
    u64 n = ((u64) 1) << n_power;
    u64* t = (u64*) malloc(sizeof(u64) * n);
    t[0] = 1;
    for (u64 j = 1; j < n; j++) t[j] = o_mult(t[j - 1], psi, &m);
    for (u64 j = 0; j < n; j++) out[j] = t[bitrev(j, n_power)];
    free(t);
}

/* util.cu:425-451 generate_intt_table: entry j = psi^-bitreverse(j) */
void o_intt_table(u64 psi, u64 q, int n_power, u64* out)
{
    omod_t m = o_mod(q);
    o_ntt_table(o_modinv(psi, &m), q, n_power, out);
}

/* util.cu:453-464 */
u64 o_n_inverse(u64 n, u64 q)
{
    omod_t m = o_mod(q);
    return o_modinv(n, &m);
}

/* defaultmodulus.cpp:12-175 (data): default chains for 128 / 192 / 256-bit security */
int o_default_modulus(u64 n, int sec_level, u64* out)
{
    static const u64 p128_4096[] = {
        0x800004001ULL, 0x800008001ULL, 0x1000002001ULL};
    static const u64 p128_8192[] = {
        0x40000084001ULL, 0x400000b0001ULL, 0x8000002c001ULL, 0x80000050001ULL,
        0x80000064001ULL};
    static const u64 p128_16384[] = {
        0x800000020001ULL, 0x8000001a8001ULL, 0x8000001e8001ULL,
        0x10000000d8001ULL, 0x1000000168001ULL, 0x10000001a0001ULL,
        0x10000001e0001ULL, 0x10000002b8001ULL, 0x10000002e8001ULL};
    static const u64 p128_32768[] = {
        0x2000000002b0001ULL, 0x2000000003a0001ULL, 0x2000000005b0001ULL,
        0x200000000640001ULL, 0x400000000270001ULL, 0x400000000350001ULL,
        0x400000000360001ULL, 0x4000000004d0001ULL, 0x400000000570001ULL,
        0x400000000660001ULL, 0x4000000008a0001ULL, 0x400000000920001ULL,
        0x400000000980001ULL, 0x400000000990001ULL, 0x400000000a40001ULL};
    static const u64 p128_65536[] = {
        0x2000000003a0001ULL, 0x200000000640001ULL, 0x200000000f80001ULL,
        0x200000001460001ULL, 0x2000000015a0001ULL, 0x2000000015e0001ULL,
        0x200000001b20001ULL, 0x200000001c00001ULL, 0x200000001ee0001ULL,
        0x400000000360001ULL, 0x400000000660001ULL, 0x4000000008a0001ULL,
        0x400000000920001ULL, 0x400000000980001ULL, 0x400000000a40001ULL,
        0x400000000c00001ULL, 0x400000000ea0001ULL, 0x400000001460001ULL,
        0x400000001700001ULL, 0x400000001740001ULL, 0x4000000017a0001ULL,
        0x400000001920001ULL, 0x400000001b00001ULL, 0x400000001b60001ULL,
        0x400000001c40001ULL, 0x400000001ee0001ULL, 0x400000001f20001ULL,
        0x4000000020c0001ULL, 0x400000002360001ULL, 0x400000002480001ULL};
    static const u64 p192_4096[] = {
        0x1000002001ULL, 0x1000042001ULL};
    static const u64 p192_8192[] = {
        0x100008c001ULL, 0x1000090001ULL, 0x10000c8001ULL, 0x2000088001ULL};
    static const u64 p192_16384[] = {
        0x20000000b0001ULL, 0x2000000178001ULL, 0x20000001a0001ULL,
        0x2000000208001ULL, 0x20000003b0001ULL, 0x20000003c8001ULL};
    static const u64 p192_32768[] = {
        0x40000000120001ULL, 0x400000001d0001ULL, 0x400000002c0001ULL,
        0x40000000480001ULL, 0x40000000540001ULL, 0x400000005c0001ULL,
        0x400000006c0001ULL, 0x400000007b0001ULL, 0x40000000890001ULL,
        0x40000000b00001ULL, 0x40000000e40001ULL};
    static const u64 p192_65536[] = {
        0x40000000120001ULL, 0x400000002c0001ULL, 0x40000000480001ULL,
        0x40000000540001ULL, 0x400000005c0001ULL, 0x400000006c0001ULL,
        0x40000000b00001ULL, 0x40000000e40001ULL, 0x40000000f60001ULL,
        0x400000010a0001ULL, 0x400000011a0001ULL, 0x40000001200001ULL,
        0x40000001340001ULL, 0x400000017a0001ULL, 0x40000001c40001ULL,
        0x40000001ca0001ULL, 0x40000001d00001ULL, 0x40000002100001ULL,
        0x400000022a0001ULL, 0x400000022e0001ULL, 0x80000000080001ULL,
        0x80000000440001ULL};
    static const u64 p256_4096[] = {
        0x8008001ULL, 0x10006001ULL};
    static const u64 p256_8192[] = {
        0x2000088001ULL, 0x20000e0001ULL, 0x4000038001ULL};
    static const u64 p256_16384[] = {
        0x200000008001ULL, 0x2000000a0001ULL, 0x2000000e0001ULL,
        0x400000008001ULL, 0x400000060001ULL};
    static const u64 p256_32768[] = {
        0x4000000120001ULL, 0x40000001b0001ULL, 0x4000000270001ULL,
        0x8000000110001ULL, 0x8000000130001ULL, 0x80000001c0001ULL,
        0x80000002c0001ULL, 0x80000004d0001ULL, 0x80000004f0001ULL};
    static const u64 p256_65536[] = {
        0x4000000120001ULL, 0x4000000420001ULL, 0x4000000660001ULL,
        0x40000007e0001ULL, 0x4000000800001ULL, 0x40000008a0001ULL,
        0x7fffffffe0001ULL, 0x80000001c0001ULL, 0x80000002c0001ULL,
        0x8000000500001ULL, 0x8000000820001ULL, 0x8000000940001ULL,
        0x8000001120001ULL, 0x80000012a0001ULL, 0x8000001360001ULL,
        0x80000014c0001ULL, 0x8000001540001ULL, 0x8000001600001ULL};
    const u64* p = NULL;
    int cnt = 0;
#define PICK(L, N) do { p = p##L##_##N; cnt = (int) (sizeof(p##L##_##N) / sizeof(u64)); } while (0)
    switch (sec_level * 100000 + (int) n) {
        case 128 * 100000 + 4096: PICK(128, 4096); break;
        case 128 * 100000 + 8192: PICK(128, 8192); break;
        case 128 * 100000 + 16384: PICK(128, 16384); break;
        case 128 * 100000 + 32768: PICK(128, 32768); break;
        case 128 * 100000 + 65536: PICK(128, 65536); break;
        case 192 * 100000 + 4096: PICK(192, 4096); break;
        case 192 * 100000 + 8192: PICK(192, 8192); break;
        case 192 * 100000 + 16384: PICK(192, 16384); break;
        case 192 * 100000 + 32768: PICK(192, 32768); break;
        case 192 * 100000 + 65536: PICK(192, 65536); break;
        case 256 * 100000 + 4096: PICK(256, 4096); break;
        case 256 * 100000 + 8192: PICK(256, 8192); break;
        case 256 * 100000 + 16384: PICK(256, 16384); break;
        case 256 * 100000 + 32768: PICK(256, 32768); break;
        case 256 * 100000 + 65536: PICK(256, 65536); break;
        default: return -1;
    }
#undef PICK
    memcpy(out, p, sizeof(u64) * cnt);
    return cnt;
}
int o_default_modulus_128(u64 n, u64* out) { return o_default_modulus(n, 128, out); }

/* util/secstdparams.h:25-79: max log2(Q*P) per degree and security level (0: none) */
int o_max_logq(u64 n, int sec_level)
{
    switch (sec_level * 100000 + (int) n) {
        case 128 * 100000 + 4096: return 109;
        case 128 * 100000 + 8192: return 218;
        case 128 * 100000 + 16384: return 438;
        case 128 * 100000 + 32768: return 881;
        case 128 * 100000 + 65536: return 1761;
        case 192 * 100000 + 4096: return 74;
        case 192 * 100000 + 8192: return 149;
        case 192 * 100000 + 16384: return 300;
        case 192 * 100000 + 32768: return 605;
        case 192 * 100000 + 65536: return 1212;
        case 256 * 100000 + 4096: return 57;
        case 256 * 100000 + 8192: return 115;
        case 256 * 100000 + 16384: return 232;
        case 256 * 100000 + 32768: return 465;
        case 256 * 100000 + 65536: return 930;
    }
    return 0;
}

/* keygeneration.cu:684-728 steps_to_galois_elt */
int o_steps_to_galois_elt(int steps, int n, int group_order)
{
    int m = n * 2;
    if (steps == 0) return m - 1;
    int sign = steps < 0;
    int pos = abs(steps);
    if (pos >= (n >> 1)) return 0;
    steps = sign ? (n >> 1) - pos : pos;
    int gen = group_order, elt = 1;
    while (steps > 0) {
        elt = elt * gen;
        elt = elt & (m - 1);
        steps--;
    }
    return elt;
}

u64 o_splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ULL;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void o_fill_poly(u64* out, u64 seed, int limb, u64 n, u64 q)
{
    for (u64 i = 0; i < n; i++)
        out[i] = o_splitmix64(seed + (((u64) limb) << 32) + i) % q;
}

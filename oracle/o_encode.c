/*
 * o_encode.c -- CPU restatement of the CKKS encoder / decoder for real vectors
 * (SURVEY.md 8f next-2).  TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h).
 *
 * Follows src/lib/host/ckks/encoder.cu:21-160, 449-513 and src/lib/kernel/encoding.cu:143-392.
 * The special FFT itself lives in the unvendored thirdparty/GPU-FFT (.gitmodules); its
 * algorithm is the one the root tables of encoder.cu:40-90 are built for -- HEAAN's
 * fftSpecial / fftSpecialInv over the rotation group 5^j -- restated here from that
 * definition.  PARITY UNPINNED against the reference (floating-point operation order inside
 * GPU-FFT is unknown); pinned by what encoding means (tests/test_oracle_encode.py: decode
 * inverts encode, polynomial product = slot-wise product, X -> X^5 rotates the slots).
 */
#include "hegpu_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { double re, im; } cx;

static void tables(const octx_t* c, cx* fwd, cx* inv, int* rev, int* log_slots_out)
{
    const u64 slots = c->n >> 1, M = 2 * c->n;
    int log_slots = 0;
    while ((1ull << log_slots) < slots) log_slots++;
    const double special_root = 2.0 * M_PI / (double) M;
    u64* rot = (u64*) malloc(sizeof(u64) * slots);
    rot[0] = 1;
    for (u64 i = 1; i < slots; i++) rot[i] = (5 * rot[i - 1]) % M;
    memset(fwd, 0, sizeof(cx) * slots);
    memset(inv, 0, sizeof(cx) * slots);
    for (int logm = 1; logm <= log_slots; logm++) {
        const u64 idx_mod = 1ull << (logm + 2), gap = M / idx_mod, offset = 1ull << (logm - 1);
        for (u64 i = 0; i < offset; i++) {
            /* separate cos() and sin() calls (volatile: no merging into sincos) */
            volatile double ang_c = (double) ((rot[i] % idx_mod) * gap) * special_root;
            volatile double ang_s = ang_c;
            const double cr = cos(ang_c), sr = sin(ang_s);
            fwd[offset + i].re = cr; fwd[offset + i].im = sr;
            inv[offset + i].re = cr; inv[offset + i].im = -sr;
        }
    }
    for (u64 i = 0; i < slots; i++) {
        u64 r = 0;
        for (int b = 0; b < log_slots; b++) r |= ((i >> b) & 1) << (log_slots - 1 - b);
        rev[i] = (int) r;
    }
    free(rot);
    *log_slots_out = log_slots;
}

/* fftSpecialInv without its bit reversal (the conversion kernel reads through reverse_order) */
static void special_ifft(cx* v, const cx* roots, u64 slots, double fix)
{
    for (u64 lenh = slots >> 1; lenh >= 1; lenh >>= 1) {
        for (u64 i = 0; i < slots; i += 2 * lenh)
            for (u64 j = 0; j < lenh; j++) {
                cx a = v[i + j], b = v[i + j + lenh], w = roots[lenh + j];
                cx d = { a.re - b.re, a.im - b.im };
                v[i + j].re = a.re + b.re; v[i + j].im = a.im + b.im;
                v[i + j + lenh].re = d.re * w.re - d.im * w.im;
                v[i + j + lenh].im = d.re * w.im + d.im * w.re;
            }
        if (lenh == 1) break;
    }
    for (u64 i = 0; i < slots; i++) { v[i].re = v[i].re * fix; v[i].im = v[i].im * fix; }
}

/* fftSpecial on bit-reversed input */
static void special_fft(cx* v, const cx* roots, u64 slots)
{
    for (u64 lenh = 1; lenh < slots; lenh <<= 1)
        for (u64 i = 0; i < slots; i += 2 * lenh)
            for (u64 j = 0; j < lenh; j++) {
                cx a = v[i + j], b = v[i + j + lenh], w = roots[lenh + j];
                cx bw = { b.re * w.re - b.im * w.im, b.re * w.im + b.im * w.re };
                v[i + j].re = a.re + bw.re; v[i + j].im = a.im + bw.im;
                v[i + j + lenh].re = a.re - bw.re; v[i + j + lenh].im = a.im - bw.im;
            }
}

/* encode_kernel_ckks_conversion, one component (encoding.cu:176-201) */
static void store_rns(const octx_t* c, u64* plain, u64 at, double value, int limbs)
{
    double v = round(value);
    int neg = signbit(v) != 0;
    v = fabs(v);
    const double two64 = 18446744073709551616.0;
    u64 lo = (u64) fmod(v, two64), hi = (u64) (v / two64);
    u128 wide = ((u128) hi << 64) | lo;
    for (int i = 0; i < limbs; i++) {
        u64 r = (u64) (wide % c->mod[i].value);
        plain[at + ((u64) i << c->n_power)] = neg ? o_sub(c->mod[i].value, r, &c->mod[i]) : r;
    }
}

/* HEEncoder<CKKS>::encode_ckks / encode_ckks_coeff (ckks/encoder.cu:100-446); plain [Q][N], NTT domain.
 *   mode 0: vector<double> into the slots (:100-160)
 *   mode 1: vector<Complex64> into the slots, message = (re, im) pairs (:294-352)
 *   mode 2: vector<double> as polynomial coefficients, at most N of them (:222-261,
 *           encode_kernel_coeff_ckks_conversion encoding.cu:106-137)
 *   mode 3: one double (or int64 cast to double) in every slot = the constant polynomial,
 *           written straight into the NTT domain (:412-446, encode_kernel_double_ckks_conversion
 *           encoding.cu:43-77) */
void o_ckks_encode_ex(const octx_t* c, int mode, const double* message, int message_size, double scale, u64* plain)
{
    const u64 slots = c->n >> 1;
    if (mode == 3) {
        for (u64 idx = 0; idx < c->n; idx++) store_rns(c, plain, idx, message[0] * scale, c->Q_size);
        return;
    }
    if (mode == 2) {
        for (u64 idx = 0; idx < c->n; idx++)
            store_rns(c, plain, idx, ((int) idx < message_size ? message[idx] : 0.0) * scale, c->Q_size);
        o_gpu_ntt(plain, plain, c->ntt_table, c->mod, c->n_power, c->Q_size, c->Q_size);
        return;
    }
    cx* fwd = (cx*) malloc(sizeof(cx) * slots);
    cx* inv = (cx*) malloc(sizeof(cx) * slots);
    int* rev = (int*) malloc(sizeof(int) * slots);
    cx* v = (cx*) malloc(sizeof(cx) * slots);
    int log_slots;
    tables(c, fwd, inv, rev, &log_slots);
    for (u64 i = 0; i < slots; i++) {
        const int in = (int) i < message_size;
        v[i].re = in ? message[mode == 1 ? 2 * i : i] : 0.0;
        v[i].im = (in && mode == 1) ? message[2 * i + 1] : 0.0;
    }
    special_ifft(v, inv, slots, scale / (double) slots);
    for (u64 idx = 0; idx < slots; idx++) {
        cx z = v[rev[idx]];
        store_rns(c, plain, idx, z.re, c->Q_size);
        store_rns(c, plain, idx + slots, z.im, c->Q_size);
    }
    o_gpu_ntt(plain, plain, c->ntt_table, c->mod, c->n_power, c->Q_size, c->Q_size);
    free(fwd); free(inv); free(rev); free(v);
}
void o_ckks_encode(const octx_t* c, const double* message, int message_size, double scale, u64* plain)
{
    o_ckks_encode_ex(c, 0, message, message_size, scale, plain);
}

/* one coefficient of encode_kernel_compose (encoding.cu:246-312): CRT composition with
 * little-endian 64-bit words, then the word-wise conversion to double */
static double compose_one(const octx_t* c, const u64* coeff, u64 at, int l, const u64* Mi, const u64* Mi_inv,
                          const u64* M, const u64* half, double inv_scale)
{
    u64 acc[O_MAX_MOD];
    memset(acc, 0, sizeof(u64) * l);
    for (int i = 0; i < l; i++) {
        u64 t = o_mult(coeff[at + ((u64) i << c->n_power)], Mi_inv[i], &c->mod[i]);
        u128 carry = 0;
        for (int k = 0; k < l; k++) {
            u128 v = (u128) Mi[(u64) i * l + k] * t + acc[k] + carry;
            acc[k] = (u64) v;
            carry = v >> 64;
        }
        int geq = 1;
        for (int k = l - 1; k >= 0; k--)
            if (acc[k] != M[k]) { geq = acc[k] > M[k]; break; }
        if (geq) {
            u64 borrow = 0;
            for (int k = 0; k < l; k++) {
                u128 d = (u128) acc[k] - M[k] - borrow;
                acc[k] = (u64) d;
                borrow = (u64) (d >> 64) & 1;
            }
        }
    }
    int neg = 1;
    for (int k = l - 1; k >= 0; k--)
        if (acc[k] != half[k]) { neg = acc[k] > half[k]; break; }
    const double two64 = 18446744073709551616.0;
    double result = 0.0, w = inv_scale;
    for (int j = 0; j < l; j++, w *= two64) {
        if (neg) {
            if (acc[j] > M[j]) { u64 d = acc[j] - M[j]; result += d ? (double) d * w : 0.0; }
            else { u64 d = M[j] - acc[j]; result -= d ? (double) d * w : 0.0; }
        } else {
            result += acc[j] ? (double) acc[j] * w : 0.0;
        }
    }
    return result;
}

/* HEEncoder<CKKS>::decode_ckks / decode_ckks_coeff (ckks/encoder.cu:449-690); plain [Q - depth][N]
 *   mode 0: N/2 real parts of the slots; mode 1: N/2 complex slots as (re, im) pairs;
 *   mode 2: the N polynomial coefficients (decode_kernel_coeff_ckks_compose encoding.cu:387-464) */
void o_ckks_decode_ex(const octx_t* c, int mode, const u64* plain, int depth, double scale, double* message);
void o_ckks_decode(const octx_t* c, const u64* plain, int depth, double scale, double* message)
{
    o_ckks_decode_ex(c, 0, plain, depth, scale, message);
}
void o_ckks_decode_ex(const octx_t* c, int mode, const u64* plain, int depth, double scale, double* message)
{
    const int l = c->Q_size - depth, np = c->n_power;
    const u64 slots = c->n >> 1;
    u64* coeff = (u64*) malloc(sizeof(u64) * ((u64) l << np));
    o_gpu_intt(plain, coeff, c->intt_table, c->mod, c->n_inv, np, l, l);
    /* level tables (ckks/context.cu:370-421 via util.cu:772-888) */
    u64* Mi = (u64*) calloc((size_t) l * l, sizeof(u64));
    u64 Mi_inv[O_MAX_MOD], M[O_MAX_MOD + 1], half[O_MAX_MOD + 1];
    for (int i = 0; i <= l; i++) {   /* i == l: the full product */
        u64 big[O_MAX_MOD + 1];
        memset(big, 0, sizeof(big));
        big[0] = 1;
        u64 m = 1;
        for (int j = 0; j < l; j++) {
            if (j == i) continue;
            u128 carry = 0;
            for (int k = 0; k <= l; k++) {
                u128 v = (u128) big[k] * c->mod[j].value + carry;
                big[k] = (u64) v;
                carry = v >> 64;
            }
            if (i < l) m = o_mult(m, c->mod[j].value % c->mod[i].value, &c->mod[i]);
        }
        if (i < l) {
            memcpy(Mi + (size_t) i * l, big, sizeof(u64) * l);
            Mi_inv[i] = o_modinv(m, &c->mod[i]);
        } else {
            memcpy(M, big, sizeof(u64) * (l + 1));
        }
    }
    { /* (M + 1) >> 1 */
        u64 t[O_MAX_MOD + 2];
        memcpy(t, M, sizeof(u64) * (l + 1));
        t[l + 1] = 0;
        for (int k = 0; k <= l; k++) if (++t[k]) break;
        for (int k = 0; k <= l; k++) half[k] = (t[k] >> 1) | (t[k + 1] << 63);
    }
    const double inv_scale = 1.0 / scale;
    if (mode == 2) {
        for (u64 idx = 0; idx < c->n; idx++) message[idx] = compose_one(c, coeff, idx, l, Mi, Mi_inv, M, half, inv_scale);
        free(coeff); free(Mi);
        return;
    }
    cx* fwd = (cx*) malloc(sizeof(cx) * slots);
    cx* inv = (cx*) malloc(sizeof(cx) * slots);
    int* rev = (int*) malloc(sizeof(int) * slots);
    cx* v = (cx*) malloc(sizeof(cx) * slots);
    int log_slots;
    tables(c, fwd, inv, rev, &log_slots);
    for (u64 idx = 0; idx < slots; idx++) {
        cx z;
        z.re = compose_one(c, coeff, idx, l, Mi, Mi_inv, M, half, inv_scale);
        z.im = compose_one(c, coeff, idx + slots, l, Mi, Mi_inv, M, half, inv_scale);
        v[rev[idx]] = z;
    }
    special_fft(v, fwd, slots);
    for (u64 i = 0; i < slots; i++) {
        if (mode == 1) { message[2 * i] = v[i].re; message[2 * i + 1] = v[i].im; }
        else message[i] = v[i].re;
    }
    free(coeff); free(Mi); free(fwd); free(inv); free(rev); free(v);
}

/* addition_constant_plain_ckks_poly / substraction_constant_plain_ckks_poly (addition.cu:219-300) and
 * cipher_constant_plain_multiplication_kernel (multiplication.cu:333-372); op 0 add, 1 sub, 2 multiply.
 * ct, out [parts][limbs][N] */
void o_ckks_constant_op(const octx_t* c, int op, const u64* ct, double value, u64* out, int limbs, int parts)
{
    double v = round(value);
    const int neg = signbit(v) != 0;
    v = fabs(v);
    const double two64 = 18446744073709551616.0;
    const u128 wide = ((u128) (u64) (v / two64) << 64) | (u64) fmod(v, two64);
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++) {
            u64 pt = (u64) (wide % c->mod[y].value);
            if (neg) pt = o_sub(c->mod[y].value, pt, &c->mod[y]);
            for (u64 i = 0; i < c->n; i++) {
                const u64 loc = i + ((u64) y << c->n_power) + (((u64) limbs * z) << c->n_power);
                if (op == 2) out[loc] = o_mult(ct[loc], pt, &c->mod[y]);
                else if (z != 0) out[loc] = ct[loc];
                else out[loc] = op == 0 ? o_add(ct[loc], pt, &c->mod[y]) : o_sub(ct[loc], pt, &c->mod[y]);
            }
        }
}

/* add_constant_plain_ckks_v2 / multiply_const_plain_ckks_v2 (ckks/operator.cu:567-724) with
 * cipher_add_by_gaussian_integer_kernel / cipher_mult_by_gaussian_integer_kernel (multiplication.cu:497-570).
 * re / im: the already scaled doubles; the reference rounds them, converts to NTL::ZZ and takes the
 * non-negative residue modulo every q_j (operator.cu:586-617) -- restated with 128-bit integers. */
static u64 zz_residue(double value, u64 q)
{
    double v = round(value);
    const int neg = signbit(v) != 0;
    v = fabs(v);
    const double two64 = 18446744073709551616.0;
    u64 r;
    if (v < two64 * two64) {
        const u128 wide = ((u128) (u64) (v / two64) << 64) | (u64) fmod(v, two64);
        r = (u64) (wide % q);
    } else { /* NTL takes any magnitude: v = mant * 2^(e - 53) exactly */
        int e;
        const double fr = frexp(v, &e);
        const u64 mant = (u64) ldexp(fr, 53);
        u128 acc = 1 % q, p = 2 % q;
        for (int sh = e - 53; sh; sh >>= 1) {
            if (sh & 1) acc = acc * p % q;
            p = p * p % q;
        }
        r = (u64) ((u128) (mant % q) * acc % q);
    }
    if (neg && r) r = q - r; /* real_mod < 0 -> += q */
    return r;
}
void o_ckks_gaussian_integer_op(const octx_t* c, int op, const u64* ct, double re, double im, u64* out, int limbs,
                                int parts)
{
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++) {
            const omod_t* m = &c->mod[y];
            const u64 psi = c->ntt_table[1 + ((u64) y << c->n_power)];
            const u64 c_real = zz_residue(re, m->value), c_imag = zz_residue(im, m->value);
            const u64 const_imag = o_mult(c_imag, psi, m);
            for (u64 i = 0; i < c->n; i++) {
                const u64 loc = i + ((u64) y << c->n_power) + (((u64) limbs * z) << c->n_power);
                const u64 k = (i < (c->n >> 1)) ? o_add(c_real, const_imag, m) : o_sub(c_real, const_imag, m);
                if (op == 0) out[loc] = (z == 0) ? o_add(ct[loc], k, m) : ct[loc];
                else out[loc] = o_mult(ct[loc], k, m);
            }
        }
}

/* cipher_mult_by_i_kernel / cipher_div_by_i_kernel (multiplication.cu:441-495) */
void o_ckks_mult_i(const octx_t* c, const u64* ct, u64* out, int limbs, int parts, int divide)
{
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++) {
            const u64 psi = c->ntt_table[1 + ((u64) y << c->n_power)];
            const u64 neg_psi = o_sub(0, psi, &c->mod[y]);
            for (u64 i = 0; i < c->n; i++) {
                const u64 loc = i + ((u64) y << c->n_power) + (((u64) limbs * z) << c->n_power);
                const int first = i < (c->n >> 1);
                const u64 w = divide ? (first ? neg_psi : psi) : (first ? psi : neg_psi);
                out[loc] = o_mult(ct[loc], w, &c->mod[y]);
            }
        }
}

// test_api.cpp -- the C++ class layer (include/heongpu/heongpu.hpp) used the way
// the reference's tests use its API (test/test_ckks_relinearization.cpp:36-111,
// test_bfv_rotation_method_1.cpp:30-86), checked bit-for-bit against the CPU
// oracle (test infrastructure).  Runs on the GPU box (pytest -m gpu wrapper).
#define HEONGPU_WITH_ZLIB 1
#include <heongpu/heongpu.hpp>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <cstdlib>
#include <cstring>
extern "C" {
#include "hegpu_oracle.h"
}

using namespace heongpu;
typedef std::vector<Data64> Vec;

static int failures = 0;
#define EXPECT(c, what)                                   \
    do {                                                  \
        if (!(c)) { printf("FAIL: %s\n", what); failures++; } \
        else printf("ok:   %s\n", what);                  \
    } while (0)

static Vec synth_ct(const Vec& primes, int limbs, int parts, size_t n, u64 seed)
{
    Vec out((size_t) parts * limbs * n);
    for (int p = 0; p < parts; p++)
        for (int j = 0; j < limbs; j++) o_fill_poly((u64*) &out[((size_t) p * limbs + j) * n], seed * 1000 + p, j, n, primes[j]);
    return out;
}
static Vec synth_key(const Vec& primes, int digits, int Qp, size_t n, u64 seed)
{
    Vec out((size_t) digits * 2 * Qp * n);
    for (int i = 0; i < digits; i++)
        for (int c = 0; c < 2; c++)
            for (int j = 0; j < Qp; j++)
                o_fill_poly((u64*) &out[(((size_t) i * 2 + c) * Qp + j) * n], seed * 100000 + i * 2 + c, j, n, primes[j]);
    return out;
}
static bool same(const Vec& a, const u64* b, size_t cnt) { return a.size() >= cnt && !memcmp(a.data(), b, cnt * 8); }

template <typename F> static bool throws_invalid(F f)
{
    try { f(); } catch (const std::invalid_argument&) { return true; } catch (...) { return false; }
    return false;
}
template <typename F> static bool throws_logic(F f)
{
    try { f(); } catch (const std::logic_error&) { return true; } catch (...) { return false; }
    return false;
}

static void ckks()
{
    constexpr auto S = Scheme::CKKS;
    const size_t n = 8192;
    HEContext<S> ctx = GenHEContext<S>(sec_level_type::none);
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_bit_sizes({40, 35, 35, 35}, {40});
    ctx->generate();
    const int Q = ctx->get_ciphertext_modulus_count(), Qp = ctx->get_key_modulus_count();
    Vec primes = ctx->get_key_modulus();
    octx_t* o = o_ctx_create(O_CKKS, 13, (const u64*) primes.data(), Q, 1, 0);

    Vec h1 = synth_ct(primes, Q, 2, n, 1), h2 = synth_ct(primes, Q, 2, n, 2), hk = synth_key(primes, Q, Qp, n, 3);
    Ciphertext<S> c1(ctx), c2(ctx), c3(ctx);
    c1.load(h1, 2, 0, std::pow(2.0, 35));
    c2.load(h2, 2, 0, std::pow(2.0, 35));
    Relinkey<S> rk(ctx);
    rk.load(hk);
    HEArithmeticOperator<S> op(ctx);

    op.multiply(c1, c2, c3);
    EXPECT(c3.size() == 3 && c3.relinearization_required() && c3.rescale_required(), "ckks multiply metadata");
    EXPECT(throws_invalid([&] { op.multiply(c3, c1, c3); }), "multiply on a non-relinearized ciphertext throws");
    op.relinearize_inplace(c3, rk);
    op.rescale_inplace(c3);
    EXPECT(c3.depth() == 1 && c3.size() == 2, "ckks depth after rescale");
    Vec got;
    c3.get_data(got);
    Vec w(3 * Q * n);
    o_ckks_multiply(o, (const u64*) h1.data(), (const u64*) h2.data(), (u64*) w.data(), 0);
    o_ckks_relinearize(o, (u64*) w.data(), (const u64*) hk.data(), 0);
    o_ckks_rescale(o, (u64*) w.data(), 0);
    EXPECT(same(got, (const u64*) w.data(), 2 * (Q - 1) * n), "ckks multiply+relinearize+rescale == oracle");
    EXPECT(std::fabs(c3.scale() - std::pow(2.0, 70) / (double) primes[Q - 1]) < 1.0, "ckks scale bookkeeping");

    // rotations: direct key (shift 1) and power-of-two chain (shift 3 = 2 + 1)
    Galoiskey<S> gk(ctx, std::vector<int>{1, 2});
    Vec k1 = synth_key(primes, Q, Qp, n, 7), k2 = synth_key(primes, Q, Qp, n, 8);
    gk.load(gk.galois_elt[1], k1);
    gk.load(gk.galois_elt[2], k2);
    Ciphertext<S> r1(ctx), r3(ctx);
    op.rotate_rows(c1, r1, gk, 1);
    r1.get_data(got);
    Vec wr(2 * Q * n), wr2(2 * Q * n);
    o_ckks_apply_galois(o, (const u64*) h1.data(), (u64*) wr.data(), (const u64*) k1.data(), gk.galois_elt[1], 0);
    EXPECT(same(got, (const u64*) wr.data(), 2 * Q * n), "ckks rotate_rows(1) == oracle");
    op.rotate_rows(c1, r3, gk, 3);
    r3.get_data(got);
    o_ckks_apply_galois(o, (const u64*) h1.data(), (u64*) wr.data(), (const u64*) k2.data(), gk.galois_elt[2], 0);
    o_ckks_apply_galois(o, (const u64*) wr.data(), (u64*) wr2.data(), (const u64*) k1.data(), gk.galois_elt[1], 0);
    EXPECT(same(got, (const u64*) wr2.data(), 2 * Q * n), "ckks rotate_rows(3) via 2+1 chain == oracle");
    EXPECT(throws_logic([&] { op.rotate_rows(c1, r3, gk, 4); }), "missing Galois key throws logic_error");

    // add / sub on a non-default stream
    hipStream_t st;
    detail::hip(hipStreamCreate(&st));
    ExecutionOptions opt;
    opt.set_stream(st);
    {
        Ciphertext<S> s(ctx, opt); // must not outlive its stream
        op.add(c1, c2, s, opt);
        s.get_data(got, st);
        Vec wa(2 * Q * n);
        o_addition((const u64*) h1.data(), (const u64*) h2.data(), (u64*) wa.data(), o->mod, 13, Q, 2);
        EXPECT(same(got, (const u64*) wa.data(), 2 * Q * n), "ckks add on a user stream == oracle");
    }
    detail::hip(hipStreamSynchronize(st));
    detail::hip(hipStreamDestroy(st));
    o_ctx_free(o);
}

static void bfv()
{
    constexpr auto S = Scheme::BFV;
    const size_t n = 4096;
    HEContext<S> ctx = GenHEContext<S>();
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_default_values(1);
    ctx->set_plain_modulus(1032193);
    ctx->generate();
    const int Q = ctx->Q_size, Qp = ctx->Q_prime_size;
    Vec primes = ctx->get_key_modulus();
    octx_t* o = o_ctx_create(O_BFV, 12, (const u64*) primes.data(), Q, 1, 1032193);
    Vec h1 = synth_ct(primes, Q, 2, n, 1), h2 = synth_ct(primes, Q, 2, n, 2), hk = synth_key(primes, Q, Qp, n, 3);
    Ciphertext<S> c1(ctx), c2(ctx);
    c1.load(h1, 2, 0);
    c2.load(h2, 2, 0);
    Relinkey<S> rk(ctx);
    rk.load(hk);
    HEArithmeticOperator<S> op(ctx);
    op.multiply_inplace(c1, c2);
    op.relinearize_inplace(c1, rk);
    Vec got;
    c1.get_data(got);
    Vec w(3 * Q * n);
    o_bfv_multiply(o, (const u64*) h1.data(), (const u64*) h2.data(), (u64*) w.data());
    o_bfv_relinearize(o, (u64*) w.data(), (const u64*) hk.data());
    EXPECT(same(got, (const u64*) w.data(), 2 * Q * n), "bfv multiply_inplace+relinearize == oracle");
    Galoiskey<S> gk(ctx, std::vector<int>{1});
    gk.load(gk.galois_elt[1], hk);
    Ciphertext<S> r(ctx);
    op.rotate_rows(c2, r, gk, 1);
    r.get_data(got);
    Vec wr(2 * Q * n);
    o_bfv_apply_galois(o, (const u64*) h2.data(), (u64*) wr.data(), (const u64*) hk.data(), gk.galois_elt[1]);
    EXPECT(same(got, (const u64*) wr.data(), 2 * Q * n), "bfv rotate_rows(1) == oracle");
    o_ctx_free(o);
}


// keygen -> encrypt -> multiply -> relinearize -> rotate -> decrypt through the class layer, the
// way the reference's tests do it (test/test_ckks_relinearization.cpp:36-111); messages are encoded
// here as scaled integer polynomials (the encoders are SURVEY.md 8f next-2).
static void ckks_pipeline()
{
    constexpr auto S = Scheme::CKKS;
    const size_t n = 4096;
    const int n_power = 12;
    HEContext<S> ctx = GenHEContext<S>(sec_level_type::none);
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_bit_sizes({40, 35, 35}, {40});
    ctx->generate();
    const int Q = ctx->get_ciphertext_modulus_count();
    Vec primes = ctx->get_key_modulus();
    octx_t* oc = o_ctx_create(O_CKKS, n_power, (const u64*) primes.data(), Q, 1, 0);

    HEKeyGenerator<S> keygen(ctx, 2026);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk);
    Galoiskey<S> gk(ctx, std::vector<int>{1});
    keygen.generate_galois_key(gk, sk);
    EXPECT(throws_logic([&] { keygen.generate_secret_key(sk); }), "Secretkey is already generated");

    const long long scale = 1LL << 16;
    std::vector<long long> m1(n, 0), m2(n, 0);
    m1[0] = 3; m1[1] = 2; m1[100] = -4;
    m2[0] = 5; m2[2] = -1;
    auto encode = [&](const std::vector<long long>& m) {
        Vec r((size_t) Q * n);
        for (int j = 0; j < Q; j++) {
            for (size_t i = 0; i < n; i++) {
                long long v = m[i] * scale;
                r[(size_t) j * n + i] = v < 0 ? primes[j] - (Data64) (-v) : (Data64) v;
            }
            o_ntt_limb((u64*) &r[(size_t) j * n], oc->ntt_table + (size_t) j * n, &oc->mod[j], n_power);
        }
        return r;
    };
    auto decode0 = [&](Plaintext<S>& pt) { // centred coefficients of limb 0
        Vec h;
        pt.get_data(h);
        o_intt_limb((u64*) h.data(), oc->intt_table, &oc->mod[0], oc->n_inv[0], n_power);
        std::vector<long long> out(n);
        for (size_t i = 0; i < n; i++)
            out[i] = h[i] > primes[0] / 2 ? -(long long) (primes[0] - h[i]) : (long long) h[i];
        return out;
    };
    auto maxerr = [&](const std::vector<long long>& got, const std::vector<long long>& want, long long f) {
        long long e = 0;
        for (size_t i = 0; i < n; i++) e = std::max(e, std::llabs(got[i] - want[i] * f));
        return e;
    };

    HEEncryptor<S> enc(ctx, pk, 7);
    HEDecryptor<S> dec(ctx, sk);
    HEArithmeticOperator<S> op(ctx);
    Plaintext<S> p1(ctx), p2(ctx), out(ctx);
    p1.load(encode(m1), 0, (double) scale);
    p2.load(encode(m2), 0, (double) scale);
    Ciphertext<S> c1(ctx), c2(ctx), c3(ctx), rot(ctx);
    enc.encrypt(c1, p1);
    enc.encrypt(c2, p2);
    dec.decrypt(out, c1);
    EXPECT(maxerr(decode0(out), m1, scale) < (1 << 12), "decrypt(encrypt(m)) = m (+ fresh noise)");

    op.multiply(c1, c2, c3);
    EXPECT(throws_invalid([&] { dec.decrypt(out, c3); }), "3-part ciphertext must be relinearized before decryption");
    op.relinearize_inplace(c3, rk);
    dec.decrypt(out, c3);
    std::vector<long long> prod(n, 0);
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            if (!m1[i] || !m2[j]) continue;
            size_t k = i + j;
            if (k >= n) prod[k - n] -= m1[i] * m2[j]; else prod[k] += m1[i] * m2[j];
        }
    EXPECT(maxerr(decode0(out), prod, scale * scale) < scale * scale / 8, "decrypt(relinearize(c1 * c2)) = m1 * m2");

    op.rotate_rows(c1, rot, gk, 1);
    dec.decrypt(out, rot);
    const int g = gk.galois_elt[1];
    std::vector<long long> want(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (!m1[i]) continue;
        const size_t r = (i * (size_t) g) % (2 * n);
        if (r >= n) want[r - n] = -m1[i]; else want[r] = m1[i];
    }
    EXPECT(maxerr(decode0(out), want, scale) < (1 << 14), "decrypt(rotate(c1)) = sigma_g(m1)");

    // hoisted rotations (host/ckks/operator.cuh:2133-2196): entry i = the input rotated by bsgs_shift[i]; shifts
    // with their own key share one decomposition, shift 3 has none and goes through the 2 + 1 chain
    {
        Galoiskey<S> gk3(ctx, std::vector<int>{1, 2, -1});
        keygen.generate_galois_key(gk3, sk);
        std::vector<int> shifts{0, 1, 2, -1, 3};
        DeviceVector<Data64> many = op.fast_single_hoisting_rotation_ckks(c1, shifts, (int) shifts.size(), gk3);
        const size_t words = 2 * (size_t) Q * n;
        bool ok = many.size() == words * shifts.size();
        for (size_t i = 0; ok && i < shifts.size(); i++) {
            Ciphertext<S> single(ctx);
            if (shifts[i] == 0) single = c1;
            else op.rotate_rows(c1, single, gk3, shifts[i]);
            Vec a(words), b;
            (void) hipMemcpy(a.data(), many.data() + i * words, words * 8, hipMemcpyDeviceToHost);
            single.get_data(b);
            ok = ok && b.size() >= words && !memcmp(a.data(), b.data(), words * 8);
        }
        EXPECT(ok, "fast_single_hoisting_rotation_ckks: every entry equals the separate rotate_rows result, bit for bit");
    }
    o_ctx_free(oc);
}

// BFV through the class layer: keygen -> encrypt -> multiply -> relinearize -> decrypt, exact
static void bfv_pipeline()
{
    constexpr auto S = Scheme::BFV;
    const size_t n = 4096;
    const Data64 t = 1032193;
    HEContext<S> ctx = GenHEContext<S>();
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_default_values(1);
    ctx->set_plain_modulus(t);
    ctx->generate();
    HEKeyGenerator<S> keygen(ctx, 11);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk);
    HEEncryptor<S> enc(ctx, pk, 12);
    HEDecryptor<S> dec(ctx, sk);
    HEArithmeticOperator<S> op(ctx);
    Vec m1(n), m2(n, 0);
    for (size_t i = 0; i < n; i++) m1[i] = (i * 2654435761ull) % t;
    m2[0] = 7; m2[1] = t - 2; // 7 - 2X
    Plaintext<S> p1(ctx), p2(ctx), out(ctx);
    p1.load(m1, 0, 0.0);
    p2.load(m2, 0, 0.0);
    Ciphertext<S> c1(ctx), c2(ctx);
    enc.encrypt(c1, p1);
    enc.encrypt(c2, p2);
    dec.decrypt(out, c1);
    Vec got;
    out.get_data(got);
    EXPECT(got == m1, "bfv decrypt(encrypt(m)) == m");
    op.multiply_inplace(c1, c2);
    op.relinearize_inplace(c1, rk);
    dec.decrypt(out, c1);
    out.get_data(got);
    Vec want(n);
    for (size_t i = 0; i < n; i++) { // (7 - 2X) * m1 mod (X^N + 1, t)
        const Data64 prev = i ? m1[i - 1] : (t - m1[n - 1]) % t;
        want[i] = (7 * m1[i] + (t - 2) * prev) % t;
    }
    EXPECT(got == want, "bfv decrypt(relinearize(c1 * c2)) == m1 * m2 mod (X^N+1, t)");

    // batching: slot-wise product and row rotation (reference example 1_basic_bfv.cpp)
    HEEncoder<S> encoder(ctx);
    Galoiskey<S> gk(ctx, std::vector<int>{1});
    keygen.generate_galois_key(gk, sk);
    std::vector<uint64_t> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = (i * 7 + 1) % t; b[i] = (i * i + 3) % t; }
    Plaintext<S> pa(ctx), pb(ctx), pr(ctx);
    encoder.encode(pa, a);
    encoder.encode(pb, b);
    Ciphertext<S> ca(ctx), cb(ctx), cr(ctx);
    enc.encrypt(ca, pa);
    enc.encrypt(cb, pb);
    op.multiply_inplace(ca, cb);
    op.relinearize_inplace(ca, rk);
    std::vector<uint64_t> slots;
    dec.decrypt(pr, ca);
    encoder.decode(slots, pr);
    bool ok = slots.size() == n;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == (a[i] * b[i]) % t;
    EXPECT(ok, "bfv slots: decode(decrypt(enc(a) * enc(b))) == a .* b");
    enc.encrypt(cb, pb);
    op.rotate_rows(cb, cr, gk, 1);
    dec.decrypt(pr, cr);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n / 2; i++)
        ok = slots[i] == b[(i + 1) % (n / 2)] && slots[n / 2 + i] == b[n / 2 + (i + 1) % (n / 2)];
    EXPECT(ok, "bfv slots: rotate_rows(1) shifts both rows left by one");
    op.rotate_columns(cb, cr, gk);
    dec.decrypt(pr, cr);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n / 2; i++) ok = slots[i] == b[n / 2 + i] && slots[n / 2 + i] == b[i];
    EXPECT(ok, "bfv slots: rotate_columns swaps the two rows");
    enc.encrypt(cb, pb);
    op.add_plain_inplace(cb, pa);
    op.multiply_plain(cb, pa, cb);
    op.sub_plain_inplace(cb, pb);
    dec.decrypt(pr, cb);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == (((a[i] + b[i]) % t) * a[i] % t + t - b[i]) % t;
    EXPECT(ok, "bfv slots: ((b + a) .* a) - b with plaintext operands");
}

// the reference's basic CKKS example flow (example/basic/4_basic_ckks.cpp) through the class layer
static void ckks_encoder_flow()
{
    constexpr auto S = Scheme::CKKS;
    const size_t n = 8192;
    HEContext<S> ctx = GenHEContext<S>(sec_level_type::none);
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_bit_sizes({60, 40, 40, 40}, {60});
    ctx->generate();
    HEKeyGenerator<S> keygen(ctx, 3);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk);
    Galoiskey<S> gk(ctx, std::vector<int>{1});
    keygen.generate_galois_key(gk, sk);
    HEEncoder<S> encoder(ctx);
    HEEncryptor<S> enc(ctx, pk, 4);
    HEDecryptor<S> dec(ctx, sk);
    HEArithmeticOperator<S> op(ctx);
    const int slots = encoder.slot_count();
    const double scale = std::pow(2.0, 40);
    std::vector<double> x(slots), y(slots), got;
    for (int i = 0; i < slots; i++) { x[i] = 0.001 * i - 2.0; y[i] = 3.0 - 0.0005 * i; }
    Plaintext<S> px(ctx), py(ctx), pr(ctx);
    encoder.encode(px, x, scale);
    encoder.encode(py, y, scale);
    Ciphertext<S> cx(ctx), cy(ctx), cr(ctx);
    enc.encrypt(cx, px);
    enc.encrypt(cy, py);
    op.multiply_inplace(cx, cy);
    op.relinearize_inplace(cx, rk);
    op.rescale_inplace(cx);
    dec.decrypt(pr, cx);
    encoder.decode(got, pr);
    double e = 0;
    for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] - x[i] * y[i]));
    EXPECT(e < 1e-5, "ckks: decode(decrypt(rescale(relin(enc(x) * enc(y))))) = x .* y");
    op.rotate_rows(cy, cr, gk, 1);
    dec.decrypt(pr, cr);
    encoder.decode(got, pr);
    e = 0;
    for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] - y[(i + 1) % slots]));
    EXPECT(e < 1e-6, "ckks: rotate_rows(1) shifts the slots left by one");
    // ciphertext (+,-,*) plaintext
    Ciphertext<S> cz(ctx);
    enc.encrypt(cz, px);
    op.add_plain_inplace(cz, py);
    dec.decrypt(pr, cz);
    encoder.decode(got, pr);
    e = 0;
    for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] - (x[i] + y[i])));
    EXPECT(e < 1e-6, "ckks: add_plain_inplace");
    op.sub_plain_inplace(cz, py);
    op.multiply_plain(cz, py, cz);
    op.rescale_inplace(cz);
    dec.decrypt(pr, cz);
    encoder.decode(got, pr);
    e = 0;
    for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] - x[i] * y[i]));
    EXPECT(e < 1e-5, "ckks: sub_plain_inplace, multiply_plain (aliased output), rescale");
    // complex constants in every slot (add_plain_v2 / multiply_plain_v2 / scale_up, operator.cuh:586-926)
    {
        std::vector<Complex64> gc;
        Ciphertext<S> cw(ctx), cv(ctx);
        enc.encrypt(cw, px);
        op.add_plain_v2(cw, Complex64(1.5, -0.25), cv);
        dec.decrypt(pr, cv);
        encoder.decode(gc, pr);
        e = 0;
        for (int i = 0; i < slots; i++) e = std::max(e, std::abs(gc[i] - Complex64(x[i] + 1.5, -0.25)));
        EXPECT(e < 1e-6, "ckks: add_plain_v2 adds the complex constant to every slot");
        op.multiply_plain_v2(cw, Complex64(0.5, 2.0), cv); // fractional: the constant is scaled by the last modulus
        EXPECT(cv.scale() > scale * 1e11, "ckks: multiply_plain_v2 with a fractional constant multiplies the scale by q_l");
        EXPECT(!cv.rescale_required(), "ckks: ... and, like the reference, does not flag the result for rescaling");
        dec.decrypt(pr, cv);
        encoder.decode(gc, pr);
        e = 0;
        for (int i = 0; i < slots; i++) e = std::max(e, std::abs(gc[i] - Complex64(x[i], 0) * Complex64(0.5, 2.0)));
        EXPECT(e < 1e-5, "ckks: multiply_plain_v2 multiplies every slot by the complex constant");
        op.multiply_plain_v2(cw, Complex64(-3.0, 0.0), cv); // integer constant: no extra scale
        EXPECT(cv.scale() == cw.scale(), "ckks: an integer constant leaves the scale unchanged");
        dec.decrypt(pr, cv);
        encoder.decode(got, pr);
        e = 0;
        for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] + 3.0 * x[i]));
        EXPECT(e < 1e-6, "ckks: multiply_plain_v2 by -3");
        op.scale_up(cw, 1024.0, cv);
        EXPECT(cv.scale() == cw.scale() * 1024.0 && cv.level() == cw.level(), "ckks: scale_up multiplies the scale");
        dec.decrypt(pr, cv);
        encoder.decode(got, pr);
        e = 0;
        for (int i = 0; i < slots; i++) e = std::max(e, std::fabs(got[i] - x[i]));
        EXPECT(e < 1e-6, "ckks: scale_up keeps the message");
    }
}

// save / load in the reference's wire format + zlib file framing (util/serializer.h)
static void serializer_round_trip()
{
    constexpr auto S = Scheme::CKKS;
    const size_t n = 4096;
    HEContext<S> ctx = GenHEContext<S>(sec_level_type::none);
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_bit_sizes({40, 35, 35}, {40});
    ctx->generate();
    Vec primes = ctx->get_key_modulus();
    const int Q = 3;
    Ciphertext<S> a(ctx), b(ctx);
    Vec data = synth_ct(primes, Q - 1, 2, n, 5);
    a.load(data, 2, 1, 1099511627776.0);
    std::stringstream ss;
    a.save(ss);
    const std::string bytes = ss.str();
    // header: u8 scheme(2) | int 4096 | int 3 | int 2 | int depth 1 | bool ntt | u8 storage(2) | double | u8 | bool x3 | u32
    const size_t header = 1 + 4 * 4 + 1 + 1 + 8 + 1 + 1 + 1 + 1 + 4;
    EXPECT(bytes.size() == header + data.size() * 8 && (unsigned char) bytes[0] == 2, "ciphertext wire size and scheme tag");
    b.load(ss);
    Vec back;
    b.get_data(back);
    EXPECT(back == data && b.depth() == 1 && b.scale() == 1099511627776.0 && b.size() == 2, "ciphertext save -> load");
    const char* path = "/tmp/hegpu_ct.bin";
    serializer::save_to_file(a, path);
    Ciphertext<S> c(ctx);
    serializer::load_from_file(c, path);
    c.get_data(back);
    EXPECT(back == data, "ciphertext file round trip (u64 size + zlib stream)");
    std::ifstream f(path, std::ios::binary);
    uint64_t sz = 0;
    f.read((char*) &sz, 8);
    f.seekg(0, std::ios::end);
    EXPECT((uint64_t) f.tellg() == 8 + sz, "file = u64 size + payload");
    Ciphertext<Scheme::BFV>* none = nullptr;
    (void) none;
    bool thrown = false;
    try { Ciphertext<S> d(ctx); std::stringstream bad("\x01garbage"); d.load(bad); } catch (const std::runtime_error&) { thrown = true; }
    EXPECT(thrown, "a BFV-tagged binary is rejected by a CKKS ciphertext");
}

// MemoryPool (the class layer's caching allocator, util/memorypool.cuh in the reference): blocks are reused,
// a block freed on one stream and taken on another is ordered by its event, and the cache can be dropped.
__global__ void fill_pattern(unsigned long long* p, size_t n, unsigned long long v)
{
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + i;
}
static void memory_pool()
{
    MemoryPool& pool = MemoryPool::instance();
    hipStream_t s1, s2;
    (void) hipStreamCreate(&s1);
    (void) hipStreamCreate(&s2);
    const size_t n = (size_t) 3 << 20; // 24 MiB
    void* a = pool.allocate(n * 8, s1);
    pool.deallocate(a, n * 8, s1);
    void* b = pool.allocate(n * 8, s1);
    EXPECT(a == b, "a freed block is handed out again for the same size");
    pool.deallocate(b, n * 8, s1);
    // cross-stream reuse: a long kernel on s1 writes the block, it is freed on s1 and taken on s2, whose
    // kernel must run after the first one (the second pattern survives)
    bool ok = true;
    std::vector<unsigned long long> h(n);
    for (int round = 0; round < 8 && ok; round++) {
        unsigned long long* x = (unsigned long long*) pool.allocate(n * 8, s1);
        for (int rep = 0; rep < 20; rep++) fill_pattern<<<(unsigned) ((n + 255) / 256), 256, 0, s1>>>(x, n, 1000 + rep);
        pool.deallocate(x, n * 8, s1);
        unsigned long long* y = (unsigned long long*) pool.allocate(n * 8, s2);
        fill_pattern<<<(unsigned) ((n + 255) / 256), 256, 0, s2>>>(y, n, 7);
        (void) hipMemcpyAsync(h.data(), y, n * 8, hipMemcpyDeviceToHost, s2);
        (void) hipStreamSynchronize(s2);
        ok = ok && (x == y) && h[0] == 7 && h[n - 1] == 7 + n - 1 && h[n / 2] == 7 + n / 2;
        pool.deallocate(y, n * 8, s2);
    }
    EXPECT(ok, "block freed on one stream and reused on another: the later kernel wins");
    // different sizes get different live blocks, nothing overlaps
    std::vector<std::pair<char*, size_t>> live;
    for (size_t sz : {(size_t) 11010048, (size_t) 22020096, (size_t) 33030144, (size_t) 11534336, (size_t) 512, (size_t) 4096})
        live.push_back({(char*) pool.allocate(sz, s1), sz});
    bool disjoint = true;
    for (size_t i = 0; i < live.size(); i++)
        for (size_t j = i + 1; j < live.size(); j++)
            disjoint = disjoint && (live[i].first + live[i].second <= live[j].first || live[j].first + live[j].second <= live[i].first);
    EXPECT(disjoint, "live buffers do not overlap (the ROCm stream-ordered pool's failure, tools/hip_pool_repro.cpp)");
    for (auto& l : live) pool.deallocate(l.first, l.second, s1);
    (void) hipDeviceSynchronize();
    pool.release_cached();
    void* c = pool.allocate(n * 8, s1);
    EXPECT(c != nullptr, "allocation after the cache was released");
    pool.deallocate(c, n * 8, s1);
    (void) hipStreamDestroy(s1);
    (void) hipStreamDestroy(s2);
}

// Every serializable object of the reference (example/basic/13_bfv_serialization.cpp,
// 14_ckks_serialization.cpp): byte layout of the headers (field order and widths of
// */context.cu, secretkey.cu, publickey.cu, evaluationkey.cu, plaintext.cu save()), round trips
// through default-constructed objects, and the loaded objects keep working.
// BFV in the NTT domain (bfv/operator.cuh:884-1110): transform_to_ntt of ciphertext and plaintext,
// pointwise multiply_plain, transform_from_ntt; multiply_power_of_X
static void bfv_ntt_domain_and_shift()
{
    constexpr auto S = Scheme::BFV;
    const int n = 8192, t = 65537;
    HEContext<S> ctx = GenHEContext<S>();
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_default_values(1);
    ctx->set_plain_modulus(t);
    ctx->generate();
    HEKeyGenerator<S> keygen(ctx, 6);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    HEEncoder<S> encoder(ctx);
    HEEncryptor<S> encryptor(ctx, pk, 7);
    HEDecryptor<S> decryptor(ctx, sk);
    HEArithmeticOperator<S> op(ctx, encoder);
    std::vector<uint64_t> m1(n), m2(n), got;
    for (int i = 0; i < n; i++) { m1[i] = (uint64_t) (i * 7 + 1) % t; m2[i] = (uint64_t) (i * 13 + 5) % t; }
    Plaintext<S> p1(ctx), p2(ctx), pr(ctx);
    encoder.encode(p1, m1);
    encoder.encode(p2, m2);
    Ciphertext<S> c1(ctx), direct(ctx), viaNtt(ctx);
    encryptor.encrypt(c1, p1);
    op.multiply_plain(c1, p2, direct);
    Ciphertext<S> cn(ctx);
    op.transform_to_ntt(c1, cn);
    EXPECT(cn.in_ntt_domain() && !c1.in_ntt_domain(), "transform_to_ntt sets the domain flag");
    bool thrown = false;
    try { op.multiply_plain(cn, p2, viaNtt); } catch (const std::logic_error&) { thrown = true; }
    EXPECT(thrown, "NTT ciphertext x coefficient plaintext is refused");
    op.transform_to_ntt_inplace(p2);
    op.multiply_plain(cn, p2, viaNtt);
    op.transform_from_ntt_inplace(viaNtt);
    Vec a, b;
    direct.get_data(a);
    viaNtt.get_data(b);
    EXPECT(a == b, "multiply_plain through the NTT domain == direct multiply_plain, bit for bit");
    decryptor.decrypt(pr, viaNtt);
    encoder.decode(got, pr);
    bool ok = true;
    for (int i = 0; i < n; i++) ok = ok && got[i] == m1[i] * m2[i] % t;
    EXPECT(ok, "and decrypts to the slot-wise product");
    // X^k on an un-batched message: coefficients move up by k with a sign flip on wrap-around
    Plaintext<S> raw(ctx);
    std::vector<Data64> poly(n, 0);
    poly[0] = 5; poly[n - 1] = 9;
    raw.load(poly, 0, 0);
    Ciphertext<S> cx(ctx), sh(ctx);
    encryptor.encrypt(cx, raw);
    op.multiply_power_of_X(cx, sh, 3);
    decryptor.decrypt(pr, sh);
    pr.get_data(a);
    EXPECT(a[3] == 5 && a[2] == (Data64) t - 9 && a[0] == 0, "multiply_power_of_X: 5 + 9 X^(N-1) -> 5 X^3 - 9 X^2");
}

static void serialize_all_objects()
{
    constexpr auto S = Scheme::BFV;
    const int n = 4096;
    HEContext<S> ctx0 = GenHEContext<S>();
    ctx0->set_poly_modulus_degree(n);
    ctx0->set_coeff_modulus_default_values(1);
    ctx0->set_plain_modulus(1032193);
    std::stringstream cs;
    ctx0->save(cs); // before generate(), like the example
    // u8 x3 | int x7 | u32 + 3 x Modulus64(24 B) | u32 + 3 x u64 | u32 (empty) | u32 + 2 int | u32 + 1 int | Modulus64
    EXPECT(cs.str().size() == 3 + 7 * 4 + (4 + 3 * 24) + (4 + 3 * 8) + 4 + (4 + 2 * 4) + (4 + 4) + 24, "BFV context wire size");
    EXPECT((unsigned char) cs.str()[0] == 1 && (unsigned char) cs.str()[1] == 1 && (unsigned char) cs.str()[2] == 1,
           "context tags: bfv, sec128, method I");
    HEContext<S> ctx = GenHEContext<S>();
    ctx->load(cs);
    EXPECT(ctx->context_generated_ && ctx->n == n && ctx->Q_size == 2 && ctx->get_plain_modulus() == 1032193 &&
               ctx->get_key_modulus() == ctx0->get_key_modulus(), "context load -> generated context with the same chain");

    HEKeyGenerator<S> keygen(ctx, 1);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    std::stringstream ss;
    sk.save(ss);
    EXPECT(ss.str().size() == 1 + 4 * 4 + 1 + 1 + 1 + 4 + (size_t) 3 * n * 8, "secret key wire size");
    auto sk2 = serializer::deserialize<Secretkey<S>>(serializer::serialize(sk));
    EXPECT(sk2.secret_key_generated_ && sk2.ring_size() == n && sk2.coeff_modulus_count() == 3, "secret key zlib round trip");
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk2);
    serializer::save_to_file(pk, "/tmp/hegpu_pk.bin");
    auto pk2 = serializer::load_from_file<Publickey<S>>("/tmp/hegpu_pk.bin");
    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk2);
    std::stringstream rs;
    rk.save(rs);
    EXPECT(rs.str().size() == 1 + 1 + 6 * 4 + 1 + 1 + 8 + rk.size() * 8, "relin key wire size");
    Relinkey<S> rk2;
    rk2.load(rs);
    std::vector<int> shifts = {1, 3};
    Galoiskey<S> gk(ctx, shifts);
    keygen.generate_galois_key(gk, sk2);
    std::stringstream gs;
    gk.save(gs);
    // header 1+1+4*4+1+4+1+1 | u32 + 2 pairs | int zero | u64 size | u32 count | 2 x (int + key) | zero key
    EXPECT(gs.str().size() == 25 + 4 + 2 * 8 + 4 + 8 + 4 + 2 * (4 + gk.size() * 8) + gk.size() * 8, "galois key wire size");
    Galoiskey<S> gk2;
    gk2.load(gs);
    EXPECT(gk2.galois_elt == gk.galois_elt && gk2.galois_elt_zero == gk.galois_elt_zero && gk2.device_location_.size() == 3,
           "galois key table after load");

    HEEncoder<S> encoder(ctx);
    HEEncryptor<S> encryptor(ctx, pk2, 2);
    HEDecryptor<S> decryptor(ctx, sk2);
    HEArithmeticOperator<S> op(ctx, encoder);
    std::vector<uint64_t> message(n, 8ULL);
    message[0] = 1; message[1] = 12; message[2] = 23; message[n / 2] = 7;
    Plaintext<S> p1(ctx);
    encoder.encode(p1, message);
    std::stringstream ps;
    p1.save(ps);
    EXPECT(ps.str().size() == 1 + 4 + 1 + 1 + 1 + 4 + (size_t) n * 8, "BFV plaintext wire size");
    Plaintext<S> p2;
    p2.load(ps);
    Ciphertext<S> c1(ctx);
    encryptor.encrypt(c1, p2);
    std::stringstream ts;
    c1.save(ts);
    Ciphertext<S> c2;
    c2.load(ts);
    op.multiply_inplace(c2, c2);
    op.relinearize_inplace(c2, rk2);
    op.rotate_rows_inplace(c2, gk2, 3);
    Plaintext<S> pr(ctx);
    decryptor.decrypt(pr, c2);
    std::vector<uint64_t> result;
    encoder.decode(result, pr);
    // row 0: [1,144,529,64,...] rotated left by 3
    EXPECT(result[0] == 64 && result[n / 2 - 3] == 1 && result[n / 2 - 2] == 144 && result[n / 2 - 1] == 529 &&
               result[n - 3] == 49, "loaded objects: encrypt -> square -> relinearize -> rotate -> decrypt");
    bool thrown = false;
    try { Relinkey<S> again; std::stringstream r2(rs.str()); again.load(r2); again.load(r2); } catch (const std::runtime_error&) { thrown = true; }
    EXPECT(thrown, "loading into a generated key is refused");
}

// Key-switching method II end to end with generated keys (two special primes), and the
// secret-key switch (example/basic/5_switchkey_methods_ckks.cpp).
static void method_II_and_switch_key()
{
    constexpr auto S = Scheme::CKKS;
    const size_t n = 8192;
    HEContext<S> ctx = GenHEContext<S>(sec_level_type::none);
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_bit_sizes({60, 40, 40, 40, 40}, {60, 60});
    ctx->generate();
    EXPECT(ctx->keyswitching_type_ == keyswitching_type::KEYSWITCHING_METHOD_II, "two special primes select method II");
    HEKeyGenerator<S> keygen(ctx, 3);
    Secretkey<S> sk(ctx), sk2(ctx);
    keygen.generate_secret_key(sk);
    keygen.generate_secret_key(sk2);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk);
    EXPECT(rk.size() == (size_t) 2 * 3 * 7 * n, "method II relin key: 3 digits of 2 primes over 7 limbs");
    Galoiskey<S> gk(ctx, std::vector<int>{2});
    keygen.generate_galois_key(gk, sk);
    Switchkey<S> swk(ctx);
    keygen.generate_switch_key(swk, sk2, sk);
    HEEncoder<S> encoder(ctx);
    HEEncryptor<S> encryptor(ctx, pk, 4);
    HEDecryptor<S> dec1(ctx, sk), dec2(ctx, sk2);
    HEArithmeticOperator<S> op(ctx, encoder);
    const double scale = std::pow(2.0, 40);
    std::vector<double> m(n / 2);
    for (size_t i = 0; i < m.size(); i++) m[i] = 0.001 * (double) (i % 1000) - 0.3;
    Plaintext<S> p(ctx);
    encoder.encode(p, m, scale);
    Ciphertext<S> c(ctx);
    encryptor.encrypt(c, p);
    op.multiply_inplace(c, c);
    op.relinearize_inplace(c, rk);
    op.rescale_inplace(c);
    op.rotate_rows_inplace(c, gk, 2);
    Plaintext<S> out(ctx);
    std::vector<double> got;
    dec1.decrypt(out, c);
    encoder.decode(got, out);
    double err = 0;
    for (size_t i = 0; i < m.size(); i++) err = std::max(err, std::fabs(got[i] - m[(i + 2) % m.size()] * m[(i + 2) % m.size()]));
    EXPECT(err < 1e-6, "method II: square -> relinearize -> rescale -> rotate under generated keys");
    { // constants, +-i, conjugation (ckks/operator.cuh:312-390, :812-925, :969-1050, :1353-1420) on a fresh ciphertext
        std::vector<Complex64> z(n / 2), zg;
        for (size_t i = 0; i < z.size(); i++) z[i] = Complex64(0.01 * (double) (i % 50), -0.02 * (double) (i % 31));
        Plaintext<S> pz(ctx);
        encoder.encode(pz, z, scale);
        Ciphertext<S> cz(ctx), t(ctx);
        encryptor.encrypt(cz, pz);
        auto worst = [&](Ciphertext<S>& ct, auto want) {
            Plaintext<S> q(ctx);
            dec1.decrypt(q, ct);
            encoder.decode(zg, q);
            double e = 0;
            for (size_t i = 0; i < z.size(); i++) e = std::max(e, std::abs(zg[i] - want(z[i])));
            return e;
        };
        op.add_plain(cz, 1.5, t);
        EXPECT(worst(t, [](Complex64 v) { return v + 1.5; }) < 1e-6, "ciphertext + constant");
        op.sub_plain(cz, 0.25, t);
        EXPECT(worst(t, [](Complex64 v) { return v - 0.25; }) < 1e-6, "ciphertext - constant");
        op.multiply_plain(cz, -3.0, t, scale);
        op.rescale_inplace(t);
        EXPECT(worst(t, [](Complex64 v) { return v * -3.0; }) < 1e-6, "ciphertext * constant, rescaled");
        op.mult_i(cz, t);
        EXPECT(worst(t, [](Complex64 v) { return v * Complex64(0, 1); }) < 1e-6, "mult_i");
        op.div_i(cz, t);
        EXPECT(worst(t, [](Complex64 v) { return v / Complex64(0, 1); }) < 1e-6, "div_i");
        op.conjugate(cz, t, gk);
        EXPECT(worst(t, [](Complex64 v) { return std::conj(v); }) < 1e-6, "conjugate (method II key for 2N-1)");
    }
    op.keyswitch(c, c, swk);
    Plaintext<S> out2(ctx);
    dec2.decrypt(out2, c);
    encoder.decode(got, out2);
    err = 0;
    for (size_t i = 0; i < m.size(); i++) err = std::max(err, std::fabs(got[i] - m[(i + 2) % m.size()] * m[(i + 2) % m.size()]));
    EXPECT(err < 1e-6, "keyswitch moves the ciphertext to the second secret key");
    std::stringstream ws;
    swk.save(ws);
    EXPECT(ws.str().size() == 1 + 1 + 4 * 4 + 1 + 1 + 8 + swk.size() * 8, "switch key wire size");
}

// TFHE through the class layer (reference test/test_tfhe_gate_boot.cpp:64-86): all gates and MUX
static void tfhe_gates()
{
    constexpr auto S = Scheme::TFHE;
    HEContext<S> ctx = GenHEContext<S>();
    HEKeyGenerator<S> keygen(ctx, 9);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Bootstrappingkey<S> bk(ctx);
    keygen.generate_bootstrapping_key(bk, sk);
    HEEncryptor<S> enc(ctx, sk);
    HEDecryptor<S> dec(ctx, sk);
    HELogicOperator<S> logic(ctx);
    std::vector<bool> x = {0, 0, 1, 1, 0, 0, 1, 1}, y = {0, 1, 0, 1, 0, 1, 0, 1}, c = {0, 0, 0, 0, 1, 1, 1, 1}, got;
    Ciphertext<S> cx(ctx), cy(ctx), cc(ctx), r(ctx);
    enc.encrypt(cx, x);
    enc.encrypt(cy, y);
    enc.encrypt(cc, c);
    dec.decrypt(cx, got);
    EXPECT(got == x, "tfhe decrypt(encrypt(bits)) == bits");
    auto check = [&](const char* name, auto f) {
        dec.decrypt(r, got);
        bool ok = got.size() == x.size();
        for (size_t i = 0; ok && i < x.size(); i++) ok = got[i] == f(x[i], y[i], c[i]);
        EXPECT(ok, name);
    };
    logic.NAND(cx, cy, r, bk); check("tfhe NAND", [](bool a, bool b, bool) { return !(a && b); });
    logic.AND(cx, cy, r, bk);  check("tfhe AND", [](bool a, bool b, bool) { return a && b; });
    logic.NOR(cx, cy, r, bk);  check("tfhe NOR", [](bool a, bool b, bool) { return !(a || b); });
    logic.OR(cx, cy, r, bk);   check("tfhe OR", [](bool a, bool b, bool) { return a || b; });
    logic.XNOR(cx, cy, r, bk); check("tfhe XNOR", [](bool a, bool b, bool) { return a == b; });
    logic.XOR(cx, cy, r, bk);  check("tfhe XOR", [](bool a, bool b, bool) { return a != b; });
    logic.NOT(cx, r);          check("tfhe NOT", [](bool a, bool, bool) { return !a; });
    logic.MUX(cx, cy, cc, r, bk); check("tfhe MUX", [](bool a, bool b, bool s) { return s ? a : b; });
}

// util/storagemanager.cuh: objects parked in pinned host memory are staged on use, go back where they were
// (keep_initial_condition_) or stay on the device, results land where ExecutionOptions::storage_ says; the
// reference's example/basic/8_default_stream_usage.cpp pattern (store_in_host / store_in_device).
static void storage_manager()
{
    constexpr auto S = Scheme::BFV;
    const size_t n = 4096;
    const Data64 t = 1032193;
    HEContext<S> ctx = GenHEContext<S>();
    ctx->set_poly_modulus_degree(n);
    ctx->set_coeff_modulus_default_values(1);
    ctx->set_plain_modulus(t);
    ctx->generate();
    HEKeyGenerator<S> keygen(ctx, 21);
    Secretkey<S> sk(ctx);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk(ctx);
    // a key generated straight into host memory (example/basic/4_switchkey_methods_bfv.cpp:70)
    keygen.generate_relin_key(rk, sk, ExecutionOptions().set_storage_type(storage_type::HOST));
    EXPECT(!rk.is_on_device(), "relin key generated with storage_type::HOST lives in host memory");
    HEEncryptor<S> enc(ctx, pk, 22);
    HEDecryptor<S> dec(ctx, sk);
    HEArithmeticOperator<S> op(ctx);
    HEEncoder<S> encoder(ctx);
    HostVector<uint64_t> a(n), b(n); // pinned
    for (size_t i = 0; i < n; i++) { a[i] = (i * 5 + 2) % t; b[i] = (i * i + 1) % t; }
    unsigned flags = 0;
    EXPECT(hipHostGetFlags(&flags, a.data()) == hipSuccess, "HostVector memory is page-locked (hipHostMalloc)");
    Plaintext<S> pa(ctx), pb(ctx), pr(ctx);
    encoder.encode(pa, a);
    encoder.encode(pb, b);
    Ciphertext<S> ca(ctx), cb(ctx), cs(ctx), cm(ctx);
    enc.encrypt(ca, pa);
    enc.encrypt(cb, pb);
    const size_t words = ca.memory_size();
    ca.store_in_host();
    EXPECT(!ca.is_on_device() && ca.memory_size() == words, "store_in_host parks the ciphertext, its size is unchanged");
    // input on the host, default options: staged for the operation, back on the host afterwards
    op.add(ca, cb, cs);
    EXPECT(!ca.is_on_device() && cs.is_on_device(), "HOST input is staged and returned; result on the device");
    HostVector<uint64_t> slots;
    dec.decrypt(pr, cs);
    encoder.decode(slots, pr);
    bool ok = slots.size() == n;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == (a[i] + b[i]) % t;
    EXPECT(ok, "add with a HOST-stored operand gives a + b");
    // result requested in host memory; relinearization key staged from the host
    op.multiply(ca, cb, cm, ExecutionOptions().set_storage_type(storage_type::HOST));
    EXPECT(!cm.is_on_device(), "ExecutionOptions::storage_ = HOST places the result in host memory");
    op.relinearize_inplace(cm, rk); // in place on a HOST object: staged, modified, copied back
    EXPECT(!cm.is_on_device() && !rk.is_on_device(), "in-place operator on HOST objects leaves them on the host");
    dec.decrypt(pr, cm);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == (a[i] * b[i]) % t;
    EXPECT(ok, "multiply + relinearize through host-parked ciphertext and key gives a .* b");
    // keep_initial_condition_ = false: the staged input stays on the device
    op.add(ca, cb, cs, ExecutionOptions().set_initial_location(false));
    EXPECT(ca.is_on_device(), "set_initial_location(false): the HOST input now lives on the device");
    ca.store_in_host();
    ca.store_in_device();
    dec.decrypt(pr, ca);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == a[i];
    EXPECT(ok && ca.is_on_device(), "store_in_host / store_in_device round trip keeps the residues");
    // copies of a parked object are parked objects with their own data
    ca.store_in_host();
    Ciphertext<S> copy = ca;
    ca.store_in_device();
    EXPECT(!copy.is_on_device(), "copy of a HOST-stored ciphertext is HOST-stored");
    dec.decrypt(pr, copy);
    encoder.decode(slots, pr);
    ok = true;
    for (size_t i = 0; ok && i < n; i++) ok = slots[i] == a[i];
    EXPECT(ok, "the copy decrypts to the same message");

    // ---- rotations through the storage manager: operator-local temporaries (rotate_rows_inplace's copy, the key
    // chain's intermediate ciphertexts) die before the operator's scope does; results assigned by copy / move (a zero
    // shift, the end of a chain) are placed per ExecutionOptions::storage_ like any other result
    Galoiskey<S> gk(ctx, std::vector<int>{1, 2});
    keygen.generate_galois_key(gk, sk);
    auto rotated = [&](const HostVector<uint64_t>& v, int shift) { // BFV batching: two rows of n / 2 slots
        HostVector<uint64_t> r(n);
        const size_t h = n / 2;
        for (size_t i = 0; i < h; i++) { r[i] = v[(i + shift) % h]; r[h + i] = v[h + (i + shift) % h]; }
        return r;
    };
    auto decrypts_to = [&](Ciphertext<S>& c, const HostVector<uint64_t>& want) {
        dec.decrypt(pr, c);
        encoder.decode(slots, pr);
        bool same = slots.size() == n;
        for (size_t i = 0; same && i < n; i++) same = slots[i] == want[i];
        return same;
    };
    Ciphertext<S> r1(ctx);
    enc.encrypt(r1, pa);
    r1.store_in_host();
    op.rotate_rows_inplace(r1, gk, 1); // direct key; HOST input, default options
    EXPECT(decrypts_to(r1, rotated(a, 1)), "rotate_rows_inplace on a HOST-stored ciphertext rotates by one");
    r1.store_in_host();
    op.rotate_rows_inplace(r1, gk, 3, ExecutionOptions().set_storage_type(storage_type::HOST)); // chain 2 + 1
    EXPECT(!r1.is_on_device(), "key-chain rotation with storage_ = HOST leaves the result in host memory");
    EXPECT(decrypts_to(r1, rotated(a, 4)), "chain rotation of a HOST-stored ciphertext (1, then 2 + 1) gives shift 4");
    Ciphertext<S> r2(ctx), r3(ctx);
    enc.encrypt(r2, pb);
    op.rotate_rows(r2, r3, gk, 0, ExecutionOptions().set_storage_type(storage_type::HOST)); // zero shift: out = in
    EXPECT(!r3.is_on_device() && r2.is_on_device(), "zero-shift result obeys storage_ = HOST, the input stays put");
    EXPECT(decrypts_to(r3, b), "zero shift copies the ciphertext");
    op.rotate_rows(r2, r3, gk, 3, ExecutionOptions().set_storage_type(storage_type::HOST));
    EXPECT(!r3.is_on_device() && decrypts_to(r3, rotated(b, 3)), "chain rotation into a HOST-placed result");
    // an operator that throws while a HOST operand is staged: the scope unwinds without copies or a second throw
    r2.store_in_host();
    bool thrown = false;
    try { op.rotate_rows(r2, r3, gk, 4); } catch (const std::logic_error&) { thrown = true; } // no key for 4
    EXPECT(thrown, "missing Galois key throws std::logic_error through the storage scope");
    EXPECT(decrypts_to(r2, b), "the staged operand is intact after the exception");
}

// One process, a thread per device (both threads on device 0 when there is only one): every thread makes its device
// current, generates its own context there, works out of that device's memory pool; the evaluation key is generated once
// and replicated by copy construction (a peer copy between devices).  Both threads must produce the same ciphertext.
static void multi_device()
{
    constexpr auto S = Scheme::CKKS;
    int ndev = 0;
    detail::hip(hipGetDeviceCount(&ndev));
    const int workers = 2;
    std::vector<std::vector<Data64>> results(workers);
    std::vector<int> ctx_dev(workers, -2), buf_dev(workers, -2);
    // keys and inputs made on device 0
    detail::hip(hipSetDevice(0));
    auto make_ctx = [] {
        HEContext<S> c = GenHEContext<S>();
        c->set_poly_modulus_degree(8192);
        c->set_coeff_modulus_bit_sizes({40, 30, 30, 30}, {40});
        c->generate();
        return c;
    };
    HEContext<S> ctx0 = make_ctx();
    HEKeyGenerator<S> keygen(ctx0, 31);
    Secretkey<S> sk(ctx0);
    keygen.generate_secret_key(sk);
    Publickey<S> pk(ctx0);
    keygen.generate_public_key(pk, sk);
    Relinkey<S> rk0(ctx0);
    keygen.generate_relin_key(rk0, sk);
    HEEncoder<S> enc0(ctx0);
    HEEncryptor<S> encryptor(ctx0, pk, 32);
    std::vector<double> m(4096);
    for (size_t i = 0; i < m.size(); i++) m[i] = 0.001 * (double) (i % 97) - 0.04;
    Plaintext<S> p(ctx0);
    enc0.encode(p, m, 1073741824.0);
    Ciphertext<S> c0(ctx0);
    encryptor.encrypt(c0, p);
    detail::hip(hipDeviceSynchronize());
    std::vector<std::thread> th;
    std::vector<std::string> errors(workers);
    for (int w = 0; w < workers; w++)
        th.emplace_back([&, w] {
            try {
                const int dev = w % ndev;
                detail::hip(hipSetDevice(dev));
                HEContext<S> ctx = make_ctx();          // tables on this thread's device
                ctx_dev[w] = ctx->device();
                Relinkey<S> rk(rk0);                    // replica on this device (peer copy when the devices differ)
                rk.set_context(ctx);
                Ciphertext<S> c(c0);                    // the input, copied to this device
                buf_dev[w] = dev;
                HEEncoder<S> enc(ctx);
                HEArithmeticOperator<S> op(ctx, enc);
                hipStream_t st;
                detail::hip(hipStreamCreate(&st));
                ExecutionOptions o;
                o.set_stream(st);
                {
                    Ciphertext<S> prod(ctx, o); // must not outlive its stream
                    op.multiply(c, c, prod, o);
                    op.relinearize_inplace(prod, rk, o);
                    op.rescale_inplace(prod, o);
                    prod.get_data(results[w], st);
                    detail::hip(hipStreamSynchronize(st));
                }
                detail::hip(hipStreamDestroy(st));
            } catch (const std::exception& e) { errors[w] = e.what(); }
        });
    for (auto& t : th) t.join();
    detail::hip(hipSetDevice(0));
    {
        // hegpu_broadcast_bytes: three replicas of an 80 MiB buffer (three 32 MiB chunks), peer access checked per edge
        const int nrep = 3;
        const size_t bytes = (size_t) 80 << 20;
        std::vector<int> devs(nrep);
        std::vector<void*> bufs(nrep, nullptr);
        std::vector<hegpu_stream> sts(nrep, nullptr);
        for (int i = 0; i < nrep; i++) {
            devs[i] = i % ndev;
            detail::hip(hipSetDevice(devs[i]));
            detail::hip(hipMalloc(&bufs[i], bytes));
            hipStream_t s;
            detail::hip(hipStreamCreate(&s));
            sts[i] = s;
        }
        detail::hip(hipSetDevice(devs[0]));
        std::vector<uint32_t> host(bytes / 4);
        for (size_t i = 0; i < host.size(); i++) host[i] = (uint32_t) (i * 2654435761u);
        detail::hip(hipMemcpyAsync(bufs[0], host.data(), bytes, hipMemcpyHostToDevice, (hipStream_t) sts[0]));
        int path = -1;
        const int rc = hegpu_broadcast_bytes(devs.data(), nrep, bufs.data(), bytes, sts.data(), &path);
        EXPECT(rc == 0, "hegpu_broadcast_bytes succeeds");
        const int shape = path & 0xff;
        EXPECT(shape == HEGPU_BCAST_FLAT || shape == HEGPU_BCAST_TREE, "the call reports the shape of the fan-out");
        EXPECT(path == hegpu_last_broadcast_path(), "the path is also kept for the calling thread");
        if (ndev == 1) EXPECT(path == (HEGPU_BCAST_FLAT | HEGPU_BCAST_SAME_DEVICE), "one device: flat, flagged as a functional run");
        else EXPECT(!(path & HEGPU_BCAST_SAME_DEVICE), "several devices are not reported as one");
        printf("    broadcast path: %s%s%s\n", shape == HEGPU_BCAST_FLAT ? "flat fan-out" : "binomial tree",
               (path & HEGPU_BCAST_STAGED) ? ", an edge WITHOUT peer access (host-staged)" : "",
               (path & HEGPU_BCAST_SAME_DEVICE) ? ", one device" : "");
        for (int i = 1; i < nrep; i++) {
            detail::hip(hipSetDevice(devs[i]));
            std::vector<uint32_t> back(bytes / 4);
            detail::hip(hipMemcpyAsync(back.data(), bufs[i], bytes, hipMemcpyDeviceToHost, (hipStream_t) sts[i]));
            detail::hip(hipStreamSynchronize((hipStream_t) sts[i]));
            EXPECT(back == host, "every replica holds the source's bytes (ordered on its own stream)");
        }
        for (int i = 0; i < nrep; i++) {
            detail::hip(hipSetDevice(devs[i]));
            detail::hip(hipStreamDestroy((hipStream_t) sts[i]));
            detail::hip(hipFree(bufs[i]));
        }
        detail::hip(hipSetDevice(0));
    }
    for (int w = 0; w < workers; w++) EXPECT(errors[w].empty(), ("device thread failed: " + errors[w]).c_str());
    EXPECT(ctx_dev[0] == 0 && ctx_dev[1] == 1 % ndev, "every thread's context lives on that thread's device");
    EXPECT(!results[0].empty() && results[0] == results[1], "both devices compute the same multiply + relinearize + rescale");
}

// HEContext::set_coeff_modulus_values (bfv/context.cu:149-265, ckks/context.cu:149-265): explicit primes equal to the
// default chain give the default chain's context -- same tables, same ciphertexts; the checks of the bit-size form apply
static void explicit_primes()
{
    constexpr auto S = Scheme::BFV;
    auto dflt = GenHEContext<S>();
    dflt->set_poly_modulus_degree(8192);
    dflt->set_coeff_modulus_default_values(1);
    dflt->set_plain_modulus(786433);
    dflt->generate();
    const std::vector<Data64> chain = dflt->get_key_modulus();
    const int Q = dflt->get_ciphertext_modulus_count();
    std::vector<Data64> q(chain.begin(), chain.begin() + Q), p(chain.begin() + Q, chain.end());
    auto expl = GenHEContext<S>();
    expl->set_poly_modulus_degree(8192);
    expl->set_coeff_modulus_values(q, p);
    expl->set_plain_modulus(786433);
    EXPECT(throws_logic([&] { expl->set_coeff_modulus_values(q, p); }), "the chain can be set once");
    expl->generate();
    EXPECT(expl->get_key_modulus() == chain && expl->get_ciphertext_modulus_count() == Q, "explicit primes are taken as given");
    for (const char* t : {"ntt_table", "intt_table", "n_inverse", "last_q_modinv", "base_Bsk", "base_change_matrix_Bsk",
                          "inv_prod_q_mod_Bsk", "Qi_t", "upper_threshold"}) {
        std::vector<uint64_t> a(1 << 20), b(1 << 20);
        const long na = hegpu_context_get(dflt->handle(), t, a.data(), (long) a.size());
        const long nb = hegpu_context_get(expl->handle(), t, b.data(), (long) b.size());
        a.resize(na > 0 ? na : 0);
        b.resize(nb > 0 ? nb : 0);
        EXPECT(na > 0 && a == b, (std::string("table ") + t + " identical for explicit primes = default chain").c_str());
    }
    // the same keys (same DRBG seed) and the same ciphertext bits on both contexts
    std::vector<Data64> ct[2];
    int k = 0;
    for (auto& ctx : {dflt, expl}) {
        HEKeyGenerator<S> keygen(ctx, 77);
        Secretkey<S> sk(ctx);
        keygen.generate_secret_key(sk);
        Publickey<S> pk(ctx);
        keygen.generate_public_key(pk, sk);
        HEEncoder<S> enc(ctx);
        HEEncryptor<S> encryptor(ctx, pk, 78);
        std::vector<uint64_t> m(8192);
        for (size_t i = 0; i < m.size(); i++) m[i] = (i * 7 + 1) % 786433;
        Plaintext<S> pt(ctx);
        enc.encode(pt, m);
        Ciphertext<S> c(ctx);
        encryptor.encrypt(c, pt);
        HEArithmeticOperator<S> op(ctx, enc);
        Ciphertext<S> prod(ctx);
        op.multiply(c, c, prod);
        prod.get_data(ct[k++]);
    }
    EXPECT(!ct[0].empty() && ct[0] == ct[1], "encrypt + multiply give the same residues on both contexts");
    // the checks of the bit-size form
    auto bad = GenHEContext<S>();
    bad->set_poly_modulus_degree(8192);
    EXPECT(throws_logic([&] { bad->set_coeff_modulus_values(q, {}); }), "P must not be empty");
    EXPECT(throws_logic([&] { bad->set_coeff_modulus_values({q[0], q[1] + 2}, p); }), "a value without a 2N-th root is refused");
    {
        std::vector<Data64> small_p = {q.back()}, big_q = {p[0], q[0]};
        bool logic = false;
        try { bad->set_coeff_modulus_values({p[0], p[0] - 0}, {65537}); } catch (const std::logic_error&) { logic = true; }
        EXPECT(logic, "P narrower than the Q digits it covers is refused (coefficient_validator)");
    }
    bool thrown = false;
    try {
        std::vector<Data64> twice(q);
        twice.insert(twice.end(), q.begin(), q.end()); // twice the budget of N = 8192 at 128-bit security
        bad->set_coeff_modulus_values(twice, p);
    } catch (const std::runtime_error&) { thrown = true; }
    EXPECT(thrown, "a chain beyond the security table is refused");
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    {
        auto bad = GenHEContext<Scheme::CKKS>();
        EXPECT(throws_logic([&] { bad->set_poly_modulus_degree(1000); }), "degree must be a power of two");
        auto sec = GenHEContext<Scheme::CKKS>();
        sec->set_poly_modulus_degree(4096);
        bool thrown = false; // thrown where the reference throws it (ckks/context.cu:95-117)
        try { sec->set_coeff_modulus_bit_sizes({40, 30, 30}, {40}); } catch (const std::runtime_error&) { thrown = true; }
        EXPECT(thrown, "140-bit chain at N=4096 violates the 128-bit security table");
    }
    ckks();
    bfv();
    ckks_pipeline();
    bfv_pipeline();
    ckks_encoder_flow();
    memory_pool();
    storage_manager();
    multi_device();
    explicit_primes();
    serializer_round_trip();
    serialize_all_objects();
    bfv_ntt_domain_and_shift();
    method_II_and_switch_key();
    tfhe_gates();
    printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}

// wire_dump.cpp -- writes what the class layer (include/heongpu/heongpu.hpp) serializes, next to what the LIVE
// objects hold, for tests/test_gpu_wire_format.py: an independent reader of the reference's wire format
// (oracle/wire.py, written from the reference's sources) parses every blob and compares each field and payload with
// the values taken from the objects through their accessors and through plain device-to-host copies -- not through
// the serializer.
//   wire_dump <dir>            every object of both schemes (needs a GPU)
//   wire_dump <dir> --context  the contexts only (host work: runs without a GPU)
#include <heongpu/heongpu.hpp>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

using namespace heongpu;

static std::string g_dir;
static FILE* g_manifest = nullptr;
static bool g_first = true;

template <typename T> static void blob(const std::string& name, const T& obj)
{
    std::stringstream ss;
    obj.save(ss);
    const std::string s = ss.str();
    std::ofstream(g_dir + "/" + name + ".bin", std::ios::binary).write(s.data(), (std::streamsize) s.size());
}
static void payload(const std::string& name, const Data64* dev, size_t count)
{
    std::vector<Data64> h(count);
    detail::hip(hipMemcpy(h.data(), dev, count * sizeof(Data64), hipMemcpyDeviceToHost));
    std::ofstream(g_dir + "/" + name + ".payload", std::ios::binary).write((const char*) h.data(), (std::streamsize) (count * 8));
}
static void entry(const std::string& name, const std::string& kind, const std::string& fields)
{
    fprintf(g_manifest, "%s\n  \"%s\": {\"kind\": \"%s\"%s%s}", g_first ? "" : ",", name.c_str(), kind.c_str(),
            fields.empty() ? "" : ", ", fields.c_str());
    g_first = false;
}
static std::string kv(const char* k, long long v) { return "\"" + std::string(k) + "\": " + std::to_string(v); }
static std::string kd(const char* k, double v)
{
    char b[64];
    snprintf(b, sizeof b, "\"%s\": %.17g", k, v);
    return b;
}
static std::string join(std::initializer_list<std::string> l)
{
    std::string s;
    for (const auto& x : l) s += (s.empty() ? "" : ", ") + x;
    return s;
}

template <Scheme S> static std::string context_fields(const HEContext<S>& c)
{
    std::string primes = "\"primes\": [";
    const auto km = c->get_key_modulus();
    for (size_t i = 0; i < km.size(); i++) primes += (i ? ", " : "") + std::to_string((unsigned long long) km[i]);
    primes += "]";
    return join({kv("n", c->n), kv("n_power", c->n_power), kv("Q_size", c->Q_size), kv("P_size", c->P_size),
                 kv("Q_prime_size", c->Q_prime_size), primes});
}

template <Scheme S> static void objects(const std::string& tag, HEContext<S> ctx)
{
    HEKeyGenerator<S> keygen(ctx, 5);
    Secretkey<S> sk(ctx), sk_old(ctx, 64);
    keygen.generate_secret_key(sk);
    keygen.generate_secret_key(sk_old);
    blob(tag + "_secretkey", sk);
    payload(tag + "_secretkey", sk.data(), (size_t) ctx->Q_prime_size * ctx->n);
    entry(tag + "_secretkey", "secretkey", join({kv("ring_size", sk.ring_size()), kv("coeff_modulus_count", sk.coeff_modulus_count()),
                                                  kv("n_power", ctx->n_power), kv("hamming_weight", ctx->n / 2)}));
    blob(tag + "_secretkey_hw64", sk_old);
    payload(tag + "_secretkey_hw64", sk_old.data(), (size_t) ctx->Q_prime_size * ctx->n);
    entry(tag + "_secretkey_hw64", "secretkey", join({kv("ring_size", ctx->n), kv("coeff_modulus_count", ctx->Q_prime_size),
                                                       kv("n_power", ctx->n_power), kv("hamming_weight", 64)}));
    Publickey<S> pk(ctx);
    keygen.generate_public_key(pk, sk);
    blob(tag + "_publickey", pk);
    payload(tag + "_publickey", pk.data(), (size_t) 2 * ctx->Q_prime_size * ctx->n);
    entry(tag + "_publickey", "publickey", join({kv("ring_size", pk.ring_size()), kv("coeff_modulus_count", pk.coeff_modulus_count())}));
    // the zlib framing of serializer::save_to_file around the same object
    serializer::save_to_file(pk, g_dir + "/" + tag + "_publickey.file");

    Relinkey<S> rk(ctx);
    keygen.generate_relin_key(rk, sk);
    blob(tag + "_relinkey", rk);
    payload(tag + "_relinkey", rk.data(), rk.size());
    const int d = ctx->P_size == 1 ? ctx->Q_size : (ctx->Q_size + (S == Scheme::BFV ? 2 : ctx->P_size) - 1) / (S == Scheme::BFV ? 2 : ctx->P_size);
    entry(tag + "_relinkey", "relinkey", join({kv("ring_size", ctx->n), kv("Q_prime_size", ctx->Q_prime_size), kv("Q_size", ctx->Q_size),
                                                kv("d", d), kv("size", (long long) rk.size()), kv("method", ctx->P_size == 1 ? 1 : 2)}));
    Switchkey<S> swk(ctx);
    keygen.generate_switch_key(swk, sk, sk_old);
    blob(tag + "_switchkey", swk);
    payload(tag + "_switchkey", swk.data(), swk.size());
    entry(tag + "_switchkey", "switchkey", join({kv("ring_size", ctx->n), kv("Q_prime_size", ctx->Q_prime_size), kv("Q_size", ctx->Q_size),
                                                  kv("d", d), kv("size", (long long) swk.size())}));

    std::vector<int> shifts = {1, -2};
    Galoiskey<S> gk(ctx, shifts);
    keygen.generate_galois_key(gk, sk);
    blob(tag + "_galoiskey", gk);
    std::string table = "\"galois_elt\": {";
    bool f = true;
    for (const auto& p : gk.galois_elt) { table += (f ? "" : ", ") + ("\"" + std::to_string(p.first) + "\": " + std::to_string(p.second)); f = false; }
    table += "}";
    std::string elts = "\"key_elements\": [";
    f = true;
    for (auto& p : gk.device_location_) {
        payload(tag + "_galoiskey_" + std::to_string(p.first), p.second.data(), gk.size());
        elts += (f ? "" : ", ") + std::to_string(p.first);
        f = false;
    }
    elts += "]";
    entry(tag + "_galoiskey", "galoiskey", join({kv("ring_size", ctx->n), kv("Q_prime_size", ctx->Q_prime_size), kv("Q_size", ctx->Q_size),
                                                  kv("d", d), kv("customized", 0), kv("group_order", gk.group_order_),
                                                  kv("galois_elt_zero", gk.galois_elt_zero), kv("size", (long long) gk.size()), table, elts}));
    std::vector<uint32_t> custom = {(uint32_t) gk.galois_elt[1], (uint32_t) (2 * ctx->n - 1)};
    Galoiskey<S> gc(ctx, custom);
    keygen.generate_galois_key(gc, sk);
    blob(tag + "_galoiskey_custom", gc);
    elts = "\"key_elements\": [";
    f = true;
    for (auto& p : gc.device_location_) {
        payload(tag + "_galoiskey_custom_" + std::to_string(p.first), p.second.data(), gc.size());
        elts += (f ? "" : ", ") + std::to_string(p.first);
        f = false;
    }
    elts += "]";
    entry(tag + "_galoiskey_custom", "galoiskey", join({kv("ring_size", ctx->n), kv("Q_prime_size", ctx->Q_prime_size), kv("Q_size", ctx->Q_size),
                                                         kv("d", d), kv("customized", 1), kv("group_order", gc.group_order_),
                                                         kv("galois_elt_zero", gc.galois_elt_zero), kv("size", (long long) gc.size()),
                                                         "\"custom_galois_elt\": [" + std::to_string(custom[0]) + ", " + std::to_string(custom[1]) + "]", elts}));

    HEEncoder<S> encoder(ctx);
    HEEncryptor<S> enc(ctx, pk, 6);
    HEArithmeticOperator<S> op(ctx, encoder);
    Plaintext<S> p(ctx);
    Ciphertext<S> c(ctx), c3(ctx);
    if constexpr (S == Scheme::BFV) {
        std::vector<uint64_t> m(ctx->n);
        for (int i = 0; i < ctx->n; i++) m[i] = (uint64_t) (i * 7 + 1) % 1032193;
        encoder.encode(p, m);
        blob(tag + "_plaintext", p);
        payload(tag + "_plaintext", p.data(), p.size());
        entry(tag + "_plaintext", "plaintext", join({kv("plain_size", (long long) p.size()), kv("in_ntt_domain", 0)}));
        enc.encrypt(c, p);
        blob(tag + "_ciphertext", c);
        payload(tag + "_ciphertext", c.data(), c.memory_size());
        entry(tag + "_ciphertext", "ciphertext", join({kv("ring_size", c.ring_size()), kv("coeff_modulus_count", c.coeff_modulus_count()),
                                                        kv("cipher_size", c.size()), kv("in_ntt_domain", c.in_ntt_domain()),
                                                        kv("relinearization_required", c.relinearization_required()),
                                                        kv("size", (long long) c.memory_size())}));
        op.multiply(c, c, c3);
        blob(tag + "_ciphertext_product", c3);
        payload(tag + "_ciphertext_product", c3.data(), c3.memory_size());
        entry(tag + "_ciphertext_product", "ciphertext", join({kv("ring_size", c3.ring_size()), kv("coeff_modulus_count", c3.coeff_modulus_count()),
                                                                kv("cipher_size", c3.size()), kv("in_ntt_domain", c3.in_ntt_domain()),
                                                                kv("relinearization_required", c3.relinearization_required()),
                                                                kv("size", (long long) c3.memory_size())}));
    } else {
        const double scale = 1073741824.0; // 2^30
        std::vector<double> m(ctx->n / 2);
        for (size_t i = 0; i < m.size(); i++) m[i] = 0.25 * (double) (i % 17) - 1.5;
        encoder.encode(p, m, scale);
        blob(tag + "_plaintext", p);
        payload(tag + "_plaintext", p.data(), p.size());
        entry(tag + "_plaintext", "plaintext", join({kv("plain_size", (long long) p.size()), kv("depth", p.depth()), kd("scale", p.scale()),
                                                      kv("in_ntt_domain", 1), kv("encoding", (int) p.encoding_type())}));
        enc.encrypt(c, p);
        auto words = [&](Ciphertext<S>& x) { return (size_t) x.size() * (x.coeff_modulus_count() - x.depth()) * x.ring_size(); };
        auto cipher_fields = [&](Ciphertext<S>& x) {
            return join({kv("ring_size", x.ring_size()), kv("coeff_modulus_count", x.coeff_modulus_count()), kv("cipher_size", x.size()),
                         kv("depth", x.depth()), kd("scale", x.scale()), kv("in_ntt_domain", x.in_ntt_domain()),
                         kv("encoding", (int) x.encoding_type()), kv("rescale_required", x.rescale_required()),
                         kv("relinearization_required", x.relinearization_required()), kv("size", (long long) words(x))});
        };
        blob(tag + "_ciphertext", c);
        payload(tag + "_ciphertext", c.data(), words(c));
        entry(tag + "_ciphertext", "ciphertext", cipher_fields(c));
        op.multiply(c, c, c3);
        blob(tag + "_ciphertext_product", c3); // three parts, un-rescaled
        payload(tag + "_ciphertext_product", c3.data(), words(c3));
        entry(tag + "_ciphertext_product", "ciphertext", cipher_fields(c3));
        op.relinearize_inplace(c3, rk);
        op.rescale_inplace(c3);
        blob(tag + "_ciphertext_depth1", c3);
        payload(tag + "_ciphertext_depth1", c3.data(), words(c3)); // the buffer itself stays three parts x four limbs long
        entry(tag + "_ciphertext_depth1", "ciphertext", cipher_fields(c3));
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: wire_dump <dir> [--context]\n"); return 2; }
    g_dir = argv[1];
    const bool context_only = argc > 2 && std::string(argv[2]) == "--context";
    g_manifest = fopen((g_dir + "/manifest.json").c_str(), "w");
    if (!g_manifest) { perror("manifest"); return 2; }
    fprintf(g_manifest, "{");
    try {
        HEContext<Scheme::BFV> b = GenHEContext<Scheme::BFV>();
        b->set_poly_modulus_degree(4096);
        b->set_coeff_modulus_default_values(1);
        b->set_plain_modulus(1032193);
        blob("bfv_context", *b); // before generate(), as example/basic/13_bfv_serialization.cpp does
        HEContext<Scheme::CKKS> c = GenHEContext<Scheme::CKKS>();
        c->set_poly_modulus_degree(8192);
        c->set_coeff_modulus_bit_sizes({40, 30, 30, 30}, {40});
        blob("ckks_context", *c);
        HEContext<Scheme::CKKS> c2 = GenHEContext<Scheme::CKKS>(sec_level_type::none);
        c2->set_poly_modulus_degree(4096);
        c2->set_coeff_modulus_bit_sizes({40, 30, 30, 30}, {40, 40}); // two special primes: method II
        blob("ckks2_context", *c2);
        if (!context_only) {
            b->generate();
            c->generate();
            c2->generate();
        }
        entry("bfv_context", "context", context_only ? join({kv("n", 4096), kv("plain_modulus", 1032193), kv("sec_level", 1), kv("method", 1)})
                                                     : join({context_fields(b), kv("plain_modulus", 1032193), kv("sec_level", 1), kv("method", 1)}));
        entry("ckks_context", "context", context_only ? join({kv("n", 8192), kv("sec_level", 1), kv("method", 1)})
                                                      : join({context_fields(c), kv("sec_level", 1), kv("method", 1)}));
        entry("ckks2_context", "context", context_only ? join({kv("n", 4096), kv("sec_level", 0), kv("method", 2)})
                                                       : join({context_fields(c2), kv("sec_level", 0), kv("method", 2)}));
        if (!context_only) {
            objects<Scheme::BFV>("bfv", b);
            objects<Scheme::CKKS>("ckks", c);
            objects<Scheme::CKKS>("ckks2", c2);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "wire_dump: %s\n", e.what());
        return 1;
    }
    fprintf(g_manifest, "\n}\n");
    fclose(g_manifest);
    return 0;
}

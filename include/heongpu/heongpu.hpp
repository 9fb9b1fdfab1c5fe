// heongpu.hpp -- C++ class layer over the C ABI (include/hegpu.h), mirroring
// the reference's public surface for the hot path so that consumer code reads
// the same:
//   heongpu::HEContext<S> ctx = heongpu::GenHEContext<S>(sec);
//   ctx->set_poly_modulus_degree(n); ctx->set_coeff_modulus_bit_sizes(Q, P); ctx->generate();
//   heongpu::Ciphertext<S> ct(ctx);  heongpu::Relinkey<S> rk(ctx);  heongpu::Galoiskey<S> gk(ctx, shifts);
//   heongpu::HEArithmeticOperator<S> op(ctx);
//   op.multiply(a, b, c); op.relinearize_inplace(c, rk); op.rescale_inplace(c); op.rotate_rows(c, d, gk, 1);
// Reference: src/include/heongpu/heongpu.hpp:9-45, util/schemes.h:15-67,
// util/storagemanager.cuh:23-97, util/devicevector.cuh:17-173,
// host/{bfv,ckks}/{context,ciphertext,evaluationkey,operator}.cuh.
//
// What is NOT here yet: key generation, encryption /
// decryption, encoders.  Until then keys and ciphertexts are filled through
// the `load()` members with data produced elsewhere (tests: seeded synthetic
// data / python big-int key generation in the reference's layouts).
//
// Header-only; link against heongpu_amd/lib/libhegpu.so and the HIP runtime.
#pragma once
#include "../hegpu.h"
#if defined(HEONGPU_WITH_ZLIB)
#include <zlib.h> // serializer::compress / decompress (reference util/serializer.cpp)
#endif
#include <hip/hip_runtime.h>
#if defined(HEONGPU_CUDA_NAMES)
#include "cuda_names.hpp" // lets unmodified consumers of the reference (benchmark/*.cpp) use their cuda* calls
#endif
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <complex>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <istream>
#include <map>
#include <ostream>
#include <sstream>
#include <random>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace heongpu {

typedef unsigned long long Data64;

enum class Scheme { BFV = 1, CKKS = 2, TFHE = 3 };                           // util/schemes.h:15-20
enum class sec_level_type { none = 0, sec128 = 128, sec192 = 192, sec256 = 256 };
enum class storage_type : std::uint8_t { HOST = 0x1, DEVICE = 0x2 };        // util/storagemanager.cuh:23-27
enum class keyswitching_type { NONE = 0, KEYSWITCHING_METHOD_I = 1, KEYSWITCHING_METHOD_II = 2 };

// util/storagemanager.cuh:34-97
struct ExecutionOptions {
    hipStream_t stream_ = nullptr;
    storage_type storage_ = storage_type::DEVICE;
    bool keep_initial_condition_ = true;
    ExecutionOptions& set_stream(hipStream_t s) { stream_ = s; return *this; }
    ExecutionOptions& set_storage_type(storage_type s) { storage_ = s; return *this; }
    ExecutionOptions& set_initial_location(bool k) { keep_initial_condition_ = k; return *this; }
};

class HipException : public std::runtime_error { // reference util/util.cuh:25-45 CudaException
  public:
    explicit HipException(const std::string& m) : std::runtime_error(m) {}
};

namespace detail {
inline void check(int rc)
{
    if (rc == 0) return;
    const std::string msg = hegpu_last_error();
    switch (rc) {
        case HEGPU_E_INVALID: throw std::invalid_argument(msg);
        case HEGPU_E_LOGIC: throw std::logic_error(msg);
        case HEGPU_E_RUNTIME: throw std::runtime_error(msg);
        default: throw HipException(msg);
    }
}
inline void hip(hipError_t e)
{
    if (e != hipSuccess) throw HipException(hipGetErrorString(e));
}
// one stream-ordered pool per device, never trimmed (the reference's RMM
// pool_memory_resource, util/memorypool.cuh:56-117)
inline void init_pool()
{
    static bool done = false;
    if (done) return;
    int dev = 0;
    hip(hipGetDevice(&dev));
    hipMemPool_t pool;
    hip(hipDeviceGetDefaultMemPool(&pool, dev));
    uint64_t keep = UINT64_MAX;
    hip(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    done = true;
}
} // namespace detail

// util/devicevector.cuh:17-173: stream-ordered device buffer
template <typename T> class DeviceVector {
  public:
    DeviceVector() = default;
    explicit DeviceVector(size_t n, hipStream_t s = nullptr) { resize(n, s); }
    DeviceVector(const std::vector<T>& h, hipStream_t s = nullptr)
    {
        resize(h.size(), s);
        if (!h.empty()) detail::hip(hipMemcpyAsync(p_, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    }
    DeviceVector(const DeviceVector& o) { copy_from(o); }
    DeviceVector& operator=(const DeviceVector& o)
    {
        if (this != &o) { release(); copy_from(o); }
        return *this;
    }
    DeviceVector(DeviceVector&& o) noexcept : p_(o.p_), n_(o.n_), s_(o.s_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceVector& operator=(DeviceVector&& o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; s_ = o.s_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceVector() { release(); }
    void resize(size_t n, hipStream_t s = nullptr)
    {
        release();
        s_ = s;
        n_ = n;
        if (n) {
            detail::init_pool();
            detail::hip(hipMallocAsync((void**) &p_, n * sizeof(T), s));
        }
    }
    T* data() const { return p_; }
    size_t size() const { return n_; }
    hipStream_t stream() const { return s_; }
    void set_stream(hipStream_t s) { s_ = s; }

  private:
    void release()
    {
        if (p_) (void) hipFreeAsync(p_, s_);
        p_ = nullptr;
        n_ = 0;
    }
    void copy_from(const DeviceVector& o)
    {
        resize(o.n_, o.s_);
        if (n_) detail::hip(hipMemcpyAsync(p_, o.p_, n_ * sizeof(T), hipMemcpyDeviceToDevice, s_));
    }
    T* p_ = nullptr;
    size_t n_ = 0;
    hipStream_t s_ = nullptr;
};

template <typename T> using HostVector = std::vector<T>; // util/hostvector.cuh:17-31 (pinned there)

// ------------------------------------------------------------------ context
template <Scheme S> class HEContextImpl { // BFV / CKKS; the TFHE specialisation is at the end of this file

  public:
    explicit HEContextImpl(sec_level_type sec = sec_level_type::sec128) : sec_level_(sec) {}
    ~HEContextImpl() { if (h_) hegpu_context_destroy(h_); }
    HEContextImpl(const HEContextImpl&) = delete;

    void set_poly_modulus_degree(size_t n) // ckks/context.cu:24-52
    {
        if (coeff_modulus_specified_ || poly_modulus_degree_specified_)
            throw std::logic_error("Poly modulus degree cannot be changed after the coeff_modulus is specified!");
        if (n == 0 || (n & (n - 1))) throw std::logic_error("Poly modulus degree have to be power of two");
        if (n > 65536 || n < 4096) throw std::logic_error("Poly modulus degree is not supported");
        n_ = n;
        poly_modulus_degree_specified_ = true;
    }
    void set_coeff_modulus_bit_sizes(const std::vector<int>& q, const std::vector<int>& p) // :54-147
    {
        if (coeff_modulus_specified_ || context_generated_ || !poly_modulus_degree_specified_)
            throw std::logic_error("Coeff_modulus cannot be changed after the context is generated!");
        if (p.empty()) throw std::logic_error("log_P_bases_bit_sizes cannot be empty!");
        q_bits_ = q;
        p_bits_ = p;
        use_default_ = false;
        coeff_modulus_specified_ = true;
    }
    void set_coeff_modulus_default_values(int p_count) // bfv/context.cu:267-374
    {
        if (coeff_modulus_specified_ || context_generated_ || !poly_modulus_degree_specified_)
            throw std::logic_error("Coeff_modulus cannot be changed after the context is generated!");
        if (p_count < 1) throw std::logic_error("P_modulus_size cannot be lower than 1!");
        default_p_ = p_count;
        use_default_ = true;
        coeff_modulus_specified_ = true;
    }
    void set_plain_modulus(int t) { plain_modulus_ = (uint64_t) t; } // bfv/context.cu:376-389
    void generate()
    {
        if (context_generated_ || !poly_modulus_degree_specified_ || !coeff_modulus_specified_)
            throw std::logic_error("Context is already generated or parameters are missing!");
        const int sec = (sec_level_ == sec_level_type::none) ? HEGPU_SEC_NONE : (int) sec_level_;
        if (sec_level_ == sec_level_type::sec192 || sec_level_ == sec_level_type::sec256)
            throw std::runtime_error("Invalid security level"); // only the 128-bit table is carried so far
        if (use_default_)
            detail::check(hegpu_context_create_default((int) S, (int) n_, default_p_, plain_modulus_, sec, &h_));
        else
            detail::check(hegpu_context_create((int) S, (int) n_, q_bits_.data(), (int) q_bits_.size(), p_bits_.data(),
                                               (int) p_bits_.size(), plain_modulus_, sec, &h_));
        detail::check(hegpu_context_upload(h_));
        n_power = (int) hegpu_context_int(h_, "n_power");
        Q_size = (int) hegpu_context_int(h_, "Q_size");
        P_size = (int) hegpu_context_int(h_, "P_size");
        Q_prime_size = (int) hegpu_context_int(h_, "Q_prime_size");
        n = (int) n_;
        keyswitching_type_ = P_size == 1 ? keyswitching_type::KEYSWITCHING_METHOD_I
                                         : keyswitching_type::KEYSWITCHING_METHOD_II;
        prime_vector_.resize(Q_prime_size);
        hegpu_context_get(h_, "modulus", (uint64_t*) prime_vector_.data(), Q_prime_size);
        context_generated_ = true;
    }
    inline int get_poly_modulus_degree() const noexcept { return n; }
    inline int get_ciphertext_modulus_count() const noexcept { return Q_size; }
    inline int get_key_modulus_count() const noexcept { return Q_prime_size; }
    inline std::vector<Data64> get_key_modulus() const noexcept { return prime_vector_; }
    inline int get_log_poly_modulus_degree() const noexcept { return n_power; }
    inline uint64_t get_plain_modulus() const noexcept { return plain_modulus_; }
    hegpu_context* handle() const { return h_; }

    int n = 0, n_power = 0, Q_size = 0, P_size = 0, Q_prime_size = 0;
    bool context_generated_ = false;
    keyswitching_type keyswitching_type_ = keyswitching_type::NONE;
    std::vector<Data64> prime_vector_;

  private:
    hegpu_context* h_ = nullptr;
    sec_level_type sec_level_;
    size_t n_ = 0;
    uint64_t plain_modulus_ = 0;
    std::vector<int> q_bits_, p_bits_;
    int default_p_ = 1;
    bool use_default_ = false, poly_modulus_degree_specified_ = false, coeff_modulus_specified_ = false;
};

template <Scheme S> using HEContext = std::shared_ptr<HEContextImpl<S>>;
template <Scheme S> HEContext<S> GenHEContext(sec_level_type sec = sec_level_type::sec128) // util/schemes.h:24-31
{
    return std::make_shared<HEContextImpl<S>>(sec);
}

// ------------------------------------------------------------------ ciphertext
template <Scheme S> class HEArithmeticOperator;

template <Scheme S> class Ciphertext { // host/{ckks,bfv}/ciphertext.cuh
    friend class HEArithmeticOperator<S>;

  public:
    explicit Ciphertext(HEContext<S> context, const ExecutionOptions& options = ExecutionOptions())
    {
        if (!context || !context->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        ring_size_ = context->n;
        coeff_modulus_count_ = context->Q_size;
        cipher_size_ = 2;
        in_ntt_domain_ = (S == Scheme::CKKS); // CKKS ciphertexts live in the NTT domain (ckks/ciphertext.cu)
        storage_type_ = options.storage_;
        device_locations_.set_stream(options.stream_);
    }
    Data64* data() { return device_locations_.data(); }
    const Data64* data() const { return device_locations_.data(); }
    size_t memory_size() const { return device_locations_.size(); }
    void memory_set(DeviceVector<Data64>&& m) { device_locations_ = std::move(m); }
    void switch_stream(hipStream_t s) { device_locations_.set_stream(s); }
    hipStream_t stream() const noexcept { return device_locations_.stream(); }
    bool is_on_device() const noexcept { return storage_type_ == storage_type::DEVICE; }
    inline int ring_size() const noexcept { return ring_size_; }
    inline int coeff_modulus_count() const noexcept { return coeff_modulus_count_; }
    inline int size() const noexcept { return cipher_size_; }
    inline int depth() const noexcept { return depth_; }
    inline double scale() const noexcept { return scale_; }
    inline bool in_ntt_domain() const noexcept { return in_ntt_domain_; }
    inline bool rescale_required() const noexcept { return rescale_required_; }
    inline bool relinearization_required() const noexcept { return relinearization_required_; }
    void get_data(std::vector<Data64>& out, hipStream_t s = nullptr) const
    {
        out.resize(device_locations_.size());
        detail::hip(hipMemcpyAsync(out.data(), device_locations_.data(), out.size() * sizeof(Data64),
                                   hipMemcpyDeviceToHost, s));
        detail::hip(hipStreamSynchronize(s));
    }
    // used by the encryptor: take ownership of freshly produced residues
    void adopt(DeviceVector<Data64>&& m, int cipher_size, int depth, double scale)
    {
        device_locations_ = std::move(m);
        cipher_size_ = cipher_size;
        depth_ = depth;
        scale_ = scale;
        ciphertext_generated_ = true;
        relinearization_required_ = false;
        rescale_required_ = false;
    }
    // Without the encryptor: fill with residues laid out [size][Q - depth][N].
    void load(const std::vector<Data64>& host, int cipher_size, int depth, double scale = 1.0, hipStream_t s = nullptr)
    {
        const size_t want = (size_t) cipher_size * (coeff_modulus_count_ - depth) * ring_size_;
        if (host.size() != want) throw std::invalid_argument("Invalid Ciphertexts size!");
        device_locations_ = DeviceVector<Data64>(host, s);
        cipher_size_ = cipher_size;
        depth_ = depth;
        scale_ = scale;
        ciphertext_generated_ = true;
        relinearization_required_ = cipher_size == 3;
    }

    // Wire format of the reference, field for field (ckks/ciphertext.cu:171-300,
    // bfv/ciphertext.cu:165-290): scheme (u8), ring size, modulus count, size [, depth] (int),
    // ntt flag (bool), storage (u8) [, scale (double), encoding (u8), rescale flag (bool)],
    // relinearization flag, generated flag (bool), element count (u32), residues (u64).
    void save(std::ostream& os) const
    {
        if (!ciphertext_generated_) throw std::runtime_error("Ciphertext is not generated so can not be serialized!");
        const std::uint8_t scheme = (std::uint8_t) S, storage = (std::uint8_t) storage_type::DEVICE, enc = 0;
        os.write((const char*) &scheme, 1);
        os.write((const char*) &ring_size_, sizeof(int));
        os.write((const char*) &coeff_modulus_count_, sizeof(int));
        os.write((const char*) &cipher_size_, sizeof(int));
        if (S == Scheme::CKKS) os.write((const char*) &depth_, sizeof(int));
        os.write((const char*) &in_ntt_domain_, sizeof(bool));
        os.write((const char*) &storage, 1);
        if (S == Scheme::CKKS) {
            os.write((const char*) &scale_, sizeof(double));
            os.write((const char*) &enc, 1);
            os.write((const char*) &rescale_required_, sizeof(bool));
        }
        os.write((const char*) &relinearization_required_, sizeof(bool));
        os.write((const char*) &ciphertext_generated_, sizeof(bool));
        const std::uint32_t count = (std::uint32_t) ((size_t) cipher_size_ * (coeff_modulus_count_ - depth_) * ring_size_);
        std::vector<Data64> host;
        get_data(host, device_locations_.stream());
        if (host.size() < count) throw std::runtime_error("Ciphertext memory is smaller than its description!");
        os.write((const char*) &count, sizeof(count));
        os.write((const char*) host.data(), sizeof(Data64) * count);
    }
    void load(std::istream& is)
    {
        if (ciphertext_generated_) throw std::runtime_error("Ciphertext has been already exist!");
        std::uint8_t scheme = 0, storage = 0, enc = 0;
        is.read((char*) &scheme, 1);
        if (scheme != (std::uint8_t) S) throw std::runtime_error("Invalid scheme binary!");
        int ring = 0, count_mod = 0;
        is.read((char*) &ring, sizeof(int));
        is.read((char*) &count_mod, sizeof(int));
        if (ring != ring_size_ || count_mod != coeff_modulus_count_)
            throw std::runtime_error("Ciphertext binary does not match the context!");
        is.read((char*) &cipher_size_, sizeof(int));
        depth_ = 0;
        if (S == Scheme::CKKS) is.read((char*) &depth_, sizeof(int));
        is.read((char*) &in_ntt_domain_, sizeof(bool));
        is.read((char*) &storage, 1);
        if (S == Scheme::CKKS) {
            is.read((char*) &scale_, sizeof(double));
            is.read((char*) &enc, 1);
            is.read((char*) &rescale_required_, sizeof(bool));
        }
        is.read((char*) &relinearization_required_, sizeof(bool));
        bool generated = false;
        is.read((char*) &generated, sizeof(bool));
        std::uint32_t count = 0;
        is.read((char*) &count, sizeof(count));
        if (!is || depth_ < 0 || depth_ >= coeff_modulus_count_ ||
            count != (std::uint32_t) ((size_t) cipher_size_ * ring_size_ * (coeff_modulus_count_ - depth_)))
            throw std::runtime_error("Ciphertext size is not correct!");
        std::vector<Data64> host(count);
        is.read((char*) host.data(), sizeof(Data64) * count);
        if (!is) throw std::runtime_error("Ciphertext binary is truncated!");
        device_locations_ = DeviceVector<Data64>(host, device_locations_.stream());
        detail::hip(hipStreamSynchronize(device_locations_.stream()));
        storage_type_ = storage_type::DEVICE;
        ciphertext_generated_ = true;
    }

  private:
    int ring_size_ = 0, coeff_modulus_count_ = 0, cipher_size_ = 2, depth_ = 0;
    double scale_ = 0;
    bool in_ntt_domain_ = false, rescale_required_ = false, relinearization_required_ = false;
    bool ciphertext_generated_ = false;
    storage_type storage_type_ = storage_type::DEVICE;
    DeviceVector<Data64> device_locations_;
};

// ------------------------------------------------------------------ keys
template <Scheme S> class Relinkey { // host/*/evaluationkey.cuh; size 2*d*Q'*N (evaluationkey.cu:30-36)
  public:
    explicit Relinkey(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        const int m = (S == Scheme::BFV) ? 2 : context_->P_size;
        d_ = context_->P_size == 1 ? context_->Q_size : (context_->Q_size + m - 1) / m;
        relinkey_size_ = (size_t) 2 * d_ * context_->Q_prime_size * context_->n;
        key_type = context_->P_size == 1 ? 1 : 2;
    }
    Data64* data() { return device_location_.data(); }
    size_t size() const { return relinkey_size_; }
    void load(const std::vector<Data64>& host, hipStream_t s = nullptr) // a key produced elsewhere
    {
        if (host.size() != relinkey_size_) throw std::invalid_argument("Invalid relinkey size!");
        device_location_ = DeviceVector<Data64>(host, s);
        relin_key_generated_ = true;
    }
    void memory_set(DeviceVector<Data64>&& m) { device_location_ = std::move(m); }
    int key_type = 1;
    bool relin_key_generated_ = false;

  private:
    HEContext<S> context_;
    int d_ = 0;
    size_t relinkey_size_ = 0;
    DeviceVector<Data64> device_location_;
};

template <Scheme S> class Galoiskey { // host/*/evaluationkey.cuh; keygeneration.cu:684-728
  public:
    Galoiskey(HEContext<S> context, const std::vector<int>& shifts) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        group_order_ = (S == Scheme::BFV) ? 3 : 5; // bfv/evaluationkey.cu:308, ckks/evaluationkey.cu:408
        for (int sh : shifts) galois_elt[sh] = hegpu_steps_to_galois_elt(sh, context_->n, group_order_);
        galois_elt_zero = hegpu_steps_to_galois_elt(0, context_->n, group_order_); // column rotation / conjugation
        const int m = (S == Scheme::BFV) ? 2 : context_->P_size;
        const int d = context_->P_size == 1 ? context_->Q_size : (context_->Q_size + m - 1) / m;
        galoiskey_size_ = (size_t) 2 * d * context_->Q_prime_size * context_->n;
    }
    size_t size() const { return galoiskey_size_; }
    void load(int galois_element, const std::vector<Data64>& host, hipStream_t s = nullptr)
    {
        if (host.size() != galoiskey_size_) throw std::invalid_argument("Invalid galoiskey size!");
        device_location_[galois_element] = DeviceVector<Data64>(host, s);
    }
    bool galois_key_generated_ = false;
    int galois_elt_zero = 0;
    std::map<int, int> galois_elt;                           // shift -> Galois element
    std::map<int, DeviceVector<Data64>> device_location_;    // Galois element -> key
    int group_order_ = 5;

  private:
    HEContext<S> context_;
    size_t galoiskey_size_ = 0;
};

// ------------------------------------------------------------------ secret / public key, plaintext
template <Scheme S> class Secretkey { // host/*/secretkey.cuh; [Q'][N], NTT domain
  public:
    explicit Secretkey(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        hamming_weight_ = context_->n >> 1; // secretkey.cu:23
    }
    Secretkey(HEContext<S> context, int hamming_weight) : Secretkey(std::move(context))
    {
        if (hamming_weight <= 0 || hamming_weight > context_->n)
            throw std::invalid_argument("hamming weight has to be in range 0 to ring size."); // secretkey.cu:43
        hamming_weight_ = hamming_weight;
    }
    Data64* data() { return device_locations_.data(); }
    const Data64* data() const { return device_locations_.data(); }
    void memory_set(DeviceVector<Data64>&& m) { device_locations_ = std::move(m); }
    int hamming_weight_ = 0;
    bool secret_key_generated_ = false;

  private:
    HEContext<S> context_;
    DeviceVector<Data64> device_locations_;
};

template <Scheme S> class Publickey { // host/*/publickey.cuh; [2][Q'][N], NTT domain
  public:
    explicit Publickey(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
    }
    Data64* data() { return device_locations_.data(); }
    const Data64* data() const { return device_locations_.data(); }
    void memory_set(DeviceVector<Data64>&& m) { device_locations_ = std::move(m); }
    bool public_key_generated_ = false;

  private:
    HEContext<S> context_;
    DeviceVector<Data64> device_locations_;
};

template <Scheme S> class Plaintext { // host/*/plaintext.cuh -- CKKS: [Q - depth][N] NTT domain (+ depth, scale); BFV: [N] mod t
  public:
    explicit Plaintext(HEContext<S> context, const ExecutionOptions& options = ExecutionOptions())
        : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        device_locations_.set_stream(options.stream_);
    }
    Data64* data() { return device_locations_.data(); }
    const Data64* data() const { return device_locations_.data(); }
    size_t size() const { return device_locations_.size(); }
    void memory_set(DeviceVector<Data64>&& m) { device_locations_ = std::move(m); }
    // until the encoders exist: residues [Q - depth][N] of the scaled message, NTT domain
    void load(const std::vector<Data64>& host, int depth, double scale, hipStream_t s = nullptr)
    {
        const size_t want = (S == Scheme::CKKS) ? (size_t) (context_->Q_size - depth) * context_->n : (size_t) context_->n;
        if (host.size() != want) throw std::invalid_argument("Invalid plaintext size!");
        device_locations_ = DeviceVector<Data64>(host, s);
        depth_ = depth;
        scale_ = scale;
        plaintext_generated_ = true;
    }
    void get_data(std::vector<Data64>& out, hipStream_t s = nullptr) const
    {
        out.resize(device_locations_.size());
        detail::hip(hipMemcpyAsync(out.data(), device_locations_.data(), out.size() * sizeof(Data64),
                                   hipMemcpyDeviceToHost, s));
        detail::hip(hipStreamSynchronize(s));
    }
    int depth_ = 0;
    double scale_ = 0;
    bool plaintext_generated_ = false;

  private:
    HEContext<S> context_;
    DeviceVector<Data64> device_locations_;
};

// ------------------------------------------------------------------ key generator / encryptor / decryptor
// Key-switching method I (P_size == 1).  Random values come from the backend's DRBG
// (csrc/drbg.hpp); like the reference's generator it is seeded from std::random_device unless
// a seed is given.
template <Scheme S> class HEKeyGenerator { // host/*/keygenerator.cuh
  public:
    explicit HEKeyGenerator(HEContext<S> context) : HEKeyGenerator(std::move(context), std::random_device{}()) {}
    HEKeyGenerator(HEContext<S> context, std::uint64_t seed) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        detail::check(hegpu_rng_create(seed, &rng_));
    }
    ~HEKeyGenerator() { hegpu_rng_destroy(rng_); }
    HEKeyGenerator(const HEKeyGenerator&) = delete;
    HEKeyGenerator& operator=(const HEKeyGenerator&) = delete;

    void generate_secret_key(Secretkey<S>& sk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (sk.secret_key_generated_) throw std::logic_error("Secretkey is already generated!");
        DeviceVector<Data64> out((size_t) context_->Q_prime_size * context_->n, o.stream_);
        Workspace ws(context_, HEGPU_OP_KEYGEN_SECRET, o.stream_);
        detail::check(hegpu_generate_secret_key(context_->handle(), rng_, sk.hamming_weight_, (uint64_t*) out.data(),
                                                ws.p(), ws.bytes(), o.stream_));
        sk.memory_set(std::move(out));
        sk.secret_key_generated_ = true;
    }
    void generate_public_key(Publickey<S>& pk, Secretkey<S>& sk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
        if (pk.public_key_generated_) throw std::logic_error("Publickey is already generated!");
        DeviceVector<Data64> out((size_t) 2 * context_->Q_prime_size * context_->n, o.stream_);
        Workspace ws(context_, HEGPU_OP_KEYGEN_PUBLIC, o.stream_);
        detail::check(hegpu_generate_public_key(context_->handle(), rng_, (const uint64_t*) sk.data(),
                                                (uint64_t*) out.data(), ws.p(), ws.bytes(), o.stream_));
        pk.memory_set(std::move(out));
        pk.public_key_generated_ = true;
    }
    void generate_relin_key(Relinkey<S>& rk, Secretkey<S>& sk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
        if (rk.relin_key_generated_) throw std::logic_error("Relinkey is already generated!");
        DeviceVector<Data64> out(rk.size(), o.stream_);
        Workspace ws(context_, HEGPU_OP_KEYGEN_SWITCH, o.stream_);
        detail::check(hegpu_generate_relin_key(context_->handle(), rng_, (const uint64_t*) sk.data(),
                                               (uint64_t*) out.data(), ws.p(), ws.bytes(), o.stream_));
        rk.memory_set(std::move(out));
        rk.relin_key_generated_ = true;
    }
    void generate_galois_key(Galoiskey<S>& gk, Secretkey<S>& sk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
        if (gk.galois_key_generated_) throw std::logic_error("Galoiskey is already generated!");
        Workspace ws(context_, HEGPU_OP_KEYGEN_SWITCH, o.stream_);
        for (auto& g : gk.galois_elt) {
            if (gk.device_location_.count(g.second)) continue;
            DeviceVector<Data64> out(gk.size(), o.stream_);
            detail::check(hegpu_generate_galois_key(context_->handle(), rng_, (const uint64_t*) sk.data(), g.second,
                                                    (uint64_t*) out.data(), ws.p(), ws.bytes(), o.stream_));
            gk.device_location_[g.second] = std::move(out);
        }
        if (!gk.device_location_.count(gk.galois_elt_zero)) { // "Columns Rotate" key (keygenerator.cu:508-560)
            DeviceVector<Data64> out(gk.size(), o.stream_);
            detail::check(hegpu_generate_galois_key(context_->handle(), rng_, (const uint64_t*) sk.data(),
                                                    gk.galois_elt_zero, (uint64_t*) out.data(), ws.p(), ws.bytes(),
                                                    o.stream_));
            gk.device_location_[gk.galois_elt_zero] = std::move(out);
        }
        gk.galois_key_generated_ = true;
    }

  private:
    struct Workspace {
        Workspace(const HEContext<S>& c, int op, hipStream_t s)
            : v((hegpu_workspace_bytes(c->handle(), op, 0, 1) + 7) / 8, s)
        {
        }
        void* p() { return v.data(); }
        size_t bytes() const { return v.size() * sizeof(Data64); }
        DeviceVector<Data64> v;
    };
    HEContext<S> context_;
    hegpu_rng* rng_ = nullptr;
};

template <Scheme S> class HEEncryptor { // host/ckks/encryptor.cuh (public-key encryption)
  public:
    HEEncryptor(HEContext<S> context, Publickey<S>& pk) : HEEncryptor(std::move(context), pk, std::random_device{}()) {}
    HEEncryptor(HEContext<S> context, Publickey<S>& pk, std::uint64_t seed) : context_(std::move(context)), pk_(&pk)
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        if (!pk.public_key_generated_) throw std::logic_error("Publickey is not generated!");
        detail::check(hegpu_rng_create(seed, &rng_));
    }
    ~HEEncryptor() { hegpu_rng_destroy(rng_); }
    HEEncryptor(const HEEncryptor&) = delete;
    HEEncryptor& operator=(const HEEncryptor&) = delete;

    void encrypt(Ciphertext<S>& ct, Plaintext<S>& pt, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!pt.plaintext_generated_ || pt.depth_ != 0) throw std::invalid_argument("Invalid plaintext size."); // encryptor.cuh:56
        DeviceVector<Data64> out((size_t) 2 * context_->Q_size * context_->n, o.stream_);
        const int opid = (S == Scheme::CKKS) ? HEGPU_OP_CKKS_ENCRYPT : HEGPU_OP_BFV_ENCRYPT;
        DeviceVector<Data64> ws((hegpu_workspace_bytes(context_->handle(), opid, 0, 1) + 7) / 8, o.stream_);
        if (S == Scheme::CKKS)
            detail::check(hegpu_ckks_encrypt(context_->handle(), rng_, (const uint64_t*) pk_->data(),
                                             (const uint64_t*) pt.data(), (uint64_t*) out.data(), ws.data(),
                                             ws.size() * sizeof(Data64), o.stream_));
        else
            detail::check(hegpu_bfv_encrypt(context_->handle(), rng_, (const uint64_t*) pk_->data(),
                                            (const uint64_t*) pt.data(), (uint64_t*) out.data(), ws.data(),
                                            ws.size() * sizeof(Data64), o.stream_));
        ct.adopt(std::move(out), 2, 0, pt.scale_);
    }

  private:
    HEContext<S> context_;
    Publickey<S>* pk_;
    hegpu_rng* rng_ = nullptr;
};

template <Scheme S> class HEDecryptor { // host/ckks/decryptor.cuh
  public:
    HEDecryptor(HEContext<S> context, Secretkey<S>& sk) : context_(std::move(context)), sk_(&sk)
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
    }
    void decrypt(Plaintext<S>& pt, Ciphertext<S>& ct, const ExecutionOptions& o = ExecutionOptions())
    {
        if (ct.size() != 2) throw std::invalid_argument("Ciphertext should be relinearized first!");
        const int l = context_->Q_size - ct.depth();
        DeviceVector<Data64> out((S == Scheme::CKKS) ? (size_t) l * context_->n : (size_t) context_->n, o.stream_);
        if (S == Scheme::CKKS) {
            detail::check(hegpu_ckks_decrypt(context_->handle(), (const uint64_t*) ct.data(),
                                             (const uint64_t*) sk_->data(), ct.depth(), (uint64_t*) out.data(),
                                             o.stream_));
        } else {
            DeviceVector<Data64> ws((hegpu_workspace_bytes(context_->handle(), HEGPU_OP_BFV_DECRYPT, 0, 1) + 7) / 8,
                                    o.stream_);
            detail::check(hegpu_bfv_decrypt(context_->handle(), (const uint64_t*) ct.data(),
                                            (const uint64_t*) sk_->data(), (uint64_t*) out.data(), ws.data(),
                                            ws.size() * sizeof(Data64), o.stream_));
        }
        pt.memory_set(std::move(out));
        pt.depth_ = ct.depth();
        pt.scale_ = ct.scale();
        pt.plaintext_generated_ = true;
    }

  private:
    HEContext<S> context_;
    Secretkey<S>* sk_;
};

// ------------------------------------------------------------------ encoders
template <Scheme S> class HEEncoder;

template <> class HEEncoder<Scheme::BFV> { // host/bfv/encoder.cuh: batching over the slots of Z_t[X]/(X^N+1)
    static constexpr Scheme S = Scheme::BFV;

  public:
    explicit HEEncoder(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
    }
    inline int slot_count() const noexcept { return context_->n; }

    void encode(Plaintext<S>& plain, const std::vector<int64_t>& message, const ExecutionOptions& o = ExecutionOptions())
    {
        if ((int) message.size() > context_->n)
            throw std::invalid_argument("Message size can not be higher than the slot count."); // bfv/encoder.cuh:60
        DeviceVector<Data64> msg(message.size() ? message.size() : 1, o.stream_);
        if (!message.empty())
            detail::hip(hipMemcpyAsync(msg.data(), message.data(), message.size() * sizeof(int64_t),
                                       hipMemcpyHostToDevice, o.stream_));
        DeviceVector<Data64> out((size_t) context_->n, o.stream_);
        detail::check(hegpu_bfv_encode(context_->handle(), (const int64_t*) msg.data(), (int) message.size(),
                                       (uint64_t*) out.data(), o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_)); // `message` may be a temporary
        plain.memory_set(std::move(out));
        plain.depth_ = 0;
        plain.plaintext_generated_ = true;
    }
    void encode(Plaintext<S>& plain, const std::vector<uint64_t>& message, const ExecutionOptions& o = ExecutionOptions())
    {
        std::vector<int64_t> m(message.begin(), message.end());
        encode(plain, m, o);
    }
    void decode(std::vector<uint64_t>& message, Plaintext<S>& plain, const ExecutionOptions& o = ExecutionOptions())
    {
        DeviceVector<Data64> out((size_t) context_->n, o.stream_), ws((size_t) context_->n, o.stream_);
        detail::check(hegpu_bfv_decode(context_->handle(), (const uint64_t*) plain.data(), (uint64_t*) out.data(),
                                       ws.data(), ws.size() * sizeof(Data64), o.stream_));
        message.resize(context_->n);
        detail::hip(hipMemcpyAsync(message.data(), out.data(), message.size() * sizeof(uint64_t),
                                   hipMemcpyDeviceToHost, o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_));
    }
    void decode(std::vector<int64_t>& message, Plaintext<S>& plain, const ExecutionOptions& o = ExecutionOptions())
    {
        std::vector<uint64_t> u;
        decode(u, plain, o);
        const uint64_t t = context_->get_plain_modulus();
        message.resize(u.size());
        for (size_t i = 0; i < u.size(); i++) // centred representative
            message[i] = u[i] > (t >> 1) ? (int64_t) u[i] - (int64_t) t : (int64_t) u[i];
    }

  private:
    HEContext<S> context_;
};

template <> class HEEncoder<Scheme::CKKS> { // host/ckks/encoder.cuh: N/2 complex slots, real vectors here
    static constexpr Scheme S = Scheme::CKKS;

  public:
    explicit HEEncoder(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
    }
    inline int slot_count() const noexcept { return context_->n >> 1; }

    void encode(Plaintext<S>& plain, const std::vector<double>& message, double scale,
                const ExecutionOptions& o = ExecutionOptions())
    {
        if ((int) message.size() > slot_count())
            throw std::invalid_argument("Vector size can not be higher than slot count!"); // ckks/encoder.cuh:74
        DeviceVector<Data64> msg(message.size() ? message.size() : 1, o.stream_);
        if (!message.empty())
            detail::hip(hipMemcpyAsync(msg.data(), message.data(), message.size() * sizeof(double),
                                       hipMemcpyHostToDevice, o.stream_));
        DeviceVector<Data64> out((size_t) context_->Q_size * context_->n, o.stream_);
        DeviceVector<Data64> ws((hegpu_workspace_bytes(context_->handle(), HEGPU_OP_CKKS_ENCODE, 0, 1) + 7) / 8,
                                o.stream_);
        detail::check(hegpu_ckks_encode(context_->handle(), (const double*) msg.data(), (int) message.size(), scale,
                                        (uint64_t*) out.data(), ws.data(), ws.size() * sizeof(Data64), o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_));
        plain.memory_set(std::move(out));
        plain.depth_ = 0;
        plain.scale_ = scale;
        plain.plaintext_generated_ = true;
    }
    void decode(std::vector<double>& message, Plaintext<S>& plain, const ExecutionOptions& o = ExecutionOptions())
    {
        DeviceVector<Data64> out((size_t) slot_count(), o.stream_);
        DeviceVector<Data64> ws(
            (hegpu_workspace_bytes(context_->handle(), HEGPU_OP_CKKS_DECODE, plain.depth_, 1) + 7) / 8, o.stream_);
        detail::check(hegpu_ckks_decode(context_->handle(), (const uint64_t*) plain.data(), plain.depth_, plain.scale_,
                                        (double*) out.data(), ws.data(), ws.size() * sizeof(Data64), o.stream_));
        message.resize(slot_count());
        detail::hip(hipMemcpyAsync(message.data(), out.data(), message.size() * sizeof(double), hipMemcpyDeviceToHost,
                                   o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_));
    }

  private:
    HEContext<S> context_;
};

// ------------------------------------------------------------------ operator
template <Scheme S> class HEArithmeticOperator { // host/{bfv,ckks}/operator.cuh
  public:
    explicit HEArithmeticOperator(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_ || !context_->context_generated_) throw std::invalid_argument("HEContext is not generated!");
    }
    // the reference's constructor takes the encoder (operator.cuh: used by its bootstrapping code only)
    HEArithmeticOperator(HEContext<S> context, HEEncoder<S>&) : HEArithmeticOperator(std::move(context)) {}

    void add(Ciphertext<S>& a, Ciphertext<S>& b, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        binary(a, b, out, 0, o);
    }
    void sub(Ciphertext<S>& a, Ciphertext<S>& b, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        binary(a, b, out, 1, o);
    }
    void negate(Ciphertext<S>& a, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        const int l = limbs(a);
        DeviceVector<Data64> m((size_t) a.cipher_size_ * l * context_->n, o.stream_);
        detail::check(hegpu_addition(context_->handle(), (const uint64_t*) a.data(), nullptr, (uint64_t*) m.data(), l,
                                     a.cipher_size_, 1, 2, o.stream_));
        copy_meta(a, out);
        out.memory_set(std::move(m));
    }

    // host/ckks/operator.cuh:632-689, host/bfv/operator.cuh:348-391
    void multiply(Ciphertext<S>& a, Ciphertext<S>& b, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        if (a.relinearization_required_ || b.relinearization_required_)
            throw std::invalid_argument("Ciphertexts can not be multiplied because of the non-linear part! Please "
                                        "use relinearization operation!");
        if (a.rescale_required_ || b.rescale_required_)
            throw std::invalid_argument("Ciphertexts can not be multiplied because of the noise! Please use rescale "
                                        "operation to get rid of additional noise!");
        if (a.depth_ != b.depth_) throw std::logic_error("Ciphertexts leveled are not equal");
        const int l = limbs(a);
        const size_t n = context_->n;
        if (a.memory_size() < 2 * n * l || b.memory_size() < 2 * n * l)
            throw std::invalid_argument("Invalid Ciphertexts size!");
        DeviceVector<Data64> m(3 * n * l, o.stream_);
        if (S == Scheme::CKKS) {
            detail::check(hegpu_ckks_multiply(context_->handle(), (const uint64_t*) a.data(), 0,
                                              (const uint64_t*) b.data(), 0, (uint64_t*) m.data(), 0, a.depth_, 1,
                                              o.stream_));
        } else {
            const size_t wsb = hegpu_workspace_bytes(context_->handle(), HEGPU_OP_BFV_MULTIPLY, 0, 1);
            DeviceVector<Data64> ws(wsb / 8, o.stream_);
            detail::check(hegpu_bfv_multiply(context_->handle(), (const uint64_t*) a.data(), 0,
                                             (const uint64_t*) b.data(), 0, (uint64_t*) m.data(), 0, 1, ws.data(), wsb,
                                             o.stream_));
        }
        const double sc = a.scale_ * b.scale_;
        copy_meta(a, out);
        out.memory_set(std::move(m));
        out.cipher_size_ = 3;
        out.scale_ = sc;
        out.relinearization_required_ = true;
        out.rescale_required_ = (S == Scheme::CKKS);
    }
    void multiply_inplace(Ciphertext<S>& a, Ciphertext<S>& b, const ExecutionOptions& o = ExecutionOptions())
    {
        multiply(a, b, a, o);
    }

    // host/ckks/operator.cuh:1053-1094: method I or II by the context's P_size
    void relinearize_inplace(Ciphertext<S>& a, Relinkey<S>& rk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!a.relinearization_required_)
            throw std::invalid_argument("Ciphertexts can not use relinearization, since no non-linear part!");
        const int l = limbs(a);
        const int op = (S == Scheme::CKKS) ? HEGPU_OP_CKKS_RELIN : HEGPU_OP_BFV_RELIN;
        const size_t wsb = hegpu_workspace_bytes(context_->handle(), op, a.depth_, 1);
        DeviceVector<Data64> ws(wsb / 8, o.stream_);
        if (S == Scheme::CKKS)
            detail::check(hegpu_ckks_relinearize_inplace(context_->handle(), (uint64_t*) a.data(),
                                                         (uint64_t) 3 * l * context_->n, (const uint64_t*) rk.data(),
                                                         a.depth_, 1, ws.data(), wsb, o.stream_));
        else
            detail::check(hegpu_bfv_relinearize_inplace(context_->handle(), (uint64_t*) a.data(),
                                                        (uint64_t) 3 * l * context_->n, (const uint64_t*) rk.data(), 1,
                                                        ws.data(), wsb, o.stream_));
        a.relinearization_required_ = false;
        a.cipher_size_ = 2;
    }

    // host/ckks/operator.cuh:1423-1445 (CKKS only)
    void rescale_inplace(Ciphertext<S>& a, const ExecutionOptions& o = ExecutionOptions())
    {
        static_assert(S == Scheme::CKKS, "rescale is a CKKS operation");
        if (!a.rescale_required_ || a.relinearization_required_)
            throw std::invalid_argument("Ciphertexts can not be rescaled because ciphertext rescaling is not required "
                                        "or relinearization is required first!");
        const int l = limbs(a);
        if (l < 2) throw std::logic_error("Ciphertext modulus can not be reducible, since there is only one modulus");
        const size_t wsb = hegpu_workspace_bytes(context_->handle(), HEGPU_OP_CKKS_RESCALE, a.depth_, 1);
        DeviceVector<Data64> ws(wsb / 8, o.stream_);
        detail::check(hegpu_ckks_rescale_inplace(context_->handle(), (uint64_t*) a.data(),
                                                 (uint64_t) 2 * l * context_->n, a.depth_, 1, ws.data(), wsb,
                                                 o.stream_));
        a.scale_ = a.scale_ / (double) context_->prime_vector_[l - 1]; // ckks/operator.cu:1235-1241
        a.depth_++;
        a.rescale_required_ = false;
    }

    // host/bfv/operator.cuh:576-660, ckks/operator.cu:1338-1378: direct key or power-of-two chain
    void rotate_rows(Ciphertext<S>& in, Ciphertext<S>& out, Galoiskey<S>& gk, int shift,
                     const ExecutionOptions& o = ExecutionOptions())
    {
        if (shift == 0) { out = in; return; }
        const int g = hegpu_steps_to_galois_elt(shift, context_->n, gk.group_order_);
        if (gk.device_location_.count(g)) { apply_galois(in, out, gk, g, o); return; }
        std::vector<int> chain; // bfv/operator.cu:692-712
        int rest = std::abs(shift);
        const int sign = shift < 0 ? -1 : 1;
        while (rest) {
            const int p2 = 1 << (int) std::log2((double) rest);
            rest -= p2;
            auto it = gk.galois_elt.find(p2 * sign);
            if (it == gk.galois_elt.end() || !gk.device_location_.count(it->second))
                throw std::logic_error("Galois key not present!");
            chain.push_back(it->second);
        }
        Ciphertext<S> cur = in;
        for (int ge : chain) {
            Ciphertext<S> nxt(cur);
            apply_galois(cur, nxt, gk, ge, o);
            cur = std::move(nxt);
        }
        out = std::move(cur);
    }
    void rotate_rows_inplace(Ciphertext<S>& a, Galoiskey<S>& gk, int shift, const ExecutionOptions& o = ExecutionOptions())
    {
        Ciphertext<S> tmp(a);
        rotate_rows(tmp, a, gk, shift, o);
    }
    void apply_galois(Ciphertext<S>& in, Ciphertext<S>& out, Galoiskey<S>& gk, int galois_elt,
                      const ExecutionOptions& o = ExecutionOptions())
    {
        auto it = gk.device_location_.find(galois_elt);
        if (it == gk.device_location_.end()) throw std::logic_error("Galois key not present!");
        if (in.relinearization_required_) throw std::invalid_argument("Ciphertext should be relinearized first!");
        const int l = limbs(in);
        const size_t n = context_->n;
        DeviceVector<Data64> m(2 * n * l, o.stream_);
        const int op = (S == Scheme::CKKS) ? HEGPU_OP_CKKS_GALOIS : HEGPU_OP_BFV_GALOIS;
        const size_t wsb = hegpu_workspace_bytes(context_->handle(), op, in.depth_, 1);
        DeviceVector<Data64> ws(wsb / 8, o.stream_);
        if (S == Scheme::CKKS)
            detail::check(hegpu_ckks_apply_galois(context_->handle(), (const uint64_t*) in.data(), 0,
                                                  (uint64_t*) m.data(), 0, (const uint64_t*) it->second.data(),
                                                  galois_elt, in.depth_, 1, ws.data(), wsb, o.stream_));
        else
            detail::check(hegpu_bfv_apply_galois(context_->handle(), (const uint64_t*) in.data(), 0,
                                                 (uint64_t*) m.data(), 0, (const uint64_t*) it->second.data(),
                                                 galois_elt, 1, ws.data(), wsb, o.stream_));
        if (&in != &out) copy_meta(in, out);
        out.memory_set(std::move(m));
    }

    // ---- ciphertext (+,-,*) plaintext (host/*/operator.cuh add_plain / sub_plain / multiply_plain)
    void add_plain(Ciphertext<S>& a, Plaintext<S>& p, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        plain_addsub(a, p, out, 0, o);
    }
    void sub_plain(Ciphertext<S>& a, Plaintext<S>& p, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        plain_addsub(a, p, out, 1, o);
    }
    void add_plain_inplace(Ciphertext<S>& a, Plaintext<S>& p, const ExecutionOptions& o = ExecutionOptions())
    {
        plain_addsub(a, p, a, 0, o);
    }
    void sub_plain_inplace(Ciphertext<S>& a, Plaintext<S>& p, const ExecutionOptions& o = ExecutionOptions())
    {
        plain_addsub(a, p, a, 1, o);
    }
    void multiply_plain(Ciphertext<S>& a, Plaintext<S>& p, Ciphertext<S>& out,
                        const ExecutionOptions& o = ExecutionOptions())
    {
        if (a.relinearization_required_) throw std::invalid_argument("Ciphertext should be relinearized first!");
        const int l = limbs(a);
        const size_t n = context_->n;
        DeviceVector<Data64> m(2 * n * l, o.stream_);
        if (S == Scheme::CKKS) {
            if (a.rescale_required_) throw std::invalid_argument("Ciphertext should be rescaled first!");
            if (p.depth_ != a.depth_) throw std::logic_error("Ciphertext and Plaintext levels are not equal");
            detail::check(hegpu_cipherplain_multiplication(context_->handle(), (const uint64_t*) a.data(),
                                                           (const uint64_t*) p.data(), (uint64_t*) m.data(), l,
                                                           o.stream_));
        } else {
            const size_t wsb = hegpu_workspace_bytes(context_->handle(), HEGPU_OP_BFV_MULTIPLY_PLAIN, 0, 1);
            DeviceVector<Data64> ws(wsb / 8, o.stream_);
            detail::check(hegpu_bfv_multiply_plain(context_->handle(), (const uint64_t*) a.data(),
                                                   (const uint64_t*) p.data(), (uint64_t*) m.data(), ws.data(), wsb,
                                                   o.stream_));
        }
        const double ps = p.scale_;
        if (&a != &out) copy_meta(a, out);
        out.memory_set(std::move(m));
        if (S == Scheme::CKKS) {
            out.scale_ = a.scale_ * ps; // ckks/operator.cuh multiply_plain
            out.rescale_required_ = true;
        }
    }
    void multiply_plain_inplace(Ciphertext<S>& a, Plaintext<S>& p, const ExecutionOptions& o = ExecutionOptions())
    {
        multiply_plain(a, p, a, o);
    }
    // BFV: swap the two rows of the slot matrix (Galois element 2N - 1, bfv/operator.cu:975-1068)
    void rotate_columns(Ciphertext<S>& in, Ciphertext<S>& out, Galoiskey<S>& gk,
                        const ExecutionOptions& o = ExecutionOptions())
    {
        apply_galois(in, out, gk, gk.galois_elt_zero, o);
    }

  private:
    int limbs(const Ciphertext<S>& a) const { return context_->Q_size - a.depth_; }
    static void copy_meta(const Ciphertext<S>& a, Ciphertext<S>& out)
    {
        out.ring_size_ = a.ring_size_;
        out.coeff_modulus_count_ = a.coeff_modulus_count_;
        out.cipher_size_ = a.cipher_size_;
        out.depth_ = a.depth_;
        out.scale_ = a.scale_;
        out.in_ntt_domain_ = a.in_ntt_domain_;
        out.rescale_required_ = a.rescale_required_;
        out.relinearization_required_ = a.relinearization_required_;
        out.ciphertext_generated_ = true;
    }
    void binary(Ciphertext<S>& a, Ciphertext<S>& b, Ciphertext<S>& out, int op, const ExecutionOptions& o)
    {
        if (a.depth_ != b.depth_) throw std::logic_error("Ciphertexts leveled are not equal");
        if (a.cipher_size_ != b.cipher_size_) throw std::invalid_argument("Ciphertexts sizes have to be equal");
        const int l = limbs(a);
        DeviceVector<Data64> m((size_t) a.cipher_size_ * l * context_->n, o.stream_);
        detail::check(hegpu_addition(context_->handle(), (const uint64_t*) a.data(), (const uint64_t*) b.data(),
                                     (uint64_t*) m.data(), l, a.cipher_size_, 1, op, o.stream_));
        copy_meta(a, out);
        out.memory_set(std::move(m));
    }
    void plain_addsub(Ciphertext<S>& a, Plaintext<S>& p, Ciphertext<S>& out, int sub, const ExecutionOptions& o)
    {
        if (a.relinearization_required_) throw std::invalid_argument("Ciphertext should be relinearized first!");
        const int l = limbs(a);
        const size_t n = context_->n;
        DeviceVector<Data64> m(2 * n * l, o.stream_);
        if (S == Scheme::CKKS) {
            if (p.depth_ != a.depth_) throw std::logic_error("Ciphertext and Plaintext levels are not equal");
            // part 0 +- plaintext, part 1 unchanged (addition.cu: addition_plain_ckks_poly)
            detail::check(hegpu_addition(context_->handle(), (const uint64_t*) a.data(), (const uint64_t*) p.data(),
                                         (uint64_t*) m.data(), l, 1, 1, sub, o.stream_));
            detail::hip(hipMemcpyAsync(m.data() + n * l, a.data() + n * l, n * l * sizeof(Data64),
                                       hipMemcpyDeviceToDevice, o.stream_));
        } else {
            detail::check(hegpu_bfv_plain_addsub(context_->handle(), (const uint64_t*) a.data(),
                                                 (const uint64_t*) p.data(), (uint64_t*) m.data(), sub, o.stream_));
        }
        if (&a != &out) copy_meta(a, out);
        out.memory_set(std::move(m));
    }
    HEContext<S> context_;
};

// ------------------------------------------------------------------ TFHE (host/tfhe/*.cuh)
// Fixed parameter set (tfhe/context.cu:15-57).  Ciphertexts are LWE samples a [shape][n], b [shape]
// on the 32-bit torus; a bit is +-1/8.
template <> class HEContextImpl<Scheme::TFHE> {
  public:
    explicit HEContextImpl(sec_level_type = sec_level_type::sec128)
    {
        detail::check(hegpu_tfhe_context_create(&h_));
        n_ = (int) hegpu_tfhe_context_int(h_, "n");
        N_ = (int) hegpu_tfhe_context_int(h_, "N");
        k_ = (int) hegpu_tfhe_context_int(h_, "k");
    }
    ~HEContextImpl() { if (h_) hegpu_tfhe_context_destroy(h_); }
    HEContextImpl(const HEContextImpl&) = delete;
    hegpu_tfhe_context* handle() const { return h_; }
    long elems(const char* what) const { return hegpu_tfhe_context_int(h_, what); }
    int n_ = 0, N_ = 0, k_ = 0;
    bool context_generated_ = true;

  private:
    hegpu_tfhe_context* h_ = nullptr;
};

template <> class Ciphertext<Scheme::TFHE> {
  public:
    explicit Ciphertext(HEContext<Scheme::TFHE> context, const ExecutionOptions& = ExecutionOptions())
    {
        if (!context) throw std::invalid_argument("HEContext is not generated!");
        n_ = context->n_;
    }
    DeviceVector<int32_t> a_device_location_, b_device_location_;
    int n_ = 0, shape_ = 0;
    bool ciphertext_generated_ = false;
};

template <> class Secretkey<Scheme::TFHE> {
  public:
    explicit Secretkey(HEContext<Scheme::TFHE> context) : context_(std::move(context))
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
    }
    DeviceVector<int32_t> lwe_key_device_location_, tlwe_key_device_location_;
    bool secret_key_generated_ = false;

  private:
    HEContext<Scheme::TFHE> context_;
};

template <Scheme S> class Bootstrappingkey;
template <> class Bootstrappingkey<Scheme::TFHE> { // boot key (prepared for the blind rotate) + key-switch key
  public:
    explicit Bootstrappingkey(HEContext<Scheme::TFHE> context) : context_(std::move(context))
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
    }
    DeviceVector<Data64> boot_key_device_location_;  // reference layout [n][k+1][l][k+1][N], NTT domain
    DeviceVector<Data64> prepared_;                  // hegpu_tfhe_prepare_bootkey
    DeviceVector<int32_t> switch_key_device_location_a_, switch_key_device_location_b_;
    bool boot_key_generated_ = false;

  private:
    HEContext<Scheme::TFHE> context_;
};

template <> class HEKeyGenerator<Scheme::TFHE> { // host/tfhe/keygenerator.cuh
    static constexpr Scheme S = Scheme::TFHE;

  public:
    explicit HEKeyGenerator(HEContext<S> context) : HEKeyGenerator(std::move(context), std::random_device{}()) {}
    HEKeyGenerator(HEContext<S> context, std::uint64_t seed) : context_(std::move(context))
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
        detail::check(hegpu_rng_create(seed, &rng_));
    }
    ~HEKeyGenerator() { hegpu_rng_destroy(rng_); }
    HEKeyGenerator(const HEKeyGenerator&) = delete;
    HEKeyGenerator& operator=(const HEKeyGenerator&) = delete;

    void generate_secret_key(Secretkey<S>& sk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (sk.secret_key_generated_) throw std::logic_error("Secretkey is already generated!");
        sk.lwe_key_device_location_ = DeviceVector<int32_t>((size_t) context_->n_, o.stream_);
        sk.tlwe_key_device_location_ = DeviceVector<int32_t>((size_t) context_->k_ * context_->N_, o.stream_);
        detail::check(hegpu_tfhe_generate_secret_key(context_->handle(), rng_, sk.lwe_key_device_location_.data(),
                                                     sk.tlwe_key_device_location_.data(), o.stream_));
        sk.secret_key_generated_ = true;
    }
    void generate_bootstrapping_key(Bootstrappingkey<S>& bk, Secretkey<S>& sk,
                                    const ExecutionOptions& o = ExecutionOptions())
    {
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
        if (bk.boot_key_generated_) throw std::logic_error("Bootstrappingkey is already generated!");
        bk.boot_key_device_location_ = DeviceVector<Data64>((size_t) context_->elems("bootkey_elems"), o.stream_);
        bk.switch_key_device_location_a_ = DeviceVector<int32_t>((size_t) context_->elems("kskey_a_elems"), o.stream_);
        bk.switch_key_device_location_b_ = DeviceVector<int32_t>((size_t) context_->elems("kskey_b_elems"), o.stream_);
        DeviceVector<Data64> ws((size_t) context_->N_, o.stream_);
        detail::check(hegpu_tfhe_generate_bootstrapping_key(
            context_->handle(), rng_, sk.lwe_key_device_location_.data(), sk.tlwe_key_device_location_.data(),
            (uint64_t*) bk.boot_key_device_location_.data(), bk.switch_key_device_location_a_.data(),
            bk.switch_key_device_location_b_.data(), ws.data(), ws.size() * sizeof(Data64), o.stream_));
        bk.prepared_ = DeviceVector<Data64>((size_t) context_->elems("prepared_bootkey_elems"), o.stream_);
        detail::check(hegpu_tfhe_prepare_bootkey(context_->handle(), (const uint64_t*) bk.boot_key_device_location_.data(),
                                                 (uint64_t*) bk.prepared_.data(), o.stream_));
        bk.boot_key_generated_ = true;
    }

  private:
    HEContext<S> context_;
    hegpu_rng* rng_ = nullptr;
};

template <> class HEEncryptor<Scheme::TFHE> { // host/tfhe/encryptor.cuh: symmetric LWE encryption of bits
    static constexpr Scheme S = Scheme::TFHE;

  public:
    HEEncryptor(HEContext<S> context, Secretkey<S>& sk) : context_(std::move(context)), sk_(&sk)
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
        detail::check(hegpu_rng_create(std::random_device{}(), &rng_));
    }
    ~HEEncryptor() { hegpu_rng_destroy(rng_); }
    HEEncryptor(const HEEncryptor&) = delete;
    HEEncryptor& operator=(const HEEncryptor&) = delete;

    void encrypt(Ciphertext<S>& ct, const std::vector<bool>& messages, const ExecutionOptions& o = ExecutionOptions())
    {
        const int shape = (int) messages.size();
        std::vector<int32_t> enc(messages.size());
        for (size_t i = 0; i < messages.size(); i++) enc[i] = messages[i] ? (1 << 29) : -(1 << 29); // +-1/8
        DeviceVector<int32_t> msg(enc, o.stream_);
        ct.a_device_location_ = DeviceVector<int32_t>((size_t) shape * context_->n_, o.stream_);
        ct.b_device_location_ = DeviceVector<int32_t>((size_t) shape, o.stream_);
        detail::check(hegpu_tfhe_encrypt(context_->handle(), rng_, sk_->lwe_key_device_location_.data(), msg.data(),
                                         shape, ct.a_device_location_.data(), ct.b_device_location_.data(), o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_));
        ct.shape_ = shape;
        ct.ciphertext_generated_ = true;
    }

  private:
    HEContext<S> context_;
    Secretkey<S>* sk_;
    hegpu_rng* rng_ = nullptr;
};

template <> class HEDecryptor<Scheme::TFHE> { // host/tfhe/decryptor.cuh
    static constexpr Scheme S = Scheme::TFHE;

  public:
    HEDecryptor(HEContext<S> context, Secretkey<S>& sk) : context_(std::move(context)), sk_(&sk)
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
        if (!sk.secret_key_generated_) throw std::logic_error("Secretkey is not generated!");
    }
    void decrypt(Ciphertext<S>& ct, std::vector<bool>& messages, const ExecutionOptions& o = ExecutionOptions())
    {
        DeviceVector<int32_t> phase((size_t) ct.shape_, o.stream_);
        detail::check(hegpu_tfhe_decrypt_phase(context_->handle(), sk_->lwe_key_device_location_.data(),
                                               ct.a_device_location_.data(), ct.b_device_location_.data(), ct.shape_,
                                               phase.data(), o.stream_));
        std::vector<int32_t> h((size_t) ct.shape_);
        detail::hip(hipMemcpyAsync(h.data(), phase.data(), h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, o.stream_));
        detail::hip(hipStreamSynchronize(o.stream_));
        messages.resize(h.size());
        for (size_t i = 0; i < h.size(); i++) messages[i] = h[i] > 0;
    }

  private:
    HEContext<S> context_;
    Secretkey<S>* sk_;
};

template <Scheme S> class HELogicOperator;
template <> class HELogicOperator<Scheme::TFHE> { // host/tfhe/operator.cuh: bootstrapped binary gates
    static constexpr Scheme S = Scheme::TFHE;

  public:
    explicit HELogicOperator(HEContext<S> context) : context_(std::move(context))
    {
        if (!context_) throw std::invalid_argument("HEContext is not generated!");
    }
#define HEONGPU_TFHE_GATE(NAME, ID)                                                                              \
    void NAME(Ciphertext<S>& in1, Ciphertext<S>& in2, Ciphertext<S>& out, Bootstrappingkey<S>& bk,               \
              const ExecutionOptions& o = ExecutionOptions())                                                    \
    {                                                                                                            \
        gate(ID, in1, in2, out, bk, o);                                                                          \
    }
    HEONGPU_TFHE_GATE(NAND, HEGPU_GATE_NAND)
    HEONGPU_TFHE_GATE(AND, HEGPU_GATE_AND)
    HEONGPU_TFHE_GATE(NOR, HEGPU_GATE_NOR)
    HEONGPU_TFHE_GATE(OR, HEGPU_GATE_OR)
    HEONGPU_TFHE_GATE(XNOR, HEGPU_GATE_XNOR)
    HEONGPU_TFHE_GATE(XOR, HEGPU_GATE_XOR)
#undef HEONGPU_TFHE_GATE
    void NOT(Ciphertext<S>& in, Ciphertext<S>& out, const ExecutionOptions& o = ExecutionOptions())
    {
        if (!in.ciphertext_generated_) throw std::runtime_error("Input is not generated!");
        DeviceVector<int32_t> a((size_t) in.shape_ * context_->n_, o.stream_), b((size_t) in.shape_, o.stream_);
        detail::check(hegpu_tfhe_gate_precompute(context_->handle(), HEGPU_GATE_NOT, a.data(), b.data(),
                                                 in.a_device_location_.data(), in.b_device_location_.data(), nullptr,
                                                 nullptr, in.shape_, o.stream_));
        finish(out, in.shape_, std::move(a), std::move(b));
    }
    void MUX(Ciphertext<S>& in1, Ciphertext<S>& in2, Ciphertext<S>& control, Ciphertext<S>& out,
             Bootstrappingkey<S>& bk, const ExecutionOptions& o = ExecutionOptions())
    {
        if (in1.shape_ != in2.shape_ || in1.shape_ != control.shape_)
            throw std::runtime_error("Ciphertexts size should be equal!");
        if (!(in1.ciphertext_generated_ && in2.ciphertext_generated_)) throw std::runtime_error("One or the inputs are generated!");
        const int shape = in1.shape_;
        DeviceVector<int32_t> a((size_t) shape * context_->n_, o.stream_), b((size_t) shape, o.stream_);
        DeviceVector<int32_t> ws(((size_t) context_->n_ + 1 + 2 * ((size_t) context_->k_ * context_->N_ + 1)) * shape,
                                 o.stream_);
        detail::check(hegpu_tfhe_mux(context_->handle(), in1.a_device_location_.data(), in1.b_device_location_.data(),
                                     in2.a_device_location_.data(), in2.b_device_location_.data(),
                                     control.a_device_location_.data(), control.b_device_location_.data(), a.data(),
                                     b.data(), (const uint64_t*) bk.prepared_.data(),
                                     bk.switch_key_device_location_a_.data(), bk.switch_key_device_location_b_.data(),
                                     shape, ws.data(), ws.size() * sizeof(int32_t), o.stream_));
        finish(out, shape, std::move(a), std::move(b));
    }

  private:
    void gate(int id, Ciphertext<S>& in1, Ciphertext<S>& in2, Ciphertext<S>& out, Bootstrappingkey<S>& bk,
              const ExecutionOptions& o)
    {
        if (in1.shape_ != in2.shape_) throw std::runtime_error("Ciphertexts size should be equal!");
        if (!(in1.ciphertext_generated_ && in2.ciphertext_generated_)) throw std::runtime_error("One or the inputs are generated!");
        if (!bk.boot_key_generated_) throw std::runtime_error("Bootstrappingkey is not generated!");
        const int shape = in1.shape_;
        DeviceVector<int32_t> a((size_t) shape * context_->n_, o.stream_), b((size_t) shape, o.stream_);
        DeviceVector<int32_t> ws(((size_t) context_->n_ + (size_t) context_->k_ * context_->N_ + 2) * shape, o.stream_);
        detail::check(hegpu_tfhe_gate(context_->handle(), id, in1.a_device_location_.data(),
                                      in1.b_device_location_.data(), in2.a_device_location_.data(),
                                      in2.b_device_location_.data(), a.data(), b.data(),
                                      (const uint64_t*) bk.prepared_.data(), bk.switch_key_device_location_a_.data(),
                                      bk.switch_key_device_location_b_.data(), shape, ws.data(),
                                      ws.size() * sizeof(int32_t), o.stream_));
        finish(out, shape, std::move(a), std::move(b));
    }
    void finish(Ciphertext<S>& out, int shape, DeviceVector<int32_t>&& a, DeviceVector<int32_t>&& b)
    {
        out.a_device_location_ = std::move(a);
        out.b_device_location_ = std::move(b);
        out.shape_ = shape;
        out.n_ = context_->n_;
        out.ciphertext_generated_ = true;
    }
    HEContext<S> context_;
};

// ------------------------------------------------------------------ serializer (util/serializer.h:20-131)
// file = u64 size + zlib stream of the object's save() bytes
namespace serializer {
inline std::vector<std::uint8_t> to_buffer(const std::stringstream& ss)
{
    const std::string str = ss.str();
    return {str.begin(), str.end()};
}
inline void from_buffer(std::stringstream& ss, const std::vector<std::uint8_t>& buffer)
{
    ss.str(std::string(buffer.begin(), buffer.end()));
}
#if defined(HEONGPU_WITH_ZLIB)
inline std::vector<std::uint8_t> compress(const std::vector<std::uint8_t>& data)
{
    uLongf bound = compressBound(data.size());
    std::vector<std::uint8_t> out(bound);
    if (::compress(out.data(), &bound, data.data(), data.size()) != Z_OK)
        throw std::runtime_error("Zlib compression failed");
    out.resize(bound);
    return out;
}
inline std::vector<std::uint8_t> decompress(const std::vector<std::uint8_t>& data)
{
    // the reference sizes the output at 4x the input and fails beyond; grow instead
    for (size_t factor = 4; factor <= 4096; factor *= 4) {
        std::vector<std::uint8_t> out(data.size() * factor + 64);
        uLongf out_size = out.size();
        const int rc = ::uncompress(out.data(), &out_size, data.data(), data.size());
        if (rc == Z_OK) {
            out.resize(out_size);
            return out;
        }
        if (rc != Z_BUF_ERROR) break;
    }
    throw std::runtime_error("Zlib decompression failed");
}
template <typename T> std::vector<std::uint8_t> serialize(const T& obj)
{
    std::stringstream ss;
    obj.save(ss);
    return compress(to_buffer(ss));
}
// the object is constructed by the caller (it needs its context)
template <typename T> void deserialize(T& obj, const std::vector<std::uint8_t>& buffer)
{
    std::stringstream ss;
    from_buffer(ss, decompress(buffer));
    obj.load(ss);
}
template <typename T> void save_to_file(const T& obj, const std::string& filename)
{
    const std::vector<std::uint8_t> data = serialize(obj);
    const std::uint64_t size = data.size();
    std::ofstream ofs(filename, std::ios::binary);
    if (!ofs) throw std::runtime_error("Cannot open file for writing: " + filename);
    ofs.write((const char*) &size, sizeof(size));
    ofs.write((const char*) data.data(), (std::streamsize) size);
}
template <typename T> void load_from_file(T& obj, const std::string& filename)
{
    std::ifstream ifs(filename, std::ios::binary);
    if (!ifs) throw std::runtime_error("Cannot open file for reading: " + filename);
    std::uint64_t size = 0;
    ifs.read((char*) &size, sizeof(size));
    std::vector<std::uint8_t> buffer(size);
    ifs.read((char*) buffer.data(), (std::streamsize) size);
    if (!ifs) throw std::runtime_error("File is truncated: " + filename);
    deserialize(obj, buffer);
}
#endif // HEONGPU_WITH_ZLIB
} // namespace serializer

} // namespace heongpu

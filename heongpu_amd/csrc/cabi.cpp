// cabi.cpp -- extern "C" entry points declared in include/hegpu.h.
#include "../../include/hegpu.h"
#include "../../include/hegpu_bench.h"
#include "context.hpp"
#include "host_params.hpp"
#include "ops.hpp"
#include <cmath>
#include "tfhe.hpp"
#include <algorithm>
#include <cerrno>
#include <climits>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cstring>
#include <sys/random.h>
#include <new>
#include <stdexcept>
#include <string>

using namespace hegpu;

struct hegpu_context {
    Context c;
};

static thread_local std::string g_err;

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

// Every device entry point runs on the device its context was uploaded to, whatever the calling thread's current
// device is (one process driving several GPUs: one context per device, hegpu_context_upload_device); the thread's
// current device is restored on return.
struct DevGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess; // a failed switch is reported by the entry point (NEED_CTX / TFHE_NEED), never ignored:
                                 // the call would otherwise launch on the caller's device with another device's pointers
    explicit DevGuard(int dev)
    {
        if (dev < 0) return;
        if ((err = hipGetDevice(&prev)) != hipSuccess) return;
        if (prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DevGuard()
    {
        if (switched) (void) hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};
static int hip_ret(hipError_t e, const char* where)
{
    if (e == hipSuccess) return 0;
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (int) e;
}

template <typename F>
static int guarded(F&& f)
{
    try {
        return f();
    } catch (const std::invalid_argument& e) {
        return fail(HEGPU_E_INVALID, e.what());
    } catch (const std::logic_error& e) {
        return fail(HEGPU_E_LOGIC, e.what());
    } catch (const std::runtime_error& e) {
        return fail(HEGPU_E_RUNTIME, e.what());
    } catch (const std::bad_alloc&) {
        return fail(HEGPU_E_RUNTIME, "out of host memory");
    }
}

extern "C" {

const char* hegpu_last_error(void) { return g_err.c_str(); }
const char* hegpu_version(void) { return "hegpu-mi355x 0.1 (gfx950)"; }

// reference util.cu:11-56 coefficient_validator
static bool coefficient_validator(const int* q, int qn, const int* p, int pn)
{
    int total_p = 0;
    for (int i = 0; i < pn; i++) total_p += p[i];
    int idx = 0;
    for (int g = 0; g < qn / pn; g++) {
        int s = 0;
        for (int j = 0; j < pn; j++) s += q[idx++];
        if (s > total_p) return false;
    }
    int s = 0;
    for (int j = 0; j < qn % pn; j++) s += q[idx++];
    return s <= total_p;
}

static void check_degree(int n, int& n_power)
{
    if (n <= 0 || (n & (n - 1))) throw std::logic_error("Poly modulus degree have to be power of two");
    if (n > 65536 || n < 4096) throw std::logic_error("Poly modulus degree is not supported");
    n_power = 31 - __builtin_clz((unsigned) n);
}

static int finish_create(hegpu_context* h, int scheme, int n_power, uint64_t plain_modulus, int q_count,
                         int p_count, hegpu_context** out)
{
    Context& c = h->c;
    c.scheme = scheme;
    c.n_power = n_power;
    c.Q_size = q_count;
    c.P_size = p_count;
    c.plain_modulus = plain_modulus;
    if (scheme == SCHEME_BFV && plain_modulus < 2) throw std::logic_error("plain modulus is not specified");
    c.build_host();
    c.seed_options_from_env(); // defaults only; hegpu_context_set_option is the interface
    *out = h;
    return 0;
}

int hegpu_context_create(int scheme, int n, const int* qb, int qn, const int* pb, int pn, uint64_t plain_modulus,
                         int sec_level, hegpu_context** out)
{
    return guarded([&]() -> int {
        if (!out) throw std::invalid_argument("null output");
        if (scheme != SCHEME_BFV && scheme != SCHEME_CKKS) throw std::invalid_argument("unknown scheme");
        int n_power;
        check_degree(n, n_power);
        if (pn <= 0 || !pb) throw std::logic_error("log_P_bases_bit_sizes cannot be empty!");
        if (qn <= 0 || !qb) throw std::logic_error("log_Q_bases_bit_sizes cannot be empty!");
        if (!coefficient_validator(qb, qn, pb, pn)) throw std::logic_error("P should be bigger than Q pairs!");
        std::vector<int> bits(qb, qb + qn);
        bits.insert(bits.end(), pb, pb + pn);
        int total = 0;
        for (int b : bits) total += b;
        if (sec_level == HEGPU_SEC_128 || sec_level == HEGPU_SEC_192 || sec_level == HEGPU_SEC_256) {
            if (host::max_logq((u64) n, sec_level) < total) // ckks/context.cu:94-119, secstdparams.h:25-79
                throw std::runtime_error("Parameters do not align with the security recommendations "
                                         "provided by the lattice-estimator");
        } else if (sec_level != HEGPU_SEC_NONE) {
            throw std::runtime_error("Invalid security level");
        }
        hegpu_context* h = new hegpu_context();
        try {
            h->c.primes = host::find_primes((u64) n, bits);
            return finish_create(h, scheme, n_power, plain_modulus, qn, pn, out);
        } catch (...) {
            delete h;
            throw;
        }
    });
}

int hegpu_context_create_default(int scheme, int n, int p_count, uint64_t plain_modulus, int sec_level,
                                 hegpu_context** out)
{
    return guarded([&]() -> int {
        if (!out) throw std::invalid_argument("null output");
        if (scheme != SCHEME_BFV && scheme != SCHEME_CKKS) throw std::invalid_argument("unknown scheme");
        int n_power;
        check_degree(n, n_power);
        if (p_count < 1) throw std::logic_error("P_modulus_size cannot be lower than 1!");
        if (sec_level != HEGPU_SEC_128 && sec_level != HEGPU_SEC_192 && sec_level != HEGPU_SEC_256)
            throw std::runtime_error("Invalid security level"); // bfv/context.cu:285-360: no default chain without a level
        std::vector<u64> chain = host::default_chain((u64) n, sec_level);
        if (chain.empty() || (int) chain.size() <= p_count) throw std::logic_error("no default chain");
        hegpu_context* h = new hegpu_context();
        try {
            h->c.primes = chain;
            return finish_create(h, scheme, n_power, plain_modulus, (int) chain.size() - p_count, p_count, out);
        } catch (...) {
            delete h;
            throw;
        }
    });
}

int hegpu_context_create_from_primes(int scheme, int n, const uint64_t* primes, int qn, int pn,
                                     uint64_t plain_modulus, hegpu_context** out)
{
    return guarded([&]() -> int {
        if (!out || !primes) throw std::invalid_argument("null argument");
        if (scheme != SCHEME_BFV && scheme != SCHEME_CKKS) throw std::invalid_argument("unknown scheme");
        int n_power;
        check_degree(n, n_power);
        if (pn <= 0 || qn <= 0) throw std::logic_error("log_P_bases_bit_sizes cannot be empty!");
        for (int i = 0; i < qn + pn; i++) {
            if (primes[i] >> 61) throw std::logic_error("invalid modulus bit size");
            if ((primes[i] - 1) % (2 * (u64) n)) throw std::logic_error("no sufficient root unity");
        }
        hegpu_context* h = new hegpu_context();
        try {
            h->c.primes.assign(primes, primes + qn + pn);
            return finish_create(h, scheme, n_power, plain_modulus, qn, pn, out);
        } catch (...) {
            delete h;
            throw;
        }
    });
}

// set_coeff_modulus_values' own checks (bfv/context.cu:149-220, ckks/context.cu:149-220): every value admits a 2N-th
// root, the widths satisfy coefficient_validator, the total width respects the security table
int hegpu_validate_coeff_modulus_values(int n, const uint64_t* primes, int qn, int pn, int sec_level)
{
    return guarded([&]() -> int {
        if (!primes) throw std::invalid_argument("null argument");
        int n_power;
        check_degree(n, n_power);
        if (pn <= 0) throw std::logic_error("log_P_bases_bit_sizes cannot be empty!");
        if (qn <= 0) throw std::logic_error("log_Q_bases_bit_sizes cannot be empty!");
        std::vector<int> bits;
        int total = 0;
        for (int i = 0; i < qn + pn; i++) {
            if (primes[i] < 2 || primes[i] >> 61) throw std::logic_error("invalid modulus bit size");
            if ((primes[i] - 1) % (2 * (u64) n)) throw std::logic_error("no sufficient root unity");
            bits.push_back(64 - __builtin_clzll(primes[i]));
            total += bits.back();
        }
        if (!coefficient_validator(bits.data(), qn, bits.data() + qn, pn))
            throw std::logic_error("Invalid parameters, P should be bigger than Q pairs!");
        if (sec_level == HEGPU_SEC_128 || sec_level == HEGPU_SEC_192 || sec_level == HEGPU_SEC_256) {
            if (host::max_logq((u64) n, sec_level) < total)
                throw std::runtime_error("Parameters do not align with the security recommendations "
                                         "provided by the lattice-estimator");
        } else if (sec_level != HEGPU_SEC_NONE) {
            throw std::runtime_error("Invalid security level");
        }
        return 0;
    });
}

void hegpu_context_destroy(hegpu_context* ctx) { delete ctx; }

int hegpu_context_set_option(hegpu_context* ctx, const char* name, int value)
{
    if (!ctx || !name) return fail(HEGPU_E_INVALID, "null argument");
    switch (ctx->c.set_option(name, value)) {
        case 0: return 0;
        case 1: return fail(HEGPU_E_INVALID, std::string("unknown option: ") + name);
        case 2: return fail(HEGPU_E_INVALID, std::string("value out of range for option ") + name);
        default: return fail(HEGPU_E_LOGIC, std::string("option ") + name + " must be set before hegpu_context_upload");
    }
}

int hegpu_context_get_option(const hegpu_context* ctx, const char* name, int* value)
{
    if (!ctx || !name || !value) return fail(HEGPU_E_INVALID, "null argument");
    if (ctx->c.get_option(name, value)) return fail(HEGPU_E_INVALID, std::string("unknown option: ") + name);
    return 0;
}

int hegpu_context_upload(hegpu_context* ctx)
{
    if (!ctx) return fail(HEGPU_E_INVALID, "null context");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        (void) hipGetLastError();
        return fail(HEGPU_E_NODEVICE, "no HIP device available: the HIP backend cannot run (no CPU fallback)");
    }
    return hip_ret(ctx->c.upload(), "context upload");
}

int hegpu_context_upload_device(hegpu_context* ctx, int device)
{
    if (!ctx) return fail(HEGPU_E_INVALID, "null context");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        (void) hipGetLastError();
        return fail(HEGPU_E_NODEVICE, "no HIP device available: the HIP backend cannot run (no CPU fallback)");
    }
    if (device < 0 || device >= cnt) return fail(HEGPU_E_INVALID, "no such device");
    if (ctx->c.uploaded)
        return ctx->c.device == device ? 0 : fail(HEGPU_E_LOGIC, "the context is already uploaded to another device");
    DevGuard g(device);
    return hip_ret(ctx->c.upload(), "context upload");
}

int hegpu_context_device(const hegpu_context* ctx) { return (ctx && ctx->c.uploaded) ? ctx->c.device : -1; }

int hegpu_context_clone(const hegpu_context* src, hegpu_context** out)
{
    return guarded([&]() -> int {
        if (!src || !out) throw std::invalid_argument("null argument");
        hegpu_context* h = new hegpu_context();
        Context& c = h->c;
        const Context& s = src->c;
        // host state and options only: the clone owns its own device tables once uploaded
        c.scheme = s.scheme; c.n_power = s.n_power; c.n = s.n;
        c.Q_size = s.Q_size; c.P_size = s.P_size; c.Qp_size = s.Qp_size; c.bsk_size = s.bsk_size;
        c.plain_modulus = s.plain_modulus; c.primes = s.primes; c.host = s.host;
        c.m2_levels = s.m2_levels; c.m2_width = s.m2_width;
        c.fused_row_mac = s.fused_row_mac; c.fused_moddown = s.fused_moddown; c.col_multi = s.col_multi;
        c.single_pass = s.single_pass; c.ntt_galois = s.ntt_galois; c.galois_scatter = s.galois_scatter;
        c.digit_split = s.digit_split; c.copy_along = s.copy_along; c.fuse_inverse = s.fuse_inverse;
        c.fp_ntt = s.fp_ntt; c.behz_split = s.behz_split; c.fused_tensor = s.fused_tensor;
        *out = h;
        return 0;
    });
}

// Evaluation keys are the only data every GPU needs a copy of.  xGMI is point-to-point: on a fully
// connected node every destination has its own link to the source, so the replication is ONE hop -- a flat fan-out,
// every copy on its destination's stream, all links busy at once (272 MiB at ~50 GB/s per direction and link: ~5 ms).
// Where some destination is not directly reachable from the source (hipDeviceCanAccessPeer) the copies form a binomial
// tree (copy i -> i + 2^r in round r) cut into <= 32 MiB chunks, so that hop r + 1 of a chunk starts as soon as that
// chunk has arrived, not when the whole buffer has.  Peer access is enabled for every edge that is used; an edge
// without it is still copied (the runtime stages it through host memory) and reported (HEGPU_BCAST_STAGED).
static thread_local int g_bcast_path = 0;
int hegpu_last_broadcast_path(void) { return g_bcast_path; }

static const size_t kBcastChunk = (size_t) 32 << 20;

// 1: direct (same device, or peer access available and now enabled on `dst` for `src`), 0: not, <0: error in *e
static int peer_edge(int src, int dst, hipError_t* e)
{
    if (src == dst) return 1;
    int can = 0;
    if ((*e = hipDeviceCanAccessPeer(&can, dst, src)) != hipSuccess) return -1;
    if (!can) return 0;
    DevGuard g(dst);
    if (g.err != hipSuccess) {
        *e = g.err;
        return -1;
    }
    const hipError_t en = hipDeviceEnablePeerAccess(src, 0);
    if (en != hipSuccess && en != hipErrorPeerAccessAlreadyEnabled) {
        *e = en;
        return -1;
    }
    (void) hipGetLastError(); // "already enabled" is not an error to leave behind
    return 1;
}

int hegpu_broadcast_bytes(const int* devices, int n, void* const* bufs, size_t bytes, const hegpu_stream* streams,
                          int* path_out)
{
    g_bcast_path = 0;
    if (path_out) *path_out = 0;
    if (!devices || !bufs || n < 1) return fail(HEGPU_E_INVALID, "null argument");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        (void) hipGetLastError();
        return fail(HEGPU_E_NODEVICE, "no HIP device available: the HIP backend cannot run (no CPU fallback)");
    }
    for (int i = 0; i < n; i++) {
        if (!bufs[i]) return fail(HEGPU_E_INVALID, "null buffer");
        if (devices[i] < 0 || devices[i] >= cnt) return fail(HEGPU_E_INVALID, "no such device");
    }
    if (n == 1 || bytes == 0) return 0;
    hipError_t e = hipSuccess;
    // ---- shape: flat when every destination is directly reachable from the source
    bool flat = true, same = true;
    for (int j = 1; j < n && flat; j++) {
        const int r = peer_edge(devices[0], devices[j], &e);
        if (r < 0) return hip_ret(e, "hegpu_broadcast: peer access");
        flat = r == 1;
    }
    for (int j = 1; j < n; j++) same = same && devices[j] == devices[0];
    std::vector<int> parent(n, 0);
    bool staged = false;
    if (!flat) {
        for (int span = 1; span < n; span <<= 1)
            for (int i = 0; i < span && i + span < n; i++) parent[i + span] = i;
        for (int j = 1; j < n; j++) {
            const int r = peer_edge(devices[parent[j]], devices[j], &e);
            if (r < 0) return hip_ret(e, "hegpu_broadcast: peer access");
            staged = staged || r == 0;
        }
    }
    // ---- copies: chunk c reaches node j on stream j, behind the event "chunk c complete at parent[j]"
    const size_t nchunks = (bytes + kBcastChunk - 1) / kBcastChunk;
    auto stream_of = [&](int i) { return streams ? (hipStream_t) streams[i] : (hipStream_t) nullptr; };
    std::vector<hipEvent_t> events;
    auto mark = [&](int i, hipEvent_t* ev) { // everything queued on stream i so far
        DevGuard g(devices[i]);
        if ((e = g.err) != hipSuccess) return;
        if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return;
        events.push_back(*ev);
        e = hipEventRecord(*ev, stream_of(i));
    };
    // arrived[j][c]; the source's one event covers all its chunks
    std::vector<std::vector<hipEvent_t>> arrived(n, std::vector<hipEvent_t>(nchunks, nullptr));
    hipEvent_t src_ready = nullptr;
    mark(0, &src_ready);
    for (size_t c = 0; c < nchunks; c++) arrived[0][c] = src_ready;
    // chunk-major order: a chunk walks down the tree before the next one is queued, so every stream sees its copies in
    // chunk order and a child never waits for more than the chunk it forwards
    for (size_t c = 0; c < nchunks && e == hipSuccess; c++) {
        const size_t off = c * kBcastChunk, len = std::min(kBcastChunk, bytes - off);
        for (int j = 1; j < n && e == hipSuccess; j++) { // parents precede children in index order
            const int pj = parent[j];
            DevGuard g(devices[j]);
            if ((e = g.err) != hipSuccess) break;
            if ((e = hipStreamWaitEvent(stream_of(j), arrived[pj][c], 0)) != hipSuccess) break;
            if ((e = hipMemcpyPeerAsync((char*) bufs[j] + off, devices[j], (const char*) bufs[pj] + off, devices[pj], len,
                                        stream_of(j))) != hipSuccess)
                break;
            bool has_child = false;
            for (int k = j + 1; k < n; k++) has_child = has_child || parent[k] == j;
            if (has_child) mark(j, &arrived[j][c]);
        }
    }
    for (hipEvent_t ev : events) (void) hipEventDestroy(ev); // released by the runtime once the recorded work has completed
    if (e != hipSuccess) return hip_ret(e, "hegpu_broadcast");
    int path = flat ? HEGPU_BCAST_FLAT : HEGPU_BCAST_TREE;
    if (staged) path |= HEGPU_BCAST_STAGED;
    if (same) path |= HEGPU_BCAST_SAME_DEVICE;
    g_bcast_path = path;
    if (path_out) *path_out = path;
    return 0;
}

int hegpu_broadcast_key(hegpu_context* const* ctxs, int n_ctx, uint64_t* const* keys, size_t elems,
                        const hegpu_stream* streams)
{
    g_bcast_path = 0;
    if (!ctxs || !keys || n_ctx < 1) return fail(HEGPU_E_INVALID, "null argument");
    std::vector<int> devs(n_ctx);
    std::vector<void*> bufs(n_ctx);
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i] || !keys[i]) return fail(HEGPU_E_INVALID, "null context or key pointer");
        if (!ctxs[i]->c.uploaded) return fail(HEGPU_E_LOGIC, "every context must be uploaded (hegpu_context_upload_device)");
        devs[i] = ctxs[i]->c.device;
        bufs[i] = keys[i];
    }
    return hegpu_broadcast_bytes(devs.data(), n_ctx, bufs.data(), elems * sizeof(u64), streams, nullptr);
}

long hegpu_context_int(const hegpu_context* ctx, const char* name)
{
    if (!ctx || !name) return -1;
    const Context& c = ctx->c;
    if (!strcmp(name, "n_power")) return c.n_power;
    if (!strcmp(name, "n")) return (long) c.n;
    if (!strcmp(name, "Q_size")) return c.Q_size;
    if (!strcmp(name, "P_size")) return c.P_size;
    if (!strcmp(name, "Q_prime_size")) return c.Qp_size;
    if (!strcmp(name, "bsk_modulus")) return c.bsk_size;
    if (!strcmp(name, "scheme")) return c.scheme;
    if (!strcmp(name, "max_logq_128")) return host::max_logq_128((int) c.n); // util/secstdparams.h
    if (!strcmp(name, "max_logq_192")) return host::max_logq(c.n, 192);
    if (!strcmp(name, "max_logq_256")) return host::max_logq(c.n, 256);
    return -1;
}

long hegpu_context_get(const hegpu_context* ctx, const char* name, uint64_t* out, long cap)
{
    if (!ctx || !name) return -1;
    auto it = ctx->c.host.find(name);
    if (it == ctx->c.host.end()) return -1;
    const long cnt = (long) it->second.size();
    if (!out) return cnt;
    if (cnt > cap) return -2;
    memcpy(out, it->second.data(), cnt * sizeof(uint64_t));
    return cnt;
}

const void* hegpu_context_device_ptr(const hegpu_context* ctx, const char* name)
{
    if (!ctx || !name) return nullptr;
    auto it = ctx->c.dev.find(name);
    return it == ctx->c.dev.end() ? nullptr : it->second;
}

int hegpu_steps_to_galois_elt(int steps, int coeff_count, int group_order)
{
    // no exception may cross the C boundary; Galois elements are odd and positive (0: step count out
    // of range, as the reference returns), so -1 is unambiguous
    int elt = -1;
    (void) guarded([&]() -> int {
        elt = host::steps_to_galois_elt(steps, coeff_count, group_order);
        return 0;
    });
    return elt;
}

#define NEED_CTX(ctx)                                                                      \
    if (!(ctx)) return fail(HEGPU_E_INVALID, "null context");                              \
    if (!(ctx)->c.uploaded) {                                                              \
        int r__ = hegpu_context_upload(ctx);                                               \
        if (r__) return r__;                                                               \
    }                                                                                      \
    DevGuard dev_guard__((ctx)->c.device);                                                 \
    if (dev_guard__.err != hipSuccess) return hip_ret(dev_guard__.err, "switching to the context's device")

static const Mod* mods_of(const Context& c, int table_set)
{
    return table_set == HEGPU_TABLES_Q_BSK ? c.plan_merge.mods : c.plan_qp.mods;
}

int hegpu_ntt(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int inverse, int batch,
              int mod_count, int mod_offset, const int* mod_order, const int* poly_order, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (table_set == HEGPU_TABLES_Q_BSK && c.scheme != SCHEME_BFV)
        return fail(HEGPU_E_INVALID, "q|Bsk tables exist only in a BFV context");
    if (table_set < 0 || table_set > HEGPU_TABLES_PLAIN || (table_set == HEGPU_TABLES_PLAIN && !c.plan_plain.count))
        return fail(HEGPU_E_INVALID, "no such table set in this context");
    NttArgs a = c.ntt_args(table_set);
    const int avail = a.mod_count;
    if (mod_count <= 0 || mod_offset < 0 || (!mod_order && mod_offset + mod_count > avail))
        return fail(HEGPU_E_INVALID, "modulus range outside the table set");
    a.in = (const u64*) in;
    a.out = (u64*) out;
    a.mod_count = mod_count;
    a.mod_offset = mod_offset;
    a.mod_order = mod_order;
    a.poly_order = poly_order;
    return hip_ret(ntt_launch(a, batch, inverse != 0, (hipStream_t) stream), "hegpu_ntt");
}

// the six gpuntt:: names (include/hegpu.h)
int hegpu_GPU_NTT(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int mod_offset, int batch,
                  int mod_count, hegpu_stream stream)
{
    return hegpu_ntt(ctx, table_set, in, out, 0, batch, mod_count, mod_offset, nullptr, nullptr, stream);
}
int hegpu_GPU_NTT_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int mod_offset, int batch, int mod_count,
                          hegpu_stream stream)
{
    return hegpu_ntt(ctx, table_set, inout, inout, 0, batch, mod_count, mod_offset, nullptr, nullptr, stream);
}
int hegpu_GPU_INTT(hegpu_context* ctx, int table_set, const uint64_t* in, uint64_t* out, int mod_offset, int batch,
                   int mod_count, hegpu_stream stream)
{
    return hegpu_ntt(ctx, table_set, in, out, 1, batch, mod_count, mod_offset, nullptr, nullptr, stream);
}
int hegpu_GPU_INTT_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int mod_offset, int batch,
                           int mod_count, hegpu_stream stream)
{
    return hegpu_ntt(ctx, table_set, inout, inout, 1, batch, mod_count, mod_offset, nullptr, nullptr, stream);
}
int hegpu_GPU_NTT_Modulus_Ordered_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int inverse,
                                          int mod_offset, int batch, int mod_count, const int* order,
                                          hegpu_stream stream)
{
    if (!order) return fail(HEGPU_E_INVALID, "modulus order table is required");
    return hegpu_ntt(ctx, table_set, inout, inout, inverse, batch, mod_count, mod_offset, order, nullptr, stream);
}
int hegpu_GPU_NTT_Poly_Ordered_Inplace(hegpu_context* ctx, int table_set, uint64_t* inout, int inverse, int mod_offset,
                                       int batch, int mod_count, const int* order, hegpu_stream stream)
{
    if (!order) return fail(HEGPU_E_INVALID, "polynomial order table is required");
    return hegpu_ntt(ctx, table_set, inout, inout, inverse, batch, mod_count, mod_offset, nullptr, order, stream);
}

int hegpu_addition(hegpu_context* ctx, const uint64_t* in1, const uint64_t* in2, uint64_t* out, int limbs,
                   int parts, int batch, int op, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (limbs <= 0 || limbs > c.Qp_size || op < 0 || op > 2) return fail(HEGPU_E_INVALID, "bad limbs/op");
    return hip_ret(rns_addition((const u64*) in1, (const u64*) in2, (u64*) out, c.plan_qp.mods, c.n_power, limbs,
                                parts, batch, op, (hipStream_t) stream),
                   "hegpu_addition");
}

int hegpu_cross_multiplication(hegpu_context* ctx, int table_set, const uint64_t* in1, uint64_t s1,
                               const uint64_t* in2, uint64_t s2, uint64_t* out, uint64_t so, int decomp_size,
                               int batch, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    const Mod* m = mods_of(c, table_set);
    if (!m) return fail(HEGPU_E_INVALID, "table set not available");
    return hip_ret(rns_cross_multiplication((const u64*) in1, s1, (const u64*) in2, s2, (u64*) out, so, m, c.n_power,
                                            decomp_size, batch, (hipStream_t) stream),
                   "hegpu_cross_multiplication");
}

int hegpu_cipher_broadcast(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                           uint64_t out_stride, int digits, int nmods, int split, int level, int batch,
                           hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    return hip_ret(rns_decompose((const u64*) in, in_stride, (u64*) out, out_stride, c.plan_qp.mods, c.n_power,
                                 digits, nmods, split, level, batch, (hipStream_t) stream),
                   "hegpu_cipher_broadcast");
}

int hegpu_keyswitch_multiply_accumulate(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride,
                                        const uint64_t* key, uint64_t* out, uint64_t out_stride, int digits,
                                        int nmods, int key_limbs, int split, int level, int batch,
                                        hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    return hip_ret(rns_keyswitch_mac((const u64*) in, in_stride, (const u64*) key, (u64*) out, out_stride,
                                     c.plan_qp.mods, c.n_power, digits, nmods, key_limbs, split, level, batch,
                                     (hipStream_t) stream),
                   "hegpu_keyswitch_multiply_accumulate");
}

int hegpu_base_conversion_DtoQtilde(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                                    uint64_t out_stride, int depth, int batch, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (c.P_size < 2) return fail(HEGPU_E_LOGIC, "method II tables exist only when P_size > 1");
    if (depth < 0 || depth >= (int) c.m2_levels.size()) return fail(HEGPU_E_INVALID, "invalid depth");
    const Context::M2Level& L = c.m2_levels[depth];
    return hip_ret(rns_base_conversion_DtoQtilde((const u64*) in, in_stride, (u64*) out, out_stride, c.plan_qp.mods,
                                                 c.d64("m2_matrix_mg") + L.off_matrix, c.d64("m2_Mi_inv") + L.off_mi,
                                                 c.d64("m2_negprod_mg") + L.off_prod, c.d32("m2_I_j") + L.off_digits,
                                                 c.d32("m2_I_location") + L.off_digits, c.n_power, L.d, L.rc,
                                                 c.Q_size - depth, depth, c.m2_width, batch, (hipStream_t) stream),
                   "hegpu_base_conversion_DtoQtilde");
}

int hegpu_divide_round_lastq(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                             uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int switchkey, int batch,
                             hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (c.P_size != 1) return fail(HEGPU_E_LOGIC, "divide_round_lastq needs a single special prime (method I)");
    return hip_ret(rns_divide_round_lastq((const u64*) in, in_stride, (const u64*) ct, ct_stride, (u64*) out,
                                          out_stride, c.plan_qp.mods, c.d64("half"), c.d64("half_mod"),
                                          c.d64("last_q_modinv"), c.n_power, c.Q_size, switchkey, batch,
                                          (hipStream_t) stream),
                   "hegpu_divide_round_lastq");
}

int hegpu_divide_round_lastq_permute(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride,
                                     const uint64_t* in2, uint64_t in2_stride, uint64_t* out,
                                     uint64_t out_stride, int galois_elt, int depth, int batch,
                                     hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (c.scheme == SCHEME_BFV && depth != 0) return fail(HEGPU_E_INVALID, "BFV ciphertexts have no depth");
    return hip_ret(rns_moddown_permute((const u64*) in, in_stride, (const u64*) in2, in2_stride, (u64*) out,
                                       out_stride, c.plan_qp.mods, c.d64("half"), c.d64("half_mod"),
                                       c.d64("last_q_modinv"), galois_elt, c.n_power, c.Qp_size - depth,
                                       c.Q_size - depth, c.Qp_size, c.Q_size, c.P_size, batch,
                                       (hipStream_t) stream),
                   "hegpu_divide_round_lastq_permute");
}

// location of depth's constants in the triangular rescale tables (reference ckks/operator.cu:1181-1187)
static int rescale_location(const Context& c, int depth)
{
    int counter = c.Q_size - 1, location = 0;
    for (int i = 0; i < depth; i++) { location += counter; counter--; }
    return location;
}
#define SEAM_CKKS(ctx, depth, batch, min_limbs)                                                                    \
    NEED_CTX(ctx);                                                                                                 \
    const Context& c = (ctx)->c;                                                                                   \
    if (c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "CKKS context required");                            \
    if ((depth) < 0 || c.Q_size - (depth) < (min_limbs)) return fail(HEGPU_E_INVALID, "invalid depth");            \
    if ((batch) < 0) return fail(HEGPU_E_INVALID, "batch must not be negative");                                   \
    if ((batch) == 0) return 0;                                                                                    \
    const int l = c.Q_size - (depth);                                                                              \
    (void) l

int hegpu_divide_round_lastq_leveled_stage_one(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                                               uint64_t out_stride, int rescale, int depth, int batch,
                                               hegpu_stream stream)
{
    SEAM_CKKS(ctx, depth, batch, rescale ? 2 : 1);
    if (rescale)
        return hip_ret(rns_moddown_stage_one((const u64*) in, in_stride, (u64*) out, out_stride, c.plan_qp.mods,
                                             c.d64("rescaled_half") + depth,
                                             c.d64("rescaled_half_mod") + rescale_location(c, depth), c.n_power, l - 1,
                                             l - 1, batch, (hipStream_t) stream),
                       "hegpu_divide_round_lastq_leveled_stage_one");
    if (c.P_size != 1) return fail(HEGPU_E_LOGIC, "the leveled stages serve a single special prime (method I)");
    return hip_ret(rns_moddown_stage_one((const u64*) in, in_stride, (u64*) out, out_stride, c.plan_qp.mods,
                                         c.d64("half"), c.d64("half_mod"), c.n_power, c.Q_size, l, batch,
                                         (hipStream_t) stream),
                   "hegpu_divide_round_lastq_leveled_stage_one");
}

int hegpu_divide_round_lastq_leveled_stage_two(hegpu_context* ctx, const uint64_t* in_last, uint64_t last_stride,
                                               const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                                               uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int switchkey,
                                               int depth, int batch, hegpu_stream stream)
{
    SEAM_CKKS(ctx, depth, batch, 1);
    if (c.P_size != 1) return fail(HEGPU_E_LOGIC, "the leveled stages serve a single special prime (method I)");
    if (!ct) return fail(HEGPU_E_INVALID, "ct must not be NULL");
    return hip_ret(rns_moddown_stage_two((const u64*) in_last, last_stride, (const u64*) in, in_stride, l + 1,
                                         (const u64*) ct, ct_stride, (u64*) out, out_stride, c.plan_qp.mods,
                                         c.d64("last_q_modinv"), c.n_power, l, switchkey ? 2 : 1, batch,
                                         (hipStream_t) stream),
                   "hegpu_divide_round_lastq_leveled_stage_two");
}

int hegpu_move_cipher_leveled(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out,
                              uint64_t out_stride, int depth, int batch, hegpu_stream stream)
{
    SEAM_CKKS(ctx, depth, batch, 2);
    return hip_ret(rns_copy_limbs((const u64*) in, (u64) l * c.n, in_stride, (u64*) out, (u64) l * c.n, out_stride,
                                  c.n_power, l - 1, 2, batch, (hipStream_t) stream),
                   "hegpu_move_cipher_leveled");
}

int hegpu_divide_round_lastq_rescale(hegpu_context* ctx, const uint64_t* in_last, uint64_t last_stride,
                                     const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                                     int depth, int batch, hegpu_stream stream)
{
    SEAM_CKKS(ctx, depth, batch, 2);
    return hip_ret(rns_moddown_stage_two((const u64*) in_last, last_stride, (const u64*) in, in_stride, l, nullptr, 0,
                                         (u64*) out, out_stride, c.plan_qp.mods,
                                         c.d64("rescaled_last_q_modinv") + rescale_location(c, depth), c.n_power, l - 1,
                                         0, batch, (hipStream_t) stream),
                   "hegpu_divide_round_lastq_rescale");
}

int hegpu_divide_round_lastq_extended(hegpu_context* ctx, const uint64_t* in, uint64_t in_stride, const uint64_t* ct,
                                      uint64_t ct_stride, uint64_t* out, uint64_t out_stride, int mode, int depth,
                                      int batch, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (mode < 0 || mode > 2) return fail(HEGPU_E_INVALID, "mode must be 0, 1 or 2");
    if (c.scheme == SCHEME_BFV && depth != 0) return fail(HEGPU_E_INVALID, "BFV ciphertexts have no depth");
    if (depth < 0 || depth >= c.Q_size) return fail(HEGPU_E_INVALID, "invalid depth");
    if (mode && !ct) return fail(HEGPU_E_INVALID, "ct must not be NULL");
    if (batch < 0) return fail(HEGPU_E_INVALID, "batch must not be negative");
    if (batch == 0) return 0;
    return hip_ret(rns_moddown_extended((const u64*) in, in_stride, (const u64*) ct, ct_stride, (u64*) out, out_stride,
                                        c.plan_qp.mods, c.d64("half"), c.d64("half_mod"), c.d64("last_q_modinv"),
                                        c.n_power, c.Qp_size - depth, c.Q_size - depth, c.Qp_size, c.Q_size, c.P_size,
                                        mode, batch, (hipStream_t) stream),
                   "hegpu_divide_round_lastq_extended");
}

int hegpu_fast_convertion(hegpu_context* ctx, const uint64_t* in1, uint64_t s1, const uint64_t* in2, uint64_t s2,
                          uint64_t* out, uint64_t so, int batch, hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "BFV context required");
    return hip_ret(rns_fast_convertion((const u64*) in1, s1, (const u64*) in2, s2, (u64*) out, so, c.behz,
                                       c.n_power, batch, (hipStream_t) stream),
                   "hegpu_fast_convertion");
}

int hegpu_fast_floor(hegpu_context* ctx, const uint64_t* in, uint64_t si, uint64_t* out, uint64_t so, int batch,
                     hegpu_stream stream)
{
    NEED_CTX(ctx);
    const Context& c = ctx->c;
    if (c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "BFV context required");
    return hip_ret(rns_fast_floor((const u64*) in, si, (u64*) out, so, c.behz, c.n_power, batch,
                                  (hipStream_t) stream),
                   "hegpu_fast_floor");
}

size_t hegpu_workspace_bytes(const hegpu_context* ctx, int op, int depth, int batch)
{
    if (!ctx || batch <= 0) return 0;
    return ops_workspace_elems(ctx->c, op, depth, batch) * sizeof(u64);
}

#define CHECK_OP(ctx, want_scheme, op, depth, batch, ws, ws_bytes)                                            \
    do {                                                                                                      \
        if ((ctx)->c.scheme != (want_scheme)) return fail(HEGPU_E_INVALID, "context scheme mismatch");        \
        if ((batch) < 0) return fail(HEGPU_E_INVALID, "batch must not be negative");                          \
        if ((batch) == 0) return 0; /* an empty batch is a no-op */                                           \
        if ((depth) < 0 || (depth) >= (ctx)->c.Q_size) return fail(HEGPU_E_INVALID, "invalid depth");         \
        if ((op) && (!(ws) || (ws_bytes) < hegpu_workspace_bytes(ctx, op, depth, batch)))                     \
            return fail(HEGPU_E_INVALID, "workspace too small");                                              \
    } while (0)

int hegpu_ckks_multiply(hegpu_context* ctx, const uint64_t* ct1, uint64_t s1, const uint64_t* ct2, uint64_t s2,
                        uint64_t* out, uint64_t so, int depth, int batch, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (depth < 0 || depth >= ctx->c.Q_size) return fail(HEGPU_E_INVALID, "invalid depth");
    return hip_ret(op_ckks_multiply(ctx->c, (const u64*) ct1, s1, (const u64*) ct2, s2, (u64*) out, so, depth, batch,
                                    (hipStream_t) stream),
                   "hegpu_ckks_multiply");
}

int hegpu_ckks_relinearize_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t cs, const uint64_t* key, int depth,
                                   int batch, void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_CKKS, OP_CKKS_RELIN, depth, batch, ws, ws_bytes);
    return hip_ret(ctx->c.P_size == 1 ? op_ckks_relinearize(ctx->c, (u64*) ct, cs, (const u64*) key, depth, batch,
                                                            (u64*) ws, (hipStream_t) stream)
                                      : op_ckks_relinearize_II(ctx->c, (u64*) ct, cs, (const u64*) key, depth, batch,
                                                               (u64*) ws, (hipStream_t) stream),
                   "hegpu_ckks_relinearize_inplace");
}

int hegpu_probe_ckks_relinearize(hegpu_context* ctx, uint64_t* ct, uint64_t cs, const uint64_t* key, int depth,
                                 int batch, void* ws, size_t ws_bytes, unsigned phases, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_CKKS, OP_CKKS_RELIN, depth, batch, ws, ws_bytes);
    if (ctx->c.P_size != 1) return fail(HEGPU_E_LOGIC, "the phase probe covers key-switching method I");
    return hip_ret(op_ckks_relinearize(ctx->c, (u64*) ct, cs, (const u64*) key, depth, batch, (u64*) ws,
                                       (hipStream_t) stream, phases),
                   "hegpu_probe_ckks_relinearize");
}

int hegpu_ckks_rescale_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t cs, int depth, int batch, void* ws,
                               size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_CKKS, OP_CKKS_RESCALE, depth, batch, ws, ws_bytes);
    if (depth >= ctx->c.Q_size - 1) return fail(HEGPU_E_LOGIC, "no modulus left to rescale by");
    return hip_ret(op_ckks_rescale(ctx->c, (u64*) ct, cs, depth, batch, (u64*) ws, (hipStream_t) stream),
                   "hegpu_ckks_rescale_inplace");
}

// The rotations read the input ciphertext while their epilogue already writes (scatters) results: no item of the result
// batch may share a word with an item of the input batch.  Items: [p + i * stride, + item) for i < batch, so layouts in
// which the two batches interleave without touching (ct and out alternating in one buffer, stride 2 * words) are fine.
// Equal strides: item i of a and item j of b overlap iff -b_item < (b - a) + (j - i) * stride < a_item -- one test per
// difference j - i.  Unequal strides: pairwise up to 1024 items, beyond that the whole spans (conservative).
static bool spans_overlap(const uint64_t* a, uint64_t a_stride, uint64_t a_item, const uint64_t* b, uint64_t b_stride,
                          uint64_t b_item, int batch)
{
    if (batch <= 0) return false;
    const __int128 W = sizeof(uint64_t);
    const __int128 a0 = (__int128) (uintptr_t) a, b0 = (__int128) (uintptr_t) b;
    const __int128 ai = (__int128) a_item * W, bi = (__int128) b_item * W;
    auto hit = [&](__int128 pa, __int128 pb) { return pa < pb + bi && pb < pa + ai; };
    if (batch == 1) return hit(a0, b0);
    {   // the whole spans first: disjoint buffers (the common case) need no per-item test
        const __int128 a1 = a0 + ((__int128) (batch - 1) * a_stride + a_item) * W;
        const __int128 b1 = b0 + ((__int128) (batch - 1) * b_stride + b_item) * W;
        if (!(a0 < b1 && b0 < a1)) return false;
    }
    if (a_stride == b_stride) {
        const __int128 st = (__int128) a_stride * W;
        for (long k = -(long) (batch - 1); k <= (long) (batch - 1); k++)
            if (hit(a0, b0 + k * st)) return true;
        return false;
    }
    if (batch <= 1024) {
        for (int i = 0; i < batch; i++)
            for (int j = 0; j < batch; j++)
                if (hit(a0 + (__int128) i * a_stride * W, b0 + (__int128) j * b_stride * W)) return true;
        return false;
    }
    return true; // beyond 1024 items with unequal strides: intersecting spans count as overlap (conservative)
}

int hegpu_ckks_apply_galois(hegpu_context* ctx, const uint64_t* ct, uint64_t cs, uint64_t* out, uint64_t so,
                            const uint64_t* key, int galois_elt, int depth, int batch, void* ws, size_t ws_bytes,
                            hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_CKKS, OP_CKKS_GALOIS, depth, batch, ws, ws_bytes);
    {
        const uint64_t words = (uint64_t) 2 * (ctx->c.Q_size - depth) * ctx->c.n;
        if (spans_overlap(ct, cs, words, out, so, words, batch))
            return fail(HEGPU_E_INVALID, "apply_galois: out must not alias ct (the result buffer must not contain the input)");
    }
    if (galois_elt <= 0 || !(galois_elt & 1) || galois_elt >= 2 * (int) ctx->c.n)
        return fail(HEGPU_E_INVALID, "apply_galois: Galois elements are odd and below 2N");
    return hip_ret(ctx->c.P_size == 1
                       ? op_ckks_apply_galois(ctx->c, (const u64*) ct, cs, (u64*) out, so, (const u64*) key,
                                              galois_elt, depth, batch, (u64*) ws, (hipStream_t) stream)
                       : op_ckks_apply_galois_II(ctx->c, (const u64*) ct, cs, (u64*) out, so, (const u64*) key,
                                                 galois_elt, depth, batch, (u64*) ws, (hipStream_t) stream),
                   "hegpu_ckks_apply_galois");
}

int hegpu_ckks_rotate_hoisted(hegpu_context* ctx, const uint64_t* ct, uint64_t cs, uint64_t* out, uint64_t so,
                              const uint64_t* const* keys, const int* galois_elts, int count, int depth, int batch,
                              void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_CKKS, OP_CKKS_GALOIS, depth, batch, ws, ws_bytes);
    if (count <= 0 || !keys || !galois_elts) return fail(HEGPU_E_INVALID, "rotate_hoisted: empty element list");
    const uint64_t words = (uint64_t) 2 * (ctx->c.Q_size - depth) * ctx->c.n;
    if (batch > 1 && so < (uint64_t) count * words) return fail(HEGPU_E_INVALID, "rotate_hoisted: out_stride too small");
    if (spans_overlap(ct, cs, words, out, so, (uint64_t) count * words, batch))
        return fail(HEGPU_E_INVALID, "rotate_hoisted: out must not alias ct (the result buffer must not contain the input)");
    for (int i = 0; i < count; i++) {
        if (galois_elts[i] == 0) continue;
        if (galois_elts[i] < 0 || !(galois_elts[i] & 1) || galois_elts[i] >= 2 * (int) ctx->c.n)
            return fail(HEGPU_E_INVALID, "rotate_hoisted: Galois elements are odd and below 2N");
        if (!keys[i]) return fail(HEGPU_E_INVALID, "rotate_hoisted: Galois key not present!");
    }
    // a workspace of HEGPU_OP_CKKS_ROTATE_HOISTED size holds four accumulators: the inner products of four
    // elements then share one read of the digits
    const bool big = ws_bytes >= hegpu_workspace_bytes(ctx, OP_CKKS_ROTATE_HOISTED, depth, batch) && ctx->c.fused_moddown &&
                     ctx->c.ntt_galois;
    return hip_ret(op_ckks_rotate_hoisted(ctx->c, (const u64*) ct, cs, (u64*) out, so, (const u64* const*) keys,
                                          galois_elts, count, depth, batch, (u64*) ws, (hipStream_t) stream, big ? 4 : 1),
                   "hegpu_ckks_rotate_hoisted");
}

int hegpu_bfv_multiply(hegpu_context* ctx, const uint64_t* ct1, uint64_t s1, const uint64_t* ct2, uint64_t s2,
                       uint64_t* out, uint64_t so, int batch, void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (batch < 0) return fail(HEGPU_E_INVALID, "batch must not be negative");
    if (batch == 0) return 0;
    if (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_BFV_MULTIPLY, 0, batch))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return hip_ret(op_bfv_multiply(ctx->c, (const u64*) ct1, s1, (const u64*) ct2, s2, (u64*) out, so, batch,
                                   (u64*) ws, (hipStream_t) stream),
                   "hegpu_bfv_multiply");
}

int hegpu_bfv_relinearize_inplace(hegpu_context* ctx, uint64_t* ct, uint64_t cs, const uint64_t* key, int batch,
                                  void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_BFV, OP_BFV_RELIN, 0, batch, ws, ws_bytes);
    return hip_ret(ctx->c.P_size == 1 ? op_bfv_relinearize(ctx->c, (u64*) ct, cs, (const u64*) key, batch, (u64*) ws,
                                                           (hipStream_t) stream)
                                      : op_bfv_relinearize_II(ctx->c, (u64*) ct, cs, (const u64*) key, batch,
                                                              (u64*) ws, (hipStream_t) stream),
                   "hegpu_bfv_relinearize_inplace");
}

int hegpu_bfv_apply_galois(hegpu_context* ctx, const uint64_t* ct, uint64_t cs, uint64_t* out, uint64_t so,
                           const uint64_t* key, int galois_elt, int batch, void* ws, size_t ws_bytes,
                           hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_OP(ctx, SCHEME_BFV, OP_BFV_GALOIS, 0, batch, ws, ws_bytes);
    {
        const uint64_t words = (uint64_t) 2 * ctx->c.Q_size * ctx->c.n;
        if (spans_overlap(ct, cs, words, out, so, words, batch))
            return fail(HEGPU_E_INVALID, "apply_galois: out must not alias ct (the result buffer must not contain the input)");
    }
    return hip_ret(ctx->c.P_size == 1
                       ? op_bfv_apply_galois(ctx->c, (const u64*) ct, cs, (u64*) out, so, (const u64*) key,
                                             galois_elt, batch, (u64*) ws, (hipStream_t) stream)
                       : op_bfv_apply_galois_II(ctx->c, (const u64*) ct, cs, (u64*) out, so, (const u64*) key,
                                                galois_elt, batch, (u64*) ws, (hipStream_t) stream),
                   "hegpu_bfv_apply_galois");
}

// ------------------------------------------------------------------ key generation / encryption / decryption
struct hegpu_rng {
    Rng r;
};

int hegpu_rng_create(uint64_t seed, hegpu_rng** out)
{
    return guarded([&]() -> int {
        if (!out) throw std::invalid_argument("null output");
        hegpu_rng* h = new hegpu_rng();
        h->r.seed = drbg_key_from_u64(seed);
        *out = h;
        return 0;
    });
}

int hegpu_rng_create_seeded(const uint8_t seed[32], hegpu_rng** out)
{
    return guarded([&]() -> int {
        if (!out || !seed) throw std::invalid_argument("null argument");
        hegpu_rng* h = new hegpu_rng();
        for (int i = 0; i < 8; i++)
            h->r.seed.k[i] = (u32) seed[4 * i] | ((u32) seed[4 * i + 1] << 8) | ((u32) seed[4 * i + 2] << 16) |
                             ((u32) seed[4 * i + 3] << 24);
        *out = h;
        return 0;
    });
}

int hegpu_rng_create_from_entropy(hegpu_rng** out)
{
    return guarded([&]() -> int {
        if (!out) throw std::invalid_argument("null output");
        uint8_t seed[32];
        size_t got = 0;
        while (got < sizeof(seed)) {
            const ssize_t r = getrandom(seed + got, sizeof(seed) - got, 0);
            if (r < 0) {
                if (errno == EINTR) continue;
                throw std::runtime_error("getrandom failed: no entropy source for the key generator");
            }
            got += (size_t) r;
        }
        const int rc = hegpu_rng_create_seeded(seed, out);
        volatile uint8_t* z = seed;
        for (size_t i = 0; i < sizeof(seed); i++) z[i] = 0;
        return rc;
    });
}

int hegpu_drbg_block(const uint8_t key[32], uint64_t stream, uint64_t index, uint32_t out[4])
{
    if (!key || !out) return fail(HEGPU_E_INVALID, "null argument");
    DrbgKey k;
    for (int i = 0; i < 8; i++)
        k.k[i] = (u32) key[4 * i] | ((u32) key[4 * i + 1] << 8) | ((u32) key[4 * i + 2] << 16) | ((u32) key[4 * i + 3] << 24);
    const DrbgOut o = drbg_block(k, stream, index);
    for (int i = 0; i < 4; i++) out[i] = o.w[i];
    return 0;
}

void hegpu_rng_destroy(hegpu_rng* rng)
{
    if (!rng) return;
    explicit_bzero(&rng->r, sizeof(rng->r)); // the 256-bit key must not stay behind in freed heap memory
    delete rng;
}

#define CHECK_KG(ctx, rng, op, ws, ws_bytes)                                                          \
    do {                                                                                              \
        if (!(rng)) return fail(HEGPU_E_INVALID, "null random generator");                            \
        if (!(ws) || (ws_bytes) < hegpu_workspace_bytes(ctx, op, 0, 1))                               \
            return fail(HEGPU_E_INVALID, "workspace too small");                                      \
    } while (0)

int hegpu_generate_secret_key(hegpu_context* ctx, hegpu_rng* rng, int hamming_weight, uint64_t* sk, void* ws,
                              size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_KG(ctx, rng, OP_KEYGEN_SECRET, ws, ws_bytes);
    if (hamming_weight <= 0 || hamming_weight > (int) ctx->c.n)
        return fail(HEGPU_E_INVALID, "hamming weight has to be in range 0 to ring size"); // secretkey.cu:43
    return hip_ret(op_gen_secret_key(ctx->c, rng->r, hamming_weight, (u64*) sk, (u64*) ws, (hipStream_t) stream),
                   "hegpu_generate_secret_key");
}

int hegpu_generate_public_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, uint64_t* pk, void* ws,
                              size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_KG(ctx, rng, OP_KEYGEN_PUBLIC, ws, ws_bytes);
    return hip_ret(op_gen_public_key(ctx->c, rng->r, (const u64*) sk, (u64*) pk, (u64*) ws, (hipStream_t) stream),
                   "hegpu_generate_public_key");
}

int hegpu_generate_relin_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, uint64_t* rk, void* ws,
                             size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_KG(ctx, rng, OP_KEYGEN_SWITCH, ws, ws_bytes);
    return hip_ret(op_gen_switch_key(ctx->c, rng->r, (const u64*) sk, 0, nullptr, (u64*) rk, (u64*) ws,
                                     (hipStream_t) stream),
                   "hegpu_generate_relin_key");
}

int hegpu_generate_switch_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* new_sk, const uint64_t* old_sk,
                              uint64_t* swk, void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_KG(ctx, rng, OP_KEYGEN_SWITCH, ws, ws_bytes);
    if (!old_sk) return fail(HEGPU_E_INVALID, "null old secret key");
    return hip_ret(op_gen_switch_key(ctx->c, rng->r, (const u64*) new_sk, 0, (const u64*) old_sk, (u64*) swk, (u64*) ws,
                                     (hipStream_t) stream),
                   "hegpu_generate_switch_key");
}

int hegpu_generate_galois_key(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* sk, int galois_elt, uint64_t* gk,
                              void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    CHECK_KG(ctx, rng, OP_KEYGEN_SWITCH, ws, ws_bytes);
    if (!(galois_elt & 1) || galois_elt <= 0 || galois_elt >= (int) (2 * ctx->c.n))
        return fail(HEGPU_E_INVALID, "galois element must be odd and below 2N");
    return hip_ret(op_gen_switch_key(ctx->c, rng->r, (const u64*) sk, galois_elt, nullptr, (u64*) gk, (u64*) ws,
                                     (hipStream_t) stream),
                   "hegpu_generate_galois_key");
}

int hegpu_ckks_encrypt(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* pk, const uint64_t* plain, uint64_t* ct,
                       void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    CHECK_KG(ctx, rng, OP_CKKS_ENCRYPT, ws, ws_bytes);
    return hip_ret(op_ckks_encrypt(ctx->c, rng->r, (const u64*) pk, (const u64*) plain, (u64*) ct, (u64*) ws,
                                   (hipStream_t) stream),
                   "hegpu_ckks_encrypt");
}

int hegpu_ckks_decrypt(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, int depth, uint64_t* plain,
                       hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (depth < 0 || depth >= ctx->c.Q_size) return fail(HEGPU_E_INVALID, "invalid depth");
    return hip_ret(op_ckks_decrypt(ctx->c, (const u64*) ct, (const u64*) sk, depth, (u64*) plain,
                                   (hipStream_t) stream),
                   "hegpu_ckks_decrypt");
}

int hegpu_bfv_encrypt(hegpu_context* ctx, hegpu_rng* rng, const uint64_t* pk, const uint64_t* plain, uint64_t* ct,
                      void* ws, size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    CHECK_KG(ctx, rng, OP_BFV_ENCRYPT, ws, ws_bytes);
    return hip_ret(op_bfv_encrypt(ctx->c, rng->r, (const u64*) pk, (const u64*) plain, (u64*) ct, (u64*) ws,
                                  (hipStream_t) stream),
                   "hegpu_bfv_encrypt");
}

int hegpu_bfv_decrypt(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, uint64_t* plain, void* ws,
                      size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_BFV_DECRYPT, 0, 1))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return guarded([&]() -> int {
        return hip_ret(op_bfv_decrypt(ctx->c, (const u64*) ct, (const u64*) sk, (u64*) plain, (u64*) ws,
                                      (hipStream_t) stream),
                       "hegpu_bfv_decrypt");
    });
}

int hegpu_bfv_encode(hegpu_context* ctx, const int64_t* message, int message_size, uint64_t* plain,
                     hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (!ctx->c.plan_plain.count)
        return fail(HEGPU_E_LOGIC, "batching needs a prime plain modulus with 2N | t - 1");
    if (message_size < 0 || message_size > (int) ctx->c.n)
        return fail(HEGPU_E_INVALID, "Message size can not be higher than the slot count."); // bfv/encoder.cuh:60
    return hip_ret(op_bfv_encode(ctx->c, (const long long*) message, message_size, (u64*) plain,
                                 (hipStream_t) stream),
                   "hegpu_bfv_encode");
}

int hegpu_bfv_decode(hegpu_context* ctx, const uint64_t* plain, uint64_t* message, void* ws, size_t ws_bytes,
                     hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (!ctx->c.plan_plain.count)
        return fail(HEGPU_E_LOGIC, "batching needs a prime plain modulus with 2N | t - 1");
    if (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_BFV_DECODE, 0, 1))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return hip_ret(op_bfv_decode(ctx->c, (const u64*) plain, (u64*) message, (u64*) ws, (hipStream_t) stream),
                   "hegpu_bfv_decode");
}

static int ckks_encode_any(hegpu_context* ctx, int mode, const double* message, int message_size, double scalar,
                           double scale, uint64_t* plain, void* ws, size_t ws_bytes, hegpu_stream stream, const char* who)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (mode == 2) {
        if (message_size < 0 || message_size > (int) ctx->c.n)
            return fail(HEGPU_E_INVALID, "Vector size can not be higher than polynomial degree!"); // ckks/encoder.cuh:80
    } else if (mode != 3 && (message_size < 0 || message_size > (int) (ctx->c.n >> 1))) {
        return fail(HEGPU_E_INVALID, "Vector size can not be higher than slot count!");            // :74
    }
    if (!(scale > 0.0)) return fail(HEGPU_E_INVALID, "Scale out of bounds");                       // :63
    if (mode < 2 && (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_CKKS_ENCODE, 0, 1)))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return hip_ret(op_ckks_encode(ctx->c, mode, message, message_size, scalar, scale, (u64*) plain, (u64*) ws,
                                  (hipStream_t) stream), who);
}

int hegpu_ckks_encode(hegpu_context* ctx, const double* message, int message_size, double scale, uint64_t* plain,
                      void* ws, size_t ws_bytes, hegpu_stream stream)
{
    return ckks_encode_any(ctx, 0, message, message_size, 0.0, scale, plain, ws, ws_bytes, stream, "hegpu_ckks_encode");
}
int hegpu_ckks_encode_complex(hegpu_context* ctx, const double* message, int message_size, double scale,
                              uint64_t* plain, void* ws, size_t ws_bytes, hegpu_stream stream)
{
    return ckks_encode_any(ctx, 1, message, message_size, 0.0, scale, plain, ws, ws_bytes, stream,
                           "hegpu_ckks_encode_complex");
}
int hegpu_ckks_encode_coeff(hegpu_context* ctx, const double* message, int message_size, double scale, uint64_t* plain,
                            hegpu_stream stream)
{
    return ckks_encode_any(ctx, 2, message, message_size, 0.0, scale, plain, nullptr, 0, stream,
                           "hegpu_ckks_encode_coeff");
}
int hegpu_ckks_encode_scalar(hegpu_context* ctx, double value, double scale, uint64_t* plain, hegpu_stream stream)
{
    return ckks_encode_any(ctx, 3, nullptr, 0, value, scale, plain, nullptr, 0, stream, "hegpu_ckks_encode_scalar");
}

static int ckks_decode_any(hegpu_context* ctx, int mode, const uint64_t* plain, int depth, double scale, double* message,
                           void* ws, size_t ws_bytes, hegpu_stream stream, const char* who)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (depth < 0 || depth >= ctx->c.Q_size) return fail(HEGPU_E_INVALID, "invalid depth");
    if (!(scale > 0.0)) return fail(HEGPU_E_INVALID, "scale must be positive");
    if (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_CKKS_DECODE, depth, 1))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return hip_ret(op_ckks_decode(ctx->c, mode, (const u64*) plain, depth, scale, message, (u64*) ws,
                                  (hipStream_t) stream), who);
}

int hegpu_ckks_decode(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message, void* ws,
                      size_t ws_bytes, hegpu_stream stream)
{
    return ckks_decode_any(ctx, 0, plain, depth, scale, message, ws, ws_bytes, stream, "hegpu_ckks_decode");
}
int hegpu_ckks_decode_complex(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message,
                              void* ws, size_t ws_bytes, hegpu_stream stream)
{
    return ckks_decode_any(ctx, 1, plain, depth, scale, message, ws, ws_bytes, stream, "hegpu_ckks_decode_complex");
}
int hegpu_ckks_decode_coeff(hegpu_context* ctx, const uint64_t* plain, int depth, double scale, double* message,
                            void* ws, size_t ws_bytes, hegpu_stream stream)
{
    return ckks_decode_any(ctx, 2, plain, depth, scale, message, ws, ws_bytes, stream, "hegpu_ckks_decode_coeff");
}

int hegpu_bfv_noise_rns(hegpu_context* ctx, const uint64_t* ct, const uint64_t* sk, uint64_t* out, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    return hip_ret(op_bfv_noise_rns(ctx->c, (const u64*) ct, (const u64*) sk, (u64*) out, (hipStream_t) stream),
                   "hegpu_bfv_noise_rns");
}

// ---- ciphertext (x) plaintext operations
int hegpu_cipherplain_multiplication(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out,
                                     int limbs, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (limbs <= 0 || limbs > ctx->c.Qp_size) return fail(HEGPU_E_INVALID, "bad limb count");
    return hip_ret(kg_pk_u((const u64*) ct, (const u64*) plain, (u64*) out, ctx->c.plan_qp.mods, ctx->c.n_power, limbs,
                           (hipStream_t) stream),
                   "hegpu_cipherplain_multiplication");
}

int hegpu_ckks_constant_op(hegpu_context* ctx, int op, const uint64_t* ct, double value, uint64_t* out, int limbs,
                           int parts, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (op < 0 || op > 2) return fail(HEGPU_E_INVALID, "unknown constant operation");
    if (limbs <= 0 || limbs > ctx->c.Q_size || parts < 2 || parts > 3) return fail(HEGPU_E_INVALID, "bad ciphertext shape");
    if (!(value == value) || value >= 3.4e38 || value <= -3.4e38) return fail(HEGPU_E_INVALID, "constant out of range");
    return hip_ret(kg_ckks_constant((const u64*) ct, value, (u64*) out, ctx->c.plan_qp.mods, ctx->c.n_power, limbs, parts,
                                    op, (hipStream_t) stream),
                   "hegpu_ckks_constant_op");
}

int hegpu_ckks_gaussian_integer_op(hegpu_context* ctx, int op, const uint64_t* ct, double re, double im, uint64_t* out,
                                   int limbs, int parts, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (op < 0 || op > 1) return fail(HEGPU_E_INVALID, "unknown constant operation");
    if (limbs <= 0 || limbs > ctx->c.Q_size || parts < 2 || parts > 3) return fail(HEGPU_E_INVALID, "bad ciphertext shape");
    for (double v : {re, im}) // any finite double, as the reference's NTL conversion (ckks/operator.cu:583-617)
        if (!std::isfinite(v)) return fail(HEGPU_E_INVALID, "constant is not a finite number");
    return guarded([&]() -> int {
        return hip_ret(kg_ckks_gaussian((const u64*) ct, re, im, (u64*) out, ctx->c.d64("psi_half"), ctx->c.plan_qp.mods,
                                        ctx->c.n_power, limbs, parts, op, (hipStream_t) stream),
                       "hegpu_ckks_gaussian_integer_op");
    });
}

int hegpu_ckks_mult_i(hegpu_context* ctx, const uint64_t* ct, uint64_t* out, int limbs, int parts, int divide,
                      hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_CKKS) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (limbs <= 0 || limbs > ctx->c.Q_size || parts < 2 || parts > 3) return fail(HEGPU_E_INVALID, "bad ciphertext shape");
    return guarded([&]() -> int {
        return hip_ret(kg_ckks_mult_i((const u64*) ct, (u64*) out, ctx->c.d64("psi_half"), ctx->c.plan_qp.mods,
                                      ctx->c.n_power, limbs, parts, divide, (hipStream_t) stream),
                       "hegpu_ckks_mult_i");
    });
}

int hegpu_bfv_plain_addsub(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out, int sub,
                           hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    const Context& c = ctx->c;
    return guarded([&]() -> int {
        return hip_ret(kg_bfv_plain_addsub((const u64*) ct, (const u64*) plain, (u64*) out, c.plan_qp.mods,
                                           c.d64("coeff_div_plain_modulus"), c.h64("Q_mod_t")[0],
                                           c.h64("upper_threshold")[0], c.plain_modulus, c.n_power, c.Q_size, sub,
                                           (hipStream_t) stream),
                       "hegpu_bfv_plain_addsub");
    });
}

int hegpu_bfv_plain_to_ntt(hegpu_context* ctx, const uint64_t* plain, uint64_t* out, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    return guarded([&]() -> int {
        return hip_ret(op_bfv_plain_to_ntt(ctx->c, (const u64*) plain, (u64*) out, (hipStream_t) stream),
                       "hegpu_bfv_plain_to_ntt");
    });
}

int hegpu_negacyclic_shift(hegpu_context* ctx, const uint64_t* in, uint64_t* out, int shift, int limbs, int parts,
                           hegpu_stream stream)
{
    NEED_CTX(ctx);
    if ((const void*) in == (const void*) out) return fail(HEGPU_E_INVALID, "negacyclic shift: out must not alias in");
    if (limbs <= 0 || limbs > ctx->c.Qp_size || parts < 1 || parts > 3) return fail(HEGPU_E_INVALID, "bad ciphertext shape");
    if (shift < 0 || shift >= (int) (2 * ctx->c.n)) return fail(HEGPU_E_INVALID, "shift must be in [0, 2N)");
    return hip_ret(kg_negacyclic_shift((const u64*) in, (u64*) out, ctx->c.plan_qp.mods, shift, ctx->c.n_power, limbs,
                                       parts, (hipStream_t) stream),
                   "hegpu_negacyclic_shift");
}

int hegpu_bfv_multiply_plain(hegpu_context* ctx, const uint64_t* ct, const uint64_t* plain, uint64_t* out, void* ws,
                             size_t ws_bytes, hegpu_stream stream)
{
    NEED_CTX(ctx);
    if (ctx->c.scheme != SCHEME_BFV) return fail(HEGPU_E_INVALID, "context scheme mismatch");
    if (!ws || ws_bytes < hegpu_workspace_bytes(ctx, OP_BFV_MULTIPLY_PLAIN, 0, 1))
        return fail(HEGPU_E_INVALID, "workspace too small");
    return guarded([&]() -> int {
        return hip_ret(op_bfv_multiply_plain(ctx->c, (const u64*) ct, (const u64*) plain, (u64*) out, (u64*) ws,
                                             (hipStream_t) stream),
                       "hegpu_bfv_multiply_plain");
    });
}

// ------------------------------------------------------------------ TFHE
struct hegpu_tfhe_context {
    TfheDev p{};
    std::vector<ulonglong2> htw, hitw, hftw, hfitw;
    ulonglong2* dtw = nullptr;
    ulonglong2* ditw = nullptr;
    ulonglong2* dftw = nullptr;
    ulonglong2* dfitw = nullptr;
    bool uploaded = false;
    int device = -1;           // the device the tables live on (the calling thread's current device at first use)
    bool allow_fp = true;      // option "fp" = 0 keeps the integer blind rotate (read by hegpu_tfhe_prepare_bootkey)
    int ks_pieces = -1;        // option "ks_pieces": workgroups per gate (group) of the key switching, -1 by launch size
    int ks_batched = -1;       // option "ks_batched": key switching with 8 / 12 / 16 gates per workgroup sharing the key rows (tfhe.hip)
    // The layout of a prepared boot key is its header word, read by the blind-rotate kernels themselves in stream order
    // (tfhe.hip: tfhe_blind_rotate) -- nothing about a buffer is remembered on the host.  A buffer whose header is
    // neither layout makes the kernel set this pinned word; the next entry of the context reports it (TFHE_NEED).
    int* bad_key = nullptr;
    // tfhe/context.cu:39-42: ks_stdev = 2^-15 sqrt(2/pi), bk_stdev = 9e-9 sqrt(2/pi)
    double ks_stdev = (1.0 / 32768.0) * 0.7978845608028654, bk_stdev = 9e-9 * 0.7978845608028654;
};

static ulonglong2 fp_pair(u64 w, u64 q)
{
    const double wd = (double) w, wi = wd / (double) q;
    u64 a, b;
    memcpy(&a, &wd, 8);
    memcpy(&b, &wi, 8);
    return make_ulonglong2(a, b);
}

int hegpu_tfhe_context_create(hegpu_tfhe_context** out)
{
    return guarded([&]() -> int {
        if (!out) throw std::invalid_argument("null output");
        hegpu_tfhe_context* h = new hegpu_tfhe_context();
        const u64 q = 1152921504606877697ULL, psi = 1689264667710614ULL; // tfhe/context.cu:23-24
        const int np = 10;
        TfheDev& p = h->p;
        p.mod = make_mod(q);
        std::vector<u64> fwd = host::power_table_bitrev(psi, q, np);
        std::vector<u64> inv = host::power_table_bitrev(host::inv_mod_prime(psi, q), q, np);
        h->htw.resize(1024);
        h->hitw.resize(1024);
        for (int j = 0; j < 1024; j++) {
            h->htw[j] = make_ulonglong2(fwd[j], shoup_companion(fwd[j], q));
            h->hitw[j] = make_ulonglong2(inv[j], shoup_companion(inv[j], q));
        }
        const u64 ninv = host::inv_mod_prime(1024, q), w1n = host::mul_mod(inv[1], ninv, q);
        p.ninv = make_ulonglong2(ninv, shoup_companion(ninv, q));
        p.w1ninv = make_ulonglong2(w1n, shoup_companion(w1n, q));
        p.n = 512; p.N = 1024; p.k = 1; p.bk_l = 2; p.bk_bg_bit = 10;
        p.half_bg = (1 << p.bk_bg_bit) >> 1;
        p.mask_mod = (1 << p.bk_bg_bit) - 1;
        long long sum = 0; // compute_offset, tfhe/context.cu:70-81
        for (int i = 1; i <= p.bk_l; i++) sum += 1LL << (32 - i * p.bk_bg_bit);
        p.offset = (int) (sum * p.half_bg);
        p.ks_base_bit = 2; p.ks_length = 8;
        // FP64 blind rotate: our own 44-bit NTT prime (tfhe.hip)
        {
            const u64 fq = host::find_primes(1024, std::vector<int>{44})[0];
            const u64 fpsi = host::minimal_primitive_root(2048, fq);
            std::vector<u64> ff = host::power_table_bitrev(fpsi, fq, np);
            std::vector<u64> fi = host::power_table_bitrev(host::inv_mod_prime(fpsi, fq), fq, np);
            h->hftw.resize(1024);
            h->hfitw.resize(1024);
            for (int j = 0; j < 1024; j++) {
                h->hftw[j] = fp_pair(ff[j], fq);
                h->hfitw[j] = fp_pair(fi[j], fq);
            }
            const u64 fn = host::inv_mod_prime(1024, fq);
            p.fprime = fq;
            p.fninv = fp_pair(fn, fq);
            p.fw1ninv = fp_pair(host::mul_mod(fi[1], fn, fq), fq);
        }
        *out = h;
        // defaults only, through the setter's own validation (a value it refuses is ignored, not stored)
        for (const char* nm : {"fp", "ks_batched", "ks_pieces"}) {
            std::string env = std::string("HEGPU_TFHE_") + nm;
            for (char& ch : env) ch = (char) toupper((unsigned char) ch);
            long v;
            if (env_long(env.c_str(), &v) && v >= INT_MIN && v <= INT_MAX) (void) hegpu_tfhe_context_set_option(h, nm, (int) v);
        }
        return 0;
    });
}

int hegpu_tfhe_context_set_option(hegpu_tfhe_context* ctx, const char* name, int value)
{
    if (!ctx || !name) return fail(HEGPU_E_INVALID, "null argument");
    if (!strcmp(name, "fp")) {
        if (value < 0 || value > 1) return fail(HEGPU_E_INVALID, "value out of range for option fp");
        ctx->allow_fp = value != 0;
    } else if (!strcmp(name, "ks_pieces")) {
        if (value < -1 || value == 0 || value > 64) return fail(HEGPU_E_INVALID, "value out of range for option ks_pieces");
        ctx->ks_pieces = value;
    } else if (!strcmp(name, "ks_batched")) {
        if (value < -1 || (value > 1 && value != 8 && value != 12 && value != 16))
            return fail(HEGPU_E_INVALID, "value out of range for option ks_batched");
        ctx->ks_batched = value;
    } else {
        return fail(HEGPU_E_INVALID, std::string("unknown option: ") + name);
    }
    return 0;
}

void hegpu_tfhe_context_destroy(hegpu_tfhe_context* ctx)
{
    if (!ctx) return;
    if (ctx->dtw) (void) hipFree(ctx->dtw);
    if (ctx->ditw) (void) hipFree(ctx->ditw);
    if (ctx->dftw) (void) hipFree(ctx->dftw);
    if (ctx->dfitw) (void) hipFree(ctx->dfitw);
    if (ctx->bad_key) (void) hipHostFree(ctx->bad_key);
    delete ctx;
}

long hegpu_tfhe_context_int(const hegpu_tfhe_context* ctx, const char* name)
{
    if (!ctx || !name) return -1;
    const TfheDev& p = ctx->p;
    if (!strcmp(name, "n")) return p.n;
    if (!strcmp(name, "N")) return p.N;
    if (!strcmp(name, "k")) return p.k;
    if (!strcmp(name, "bk_l")) return p.bk_l;
    if (!strcmp(name, "bk_bg_bit")) return p.bk_bg_bit;
    if (!strcmp(name, "ks_base_bit")) return p.ks_base_bit;
    if (!strcmp(name, "ks_length")) return p.ks_length;
    if (!strcmp(name, "offset")) return p.offset;
    if (!strcmp(name, "bootkey_elems")) return (long) p.n * (p.k + 1) * p.bk_l * (p.k + 1) * p.N;
    if (!strcmp(name, "prepared_bootkey_elems"))
        return (long) TFHE_PREP_HEADER + 2L * p.n * (p.k + 1) * p.bk_l * (p.k + 1) * p.N;
    if (!strcmp(name, "kskey_b_elems")) return (long) p.N * p.k * p.ks_length * ((1 << p.ks_base_bit) - 1);
    if (!strcmp(name, "kskey_a_elems")) return (long) p.N * p.k * p.ks_length * ((1 << p.ks_base_bit) - 1) * p.n;
    return -1;
}

uint64_t hegpu_tfhe_prime(const hegpu_tfhe_context* ctx) { return ctx ? ctx->p.mod.q : 0; }

static int tfhe_need(hegpu_tfhe_context* ctx)
{
    if (!ctx) return fail(HEGPU_E_INVALID, "null context");
    if (ctx->uploaded) return 0; // (the bad-key flag is NOT looked at here: hegpu_tfhe_status / the bootstrapping entries own it)
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        (void) hipGetLastError();
        return fail(HEGPU_E_NODEVICE, "no HIP device available: the HIP backend cannot run (no CPU fallback)");
    }
    hipError_t e;
    if ((e = hipMalloc((void**) &ctx->dtw, 1024 * sizeof(ulonglong2))) != hipSuccess) return hip_ret(e, "tfhe upload");
    if ((e = hipMalloc((void**) &ctx->ditw, 1024 * sizeof(ulonglong2))) != hipSuccess) return hip_ret(e, "tfhe upload");
    (void) hipMemcpy(ctx->dtw, ctx->htw.data(), 1024 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    (void) hipMemcpy(ctx->ditw, ctx->hitw.data(), 1024 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    if ((e = hipMalloc((void**) &ctx->dftw, 1024 * sizeof(ulonglong2))) != hipSuccess) return hip_ret(e, "tfhe upload");
    if ((e = hipMalloc((void**) &ctx->dfitw, 1024 * sizeof(ulonglong2))) != hipSuccess) return hip_ret(e, "tfhe upload");
    (void) hipMemcpy(ctx->dftw, ctx->hftw.data(), 1024 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    (void) hipMemcpy(ctx->dfitw, ctx->hfitw.data(), 1024 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    ctx->p.tw = ctx->dtw;
    ctx->p.itw = ctx->ditw;
    ctx->p.ftw = ctx->dftw;
    ctx->p.fitw = ctx->dfitw;
    if ((e = hipHostMalloc((void**) &ctx->bad_key, sizeof(int), hipHostMallocMapped)) != hipSuccess) return hip_ret(e, "tfhe upload");
    *ctx->bad_key = 0;
    ctx->p.bad_key = ctx->bad_key;
    (void) hipGetDevice(&ctx->device);
    ctx->uploaded = true;
    return 0;
}
// The blind rotate reads the prepared key's layout ON THE DEVICE; a buffer that is no prepared key makes the kernel set the
// context's pinned flag and write nothing.  The flag is context-wide (not per stream) and is visible once the kernel that
// set it has finished.  Who consumes it (round 5 let ANY later entry of the context consume it -- an unrelated
// encrypt, or a nested call inside hegpu_tfhe_gate, reported and swallowed it):
//   * hegpu_tfhe_status(ctx, stream): drains `stream`, returns HEGPU_E_INVALID if the flag is set and clears it -- the way
//     to learn about a bad call AT that call (call + status), as the reference checks after each launch (util.cuh:47-55);
//   * the entry of hegpu_tfhe_bootstrapping / _gate / _mux themselves, before they queue anything: a caller that never
//     asks still hears about it at its next bootstrapping, and no gate is left half-queued.
static int tfhe_take_bad_key(hegpu_tfhe_context* ctx)
{
    if (ctx->bad_key && __atomic_exchange_n(ctx->bad_key, 0, __ATOMIC_ACQ_REL))
        return fail(HEGPU_E_INVALID, "a bootstrapping call of this context was handed a buffer that is not a prepared boot "
                                     "key (header word neither 0 nor 1): its outputs were not written");
    return 0;
}
#define TFHE_NEED(ctx)        \
    int r = tfhe_need(ctx);   \
    if (r) return r;          \
    DevGuard dev_guard__((ctx)->device); \
    if (dev_guard__.err != hipSuccess) return hip_ret(dev_guard__.err, "switching to the TFHE context's device")

// tfhe/operator.cu:317-323
static int32_t encode_to_torus32(uint32_t mu, uint32_t m_size)
{
    uint64_t interval = ((1ULL << 63) / m_size) * 2;
    uint64_t phase64 = mu * interval;
    return (int32_t) (phase64 >> 32);
}

int hegpu_tfhe_prepare_bootkey(hegpu_tfhe_context* ctx, const uint64_t* boot_key, uint64_t* prepared,
                               hegpu_stream stream)
{
    TFHE_NEED(ctx);
    const TfheDev& p = ctx->p;
    const u64 polys = (u64) p.n * (p.k + 1) * p.bk_l * (p.k + 1);
    int fmt = -1;
    return hip_ret(tfhe_prepare_bootkey(p, (const u64*) boot_key, (u64*) prepared, polys, ctx->allow_fp, &fmt,
                                        (hipStream_t) stream),
                   "hegpu_tfhe_prepare_bootkey");
}

// A query, not part of any launch path: the header word of a prepared key, read after the device has drained (so that it
// is ordered behind whatever stream wrote the buffer).  Every failure is -1 (error codes leaked out as "formats").
int hegpu_tfhe_prepared_format(hegpu_tfhe_context* ctx, const uint64_t* prepared, int refresh)
{
    (void) refresh; // nothing is remembered any more (kept for the ABI)
    if (tfhe_need(ctx)) return -1;
    DevGuard dev_guard__(ctx->device);
    if (dev_guard__.err != hipSuccess) {
        (void) hip_ret(dev_guard__.err, "switching to the TFHE context's device");
        return -1;
    }
    if (!prepared) {
        (void) fail(HEGPU_E_INVALID, "null prepared boot key");
        return -1;
    }
    uint64_t w = ~0ULL;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(&w, prepared, sizeof(w), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void) hip_ret(e, "reading the prepared boot key's header");
        return -1;
    }
    if (w > 1) {
        (void) fail(HEGPU_E_INVALID, "not a prepared boot key (header word is neither 0 nor 1)");
        return -1;
    }
    return (int) w;
}

int hegpu_tfhe_gate_precompute(hegpu_tfhe_context* ctx, int gate, int32_t* out_a, int32_t* out_b,
                               const int32_t* a1, const int32_t* b1, const int32_t* a2, const int32_t* b2,
                               int shape, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if (shape <= 0) return 0; // an empty batch is a no-op (and the cheapest first call: it places the context's tables)
    int enc, s1, s2, m = 1;
    const int e8 = encode_to_torus32(1, 8), e4 = encode_to_torus32(1, 4);
    switch (gate) { // tfhe/operator.cu:24-198
        case HEGPU_GATE_NAND: enc = e8; s1 = -1; s2 = -1; break;
        case HEGPU_GATE_AND: enc = -e8; s1 = 1; s2 = 1; break;
        case HEGPU_GATE_AND_FIRST_NOT: enc = -e8; s1 = -1; s2 = 1; break;
        case HEGPU_GATE_NOR: enc = -e8; s1 = -1; s2 = -1; break;
        case HEGPU_GATE_OR: enc = e8; s1 = 1; s2 = 1; break;
        case HEGPU_GATE_XNOR: enc = -e4; s1 = -1; s2 = -1; m = 2; break;
        case HEGPU_GATE_XOR: enc = e4; s1 = 1; s2 = 1; m = 2; break;
        case HEGPU_GATE_NOT: enc = 0; s1 = -1; s2 = 1; a2 = nullptr; b2 = nullptr; break;
        default: return fail(HEGPU_E_INVALID, "unknown gate");
    }
    return hip_ret(tfhe_gate_pre(out_a, out_b, a1, b1, a2, b2, enc, s1, s2, m, ctx->p.n, shape, (hipStream_t) stream),
                   "hegpu_tfhe_gate_precompute");
}

int hegpu_tfhe_status(hegpu_tfhe_context* ctx, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    hipError_t e = hipStreamSynchronize((hipStream_t) stream);
    if (e != hipSuccess) return hip_ret(e, "hegpu_tfhe_status");
    return tfhe_take_bad_key(ctx);
}

// (no look at the flag: the nested calls of hegpu_tfhe_gate / _mux)
static int tfhe_bootstrapping_queue(hegpu_tfhe_context* ctx, const int32_t* in_a, const int32_t* in_b,
                                    const uint64_t* prepared_boot_key, int32_t* out_a, int32_t* out_b, int shape,
                                    hegpu_stream stream)
{
    if (shape <= 0) return 0;
    if (!prepared_boot_key) return fail(HEGPU_E_INVALID, "null prepared boot key");
    return hip_ret(tfhe_blind_rotate(ctx->p, in_a, in_b, (const u64*) prepared_boot_key, out_a, out_b,
                                     encode_to_torus32(1, 8), shape, (hipStream_t) stream),
                   "hegpu_tfhe_bootstrapping");
}

int hegpu_tfhe_bootstrapping(hegpu_tfhe_context* ctx, const int32_t* in_a, const int32_t* in_b,
                             const uint64_t* prepared_boot_key, int32_t* out_a, int32_t* out_b, int shape,
                             hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if ((r = tfhe_take_bad_key(ctx))) return r; // an earlier call's bad key, before anything is queued
    return tfhe_bootstrapping_queue(ctx, in_a, in_b, prepared_boot_key, out_a, out_b, shape, stream);
}

int hegpu_tfhe_key_switching(hegpu_tfhe_context* ctx, const int32_t* in_a, const int32_t* in_b, int32_t* out_a,
                             int32_t* out_b, const int32_t* ks_a, const int32_t* ks_b, int shape,
                             hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if (shape > 0) {
        if (!in_a || !in_b || !out_a || !out_b || !ks_a || !ks_b) return fail(HEGPU_E_INVALID, "hegpu_tfhe_key_switching: null pointer");
        // the split forms zero the outputs before any input is read (tfhe.hip): in and out must be distinct buffers
        // (address ranges compared as integers: the pointers may belong to unrelated allocations)
        const uintptr_t ia = (uintptr_t) in_a, ib = (uintptr_t) in_b, oa = (uintptr_t) out_a, ob = (uintptr_t) out_b;
        const size_t ia_n = (size_t) shape * ctx->p.k * ctx->p.N * 4, ib_n = (size_t) shape * 4, oa_n = (size_t) shape * ctx->p.n * 4;
        auto hit = [](uintptr_t a, size_t an, uintptr_t b, size_t bn) { return a < b + bn && b < a + an; };
        if (hit(ia, ia_n, oa, oa_n) || hit(ia, ia_n, ob, ib_n) || hit(ib, ib_n, oa, oa_n) || hit(ib, ib_n, ob, ib_n))
            return fail(HEGPU_E_INVALID, "hegpu_tfhe_key_switching: the output sample overlaps the input sample");
    }
    return hip_ret(tfhe_key_switching(ctx->p, in_a, in_b, out_a, out_b, ks_a, ks_b, shape, ctx->ks_batched, ctx->ks_pieces, (hipStream_t) stream),
                   "hegpu_tfhe_key_switching");
}

int hegpu_tfhe_gate(hegpu_tfhe_context* ctx, int gate, const int32_t* in1_a, const int32_t* in1_b,
                    const int32_t* in2_a, const int32_t* in2_b, int32_t* out_a, int32_t* out_b,
                    const uint64_t* prepared_boot_key, const int32_t* ks_a, const int32_t* ks_b, int shape, void* ws,
                    size_t ws_bytes, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    const TfheDev& p = ctx->p;
    if (gate == HEGPU_GATE_NOT) // NOT needs no bootstrapping (tfhe/operator.cuh:640-686)
        return hegpu_tfhe_gate_precompute(ctx, gate, out_a, out_b, in1_a, in1_b, nullptr, nullptr, shape, stream);
    if ((r = tfhe_take_bad_key(ctx))) return r; // an earlier call's bad key, before anything of this gate is queued
    const size_t need = ((size_t) p.n + (size_t) p.k * p.N + 2) * shape * sizeof(int32_t);
    if (!ws || ws_bytes < need) return fail(HEGPU_E_INVALID, "workspace too small");
    int32_t* t_a = (int32_t*) ws;
    int32_t* t_b = t_a + (size_t) p.n * shape;
    int32_t* e_a = t_b + shape;
    int32_t* e_b = e_a + (size_t) p.k * p.N * shape;
    if ((r = hegpu_tfhe_gate_precompute(ctx, gate, t_a, t_b, in1_a, in1_b, in2_a, in2_b, shape, stream))) return r;
    if ((r = tfhe_bootstrapping_queue(ctx, t_a, t_b, prepared_boot_key, e_a, e_b, shape, stream))) return r;
    return hegpu_tfhe_key_switching(ctx, e_a, e_b, out_a, out_b, ks_a, ks_b, shape, stream);
}

// ---- TFHE front end (reference tfhe/keygenerator.cu, encryptor.cu, decryptor.cu, operator.cuh:676-800)
static const double TFHE_IH = 1.1547005383792517; // sqrt(16/12), see drbg_torus_gaussian

int hegpu_tfhe_generate_secret_key(hegpu_tfhe_context* ctx, hegpu_rng* rng, int32_t* lwe_key, int32_t* tlwe_key,
                                   hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if (!rng) return fail(HEGPU_E_INVALID, "null random generator");
    const u64 s0 = rng->r.stream;
    rng->r.stream += 2;
    return hip_ret(tfhe_gen_secret(lwe_key, tlwe_key, ctx->p.n, ctx->p.k * ctx->p.N, rng->r.seed, s0,
                                   (hipStream_t) stream),
                   "hegpu_tfhe_generate_secret_key");
}

int hegpu_tfhe_generate_bootstrapping_key(hegpu_tfhe_context* ctx, hegpu_rng* rng, const int32_t* lwe_key,
                                          const int32_t* tlwe_key, uint64_t* boot_key, int32_t* ks_a, int32_t* ks_b,
                                          void* ws, size_t ws_bytes, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if (!rng) return fail(HEGPU_E_INVALID, "null random generator");
    const TfheDev& p = ctx->p;
    if (!ws || ws_bytes < (size_t) p.N * sizeof(u64)) return fail(HEGPU_E_INVALID, "workspace too small");
    const u64 s0 = rng->r.stream;
    rng->r.stream += 4;
    hipError_t e = tfhe_gen_bootkey(p, (u64*) boot_key, lwe_key, tlwe_key, (u64*) ws, ctx->bk_stdev / TFHE_IH,
                                    rng->r.seed, s0, s0 + 1, (hipStream_t) stream);
    if (e != hipSuccess) return hip_ret(e, "hegpu_tfhe_generate_bootstrapping_key");
    const u64 rows = (u64) p.N * p.k * p.ks_length * ((1 << p.ks_base_bit) - 1);
    return hip_ret(tfhe_lwe_encrypt(ks_a, ks_b, lwe_key, nullptr, 1, tlwe_key, p.ks_base_bit, p.ks_length, p.n, rows,
                                    ctx->ks_stdev / TFHE_IH, rng->r.seed, s0 + 2, s0 + 3, (hipStream_t) stream),
                   "hegpu_tfhe_generate_bootstrapping_key");
}

int hegpu_tfhe_encrypt(hegpu_tfhe_context* ctx, hegpu_rng* rng, const int32_t* lwe_key, const int32_t* messages,
                       int shape, int32_t* out_a, int32_t* out_b, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    if (!rng) return fail(HEGPU_E_INVALID, "null random generator");
    if (shape <= 0) return fail(HEGPU_E_INVALID, "shape must be positive");
    const u64 s0 = rng->r.stream;
    rng->r.stream += 2;
    return hip_ret(tfhe_lwe_encrypt(out_a, out_b, lwe_key, messages, 0, nullptr, 0, 1, ctx->p.n, (u64) shape,
                                    ctx->ks_stdev / TFHE_IH, rng->r.seed, s0, s0 + 1, (hipStream_t) stream),
                   "hegpu_tfhe_encrypt");
}

int hegpu_tfhe_decrypt_phase(hegpu_tfhe_context* ctx, const int32_t* lwe_key, const int32_t* a, const int32_t* b,
                             int shape, int32_t* phase, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    return hip_ret(tfhe_lwe_phase(a, b, lwe_key, phase, ctx->p.n, shape, (hipStream_t) stream),
                   "hegpu_tfhe_decrypt_phase");
}

int hegpu_tfhe_mux(hegpu_tfhe_context* ctx, const int32_t* in1_a, const int32_t* in1_b, const int32_t* in2_a,
                   const int32_t* in2_b, const int32_t* c_a, const int32_t* c_b, int32_t* out_a, int32_t* out_b,
                   const uint64_t* prepared_boot_key, const int32_t* ks_a, const int32_t* ks_b, int shape, void* ws,
                   size_t ws_bytes, hegpu_stream stream)
{
    TFHE_NEED(ctx);
    const TfheDev& p = ctx->p;
    const size_t kN = (size_t) p.k * p.N;
    const size_t need = ((size_t) p.n + 1 + 2 * (kN + 1)) * shape * sizeof(int32_t);
    if (!ws || ws_bytes < need) return fail(HEGPU_E_INVALID, "workspace too small");
    int32_t* t_a = (int32_t*) ws;
    int32_t* t_b = t_a + (size_t) p.n * shape;
    int32_t* e1_a = t_b + shape;
    int32_t* e1_b = e1_a + kN * shape;
    int32_t* e2_a = e1_b + shape;
    int32_t* e2_b = e2_a + kN * shape;
    if ((r = tfhe_take_bad_key(ctx))) return r;
    // AND(c, in1) and AND(NOT c, in2), bootstrapped; OR of the two extracted samples; key switch
    if ((r = hegpu_tfhe_gate_precompute(ctx, HEGPU_GATE_AND, t_a, t_b, c_a, c_b, in1_a, in1_b, shape, stream))) return r;
    if ((r = tfhe_bootstrapping_queue(ctx, t_a, t_b, prepared_boot_key, e1_a, e1_b, shape, stream))) return r;
    if ((r = hegpu_tfhe_gate_precompute(ctx, HEGPU_GATE_AND_FIRST_NOT, t_a, t_b, c_a, c_b, in2_a, in2_b, shape,
                                        stream)))
        return r;
    if ((r = tfhe_bootstrapping_queue(ctx, t_a, t_b, prepared_boot_key, e2_a, e2_b, shape, stream))) return r;
    hipError_t e = tfhe_gate_pre(e1_a, e1_b, e1_a, e1_b, e2_a, e2_b, encode_to_torus32(1, 8), 1, 1, 1, (int) kN, shape,
                                 (hipStream_t) stream); // OR_pre_computation on the N-dimensional samples
    if (e != hipSuccess) return hip_ret(e, "hegpu_tfhe_mux");
    return hegpu_tfhe_key_switching(ctx, e1_a, e1_b, out_a, out_b, ks_a, ks_b, shape, stream);
}

} // extern "C"

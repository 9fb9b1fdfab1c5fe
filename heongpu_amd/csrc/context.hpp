// context.hpp -- parameter set + device-resident tables of one HE context.
//
// Host-side counterpart of the reference's HEContextImpl<BFV|CKKS>::generate()
// (reference src/lib/host/ckks/context.cu:272-440, bfv/context.cu:396-705):
// prime chain, psi, NTT tables, mod-down / rescale constants, the CKKS order
// tables (ckks/operator.cu:24-56) and the BFV BEHZ constants
// (bfv/context.cu:939-1347).  Host tables are kept under the reference's
// member names so tests can compare them one by one.
#pragma once
#include "modarith.cuh"
#include "ntt.hpp"
#include "rns.hpp"
#include <map>
#include <string>
#include "drbg.hpp"
#include <vector>

namespace hegpu {

// context.cpp: the value of an environment variable that holds a whole decimal integer (false otherwise)
bool env_long(const char* name, long* out);


enum Scheme { SCHEME_BFV = 1, SCHEME_CKKS = 2 };

struct NttPlan {
    // device arrays, one entry / table per modulus
    Mod* mods = nullptr;
    ulonglong2* tw = nullptr;
    ulonglong2* itw = nullptr;
    ulonglong2* twB = nullptr;  // row-pass last-four-stages layout (see ntt.hpp)
    ulonglong2* itwB = nullptr;
    double* twB8 = nullptr;     // twB's layout as plain doubles (FP64 moduli; zeros elsewhere): 8 bytes per twiddle for the row pass
    ulonglong2* ninv = nullptr;
    ulonglong2* w1ninv = nullptr;
    int count = 0;
    int has_fp = 0, has_int = 0; // moduli on the FP64 / on the integer butterflies
    std::vector<unsigned char> fp; // host copy of Mod::fp per modulus
    // integer butterflies: moduli up to this run the whole forward transform correction-free (NttArgs::lazy_q_max)
    u64 lazy_q_max = 0;
};

struct Context {
    int scheme = 0;
    int n_power = 0;
    u64 n = 0;
    int Q_size = 0, P_size = 0, Qp_size = 0;
    int bsk_size = 0;
    u64 plain_modulus = 0;
    std::vector<u64> primes; // Q then P
    std::map<std::string, std::vector<u64>> host; // named host tables
    // key-switching method II (P_size > 1): per-depth digit partition + table
    // offsets into the flattened host/device arrays "m2_*"
    struct M2Level {
        int d = 0, rc = 0;
        int off_digits = 0; // into m2_I_j / m2_I_location
        int off_mi = 0;     // into m2_Mi_inv
        int off_matrix = 0; // into m2_matrix
        int off_prod = 0;   // into m2_prod
    };
    std::vector<M2Level> m2_levels;
    int m2_width = 0; // digit width m: 2 for BFV, P_size for CKKS
    // use the fused "row pass + key-switch MAC" kernel (HEGPU_FUSED_ROW_MAC=0 disables)
    int fused_row_mac = -1;    // HEGPU_FUSED_ROW_MAC: 1 / 0 force the fused / the reference's key-switch sequence, otherwise by launch size (ops.cpp: use_fused_row_mac)
    bool fused_moddown = true; // HEGPU_FUSED_MODDOWN=0: separate stage-two kernel
    int col_multi = -1;        // HEGPU_COL_MULTI: form of the decomposing column pass (NttArgs::col_multi)
    int single_pass = -1;      // HEGPU_SINGLE_PASS: 1 / 0 force the single pass / the two passes for N <= 2^14, otherwise by launch size (NttArgs::single_pass)
    bool ntt_galois = true;    // HEGPU_NTT_GALOIS=0: CKKS rotations in the reference's order (permutation in the coefficient domain)
    bool galois_scatter = true; // HEGPU_GALOIS_SCATTER=0: the NTT-domain permutation as a kernel of its own (gather) instead of the mod-down epilogue's store
    int digit_split = -1;      // HEGPU_DIGIT_SPLIT: 0 never, 2 / 4 always that many workgroups per fused key-switch unit, -1 by launch size
    bool copy_along = true;    // HEGPU_COPY_ALONG=0: the rescale's copy of the kept limbs always has its own launch
    bool fuse_inverse = true;  // HEGPU_FUSE_INVERSE=0: the INTT feeding a decomposing launch runs on its own
    bool fused_tensor = true;  // BFV multiply: the tensor product as the load transform of the inverse transform (0: its own kernel)
    bool fp_ntt = true;        // fp_ntt = 0: every modulus on the integer butterflies (read when the tables are built)
    int behz_split = -1;       // BFV BEHZ kernels: rows over four wavefronts (1), one thread per coefficient (0), by launch size (-1)
    // The fields above are options: hegpu_context_set_option (include/hegpu.h); the environment variables named in
    // the comments only seed their defaults when a context is created (seed_options_from_env).
    void seed_options_from_env();
    // 0 ok, 1 unknown name, 2 value out of range, 3 too late (the tables are already on the device)
    int set_option(const char* name, int value);
    int get_option(const char* name, int* value) const;
    GaussCdt gauss_cdt{}; // rounded Gaussian, sigma = 3.2 (drbg.hpp)

    // ---- device state (valid after upload())
    bool uploaded = false;
    int device = -1;
    NttPlan plan_qp;    // tables for the Q' chain
    NttPlan plan_merge; // BFV: [q_0..q_{Q-1}, Bsk...]
    NttPlan plan_plain; // BFV batching: the plain modulus t alone (bfv/context.cu:489-499), when 2N | t-1
    std::map<std::string, void*> dev; // named device arrays (u64 / int)
    Mod* bsk_mods = nullptr;
    BehzDev behz{};

    ~Context();
    void build_host();           // derive every host table from `primes`
    hipError_t upload();         // allocate + copy device tables (current device)
    void release_device();

    const u64* d64(const char* name) const;
    // host copy of a named table (throws if absent)
    const std::vector<u64>& h64(const char* name) const { return host.at(name); }
    const int* d32(const char* name) const;
    NttArgs ntt_args(int table_set) const; // 0 = Q' chain, 1 = merged q|Bsk
};

} // namespace hegpu

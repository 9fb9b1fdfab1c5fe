// host_params.hpp -- host-side parameter derivation for the HIP backend.
//
// Product code (NOT the oracle): an independent implementation of the
// reference's deterministic parameter generators so that every constant the
// kernels read is identical to the reference's:
//   prime chains        reference src/lib/util/util.cu:219-310
//   minimal 2N-th root  util.cu:312-380
//   psi power tables    util.cu:398-451 (bit-reversed order)
//   default chains      src/lib/util/defaultmodulus.cpp:12-175
#pragma once
#include <cstdint>
#include <vector>

namespace hegpu {
namespace host {

typedef unsigned long long u64;

u64 mul_mod(u64 a, u64 b, u64 q);
u64 pow_mod(u64 a, u64 e, u64 q);
u64 inv_mod_prime(u64 a, u64 q); // q prime
u64 inv_mod_pow2_32(u64 a);      // inverse modulo 2^32 (a odd)
bool is_prime(u64 v);
// SEAL-style chain: for each distinct bit size scan down from
// floor((2^b-1)/2N)*2N+1 in steps of 2N; hand out smallest-first.
std::vector<u64> find_primes(u64 n, const std::vector<int>& bit_sizes);
std::vector<u64> internal_primes(u64 n, int count); // 61-bit
u64 minimal_primitive_root(u64 degree, u64 q);
// out[j] = base^bitreverse(j, n_power) mod q
std::vector<u64> power_table_bitrev(u64 base, u64 q, int n_power);
// default chain for n in {4096..65536} at 128 / 192 / 256-bit security; empty if none
std::vector<u64> default_chain(u64 n, int sec_level);
std::vector<u64> default_chain_128(u64 n);
// max total coefficient-modulus bits (reference secstdparams.h:25-79); 0 if none
int max_logq(u64 n, int sec_level);
int max_logq_128(u64 n);
int steps_to_galois_elt(int steps, int n, int group_order);

} // namespace host
} // namespace hegpu

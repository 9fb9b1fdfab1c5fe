// host_params.cpp -- see host_params.hpp.  Product code, independent of oracle/.
#include "host_params.hpp"
#include <algorithm>
#include <map>
#include <stdexcept>

namespace hegpu {
namespace host {

typedef unsigned __int128 u128;

u64 mul_mod(u64 a, u64 b, u64 q) { return (u64) (((u128) a * b) % q); }

u64 pow_mod(u64 a, u64 e, u64 q)
{
    u64 acc = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) acc = mul_mod(acc, a, q);
        a = mul_mod(a, a, q);
    }
    return acc;
}

u64 inv_mod_prime(u64 a, u64 q) { return pow_mod(a, q - 2, q); }

// Newton iteration for the inverse of an odd number modulo 2^32
// (reference bfv/context.cu:1037-1053 uses an extended gcd for m_tilde = 2^32).
u64 inv_mod_pow2_32(u64 a)
{
    u64 x = a; // correct to 3 bits for odd a
    for (int i = 0; i < 5; i++) x *= 2 - a * x;
    return x & 0xFFFFFFFFULL;
}

// deterministic Miller-Rabin, exact for 64-bit inputs (the reference draws
// random witnesses, util.cu:127-166; the verdict is the same)
bool is_prime(u64 v)
{
    if (v < 2) return false;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : small) {
        if (v == p) return true;
        if (v % p == 0) return false;
    }
    u64 d = v - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (u64 a : small) {
        u64 x = pow_mod(a, d, v);
        if (x == 1 || x == v - 1) continue;
        bool composite = true;
        for (int i = 1; i < r; i++) {
            x = mul_mod(x, x, v);
            if (x == v - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

std::vector<u64> find_primes(u64 n, const std::vector<int>& bit_sizes)
{
    std::map<int, int> want;
    for (int b : bit_sizes) {
        if (b < 30 || b > 61) throw std::logic_error("invalid modulus bit size");
        ++want[b];
    }
    const u64 step = 2 * n;
    std::map<int, std::vector<u64>> pool; // descending per bit size
    for (auto& kv : want) {
        int b = kv.first;
        u64 cand = ((((u64) 1) << b) - 1) / step * step + 1;
        u64 floor_ = ((u64) 1) << (b - 1);
        std::vector<u64>& dst = pool[b];
        while ((int) dst.size() < kv.second && cand > floor_) {
            if (is_prime(cand)) dst.push_back(cand);
            cand -= step;
        }
        if ((int) dst.size() < kv.second)
            throw std::logic_error("failed to find enough qualifying primes");
    }
    std::vector<u64> chain;
    for (int b : bit_sizes) {
        chain.push_back(pool[b].back());
        pool[b].pop_back();
    }
    return chain;
}

std::vector<u64> internal_primes(u64 n, int count)
{
    return find_primes(n, std::vector<int>(count, 61));
}

u64 minimal_primitive_root(u64 degree, u64 q)
{
    if ((q - 1) % degree) throw std::logic_error("no sufficient root unity");
    const u64 cofactor = (q - 1) / degree;
    u64 root = 0;
    for (u64 g = 2; g < q && !root; ++g) {
        u64 c = pow_mod(g, cofactor, q);
        if (pow_mod(c, degree / 2, q) == q - 1) root = c;
    }
    // all primitive degree-th roots are the odd powers of one of them
    const u64 sq = mul_mod(root, root, q);
    u64 best = root, cur = root;
    for (u64 i = 1; i < degree / 2; ++i) {
        cur = mul_mod(cur, sq, q);
        best = std::min(best, cur);
    }
    return best;
}

static inline u64 bit_reverse(u64 x, int bits)
{
    u64 r = 0;
    for (int i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}

std::vector<u64> power_table_bitrev(u64 base, u64 q, int n_power)
{
    const u64 n = ((u64) 1) << n_power;
    std::vector<u64> out(n);
    u64 p = 1;
    for (u64 e = 0; e < n; ++e) {
        out[bit_reverse(e, n_power)] = p;
        p = mul_mod(p, base, q);
    }
    return out;
}

// default prime chains per security level (reference src/lib/util/defaultmodulus.cpp:12-175; the
// values are the reference's constants -- tests/golden/reference_constants.json holds them as extracted
// from the reference tree and tests/test_oracle_golden.py compares)
std::vector<u64> default_chain(u64 n, int sec_level)
{
    if (sec_level == 128) {
        switch (n) {
            case 4096:
                return {0x800004001ULL, 0x800008001ULL, 0x1000002001ULL};
            case 8192:
                return {0x40000084001ULL, 0x400000b0001ULL, 0x8000002c001ULL, 0x80000050001ULL, 0x80000064001ULL};
            case 16384:
                return {0x800000020001ULL, 0x8000001a8001ULL, 0x8000001e8001ULL, 0x10000000d8001ULL,
                        0x1000000168001ULL, 0x10000001a0001ULL, 0x10000001e0001ULL, 0x10000002b8001ULL,
                        0x10000002e8001ULL};
            case 32768:
                return {0x2000000002b0001ULL, 0x2000000003a0001ULL, 0x2000000005b0001ULL, 0x200000000640001ULL,
                        0x400000000270001ULL, 0x400000000350001ULL, 0x400000000360001ULL, 0x4000000004d0001ULL,
                        0x400000000570001ULL, 0x400000000660001ULL, 0x4000000008a0001ULL, 0x400000000920001ULL,
                        0x400000000980001ULL, 0x400000000990001ULL, 0x400000000a40001ULL};
            case 65536:
                return {0x2000000003a0001ULL, 0x200000000640001ULL, 0x200000000f80001ULL, 0x200000001460001ULL,
                        0x2000000015a0001ULL, 0x2000000015e0001ULL, 0x200000001b20001ULL, 0x200000001c00001ULL,
                        0x200000001ee0001ULL, 0x400000000360001ULL, 0x400000000660001ULL, 0x4000000008a0001ULL,
                        0x400000000920001ULL, 0x400000000980001ULL, 0x400000000a40001ULL, 0x400000000c00001ULL,
                        0x400000000ea0001ULL, 0x400000001460001ULL, 0x400000001700001ULL, 0x400000001740001ULL,
                        0x4000000017a0001ULL, 0x400000001920001ULL, 0x400000001b00001ULL, 0x400000001b60001ULL,
                        0x400000001c40001ULL, 0x400000001ee0001ULL, 0x400000001f20001ULL, 0x4000000020c0001ULL,
                        0x400000002360001ULL, 0x400000002480001ULL};
        }
    } else if (sec_level == 192) {
        switch (n) {
            case 4096:
                return {0x1000002001ULL, 0x1000042001ULL};
            case 8192:
                return {0x100008c001ULL, 0x1000090001ULL, 0x10000c8001ULL, 0x2000088001ULL};
            case 16384:
                return {0x20000000b0001ULL, 0x2000000178001ULL, 0x20000001a0001ULL, 0x2000000208001ULL,
                        0x20000003b0001ULL, 0x20000003c8001ULL};
            case 32768:
                return {0x40000000120001ULL, 0x400000001d0001ULL, 0x400000002c0001ULL, 0x40000000480001ULL,
                        0x40000000540001ULL, 0x400000005c0001ULL, 0x400000006c0001ULL, 0x400000007b0001ULL,
                        0x40000000890001ULL, 0x40000000b00001ULL, 0x40000000e40001ULL};
            case 65536:
                return {0x40000000120001ULL, 0x400000002c0001ULL, 0x40000000480001ULL, 0x40000000540001ULL,
                        0x400000005c0001ULL, 0x400000006c0001ULL, 0x40000000b00001ULL, 0x40000000e40001ULL,
                        0x40000000f60001ULL, 0x400000010a0001ULL, 0x400000011a0001ULL, 0x40000001200001ULL,
                        0x40000001340001ULL, 0x400000017a0001ULL, 0x40000001c40001ULL, 0x40000001ca0001ULL,
                        0x40000001d00001ULL, 0x40000002100001ULL, 0x400000022a0001ULL, 0x400000022e0001ULL,
                        0x80000000080001ULL, 0x80000000440001ULL};
        }
    } else if (sec_level == 256) {
        switch (n) {
            case 4096:
                return {0x8008001ULL, 0x10006001ULL};
            case 8192:
                return {0x2000088001ULL, 0x20000e0001ULL, 0x4000038001ULL};
            case 16384:
                return {0x200000008001ULL, 0x2000000a0001ULL, 0x2000000e0001ULL, 0x400000008001ULL,
                        0x400000060001ULL};
            case 32768:
                return {0x4000000120001ULL, 0x40000001b0001ULL, 0x4000000270001ULL, 0x8000000110001ULL,
                        0x8000000130001ULL, 0x80000001c0001ULL, 0x80000002c0001ULL, 0x80000004d0001ULL,
                        0x80000004f0001ULL};
            case 65536:
                return {0x4000000120001ULL, 0x4000000420001ULL, 0x4000000660001ULL, 0x40000007e0001ULL,
                        0x4000000800001ULL, 0x40000008a0001ULL, 0x7fffffffe0001ULL, 0x80000001c0001ULL,
                        0x80000002c0001ULL, 0x8000000500001ULL, 0x8000000820001ULL, 0x8000000940001ULL,
                        0x8000001120001ULL, 0x80000012a0001ULL, 0x8000001360001ULL, 0x80000014c0001ULL,
                        0x8000001540001ULL, 0x8000001600001ULL};
        }
    }
    return {};
}
std::vector<u64> default_chain_128(u64 n) { return default_chain(n, 128); }

// max total coefficient-modulus bits (reference util/secstdparams.h:25-79)
int max_logq(u64 n, int sec_level)
{
    static const int table[3][5] = {{109, 218, 438, 881, 1761}, {74, 149, 300, 605, 1212}, {57, 115, 232, 465, 930}};
    const int row = sec_level == 128 ? 0 : sec_level == 192 ? 1 : sec_level == 256 ? 2 : -1;
    int col = -1;
    switch (n) {
        case 4096: col = 0; break;
        case 8192: col = 1; break;
        case 16384: col = 2; break;
        case 32768: col = 3; break;
        case 65536: col = 4; break;
    }
    return (row < 0 || col < 0) ? 0 : table[row][col];
}
int max_logq_128(u64 n) { return max_logq(n, 128); }

int steps_to_galois_elt(int steps, int n, int group_order)
{
    const int m = 2 * n;
    if (steps == 0) return m - 1;
    int k = steps < 0 ? -steps : steps;
    if (k >= n / 2) return 0; // reference prints an error and returns 0
    if (steps < 0) k = n / 2 - k;
    long long g = 1;
    for (int i = 0; i < k; ++i) g = (g * group_order) & (m - 1);
    return (int) g;
}

} // namespace host
} // namespace hegpu

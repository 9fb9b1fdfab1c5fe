// context.cpp -- see context.hpp.
#include "context.hpp"
#include <cerrno>
#include <climits>
#include <cstdlib>
#include "host_params.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace hegpu {

using host::inv_mod_prime;
using host::mul_mod;

typedef std::vector<u64> vec;

static void append(vec& dst, const vec& src) { dst.insert(dst.end(), src.begin(), src.end()); }

void Context::build_host()
{
    n = ((u64) 1) << n_power;
    Qp_size = Q_size + P_size;
    const int Q = Q_size, P = P_size, Qp = Qp_size;
    if ((int) primes.size() != Qp) throw std::logic_error("prime chain size mismatch");
    host.clear();
    host["modulus"] = primes;

    vec psi, ninv, fwd, inv;
    for (int i = 0; i < Qp; i++) {
        const u64 q = primes[i];
        const u64 r = host::minimal_primitive_root(2 * n, q);
        psi.push_back(r);
        append(fwd, host::power_table_bitrev(r, q, n_power));
        append(inv, host::power_table_bitrev(inv_mod_prime(r, q), q, n_power));
        ninv.push_back(inv_mod_prime(n % q, q));
    }
    host["psi"] = psi;
    host["n_inverse"] = ninv;
    host["ntt_table"] = fwd;
    { // psi^(N/2) per modulus = entry 1 of the bit-reversed forward table (the "i" of mult_i / div_i)
        vec ph;
        for (int j = 0; j < Qp; j++) ph.push_back(fwd[(size_t) j * n + 1]);
        host["psi_half"] = ph;
    }
    host["intt_table"] = inv;

    // mod-down by the special primes, last P prime first (util.cu:701-767)
    vec lqm, half, half_mod, factor;
    for (int i = 0; i < P; i++) {
        const u64 p = primes[Qp - 1 - i];
        half.push_back(p >> 1);
        for (int j = 0; j < (Qp - 1) - i; j++) {
            lqm.push_back(inv_mod_prime(p % primes[j], primes[j]));
            half_mod.push_back((p >> 1) % primes[j]);
        }
        for (int j = 0; j < Q; j++) factor.push_back(p % primes[j]);
    }
    host["last_q_modinv"] = lqm;
    host["half"] = half;
    host["half_mod"] = half_mod;
    host["factor"] = factor;

    if (scheme == SCHEME_CKKS) {
        // per-depth rescale constants (ckks/context.cu:342-368)
        vec r_inv, r_half_mod, r_half;
        for (int d = 0; d < Q - 1; d++) {
            const int last = (Q - 1) - d;
            const u64 ql = primes[last];
            r_half.push_back(ql >> 1);
            for (int i = 0; i < last; i++) {
                r_inv.push_back(inv_mod_prime(ql % primes[i], primes[i]));
                r_half_mod.push_back((ql >> 1) % primes[i]);
            }
        }
        host["rescaled_last_q_modinv"] = r_inv;
        host["rescaled_half_mod"] = r_half_mod;
        host["rescaled_half"] = r_half;
        // ---- decoding: CRT composition tables per level (ckks/context.cu:370-421, util.cu:772-888):
        // for l = Q - depth limbs, Mi[i] = prod_{j != i} q_j and M = prod q_j as l little-endian
        // 64-bit words, Mi_inv[i] = (Mi[i] mod q_i)^-1, upper_half_threshold = (M + 1) >> 1
        {
            auto mul_word = [](std::vector<u64>& big, u64 f) {
                unsigned __int128 carry = 0;
                for (u64& w : big) {
                    unsigned __int128 v = (unsigned __int128) w * f + carry;
                    w = (u64) v;
                    carry = v >> 64;
                }
                if (carry) big.push_back((u64) carry);
            };
            vec Mi, Mi_inv, uht, dmod;
            for (int d = 0; d < Q; d++) {
                const int l = Q - d;
                for (int i = 0; i < l; i++) {
                    std::vector<u64> big{1};
                    u64 m = 1;
                    for (int j = 0; j < l; j++)
                        if (j != i) {
                            mul_word(big, primes[j]);
                            m = mul_mod(m, primes[j] % primes[i], primes[i]);
                        }
                    big.resize(l, 0);
                    Mi.insert(Mi.end(), big.begin(), big.end());
                    Mi_inv.push_back(inv_mod_prime(m, primes[i]));
                }
                std::vector<u64> M{1};
                for (int j = 0; j < l; j++) mul_word(M, primes[j]);
                M.resize(l, 0);
                dmod.insert(dmod.end(), M.begin(), M.end());
                std::vector<u64> h = M; // (M + 1) >> 1
                h.push_back(0);
                for (size_t k = 0; k < h.size(); k++)
                    if (++h[k]) break;
                for (size_t k = 0; k + 1 < h.size(); k++) h[k] = (h[k] >> 1) | (h[k + 1] << 63);
                h.resize(l);
                uht.insert(uht.end(), h.begin(), h.end());
            }
            host["Mi"] = Mi;
            host["Mi_inv"] = Mi_inv;
            host["upper_half_threshold"] = uht;
            host["decryption_modulus"] = dmod;
        }
        // ---- encoding: special-FFT roots in the rotation-group order and the slot bit reversal
        // (ckks/encoder.cu:21-98); doubles stored as bit patterns, (re, im) interleaved
        {
            const u64 slots = n >> 1, M = 2 * n;
            int log_slots = 0;
            while ((1ull << log_slots) < slots) log_slots++;
            const double special_root = 2.0 * M_PI / (double) M;
            std::vector<u64> rot(slots);
            rot[0] = 1;
            for (u64 i = 1; i < slots; i++) rot[i] = (5 * rot[i - 1]) % M;
            vec fr(2 * slots, 0), ir(2 * slots, 0), rev(slots);
            auto put = [](vec& v, u64 at, double re, double im) {
                memcpy(&v[2 * at], &re, 8);
                memcpy(&v[2 * at + 1], &im, 8);
            };
            for (int logm = 1; logm <= log_slots; logm++) {
                const u64 idx_mod = 1ull << (logm + 2), gap = M / idx_mod, offset = 1ull << (logm - 1);
                for (u64 i = 0; i < offset; i++) {
                    // two separate libm calls on purpose (volatile keeps the compiler from merging
                    // them into one sincos, whose last bit may differ): the table must be
                    // reproducible by any other program calling cos() and sin()
                    volatile double ang_c = (double) ((rot[i] % idx_mod) * gap) * special_root;
                    volatile double ang_s = ang_c;
                    const double cr = cos(ang_c), sr = sin(ang_s);
                    put(fr, offset + i, cr, sr);
                    put(ir, offset + i, cr, -sr);
                }
            }
            for (u64 i = 0; i < slots; i++) {
                u64 r = 0;
                for (int b = 0; b < log_slots; b++) r |= ((i >> b) & 1) << (log_slots - 1 - b);
                rev[i] = r;
            }
            host["special_fft_roots_table"] = fr;
            host["special_ifft_roots_table"] = ir;
            host["reverse_order"] = rev;
        }
        // modulus-order / polynomial-order tables (ckks/operator.cu:24-56)
        vec ploc, iloc;
        for (int d = 0; d < Q; d++) {
            for (int j = 0; j < Q - d; j++) ploc.push_back(j);
            for (int j = 0; j < P; j++) ploc.push_back(Q + j);
        }
        for (int i = 0; i < Qp - 1; i++) {
            const int c = Qp - i;
            iloc.push_back(c - 1);
            iloc.push_back(2 * c - 1);
        }
        host["new_prime_locations"] = ploc;
        host["new_input_locations"] = iloc;
    }

    if (P > 1) {
        // KeySwitchParameterGenerator (reference src/lib/kernel/contextpool.cpp:11-438):
        // greedy digits of m primes (m = 2 for BFV, P_size for CKKS), per depth for CKKS
        m2_width = (scheme == SCHEME_BFV) ? 2 : P;
        const int levels = (scheme == SCHEME_BFV) ? 1 : Q;
        vec Ij, Iloc, mi, matrix, prod, matrix_mg, negprod_mg;
        auto R64 = [](u64 m) { return (u64) ((((unsigned __int128) 1) << 64) % m); };
        m2_levels.clear();
        for (int lvl = 0; lvl < levels; lvl++) {
            const int l = Q - lvl, rc = Qp - lvl;
            vec base(primes.begin(), primes.begin() + l);
            base.insert(base.end(), primes.begin() + Q, primes.end());
            M2Level L;
            L.rc = rc;
            L.off_digits = (int) Ij.size();
            L.off_mi = (int) mi.size();
            L.off_matrix = (int) matrix.size();
            L.off_prod = (int) prod.size();
            for (int s0 = 0; s0 < l; s0 += m2_width) {
                const int cnt = std::min(m2_width, l - s0);
                Ij.push_back(cnt);
                Iloc.push_back(s0);
                L.d++;
                for (int k = 0; k < rc; k++)
                    for (int i = 0; i < cnt; i++) {
                        u64 acc = 1;
                        for (int j = 0; j < cnt; j++)
                            if (j != i) acc = mul_mod(acc, base[s0 + j] % base[k], base[k]);
                        matrix.push_back(acc);
                        matrix_mg.push_back(mul_mod(acc, R64(base[k]), base[k]));
                    }
                for (int i = 0; i < cnt; i++) {
                    u64 acc = 1;
                    for (int j = 0; j < cnt; j++)
                        if (j != i) acc = mul_mod(acc, base[s0 + j] % base[s0 + i], base[s0 + i]);
                    mi.push_back(inv_mod_prime(acc, base[s0 + i]));
                }
                for (int k = 0; k < rc; k++) {
                    u64 acc = 1;
                    for (int j = 0; j < cnt; j++) acc = mul_mod(acc, base[s0 + j] % base[k], base[k]);
                    prod.push_back(acc);
                    negprod_mg.push_back(mul_mod((base[k] - acc) % base[k], R64(base[k]), base[k]));
                }
            }
            m2_levels.push_back(L);
        }
        // The mod-down by the P special primes, one at a time (x <- (x - t_i) / P_(i), t_i = lh_i mod q - half_mod_i,
        // P_(i) = primes[Qp-1-i]), flattened for a limb q_y of Q:
        //     x_final = W0 * (x - u),   W0 = prod_i P_(i)^-1,   u = sum_i lh_i * G_i - C,
        //     G_i = prod_{k<i} P_(k),   C = sum_i half_mod_i * G_i                              (all mod q_y)
        // so that u can be transformed on its own and the division become the epilogue of that transform
        // (ops.cpp: ckks_moddown_multi).  Tables indexed by the absolute modulus index y < Q.
        {
            vec W0(Q), G((size_t) Q * P), C(Q);
            for (int y = 0; y < Q; y++) {
                const u64 q = primes[y];
                u64 w = 1, g = 1, c = 0;
                int loc = 0;
                for (int i = 0; i < P; i++) {
                    G[(size_t) y * P + i] = g;
                    c = (c + mul_mod(host["half_mod"][loc + y], g, q)) % q;
                    w = mul_mod(w, host["last_q_modinv"][loc + y], q);
                    g = mul_mod(g, primes[Qp - 1 - i] % q, q);
                    loc += (Qp - 1) - i;
                }
                W0[y] = w;
                C[y] = c;
            }
            host["m2_md_W0"] = W0;
            host["m2_md_G"] = G;
            host["m2_md_C"] = C;
        }
        host["m2_I_j"] = Ij;
        host["m2_I_location"] = Iloc;
        host["m2_Mi_inv"] = mi;
        host["m2_matrix"] = matrix;
        host["m2_prod"] = prod;
        // the same rows with the 2^64 of a Montgomery reduction multiplied in, and -prod likewise: the kernel forms
        // out_k = sum_i y_i M_ik + r (q_k - prod_k) as ONE lazy 128-bit sum and reduces it once (rns.hip)
        host["m2_matrix_mg"] = matrix_mg;
        host["m2_negprod_mg"] = negprod_mg;
    }

    if (scheme == SCHEME_BFV) {
        const u64 t = plain_modulus;
        const u64 mt = ((u64) 1) << 32; // m_tilde (bfv/context.cu:510)
        int total_bits = 0;
        for (u64 q : primes) total_bits += 64 - __builtin_clzll(q);
        const int t_bits = 64 - __builtin_clzll(t);
        int bsk = Qp;
        if (t_bits + total_bits + 32 >= 61 * Q + 61) bsk++; // bfv/context.cu:518-525
        bsk_size = bsk;
        vec ip = host::internal_primes(n, bsk + 1);
        vec B(ip.begin(), ip.begin() + bsk);
        const u64 gamma = ip[bsk];
        const u64 msk = B[bsk - 1];
        host["base_Bsk"] = B;
        host["gamma"] = vec{gamma};
        vec Bpsi;
        for (u64 b : B) Bpsi.push_back(host::minimal_primitive_root(2 * n, b));
        host["base_Bsk_psi"] = Bpsi;

        vec m_q_Bsk, inv_punct, m_mt, inv_mt_B, prod_q_B, inv_prod_q_B, m_B_q, m_msk, inv_punct_B, prod_B_q;
        for (int k = 0; k < bsk; k++)
            for (int i = 0; i < Q; i++) {
                u64 acc = 1;
                for (int j = 0; j < Q; j++)
                    if (j != i) acc = mul_mod(acc, primes[j], B[k]);
                m_q_Bsk.push_back(acc);
            }
        for (int i = 0; i < Q; i++) {
            u64 acc = 1, acc_mt = 1;
            for (int j = 0; j < Q; j++)
                if (j != i) {
                    acc = mul_mod(acc, primes[j] % primes[i], primes[i]);
                    acc_mt = mul_mod(acc_mt, primes[j] % mt, mt);
                }
            inv_punct.push_back(inv_mod_prime(acc, primes[i]));
            m_mt.push_back(acc_mt);
        }
        u64 prod_q_mt = 1;
        for (int i = 0; i < Q; i++) prod_q_mt = mul_mod(prod_q_mt, primes[i] % mt, mt);
        const u64 inv_prod_q_mt = host::inv_mod_pow2_32(prod_q_mt);
        for (int i = 0; i < bsk; i++) {
            inv_mt_B.push_back(inv_mod_prime(mt % B[i], B[i]));
            u64 acc = 1;
            for (int j = 0; j < Q; j++) acc = mul_mod(acc, primes[j], B[i]);
            prod_q_B.push_back(acc);
            inv_prod_q_B.push_back(inv_mod_prime(acc, B[i]));
        }
        for (int k = 0; k < Q; k++)
            for (int i = 0; i < bsk - 1; i++) {
                u64 acc = 1;
                for (int j = 0; j < bsk - 1; j++)
                    if (j != i) acc = mul_mod(acc, B[j] % primes[k], primes[k]);
                m_B_q.push_back(acc);
            }
        for (int i = 0; i < bsk - 1; i++) {
            u64 a1 = 1, a2 = 1;
            for (int j = 0; j < bsk - 1; j++)
                if (j != i) {
                    a1 = mul_mod(a1, B[j], msk);
                    a2 = mul_mod(a2, B[j], B[i]);
                }
            m_msk.push_back(a1);
            inv_punct_B.push_back(inv_mod_prime(a2, B[i]));
        }
        u64 prod_B_msk = 1;
        for (int i = 0; i < bsk - 1; i++) prod_B_msk = mul_mod(prod_B_msk, B[i], msk);
        for (int i = 0; i < Q; i++) {
            u64 acc = 1;
            for (int j = 0; j < bsk - 1; j++) acc = mul_mod(acc, B[j] % primes[i], primes[i]);
            prod_B_q.push_back(acc);
        }
        host["base_change_matrix_Bsk"] = m_q_Bsk;
        host["inv_punctured_prod_mod_base_array"] = inv_punct;
        host["base_change_matrix_m_tilde"] = m_mt;
        host["inv_prod_q_mod_m_tilde"] = vec{inv_prod_q_mt};
        host["inv_m_tilde_mod_Bsk"] = inv_mt_B;
        host["prod_q_mod_Bsk"] = prod_q_B;
        host["inv_prod_q_mod_Bsk"] = inv_prod_q_B;
        host["base_change_matrix_q"] = m_B_q;
        host["base_change_matrix_msk"] = m_msk;
        host["inv_punctured_prod_mod_B_array"] = inv_punct_B;
        host["inv_prod_B_mod_m_sk"] = vec{inv_mod_prime(prod_B_msk, msk)};
        host["prod_B_mod_q"] = prod_B_q;
        // Products of constants that the reference multiplies in one after the other
        // (multiplication.cu:37-41, 160-164, 196-205): one Barrett product per coefficient instead of two.
        // The values are canonical residues either way, so the kernels' outputs do not change.
        {
            vec fc_in, ff_in, ff_mid;
            const u64 mt = ((u64) 1) << 32; // m_tilde
            for (int i = 0; i < Q; i++) {
                fc_in.push_back(mul_mod(mt % primes[i], inv_punct[i], primes[i]));
                ff_in.push_back(mul_mod(plain_modulus % primes[i], inv_punct[i], primes[i]));
            }
            for (int i = 0; i < bsk - 1; i++) ff_mid.push_back(mul_mod(inv_prod_q_B[i], inv_punct_B[i], B[i]));
            host["behz_mtilde_inv_punct"] = fc_in;
            host["behz_t_inv_punct"] = ff_in;
            host["behz_invq_inv_punct_B"] = ff_mid;
            vec msk_q;
            for (int i = 0; i < Q; i++) msk_q.push_back(msk % primes[i]);
            host["behz_msk_mod_q"] = msk_q;
            // whole rows with their trailing factors multiplied in (rns.hpp BehzDev::fc_matrix ...): a row of either base
            // conversion becomes one lazy 128-bit sum and one reduction; every value the kernels store is the canonical
            // residue of the same integer expression as in the reference (multiplication.cu:37-90, 160-205)
            // ... and every such table carries a factor 2^64 (mod its modulus): the kernels bring a lazy sum down with a
            // Montgomery reduction (modarith.cuh redc128), which divides by 2^64
            auto R = [](u64 m) { return (u64) ((((unsigned __int128) 1) << 64) % m); };
            vec fc_m, fc_c1, ff_m, ff_tc, ff_qm, ff_mskm, ff_pB, ff_npB;
            for (int i = 0; i < bsk; i++) {
                const u64 p = B[i], r = R(p);
                const u64 c = i < bsk - 1 ? ff_mid[i] : inv_prod_q_B[i];
                for (int j = 0; j < Q; j++) {
                    const u64 m = m_q_Bsk[(size_t) i * Q + j];
                    fc_m.push_back(mul_mod(mul_mod(m, inv_mt_B[i], p), r, p));
                    ff_m.push_back(mul_mod((p - mul_mod(m, c, p)) % p, r, p));
                }
                fc_c1.push_back(mul_mod(mul_mod(prod_q_B[i], inv_mt_B[i], p), r, p));
                ff_tc.push_back(mul_mod(mul_mod(plain_modulus % p, c, p), r, p));
            }
            for (int k = 0; k < Q; k++) {
                const u64 q = primes[k], r = R(q);
                for (int i = 0; i < bsk - 1; i++) ff_qm.push_back(mul_mod(m_B_q[(size_t) k * (bsk - 1) + i], r, q));
                ff_pB.push_back(mul_mod(prod_B_q[k], r, q));
                ff_npB.push_back(mul_mod(q - prod_B_q[k], r, q));
            }
            for (int i = 0; i < bsk - 1; i++) ff_mskm.push_back(mul_mod(m_msk[i], R(msk), msk));
            // Contract of the lazy sums (rns.hip dot128 + acc128 -> redc128): every operand below 2^61 (prime widths are
            // checked at creation, the internal primes are 61-bit) and the worst-case sum of a row below 2^128.  With the
            // moduli at hand that is exact arithmetic, not an estimate: a base of 64 primes just below 2^61 plus the
            // trailing term does NOT fit (65 * 2^122 > 2^128), shorter or narrower bases do.
            auto fits = [](const vec& in_mods, size_t cnt, u64 out_mod, u64 extra_a, u64 extra_b) {
                unsigned __int128 s = (unsigned __int128) extra_a * extra_b;
                for (size_t j = 0; j < cnt; j++) {
                    const unsigned __int128 t = (unsigned __int128) (in_mods[j] - 1) * (out_mod - 1);
                    if (s + t < s) return false;
                    s += t;
                }
                return true;
            };
            bool ok = true;
            vec qv(primes.begin(), primes.begin() + Q);
            for (int i = 0; i < bsk; i++) ok = ok && fits(qv, Q, B[i], B[i], B[i] - 1);     // fc rows, first ff rows
            ok = ok && fits(B, bsk - 1, msk, 0, 0);                                          // the m_sk row
            for (int k = 0; k < Q; k++) ok = ok && fits(B, bsk - 1, primes[k], msk, primes[k] - 1); // second ff rows
            if (!ok)
                throw std::invalid_argument("BFV base too long for its prime widths: a lazy 128-bit row sum of the BEHZ base "
                                            "conversions would overflow (use fewer or narrower primes)");
            host["behz_fc_matrix"] = fc_m;
            host["behz_fc_c1"] = fc_c1;
            host["behz_ff_matrix"] = ff_m;
            host["behz_ff_tc"] = ff_tc;
            host["behz_ff_q_matrix"] = ff_qm;
            host["behz_ff_msk_matrix"] = ff_mskm;
            host["behz_ff_prod_B"] = ff_pB;
            host["behz_ff_neg_prod_B"] = ff_npB;
        }

        // merged base [q | Bsk] with its NTT tables (bfv/context.cu:1210-1241)
        vec mm(primes.begin(), primes.begin() + Q), mpsi(psi.begin(), psi.begin() + Q), mfwd, minv, mninv;
        append(mm, B);
        append(mpsi, Bpsi);
        for (size_t i = 0; i < mm.size(); i++) {
            append(mfwd, host::power_table_bitrev(mpsi[i], mm[i], n_power));
            append(minv, host::power_table_bitrev(inv_mod_prime(mpsi[i], mm[i]), mm[i], n_power));
            mninv.push_back(inv_mod_prime(n % mm[i], mm[i]));
        }
        host["q_Bsk_merge_modulus"] = mm;
        host["q_Bsk_merge_ntt_tables"] = mfwd;
        host["q_Bsk_merge_intt_tables"] = minv;
        host["q_Bsk_n_inverse"] = mninv;

        // ---- batching (bfv/context.cu:489-499, bfv/encoder.cu:21-46): NTT tables of the plain
        // modulus and the slot -> coefficient-position map; only when t is a prime with 2N | t-1
        if (host::is_prime(t) && (t - 1) % (2 * n) == 0) {
            const u64 ppsi = host::minimal_primitive_root(2 * n, t);
            host["plain_modulus2"] = vec{t};
            host["plain_psi"] = vec{ppsi};
            host["plain_ntt_tables"] = host::power_table_bitrev(ppsi, t, n_power);
            host["plain_intt_tables"] = host::power_table_bitrev(inv_mod_prime(ppsi, t), t, n_power);
            host["n_plain_inverse"] = vec{inv_mod_prime(n % t, t)};
            vec loc;
            const u64 m = 2 * n;
            u64 pos = 1;
            auto brev = [&](u64 v) {
                u64 r = 0;
                for (int b = 0; b < n_power; b++) r |= ((v >> b) & 1) << (n_power - 1 - b);
                return r;
            };
            for (u64 i = 0; i < n / 2; i++) {
                loc.push_back(brev((pos - 1) >> 1));
                pos = (pos * 3) & (m - 1);
            }
            for (u64 i = n / 2; i < n; i++) {
                loc.push_back(brev((m - pos - 1) >> 1));
                pos = (pos * 3) & (m - 1);
            }
            host["encoding_location"] = loc;
        }

        // ---- encryption / decryption constants (bfv/context.cu:501-516, 605-620, 939-983, 1239-1343)
        {
            u64 Q_mod_t = 1;
            for (int i = 0; i < Q; i++) Q_mod_t = mul_mod(Q_mod_t, primes[i] % t, t);
            host["Q_mod_t"] = vec{Q_mod_t};
            host["upper_threshold"] = vec{(t + 1) >> 1};
            vec inc; // plain_upper_half_increment (bfv/context.cu:503-508)
            for (int i = 0; i < Q; i++) inc.push_back(primes[i] - t);
            host["upper_halfincrement"] = inc;
            // floor(prod(q) / t) mod q_i with a little-endian multi-word integer
            std::vector<u64> big{1};
            for (int i = 0; i < Q; i++) {
                unsigned __int128 carry = 0;
                for (u64& w : big) {
                    unsigned __int128 v = (unsigned __int128) w * primes[i] + carry;
                    w = (u64) v;
                    carry = v >> 64;
                }
                if (carry) big.push_back((u64) carry);
            }
            {
                unsigned __int128 rem = 0; // divide by t, most significant word first
                for (size_t k = big.size(); k-- > 0;) {
                    unsigned __int128 cur = (rem << 64) | big[k];
                    big[k] = (u64) (cur / t);
                    rem = cur % t;
                }
            }
            vec cdiv;
            for (int i = 0; i < Q; i++) {
                unsigned __int128 rem = 0;
                for (size_t k = big.size(); k-- > 0;) rem = ((rem << 64) | big[k]) % primes[i];
                cdiv.push_back((u64) rem);
            }
            host["coeff_div_plain_modulus"] = cdiv;
            vec Qi_t, Qi_gamma, Qi_inverse;
            for (int i = 0; i < Q; i++) {
                u64 a = 1, b = 1, c = 1;
                for (int j = 0; j < Q; j++)
                    if (j != i) {
                        a = mul_mod(a, primes[j] % t, t);
                        b = mul_mod(b, primes[j] % gamma, gamma);
                        c = mul_mod(c, inv_mod_prime(primes[j] % primes[i], primes[i]), primes[i]);
                    }
                Qi_t.push_back(a);
                Qi_gamma.push_back(b);
                Qi_inverse.push_back(c);
            }
            host["Qi_t"] = Qi_t;
            host["Qi_gamma"] = Qi_gamma;
            host["Qi_inverse"] = Qi_inverse;
            u64 mt_ = 1, mg_ = 1;
            for (int i = 0; i < Q; i++) {
                mt_ = mul_mod(mt_, inv_mod_prime(primes[i] % t, t), t);
                mg_ = mul_mod(mg_, inv_mod_prime(primes[i] % gamma, gamma), gamma);
            }
            host["mulq_inv_t"] = vec{t - mt_};
            host["mulq_inv_gamma"] = vec{gamma - mg_};
            host["inv_gamma"] = vec{inv_mod_prime(gamma % t, t)};
        }
    }
}

// ------------------------------------------------------------------ device
template <typename T>
static hipError_t to_device(const std::vector<T>& h, T** d)
{
    *d = nullptr;
    if (h.empty()) return hipSuccess;
    hipError_t e = hipMalloc((void**) d, h.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

static hipError_t build_plan(NttPlan& p, const vec& mods, const vec& fwd, const vec& inv, const vec& ninv,
                             int n_power, bool fp_on)
{
    const u64 n = ((u64) 1) << n_power;
    const int cnt = (int) mods.size();
    std::vector<Mod> hm(cnt);
    std::vector<ulonglong2> htw((size_t) cnt * n), hitw((size_t) cnt * n), hn(cnt), hw(cnt);
    const u64 rows = n / 256, perB = rows * 15 * 16; // re-laid entries per modulus
    std::vector<ulonglong2> htwB((size_t) cnt * perB), hitwB((size_t) cnt * perB);
    std::vector<double> htwB8((size_t) cnt * perB, 0.0);
    // fp_on == false (option fp_ntt = 0) keeps every modulus on the integer butterflies
    for (int k = 0; k < cnt; k++) {
        const u64 q = mods[k];
        hm[k] = make_mod(q);
        // moduli below 2^50: forward transform in FP64 (ntt.hip), the forward
        // tables hold (double(w), RN(w/q)) instead of (w, Shoup companion)
        // (THE place that admits a modulus to the FP64 path: every exactness bound of fpmod.cuh / ntt.hip is stated for
        // q < 2^50 -- 8 q <= 2^53 -- and machine-checked for it: tests/fp_model.py, tests/test_gpu_fp_audit.py)
        static_assert(FP_MAX_MODULUS_BITS == 50, "the bounds in fpmod.cuh, ntt.hip and tests/fp_model.py are derived for q < 2^50");
        hm[k].fp = (fp_on && hm[k].bit <= FP_MAX_MODULUS_BITS && (q >> FP_MAX_MODULUS_BITS) == 0) ? 1 : 0;
        for (u64 j = 0; j < n; j++) {
            const u64 w = fwd[k * n + j], iw = inv[k * n + j];
            if (hm[k].fp) {
                const double wd = (double) w, wi = wd / (double) q;
                u64 a, b;
                memcpy(&a, &wd, 8);
                memcpy(&b, &wi, 8);
                htw[k * n + j] = make_ulonglong2(a, b);
            } else {
                htw[k * n + j] = make_ulonglong2(w, shoup_companion(w, q));
            }
            if (hm[k].fp) { // the inverse transform of an FP64 modulus reads (double(w), RN(w/q)) pairs as well
                const double wd = (double) iw, wi = wd / (double) q;
                u64 a, b;
                memcpy(&a, &wd, 8);
                memcpy(&b, &wi, 8);
                hitw[k * n + j] = make_ulonglong2(a, b);
            } else {
                hitw[k * n + j] = make_ulonglong2(iw, shoup_companion(iw, q));
            }
        }
        for (u64 c = 0; c < rows; c++)
            for (int st = 0; st < 4; st++)
                for (int b = 0; b < (1 << st); b++)
                    for (int j = 0; j < 16; j++) {
                        const u64 src = ((((rows + c) * 16 + j) << st) + b);
                        const u64 dst = k * perB + (c * 15 + ((1 << st) - 1 + b)) * 16 + j;
                        htwB[dst] = htw[k * n + src];
                        hitwB[dst] = hitw[k * n + src];
                        if (hm[k].fp) htwB8[dst] = (double) fwd[k * n + src];
                    }
        const u64 w1n = host::mul_mod(inv[k * n + 1], ninv[k], q);
        if (hm[k].fp) {
            auto pair = [&](u64 v) {
                const double vd = (double) v, vi = vd / (double) q;
                u64 a, b;
                memcpy(&a, &vd, 8);
                memcpy(&b, &vi, 8);
                return make_ulonglong2(a, b);
            };
            hn[k] = pair(ninv[k]);
            hw[k] = pair(w1n);
        } else {
            hn[k] = make_ulonglong2(ninv[k], shoup_companion(ninv[k], q));
            hw[k] = make_ulonglong2(w1n, shoup_companion(w1n, q));
        }
    }
    p.count = cnt;
    p.has_fp = p.has_int = 0;
    for (int k = 0; k < cnt; k++) (hm[k].fp ? p.has_fp : p.has_int) = 1;
    p.fp.assign(cnt, 0);
    for (int k = 0; k < cnt; k++) p.fp[k] = hm[k].fp ? 1 : 0;
    // Correction-free forward butterflies add at most 4q to the bound of a value per stage (ntt.hip: ct_bfly), so a
    // modulus may skip every conditional subtraction when in + 4 * log2(N) * q stays below 2^64; `in` is at most twice
    // the largest modulus of the plan (canonical inputs, or digits that are residues of another prime of the plan).
    u64 qmax = 0;
    for (int k = 0; k < cnt; k++) qmax = mods[k] > qmax ? mods[k] : qmax;
    p.lazy_q_max = (~0ull - 2 * qmax) / (4ull * (u64) n_power);
    hipError_t e;
    if ((e = to_device(hm, &p.mods)) != hipSuccess) return e;
    if ((e = to_device(htw, &p.tw)) != hipSuccess) return e;
    if ((e = to_device(hitw, &p.itw)) != hipSuccess) return e;
    if ((e = to_device(htwB, &p.twB)) != hipSuccess) return e;
    if ((e = to_device(hitwB, &p.itwB)) != hipSuccess) return e;
    if ((e = to_device(htwB8, &p.twB8)) != hipSuccess) return e;
    if ((e = to_device(hn, &p.ninv)) != hipSuccess) return e;
    return to_device(hw, &p.w1ninv);
}

static void free_plan(NttPlan& p)
{
    if (p.mods) (void) hipFree(p.mods);
    if (p.tw) (void) hipFree(p.tw);
    if (p.itw) (void) hipFree(p.itw);
    if (p.twB) (void) hipFree(p.twB);
    if (p.itwB) (void) hipFree(p.itwB);
    if (p.twB8) (void) hipFree(p.twB8);
    if (p.ninv) (void) hipFree(p.ninv);
    if (p.w1ninv) (void) hipFree(p.w1ninv);
    p = NttPlan();
}

hipError_t Context::upload()
{
    if (uploaded) return hipSuccess;
    // cdt[k] = floor(2^63 * P(|round(N(0, 3.2^2))| <= k))  (secstdparams.h:22: error_std_dev = 3.2)
    for (int k = 0; k < DRBG_GAUSS_MAX; k++)
        gauss_cdt.t[k] = (u64) (erf(((double) k + 0.5) / (3.2 * 1.4142135623730951)) * 9223372036854775808.0);
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if ((e = build_plan(plan_qp, host["modulus"], host["ntt_table"], host["intt_table"], host["n_inverse"],
                        n_power, fp_ntt)) != hipSuccess)
        return e;
    static const char* u64_tables[] = {"psi_half",
                                       "last_q_modinv",
                                       "half",
                                       "half_mod",
                                       "factor",
                                       "m2_md_W0",
                                       "m2_md_G",
                                       "m2_md_C",
                                       "rescaled_last_q_modinv",
                                       "rescaled_half_mod",
                                       "rescaled_half",
                                       "base_change_matrix_Bsk",
                                       "inv_punctured_prod_mod_base_array",
                                       "base_change_matrix_m_tilde",
                                       "inv_m_tilde_mod_Bsk",
                                       "prod_q_mod_Bsk",
                                       "inv_prod_q_mod_Bsk",
                                       "base_change_matrix_q",
                                       "base_change_matrix_msk",
                                       "inv_punctured_prod_mod_B_array",
                                       "behz_mtilde_inv_punct",
                                       "behz_t_inv_punct",
                                       "behz_invq_inv_punct_B",
                                       "behz_msk_mod_q",
                                       "behz_fc_matrix",
                                       "behz_fc_c1",
                                       "behz_ff_matrix",
                                       "behz_ff_tc",
                                       "behz_ff_q_matrix",
                                       "behz_ff_msk_matrix",
                                       "behz_ff_prod_B",
                                       "behz_ff_neg_prod_B",
                                       "prod_B_mod_q",
                                       "Mi",
                                       "Mi_inv",
                                       "upper_half_threshold",
                                       "decryption_modulus",
                                       "special_fft_roots_table",
                                       "special_ifft_roots_table",
                                       "coeff_div_plain_modulus",
                                       "upper_halfincrement",
                                       "Qi_t",
                                       "Qi_gamma",
                                       "Qi_inverse",
                                       "m2_Mi_inv",
                                       "m2_matrix",
                                       "m2_prod",
                                       "m2_matrix_mg",
                                       "m2_negprod_mg"};
    for (const char* nm : u64_tables) {
        auto it = host.find(nm);
        if (it == host.end()) continue;
        u64* d = nullptr;
        if ((e = to_device(it->second, &d)) != hipSuccess) return e;
        dev[nm] = d;
    }
    for (const char* nm : {"new_prime_locations", "new_input_locations", "m2_I_j", "m2_I_location", "encoding_location",
                           "reverse_order"}) {
        auto it = host.find(nm);
        if (it == host.end()) continue;
        std::vector<int> v(it->second.begin(), it->second.end());
        int* d = nullptr;
        if ((e = to_device(v, &d)) != hipSuccess) return e;
        dev[nm] = d;
    }
    if (scheme == SCHEME_BFV) {
        if ((e = build_plan(plan_merge, host["q_Bsk_merge_modulus"], host["q_Bsk_merge_ntt_tables"],
                            host["q_Bsk_merge_intt_tables"], host["q_Bsk_n_inverse"], n_power, fp_ntt)) != hipSuccess)
            return e;
        if (host.count("plain_modulus2") &&
            (e = build_plan(plan_plain, host["plain_modulus2"], host["plain_ntt_tables"], host["plain_intt_tables"],
                            host["n_plain_inverse"], n_power, fp_ntt)) != hipSuccess)
            return e;
        behz.ibase = plan_merge.mods;
        behz.obase = plan_merge.mods + Q_size;
        behz.m_tilde = make_mod(((u64) 1) << 32);
        behz.plain = make_mod(plain_modulus);
        behz.inv_prod_q_mod_m_tilde = host["inv_prod_q_mod_m_tilde"][0];
        behz.inv_prod_B_mod_m_sk = host["inv_prod_B_mod_m_sk"][0];
        behz.inv_m_tilde_mod_Bsk = d64("inv_m_tilde_mod_Bsk");
        behz.prod_q_mod_Bsk = d64("prod_q_mod_Bsk");
        behz.base_change_matrix_Bsk = d64("base_change_matrix_Bsk");
        behz.base_change_matrix_m_tilde = d64("base_change_matrix_m_tilde");
        behz.inv_punctured_prod_mod_base_array = d64("inv_punctured_prod_mod_base_array");
        behz.inv_prod_q_mod_Bsk = d64("inv_prod_q_mod_Bsk");
        behz.inv_punctured_prod_mod_B_array = d64("inv_punctured_prod_mod_B_array");
        behz.base_change_matrix_q = d64("base_change_matrix_q");
        behz.base_change_matrix_msk = d64("base_change_matrix_msk");
        behz.prod_B_mod_q = d64("prod_B_mod_q");
        behz.mtilde_inv_punct = d64("behz_mtilde_inv_punct");
        behz.t_inv_punct = d64("behz_t_inv_punct");
        behz.invq_inv_punct_B = d64("behz_invq_inv_punct_B");
        behz.msk_mod_q = d64("behz_msk_mod_q");
        behz.fc_matrix = d64("behz_fc_matrix");
        behz.fc_c1 = d64("behz_fc_c1");
        behz.ff_matrix = d64("behz_ff_matrix");
        behz.ff_tc = d64("behz_ff_tc");
        behz.ff_q_matrix = d64("behz_ff_q_matrix");
        behz.ff_msk_matrix = d64("behz_ff_msk_matrix");
        behz.ff_prod_B = d64("behz_ff_prod_B");
        behz.ff_neg_prod_B = d64("behz_ff_neg_prod_B");
        behz.ibase_size = Q_size;
        behz.obase_size = bsk_size;
        behz.split = behz_split;
    }
    uploaded = true;
    return hipDeviceSynchronize();
}

// ---- options (hegpu_context_set_option).  Environment variables HEGPU_<NAME> only seed the defaults, once, when the
// context is created; nothing on a call path reads the environment.
namespace {
struct OptDesc { const char* name; const char* env; int lo, hi; };
const OptDesc kOptions[] = {
    {"fused_row_mac", "HEGPU_FUSED_ROW_MAC", -1, 1}, {"fused_moddown", "HEGPU_FUSED_MODDOWN", 0, 1},
    {"col_multi", "HEGPU_COL_MULTI", -1, 1},         {"single_pass", "HEGPU_SINGLE_PASS", -1, 1},
    {"ntt_galois", "HEGPU_NTT_GALOIS", 0, 1},        {"galois_scatter", "HEGPU_GALOIS_SCATTER", 0, 1},
    {"fuse_inverse", "HEGPU_FUSE_INVERSE", 0, 1},    {"copy_along", "HEGPU_COPY_ALONG", 0, 1},
    {"digit_split", "HEGPU_DIGIT_SPLIT", -1, 4},     {"fp_ntt", "HEGPU_FP_NTT", 0, 1},
    {"behz_split", "HEGPU_BEHZ_SPLIT", -1, 1},       {"fused_tensor", "HEGPU_FUSED_TENSOR", 0, 1},
};
} // namespace

int Context::set_option(const char* name, int value)
{
    if (!name) return 1;
    const OptDesc* d = nullptr;
    for (const OptDesc& o : kOptions)
        if (!strcmp(o.name, name)) d = &o;
    if (!d) return 1;
    if (value < d->lo || value > d->hi) return 2;
    const std::string nm(name);
    if (nm == "fp_ntt") {
        if (uploaded) return 3; // decides the layout of the twiddle tables
        fp_ntt = value != 0;
    } else if (nm == "fused_row_mac") fused_row_mac = value;
    else if (nm == "fused_moddown") fused_moddown = value != 0;
    else if (nm == "col_multi") col_multi = value;
    else if (nm == "single_pass") single_pass = value;
    else if (nm == "ntt_galois") ntt_galois = value != 0;
    else if (nm == "galois_scatter") galois_scatter = value != 0;
    else if (nm == "fuse_inverse") fuse_inverse = value != 0;
    else if (nm == "copy_along") copy_along = value != 0;
    else if (nm == "fused_tensor") fused_tensor = value != 0;
    else if (nm == "digit_split") {
        if (value == 1 || value == 3) return 2;
        digit_split = value;
    } else if (nm == "behz_split") {
        behz_split = value;
        behz.split = value;
    }
    return 0;
}

int Context::get_option(const char* name, int* value) const
{
    if (!name || !value) return 1;
    const std::string nm(name);
    if (nm == "fp_ntt") *value = fp_ntt;
    else if (nm == "fused_row_mac") *value = fused_row_mac;
    else if (nm == "fused_moddown") *value = fused_moddown;
    else if (nm == "col_multi") *value = col_multi;
    else if (nm == "single_pass") *value = single_pass;
    else if (nm == "ntt_galois") *value = ntt_galois;
    else if (nm == "galois_scatter") *value = galois_scatter;
    else if (nm == "fuse_inverse") *value = fuse_inverse;
    else if (nm == "copy_along") *value = copy_along;
    else if (nm == "fused_tensor") *value = fused_tensor;
    else if (nm == "digit_split") *value = digit_split;
    else if (nm == "behz_split") *value = behz_split;
    else return 1;
    return 0;
}

// The one place the library reads the environment: a variable that is set to a whole decimal integer (optional sign,
// nothing else -- "yes", "", "1x" are not numbers and are ignored, they do not mean 0).
bool env_long(const char* name, long* out)
{
    const char* e = getenv(name);
    if (!e || !*e) return false;
    char* end = nullptr;
    errno = 0;
    const long v = strtol(e, &end, 10);
    if (errno || end == e || *end) return false;
    *out = v;
    return true;
}

void Context::seed_options_from_env()
{
    for (const OptDesc& o : kOptions) {
        long v;
        if (env_long(o.env, &v) && v >= INT_MIN && v <= INT_MAX)
            (void) set_option(o.name, (int) v); // a value the setter refuses (out of range) is ignored
    }
}

void Context::release_device()
{
    free_plan(plan_qp);
    free_plan(plan_merge);
    free_plan(plan_plain);
    for (auto& kv : dev)
        if (kv.second) (void) hipFree(kv.second);
    dev.clear();
    uploaded = false;
}

Context::~Context()
{
    if (uploaded) release_device();
}

const u64* Context::d64(const char* name) const
{
    auto it = dev.find(name);
    return it == dev.end() ? nullptr : (const u64*) it->second;
}
const int* Context::d32(const char* name) const
{
    auto it = dev.find(name);
    return it == dev.end() ? nullptr : (const int*) it->second;
}

NttArgs Context::ntt_args(int table_set) const
{
    const NttPlan& p = table_set == 2 ? plan_plain : (table_set ? plan_merge : plan_qp);
    NttArgs a{};
    a.mods = p.mods;
    a.tw = p.tw;
    a.itw = p.itw;
    a.twB = p.twB;
    a.itwB = p.itwB;
    a.twB8 = p.twB8;
    a.ninv = p.ninv;
    a.w1ninv = p.w1ninv;
    a.n_power = n_power;
    a.mod_count = p.count;
    a.col_multi = col_multi;
    a.single_pass = single_pass;
    a.plan_has_fp = p.has_fp;
    a.plan_has_int = p.has_int;
    a.lazy_q_max = p.lazy_q_max;
    return a;
}

} // namespace hegpu

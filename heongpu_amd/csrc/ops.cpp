// ops.cpp -- operator sequences (the reference's host orchestration), batched
// over independent ciphertexts and free of allocation: all temporaries live in
// a caller-provided workspace so the sequences can be captured in a hipGraph.
#include "ops.hpp"
#include <vector>
#include <utility>

namespace hegpu {

#define TRY(x)                           \
    do {                                 \
        hipError_t e__ = (x);            \
        if (e__ != hipSuccess) return e__; \
    } while (0)

static int prime_loc_offset(const Context& c, int depth)
{
    int counter = c.Qp_size, location = 0; // reference ckks/operator.cu:949-955
    for (int i = 0; i < depth; i++) { location += counter; counter--; }
    return location;
}

// The fused row pass + inner product runs one workgroup per (ciphertext, target modulus, 16-row tile) that walks
// all digits; a launch too small to fill the chip finishes sooner as the reference's sequence -- the transform of
// all digits x moduli as independent limbs, then the element-wise inner product.  512 workgroups of this kernel are
// resident at a time (two per CU); measured crossover (tools/relin_sweep.py, relinearize of B ciphertexts, fused
// against unfused): N = 2^14, Q = 8: beyond B = 16 (36 B workgroups); N = 2^15, Q = 15: B = 4 388 / 360 us, B = 8
// 609 / 685 (128 B); N = 2^16, Q = 16: B = 1 401 / 288, B = 2 476 / 452, B = 4 688 / 788 (272 B); Q = 30: B = 1
// 804 / 758, B = 2 1098 / 1194 (496 B) -- one and a half rounds of resident workgroups.  Chains of integer-butterfly
// moduli (the 58/59-bit default chains) cross earlier, their transforms being the larger part of either path:
// N = 2^15, Q = 14: B = 2 242 / 220, B = 4 361 / 377 (120 B); N = 2^16, Q = 14: B = 1 287 / 274, B = 2 388 / 426
// (240 B); Q = 29: B = 1 756 / 866 (480 B).
static long fused_row_mac_need(const Context& c)
{
    size_t fp = 0;
    for (unsigned char f : c.plan_qp.fp) fp += f ? 1 : 0;
    return (2 * fp > c.plan_qp.fp.size()) ? 768 : 400;
}
// Between the two: the fused kernel with 2 (or, forced, 4) workgroups per (ciphertext, limb slot, tile), each over a
// part of the digits, and a pass that adds the partial sums (KsMacArgs::splits).  Measured (tools/relin_sweep.py,
// us per launch, unfused / fused / two pieces / four pieces): N = 2^16, Q = 16: B = 1 289 / 368 / 307 / 289, B = 2
// 456 / 457 / 410 / 410; Q = 30: B = 1 766 / 810 / 718 / 687, B = 2 1217 / 1123 / 1044 / 1027; N = 2^15, Q = 15: B = 1
// 154 / 251 / 195 / 171, B = 2 219 / 307 / 255 / 234; chains of integer-butterfly moduli: never ahead, and the order
// of fused and unfused changes from box to box there.  Round 3, with the integer and the FP64 moduli of a split
// launch in one grid (ks_row_mac_split; before, two half-empty grids one after the other): unfused / fused / four
// pieces, N = 2^16, Q = 16: B = 1 294 / 409 / 285, B = 2 467 / 476 / 409, B = 4 813 / 702 / 701; Q = 30: B = 1
// 806 / 832 / 690, B = 2 1277 / 1138 / 1029, B = 4 2283 / 1864 / 1829; N = 2^15, Q = 15: B = 1 155 / 252 / 154, B = 2
// 221 / 302 / 232, B = 4 370 / 394 / 347 (tools/relin_small.py).  So: four pieces, on FP64-path chains at N = 2^16,
// for launches below two rounds of the fused path's size that reach one round that way.  Returns 0 (no split) or
// the number of pieces.
static int fused_digit_splits(const Context& c, int rc, int digits, int batch)
{
    if (c.digit_split == 0) return 0;
    if (c.digit_split > 0) return digits >= 2 * c.digit_split ? c.digit_split : 0;
    if (c.fused_row_mac >= 0 || c.n_power < 16 || digits < 8) return 0;
    const long wgs = (long) batch * rc * (long) (c.n >> 12), need = fused_row_mac_need(c);
    return (need == 768 && wgs < 2 * need && 4 * wgs >= need) ? 4 : 0;
}
static bool use_fused_row_mac(const Context& c, int rc, int batch)
{
    if (c.fused_row_mac >= 0) return c.fused_row_mac != 0;
    return (long) batch * rc * (long) (c.n >> 12) >= fused_row_mac_need(c);
}

// The target slots of a decomposing launch over the Q' chain whose moduli run on the integer butterflies
// (NttArgs::int_slots).  `order`: host copy of the launch's mod_order (nullptr: slot k is modulus k).
static void fill_int_slots(const Context& c, NttArgs& a, const u64* order)
{
    int cnt = 0;
    for (int k = 0; k < a.decomp_mods; k++) {
        const int m = a.mod_offset + (order ? (int) order[k] : k);
        if (m < 0 || m >= (int) c.plan_qp.fp.size()) { a.int_slot_count = 0; return; }
        if (c.plan_qp.fp[m]) continue;
        if (cnt == 8) { a.int_slot_count = 0; return; } // more than the list holds: full grid
        a.int_slots[cnt++] = k;
    }
    a.int_slot_count = cnt ? cnt : -1;
}

// Forward NTT of the key-switch digits followed by the inner product with the
// key.  `a` is the fully configured forward transform (plain or decomposing)
// whose output is [digits][rc][N] per ciphertext at a.out; the result
// [2][rc][N] goes to `acc`.  Fused path: column pass + ks_row_mac; otherwise
// two-pass NTT + rns_keyswitch_mac.
// `ident` (with a.skip_identity): the NTT-domain limbs [digits][N] of the
// polynomial being decomposed, `ident_stride` apart; digit d at modulus d is
// taken from there instead of being transformed.
// `which`: 1 = column pass only, 2 = row pass + inner product only, 3 = both (the measurement seam of
// hegpu_probe_ckks_relinearize; the unfused path ignores it)
static hipError_t keyswitch_ntt_mac(const Context& c, NttArgs a, const u64* key, u64* acc, u64 acc_stride,
                                    int digits, int rc, int split, int level, const u64* ident, u64 ident_stride,
                                    int batch, hipStream_t st, int which = 3)
{
    const int ppi = digits * rc;
    const int skip_identity = ident ? 1 : 0;
    a.skip_identity = skip_identity;
    const int splits = fused_digit_splits(c, rc, digits, batch);
    if (!splits && !use_fused_row_mac(c, rc, batch)) {
        // identity digits (digit d at modulus d: the transform gives back the NTT-domain limb): copied instead of
        // transformed -- unless this path was chosen for a small launch, where one kernel less is worth more than
        // one transform in ten (same residues either way)
        if (ident && c.fused_row_mac == 0)
            TRY(rns_copy_diag(ident, ident_stride, a.out, a.out_item_stride, c.n_power, digits, rc, batch, st));
        else a.skip_identity = 0;
        TRY(ntt_launch(a, ppi * batch, false, st));
        return rns_keyswitch_mac(a.out, a.out_item_stride, key, acc, acc_stride, c.plan_qp.mods, c.n_power, digits, rc,
                                 c.Qp_size, split, level, batch, st);
    }
    const int chunk = 65535 / ppi; // gridDim.y limit of the column pass
    if (chunk < 1) return hipErrorInvalidValue;
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        NttArgs ca = a;
        ca.in = a.in + (u64) b0 * a.in_item_stride;
        ca.out = a.out + (u64) b0 * a.out_item_stride;
        if (which & 1) TRY(ntt_launch_fwd_col(ca, ppi * nb, st));
        if (!(which & 2)) continue;
        KsMacArgs k{};
        k.in = ca.out; k.in_item_stride = a.out_item_stride; k.key = key;
        k.out = acc + (u64) b0 * acc_stride; k.out_item_stride = acc_stride;
        k.mods = c.plan_qp.mods; k.tw = c.plan_qp.tw; k.twB = c.plan_qp.twB; k.twB8 = c.plan_qp.twB8; k.mod_order = a.mod_order;
        k.lazy_q_max = a.lazy_q_max; k.n_power = c.n_power; k.digits = digits; k.rc = rc; k.key_limbs = c.Qp_size; k.skip_identity = skip_identity;
        k.ident = ident ? ident + (u64) b0 * ident_stride : nullptr; k.ident_item_stride = ident_stride;
        k.splits = splits;
        k.no_fp = !a.plan_has_fp;
        k.no_int = !a.plan_has_int || a.int_slot_count < 0;
        k.int_slot_count = a.int_slot_count > 0 ? a.int_slot_count : 0;
        for (int q = 0; q < 8; q++) k.int_slots[q] = a.int_slots[q];
        TRY(ks_row_mac_launch(k, nb, st));
        if (splits > 1)
            TRY(rns_sum_partials(ca.out, a.out_item_stride, k.out, acc_stride, c.plan_qp.mods, a.mod_order, c.n_power,
                                 digits, rc, splits, nb, st));
    }
    return hipSuccess;
}

size_t ops_workspace_elems(const Context& c, int op, int depth, int batch)
{
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const int L = Q + c.bsk_size;
    u64 per = 0;
    switch (op) {
        case OP_CKKS_RELIN: per = ((u64) l * rc + 2 * rc) * n; break;
        case OP_CKKS_RESCALE: per = ((u64) 2 * (l - 1) + 2 * l) * n; break;
        case OP_CKKS_GALOIS: per = ((u64) 2 * l + (u64) l * rc + 2 * rc) * n; break;
        case OP_CKKS_ROTATE_HOISTED: per = ((u64) 2 * l + (u64) l * rc + 4 * 2 * rc) * n; break; // four accumulators
        case OP_BFV_MULTIPLY: per = (u64) 7 * L * n; break;
        case OP_BFV_RELIN: per = ((u64) Q * Qp + 2 * Qp) * n; break;
        case OP_BFV_GALOIS: per = ((u64) Q * Qp + 2 * Qp) * n; break;
        case OP_KEYGEN_SECRET: per = n; break;                      // 2 x hamming weight ints
        case OP_KEYGEN_PUBLIC: per = (u64) 2 * Qp * n; break;        // e, a
        case OP_KEYGEN_SWITCH: per = (u64) 2 * Q * Qp * n; break;    // e, a per digit
        case OP_CKKS_ENCRYPT:
        case OP_BFV_ENCRYPT: per = (u64) 5 * Qp * n; break;          // u, e[2], pk*u[2]
        case OP_BFV_DECRYPT: per = (u64) Q * n; break;               // c1*s
        case OP_BFV_DECODE: per = n; break;
        case OP_BFV_MULTIPLY_PLAIN: per = (u64) Q * n; break;        // lifted + transformed plaintext
        case OP_CKKS_ENCODE: per = n; break;                         // N/2 complex doubles
        case OP_CKKS_DECODE: per = (u64) (l + 1) * n; break;         // coefficient-domain copy + complex
        default: return 0;
    }
    return per * (u64) batch;
}

hipError_t op_ckks_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                            int depth, int batch, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    return rns_cross_multiplication(ct1, s1, ct2, s2, out, so, c.plan_qp.mods, c.n_power, c.Q_size - depth, batch,
                                    st);
}

// Key switch (method I) of one NTT-domain polynomial per item, the shared core of relinearize and rotate
// (reference ckks/operator.cu:919-1020 and :1461-1545):
//   out[part p] = moddown(sum_i digit_i(src) * key[i][p]) + (p < add_parts ? add[part p] : 0),   p = 0, 1
// src: [l][N] per item, src_stride apart; add / out: [2][l][N] per item.  temp1: [l][rc][N], temp2: [2][rc][N]
// per item, both `per` apart.  out may alias add.
static u64 inv_mod_2n(u64 g, u64 two_n);
static hipError_t ckks_keyswitch_core(const Context& c, const u64* src, u64 src_stride, const u64* add, u64 add_stride,
                                      int add_parts, u64* outp, u64 out_stride, const u64* key, int depth, int batch,
                                      u64* temp1, u64* temp2, u64 per, hipStream_t st, unsigned phases,
                                      int galois_elt = 0)
{
    const int np = c.n_power;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const Mod* mods = c.plan_qp.mods;

    NttArgs a = c.ntt_args(0);
    // INTT(src), batch l per item (:919) -- out of place into the (not yet used)
    // accumulator region: src itself stays in the NTT domain because digit d
    // re-reduced into its own modulus d and transformed back is that limb
    a.in = src; a.out = temp2; a.mod_count = l; a.polys_per_item = l;
    a.in_item_stride = src_stride; a.out_item_stride = per;
    // digit decomposition src -> [l][rc][N] fused into the forward NTT's
    // first load; modulus order skips dropped primes                (:932-960)
    NttArgs dgt = c.ntt_args(0);
    dgt.in = temp2; dgt.out = temp1; dgt.mod_count = rc; dgt.polys_per_item = l * rc; dgt.decomp_mods = rc;
    dgt.in_item_stride = per; dgt.out_item_stride = per;
    dgt.mod_order = c.d32("new_prime_locations") + prime_loc_offset(c, depth);
    fill_int_slots(c, dgt, c.h64("new_prime_locations").data() + prime_loc_offset(c, depth));
    // When the decomposing column pass is the multi-modulus kernel (one launch for the whole batch), it also
    // finishes the inverse transform of its source tiles: only the row stages of the INTT run on their own.
    const bool fuse_inv = use_fused_row_mac(c, rc, batch) && c.fuse_inverse && (long) l * rc * batch <= 65535 &&
                          ntt_decomp_uses_multi(dgt, l * rc * batch);
    if (phases & RELIN_PHASE_INTT_C2) {
        if (fuse_inv) TRY(ntt_launch_inv_rows(a, l * batch, st));
        else TRY(ntt_launch(a, l * batch, true, st));
    }
    dgt.src_inv = fuse_inv ? 1 : 0;
    dgt.single_decomp_ok = 1; // temp2 -> temp1, no epilogue
    a = dgt;
    // forward NTT of the digits + inner product with the key      (:956-988)
    const int which = ((phases & RELIN_PHASE_COLUMN) ? 1 : 0) | ((phases & RELIN_PHASE_ROW_MAC) ? 2 : 0);
    if (which) TRY(keyswitch_ntt_mac(c, a, key, temp2, per, l, rc, l, depth, src, src_stride, batch, st, which));
    // INTT of the two P-limb polynomials only                        (:996)
    a = c.ntt_args(0);
    a.in = temp2; a.out = temp2; a.mod_count = 1; a.mod_offset = Q; a.polys_per_item = 2;
    a.in_item_stride = a.out_item_stride = per;
    a.poly_order = c.d32("new_input_locations") + 2 * depth;
    // the same fusion for the mod-down transform, whose source is the inverse transform of the P limbs
    NttArgs md = c.ntt_args(0);
    md.in = temp2; md.out = temp1; md.mod_count = l; md.polys_per_item = 2 * l;
    md.in_item_stride = md.out_item_stride = per;
    md.decomp_mods = l; md.decomp_in_mul = l + 1; md.decomp_in_add = l;
    md.half_on = 1; md.half_src_mod = Q;
    fill_int_slots(c, md, nullptr);
    const bool fuse_inv_p = c.fused_moddown && c.fuse_inverse && (long) 2 * l * batch <= 65535 &&
                            ntt_decomp_uses_multi(md, 2 * l * batch);
    if (phases & RELIN_PHASE_INTT_P) {
        if (fuse_inv_p) TRY(ntt_launch_inv_rows(a, 2 * batch, st));
        else TRY(ntt_launch(a, 2 * batch, true, st));
    }
    if (!(phases & RELIN_PHASE_MODDOWN)) return hipSuccess;
    // stage one: P limb (+half) reduced into every q_j               (:1003)
    if (!c.fused_moddown)
        TRY(rns_moddown_stage_one(temp2, per, temp1, per, mods, c.d64("half"), c.d64("half_mod"), np, Q, l, batch,
                                  st));
    // forward NTT of that (:1011) with stage one as its load transform and stage two -- (x - last) * P^-1 + add, written to
    // out parts 0,1 (:1015) -- as the epilogue of its row pass
    a = c.ntt_args(0);
    a.in = temp1; a.out = temp1; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = a.out_item_stride = per;
    if (c.fused_moddown) {
        // input: the P limb (slot l) of each of the two parts of temp2 [2][l+1][N]
        a.in = temp2; a.decomp_mods = l; a.decomp_in_mul = l + 1; a.decomp_in_add = l;
        a.half_on = 1; a.half_src_mod = Q; a.half = c.h64("half")[0]; a.half_mod = c.d64("half_mod");
        a.epi.on = 1;
        a.epi.ks = temp2; a.epi.ks_item_stride = per; a.epi.ks_part_limbs = l + 1;
        a.epi.ct = add; a.epi.ct_item_stride = add_stride; a.epi.ct_parts = add_parts;
        a.epi.out = outp; a.epi.out_item_stride = out_stride;
        a.epi.inv = c.d64("last_q_modinv");
        a.epi.limbs = l;
        a.epi.galois_inv = galois_elt ? (unsigned) inv_mod_2n((u64) galois_elt, 2 * c.n) : 0u;
        a.src_inv = fuse_inv_p ? 1 : 0;
        a.single_decomp_ok = 1; // source: the P-limb slots of temp2; the epilogue reads its Q-limb slots and `add`, writes `outp`
        fill_int_slots(c, a, nullptr);
        return ntt_launch(a, 2 * l * batch, false, st);
    }
    if (add_parts != 2 || galois_elt) return hipErrorInvalidValue; // the stand-alone stage two adds both parts (relinearize only)
    TRY(ntt_launch(a, 2 * l * batch, false, st));
    return rns_moddown_stage_two(temp1, per, temp2, per, l + 1, add, add_stride, outp, out_stride, mods,
                                 c.d64("last_q_modinv"), np, l, 1, batch, st);
}

// reference ckks/operator.cu:899-1023
hipError_t op_ckks_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                               hipStream_t st, unsigned phases)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const u64 n = c.n;
    const int l = c.Q_size - depth, rc = c.Qp_size - depth;
    const u64 per = ((u64) l * rc + 2 * rc) * n;
    u64* temp1 = ws;                      // [l][rc][N] per item, later [2][l][N]
    u64* temp2 = ws + (u64) l * rc * n;   // [2][rc][N] per item
    return ckks_keyswitch_core(c, ct + ((u64) l << (c.n_power + 1)), cs, ct, cs, 2, ct, cs, key, depth, batch, temp1,
                               temp2, per, st, phases);
}

// reference ckks/operator.cu:1156-1244
hipError_t op_ckks_rescale(const Context& c, u64* ct, u64 cs, int depth, int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, P = c.P_size;
    const int l = Q - depth;
    int counter = Q - 1, location = 0;
    for (int i = 0; i < depth; i++) { location += counter; counter--; }
    const u64 per = ((u64) 2 * (l - 1) + 2 * l) * n;
    u64* temp1 = ws;                         // [2][l-1][N]
    u64* temp2 = ws + (u64) 2 * (l - 1) * n; // copy of ct, part stride l
    const Mod* mods = c.plan_qp.mods;

    NttArgs a = c.ntt_args(0);
    a.in = ct; a.out = ct; a.mod_count = 1; a.mod_offset = l - 1; a.polys_per_item = 2;
    a.in_item_stride = a.out_item_stride = cs;
    a.poly_order = c.d32("new_input_locations") + (depth + P) * 2;
    TRY(ntt_launch(a, 2 * batch, true, st));                                               // :1197
    if (c.fused_moddown) {
        // stage one (:1205) as the load transform and stage two (:1225) as the row-pass epilogue
        // of the forward NTT (:1214); the copy of the kept limbs (:1219) comes first because the
        // epilogue writes the compacted ciphertext over them
        a = c.ntt_args(0);
        a.in = ct; a.out = temp1; a.mod_count = l - 1; a.polys_per_item = 2 * (l - 1);
        a.in_item_stride = cs; a.out_item_stride = per;
        a.decomp_mods = l - 1; a.decomp_in_mul = l; a.decomp_in_add = l - 1;
        a.half_on = 1; a.half_src_mod = l - 1; a.half = c.h64("rescaled_half")[depth];
        a.half_mod = c.d64("rescaled_half_mod") + location;
        // The copy rides on the column pass when that is the per-polynomial kernel (one workgroup per kept limb
        // and tile: C2, 4.7 us of launch less); the multi-modulus kernel keeps its own launch for it.
        if (c.copy_along && !ntt_decomp_uses_multi(a, 2 * (l - 1) * batch)) {
            a.copy_src = ct; a.copy_src_item_stride = cs;
            a.copy_dst = temp2; a.copy_dst_item_stride = per;
            a.copy_part_limbs = l;
        } else {
            TRY(rns_copy_limbs(ct, (u64) l * n, cs, temp2, (u64) l * n, per, np, l - 1, 2, batch, st));
        }
        a.epi.on = 1;
        a.epi.ks = temp2; a.epi.ks_item_stride = per; a.epi.ks_part_limbs = l;
        a.epi.ct = nullptr; a.epi.ct_item_stride = 0;
        a.epi.out = ct; a.epi.out_item_stride = cs;
        a.epi.inv = c.d64("rescaled_last_q_modinv") + location;
        a.epi.limbs = l - 1;
        return ntt_launch(a, 2 * (l - 1) * batch, false, st);
    }
    TRY(rns_moddown_stage_one(ct, cs, temp1, per, mods, c.d64("rescaled_half") + depth,
                              c.d64("rescaled_half_mod") + location, np, l - 1, l - 1, batch, st)); // :1205
    a = c.ntt_args(0);
    a.in = temp1; a.out = temp1; a.mod_count = l - 1; a.polys_per_item = 2 * (l - 1);
    a.in_item_stride = a.out_item_stride = per;
    TRY(ntt_launch(a, 2 * (l - 1) * batch, false, st));                                    // :1214
    TRY(rns_copy_limbs(ct, (u64) l * n, cs, temp2, (u64) l * n, per, np, l - 1, 2, batch, st)); // :1219
    return rns_moddown_stage_two(temp1, per, temp2, per, l, nullptr, 0, ct, cs, mods,
                                 c.d64("rescaled_last_q_modinv") + location, np, l - 1, 0, batch, st); // :1225
}

// reference ckks/operator.cu:1422-1559
hipError_t op_ckks_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                int galois_elt, int depth, int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const u64 per = ((u64) 2 * l + (u64) l * rc + 2 * rc) * n;
    u64* temp0 = ws;                       // [2][l][N] coefficient-domain copy of ct
    u64* temp2 = temp0 + (u64) 2 * l * n;  // [l][rc][N]
    u64* temp3 = temp2 + (u64) l * rc * n; // [2][rc][N]
    const Mod* mods = c.plan_qp.mods;
    const int* order = c.d32("new_prime_locations") + prime_loc_offset(c, depth);

    if (c.fused_moddown && c.ntt_galois) {
        // The key switch of c1 exactly as relinearize does it (c0 added to part 0 by the mod-down epilogue), all
        // in the NTT domain, and the automorphism last, as a slot gather of both parts.  Against the reference's
        // order (INTT of both parts, ..., INTT of 2 rc limbs, mod-down + permute in the coefficient domain, NTT
        // of 2 l limbs) this never transforms c0, inverse-transforms 2 instead of 2 rc accumulator limbs and has
        // no scattered stores; the residues are the same (the permutation commutes with the transform, the
        // mod-down is the same exact integer function either way).
        if (c.galois_scatter)
            return ckks_keyswitch_core(c, ct + (u64) l * n, cs, ct, cs, 1, out, so, key, depth, batch, temp2, temp3, per, st,
                                       RELIN_PHASE_ALL, galois_elt);
        TRY(ckks_keyswitch_core(c, ct + (u64) l * n, cs, ct, cs, 1, temp0, per, key, depth, batch, temp2, temp3, per, st,
                                RELIN_PHASE_ALL));
        return rns_permute_ntt(temp0, per, out, so, galois_elt, np, 2 * l, batch, st);
    }
    NttArgs a = c.ntt_args(0);
    a.in = ct; a.out = temp0; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = cs; a.out_item_stride = per;
    TRY(ntt_launch(a, 2 * l * batch, true, st));                                           // :1461
    a = c.ntt_args(0); // ckks_duplicate_kernel fused into the NTT load          :1467-1494
    a.in = temp0 + (u64) l * n; a.out = temp2; a.mod_count = rc; a.polys_per_item = l * rc; a.decomp_mods = rc;
    a.in_item_stride = a.out_item_stride = per;
    a.mod_order = order;
    TRY(keyswitch_ntt_mac(c, a, key, temp3, per, l, rc, l, depth, ct + (u64) l * n, cs, batch, st)); // :1490-1520
    a.decomp_mods = 0;
    a.skip_identity = 0;
    a.in = temp3; a.out = temp3; a.polys_per_item = 2 * rc;
    TRY(ntt_launch(a, 2 * rc * batch, true, st));                                          // :1524
    TRY(rns_moddown_permute(temp3, per, temp0, per, out, so, mods, c.d64("half"), c.d64("half_mod"),
                            c.d64("last_q_modinv"), galois_elt, np, rc, l, Qp, Q, c.P_size, batch, st)); // :1530
    a = c.ntt_args(0);
    a.in = out; a.out = out; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = a.out_item_stride = so;
    return ntt_launch(a, 2 * l * batch, false, st);                                        // :1541
}

// reference bfv/operator.cu:336-430
hipError_t op_bfv_multiply(const Context& c, const u64* ct1, u64 s1, const u64* ct2, u64 s2, u64* out, u64 so,
                           int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int L = c.Q_size + c.bsk_size;
    const u64 per = (u64) 7 * L * n;
    u64* temp1 = ws;                   // [4][L][N]
    u64* temp2 = ws + (u64) 4 * L * n; // [3][L][N]
    TRY(rns_fast_convertion(ct1, s1, ct2, s2, temp1, per, c.behz, np, batch, st));         // :364
    NttArgs a = c.ntt_args(1);
    a.in = temp1; a.out = temp1; a.mod_count = L; a.polys_per_item = 4 * L;
    a.in_item_stride = a.out_item_stride = per;
    TRY(ntt_launch(a, 4 * L * batch, false, st));                                          // :393
    a.in = temp2; a.out = temp2; a.polys_per_item = 3 * L;
    if (c.fused_tensor) {
        // the tensor product (:399) as the load transform of the inverse transform (:410): [3][L][N] is never stored
        a.tensor_in = temp1; a.tensor_item_stride = per; a.tensor_limbs = L;
    } else {
        TRY(rns_cross_multiplication(temp1, per, temp1 + (u64) 2 * L * n, per, temp2, per, c.plan_merge.mods, np, L,
                                     batch, st));                                          // :399
    }
    TRY(ntt_launch(a, 3 * L * batch, true, st));                                           // :410
    return rns_fast_floor(temp2, per, out, so, c.behz, np, batch, st);                     // :416
}

// BFV key switching, tail: acc [2][Q'][N] (NTT domain, `per` apart) -> INTT -> divide by the special prime with
// rounding -> + ct (parts below add_parts) [-> Galois permutation] -> out [2][Q][N].  The reference runs the INTT
// of all 2 Q' limbs and then divide_round_lastq(_permute_bfv)_kernel (bfv/operator.cu:571-576, 846-853); here the
// two P limbs are inverse-transformed first and the division is the epilogue of the Q limbs' inverse transform
// (NttInvEpilogue): the coefficient-domain accumulator is never written or read back.
static hipError_t bfv_intt_moddown(const Context& c, u64* acc, u64 per, const u64* ct, u64 cs, int add_parts, u64* out,
                                   u64 so, int galois_elt, int batch, hipStream_t st)
{
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    for (int part = 0; part < 2; part++) {
        NttArgs a = c.ntt_args(0);
        a.in = a.out = acc + (u64) (part * Qp + Q) * n;
        a.mod_count = 1; a.mod_offset = Q; a.polys_per_item = 1;
        a.in_item_stride = a.out_item_stride = per;
        TRY(ntt_launch(a, batch, true, st));
    }
    NttArgs a = c.ntt_args(0);
    a.in = a.out = acc; a.mod_count = Qp; a.polys_per_item = 2 * Qp;
    a.in_item_stride = a.out_item_stride = per;
    a.iepi.on = 1; a.iepi.limbs = Q; a.iepi.add_parts = add_parts; a.iepi.p_mod = Q; a.iepi.galois_elt = galois_elt;
    a.iepi.half = c.h64("half")[0]; a.iepi.half_mod = c.d64("half_mod"); a.iepi.inv = c.d64("last_q_modinv");
    a.iepi.ct = ct; a.iepi.ct_item_stride = cs;
    a.iepi.out = out; a.iepi.out_item_stride = so;
    return ntt_launch(a, 2 * Qp * batch, true, st);
}

// reference bfv/operator.cu:505-583
hipError_t op_bfv_relinearize(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                              hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const u64 per = ((u64) Q * Qp + 2 * Qp) * n;
    u64* temp1 = ws;
    u64* temp2 = ws + (u64) Q * Qp * n;
    const Mod* mods = c.plan_qp.mods;
    NttArgs a = c.ntt_args(0); // cipher_broadcast_kernel fused into the NTT load    :515-534
    a.in = ct + ((u64) Q << (np + 1)); a.out = temp1; a.mod_count = Qp; a.polys_per_item = Q * Qp;
    a.decomp_mods = Qp;
    a.in_item_stride = cs; a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp2, per, Q, Qp, Qp, 0, nullptr, 0, batch, st));             // :531-566
    if (c.P_size == 1 && c.fused_moddown)
        return bfv_intt_moddown(c, temp2, per, ct, cs, 2, ct, cs, 0, batch, st);            // :571-576 in one transform
    a.decomp_mods = 0; a.in_item_stride = per;
    a.in = temp2; a.out = temp2; a.polys_per_item = 2 * Qp;
    TRY(ntt_launch(a, 2 * Qp * batch, true, st));                                          // :571
    return rns_divide_round_lastq(temp2, per, ct, cs, ct, cs, mods, c.d64("half"), c.d64("half_mod"),
                                  c.d64("last_q_modinv"), np, Q, 0, batch, st);            // :576
}

// reference bfv/operator.cu:771-864 (c0 is read straight from the input
// ciphertext instead of being copied aside: out must not alias ct)
hipError_t op_bfv_apply_galois(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                               int galois_elt, int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const u64 per = ((u64) Q * Qp + 2 * Qp) * n;
    u64* temp1 = ws;
    u64* temp2 = ws + (u64) Q * Qp * n;
    const Mod* mods = c.plan_qp.mods;
    NttArgs a = c.ntt_args(0); // bfv_duplicate_kernel fused into the NTT load       :789-808
    a.in = ct + (u64) Q * n; a.out = temp1; a.mod_count = Qp; a.polys_per_item = Q * Qp;
    a.decomp_mods = Qp;
    a.in_item_stride = cs; a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp2, per, Q, Qp, Qp, 0, nullptr, 0, batch, st));             // :805-840
    if (c.P_size == 1 && c.fused_moddown)
        return bfv_intt_moddown(c, temp2, per, ct, cs, 1, out, so, galois_elt, batch, st);  // :846-853 in one transform
    a.decomp_mods = 0; a.in_item_stride = per;
    a.in = temp2; a.out = temp2; a.polys_per_item = 2 * Qp;
    TRY(ntt_launch(a, 2 * Qp * batch, true, st));                                          // :846
    return rns_moddown_permute(temp2, per, ct, cs, out, so, mods, c.d64("half"), c.d64("half_mod"),
                               c.d64("last_q_modinv"), galois_elt, np, Qp, Q, Qp, Q, c.P_size, batch, st); // :853
}

// ------------------------------------------------------------------ method II (P_size > 1)
static hipError_t dtoq(const Context& c, int lvl, const u64* in, u64 in_stride, u64* out, u64 out_stride, int l,
                       int level, int batch, hipStream_t st)
{
    const Context::M2Level& L = c.m2_levels[lvl];
    return rns_base_conversion_DtoQtilde(in, in_stride, out, out_stride, c.plan_qp.mods,
                                         c.d64("m2_matrix_mg") + L.off_matrix, c.d64("m2_Mi_inv") + L.off_mi,
                                         c.d64("m2_negprod_mg") + L.off_prod, c.d32("m2_I_j") + L.off_digits,
                                         c.d32("m2_I_location") + L.off_digits, c.n_power, L.d, L.rc, l, level,
                                         c.m2_width, batch, st);
}

// reference ckks/operator.cu:1025-1154
// CKKS key switching with several special primes, tail: acc [2][rc][N] (NTT domain) -> out [2][l][N] =
// moddown(acc) + ct (parts below add_parts; 0 = both) [-> Galois automorphism].  The reference runs the INTT of all
// 2 rc limbs, divide_round_lastq_extended_leveled_kernel, the NTT of 2 l limbs and an addition
// (ckks/operator.cu:1131-1149).  Here only the 2 P special limbs are inverse-transformed; from them one kernel
// forms, per limb of Q, the value u whose transform the forward pass's epilogue subtracts from the accumulated limb
// before multiplying by W0 = prod P_i^-1 (context.cpp m2_md_*): the same exact integer function, so the same
// residues.  scratch: [2][l][N] per item (`per` apart).
static hipError_t ckks_moddown_multi(const Context& c, u64* acc, u64* scratch, u64 per, const u64* ct, u64 cs,
                                     int add_parts, u64* out, u64 so, int depth, int galois_elt, int batch,
                                     hipStream_t st)
{
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size, P = c.P_size;
    const int l = Q - depth, rc = Qp - depth;
    for (int part = 0; part < 2; part++) {
        NttArgs a = c.ntt_args(0);
        a.in = a.out = acc + (u64) (part * rc + l) * n;
        a.mod_count = P; a.mod_offset = Q; a.polys_per_item = P;
        a.in_item_stride = a.out_item_stride = per;
        TRY(ntt_launch(a, P * batch, true, st));
    }
    TRY(rns_moddown_multi_stage_one(acc, per, scratch, per, c.plan_qp.mods, c.d64("half"), c.d64("half_mod"),
                                    c.d64("last_q_modinv"), c.d64("m2_md_G"), c.d64("m2_md_C"), c.n_power, rc, l, Qp, Q,
                                    P, batch, st));
    NttArgs a = c.ntt_args(0);
    a.in = a.out = scratch; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = a.out_item_stride = per;
    a.epi.on = 1;
    a.epi.ks = acc; a.epi.ks_item_stride = per; a.epi.ks_part_limbs = rc;
    a.epi.ct = ct; a.epi.ct_item_stride = cs; a.epi.ct_parts = add_parts;
    a.epi.out = out; a.epi.out_item_stride = so;
    a.epi.inv = c.d64("m2_md_W0");
    a.epi.limbs = l;
    a.epi.galois_inv = galois_elt ? (unsigned) inv_mod_2n((u64) galois_elt, 2 * c.n) : 0u;
    return ntt_launch(a, 2 * l * batch, false, st);
}

hipError_t op_ckks_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int depth, int batch, u64* ws,
                                  hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const int d = c.m2_levels[depth].d;
    const u64 per = ((u64) l * rc + 2 * rc) * n;
    u64* temp1 = ws;
    u64* temp2 = ws + (u64) l * rc * n;
    u64* c2 = ct + ((u64) l << (np + 1));
    const Mod* mods = c.plan_qp.mods;
    const int* order = c.d32("new_prime_locations") + prime_loc_offset(c, depth);
    NttArgs a = c.ntt_args(0);
    a.in = c2; a.out = c2; a.mod_count = l; a.polys_per_item = l;
    a.in_item_stride = a.out_item_stride = cs;
    TRY(ntt_launch(a, l * batch, true, st));                                               // :1052
    TRY(dtoq(c, depth, c2, cs, temp1, per, l, depth, batch, st));                          // :1065
    a = c.ntt_args(0);
    a.in = temp1; a.out = temp1; a.mod_count = rc; a.polys_per_item = d * rc; a.mod_order = order;
    a.in_item_stride = a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp2, per, d, rc, l, depth, nullptr, 0, batch, st));          // :1095-1125
    if (c.fused_moddown) // temp1 ([d][rc][N], the digits) is free again: scratch of the mod-down
        return ckks_moddown_multi(c, temp2, temp1, per, ct, cs, 0, ct, cs, depth, 0, batch, st);
    a.in = temp2; a.out = temp2; a.polys_per_item = 2 * rc;
    TRY(ntt_launch(a, 2 * rc * batch, true, st));                                          // :1131
    TRY(rns_moddown_extended(temp2, per, nullptr, 0, temp1, per, mods, c.d64("half"), c.d64("half_mod"),
                             c.d64("last_q_modinv"), np, rc, l, Qp, Q, c.P_size, 0, batch, st)); // :1136
    a = c.ntt_args(0);
    a.in = temp1; a.out = temp1; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = a.out_item_stride = per;
    TRY(ntt_launch(a, 2 * l * batch, false, st));                                          // :1145
    // addition(temp1, ct, ct): per-item strides differ, so one launch per item batch via copy-free add
    return rns_addition_strided(temp1, per, ct, cs, ct, cs, mods, np, l, 2, batch, st);     // :1149
}

// reference ckks/operator.cu:1561-1720
hipError_t op_ckks_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                   int galois_elt, int depth, int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const int d = c.m2_levels[depth].d;
    const u64 per = ((u64) 2 * l + (u64) l * rc + 2 * rc) * n;
    u64* temp0 = ws;
    u64* temp3 = temp0 + (u64) 2 * l * n;
    u64* temp4 = temp3 + (u64) l * rc * n;
    const Mod* mods = c.plan_qp.mods;
    const int* order = c.d32("new_prime_locations") + prime_loc_offset(c, depth);
    const bool ntt_domain = c.fused_moddown && c.ntt_galois;
    NttArgs a = c.ntt_args(0);
    a.in = ct; a.out = temp0; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = cs; a.out_item_stride = per;
    if (ntt_domain) { // c0 stays in the NTT domain (see op_ckks_apply_galois): only c1 is needed as coefficients
        a.in = ct + (u64) l * n; a.out = temp0 + (u64) l * n; a.polys_per_item = l;
        TRY(ntt_launch(a, l * batch, true, st));
    } else {
        TRY(ntt_launch(a, 2 * l * batch, true, st));
    }
    TRY(dtoq(c, depth, temp0 + (u64) l * n, per, temp3, per, l, depth, batch, st));
    a = c.ntt_args(0);
    a.in = temp3; a.out = temp3; a.mod_count = rc; a.polys_per_item = d * rc; a.mod_order = order;
    a.in_item_stride = a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp4, per, d, rc, l, depth, nullptr, 0, batch, st));
    if (ntt_domain) // temp0 is free again: scratch of the mod-down; c0 added and the automorphism applied by its epilogue
        return ckks_moddown_multi(c, temp4, temp0, per, ct, cs, 1, out, so, depth, galois_elt, batch, st);
    a.in = temp4; a.out = temp4; a.polys_per_item = 2 * rc;
    TRY(ntt_launch(a, 2 * rc * batch, true, st));
    TRY(rns_moddown_permute(temp4, per, temp0, per, out, so, mods, c.d64("half"), c.d64("half_mod"),
                            c.d64("last_q_modinv"), galois_elt, np, rc, l, Qp, Q, c.P_size, batch, st));
    a = c.ntt_args(0);
    a.in = out; a.out = out; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = a.out_item_stride = so;
    return ntt_launch(a, 2 * l * batch, false, st);
}

// Hoisted rotations: fast_single_hoisting_rotation_ckks_method_I / _II (reference ckks/operator.cu:4674-4953,
// 5092-5446).  The reference computes, for every requested Galois element, the same INTT of the ciphertext, the
// same digit decomposition (method I: ckks_duplicate_kernel, method II: base_conversion_DtoQtilde) and the same
// forward NTT of the digits -- none of them depends on the element -- and then the key inner product, the INTT,
// the mod-down + permutation and the final NTT that do.  Here the shared part runs once: the coefficient-domain
// ciphertext and the NTT-domain digits stay in the workspace while the per-element part walks the keys, so every
// output is bit-identical to the reference's (and to `count` separate hegpu_ckks_apply_galois calls).
// out: [count][2][l][N] per ciphertext (`so` apart), entry i at i * 2 l N; galois_elts[i] == 0 copies the input
// (global_memory_replace_kernel, :4708 / :5141).  Workspace: OP_CKKS_GALOIS.
hipError_t op_ckks_rotate_hoisted(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* const* keys,
                                  const int* galois_elts, int count, int depth, int batch, u64* ws, hipStream_t st,
                                  int group)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int l = Q - depth, rc = Qp - depth;
    const bool m2 = c.P_size > 1;
    const int digits = m2 ? c.m2_levels[depth].d : l;
    if (group != 1 && group != 4) return hipErrorInvalidValue;
    if (digits > 16) group = 1; // rns_keyswitch_mac_keys keeps the digits of a coefficient in 16 registers
    const u64 acc_words = (u64) 2 * rc * n;
    const u64 per = ((u64) 2 * l + (u64) l * rc) * n + group * acc_words;
    u64* temp0 = ws;                       // [2][l][N] coefficient-domain copy of ct
    u64* temp2 = temp0 + (u64) 2 * l * n;  // [digits][rc][N] NTT-domain digits
    u64* temp3 = temp2 + (u64) l * rc * n; // [group][2][rc][N]
    const Mod* mods = c.plan_qp.mods;
    const int* order = c.d32("new_prime_locations") + prime_loc_offset(c, depth);
    const u64 ct_words = (u64) 2 * l * n;
    bool any = false;
    for (int i = 0; i < count; i++) {
        if (galois_elts[i] == 0)
            TRY(rns_copy_limbs(ct, (u64) l * n, cs, out + (u64) i * ct_words, (u64) l * n, so, np, l, 2, batch, st));
        else if (!keys[i]) return hipErrorInvalidValue;
        else any = true;
    }
    if (!any) return hipSuccess;
    {   // a single element has nothing to share: the fused key-switch path of the plain operator is faster
        int nz = 0, which = -1;
        for (int i = 0; i < count; i++)
            if (galois_elts[i] != 0) { nz++; which = i; }
        if (nz == 1) {
            u64* oi = out + (u64) which * ct_words;
            return m2 ? op_ckks_apply_galois_II(c, ct, cs, oi, so, keys[which], galois_elts[which], depth, batch, ws, st)
                      : op_ckks_apply_galois(c, ct, cs, oi, so, keys[which], galois_elts[which], depth, batch, ws, st);
        }
    }

    // With the mod-down fused into its transform every element stays in the NTT domain -- inner product, INTT of
    // the special limbs, mod-down transform whose epilogue adds c0 and scatters through the automorphism (see
    // op_ckks_apply_galois / ckks_moddown_multi).  c0 is never transformed, so only c1 goes through the shared INTT.
    const bool ntt_domain = c.fused_moddown && c.ntt_galois;
    // ---- shared: INTT of both parts (of c1 only: ntt_domain), digits, forward NTT of the digits
    NttArgs a = c.ntt_args(0);
    a.in = ct; a.out = temp0; a.mod_count = l; a.polys_per_item = 2 * l;
    a.in_item_stride = cs; a.out_item_stride = per;
    if (ntt_domain) {
        a.in = ct + (u64) l * n; a.out = temp0 + (u64) l * n; a.polys_per_item = l;
        TRY(ntt_launch(a, l * batch, true, st));
    } else {
        TRY(ntt_launch(a, 2 * l * batch, true, st));
    }
    a = c.ntt_args(0);
    a.out = temp2; a.mod_count = rc; a.polys_per_item = digits * rc; a.mod_order = order;
    a.in_item_stride = a.out_item_stride = per;
    if (m2) {
        TRY(dtoq(c, depth, temp0 + (u64) l * n, per, temp2, per, l, depth, batch, st));
        a.in = temp2;
    } else {
        // digit d at modulus d is the NTT-domain limb of c1 itself
        a.in = temp0 + (u64) l * n; a.decomp_mods = rc; a.skip_identity = 1;
        TRY(rns_copy_diag(ct + (u64) l * n, cs, temp2, per, np, l, rc, batch, st));
    }
    TRY(ntt_launch(a, digits * rc * batch, false, st));

    // ---- per Galois element; with room for four accumulators the inner products of four elements share one read
    // of the digits (rns_keyswitch_mac_keys)
    int pending[4], npend = 0, next = 0;
    for (int i = 0; i < count; i++) {
        if (galois_elts[i] == 0) continue;
        u64* oi = out + (u64) i * ct_words;
        u64* acc = temp3;
        if (group > 1) {
            if (npend == next) { // start a new group
                npend = next = 0;
                const u64* gk[4];
                for (int j = i; j < count && npend < group; j++)
                    if (galois_elts[j] != 0) { pending[npend] = j; gk[npend++] = keys[j]; }
                TRY(rns_keyswitch_mac_keys(temp2, per, gk, npend, temp3, per, acc_words, mods, np, digits, rc, Qp, l, depth,
                                           batch, st));
            }
            acc = temp3 + (u64) next * acc_words; // pending[next] == i
            next++;
        } else {
            TRY(rns_keyswitch_mac(temp2, per, keys[i], temp3, per, mods, np, digits, rc, Qp, l, depth, batch, st));
        }
        if (ntt_domain) {
            // temp0 is free once the digits exist: scratch of the mod-down transform
            if (m2) TRY(ckks_moddown_multi(c, acc, temp0, per, ct, cs, 1, oi, so, depth, galois_elts[i], batch, st));
            else
                TRY(ckks_keyswitch_core(c, nullptr, 0, ct, cs, 1, oi, so, nullptr, depth, batch, temp0, acc, per, st,
                                        RELIN_PHASE_INTT_P | RELIN_PHASE_MODDOWN, galois_elts[i]));
            continue;
        }
        if (group > 1) return hipErrorInvalidValue; // the reference-order tail works on one accumulator
        NttArgs b = c.ntt_args(0);
        b.in = temp3; b.out = temp3; b.mod_count = rc; b.polys_per_item = 2 * rc; b.mod_order = order;
        b.in_item_stride = b.out_item_stride = per;
        TRY(ntt_launch(b, 2 * rc * batch, true, st));
        TRY(rns_moddown_permute(temp3, per, temp0, per, oi, so, mods, c.d64("half"), c.d64("half_mod"),
                                c.d64("last_q_modinv"), galois_elts[i], np, rc, l, Qp, Q, c.P_size, batch, st));
        b = c.ntt_args(0);
        b.in = oi; b.out = oi; b.mod_count = l; b.polys_per_item = 2 * l;
        b.in_item_stride = b.out_item_stride = so;
        TRY(ntt_launch(b, 2 * l * batch, false, st));
    }
    return hipSuccess;
}

// BFV key switching with several special primes, tail: acc [2][Q'][N] (NTT domain) -> out [2][Q][N] = moddown(acc) + ct
// (parts below add_parts) [-> Galois permutation].  The reference inverse-transforms all 2 Q' limbs and runs
// divide_round_lastq_extended_kernel / divide_round_lastq_permute_bfv_kernel (bfv/operator.cu:657-667, 948-963).  Here
// the 2 P special limbs are inverse-transformed first, one kernel forms u = sum_i lh_i G_i - C per limb of Q from them
// (the chain among the special limbs run once per coefficient, context.cpp m2_md_*), and the division (x - u) * W0,
// the added ciphertext and the permutation are the epilogue of the Q limbs' inverse transform (NttInvEpilogue::u):
// the same exact integer function, the coefficient-domain accumulator is never written or read back.
// scratch: [2][Q][N] per item (`per` apart).
static hipError_t bfv_intt_moddown_multi(const Context& c, u64* acc, u64* scratch, u64 per, const u64* ct, u64 cs,
                                         int add_parts, u64* out, u64 so, int galois_elt, int batch, hipStream_t st)
{
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size, P = c.P_size;
    for (int part = 0; part < 2; part++) {
        NttArgs a = c.ntt_args(0);
        a.in = a.out = acc + (u64) (part * Qp + Q) * n;
        a.mod_count = P; a.mod_offset = Q; a.polys_per_item = P;
        a.in_item_stride = a.out_item_stride = per;
        TRY(ntt_launch(a, P * batch, true, st));
    }
    TRY(rns_moddown_multi_stage_one(acc, per, scratch, per, c.plan_qp.mods, c.d64("half"), c.d64("half_mod"),
                                    c.d64("last_q_modinv"), c.d64("m2_md_G"), c.d64("m2_md_C"), c.n_power, Qp, Q, Qp, Q, P,
                                    batch, st));
    NttArgs a = c.ntt_args(0);
    a.in = a.out = acc; a.mod_count = Qp; a.polys_per_item = 2 * Qp;
    a.in_item_stride = a.out_item_stride = per;
    a.iepi.on = 1; a.iepi.limbs = Q; a.iepi.p_count = P; a.iepi.add_parts = add_parts; a.iepi.p_mod = Q;
    a.iepi.galois_elt = galois_elt;
    a.iepi.inv = c.d64("m2_md_W0");
    a.iepi.u = scratch; a.iepi.u_item_stride = per;
    a.iepi.ct = ct; a.iepi.ct_item_stride = cs;
    a.iepi.out = out; a.iepi.out_item_stride = so;
    return ntt_launch(a, 2 * Qp * batch, true, st);
}

// reference bfv/operator.cu:585-672
hipError_t op_bfv_relinearize_II(const Context& c, u64* ct, u64 cs, const u64* key, int batch, u64* ws,
                                 hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int d = c.m2_levels[0].d;
    const u64 per = ((u64) Q * Qp + 2 * Qp) * n;
    u64* temp1 = ws;
    u64* temp2 = ws + (u64) Q * Qp * n;
    const Mod* mods = c.plan_qp.mods;
    TRY(dtoq(c, 0, ct + ((u64) Q << (np + 1)), cs, temp1, per, Q, 0, batch, st));          // :599
    NttArgs a = c.ntt_args(0);
    a.in = temp1; a.out = temp1; a.mod_count = Qp; a.polys_per_item = d * Qp;
    a.in_item_stride = a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp2, per, d, Qp, Qp, 0, nullptr, 0, batch, st));             // :619-650
    if (c.fused_moddown) // temp1 (the digits) is free again: scratch of the mod-down
        return bfv_intt_moddown_multi(c, temp2, temp1, per, ct, cs, 2, ct, cs, 0, batch, st);        // :657-667 in one transform
    a.in = temp2; a.out = temp2; a.polys_per_item = 2 * Qp;
    TRY(ntt_launch(a, 2 * Qp * batch, true, st));                                          // :657
    return rns_moddown_extended(temp2, per, ct, cs, ct, cs, mods, c.d64("half"), c.d64("half_mod"),
                                c.d64("last_q_modinv"), np, Qp, Q, Qp, Q, c.P_size, 1, batch, st); // :662
}

// reference bfv/operator.cu:866-973
hipError_t op_bfv_apply_galois_II(const Context& c, const u64* ct, u64 cs, u64* out, u64 so, const u64* key,
                                  int galois_elt, int batch, u64* ws, hipStream_t st)
{
    if (batch <= 0) return hipSuccess; // an empty batch is a no-op, not an invalid launch
    const int np = c.n_power;
    const u64 n = c.n;
    const int Q = c.Q_size, Qp = c.Qp_size;
    const int d = c.m2_levels[0].d;
    const u64 per = ((u64) Q * Qp + 2 * Qp) * n;
    u64* temp2 = ws;
    u64* temp3 = ws + (u64) Q * Qp * n;
    const Mod* mods = c.plan_qp.mods;
    TRY(dtoq(c, 0, ct + (u64) Q * n, cs, temp2, per, Q, 0, batch, st));
    NttArgs a = c.ntt_args(0);
    a.in = temp2; a.out = temp2; a.mod_count = Qp; a.polys_per_item = d * Qp;
    a.in_item_stride = a.out_item_stride = per;
    TRY(keyswitch_ntt_mac(c, a, key, temp3, per, d, Qp, Qp, 0, nullptr, 0, batch, st));
    if (c.fused_moddown) // temp2 (the digits) is free again; c0 added to part 0, the permutation as the scatter of the store
        return bfv_intt_moddown_multi(c, temp3, temp2, per, ct, cs, 1, out, so, galois_elt, batch, st);
    a.in = temp3; a.out = temp3; a.polys_per_item = 2 * Qp;
    TRY(ntt_launch(a, 2 * Qp * batch, true, st));
    return rns_moddown_permute(temp3, per, ct, cs, out, so, mods, c.d64("half"), c.d64("half_mod"),
                               c.d64("last_q_modinv"), galois_elt, np, Qp, Q, Qp, Q, c.P_size, batch, st);
}

// ------------------------------------------------------------------ keygen / encrypt / decrypt
static u64 inv_mod_2n(u64 g, u64 two_n)
{
    // g odd, two_n a power of two: Newton iteration
    u64 x = g;
    for (int i = 0; i < 6; i++) x *= 2 - g * x;
    return x & (two_n - 1);
}

hipError_t op_gen_secret_key(const Context& c, Rng& r, int hamming_weight, u64* sk, u64* ws, hipStream_t st)
{
    const int n = (int) c.n;
    if (hamming_weight <= 0 || hamming_weight > n) return hipErrorInvalidValue;
    // partial Fisher-Yates over the coefficient indices (the reference's v2 generator
    // does the same on the host with mt19937, ckks/keygenerator.cu:100-118)
    std::vector<int> index(n), host(2 * (size_t) hamming_weight);
    for (int i = 0; i < n; i++) index[i] = i;
    const u64 stream = r.stream++;
    for (int i = 0; i < hamming_weight; i++) {
        const DrbgOut o = drbg_block(r.seed, stream, (u64) i);
        const int j = i + (int) (((u64) o.w[0] * (u64) (n - i)) >> 32);
        std::swap(index[i], index[j]);
        host[i] = index[i];
        host[hamming_weight + i] = (o.w[1] & 1) ? 1 : -1;
    }
    int* dpos = reinterpret_cast<int*>(ws);
    TRY(hipMemcpyAsync(dpos, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, st));
    TRY(hipStreamSynchronize(st)); // the staging vector dies with this call
    TRY(kg_secret_rns(dpos, dpos + hamming_weight, hamming_weight, sk, c.plan_qp.mods, c.n_power, c.Qp_size, st));
    NttArgs a = c.ntt_args(0);
    a.in = sk; a.out = sk; a.mod_count = c.Qp_size;
    return ntt_launch(a, c.Qp_size, false, st);
}

hipError_t op_gen_public_key(const Context& c, Rng& r, const u64* sk, u64* pk, u64* ws, hipStream_t st)
{
    const int Qp = c.Qp_size;
    u64* e = ws;
    u64* av = ws + (u64) Qp * c.n;
    TRY(kg_uniform(av, c.plan_qp.mods, c.n_power, Qp, 1, r.seed, r.stream++, st));
    TRY(kg_gaussian(e, c.plan_qp.mods, c.n_power, Qp, 1, r.seed, r.stream++, c.gauss_cdt, st));
    NttArgs a = c.ntt_args(0);
    a.in = e; a.out = e; a.mod_count = Qp;
    TRY(ntt_launch(a, Qp, false, st));
    return kg_publickey(pk, sk, e, av, c.plan_qp.mods, c.n_power, Qp, st);
}

hipError_t op_gen_switch_key(const Context& c, Rng& r, const u64* sk, int galois_elt, const u64* old_sk, u64* key,
                             u64* ws, hipStream_t st)
{
    const int Q = c.Q_size, Qp = c.Qp_size;
    // method I: one digit per ciphertext prime; method II: the depth-0 digit partition
    // (ckks/keygenerator.cu:326-414 uses d_leveled[0] and Sk_pair_leveled[0])
    const int d = c.P_size == 1 ? Q : c.m2_levels[0].d;
    const int width = c.P_size == 1 ? 1 : c.m2_width;
    u64* e = ws;
    u64* av = ws + (u64) d * Qp * c.n;
    TRY(kg_uniform(av, c.plan_qp.mods, c.n_power, Qp, d, r.seed, r.stream++, st));
    TRY(kg_gaussian(e, c.plan_qp.mods, c.n_power, Qp, d, r.seed, r.stream++, c.gauss_cdt, st));
    NttArgs a = c.ntt_args(0);
    a.in = e; a.out = e; a.mod_count = Qp;
    TRY(ntt_launch(a, d * Qp, false, st));
    const int inv = galois_elt ? (int) inv_mod_2n((u64) galois_elt, 2 * c.n) : 0; // keygenerator.cu:474
    return kg_switchkey(key, sk, e, av, c.plan_qp.mods, c.d64("factor"), inv, old_sk, c.n_power, Qp, d, width, Q,
                        c.P_size, st);
}

// (pk*u + e) / P with rounding: the common front of both encryptions (encryptor.cu:52-100)
static hipError_t encrypt_zero(const Context& c, Rng& r, const u64* pk, u64* ct, u64* ws, hipStream_t st)
{
    const int np = c.n_power, Q = c.Q_size, Qp = c.Qp_size;
    const u64 n = c.n;
    u64* u = ws;                       // [Q'][N]
    u64* e = u + (u64) Qp * n;         // [2][Q'][N]
    u64* pku = e + (u64) 2 * Qp * n;   // [2][Q'][N]
    const Mod* mods = c.plan_qp.mods;
    TRY(kg_ternary(u, mods, np, Qp, 1, r.seed, r.stream++, st));
    TRY(kg_gaussian(e, mods, np, Qp, 2, r.seed, r.stream++, c.gauss_cdt, st));
    NttArgs a = c.ntt_args(0);
    a.in = u; a.out = u; a.mod_count = Qp;
    TRY(ntt_launch(a, Qp, false, st));
    TRY(kg_pk_u(pk, u, pku, mods, np, Qp, st));
    a.in = pku; a.out = pku;
    TRY(ntt_launch(a, 2 * Qp, true, st));
    TRY(rns_addition(pku, e, pku, mods, np, Qp, 2, 1, 0, st));
    return rns_moddown_extended(pku, 0, nullptr, 0, ct, 0, mods, c.d64("half"), c.d64("half_mod"),
                                c.d64("last_q_modinv"), np, Qp, Q, Qp, Q, c.P_size, 0, 1, st);
}

hipError_t op_ckks_encrypt(const Context& c, Rng& r, const u64* pk, const u64* plain, u64* ct, u64* ws,
                           hipStream_t st)
{
    TRY(encrypt_zero(c, r, pk, ct, ws, st));                                               // :52-100
    NttArgs a = c.ntt_args(0);
    a.in = ct; a.out = ct; a.mod_count = c.Q_size;
    TRY(ntt_launch(a, 2 * c.Q_size, false, st));                                           // :103
    return kg_message_add(ct, plain, c.plan_qp.mods, c.n_power, c.Q_size, st);             // :107
}

hipError_t op_bfv_encrypt(const Context& c, Rng& r, const u64* pk, const u64* plain, u64* ct, u64* ws,
                          hipStream_t st)
{
    TRY(encrypt_zero(c, r, pk, ct, ws, st));
    return kg_bfv_message_add(ct, plain, c.plan_qp.mods, c.d64("coeff_div_plain_modulus"), c.h64("Q_mod_t")[0],
                              c.h64("upper_threshold")[0], c.plain_modulus, c.n_power, c.Q_size, st);
}

hipError_t op_bfv_decrypt(const Context& c, const u64* ct, const u64* sk, u64* plain, u64* ws, hipStream_t st)
{
    const int np = c.n_power, Q = c.Q_size;
    const u64* ct1 = ct + ((u64) Q << np);
    u64* t1 = ws; // [Q][N]
    NttArgs a = c.ntt_args(0);
    a.in = ct1; a.out = t1; a.mod_count = Q;
    TRY(ntt_launch(a, Q, false, st));                                                      // :62
    TRY(kg_sk_multiplication(t1, sk, t1, c.plan_qp.mods, np, Q, st));                      // :65
    a.in = t1; a.out = t1;
    TRY(ntt_launch(a, Q, true, st));                                                       // :101
    BfvDecryptDev d{};
    d.plain = make_mod(c.plain_modulus);
    d.gamma = make_mod(c.h64("gamma")[0]);
    d.Qi_t = c.d64("Qi_t"); d.Qi_gamma = c.d64("Qi_gamma"); d.Qi_inverse = c.d64("Qi_inverse");
    d.mulq_inv_t = c.h64("mulq_inv_t")[0];
    d.mulq_inv_gamma = c.h64("mulq_inv_gamma")[0];
    d.inv_gamma = c.h64("inv_gamma")[0];
    return kg_bfv_decryption(ct, t1, plain, c.plan_qp.mods, d, np, Q, st);                 // :107
}

hipError_t op_bfv_noise_rns(const Context& c, const u64* ct, const u64* sk, u64* out, hipStream_t st)
{
    const int np = c.n_power, Q = c.Q_size;
    NttArgs a = c.ntt_args(0);
    a.in = ct + ((u64) Q << np); a.out = out; a.mod_count = Q;
    TRY(ntt_launch(a, Q, false, st));
    TRY(kg_sk_multiplication(out, sk, out, c.plan_qp.mods, np, Q, st));
    a.in = out; a.out = out;
    TRY(ntt_launch(a, Q, true, st));
    return kg_coeff_multadd(ct, out, out, c.plain_modulus, c.plan_qp.mods, np, Q, st);
}

hipError_t op_bfv_encode(const Context& c, const long long* message, int message_size, u64* plain, hipStream_t st)
{
    if (!c.plan_plain.count) return hipErrorNotSupported;
    if (message_size < 0 || message_size > (int) c.n) return hipErrorInvalidValue;
    TRY(kg_bfv_encode_scatter(plain, message, c.d32("encoding_location"), c.plain_modulus, message_size, c.n_power,
                              st));                                                        // :66
    NttArgs a = c.ntt_args(2);
    a.in = plain; a.out = plain; a.mod_count = 1;
    return ntt_launch(a, 1, true, st);                                                     // :81
}

hipError_t op_bfv_decode(const Context& c, const u64* plain, u64* message, u64* ws, hipStream_t st)
{
    if (!c.plan_plain.count) return hipErrorNotSupported;
    NttArgs a = c.ntt_args(2);
    a.in = plain; a.out = ws; a.mod_count = 1;
    TRY(ntt_launch(a, 1, false, st));                                                      // :234
    return kg_bfv_decode_gather(message, ws, c.d32("encoding_location"), c.n_power, st);   // :239
}

hipError_t op_bfv_multiply_plain(const Context& c, const u64* ct, const u64* plain, u64* out, u64* ws, hipStream_t st)
{
    const int np = c.n_power, Q = c.Q_size;
    u64* pl = ws; // [Q][N]
    TRY(kg_bfv_threshold(plain, pl, c.plan_qp.mods, c.d64("upper_halfincrement"), c.h64("upper_threshold")[0], np, Q,
                         st));                                                             // :454
    NttArgs a = c.ntt_args(0);
    a.in = pl; a.out = pl; a.mod_count = Q;
    TRY(ntt_launch(a, Q, false, st));                                                      // :479
    a.in = ct; a.out = out;
    TRY(ntt_launch(a, 2 * Q, false, st));                                                  // :484
    TRY(kg_pk_u(out, pl, out, c.plan_qp.mods, np, Q, st));                                 // :489 cipherplain_kernel
    a.in = out; a.out = out;
    return ntt_launch(a, 2 * Q, true, st);                                                 // :496
}

// HEOperator<BFV>::transform_to_ntt_bfv_plain (bfv/operator.cu:1398-1431): threshold lift mod every q_j, forward NTT
hipError_t op_bfv_plain_to_ntt(const Context& c, const u64* plain, u64* out, hipStream_t st)
{
    TRY(kg_bfv_threshold(plain, out, c.plan_qp.mods, c.d64("upper_halfincrement"), c.h64("upper_threshold")[0],
                         c.n_power, c.Q_size, st));
    NttArgs a = c.ntt_args(0);
    a.in = out; a.out = out; a.mod_count = c.Q_size;
    return ntt_launch(a, c.Q_size, false, st);
}

static int log2i(u64 v)
{
    int r = 0;
    while ((1ull << r) < v) r++;
    return r;
}

// mode: 0 real slots, 1 complex slots ((re, im) pairs), 2 polynomial coefficients, 3 one value in every slot
// (message == host pointer to that value is NOT used: the value comes in `scalar`)
hipError_t op_ckks_encode(const Context& c, int mode, const double* message, int message_size, double scalar,
                          double scale, u64* plain, u64* ws, hipStream_t st)
{
    const int slots = (int) (c.n >> 1), Q = c.Q_size;
    if (mode == 3)                                                                         // encoder.cu:412-446
        return en_coeff_conversion(plain, nullptr, 0, scalar * scale, c.plan_qp.mods, Q, c.n_power, st);
    NttArgs a = c.ntt_args(0);
    a.in = plain; a.out = plain; a.mod_count = Q;
    if (mode == 2) {                                                                       // :222-261
        if (message_size < 0 || message_size > (int) c.n) return hipErrorInvalidValue;
        TRY(en_coeff_conversion(plain, message, message_size, scale, c.plan_qp.mods, Q, c.n_power, st));
        return ntt_launch(a, Q, false, st);
    }
    if (message_size < 0 || message_size > slots) return hipErrorInvalidValue;
    void* cbuf = ws; // slots complex doubles = N words
    const double fix = scale / (double) slots;                                             // :127
    // double -> complex (:120) in the transform's first load
    TRY(en_special_fft(cbuf, c.d64("special_ifft_roots_table"), log2i(slots), true, fix, st, message, message_size, mode == 1));
    TRY(en_conversion(plain, cbuf, c.plan_qp.mods, Q, c.d32("reverse_order"), c.n_power, st)); // :138
    return ntt_launch(a, Q, false, st);                                                    // :153
}

// mode: 0 the N/2 real parts, 1 the N/2 complex slots, 2 the N coefficients
hipError_t op_ckks_decode(const Context& c, int mode, const u64* plain, int depth, double scale, double* message,
                          u64* ws, hipStream_t st)
{
    const int slots = (int) (c.n >> 1), l = c.Q_size - depth;
    if (l < 1) return hipErrorInvalidValue;
    u64* coeff = ws;                      // [l][N]
    void* cbuf = ws + (u64) l * c.n;      // slots complex doubles
    NttArgs a = c.ntt_args(0);
    a.in = plain; a.out = coeff; a.mod_count = l;
    TRY(ntt_launch(a, l, true, st));                                                       // :469
    int counter = c.Q_size, loc1 = 0, loc2 = 0;                                            // :474-482
    for (int i = 0; i < depth; i++) { loc1 += counter; loc2 += counter * counter; counter--; }
    if (mode == 2)                                                                         // :586-635
        return en_coeff_compose(message, coeff, c.plan_qp.mods, c.d64("Mi_inv") + loc1, c.d64("Mi") + loc2,
                                c.d64("upper_half_threshold") + loc1, c.d64("decryption_modulus") + loc1, l, scale,
                                c.n_power, st);
    TRY(en_compose(cbuf, coeff, c.plan_qp.mods, c.d64("Mi_inv") + loc1, c.d64("Mi") + loc2,
                   c.d64("upper_half_threshold") + loc1, c.d64("decryption_modulus") + loc1, l, scale,
                   c.d32("reverse_order"), c.n_power, st));                                // :485
    // :502, complex -> double (:505) in the transform's last store
    return en_special_fft(cbuf, c.d64("special_fft_roots_table"), log2i(slots), false, 1.0, st, nullptr, 0, 0, message, mode == 1);
}

hipError_t op_ckks_decrypt(const Context& c, const u64* ct, const u64* sk, int depth, u64* plain, hipStream_t st)
{
    const int l = c.Q_size - depth;
    if (l < 1) return hipErrorInvalidValue;
    return kg_sk_multiplication_ckks(ct, plain, sk, c.plan_qp.mods, c.n_power, l, st);
}

} // namespace hegpu

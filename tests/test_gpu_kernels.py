"""GPU parity tests that close the holes VERDICT round 1 listed:

* config C4's exact chain (CKKS N=2^16, {60,50x15}|{60}) through multiply ->
  relinearize -> rotate, bit for bit against the oracle;
* the operator sequences with each fusion switched off (HEGPU_FUSED_ROW_MAC=0,
  HEGPU_FUSED_MODDOWN=0, HEGPU_FP_NTT=0), so that the stand-alone kernels
  (k_decompose, k_keyswitch_mac, k_moddown_stage_one/two, k_copy_diag) and the
  integer butterflies on a < 2^50 chain run on the device;
* one direct oracle comparison per kernel-level C-ABI entry of include/hegpu.h
  (the entries INTEGRATION.md tells a maintainer to bind);
* config C5 at a grid that fills the GPU (4096 gates).
"""
import contextlib
import ctypes
import os

import numpy as np
import pytest

from helpers import backend_switches, extreme_limbs, synth_ct, synth_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _ckks(hg, oracle, n, log_q, log_p, sec=None):
    c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_128 if sec is None else sec)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, c.n_power, primes, len(log_q), len(log_p))
    c.upload()
    return c, o, primes


def _bfv(hg, oracle, n, t):
    c = hg.Context.from_default(hg.BFV, n, 1, t)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, c.Q_size, c.P_size, t)
    c.upload()
    return c, o, primes


# ------------------------------------------------------------------ config C4, exact chain
@pytest.mark.parametrize("col_multi,batch", [(None, 3), (1, 3), (0, 3), (None, 2), (None, 1)],
                         ids=["auto", "col_multi", "col_per_poly", "two_ciphertexts", "one_ciphertext"])
def test_c4_chain_multiply_relinearize_rotate(hg, oracle, torch, col_multi, batch):
    """BASELINE.json config C4 / bench.py's workload: CKKS N=2^16, Q = {60, 50 x 15}, P = {60}, depth 0,
    batch 3 (two distinct pairs + a twin): multiply -> relinearize_inplace -> rotate by one slot, every
    limb compared with the oracle.  Both forms of the decomposing column pass (bench.py's batch of 64
    takes the multi-modulus one, a batch of 3 would not on its own); with one to five ciphertexts the launch-size
    rules pick the fused key switch in four pieces per unit, integer and FP64 moduli in one grid (ops.cpp:
    fused_digit_splits, ks_row_mac_split); the unfused sequence is forced by the switch sets below."""
    n = 65536
    with backend_switches(**({} if col_multi is None else dict(HEGPU_COL_MULTI=col_multi))):
        c, o, primes = _ckks(hg, oracle, n, [60] + [50] * 15, [60])
    Q, Qp = 16, 17
    assert (c.Q_size, c.Q_prime_size) == (Q, Qp)
    uniq = min(2, batch)
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 4)
    ct1 = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(uniq)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(uniq)]
    d1 = hg.to_device(np.concatenate([ct1[b % uniq] for b in range(batch)]))
    d2 = hg.to_device(np.concatenate([ct2[b % uniq] for b in range(batch)]))
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, 0, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    want = [o.ckks_multiply(ct1[b], ct2[b], 0) for b in range(uniq)]
    for b in range(batch):
        assert np.array_equal(got[b], want[b % uniq]), ("multiply", b)
    c.ckks_relinearize_inplace(out, 3 * Q * n, hg.to_device(key), 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(uniq):
        o.ckks_relinearize(want[b], key, 0)
    for b in range(batch):
        assert np.array_equal(got[b][:2 * Q * n], want[b % uniq][:2 * Q * n]), ("relinearize", b)
    g = hg.steps_to_galois_elt(1, n, 5)
    rot = torch.empty(batch * 2 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(out, 3 * Q * n, rot, 2 * Q * n, hg.to_device(gkey), g, 0, batch,
                        c.workspace(hg.OP_CKKS_GALOIS, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(rot).reshape(batch, -1)
    for b in range(batch):
        w = o.ckks_apply_galois(want[b % uniq][:2 * Q * n].copy(), gkey, g, 0)
        assert np.array_equal(got[b], w), ("rotate", b)


# ------------------------------------------------------------------ fusions switched off
@pytest.mark.parametrize("sw,batch,key_kind", [
    (dict(), 8, "max"), (dict(), 8, "random"), (dict(), 2, "max"), (dict(HEGPU_FP_NTT=0), 8, "max"),
    (dict(HEGPU_FUSED_ROW_MAC=0), 4, "max")],
    ids=["default_batch8_key_max", "default_batch8_key_random", "default_batch2_split", "integer_butterflies", "unfused"])
def test_c4_shape_key_switch_extreme_values(hg, oracle, torch, sw, batch, key_kind):
    """The fused key switch at config C4's EXACT shape -- N = 2^16 (FpColSched<8>), {60, 50 x 15 | 60}, 16 digits: the
    three-digit re-centring cadence of ks_row_mac_fp five times over -- with inputs at their extremes instead of
    random ones: every residue q - 1, alternating 0 / q - 1, a single spike, q / 2, each in the NTT
    domain and in the coefficient domain (so that the decomposed digits themselves are the extremes), against a key of
    all q - 1 and a random one.  Eight ciphertexts take the kernels of the bench (ntt_fwd_col_multi<8> +
    ks_row_mac_fp), two the small-launch form (ks_row_mac_split); the same under HEGPU_FP_NTT=0 (integer butterflies
    on the same chain) and with the fusion off.  Canonical mult / add semantics: switchkey.cu:61-162.
    (tests/test_gpu_fp_audit.py runs these inputs through the instrumented build and records the magnitudes.)"""
    n = 65536
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [60] + [50] * 15, [60])
    Q, Qp = 16, 17
    pats = ["max", "alt", "spike", "max_coeff", "alt_coeff", "alt3_coeff", "half_coeff", "random"][:batch] if batch >= 4 \
        else ["max_coeff", "alt_coeff"]
    if key_kind == "max":
        key = np.concatenate([np.full(n, primes[j] - 1, dtype=np.uint64) for _ in range(Q) for _c in range(2) for j in range(Qp)])
    else:
        key = synth_key(primes, Q, Qp, n, 3)
    cts = [np.concatenate([extreme_limbs(c, primes, range(Q), n, pat, 31 * i + p) for p in range(3)]) for i, pat in enumerate(pats)]
    d = hg.to_device(np.concatenate(cts))
    c.ckks_relinearize_inplace(d, 3 * Q * n, hg.to_device(key), 0, len(cts), c.workspace(hg.OP_CKKS_RELIN, 0, len(cts)))
    torch.cuda.synchronize()
    got = hg.to_host(d).reshape(len(cts), -1)
    for b, pat in enumerate(pats):
        want = o.ckks_relinearize(cts[b].copy(), key, 0)
        assert np.array_equal(got[b][:2 * Q * n], want[:2 * Q * n]), (pat, key_kind)


_SWITCHES = [
    dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_SINGLE_PASS=1),  # forced: a small launch would pick the other forms
    dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_SINGLE_PASS=0),
    dict(HEGPU_SINGLE_PASS=0),
    dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1),
    dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1, HEGPU_FP_NTT=0),
    dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1, HEGPU_FUSE_INVERSE=0),
    dict(HEGPU_NTT_GALOIS=0),
    dict(HEGPU_GALOIS_SCATTER=0),
    dict(HEGPU_FUSED_ROW_MAC=0),
    dict(HEGPU_FUSED_MODDOWN=0),
    dict(HEGPU_COPY_ALONG=0),  # the rescale's copy of the kept limbs as its own launch
    dict(HEGPU_DIGIT_SPLIT=2),  # fused key switch, two / four workgroups per unit over parts of the digits
    dict(HEGPU_DIGIT_SPLIT=4, HEGPU_COL_MULTI=1),
    dict(HEGPU_FP_NTT=0),
    dict(HEGPU_FUSED_ROW_MAC=0, HEGPU_FUSED_MODDOWN=0, HEGPU_FP_NTT=0),
    # decomposing launches as ONE pass (ntt_fwd_single<S1, true>): the digits of the unfused key switch and the
    # mod-down transform with its epilogue, FP64 and integer moduli
    dict(HEGPU_FUSED_ROW_MAC=0, HEGPU_SINGLE_PASS=1, HEGPU_COL_MULTI=0),
    dict(HEGPU_FUSED_ROW_MAC=0, HEGPU_SINGLE_PASS=1, HEGPU_COL_MULTI=0, HEGPU_FP_NTT=0),
]


@pytest.mark.parametrize("sw", _SWITCHES, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
@pytest.mark.parametrize("depth", [0, 2])
def test_ckks_sequence_with_fusions_off(hg, oracle, torch, sw, depth):
    """Config C2 chain (CKKS N=2^14, {50,40x7}|{50}): multiply -> relinearize -> rescale -> rotate with a
    fusion disabled: the unfused path launches the reference's kernel sequence one to one
    (k_copy_diag + two-pass NTT + k_keyswitch_mac, k_moddown_stage_one / stage_two, integer butterflies)."""
    n = 16384
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [50] + [40] * 7, [50])
    Q, Qp = 8, 9
    l = Q - depth
    batch = 2
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 5)
    ct1 = [synth_ct(primes, range(l), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(l), 2, n, 2 + 10 * b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * l * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, 2 * l * n, d2, 2 * l * n, out, 3 * l * n, depth, batch)
    c.ckks_relinearize_inplace(out, 3 * l * n, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    want = [o.ckks_relinearize(o.ckks_multiply(ct1[b], ct2[b], depth), key, depth) for b in range(batch)]
    for b in range(batch):
        assert np.array_equal(got[b][:2 * l * n], want[b][:2 * l * n]), "relinearize"
    g = hg.steps_to_galois_elt(3, n, 5)
    rot = torch.empty(batch * 2 * l * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(d1, 2 * l * n, rot, 2 * l * n, hg.to_device(gkey), g, depth, batch,
                        c.workspace(hg.OP_CKKS_GALOIS, depth, batch))
    torch.cuda.synchronize()
    got_r = hg.to_host(rot).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(got_r[b], o.ckks_apply_galois(ct1[b], gkey, g, depth)), "rotate"
    c.ckks_rescale_inplace(out, 3 * l * n, depth, batch, c.workspace(hg.OP_CKKS_RESCALE, depth, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        w = o.ckks_rescale(want[b][:2 * l * n].copy(), depth)
        assert np.array_equal(got[b][:2 * (l - 1) * n], w[:2 * (l - 1) * n]), "rescale"


@pytest.mark.parametrize("sw", [dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1),
                                dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=0, HEGPU_SINGLE_PASS=1),
                                dict(HEGPU_DIGIT_SPLIT=2)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
@pytest.mark.parametrize("n_power,depth", [(12, 0), (13, 1), (15, 0), (15, 2)])
def test_fused_key_switch_at_every_degree(hg, oracle, torch, sw, n_power, depth):
    """The fused forms (multi-modulus decomposing column pass finishing the inverse transform of its source, row
    pass + inner product, mod-down as load transform / epilogue) are picked by launch size, which the small tests
    never reach below N = 2^14: forced here at N = 2^12, 2^13, 2^15 (the kernels index tiles, rows and twiddles by
    N) on a chain with a 60-bit first prime and a 60-bit special prime next to FP64-path primes, leveled."""
    n = 1 << n_power
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [60, 45, 45, 45, 45], [60], sec=hg.SEC_NONE)
    Q, Qp = 5, 6
    l = Q - depth
    batch = 2
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 5)
    ct1 = [synth_ct(primes, range(l), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(l), 2, n, 2 + 10 * b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * l * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, 2 * l * n, d2, 2 * l * n, out, 3 * l * n, depth, batch)
    c.ckks_relinearize_inplace(out, 3 * l * n, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
    g = hg.steps_to_galois_elt(2, n, 5)
    rot = torch.empty(batch * 2 * l * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(d1, 2 * l * n, rot, 2 * l * n, hg.to_device(gkey), g, depth, batch,
                        c.workspace(hg.OP_CKKS_GALOIS, depth, batch))
    torch.cuda.synchronize()
    got, got_r = hg.to_host(out).reshape(batch, -1), hg.to_host(rot).reshape(batch, -1)
    want = [o.ckks_relinearize(o.ckks_multiply(ct1[b], ct2[b], depth), key, depth) for b in range(batch)]
    for b in range(batch):
        assert np.array_equal(got[b][:2 * l * n], want[b][:2 * l * n]), "relinearize"
        assert np.array_equal(got_r[b], o.ckks_apply_galois(ct1[b], gkey, g, depth)), "rotate"
    if l > 1:
        c.ckks_rescale_inplace(out, 3 * l * n, depth, batch, c.workspace(hg.OP_CKKS_RESCALE, depth, batch))
        torch.cuda.synchronize()
        got = hg.to_host(out).reshape(batch, -1)
        for b in range(batch):
            w = o.ckks_rescale(want[b][:2 * l * n].copy(), depth)
            assert np.array_equal(got[b][:2 * (l - 1) * n], w[:2 * (l - 1) * n]), "rescale"


@pytest.mark.parametrize("split", [0, 1], ids=["one_thread_per_coefficient", "rows_over_four_wavefronts"])
@pytest.mark.parametrize("n_power", [12, 14, 15])
def test_bfv_multiply_both_behz_forms(hg, oracle, torch, split, n_power):
    """BFV multiply on the default chains (Q = 2 / 8 / 14 with 3 / 9 / 15 auxiliary primes): the two BEHZ kernels
    with one thread per coefficient and with the rows of the base conversions spread over the four wavefronts of a
    workgroup (picked for launches below 320 workgroups; the context option behz_split may be changed between calls)."""
    n, t = 1 << n_power, 786433
    c, o, primes = _bfv(hg, oracle, n, t)
    Q = c.Q_size
    batch = 2
    ct1 = [synth_ct(primes, range(Q), 2, n, 21 + b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 31 + b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.set_option("behz_split", split)
    c.bfv_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, batch, c.workspace(hg.OP_BFV_MULTIPLY, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(got[b], o.bfv_multiply(ct1[b], ct2[b])), b


@pytest.mark.parametrize("sw", [dict(), dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_SINGLE_PASS=1), dict(HEGPU_SINGLE_PASS=0),
                                dict(HEGPU_DIGIT_SPLIT=4)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()) or "default")
@pytest.mark.parametrize("n_power", [13, 14, 16])
def test_bfv_key_switch_at_every_degree(hg, oracle, torch, sw, n_power):
    """BFV relinearize + rotate on the default chains of N = 2^13, 2^14, 2^16 (C1 covers 2^12, C3 2^15): the
    inverse transform that carries the mod-down and the Galois permutation as its epilogue exists as a single-pass
    kernel (N <= 2^14) and as the column pass of the two-pass form, each indexed by N."""
    n, t = 1 << n_power, 786433
    with backend_switches(**sw):
        c, o, primes = _bfv(hg, oracle, n, t)
    Q, Qp = c.Q_size, c.Q_prime_size
    batch = 2
    key, gkey = synth_key(primes, Q, Qp, n, 3), synth_key(primes, Q, Qp, n, 4)
    ct3 = [synth_ct(primes, range(Q), 3, n, 11 + b) for b in range(batch)]
    d = hg.to_device(np.concatenate(ct3))
    c.bfv_relinearize_inplace(d, 3 * Q * n, hg.to_device(key), batch, c.workspace(hg.OP_BFV_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(d).reshape(batch, -1)
    want = [o.bfv_relinearize(ct3[b].copy(), key)[:2 * Q * n] for b in range(batch)]
    for b in range(batch):
        assert np.array_equal(got[b][:2 * Q * n], want[b]), "relinearize"
    g = hg.steps_to_galois_elt(1, n, 3)
    src = hg.to_device(np.concatenate(want))
    rot = torch.empty_like(src)
    c.bfv_apply_galois(src, 2 * Q * n, rot, 2 * Q * n, hg.to_device(gkey), g, batch, c.workspace(hg.OP_BFV_GALOIS, 0, batch))
    torch.cuda.synchronize()
    gr = hg.to_host(rot).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(gr[b], o.bfv_apply_galois(np.ascontiguousarray(want[b]), gkey, g)), "rotate"


@pytest.mark.parametrize("sw", [dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_SINGLE_PASS=1), dict(HEGPU_FUSED_ROW_MAC=0),
                                dict(HEGPU_FP_NTT=0), dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1),
                                dict(HEGPU_SINGLE_PASS=0), dict(HEGPU_FUSED_MODDOWN=0),
                                dict(HEGPU_FUSED_MODDOWN=0, HEGPU_SINGLE_PASS=1)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_bfv_sequence_with_fusions_off(hg, oracle, torch, sw):
    """Config C1 shapes (BFV N=2^12 default chain): relinearize + rotate through the unfused key switch."""
    n, t = 4096, 1032193
    with backend_switches(**sw):
        c, o, primes = _bfv(hg, oracle, n, t)
    Q, Qp, batch = 2, 3, 2
    key = synth_key(primes, Q, Qp, n, 3)
    ct1 = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, batch, c.workspace(hg.OP_BFV_MULTIPLY, 0, batch))
    c.bfv_relinearize_inplace(out, 3 * Q * n, hg.to_device(key), batch, c.workspace(hg.OP_BFV_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        w = o.bfv_relinearize(o.bfv_multiply(ct1[b], ct2[b]), key)
        assert np.array_equal(got[b][:2 * Q * n], w[:2 * Q * n]), "bfv relinearize"
    g = hg.steps_to_galois_elt(1, n, 3)
    rot = torch.empty(batch * 2 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_apply_galois(d1, 2 * Q * n, rot, 2 * Q * n, hg.to_device(key), g, batch, c.workspace(hg.OP_BFV_GALOIS, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(rot).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(got[b], o.bfv_apply_galois(ct1[b], key, g)), "bfv rotate"


@pytest.mark.parametrize("depth", [0, 1])
def test_keyswitch_mixed_widths_multi_modulus_column_pass(hg, oracle, torch, depth):
    """The multi-modulus column pass on a chain that mixes FP64 targets (50/36/45/49 bits) with integer
    ones (60/55 bits) and narrow with wide source digits, inputs at their extremes (every residue q-1,
    0/q-1 patterns, a key of all q-1), N = 2^12 and 2^14."""
    for n in (4096, 16384):
        bits = [60, 50, 36, 45, 55, 49]
        with backend_switches(HEGPU_COL_MULTI=1, HEGPU_FUSED_ROW_MAC=1):
            c, o, primes = _ckks(hg, oracle, n, bits, [60], sec=hg.SEC_NONE)
        Q, Qp = len(bits), len(bits) + 1
        l = Q - depth
        key = np.concatenate([np.full(n, primes[j] - 1, dtype=np.uint64) for _ in range(Q) for _c in range(2)
                              for j in range(Qp)])
        cts = []
        full = np.concatenate([np.full(n, primes[j] - 1, dtype=np.uint64) for _p in range(3) for j in range(l)])
        cts.append(full)
        pat = full.copy().reshape(3 * l, n); pat[:, ::3] = 0
        cts.append(pat.reshape(-1))
        cts.append(synth_ct(primes, range(l), 3, n, 77))
        batch = len(cts)
        d = hg.to_device(np.concatenate(cts))
        c.ckks_relinearize_inplace(d, 3 * l * n, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
        torch.cuda.synchronize()
        got = hg.to_host(d).reshape(batch, -1)
        for b in range(batch):
            want = o.ckks_relinearize(cts[b].copy(), key, depth)
            assert np.array_equal(got[b][:2 * l * n], want[:2 * l * n]), (n, b)
        if l >= 2:
            cc = [x[:2 * l * n].copy() for x in cts]
            d = hg.to_device(np.concatenate(cc))
            c.ckks_rescale_inplace(d, 2 * l * n, depth, batch, c.workspace(hg.OP_CKKS_RESCALE, depth, batch))
            torch.cuda.synchronize()
            got = hg.to_host(d).reshape(batch, -1)
            for b in range(batch):
                w = o.ckks_rescale(cc[b].copy(), depth)
                assert np.array_equal(got[b][:2 * (l - 1) * n], w[:2 * (l - 1) * n]), ("rescale", n, b)


# ------------------------------------------------------------------ kernel-level C-ABI entries
def _limbs(oracle, primes, limb_ids, n, seed, count=1):
    """`count` polynomials per listed limb, canonical residues: [count][len(limb_ids)][n]"""
    return np.concatenate([oracle.fill_poly(seed + 97 * k, lid, n, primes[lid]) for k in range(count)
                           for lid in limb_ids])


def test_kernel_cross_multiplication(hg, oracle, torch):
    """hegpu_cross_multiplication (multiplication.cu:102-126) on the Q' tables and on the merged q|Bsk tables."""
    n = 4096
    c, o, primes = _bfv(hg, oracle, n, 1032193)
    batch = 2
    for table_set, mods_list in ((hg.TABLES_QP, primes), (hg.TABLES_Q_BSK, [int(v) for v in c.table("q_Bsk_merge_modulus")])):
        L = len(mods_list)
        a = [_limbs(oracle, mods_list, list(range(L)) * 2, n, 11 + b) for b in range(batch)]
        b_ = [_limbs(oracle, mods_list, list(range(L)) * 2, n, 31 + b) for b in range(batch)]
        mods = o.mods(mods_list)
        out = torch.empty(batch * 3 * L * n, dtype=torch.int64, device="cuda")
        c.cross_multiplication(hg.to_device(np.concatenate(a)), 2 * L * n, hg.to_device(np.concatenate(b_)), 2 * L * n,
                               out, 3 * L * n, L, batch, table_set=table_set)
        torch.cuda.synchronize()
        got = hg.to_host(out).reshape(batch, -1)
        for b in range(batch):
            want = np.zeros(3 * L * n, dtype=np.uint64)
            o.L.o_cross_multiplication(a[b].ctypes.data, b_[b].ctypes.data, want.ctypes.data, mods, 12, L)
            assert np.array_equal(got[b], want), (table_set, b)


def test_kernel_cipher_broadcast_and_keyswitch_mac(hg, oracle, torch):
    """hegpu_cipher_broadcast (cipher_broadcast_kernel switchkey.cu:11-27) and
    hegpu_keyswitch_multiply_accumulate (:61-162), non-leveled forms as bfv/operator.cu:521-561 calls them."""
    n = 4096
    c, o, primes = _bfv(hg, oracle, n, 1032193)
    Q, Qp, batch = 2, 3, 3
    src = [_limbs(oracle, primes, range(Q), n, 5 + b) for b in range(batch)]
    out = torch.empty(batch * Q * Qp * n, dtype=torch.int64, device="cuda")
    c.cipher_broadcast(hg.to_device(np.concatenate(src)), Q * n, out, Q * Qp * n, Q, Qp, Qp, 0, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    want = []
    for b in range(batch):
        w = np.zeros(Q * Qp * n, dtype=np.uint64)
        o.L.o_cipher_broadcast(src[b].ctypes.data, w.ctypes.data, o.qp_mods, 12, Q, Qp)
        assert np.array_equal(got[b], w), ("broadcast", b)
        want.append(w)
    # the inner product takes NTT-domain digits: any canonical residues do
    key = synth_key(primes, Q, Qp, n, 3)
    dig = [np.concatenate([_limbs(oracle, primes, range(Qp), n, 70 + 7 * b + d) for d in range(Q)]) for b in range(batch)]
    acc = torch.empty(batch * 2 * Qp * n, dtype=torch.int64, device="cuda")
    c.keyswitch_multiply_accumulate(hg.to_device(np.concatenate(dig)), Q * Qp * n, hg.to_device(key), acc, 2 * Qp * n,
                                    Q, Qp, Qp, Qp, 0, batch)
    torch.cuda.synchronize()
    got = hg.to_host(acc).reshape(batch, -1)
    for b in range(batch):
        w = np.zeros(2 * Qp * n, dtype=np.uint64)
        o.L.o_keyswitch_mac(dig[b].ctypes.data, key.ctypes.data, w.ctypes.data, o.qp_mods, 12, Qp, Q)
        assert np.array_equal(got[b], w), ("mac", b)


@pytest.mark.parametrize("depth", [0, 1, 3])
def test_kernel_leveled_broadcast_and_mac(hg, oracle, torch, depth):
    """cipher_broadcast_leveled_kernel (switchkey.cu:29-59) and keyswitch_multiply_accumulate_leveled_kernel
    (:164-285) through the same two entries (split = l, level = depth), CKKS N=2^13 {40,35x4}|{40}."""
    n = 8192
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35, 35], [40], sec=hg.SEC_NONE)
    Q, Qp = 5, 6
    l, rc = Q - depth, Qp - depth
    batch = 2
    src = [_limbs(oracle, primes, range(l), n, 9 + b) for b in range(batch)]
    out = torch.empty(batch * l * rc * n, dtype=torch.int64, device="cuda")
    c.cipher_broadcast(hg.to_device(np.concatenate(src)), l * n, out, l * rc * n, l, rc, l, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        w = np.zeros(l * rc * n, dtype=np.uint64)
        o.L.o_cipher_broadcast_leveled(src[b].ctypes.data, w.ctypes.data, o.qp_mods, Qp, rc, 13, l)
        assert np.array_equal(got[b], w), ("broadcast", b)
    key = synth_key(primes, Q, Qp, n, 3)
    limb_ids = list(range(l)) + [Q]  # rows 0..l-1 use q_y, row l the special prime
    dig = [np.concatenate([_limbs(oracle, primes, limb_ids, n, 40 + 7 * b + d) for d in range(l)]) for b in range(batch)]
    acc = torch.empty(batch * 2 * rc * n, dtype=torch.int64, device="cuda")
    c.keyswitch_multiply_accumulate(hg.to_device(np.concatenate(dig)), l * rc * n, hg.to_device(key), acc, 2 * rc * n,
                                    l, rc, Qp, l, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(acc).reshape(batch, -1)
    for b in range(batch):
        w = np.zeros(2 * rc * n, dtype=np.uint64)
        o.L.o_keyswitch_mac_leveled(dig[b].ctypes.data, key.ctypes.data, w.ctypes.data, o.qp_mods, Qp, l, 13)
        assert np.array_equal(got[b], w), ("mac", b)


@pytest.mark.parametrize("switchkey", [0, 1])
def test_kernel_divide_round_lastq(hg, oracle, torch, switchkey):
    """hegpu_divide_round_lastq (divide_round_lastq_kernel / _switchkey_kernel, switchkey.cu:400-478)."""
    n = 4096
    c, o, primes = _bfv(hg, oracle, n, 1032193)
    Q, Qp, batch = 2, 3, 2
    src = [_limbs(oracle, primes, list(range(Qp)) * 2, n, 3 + b) for b in range(batch)]
    for s in src:
        s[:4] = [0, primes[0] - 1, 1, primes[0] // 2]
        s[Q * n:Q * n + 3] = [0, primes[Q] - 1, primes[Q] // 2]  # the P limb at its corners
    cts = [synth_ct(primes, range(Q), 2, n, 50 + b) for b in range(batch)]
    out = torch.empty(batch * 2 * Q * n, dtype=torch.int64, device="cuda")
    c.divide_round_lastq(hg.to_device(np.concatenate(src)), 2 * Qp * n, hg.to_device(np.concatenate(cts)), 2 * Q * n, out,
                         2 * Q * n, switchkey, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    half, half_mod, inv = o.table("half"), o.table("half_mod"), o.table("last_q_modinv")
    for b in range(batch):
        w = np.zeros(2 * Q * n, dtype=np.uint64)
        o.L.o_divide_round_lastq(src[b].ctypes.data, cts[b].ctypes.data, w.ctypes.data, o.qp_mods, half.ctypes.data,
                                 half_mod.ctypes.data, inv.ctypes.data, 12, Q, switchkey)
        assert np.array_equal(got[b], w), b


def _permute_case(hg, oracle, torch, c, o, primes, n, depth, g):
    Q, Qp, P = c.Q_size, c.Q_prime_size, c.P_size
    l, rc = Q - depth, Qp - depth
    np_ = c.n_power
    batch = 2
    limb_ids = list(range(l)) + list(range(Q, Qp))
    src = [_limbs(oracle, primes, limb_ids * 2, n, 13 + b) for b in range(batch)]
    for s in src:
        s[0] = 0  # q - 0 = q is stored un-reduced by the reference's negation
    in2 = [_limbs(oracle, primes, range(l), n, 60 + b) for b in range(batch)]
    out = torch.empty(batch * 2 * l * n, dtype=torch.int64, device="cuda")
    c.divide_round_lastq_permute(hg.to_device(np.concatenate(src)), 2 * rc * n, hg.to_device(np.concatenate(in2)), l * n,
                                 out, 2 * l * n, g, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    half, half_mod, inv = o.table("half"), o.table("half_mod"), o.table("last_q_modinv")
    for b in range(batch):
        w = np.zeros(2 * l * n, dtype=np.uint64)
        o.L.o_divide_round_lastq_permute(src[b].ctypes.data, in2[b].ctypes.data, w.ctypes.data, o.qp_mods,
                                         half.ctypes.data, half_mod.ctypes.data, inv.ctypes.data, g, np_, rc, l, Qp, Q, P)
        assert np.array_equal(got[b], w), (depth, g, b)


def test_kernel_divide_round_lastq_permute(hg, oracle, torch):
    """hegpu_divide_round_lastq_permute (divide_round_lastq_permute_{bfv,ckks}_kernel, switchkey.cu:1621-1813):
    BFV, CKKS at depth 0 / 2, and two special primes (the method II callers, P_size = 2)."""
    n = 4096
    c, o, primes = _bfv(hg, oracle, n, 1032193)
    for g in (3, 2 * n - 1):
        _permute_case(hg, oracle, torch, c, o, primes, n, 0, g)
    n = 8192
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35, 35], [40], sec=hg.SEC_NONE)
    for depth in (0, 2):
        _permute_case(hg, oracle, torch, c, o, primes, n, depth, hg.steps_to_galois_elt(-2, n, 5))
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35, 35], [40, 40], sec=hg.SEC_NONE)
    for depth in (0, 1):
        _permute_case(hg, oracle, torch, c, o, primes, n, depth, hg.steps_to_galois_elt(1, n, 5))


def _rescale_location(Q, depth):
    counter, location = Q - 1, 0
    for _ in range(depth):
        location += counter
        counter -= 1
    return location


@pytest.mark.parametrize("depth", [0, 2])
def test_kernel_leveled_moddown_stages(hg, oracle, torch, depth):
    """hegpu_divide_round_lastq_leveled_stage_one (relinearize and rescale forms), _stage_two (plain and switchkey),
    hegpu_move_cipher_leveled and hegpu_divide_round_lastq_rescale (switchkey.cu:678-815), each against the oracle's
    restatement with the caller-offset tables of ckks/operator.cu:1003-1015 and :1205-1232."""
    n = 8192
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35, 35], [40], sec=hg.SEC_NONE)
    Q, np_ = c.Q_size, c.n_power
    l = Q - depth
    batch = 2
    dev = lambda parts: hg.to_device(np.concatenate(parts))
    # ---- stage one, relinearize form: in [2][l+1][N] (limbs 0..l-1 and the special prime), out [2][l][N]
    ids = list(range(l)) + [Q]
    src = [_limbs(oracle, primes, ids * 2, n, 5 + b) for b in range(batch)]
    for s_ in src:
        s_[l * n:l * n + 3] = [0, primes[Q] - 1, primes[Q] // 2]
    out = torch.empty(batch * 2 * l * n, dtype=torch.int64, device="cuda")
    c.divide_round_lastq_leveled_stage_one(dev(src), 2 * (l + 1) * n, out, 2 * l * n, 0, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    half, half_mod, inv = o.table("half"), o.table("half_mod"), o.table("last_q_modinv")
    for b in range(batch):
        w = np.zeros(2 * l * n, dtype=np.uint64)
        o.L.o_divide_round_lastq_leveled_stage_one(src[b].ctypes.data, w.ctypes.data, o.qp_mods, half.ctypes.data,
                                                   half_mod.ctypes.data, np_, Q, l)
        assert np.array_equal(got[b], w), ("stage one", depth, b)
    # ---- stage two: (in - last) * P^-1 + ct, both parts / part 0 only
    last = [_limbs(oracle, primes, list(range(l)) * 2, n, 15 + b) for b in range(batch)]
    cts = [synth_ct(primes, range(l), 2, n, 70 + b) for b in range(batch)]
    for sk in (0, 1):
        c.divide_round_lastq_leveled_stage_two(dev(last), 2 * l * n, dev(src), 2 * (l + 1) * n, dev(cts), 2 * l * n, out,
                                               2 * l * n, sk, depth, batch)
        torch.cuda.synchronize()
        got = hg.to_host(out).reshape(batch, -1)
        for b in range(batch):
            w = np.zeros(2 * l * n, dtype=np.uint64)
            o.L.o_divide_round_lastq_leveled_stage_two(last[b].ctypes.data, src[b].ctypes.data, cts[b].ctypes.data,
                                                       w.ctypes.data, o.qp_mods, inv.ctypes.data, np_, l, sk)
            assert np.array_equal(got[b], w), ("stage two", depth, sk, b)
    # ---- rescale: stage one on [2][l][N] -> [2][l-1][N], copy of the kept limbs, divide
    loc = _rescale_location(Q, depth)
    rhalf, rhm, rinv = o.table("rescaled_half"), o.table("rescaled_half_mod"), o.table("rescaled_last_q_modinv")
    ct_in = [synth_ct(primes, range(l), 2, n, 90 + b) for b in range(batch)]
    for s_ in ct_in:
        s_[(l - 1) * n:(l - 1) * n + 3] = [0, primes[l - 1] - 1, primes[l - 1] // 2]
    out1 = torch.empty(batch * 2 * (l - 1) * n, dtype=torch.int64, device="cuda")
    c.divide_round_lastq_leveled_stage_one(dev(ct_in), 2 * l * n, out1, 2 * (l - 1) * n, 1, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out1).reshape(batch, -1)
    for b in range(batch):
        w = np.zeros(2 * (l - 1) * n, dtype=np.uint64)
        o.L.o_divide_round_lastq_leveled_stage_one(ct_in[b].ctypes.data, w.ctypes.data, o.qp_mods,
                                                   rhalf.ctypes.data + 8 * depth, rhm.ctypes.data + 8 * loc, np_, l - 1,
                                                   l - 1)
        assert np.array_equal(got[b], w), ("rescale stage one", depth, b)
    moved = torch.full((batch * 2 * l * n,), 7, dtype=torch.int64, device="cuda")
    c.move_cipher_leveled(dev(ct_in), 2 * l * n, moved, 2 * l * n, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(moved).reshape(batch, -1)
    for b in range(batch):
        w = np.full(2 * l * n, 7, dtype=np.uint64)
        o.L.o_move_cipher_leveled(ct_in[b].ctypes.data, w.ctypes.data, np_, l - 1)
        assert np.array_equal(got[b], w), ("move", depth, b)   # the dropped limb's slot stays untouched
    last = [_limbs(oracle, primes, list(range(l - 1)) * 2, n, 25 + b) for b in range(batch)]
    c.divide_round_lastq_rescale(dev(last), 2 * (l - 1) * n, dev(ct_in), 2 * l * n, out1, 2 * (l - 1) * n, depth, batch)
    torch.cuda.synchronize()
    got = hg.to_host(out1).reshape(batch, -1)
    for b in range(batch):
        w = np.zeros(2 * (l - 1) * n, dtype=np.uint64)
        o.L.o_divide_round_lastq_rescale(last[b].ctypes.data, ct_in[b].ctypes.data, w.ctypes.data, o.qp_mods,
                                         rinv.ctypes.data + 8 * loc, np_, l - 1)
        assert np.array_equal(got[b], w), ("rescale", depth, b)


def test_kernel_divide_round_lastq_extended(hg, oracle, torch):
    """hegpu_divide_round_lastq_extended: divide_round_lastq_extended_kernel (+ ct, BFV method II relinearize),
    _extended_switchkey_kernel (+ ct on part 0) and _extended_leveled_kernel (CKKS method II, depth 0 / 1)
    (switchkey.cu:480-611, 1222-1282) with two and three special primes."""
    n = 8192
    cases = []
    for log_q, log_p in (([40, 35, 35, 35, 35], [40, 40]), ([45, 40, 40, 40, 40, 40], [45, 45, 45])):
        c, o, primes = _ckks(hg, oracle, n, log_q, log_p, sec=hg.SEC_NONE)
        cases += [(c, o, primes, depth, (0, 1, 2)) for depth in (0, 1)]
    cb = hg.Context.from_default(hg.BFV, n, 2, 1032193)   # BFV, two special primes: modes 1 and 2 as its operators use them
    pb = [int(x) for x in cb.table("modulus")]
    ob_ = oracle.OracleContext(oracle.BFV, cb.n_power, pb, cb.Q_size, cb.P_size, 1032193)
    cb.upload()
    cases.append((cb, ob_, pb, 0, (1, 2)))
    for c, o, primes, depth, modes in cases:
        Q, Qp = c.Q_size, c.Q_prime_size
        l, rc = Q - depth, Qp - depth
        batch = 2
        ids = list(range(l)) + list(range(Q, Qp))
        src = [_limbs(oracle, primes, ids * 2, n, 33 + b) for b in range(batch)]
        for s_ in src:
            s_[l * n:l * n + 3] = [0, primes[Q] - 1, primes[Q] // 2]
        cts = [synth_ct(primes, range(l), 2, n, 44 + b) for b in range(batch)]
        out = torch.empty(batch * 2 * l * n, dtype=torch.int64, device="cuda")
        for mode in modes:
            c.divide_round_lastq_extended(hg.to_device(np.concatenate(src)), 2 * rc * n, hg.to_device(np.concatenate(cts)),
                                          2 * l * n, out, 2 * l * n, mode, depth, batch)
            torch.cuda.synchronize()
            got = hg.to_host(out).reshape(batch, -1)
            for b in range(batch):
                w = np.zeros(2 * l * n, dtype=np.uint64)
                o.L.o_divide_round_lastq_extended(o.h, src[b].ctypes.data, cts[b].ctypes.data, w.ctypes.data, rc, l, mode)
                assert np.array_equal(got[b], w), (c.P_size, depth, mode, b)


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_kernel_base_conversion_DtoQtilde(hg, oracle, torch, depth):
    """hegpu_base_conversion_DtoQtilde (base_conversion_DtoQtilde_{bfv,leveled}_kernel, switchkey.cu:872-927,
    985-1046): digits of P_size primes -> Q~ with the float32 overflow estimate; P_size = 2 and 3."""
    n = 8192
    for log_q, log_p in (([40, 35, 35, 35, 35], [40, 40]), ([45, 40, 40, 40, 40, 40, 40], [45, 45, 45])):
        c, o, primes = _ckks(hg, oracle, n, log_q, log_p, sec=hg.SEC_NONE)
        Q, P = len(log_q), len(log_p)
        l, rc = Q - depth, Q + P - depth
        d = -(-l // P)
        batch = 2
        src = [_limbs(oracle, primes, range(l), n, 21 + b) for b in range(batch)]
        for s in src:
            s[:3] = [0, primes[0] - 1, primes[0] // 2]
        out = torch.empty(batch * d * rc * n, dtype=torch.int64, device="cuda")
        c.base_conversion_DtoQtilde(hg.to_device(np.concatenate(src)), l * n, out, d * rc * n, depth, batch)
        torch.cuda.synchronize()
        got = hg.to_host(out).reshape(batch, -1)
        for b in range(batch):
            w = np.zeros(d * rc * n, dtype=np.uint64)
            o.L.o_base_conversion_DtoQtilde(o.h, src[b].ctypes.data, w.ctypes.data, depth)
            assert np.array_equal(got[b], w), (P, depth, b)


def test_kernel_fast_convertion_and_fast_floor(hg, oracle, torch):
    """hegpu_fast_convertion / hegpu_fast_floor (BEHZ, multiplication.cu:10-100, 128-272) on their own,
    N=2^12 default chain (Bsk = 3 primes of 61 bits) and N=2^13 (Q=4)."""
    for n, t in ((4096, 1032193), (8192, 65537)):
        c, o, primes = _bfv(hg, oracle, n, t)
        Q = c.Q_size
        mm = [int(v) for v in c.table("q_Bsk_merge_modulus")]
        L = len(mm)
        batch = 2
        ct1 = [synth_ct(primes, range(Q), 2, n, 1 + b) for b in range(batch)]
        ct2 = [synth_ct(primes, range(Q), 2, n, 9 + b) for b in range(batch)]
        for x in ct1:
            x[:3] = [0, primes[0] - 1, primes[0] // 2]
        out = torch.empty(batch * 4 * L * n, dtype=torch.int64, device="cuda")
        c.fast_convertion(hg.to_device(np.concatenate(ct1)), 2 * Q * n, hg.to_device(np.concatenate(ct2)), 2 * Q * n, out,
                          4 * L * n, batch)
        torch.cuda.synchronize()
        got = hg.to_host(out).reshape(batch, -1)
        for b in range(batch):
            w = np.zeros(4 * L * n, dtype=np.uint64)
            o.L.o_fast_convertion(o.h, ct1[b].ctypes.data, ct2[b].ctypes.data, w.ctypes.data)
            assert np.array_equal(got[b], w), ("fast_convertion", n, b)
        src = [_limbs(oracle, mm, list(range(L)) * 3, n, 33 + b) for b in range(batch)]
        for s in src:
            s[Q * n:Q * n + 2] = [0, mm[Q] - 1]
        fl = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
        c.fast_floor(hg.to_device(np.concatenate(src)), 3 * L * n, fl, 3 * Q * n, batch)
        torch.cuda.synchronize()
        got = hg.to_host(fl).reshape(batch, -1)
        for b in range(batch):
            w = np.zeros(3 * Q * n, dtype=np.uint64)
            o.L.o_fast_floor(o.h, src[b].ctypes.data, w.ctypes.data)
            assert np.array_equal(got[b], w), ("fast_floor", n, b)


# ------------------------------------------------------------------ config C5 at a grid that fills the GPU
def test_c5_tfhe_4096_gates(hg, oracle, torch):
    """Config C5 per-GPU share and more: 4096 concurrent NAND gates (pre-computation -> blind rotate ->
    sample extraction -> key switching) with a real torus32 boot key (FP64 blind rotate).  64 distinct input pairs;
    16 gates are compared with the oracle bit for bit, the rest through twin consistency (inputs repeat with
    period `uniq`, so gate i must equal gate i mod uniq)."""
    t = hg.TfheContext()
    o = oracle.OracleTfhe()
    rng = np.random.default_rng(5)
    polys = t.int("bootkey_elems") // 1024
    coeff = rng.integers(-2**31, 2**31, (polys, 1024), dtype=np.int64).astype(np.int32)
    bk = np.concatenate([o.to_ntt(coeff[i]) for i in range(polys)])
    prepared = t.prepare_bootkey(hg.to_device(bk))
    assert t.prepared_is_fp64(prepared)
    ks_a = rng.integers(-2**31, 2**31, t.int("kskey_a_elems"), dtype=np.int64).astype(np.int32)
    ks_b = rng.integers(-2**31, 2**31, t.int("kskey_b_elems"), dtype=np.int64).astype(np.int32)
    shape, uniq, checked = 4096, 64, 16
    r32 = lambda k: rng.integers(-2**31, 2**31, k, dtype=np.int64).astype(np.int32)
    a1u, a2u, b1u, b2u = r32(uniq * 512), r32(uniq * 512), r32(uniq), r32(uniq)
    rep = shape // uniq
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a1, a2 = dev(np.tile(a1u, rep)), dev(np.tile(a2u, rep))
    b1, b2 = dev(np.tile(b1u, rep)), dev(np.tile(b2u, rep))
    out_a = torch.empty(shape * 512, dtype=torch.int32, device="cuda")
    out_b = torch.empty(shape, dtype=torch.int32, device="cuda")
    ws = torch.empty((512 + 1024 + 2) * shape, dtype=torch.int32, device="cuda")
    t.gate(hg.GATE_NAND, a1, b1, a2, b2, out_a, out_b, prepared, dev(ks_a), dev(ks_b), shape, ws)
    torch.cuda.synchronize()
    ga, gb = out_a.cpu().numpy().reshape(shape, 512), out_b.cpu().numpy()
    want_a, want_b = o.gate(hg.GATE_NAND, a1u[:checked * 512], b1u[:checked], a2u[:checked * 512], b2u[:checked], bk,
                            ks_a, ks_b)
    assert np.array_equal(ga[:checked].reshape(-1), want_a) and np.array_equal(gb[:checked], want_b)
    assert np.array_equal(ga, np.tile(ga[:uniq], (rep, 1))), "a gate differs from its twin"
    assert np.array_equal(gb, np.tile(gb[:uniq], rep))


@pytest.mark.parametrize("n,fused", [(4096, 0), (8192, 0), (16384, 0), (32768, 1), (32768, 0), (65536, 1)])
def test_bfv_multiply_tensor_fusion(hg, oracle, torch, n, fused):
    """multiply_bfv on the default chains with the tensor product as the load transform of the inverse transform
    (option fused_tensor = 1, the default: single pass at N <= 2^14 -- covered by every other BFV test -- and the
    two-pass row kernel ntt_inv_row_tensor at 2^15 / 2^16) and as the reference's kernel of its own (= 0)."""
    t = 786433
    with backend_switches(HEGPU_FUSED_TENSOR=fused):
        c, o, primes = _bfv(hg, oracle, n, t)
    Q, batch = c.Q_size, 2
    ct1 = [synth_ct(primes, range(Q), 2, n, 5 + b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 9 + b) for b in range(batch)]
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_multiply(hg.to_device(np.concatenate(ct1)), 2 * Q * n, hg.to_device(np.concatenate(ct2)), 2 * Q * n, out, 3 * Q * n,
                   batch, c.workspace(hg.OP_BFV_MULTIPLY, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(got[b], o.bfv_multiply(ct1[b], ct2[b])), (n, fused, b)


def test_bfv_n14_multiply_at_the_bench_shape(hg, oracle, torch):
    """North-star target 2 at the shape bench.py times: BFV N=2^14, default chain (Q=8, Bsk=9), 256 pairs in one call
    (single-pass transforms of 4 x 17 x 256 limbs, the one-thread-per-coefficient BEHZ kernels).  Four distinct
    pairs against the oracle, the other 252 against their twins."""
    n, t, B, U = 1 << 14, 786433, 256, 4
    c, o, primes = _bfv(hg, oracle, n, t)
    Q = c.Q_size
    a = [synth_ct(primes, range(Q), 2, n, 1 + 10 * u) for u in range(U)]
    b = [synth_ct(primes, range(Q), 2, n, 2 + 10 * u) for u in range(U)]
    d1 = hg.to_device(np.concatenate(a)).repeat(B // U)
    d2 = hg.to_device(np.concatenate(b)).repeat(B // U)
    out = torch.empty(B * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.bfv_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, B, c.workspace(hg.OP_BFV_MULTIPLY, 0, B))
    torch.cuda.synchronize()
    v = out.view(B // U, U, 3 * Q * n)
    assert bool(torch.equal(v, v[:1].expand(B // U, U, 3 * Q * n))), "an item differs from its twin"
    got = hg.to_host(v[0])
    for u in range(U):
        assert np.array_equal(got[u], o.bfv_multiply(a[u], b[u])), u


def test_bench_line_proves_its_own_work():
    """`bench.py --steps 1 --warmup 0 --no-secondary` on this GPU: rc 0 and a line whose every timed output was
    checked -- all 64 against the CPU oracle, 60 of them also against their twins."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = os.path.join(root, "gpurun_out", "bench_detail_test.json")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-secondary",
                        "--detail", detail], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # as the driver reads it: the last 8000 characters of stdout, the last line that starts with `{`
    text = [ln for ln in r.stdout[-8000:].splitlines() if ln.startswith("{")][-1]
    assert len(text) < 4096 and r.stdout.rstrip().endswith(text), len(text)
    line = json.loads(text)
    full = json.load(open(detail))
    assert full["value"] == pytest.approx(line["value"], rel=1e-4) and "in_step" in full and "in_step" not in line
    assert line["distinct_devices"] == 1 and line["ranks"] == 1
    assert line["cpu_baseline"]["gpu_matches_cpu_bit_exact"] and line["cpu_baseline"]["kind"] == "port"
    chk = line["checked_items"]
    assert chk["oracle_equal"] and chk["twins_equal"] and chk["total"] == 64 and chk["oracle_compared"] >= 2
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["roofline"]["unit"] == "GB/s"


@pytest.mark.parametrize("split", [0, 1], ids=["one_thread_per_coefficient", "rows_over_four_wavefronts"])
@pytest.mark.parametrize("Q,bits", [(42, 30), (58, 60)], ids=["42x30bit", "58x60bit"])
def test_bfv_multiply_with_more_than_40_base_primes(hg, oracle, torch, split, Q, bits):
    """BEHZ with Q + |Bsk| beyond 40 moduli (the reference allows MAX_BSK_SIZE = 64, defines.h:26):
    BFV N=2^12, 42 primes of 30 bits + one special prime, multiply against the oracle; and 58 primes of 60 bits with 59
    base primes of 61 bits -- lazy row sums of 59 terms of up to 121 bits whose high word passes 2^63 (the
    Montgomery reduction of round 3 assumed hi < 2^63; redc128 takes any 128-bit sum now, and the context refuses a base
    whose worst-case sum would not fit 128 bits: tests/test_cabi.py)."""
    n, t = 4096, 65537
    c = hg.Context.from_bit_sizes(hg.BFV, n, [bits] * Q, [bits + 1 if bits < 60 else 60], plain_modulus=t, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, 12, primes, Q, 1, t)
    c.upload()
    L = len(c.table("q_Bsk_merge_modulus"))
    assert L - Q > 40, (L, Q)
    ct1 = synth_ct(primes, range(Q), 2, n, 1)
    ct2 = synth_ct(primes, range(Q), 2, n, 2)
    ct1[:4] = [primes[0] - 1, 0, primes[0] - 1, 1]  # extreme residues among the random ones
    out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    c.set_option("behz_split", split)
    c.bfv_multiply(hg.to_device(ct1), 2 * Q * n, hg.to_device(ct2), 2 * Q * n, out, 3 * Q * n, 1,
                   c.workspace(hg.OP_BFV_MULTIPLY, 0, 1))
    torch.cuda.synchronize()
    want = o.bfv_multiply(ct1, ct2)
    assert np.array_equal(hg.to_host(out), want)


@pytest.mark.parametrize("split", [0, 1], ids=["one_thread_per_coefficient", "rows_over_four_wavefronts"])
@pytest.mark.parametrize("Q", [17, 22, 26, 30])
def test_bfv_multiply_base_sizes_between_the_kernel_instances(hg, oracle, torch, Q, split):
    """The BEHZ kernels exist for padded base sizes (steps of 4 between 16 and 32; the plain fast_floor sends 25..28
    to the 32-slot instance): bases of 17, 22, 26 and 30 primes of 30 bits, N = 2^12, both forms."""
    n, t = 4096, 65537
    c = hg.Context.from_bit_sizes(hg.BFV, n, [30] * Q, [31], plain_modulus=t, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, 12, primes, Q, 1, t)
    c.upload()
    ct1, ct2 = synth_ct(primes, range(Q), 2, n, 3), synth_ct(primes, range(Q), 2, n, 4)
    out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
    c.set_option("behz_split", split)
    c.bfv_multiply(hg.to_device(ct1), 2 * Q * n, hg.to_device(ct2), 2 * Q * n, out, 3 * Q * n, 1,
                   c.workspace(hg.OP_BFV_MULTIPLY, 0, 1))
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(out), o.bfv_multiply(ct1, ct2))


def test_six_gpuntt_entry_points_by_name(hg, oracle, torch):
    """hegpu_GPU_NTT / _Inplace / GPU_INTT / _Inplace / GPU_NTT_Modulus_Ordered_Inplace /
    GPU_NTT_Poly_Ordered_Inplace (include/hegpu.h: the names a maintainer binds call site by call site),
    each against the oracle's restatement of the same gpuntt call, CKKS N=2^13 {40,35x3}|{40}, depth 1."""
    from heongpu_amd import _lib
    L = _lib.load()
    n, Q, Qp, depth = 8192, 4, 5, 1
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35], [40])
    h = c._h
    st = torch.cuda.current_stream().cuda_stream
    rc = Qp - depth
    batch = 2 * Qp
    x = np.concatenate([oracle.fill_poly(7 + i, i % Qp, n, primes[i % Qp]) for i in range(batch)])
    want = o.ntt(x.copy(), batch, Qp)
    d, out = hg.to_device(x), torch.empty(batch * n, dtype=torch.int64, device="cuda")
    assert L.hegpu_GPU_NTT(h, hg.TABLES_QP, d.data_ptr(), out.data_ptr(), 0, batch, Qp, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(out), want) and np.array_equal(hg.to_host(d), x)
    assert L.hegpu_GPU_INTT(h, hg.TABLES_QP, out.data_ptr(), d.data_ptr(), 0, batch, Qp, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(d), x)
    assert L.hegpu_GPU_NTT_Inplace(h, hg.TABLES_QP, d.data_ptr(), 0, batch, Qp, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(d), want)
    assert L.hegpu_GPU_INTT_Inplace(h, hg.TABLES_QP, d.data_ptr(), 0, batch, Qp, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(d), x)
    # caller-offset tables: the two P-limb polynomials only (ckks/operator.cu:996 passes tables + Q)
    y = np.concatenate([oracle.fill_poly(50 + i, Q, n, primes[Q]) for i in range(2)])
    dy = hg.to_device(y)
    assert L.hegpu_GPU_NTT_Inplace(h, hg.TABLES_QP, dy.data_ptr(), Q, 2, 1, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(dy), o.ntt(y.copy(), 2, 1, mod_offset=Q))
    # modulus ordered, forward and inverse (ckks/operator.cu:956,1524)
    order = [int(v) for v in o.table("new_prime_locations")][Qp:Qp + rc]
    z = np.concatenate([oracle.fill_poly(11 + i, 0, n, primes[order[i % rc]]) for i in range(3 * rc)])
    wz = z.copy()
    ord_arr = np.array(order, dtype=np.int32)
    tab, itab, ninv = o.table("ntt_table"), o.table("intt_table"), o.table("n_inverse")
    o.L.o_gpu_ntt_modulus_ordered(wz.ctypes.data, tab.ctypes.data, o.qp_mods, ninv.ctypes.data, 0, 13, 3 * rc, rc, ord_arr.ctypes.data)
    dz = hg.to_device(z)
    dev_order = c.device_ptr("new_prime_locations") + 4 * Qp
    assert L.hegpu_GPU_NTT_Modulus_Ordered_Inplace(h, hg.TABLES_QP, dz.data_ptr(), 0, 0, 3 * rc, rc, dev_order, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(dz), wz)
    assert L.hegpu_GPU_NTT_Modulus_Ordered_Inplace(h, hg.TABLES_QP, dz.data_ptr(), 1, 0, 3 * rc, rc, dev_order, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(dz), z)
    # poly ordered: INTT of slots {rc-1, 2rc-1} with the P prime (relinearize, ckks/operator.cu:996)
    slots = np.array([rc - 1, 2 * rc - 1], dtype=np.int32)
    w = np.concatenate([oracle.fill_poly(90 + i, 0, n, primes[Qp - 1]) for i in range(2 * rc)])
    ww = w.copy()
    o.L.o_gpu_ntt_poly_ordered(ww.ctypes.data, itab.ctypes.data + Q * n * 8, o.mods_addr(Q), ninv.ctypes.data + Q * 8, 1, 13, 2, 1,
                               slots.ctypes.data)
    dw = hg.to_device(w)
    dev_slots = c.device_ptr("new_input_locations") + 4 * 2 * depth
    assert L.hegpu_GPU_NTT_Poly_Ordered_Inplace(h, hg.TABLES_QP, dw.data_ptr(), 1, Q, 2, 1, dev_slots, st) == 0
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(dw), ww)
    assert L.hegpu_GPU_NTT_Poly_Ordered_Inplace(h, hg.TABLES_QP, dw.data_ptr(), 1, Q, 2, 1, None, st) != 0


# ------------------------------------------------------------------ hoisted rotations
@pytest.mark.parametrize("depth", [0, 2])
@pytest.mark.parametrize("method", ["I", "II"])
@pytest.mark.parametrize("grouped", [False, True], ids=["one_accumulator", "four_accumulators"])
def test_hoisted_rotations(hg, oracle, torch, depth, method, grouped):
    """hegpu_ckks_rotate_hoisted (fast_single_hoisting_rotation_ckks_method_I / _II, ckks/operator.cu:4674-5446):
    one decomposition + digit NTT shared by several Galois elements; every entry bit-identical to the oracle's
    restatement of the reference loop AND to separate hegpu_ckks_apply_galois calls.  CKKS N=2^13, batch 2,
    elements for shifts (0, 1, -2, 5) + conjugation, one key per element."""
    n = 8192
    log_p = [40] if method == "I" else [40, 40]
    c, o, primes = _ckks(hg, oracle, n, [40, 35, 35, 35, 35], log_p, sec=hg.SEC_NONE)
    Q, Qp = 5, 5 + len(log_p)
    l = Q - depth
    digits = Q if method == "I" else -(-Q // 2)
    batch = 2
    elts = [0, hg.steps_to_galois_elt(1, n, 5), hg.steps_to_galois_elt(-2, n, 5), hg.steps_to_galois_elt(5, n, 5), 2 * n - 1,
            hg.steps_to_galois_elt(3, n, 5), 0, hg.steps_to_galois_elt(-1, n, 5)]  # six keys: a group of four and one of two
    keys = [None if g == 0 else synth_key(primes, digits, Qp, n, 20 + i) for i, g in enumerate(elts)]
    dkeys = [None if k is None else hg.to_device(k) for k in keys]
    cts = [synth_ct(primes, range(l), 2, n, 5 + b) for b in range(batch)]
    d = hg.to_device(np.concatenate(cts))
    words = 2 * l * n
    out = torch.empty(batch * len(elts) * words, dtype=torch.int64, device="cuda")
    # the larger workspace holds four accumulators: four inner products per read of the digits
    ws = c.workspace(hg.OP_CKKS_ROTATE_HOISTED if grouped else hg.OP_CKKS_GALOIS, depth, batch)
    c.ckks_rotate_hoisted(d, words, out, len(elts) * words, dkeys, elts, depth, batch, ws)
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, len(elts), words)
    one = torch.empty(batch * words, dtype=torch.int64, device="cuda")
    for b in range(batch):
        want = o.ckks_rotate_hoisted(cts[b], keys, elts, depth).reshape(len(elts), words)
        for i in range(len(elts)):
            assert np.array_equal(got[b, i], want[i]), (method, depth, b, i)
    for i in range(1, len(elts)):  # the same elements one by one through the plain operator
        if elts[i] == 0:
            continue
        c.ckks_apply_galois(d, words, one, words, dkeys[i], elts[i], depth, batch, ws)
        torch.cuda.synchronize()
        assert np.array_equal(hg.to_host(one).reshape(batch, words), got[:, i]), ("vs apply_galois", i)


@pytest.mark.parametrize("sw", [dict(), dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()) or "default")
@pytest.mark.parametrize("n_power,log_p", [(12, [60, 50]), (14, [60, 50, 50]), (16, [60, 50])])
def test_method_II_and_hoisting_at_other_degrees(hg, oracle, torch, sw, n_power, log_p):
    """Key switching method II (relinearize, rotate, hoisted rotations) away from N = 2^13: the digit -> Q~ base
    conversion, the multi-prime mod-down in the NTT domain and the key-stationary inner product at N = 2^12, 2^14
    and 2^16, by launch size and with the fused forms forced; depth 1, a 60-bit prime in Q (and in P at 2^14)."""
    n = 1 << n_power
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [60, 45, 45, 45, 45, 45], log_p, sec=hg.SEC_NONE)
    Q, P, depth = 6, len(log_p), 1
    Qp, l, digits = Q + P, Q - depth, -(-Q // P)
    batch = 2
    key = synth_key(primes, digits, Qp, n, 3)
    ct1 = [synth_ct(primes, range(l), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(l), 2, n, 2 + 10 * b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * l * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, 2 * l * n, d2, 2 * l * n, out, 3 * l * n, depth, batch)
    c.ckks_relinearize_inplace(out, 3 * l * n, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    for b in range(batch):
        w = o.ckks_multiply(ct1[b], ct2[b], depth)
        o.ckks_relinearize_II(w, key, depth)
        assert np.array_equal(got[b][:2 * l * n], w[:2 * l * n]), "method II relinearize"
    elts = [hg.steps_to_galois_elt(1, n, 5), 0, hg.steps_to_galois_elt(-3, n, 5), 2 * n - 1, hg.steps_to_galois_elt(7, n, 5)]
    keys = [None if g == 0 else synth_key(primes, digits, Qp, n, 30 + i) for i, g in enumerate(elts)]
    dkeys = [None if k is None else hg.to_device(k) for k in keys]
    words = 2 * l * n
    hout = torch.empty(batch * len(elts) * words, dtype=torch.int64, device="cuda")
    c.ckks_rotate_hoisted(d1, words, hout, len(elts) * words, dkeys, elts, depth, batch,
                          c.workspace(hg.OP_CKKS_ROTATE_HOISTED, depth, batch))
    one = torch.empty(batch * words, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(d1, words, one, words, dkeys[0], elts[0], depth, batch, c.workspace(hg.OP_CKKS_GALOIS, depth, batch))
    torch.cuda.synchronize()
    gh = hg.to_host(hout).reshape(batch, len(elts), words)
    go = hg.to_host(one).reshape(batch, words)
    for b in range(batch):
        want = o.ckks_rotate_hoisted(ct1[b], keys, elts, depth).reshape(len(elts), words)
        for i in range(len(elts)):
            assert np.array_equal(gh[b, i], want[i]), ("hoisted", b, i)
        assert np.array_equal(go[b], o.ckks_apply_galois_II(ct1[b], keys[0], elts[0], depth)), "method II rotate"


@pytest.mark.parametrize("single", [1, 0], ids=["single_pass", "two_pass"])
@pytest.mark.parametrize("n_power", [12, 13, 14])
def test_ntt_small_degrees_both_forms(hg, oracle, torch, n_power, single):
    """N <= 2^14: the LDS-resident single pass (ntt_fwd_single) and the two passes give the oracle's transform,
    FP64 moduli (30 / 45 / 50 bits), lazy integer (55) and full-range integer (60 / 61-bit Bsk) ones, batches
    that wrap around the modulus list, plain and modulus-ordered."""
    n = 1 << n_power
    with backend_switches(HEGPU_SINGLE_PASS=single):
        c, o, primes = _ckks(hg, oracle, n, [60, 30, 45, 50, 55], [60], sec=hg.SEC_NONE)
    Qp = 6
    batch = 3 * Qp + 2
    x = np.concatenate([oracle.fill_poly(7 + i, i % Qp, n, primes[i % Qp]) for i in range(batch)])
    x[:4] = [0, primes[0] - 1, 1, primes[0] // 2]
    want = o.ntt(x.copy(), batch, Qp)
    d = hg.to_device(x)
    out = torch.empty_like(d)
    c.ntt(d, out, False, batch, Qp)
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(out), want)
    c.ntt(out, out, True, batch, Qp)
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(out), x)
    # extremes on an FP64 modulus: all q-1 and alternating 0 / q-1
    q = primes[3]
    y = np.concatenate([np.full(n, q - 1, dtype=np.uint64), np.tile(np.array([0, q - 1], dtype=np.uint64), n // 2)])
    wy = o.ntt(y.copy(), 2, 1, mod_offset=3)
    dy = hg.to_device(y)
    c.ntt(dy, dy, False, 2, 1, mod_offset=3)
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(dy), wy)


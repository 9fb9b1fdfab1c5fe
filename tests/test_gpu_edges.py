"""Edge cases of the operator entries on the GPU: empty batches, the deepest level (one limb left), a one-prime
chain, the smallest and the largest ring, the smallest prime size the reference accepts (30 bits) next to the
largest (60), a batch beyond one launch's grid -- each compared limb for limb with the oracle."""
import numpy as np
import pytest

from helpers import backend_switches, synth_ct, synth_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _ckks(hg, oracle, n, log_q, log_p):
    c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, c.n_power, primes, len(log_q), len(log_p))
    c.upload()
    return c, o, primes


def test_empty_batch_is_a_no_op(hg, oracle, torch):
    """batch == 0: every operator entry returns success and touches nothing."""
    n = 4096
    c, o, primes = _ckks(hg, oracle, n, [40, 30, 30], [40])
    Q, Qp = 3, 4
    sentinel = 0x5A5A5A5A5A5A5A5A
    out = torch.full((3 * Q * n,), sentinel, dtype=torch.int64, device="cuda")
    src = hg.to_device(synth_ct(primes, range(Q), 3, n, 1))
    key = hg.to_device(synth_key(primes, Q, Qp, n, 2))
    ws = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
    c.ckks_multiply(src, 2 * Q * n, src, 2 * Q * n, out, 3 * Q * n, 0, 0)
    c.ckks_relinearize_inplace(out, 3 * Q * n, key, 0, 0, ws)
    c.ckks_rescale_inplace(out, 3 * Q * n, 0, 0, ws)
    c.ckks_apply_galois(src, 2 * Q * n, out, 2 * Q * n, key, 3, 0, 0, ws)
    c.ntt(out, out, False, 0, Q)
    torch.cuda.synchronize()
    assert bool((out == sentinel).all())


@pytest.mark.parametrize("n", [4096, 65536])
def test_deepest_level_and_one_prime_chain(hg, oracle, torch, n):
    """l = 1 (depth Q - 1): relinearize and rotate on the last remaining limb; rescale from two limbs to one;
    and a chain that only ever has one prime (Q = 1, P = 1).  Smallest and largest ring."""
    c, o, primes = _ckks(hg, oracle, n, [45, 30, 36], [45])
    Q, Qp = 3, 4
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 4)
    g = hg.steps_to_galois_elt(1, n, 5)
    for depth in (2, 1):
        l = Q - depth
        a, b = synth_ct(primes, range(l), 2, n, 10 + depth), synth_ct(primes, range(l), 2, n, 20 + depth)
        out = torch.empty(3 * l * n, dtype=torch.int64, device="cuda")
        c.ckks_multiply(hg.to_device(a), 2 * l * n, hg.to_device(b), 2 * l * n, out, 3 * l * n, depth, 1)
        c.ckks_relinearize_inplace(out, 3 * l * n, hg.to_device(key), depth, 1, c.workspace(hg.OP_CKKS_RELIN, depth, 1))
        torch.cuda.synchronize()
        want = o.ckks_relinearize(o.ckks_multiply(a, b, depth), key, depth)
        got = hg.to_host(out)
        assert np.array_equal(got[:2 * l * n], want[:2 * l * n]), ("relinearize", depth)
        rot = torch.empty(2 * l * n, dtype=torch.int64, device="cuda")
        c.ckks_apply_galois(hg.to_device(a), 2 * l * n, rot, 2 * l * n, hg.to_device(gkey), g, depth, 1,
                            c.workspace(hg.OP_CKKS_GALOIS, depth, 1))
        torch.cuda.synchronize()
        assert np.array_equal(hg.to_host(rot), o.ckks_apply_galois(a, gkey, g, depth)), ("rotate", depth)
        if l == 2:
            c.ckks_rescale_inplace(out, 3 * l * n, depth, 1, c.workspace(hg.OP_CKKS_RESCALE, depth, 1))
            torch.cuda.synchronize()
            w = o.ckks_rescale(want[:2 * l * n].copy(), depth)
            assert np.array_equal(hg.to_host(out)[:2 * n], w[:2 * n]), "rescale to one limb"
    c.close()
    c, o, primes = _ckks(hg, oracle, n, [50], [51])
    key = synth_key(primes, 1, 2, n, 5)
    a, b = synth_ct(primes, range(1), 2, n, 6), synth_ct(primes, range(1), 2, n, 7)
    out = torch.empty(3 * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(hg.to_device(a), 2 * n, hg.to_device(b), 2 * n, out, 3 * n, 0, 1)
    c.ckks_relinearize_inplace(out, 3 * n, hg.to_device(key), 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
    torch.cuda.synchronize()
    want = o.ckks_relinearize(o.ckks_multiply(a, b, 0), key, 0)
    assert np.array_equal(hg.to_host(out)[:2 * n], want[:2 * n]), "Q = 1"


def test_smallest_and_largest_prime_sizes_together(hg, oracle, torch):
    """30-bit primes (the reference's minimum, ckks/context.cu:100-110) next to 60-bit ones: narrow FP64 targets
    fed by wide digits and the other way round, through multiply -> relinearize -> rescale -> rotate."""
    n = 8192
    bits = [60, 30, 30, 31, 60, 30]
    c, o, primes = _ckks(hg, oracle, n, bits, [60])
    Q, Qp = len(bits), len(bits) + 1
    key, gkey = synth_key(primes, Q, Qp, n, 3), synth_key(primes, Q, Qp, n, 4)
    batch = 2
    a = [synth_ct(primes, range(Q), 2, n, 30 + i) for i in range(batch)]
    b = [synth_ct(primes, range(Q), 2, n, 40 + i) for i in range(batch)]
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(hg.to_device(np.concatenate(a)), 2 * Q * n, hg.to_device(np.concatenate(b)), 2 * Q * n, out, 3 * Q * n, 0, batch)
    c.ckks_relinearize_inplace(out, 3 * Q * n, hg.to_device(key), 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    want = [o.ckks_relinearize(o.ckks_multiply(a[i], b[i], 0), key, 0) for i in range(batch)]
    for i in range(batch):
        assert np.array_equal(got[i][:2 * Q * n], want[i][:2 * Q * n]), ("relinearize", i)
    c.ckks_rescale_inplace(out, 3 * Q * n, 0, batch, c.workspace(hg.OP_CKKS_RESCALE, 0, batch))
    g = hg.steps_to_galois_elt(-3, n, 5)
    rot = torch.empty(batch * 2 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(hg.to_device(np.concatenate(a)), 2 * Q * n, rot, 2 * Q * n, hg.to_device(gkey), g, 0, batch,
                        c.workspace(hg.OP_CKKS_GALOIS, 0, batch))
    torch.cuda.synchronize()
    got, gr = hg.to_host(out).reshape(batch, -1), hg.to_host(rot).reshape(batch, -1)
    for i in range(batch):
        w = o.ckks_rescale(want[i][:2 * Q * n].copy(), 0)
        assert np.array_equal(got[i][:2 * (Q - 1) * n], w[:2 * (Q - 1) * n]), ("rescale", i)
        assert np.array_equal(gr[i], o.ckks_apply_galois(a[i], gkey, g, 0)), ("rotate", i)


def test_batch_beyond_one_grid(hg, oracle, torch):
    """More polynomials than one launch's grid holds (65535): a plain NTT of 70 000 limbs at N = 2^12 is cut into
    pieces on modulus-cycle boundaries; sampled limbs against the oracle, the round trip for all of them."""
    n, limbs = 4096, 70000
    c, o, primes = _ckks(hg, oracle, n, [40, 30, 30, 30], [40])
    mc = 5
    assert limbs % mc == 0
    rng = np.random.default_rng(11)
    x = np.empty(limbs * n, dtype=np.uint64)
    v = x.reshape(limbs, n)
    for j in range(mc):
        v[j::mc] = rng.integers(0, primes[j], (limbs // mc, n), dtype=np.uint64)
    d = hg.to_device(x)
    y = torch.empty_like(d)
    c.ntt(d, y, False, limbs, mc)
    torch.cuda.synchronize()
    got = hg.to_host(y).reshape(limbs, n)
    for i in (0, 1, 4, 32767, 65534, 65535, 65536, 69999):
        assert np.array_equal(got[i], o.ntt(v[i].copy(), 1, 1, mod_offset=i % mc)), i
    c.ntt(y, y, True, limbs, mc)
    torch.cuda.synchronize()
    assert np.array_equal(hg.to_host(y), x)


@pytest.mark.parametrize("sw", [dict(), dict(HEGPU_COL_MULTI=0)], ids=["default", "per_polynomial_column_pass"])
@pytest.mark.parametrize("log_q,log_p", [([40, 30, 30], [40]), ([40, 30, 30, 30], [40, 40])], ids=["method_I", "method_II"])
def test_epilogue_launches_beyond_one_grid(hg, oracle, torch, log_q, log_p, sw):
    """Relinearize and rescale of so many ciphertexts that their mod-down transforms (2 l resp. 2 (l - 1)
    polynomials per ciphertext, with per-ciphertext epilogue operands) exceed one grid (65535 polynomials) and are
    cut into pieces: every per-item pointer of the launch has to move with the piece.  The batch repeats 24
    distinct ciphertexts, so all of them are checked against the oracle and every other item against its twin."""
    n = 4096
    with backend_switches(**sw):   # per-polynomial column pass: the rescale's copy rides on it (NttArgs::copy_src)
        c, o, primes = _ckks(hg, oracle, n, log_q, log_p)
    Q, P = len(log_q), len(log_p)
    Qp = Q + P
    batch, distinct = 16400, 24            # relinearize: 2 Q * 16400, rescale: 2 (Q - 1) * 16400 >= 65600 polynomials
    key = synth_key(primes, -(-Q // P), Qp, n, 3)
    relin = o.ckks_relinearize if P == 1 else o.ckks_relinearize_II
    cts = [synth_ct(primes, range(Q), 3, n, 40 + i) for i in range(distinct)]
    base = hg.to_device(np.concatenate(cts)).reshape(distinct, 3 * Q * n)
    d = base.repeat((batch + distinct - 1) // distinct, 1)[:batch].contiguous().reshape(-1)
    c.ckks_relinearize_inplace(d, 3 * Q * n, hg.to_device(key), 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = d.reshape(batch, 3 * Q * n)[:, :2 * Q * n]
    want = []
    for i in range(distinct):
        w = cts[i].copy()
        relin(w, key, 0)
        want.append(w[:2 * Q * n])
        assert np.array_equal(hg.to_host(got[i]), want[i]), ("relinearize", i)
    twin = got[:distinct].repeat((batch + distinct - 1) // distinct, 1)[:batch]
    assert bool((got == twin).all()), "relinearize: an item beyond the first piece differs from its twin"
    c.ckks_rescale_inplace(d, 3 * Q * n, 0, batch, c.workspace(hg.OP_CKKS_RESCALE, 0, batch))
    torch.cuda.synchronize()
    got = d.reshape(batch, 3 * Q * n)[:, :2 * (Q - 1) * n]
    for i in range(distinct):
        assert np.array_equal(hg.to_host(got[i]), o.ckks_rescale(want[i].copy(), 0)[:2 * (Q - 1) * n]), ("rescale", i)
    twin = got[:distinct].repeat((batch + distinct - 1) // distinct, 1)[:batch]
    assert bool((got == twin).all()), "rescale: an item beyond the first piece differs from its twin"


def test_bfv_epilogue_launch_beyond_one_grid(hg, oracle, torch):
    """The same for BFV: the inverse transform that carries the mod-down as its epilogue (NttInvEpilogue, 2 (Q + 1)
    polynomials per ciphertext with per-ciphertext operands) in relinearize and rotate of 11 000 ciphertexts of
    N = 4096 (66 000 polynomials: two pieces)."""
    n, t = 4096, 1032193
    c = hg.Context.from_default(hg.BFV, n, 1, t)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, c.Q_size, c.P_size, t)
    c.upload()
    Q, Qp = c.Q_size, c.Q_prime_size
    batch, distinct = 11000, 16
    assert 2 * (Q + 1) * batch > 65535
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 4)
    g = hg.steps_to_galois_elt(1, n, 3)
    cts = [synth_ct(primes, range(Q), 3, n, 60 + i) for i in range(distinct)]
    base = hg.to_device(np.concatenate(cts)).reshape(distinct, 3 * Q * n)
    reps = (batch + distinct - 1) // distinct
    d = base.repeat(reps, 1)[:batch].contiguous().reshape(-1)
    c.bfv_relinearize_inplace(d, 3 * Q * n, hg.to_device(key), batch, c.workspace(hg.OP_BFV_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = d.reshape(batch, 3 * Q * n)[:, :2 * Q * n]
    want = []
    for i in range(distinct):
        w = o.bfv_relinearize(cts[i].copy(), key)
        want.append(np.ascontiguousarray(w[:2 * Q * n]))
        assert np.array_equal(hg.to_host(got[i]), want[i]), ("relinearize", i)
    assert bool((got == got[:distinct].repeat(reps, 1)[:batch]).all()), "relinearize: an item of the second piece differs"
    src = got.contiguous().reshape(-1)
    rot = torch.empty_like(src)
    c.bfv_apply_galois(src, 2 * Q * n, rot, 2 * Q * n, hg.to_device(gkey), g, batch, c.workspace(hg.OP_BFV_GALOIS, 0, batch))
    torch.cuda.synchronize()
    gr = rot.reshape(batch, 2 * Q * n)
    for i in range(distinct):
        assert np.array_equal(hg.to_host(gr[i]), o.bfv_apply_galois(want[i], gkey, g)), ("rotate", i)
    assert bool((gr == gr[:distinct].repeat(reps, 1)[:batch]).all()), "rotate: an item of the second piece differs"


@pytest.mark.parametrize("batch", [3, 40])
def test_padded_item_strides(hg, oracle, torch, batch):
    """Ciphertexts that are NOT back to back: every operator takes element strides between the items of a batch.
    Items 1000 (inputs) / 3000 (outputs) elements further apart than their size, the gaps filled with a sentinel
    that must survive; results against the oracle.  Both launch-size regimes of the key switch (3 and 40 items)."""
    n = 8192
    c, o, primes = _ckks(hg, oracle, n, [50, 40, 40, 40], [50])
    Q, Qp = 4, 5
    SENT = 0x5555555555555555
    key = synth_key(primes, Q, Qp, n, 3)
    gkey = synth_key(primes, Q, Qp, n, 4)
    g = hg.steps_to_galois_elt(1, n, 5)
    dkey, dgkey = hg.to_device(key), hg.to_device(gkey)
    ct1 = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(batch)]
    s_in, s_out = 2 * Q * n + 1000, 3 * Q * n + 3000
    def strided(items, stride):
        buf = torch.full((batch * stride,), SENT, dtype=torch.int64, device="cuda")
        for b, x in enumerate(items):
            buf[b * stride:b * stride + len(x)] = hg.to_device(x)
        return buf
    def gaps_intact(buf, stride, used):
        v = buf.reshape(batch, stride)[:, used:]
        return bool((v == SENT).all())
    d1, d2 = strided(ct1, s_in), strided(ct2, s_in)
    out = torch.full((batch * s_out,), SENT, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, s_in, d2, s_in, out, s_out, 0, batch)
    c.ckks_relinearize_inplace(out, s_out, dkey, 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, s_out)
    want = []
    for b in range(batch):
        w = o.ckks_multiply(ct1[b], ct2[b], 0)
        o.ckks_relinearize(w, key, 0)
        want.append(w)
        assert np.array_equal(got[b][:2 * Q * n], w[:2 * Q * n]), ("relinearize", b)
    assert gaps_intact(out, s_out, 3 * Q * n) and gaps_intact(d1, s_in, 2 * Q * n) and gaps_intact(d2, s_in, 2 * Q * n)
    c.ckks_rescale_inplace(out, s_out, 0, batch, c.workspace(hg.OP_CKKS_RESCALE, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, s_out)
    resc = []
    for b in range(batch):
        r = o.ckks_rescale(want[b][:2 * Q * n].copy(), 0)[:2 * (Q - 1) * n]
        resc.append(r)
        assert np.array_equal(got[b][:2 * (Q - 1) * n], r), ("rescale", b)
    assert gaps_intact(out, s_out, 3 * Q * n)
    s_rot = 2 * (Q - 1) * n + 500
    rot = torch.full((batch * s_rot,), SENT, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(out, s_out, rot, s_rot, dgkey, g, 1, batch, c.workspace(hg.OP_CKKS_GALOIS, 1, batch))
    torch.cuda.synchronize()
    gr = hg.to_host(rot).reshape(batch, s_rot)
    for b in range(batch):
        assert np.array_equal(gr[b][:2 * (Q - 1) * n], o.ckks_apply_galois(resc[b], gkey, g, 1)), ("rotate", b)
    assert gaps_intact(rot, s_rot, 2 * (Q - 1) * n) and gaps_intact(out, s_out, 3 * Q * n)


def test_padded_item_strides_method_II_and_bfv(hg, oracle, torch):
    """Padded item strides (see above) through key switching method II (CKKS, two special primes, depth 1) and the
    BFV operators (multiply, relinearize, rotate)."""
    SENT = 0x5555555555555555
    batch = 3
    def strided(items, stride):
        buf = torch.full((batch * stride,), SENT, dtype=torch.int64, device="cuda")
        for b, x in enumerate(items):
            buf[b * stride:b * stride + len(x)] = hg.to_device(x)
        return buf
    def gaps_intact(buf, stride, used):
        return bool((buf.reshape(batch, stride)[:, used:] == SENT).all())
    # ---- CKKS method II
    n = 8192
    c = hg.Context.from_bit_sizes(hg.CKKS, n, [40, 35, 35, 35, 35], [40, 40], sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.CKKS, c.n_power, primes, c.Q_size, c.P_size)
    c.upload()
    Q, P, depth = 5, 2, 1
    Qp, l, d0 = Q + P, Q - depth, -(-Q // P)
    key = synth_key(primes, d0, Qp, n, 3)
    ct1 = [synth_ct(primes, range(l), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(l), 2, n, 2 + 10 * b) for b in range(batch)]
    s_in, s_out = 2 * l * n + 700, 3 * l * n + 900
    d1, d2 = strided(ct1, s_in), strided(ct2, s_in)
    out = torch.full((batch * s_out,), SENT, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, s_in, d2, s_in, out, s_out, depth, batch)
    c.ckks_relinearize_inplace(out, s_out, hg.to_device(key), depth, batch, c.workspace(hg.OP_CKKS_RELIN, depth, batch))
    g = hg.steps_to_galois_elt(1, n, 5)
    rot = torch.full((batch * s_in,), SENT, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(d1, s_in, rot, s_in, hg.to_device(key), g, depth, batch, c.workspace(hg.OP_CKKS_GALOIS, depth, batch))
    torch.cuda.synchronize()
    got, gr = hg.to_host(out).reshape(batch, s_out), hg.to_host(rot).reshape(batch, s_in)
    for b in range(batch):
        w = o.ckks_multiply(ct1[b], ct2[b], depth)
        o.ckks_relinearize_II(w, key, depth)
        assert np.array_equal(got[b][:2 * l * n], w[:2 * l * n]), ("method II relinearize", b)
        assert np.array_equal(gr[b][:2 * l * n], o.ckks_apply_galois_II(ct1[b], key, g, depth)), ("method II rotate", b)
    assert gaps_intact(out, s_out, 3 * l * n) and gaps_intact(rot, s_in, 2 * l * n) and gaps_intact(d1, s_in, 2 * l * n)
    # ---- BFV
    n, t = 4096, 1032193
    c = hg.Context.from_default(hg.BFV, n, 1, t)
    primes = [int(x) for x in c.table("modulus")]
    o = oracle.OracleContext(oracle.BFV, c.n_power, primes, c.Q_size, c.P_size, t)
    c.upload()
    Q, Qp = c.Q_size, c.Q_prime_size
    key, gkey = synth_key(primes, Q, Qp, n, 3), synth_key(primes, Q, Qp, n, 4)
    ct1 = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(batch)]
    s_in, s_out = 2 * Q * n + 300, 3 * Q * n + 1100
    d1, d2 = strided(ct1, s_in), strided(ct2, s_in)
    out = torch.full((batch * s_out,), SENT, dtype=torch.int64, device="cuda")
    c.bfv_multiply(d1, s_in, d2, s_in, out, s_out, batch, c.workspace(hg.OP_BFV_MULTIPLY, 0, batch))
    c.bfv_relinearize_inplace(out, s_out, hg.to_device(key), batch, c.workspace(hg.OP_BFV_RELIN, 0, batch))
    g = hg.steps_to_galois_elt(1, n, 3)
    rot = torch.full((batch * s_in,), SENT, dtype=torch.int64, device="cuda")
    c.bfv_apply_galois(d1, s_in, rot, s_in, hg.to_device(gkey), g, batch, c.workspace(hg.OP_BFV_GALOIS, 0, batch))
    torch.cuda.synchronize()
    got, gr = hg.to_host(out).reshape(batch, s_out), hg.to_host(rot).reshape(batch, s_in)
    for b in range(batch):
        w = o.bfv_relinearize(o.bfv_multiply(ct1[b], ct2[b]), key)
        assert np.array_equal(got[b][:2 * Q * n], w[:2 * Q * n]), ("bfv multiply + relinearize", b)
        assert np.array_equal(gr[b][:2 * Q * n], o.bfv_apply_galois(ct1[b], gkey, g)), ("bfv rotate", b)
    assert gaps_intact(out, s_out, 3 * Q * n) and gaps_intact(rot, s_in, 2 * Q * n) and gaps_intact(d1, s_in, 2 * Q * n)


def test_c4_chain_beyond_one_column_pass_grid(hg, oracle, torch):
    """Config C4's chain with more ciphertexts than one launch of the decomposing column pass takes (digits x Q'
    polynomials per ciphertext: 65535 / 272 = 240): 250 ciphertexts, so the fused key switch runs in two pieces
    and the inverse transform of c2 on its own.  Five distinct inputs: two checked against the oracle, all items
    against their twins."""
    n = 1 << 16
    c, o, primes = _ckks(hg, oracle, n, [60] + [50] * 15, [60])
    Q, Qp = 16, 17
    batch, distinct = 250, 5
    key = synth_key(primes, Q, Qp, n, 3)
    cts = [synth_ct(primes, range(Q), 3, n, 70 + i) for i in range(distinct)]
    base = hg.to_device(np.concatenate(cts)).reshape(distinct, 3 * Q * n)
    reps = (batch + distinct - 1) // distinct
    d = base.repeat(reps, 1)[:batch].contiguous().reshape(-1)
    c.ckks_relinearize_inplace(d, 3 * Q * n, hg.to_device(key), 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = d.reshape(batch, 3 * Q * n)[:, :2 * Q * n]
    for i in (0, 4):
        w = cts[i].copy()
        o.ckks_relinearize(w, key, 0)
        assert np.array_equal(hg.to_host(got[i]), w[:2 * Q * n]), ("relinearize", i)
    assert bool((got == got[:distinct].repeat(reps, 1)[:batch]).all()), "an item of the second piece differs from its twin"
    # the same batch through rotate (key switch first, then the slot scatter) and hoisted rotations (two elements)
    del d
    gkeys = [synth_key(primes, Q, Qp, n, 8), synth_key(primes, Q, Qp, n, 9)]
    elts = [hg.steps_to_galois_elt(1, n, 5), 2 * n - 1]
    words = 2 * Q * n
    src = hg.to_device(np.concatenate([x[:words] for x in cts])).reshape(distinct, words).repeat(reps, 1)[:batch].contiguous().reshape(-1)
    rot = torch.empty(batch * words, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(src, words, rot, words, hg.to_device(gkeys[0]), elts[0], 0, batch, c.workspace(hg.OP_CKKS_GALOIS, 0, batch))
    torch.cuda.synchronize()
    gr = rot.reshape(batch, words)
    assert np.array_equal(hg.to_host(gr[3]), o.ckks_apply_galois(np.ascontiguousarray(cts[3][:words]), gkeys[0], elts[0], 0)), "rotate"
    assert bool((gr == gr[:distinct].repeat(reps, 1)[:batch]).all()), "rotate: an item of the second piece differs"
    hout = torch.empty(batch * 2 * words, dtype=torch.int64, device="cuda")
    c.ckks_rotate_hoisted(src, words, hout, 2 * words, [hg.to_device(k) for k in gkeys], elts, 0, batch,
                          c.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, batch))
    torch.cuda.synchronize()
    gh = hout.reshape(batch, 2 * words)
    assert bool((gh[:, :words] == gr).all()), "hoisted element 0 differs from apply_galois"
    assert np.array_equal(hg.to_host(gh[2][words:]), o.ckks_apply_galois(np.ascontiguousarray(cts[2][:words]), gkeys[1], elts[1], 0)), "hoisted"
    assert bool((gh == gh[:distinct].repeat(reps, 1)[:batch]).all()), "hoisted: an item of the second piece differs"


@pytest.mark.parametrize("sw", [dict(), dict(HEGPU_GALOIS_SCATTER=0), dict(HEGPU_NTT_GALOIS=0)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()) or "default")
def test_arbitrary_galois_elements(hg, oracle, torch, sw):
    """Any odd Galois element below 2N, not only the powers of 5 and 2N - 1 a rotation key set holds: the slot
    scatter (default), the slot gather and the reference's coefficient-domain permutation against the oracle for
    24 random elements plus 1, 3, N - 1, N + 1, 2N - 3 and 2N - 1 (CKKS N = 2^12; BFV N = 2^12)."""
    n = 4096
    rng = np.random.default_rng(5)
    elts = [1, 3, n - 1, n + 1, 2 * n - 3, 2 * n - 1] + [int(2 * v + 1) for v in rng.integers(0, n, 24)]
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [40, 30, 30], [40])
        cb = hg.Context.from_default(hg.BFV, n, 1, 1032193)
        cb.upload()
    pb = [int(x) for x in cb.table("modulus")]
    ob = oracle.OracleContext(oracle.BFV, cb.n_power, pb, cb.Q_size, cb.P_size, 1032193)
    Q, Qp = 3, 4
    key = synth_key(primes, Q, Qp, n, 9)
    ct = synth_ct(primes, range(Q), 2, n, 77)
    d, dk = hg.to_device(ct), hg.to_device(key)
    out = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
    ws = c.workspace(hg.OP_CKKS_GALOIS, 0, 1)
    Qb, Qpb = cb.Q_size, cb.Q_prime_size
    keyb = synth_key(pb, Qb, Qpb, n, 8)
    ctb = synth_ct(pb, range(Qb), 2, n, 78)
    db, dkb = hg.to_device(ctb), hg.to_device(keyb)
    outb = torch.empty(2 * Qb * n, dtype=torch.int64, device="cuda")
    wsb = cb.workspace(hg.OP_BFV_GALOIS, 0, 1)
    for g in elts:
        c.ckks_apply_galois(d, 2 * Q * n, out, 2 * Q * n, dk, g, 0, 1, ws)
        cb.bfv_apply_galois(db, 2 * Qb * n, outb, 2 * Qb * n, dkb, g, 1, wsb)
        torch.cuda.synchronize()
        assert np.array_equal(hg.to_host(out), o.ckks_apply_galois(ct, key, g, 0)), ("ckks", g)
        assert np.array_equal(hg.to_host(outb), ob.bfv_apply_galois(ctb, keyb, g)), ("bfv", g)


@pytest.mark.parametrize("sw", [dict(), dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=1), dict(HEGPU_FUSED_ROW_MAC=1, HEGPU_COL_MULTI=0)],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()) or "default")
def test_long_chain_many_digits(hg, oracle, torch, sw):
    """40 primes in Q (method I: 40 digits, beyond the 32 up to which the integer row pass stays un-reduced and the
    16 up to which hoisting groups four keys): a 59-bit first prime (just above 2^58: the lazy row stages) and a
    60-bit special prime on the integer butterflies, 38 x 30-bit primes and one 50-bit on the FP64 path.
    multiply -> relinearize -> rescale -> rotate and hoisted rotations, N = 2^12."""
    n = 4096
    with backend_switches(**sw):
        c, o, primes = _ckks(hg, oracle, n, [59] + [30] * 38 + [50], [60])
    Q, Qp = 40, 41
    batch = 2
    key, gkey = synth_key(primes, Q, Qp, n, 3), synth_key(primes, Q, Qp, n, 4)
    ct1 = [synth_ct(primes, range(Q), 2, n, 1 + 10 * b) for b in range(batch)]
    ct2 = [synth_ct(primes, range(Q), 2, n, 2 + 10 * b) for b in range(batch)]
    d1, d2 = hg.to_device(np.concatenate(ct1)), hg.to_device(np.concatenate(ct2))
    out = torch.empty(batch * 3 * Q * n, dtype=torch.int64, device="cuda")
    c.ckks_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, 0, batch)
    c.ckks_relinearize_inplace(out, 3 * Q * n, hg.to_device(key), 0, batch, c.workspace(hg.OP_CKKS_RELIN, 0, batch))
    torch.cuda.synchronize()
    got = hg.to_host(out).reshape(batch, -1)
    want = []
    for b in range(batch):
        w = o.ckks_multiply(ct1[b], ct2[b], 0)
        o.ckks_relinearize(w, key, 0)
        want.append(w)
        assert np.array_equal(got[b][:2 * Q * n], w[:2 * Q * n]), ("relinearize", b)
    c.ckks_rescale_inplace(out, 3 * Q * n, 0, batch, c.workspace(hg.OP_CKKS_RESCALE, 0, batch))
    g = hg.steps_to_galois_elt(1, n, 5)
    elts = [g, 2 * n - 1]
    keys = [gkey, synth_key(primes, Q, Qp, n, 6)]
    words = 2 * Q * n
    hout = torch.empty(batch * 2 * words, dtype=torch.int64, device="cuda")
    c.ckks_rotate_hoisted(d1, words, hout, 2 * words, [hg.to_device(k) for k in keys], elts, 0, batch,
                          c.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, batch))
    rot = torch.empty(batch * words, dtype=torch.int64, device="cuda")
    c.ckks_apply_galois(d1, words, rot, words, hg.to_device(gkey), g, 0, batch, c.workspace(hg.OP_CKKS_GALOIS, 0, batch))
    torch.cuda.synchronize()
    got, gh, gr = hg.to_host(out).reshape(batch, -1), hg.to_host(hout).reshape(batch, 2, words), hg.to_host(rot).reshape(batch, words)
    for b in range(batch):
        r = o.ckks_rescale(want[b][:2 * Q * n].copy(), 0)
        assert np.array_equal(got[b][:2 * (Q - 1) * n], r[:2 * (Q - 1) * n]), ("rescale", b)
        wr = o.ckks_apply_galois(ct1[b], gkey, g, 0)
        assert np.array_equal(gr[b], wr), ("rotate", b)
        wh = o.ckks_rotate_hoisted(ct1[b], keys, elts, 0).reshape(2, words)
        assert np.array_equal(gh[b, 0], wh[0]) and np.array_equal(gh[b, 1], wh[1]), ("hoisted", b)


def test_options_change_the_launches_not_the_result(hg, oracle, torch):
    """hegpu_context_set_option on an uploaded context: every call-time option may change between calls on the same
    context and the residues stay the oracle's; fp_ntt (the layout of the uploaded tables) is refused after upload."""
    n = 8192
    c, o, primes = _ckks(hg, oracle, n, [50, 40, 40, 40], [50])
    Q, Qp = 4, 5
    with pytest.raises(hg.HEError) as e:
        c.set_option("fp_ntt", 0)
    assert e.value.code == hg.E_LOGIC
    ct1, ct2 = synth_ct(primes, range(Q), 2, n, 1), synth_ct(primes, range(Q), 2, n, 2)
    key = synth_key(primes, Q, Qp, n, 3)
    want = o.ckks_multiply(ct1, ct2, 0)
    o.ckks_relinearize(want, key, 0)
    d1, d2, dk = hg.to_device(ct1), hg.to_device(ct2), hg.to_device(key)
    ws = c.workspace(hg.OP_CKKS_RELIN, 0, 1)
    for opts in (dict(), dict(fused_row_mac=1, col_multi=1), dict(fused_row_mac=0, fused_moddown=0), dict(single_pass=0),
                 dict(fused_row_mac=1, col_multi=0, fuse_inverse=0), dict(fused_row_mac=-1, col_multi=-1, fused_moddown=1,
                                                                           single_pass=-1, fuse_inverse=1)):
        for k, v in opts.items():
            c.set_option(k, v)
        out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_multiply(d1, 2 * Q * n, d2, 2 * Q * n, out, 3 * Q * n, 0, 1)
        c.ckks_relinearize_inplace(out, 3 * Q * n, dk, 0, 1, ws)
        torch.cuda.synchronize()
        assert np.array_equal(hg.to_host(out)[:2 * Q * n], want[:2 * Q * n]), opts


def test_rotation_inputs_and_results_interleaved_in_one_buffer(hg, oracle, torch):
    """apply_galois / rotate_hoisted refuse a result batch that shares a word with the input batch -- per ITEM: ct and out
    alternating in one buffer (stride 2 x words, no word shared) is a legal layout and computes the oracle's values; an
    out that starts inside an item of ct is refused with E_INVALID, whatever the strides."""
    n, batch = 4096, 3
    c, o, primes = _ckks(hg, oracle, n, [40, 30, 30], [40])
    Q, Qp = 3, 4
    words = 2 * Q * n
    gk = synth_key(primes, Q, Qp, n, 9)
    g = hg.steps_to_galois_elt(1, n, 5)
    cts = [synth_ct(primes, range(Q), 2, n, 5 + b) for b in range(batch)]
    buf = torch.zeros(2 * batch * words, dtype=torch.int64, device="cuda")
    v = buf.view(batch, 2, words)
    for b in range(batch):
        v[b, 0].copy_(hg.to_device(cts[b]))
    ws = c.workspace(hg.OP_CKKS_GALOIS, 0, batch)
    key = hg.to_device(gk)
    c.ckks_apply_galois(buf, 2 * words, buf[words:], 2 * words, key, g, 0, batch, ws)
    torch.cuda.synchronize()
    got = hg.to_host(v[:, 1].contiguous()).reshape(batch, -1)
    for b in range(batch):
        assert np.array_equal(got[b], o.ckks_apply_galois(cts[b], gk, g, 0)), b
    for off in (0, 1, words - 1, 2 * words, 2 * words + words - 1):  # inside item 0 or item 1 of ct
        with pytest.raises(hg.HEError) as e:
            c.ckks_apply_galois(buf, 2 * words, buf[off:], 2 * words, key, g, 0, batch, ws)
        assert e.value.code == hg.E_INVALID, off
    with pytest.raises(hg.HEError):  # unequal strides, out's second item lands on ct's second item
        c.ckks_apply_galois(buf, 2 * words, buf[words:], words, key, g, 0, 2, ws)
    # hoisted: count results per item
    out = torch.zeros(batch * 2 * words, dtype=torch.int64, device="cuda")
    with pytest.raises(hg.HEError):
        c.ckks_rotate_hoisted(buf, 2 * words, buf[words:], 2 * words, [key, key], [g, g], 0, batch,
                              c.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, batch))  # 2 results per item do not fit the gaps
    c.ckks_rotate_hoisted(buf, 2 * words, out, 2 * words, [key, key], [g, g], 0, batch,
                          c.workspace(hg.OP_CKKS_ROTATE_HOISTED, 0, batch))
    torch.cuda.synchronize()
    hv = hg.to_host(out).reshape(batch, 2, -1)
    for b in range(batch):
        assert np.array_equal(hv[b, 0], got[b]) and np.array_equal(hv[b, 1], got[b])

"""GPU parity, config C5 shapes: TFHE gate bootstrapping (pre-computation ->
blind rotate + sample extraction -> key switching) through the C ABI vs the CPU
oracle, bit-exact on the int32 torus.  Two kinds of seeded boot keys: a real
one (torus32 coefficients, NTT'd by the oracle: takes the FP64 blind rotate) and
arbitrary 60-bit residues (the arithmetic is exact for any key material: takes
the integer blind rotate with the reference's prime)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["torus32", "residues60"])
def setup(hg, oracle, request):
    import torch
    assert torch.cuda.is_available()
    t = hg.TfheContext()
    o = oracle.OracleTfhe()
    assert t.prime == o.prime
    rng = np.random.default_rng(11)
    if request.param == "torus32":
        polys = t.int("bootkey_elems") // 1024
        coeff = rng.integers(-2**31, 2**31, (polys, 1024), dtype=np.int64).astype(np.int32)
        coeff[0, :4] = [-2**31, 2**31 - 1, 0, -1]  # range corners of the lo/hi split
        coeff[1, :] = -2**31
        coeff[2, :] = 2**31 - 1
        bk = np.concatenate([o.to_ntt(coeff[i]) for i in range(polys)])
    else:
        bk = rng.integers(0, o.prime, t.int("bootkey_elems"), dtype=np.uint64)
    prepared = t.prepare_bootkey(hg.to_device(bk))
    assert t.prepared_is_fp64(prepared) == (request.param == "torus32")
    ks_a = rng.integers(-2**31, 2**31, t.int("kskey_a_elems"), dtype=np.int64).astype(np.int32)
    ks_b = rng.integers(-2**31, 2**31, t.int("kskey_b_elems"), dtype=np.int64).astype(np.int32)
    return t, o, rng, bk, ks_a, ks_b


def _dev32(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_blind_rotate_and_key_switch(hg, setup):
    import torch
    t, o, rng, bk, ks_a, ks_b = setup
    shape = 6
    a = rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    # exercise the modulus-switch corner cases: a~ = 0, N, 2N; b~ = 0 and N
    a[0], a[1], a[2], a[3] = 0, -2**31, -1, 2**20
    b[0], b[1] = 0, -2**31
    want_a, want_b = o.bootstrapping(a, b, bk)
    prepared = t.prepare_bootkey(hg.to_device(bk))
    out_a = torch.empty(shape * 1024, dtype=torch.int32, device="cuda")
    out_b = torch.empty(shape, dtype=torch.int32, device="cuda")
    t.bootstrapping(_dev32(a), _dev32(b), prepared, out_a, out_b, shape)
    torch.cuda.synchronize()
    assert np.array_equal(out_b.cpu().numpy(), want_b)
    assert np.array_equal(out_a.cpu().numpy(), want_a)
    ks_want_a, ks_want_b = o.key_switching(want_a, want_b, ks_a, ks_b)
    ka = torch.empty(shape * 512, dtype=torch.int32, device="cuda")
    kb = torch.empty(shape, dtype=torch.int32, device="cuda")
    t.key_switching(out_a, out_b, ka, kb, _dev32(ks_a), _dev32(ks_b), shape)
    torch.cuda.synchronize()
    assert np.array_equal(ka.cpu().numpy(), ks_want_a)
    assert np.array_equal(kb.cpu().numpy(), ks_want_b)


@pytest.mark.parametrize("gate", [0, 1, 2, 3, 4, 5, 6])
def test_full_gates(hg, setup, gate):
    import torch
    t, o, rng, bk, ks_a, ks_b = setup
    shape = 3
    a1 = rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)
    a2 = rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)
    b1 = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    b2 = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    want_a, want_b = o.gate(gate, a1, b1, a2, b2, bk, ks_a, ks_b)
    prepared = t.prepare_bootkey(hg.to_device(bk))
    out_a = torch.empty(shape * 512, dtype=torch.int32, device="cuda")
    out_b = torch.empty(shape, dtype=torch.int32, device="cuda")
    ws = torch.empty((512 + 1024 + 2) * shape, dtype=torch.int32, device="cuda")
    t.gate(gate, _dev32(a1), _dev32(b1), _dev32(a2), _dev32(b2), out_a, out_b, prepared, _dev32(ks_a), _dev32(ks_b),
           shape, ws)
    torch.cuda.synchronize()
    assert np.array_equal(out_a.cpu().numpy(), want_a)
    assert np.array_equal(out_b.cpu().numpy(), want_b)


def test_not_gate(hg, setup):
    import torch
    t, o, rng, bk, ks_a, ks_b = setup
    shape = 4
    a = rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    out_a = torch.empty(shape * 512, dtype=torch.int32, device="cuda")
    out_b = torch.empty(shape, dtype=torch.int32, device="cuda")
    t.gate_precompute(hg.GATE_NOT, out_a, out_b, _dev32(a), _dev32(b), None, None, shape)
    torch.cuda.synchronize()
    assert np.array_equal(out_a.cpu().numpy(), (-a.astype(np.int64)).astype(np.int32))
    assert np.array_equal(out_b.cpu().numpy(), (-b.astype(np.int64)).astype(np.int32))

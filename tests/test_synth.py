"""heongpu_amd.synth (bench.py's input generator: numpy on the host, torch on the device) produces exactly the
oracle's o_fill_poly values, which is what tests/helpers.py feeds the parity tests."""
import numpy as np

from helpers import synth_ct, synth_key


def test_numpy_and_torch_generators_equal_the_oracles(oracle):
    import torch

    from heongpu_amd import synth
    primes = [0x800004001, 0x7FFFFFFFFFE0001 >> 3 | 1, (1 << 60) - 93, (1 << 30) + 3, 0x3FFFFFFFFFFC0001]
    n = 1024
    for seed, limb in ((0, 0), (7, 3), (2**40 + 5, 4), (123456789, 2)):
        want = oracle.fill_poly(seed, limb, n, primes[limb])
        assert np.array_equal(synth.fill_poly_np(seed, limb, n, primes[limb]), want)
        got = synth.fill_polys_t(torch, [seed], [limb], [primes[limb]], n, "cpu")[0].numpy().view(np.uint64)
        assert np.array_equal(got, want)
    ct = synth_ct(primes, range(3), 2, n, 11)
    assert np.array_equal(synth.synth_ct_np(primes, range(3), 2, n, 11), ct)
    assert np.array_equal(synth.synth_ct_t(torch, primes, range(3), 2, n, 11, "cpu").numpy().view(np.uint64), ct)
    key = synth_key(primes, 3, 4, n, 5)
    assert np.array_equal(synth.synth_key_np(primes, 3, 4, n, 5), key)
    assert np.array_equal(synth.synth_key_t(torch, primes, 3, 4, n, 5, "cpu", rows_per_call=5).numpy().view(np.uint64), key)

"""Shared helpers for the parity tests: seeded synthetic ciphertexts/keys
(splitmix64(seed + limb*2^32 + idx) mod q_limb)."""
import numpy as np

from oracle import binding as ob


def synth_ct(primes, limb_ids, parts, n, seed):
    """[parts][len(limb_ids)][n] uint64, limb j reduced mod primes[limb_ids[j]]."""
    out = np.zeros((parts, len(limb_ids), n), dtype=np.uint64)
    for p in range(parts):
        for j, lid in enumerate(limb_ids):
            out[p, j] = ob.fill_poly(seed * 1000 + p, lid, n, primes[lid])
    return out.reshape(-1)


def synth_key(primes, Q, Qp, n, seed):
    """evaluation key [Q digits][2][Q' limbs][n] (keygeneration.cu:145-185 layout)."""
    out = np.zeros((Q, 2, Qp, n), dtype=np.uint64)
    for i in range(Q):
        for c in range(2):
            for j in range(Qp):
                out[i, c, j] = ob.fill_poly(seed * 100000 + i * 2 + c, j, n, primes[j])
    return out.reshape(-1)


import contextlib  # noqa: E402


@contextlib.contextmanager
def backend_switches(**kw):
    """Options for every context CREATED inside the block (hegpu_context_set_option through
    heongpu_amd.default_options).  The keys are the option names in capitals with the HEGPU_ prefix of the environment
    variables that seed the same defaults (HEGPU_COL_MULTI=1 -> option "col_multi" = 1); the environment itself is
    not touched."""
    import heongpu_amd as hg
    opts = {}
    for k, v in kw.items():
        assert k.startswith("HEGPU_"), k
        opts[k[len("HEGPU_"):].lower()] = int(v)
    with hg.default_options(**opts):
        yield


def extreme_limbs(c, primes, limb_ids, n, pattern, seed):
    """[len(limb_ids)][n] residues at their extremes: 'max' every residue q - 1, 'alt' 0 / q - 1, 'alt3', 'spike' one
    q - 1, 'half' q/2 and q/2 + 1, else seeded random.  Patterns ending in '_coeff' are built in the COEFFICIENT domain
    and transformed on the device, so that the digits a key switch decomposes (after its inverse transform) are the
    extremes."""
    import torch
    import heongpu_amd as hg
    rows = []
    for j, lid in enumerate(limb_ids):
        q = primes[lid]
        base = pattern.replace("_coeff", "")
        if base == "max":
            v = np.full(n, q - 1, dtype=np.uint64)
        elif base == "alt":
            v = np.zeros(n, dtype=np.uint64)
            v[::2] = q - 1
        elif base == "alt3":
            v = np.full(n, q - 1, dtype=np.uint64)
            v[::3] = 0
        elif base == "spike":
            v = np.zeros(n, dtype=np.uint64)
            v[(seed * 7919 + 13 * j) % n] = q - 1
        elif base == "half":
            v = np.full(n, q // 2, dtype=np.uint64)
            v[1::2] = q // 2 + 1
        else:
            v = ob.fill_poly(seed * 1000 + 17, lid, n, q)
        rows.append(v)
    x = np.concatenate(rows)
    if pattern.endswith("_coeff"):
        d = hg.to_device(x)
        assert list(limb_ids) == list(range(len(limb_ids)))
        c.ntt(d, d, False, len(limb_ids), len(limb_ids))
        torch.cuda.synchronize()
        x = hg.to_host(d)
    return x

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
import os  # noqa: E402


@contextlib.contextmanager
def backend_switches(**kw):
    """The switches are read when a context is uploaded (csrc/context.cpp: Context::upload,
    build_plan), so the context must be created inside the block."""
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

"""Seeded synthetic polynomials for bench.py and the tests: coefficient idx of limb `limb` is
splitmix64(seed + limb * 2^32 + idx) mod q_limb.  Two generators of the same values: numpy on the host (what a CPU
checker is fed) and torch on whatever device a tensor lives on (272 MiB of key material are produced where they are
used instead of through the host).  No dependency on the oracle: tests/test_synth.py pins both against it."""
import numpy as np

_M64 = (1 << 64) - 1
_GOLD, _C1, _C2 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def splitmix64_np(x):
    with np.errstate(over="ignore"):
        z = (x + np.uint64(_GOLD)).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_C1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_C2)
        return z ^ (z >> np.uint64(31))


def fill_poly_np(seed, limb, n, q):
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed & _M64) + (np.uint64(limb) << np.uint64(32))
    return splitmix64_np(idx) % np.uint64(q)


def ct_seed(seed, part):
    return seed * 1000 + part


def key_seed(seed, digit, part):
    return seed * 100000 + digit * 2 + part


def synth_ct_np(primes, limb_ids, parts, n, seed):
    """[parts][len(limb_ids)][n] uint64, limb j reduced mod primes[limb_ids[j]] (tests/helpers.synth_ct)."""
    out = np.empty((parts, len(limb_ids), n), dtype=np.uint64)
    for p in range(parts):
        for j, lid in enumerate(limb_ids):
            out[p, j] = fill_poly_np(ct_seed(seed, p), lid, n, primes[lid])
    return out.reshape(-1)


def synth_key_np(primes, Q, Qp, n, seed):
    """evaluation key [Q digits][2][Q' limbs][n] (keygeneration.cu:145-185 layout; tests/helpers.synth_key)."""
    out = np.empty((Q, 2, Qp, n), dtype=np.uint64)
    for i in range(Q):
        for c in range(2):
            for j in range(Qp):
                out[i, c, j] = fill_poly_np(key_seed(seed, i, c), j, n, primes[j])
    return out.reshape(-1)


# ---- the same values with torch (int64 tensors carrying the uint64 bit patterns, as everywhere in this package)
def _i64(v):
    v &= _M64
    return v - (1 << 64) if v >> 63 else v


def _lsr(torch, z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _splitmix64_t(torch, x):
    z = x + _i64(_GOLD)
    z = (z ^ _lsr(torch, z, 30)) * _i64(_C1)
    z = (z ^ _lsr(torch, z, 27)) * _i64(_C2)
    return z ^ _lsr(torch, z, 31)


def _umod_t(torch, z, q):
    """unsigned z mod q for q < 2^62, on int64 bit patterns"""
    zh = _lsr(torch, z, 1)
    return ((zh % q) * 2 + (z & 1)) % q


def fill_polys_t(torch, seeds, limbs, moduli, n, device):
    """len(seeds) polynomials at once: [len][n] int64 on `device`; row r = fill_poly(seeds[r], limbs[r], n, moduli[r])."""
    base = torch.tensor([_i64(int(s) + (int(l) << 32)) for s, l in zip(seeds, limbs)], dtype=torch.int64, device=device)
    q = torch.tensor([int(m) for m in moduli], dtype=torch.int64, device=device)
    idx = torch.arange(n, dtype=torch.int64, device=device)
    z = _splitmix64_t(torch, base[:, None] + idx[None, :])
    return _umod_t(torch, z, q[:, None])


def synth_ct_t(torch, primes, limb_ids, parts, n, seed, device):
    seeds = [ct_seed(seed, p) for p in range(parts) for _ in limb_ids]
    limbs = [lid for _ in range(parts) for lid in limb_ids]
    return fill_polys_t(torch, seeds, limbs, [primes[l] for l in limbs], n, device).reshape(-1)


def synth_key_t(torch, primes, Q, Qp, n, seed, device, rows_per_call=64):
    """[Q][2][Qp][n] int64 on `device`, produced there in pieces of `rows_per_call` limbs"""
    out = torch.empty(Q * 2 * Qp * n, dtype=torch.int64, device=device)
    rows = [(key_seed(seed, i, c), j) for i in range(Q) for c in range(2) for j in range(Qp)]
    for r0 in range(0, len(rows), rows_per_call):
        chunk = rows[r0:r0 + rows_per_call]
        out[r0 * n:(r0 + len(chunk)) * n] = fill_polys_t(torch, [s for s, _ in chunk], [j for _, j in chunk],
                                                         [primes[j] for _, j in chunk], n, device).reshape(-1)
    return out

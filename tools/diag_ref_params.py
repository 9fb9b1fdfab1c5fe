"""GPU vs oracle on the parameter sets of the reference's own tests (test/test_*.cpp)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import heongpu_amd as hg
from oracle import binding as ob

def ckks(n, log_q, log_p, what):
    c = hg.Context.from_bit_sizes(hg.CKKS, n, log_q, log_p, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    Q, P = len(log_q), len(log_p)
    o = ob.OracleContext(ob.CKKS, c.n_power, primes, Q, P)
    c.upload()
    rg, ro = hg.Rng(11), ob.ORng(11)
    sk, sk_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    print(" sk", np.array_equal(hg.to_host(sk), sk_o), flush=True)
    pk, pk_o = c.generate_public_key(rg, sk), o.gen_public_key(ro, sk_o)
    print(" pk", np.array_equal(hg.to_host(pk), pk_o), flush=True)
    plain = np.concatenate([ob.fill_poly(3, j, n, primes[j]) for j in range(Q)])
    ct = c.ckks_encrypt(rg, pk, hg.to_device(plain)); ct_o = o.ckks_encrypt(ro, pk_o, plain)
    print(" enc", np.array_equal(hg.to_host(ct), ct_o), flush=True)
    if "relin" in what:
        rk, rk_o = c.generate_relin_key(rg, sk), o.gen_switch_key(ro, sk_o, 0)
        print(" rk", np.array_equal(hg.to_host(rk), rk_o), flush=True)
        out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_multiply(ct, 2 * Q * n, ct, 2 * Q * n, out, 3 * Q * n, 0, 1)
        ct3 = o.ckks_multiply(ct_o, ct_o, 0)
        print(" mul", np.array_equal(hg.to_host(out), ct3), flush=True)
        c.ckks_relinearize_inplace(out, 3 * Q * n, rk, 0, 1, c.workspace(hg.OP_CKKS_RELIN, 0, 1))
        (o.ckks_relinearize if P == 1 else o.ckks_relinearize_II)(ct3, rk_o, 0)
        print(" relin", np.array_equal(hg.to_host(out)[:2 * Q * n], ct3[:2 * Q * n]), flush=True)
        c.ckks_rescale_inplace(out, 3 * Q * n, 0, 1, c.workspace(hg.OP_CKKS_RESCALE, 0, 1))
        r = o.ckks_rescale(ct3[:2 * Q * n].copy(), 0) if hasattr(o, "ckks_rescale") else None
        if r is not None:
            print(" rescale", np.array_equal(hg.to_host(out)[:2 * (Q - 1) * n], r[:2 * (Q - 1) * n]), flush=True)
    if "rot" in what:
        gal = hg.steps_to_galois_elt(1, n, 5)
        gk, gk_o = c.generate_galois_key(rg, sk, gal), o.gen_switch_key(ro, sk_o, gal)
        print(" gk", np.array_equal(hg.to_host(gk), gk_o), flush=True)
        rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
        c.ckks_apply_galois(ct, 2 * Q * n, rot, 2 * Q * n, gk, gal, 0, 1, c.workspace(hg.OP_CKKS_GALOIS, 0, 1))
        want = (o.ckks_apply_galois if P == 1 else o.ckks_apply_galois_II)(ct_o, gk_o, gal, 0)
        print(" rot", np.array_equal(hg.to_host(rot), want), flush=True)

def bfv(n, log_q, log_p, t, what):
    c = hg.Context.from_bit_sizes(hg.BFV, n, log_q, log_p, plain_modulus=t, sec=hg.SEC_NONE)
    primes = [int(x) for x in c.table("modulus")]
    Q, P = len(log_q), len(log_p)
    o = ob.OracleContext(ob.BFV, c.n_power, primes, Q, P, t)
    c.upload()
    rg, ro = hg.Rng(12), ob.ORng(12)
    sk, sk_o = c.generate_secret_key(rg), o.gen_secret_key(ro)
    pk, pk_o = c.generate_public_key(rg, sk), o.gen_public_key(ro, sk_o)
    print(" pk", np.array_equal(hg.to_host(pk), pk_o), flush=True)
    msg = np.random.default_rng(1).integers(0, t, n).astype(np.uint64)
    ct = c.bfv_encrypt(rg, pk, hg.to_device(msg)); ct_o = o.bfv_encrypt(ro, pk_o, msg)
    print(" enc", np.array_equal(hg.to_host(ct), ct_o), flush=True)
    dec = c.bfv_decrypt(ct, sk)
    print(" dec", np.array_equal(hg.to_host(dec), o.bfv_decrypt(ct_o, sk_o)), np.array_equal(hg.to_host(dec), msg), flush=True)
    if "relin" in what:
        rk, rk_o = c.generate_relin_key(rg, sk), o.gen_switch_key(ro, sk_o, 0)
        out = torch.empty(3 * Q * n, dtype=torch.int64, device="cuda")
        c.bfv_multiply(ct, 2 * Q * n, ct, 2 * Q * n, out, 3 * Q * n, 1, c.workspace(hg.OP_BFV_MULTIPLY, 0, 1))
        ct3 = o.bfv_multiply(ct_o, ct_o)
        print(" mul", np.array_equal(hg.to_host(out), ct3), flush=True)
        c.bfv_relinearize_inplace(out, 3 * Q * n, rk, 1, c.workspace(hg.OP_BFV_RELIN, 0, 1))
        (o.bfv_relinearize if P == 1 else o.bfv_relinearize_II)(ct3, rk_o)
        print(" relin", np.array_equal(hg.to_host(out)[:2 * Q * n], ct3[:2 * Q * n]), flush=True)
    if "rot" in what:
        gal = hg.steps_to_galois_elt(1, n, 3)
        gk, gk_o = c.generate_galois_key(rg, sk, gal), o.gen_switch_key(ro, sk_o, gal)
        rot = torch.empty(2 * Q * n, dtype=torch.int64, device="cuda")
        c.bfv_apply_galois(ct, 2 * Q * n, rot, 2 * Q * n, gk, gal, 1, c.workspace(hg.OP_BFV_GALOIS, 0, 1))
        want = (o.bfv_apply_galois if P == 1 else o.bfv_apply_galois_II)(ct_o, gk_o, gal)
        print(" rot", np.array_equal(hg.to_host(rot), want), flush=True)

which = sys.argv[1:] or ["a", "b", "c", "d"]
if "a" in which:
    print("CKKS 65536 Q=37 P=1"); ckks(65536, [59] + [45] * 36, [59], "relin")
if "b" in which:
    print("CKKS 32768 Q=19 P=2"); ckks(32768, [59] + [40] * 18, [59, 59], "rot relin")
if "c" in which:
    print("BFV 32768 Q=14"); bfv(32768, [58] * 4 + [59] * 10, [59], 786433, "rot relin")
if "d" in which:
    print("BFV 65536 Q=29"); bfv(65536, [58] * 9 + [59] * 20, [59], 786433, "relin")

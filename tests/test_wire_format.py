"""SURVEY 8f next-3, the serializer: blobs written by the class layer (include/heongpu/heongpu.hpp save / serializer::
save_to_file) are parsed by oracle/wire.py -- a reader written from the REFERENCE's save functions alone -- and every
field and payload is compared with what the live objects hold (tests/cpp/wire_dump.cpp takes those through accessors and
plain device-to-host copies, not through the serializer).  That replaces "round trip with itself" by a comparison
against a second implementation of the format.  What stays unverifiable: sizeof(Modulus64) = 24 (GPU-NTT is not in the
reference tree); the reader assumes it and checks the three words against GPU-NTT's recalled Barrett definition."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "heongpu_amd", "lib", "wire_dump")


def _run(tmp_path, *flags):
    assert os.path.exists(DUMP), "build() compiles tests/cpp/wire_dump.cpp"
    r = subprocess.run([DUMP, str(tmp_path), *flags], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    return json.load(open(tmp_path / "manifest.json"))


def _check_context(tmp_path, name, want, hg):
    d = wire.parse_context(open(tmp_path / (name + ".bin"), "rb").read())
    scheme = name.split("_")[0].rstrip("2")
    assert d["scheme"] == scheme
    assert d["sec_level"] == {0: "none", 1: "sec128"}[want["sec_level"]]
    assert d["keyswitching_type"] == {1: "KEYSWITCHING_METHOD_I", 2: "KEYSWITCHING_METHOD_II"}[want["method"]]
    n = want["n"]
    assert d["n"] == n and d["n_power"] == n.bit_length() - 1
    assert d["Q_prime_size"] == d["Q_size"] + d["P_size"] == d["coeff_modulus"] == len(d["prime_vector"])
    assert all(wire.modulus64_is_consistent(m) for m in d["prime_vector"])
    primes = [m["value"] for m in d["prime_vector"]]
    assert d["base_q"] == primes[:d["Q_size"]] or d["base_q"] == primes  # the reference pushes the whole chain (context.cu:139-147)
    assert d["total_coeff_bit_count"] == sum(d["Q_mod_bit_sizes"]) + sum(d["P_mod_bit_sizes"])
    assert [p.bit_length() for p in primes] == d["Q_mod_bit_sizes"] + d["P_mod_bit_sizes"] or scheme == "bfv"
    if scheme == "bfv":
        assert d["plain_modulus"]["value"] == want["plain_modulus"] and wire.modulus64_is_consistent(d["plain_modulus"])
        chain = [int(v) for v in hg.Context.from_default(hg.BFV, n, 1, want["plain_modulus"]).table("modulus")]
        assert primes == chain  # the chain the context tables are built from, through the C ABI
    if "primes" in want:  # the generated context's own key modulus
        assert primes == want["primes"] and (d["Q_size"], d["P_size"]) == (want["Q_size"], want["P_size"])
    return d


def test_context_blobs_parse_with_the_independent_reader(tmp_path, hg):
    """host-only part (no GPU): a context is serializable before generate()"""
    man = _run(tmp_path, "--context")
    for name, want in man.items():
        _check_context(tmp_path, name, want, hg)
    # a reader must refuse what is not the format
    blob = bytearray(open(tmp_path / "ckks_context.bin", "rb").read())
    with pytest.raises(ValueError):
        wire.parse_context(bytes(blob) + b"\0")
    blob[0] = 9
    with pytest.raises(ValueError):
        wire.parse_context(bytes(blob))


def _payload(tmp_path, name):
    return open(tmp_path / (name + ".payload"), "rb").read()


@pytest.mark.gpu
def test_every_object_parses_and_matches_the_live_object(tmp_path, hg):
    man = _run(tmp_path)
    seen = set()
    for name, want in man.items():
        kind = want["kind"]
        seen.add((name.split("_")[0], kind))
        if kind == "context":
            _check_context(tmp_path, name, want, hg)
            continue
        scheme = name.split("_")[0].rstrip("2")
        d = wire.PARSERS[kind](open(tmp_path / (name + ".bin"), "rb").read())
        assert d["scheme"] == scheme and d["generated"] is True and d["storage_type"] == "DEVICE", name
        if kind == "secretkey":
            for f in ("ring_size", "coeff_modulus_count", "n_power", "hamming_weight"):
                assert d[f] == want[f], (name, f)
            assert d["in_ntt_domain"] and d["size"] == want["ring_size"] * want["coeff_modulus_count"]
            assert d["payload"] == _payload(tmp_path, name)
            # the payload is a ternary key of that Hamming weight in the NTT domain: limbs are not all equal, none is zero
            limbs = np.frombuffer(d["payload"], dtype=np.uint64).reshape(want["coeff_modulus_count"], -1)
            assert all(l.any() for l in limbs)
        elif kind == "publickey":
            assert d["ring_size"] == want["ring_size"] and d["coeff_modulus_count"] == want["coeff_modulus_count"]
            assert d["in_ntt_domain"] and d["size"] == 2 * want["ring_size"] * want["coeff_modulus_count"]
            assert d["payload"] == _payload(tmp_path, name)
            # serializer::save_to_file: u64 size + one zlib stream holding exactly the save() bytes
            framed = open(tmp_path / (name + ".file"), "rb").read()
            assert wire.unframe_file(framed) == open(tmp_path / (name + ".bin"), "rb").read()
        elif kind in ("relinkey", "switchkey"):
            for f in ("ring_size", "Q_prime_size", "Q_size", "d", "size"):
                assert d[f] == want[f], (name, f)
            assert d["size"] == 2 * d["d"] * d["Q_prime_size"] * d["ring_size"]  # evaluationkey.cu:30-36
            if kind == "relinkey":
                assert d["key_type"] == {1: "KEYSWITCHING_METHOD_I", 2: "KEYSWITCHING_METHOD_II"}[want["method"]]
            assert d["payload"] == _payload(tmp_path, name)
        elif kind == "galoiskey":
            for f in ("ring_size", "Q_prime_size", "Q_size", "d", "group_order", "galois_elt_zero", "size"):
                assert d[f] == want[f], (name, f)
            assert d["customized"] == bool(want["customized"])
            assert d["group_order"] == (3 if scheme == "bfv" else 5)  # bfv/evaluationkey.cu:308, ckks/evaluationkey.cu:408
            assert d["galois_elt_zero"] == 2 * d["ring_size"] - 1
            if d["customized"]:
                assert d["custom_galois_elt"] == want["custom_galois_elt"]
            else:
                assert d["galois_elt"] == {int(k): v for k, v in want["galois_elt"].items()}
                for shift, elt in d["galois_elt"].items():  # keygeneration.cu:684-728 steps_to_galois_elt
                    assert elt == hg.steps_to_galois_elt(shift, d["ring_size"], d["group_order"])
            live = set(want["key_elements"])
            assert set(d["keys"]) | {d["galois_elt_zero"]} == live and d["galois_elt_zero"] not in d["keys"]
            for elt, data in d["keys"].items():
                assert data == _payload(tmp_path, "%s_%d" % (name, elt)), (name, elt)
            assert d["zero_key"] == _payload(tmp_path, "%s_%d" % (name, d["galois_elt_zero"]))
        elif kind == "plaintext":
            assert d["plain_size"] == d["size"] == want["plain_size"] and d["in_ntt_domain"] == bool(want["in_ntt_domain"])
            if scheme == "ckks":
                assert d["depth"] == want["depth"] and d["scale"] == want["scale"]
                assert d["encoding"] == {0: "SLOT", 1: "COEFFICIENT"}[want["encoding"]]
            assert d["payload"] == _payload(tmp_path, name)
        elif kind == "ciphertext":
            for f in ("ring_size", "coeff_modulus_count", "cipher_size", "size"):
                assert d[f] == want[f], (name, f)
            assert d["in_ntt_domain"] == bool(want["in_ntt_domain"])
            assert d["relinearization_required"] == bool(want["relinearization_required"]) == (d["cipher_size"] == 3)
            depth = d.get("depth", 0)
            assert d["size"] == d["cipher_size"] * (d["coeff_modulus_count"] - depth) * d["ring_size"]  # ciphertext.cu:204-205
            if scheme == "ckks":
                assert d["depth"] == want["depth"] and d["scale"] == want["scale"]
                assert d["rescale_required"] == bool(want["rescale_required"])
                assert d["encoding"] == {0: "SLOT", 1: "COEFFICIENT"}[want["encoding"]]
            assert d["payload"] == _payload(tmp_path, name)
        else:
            raise AssertionError(kind)
    kinds = {"context", "secretkey", "publickey", "relinkey", "switchkey", "galoiskey", "plaintext", "ciphertext"}
    for scheme in ("bfv", "ckks", "ckks2"):
        assert {k for s, k in seen if s == scheme} == kinds, scheme

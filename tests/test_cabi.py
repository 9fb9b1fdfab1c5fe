"""CPU: the C-ABI library loads and exports every symbol declared in
include/hegpu.h; host-only entry points behave like the reference's
exceptions (no compute calls here -- there is no GPU on this box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    # the drop-in boundary + the bench-only measurement seam (hegpu_bench.h, one entry)
    src = open(os.path.join(ROOT, "include", "hegpu.h")).read() + open(os.path.join(ROOT, "include", "hegpu_bench.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hegpu_[A-Za-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(hg):
    from heongpu_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    bound = {s[0] for s in _lib.SIGNATURES}
    for name in names:
        assert hasattr(lib, name), f"{name} declared in hegpu.h but not exported by libhegpu.so"
        assert name in bound, f"{name} has no ctypes signature"
    assert bound <= set(names), "ctypes binds a symbol the header does not declare"


def test_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C, no torch/C++ types."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write('#include "hegpu.h"\nint main(void){return 0;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", c,
                               "-o", os.path.join(d, "t.o")])


def test_context_errors_mirror_reference_exceptions(hg):
    # std::logic_error cases (reference ckks/context.cu:33-44, util.cu:11-56)
    for n in (1000, 2048, 131072):
        with pytest.raises(hg.HEError) as e:
            hg.Context.from_bit_sizes(hg.CKKS, n, [40, 30], [40], sec=hg.SEC_NONE)
        assert e.value.code == hg.E_LOGIC
    with pytest.raises(hg.HEError) as e:
        hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [], sec=hg.SEC_NONE)
    assert e.value.code == hg.E_LOGIC and "cannot be empty" in str(e.value)
    with pytest.raises(hg.HEError) as e:  # P must cover every group of |P| Q primes
        hg.Context.from_bit_sizes(hg.CKKS, 8192, [50, 50], [40], sec=hg.SEC_NONE)
    assert e.value.code == hg.E_LOGIC and "bigger than Q" in str(e.value)
    with pytest.raises(hg.HEError) as e:  # invalid modulus bit size (util.cu:252-256)
        hg.Context.from_bit_sizes(hg.CKKS, 8192, [29, 29], [40], sec=hg.SEC_NONE)
    assert e.value.code == hg.E_LOGIC
    # std::runtime_error: security check (ckks/context.cu:94-119, secstdparams.h:25-41)
    with pytest.raises(hg.HEError) as e:
        hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30, 30], [40], sec=hg.SEC_128)
    assert e.value.code == hg.E_RUNTIME and "security" in str(e.value)
    hg.Context.from_bit_sizes(hg.CKKS, 4096, [36, 36], [37], sec=hg.SEC_128)  # 109 bits: allowed
    with pytest.raises(hg.HEError):  # BFV needs a plain modulus
        hg.Context.from_default(hg.BFV, 4096, 1, 0)


def test_default_chain_and_properties(hg):
    c = hg.Context.from_default(hg.BFV, 32768, 1, 786433)  # config C3 parameters
    assert (c.n_power, c.Q_size, c.P_size, c.Q_prime_size) == (15, 14, 1, 15)
    assert c.bsk_modulus in (15, 16)
    assert int(c.table("modulus")[0]) == 0x2000000002b0001
    assert c.workspace_bytes(hg.OP_BFV_GALOIS, 0, 64) == 64 * (14 * 15 + 2 * 15) * 32768 * 8


def test_galois_elements(hg, oracle):
    for n in (4096, 65536):
        for steps in (0, 1, 2, -1, -7, 100):
            for order in (3, 5):
                assert hg.steps_to_galois_elt(steps, n, order) == oracle.lib().o_steps_to_galois_elt(steps, n, order)


def test_no_device_means_loud_failure(hg):
    """without a GPU every device entry point must fail loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = hg.Context.from_bit_sizes(hg.CKKS, 4096, [36, 36], [37])
    with pytest.raises(hg.HEError) as e:
        c.upload()
    assert e.value.code == hg.E_NODEVICE
    with pytest.raises(hg.HEError) as e:
        c.ntt(0, 0, False, 1, 1, stream=0)
    assert e.value.code == hg.E_NODEVICE


def test_drbg_is_chacha20_rfc8439_vector():
    """The DRBG's PRF is the ChaCha20 block function (csrc/drbg.hpp): RFC 8439 section 2.3.2 test vector
    (key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00), whose counter / nonce words
    12..15 are (index lo, index hi, stream lo, stream hi) here.  The CPU oracle's independent
    implementation must give the same words."""
    import ctypes
    from heongpu_amd import _lib
    from oracle import binding as ob
    key = bytes(range(32))
    index = 1 | (0x09000000 << 32)
    stream = 0x4A000000
    out = (ctypes.c_uint32 * 4)()
    assert _lib.load().hegpu_drbg_block(key, stream, index, out) == 0
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3]
    assert list(out) == want
    kw = (ctypes.c_uint32 * 8)(*[int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)])
    o = (ctypes.c_uint32 * 4)()
    ob.lib().o_drbg_block(kw, stream, index, o)
    assert list(o) == want


def test_entropy_seeded_generators_differ():
    """hegpu_rng_create_from_entropy draws 256 bits from the OS: two generators never share a key
    (checked through the only observable without a device: creation succeeds and handles are distinct)."""
    import ctypes
    from heongpu_amd import _lib
    lib = _lib.load()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.hegpu_rng_create_from_entropy(ctypes.byref(a)) == 0
    assert lib.hegpu_rng_create_from_entropy(ctypes.byref(b)) == 0
    assert a.value and b.value and a.value != b.value
    lib.hegpu_rng_destroy(a)
    lib.hegpu_rng_destroy(b)
    assert lib.hegpu_rng_create_seeded(None, ctypes.byref(a)) != 0


def test_options_are_an_interface_not_the_environment(hg):
    """hegpu_context_set_option / get_option (include/hegpu.h): names, ranges, error codes; the environment only seeds
    the defaults of a context at creation, a later change of the variable does nothing; no product source reads the
    environment on a call path."""
    import subprocess
    import sys
    c = hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [40], sec=hg.SEC_NONE)
    defaults = {"fused_row_mac": -1, "fused_moddown": 1, "col_multi": -1, "single_pass": -1, "ntt_galois": 1,
                "galois_scatter": 1, "fuse_inverse": 1, "copy_along": 1, "digit_split": -1, "fp_ntt": 1, "behz_split": -1, "fused_tensor": 1}
    for k, v in defaults.items():
        assert c.get_option(k) == v, k
    for k, v in (("fused_row_mac", 0), ("col_multi", 1), ("digit_split", 4), ("fp_ntt", 0), ("behz_split", 1)):
        c.set_option(k, v)
        assert c.get_option(k) == v
    for name, value in (("no_such_option", 1), ("col_multi", 2), ("digit_split", 3), ("fused_moddown", -1)):
        with pytest.raises(hg.HEError) as e:
            c.set_option(name, value)
        assert e.value.code == hg.E_INVALID
    with hg.default_options(single_pass=0):
        assert hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [40], sec=hg.SEC_NONE).get_option("single_pass") == 0
    assert hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [40], sec=hg.SEC_NONE).get_option("single_pass") == -1
    # the environment seeds defaults at creation (a fresh process: the variable is set before the library reads it)
    code = ("import os, heongpu_amd as hg\n"
            "c = hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [40], sec=hg.SEC_NONE)\n"
            "os.environ['HEGPU_COL_MULTI'] = '0'\n"
            "d = hg.Context.from_bit_sizes(hg.CKKS, 4096, [40, 30], [40], sec=hg.SEC_NONE)\n"
            "print(c.get_option('col_multi'), c.get_option('fp_ntt'), d.get_option('col_multi'))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                         env=dict(os.environ, HEGPU_COL_MULTI="1", HEGPU_FP_NTT="0"))
    assert out.returncode == 0, out.stderr[-800:]
    assert out.stdout.split() == ["1", "0", "0"]
    # getenv appears only where defaults are seeded (context / TFHE context creation) and in the class layer's
    # debugging aids of the memory pool
    allowed = {"context.cpp": 1}  # env_long, the one reader (context and TFHE defaults)
    csrc = os.path.join(ROOT, "heongpu_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".hpp", ".cuh")):
            n = len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, f)).read()))
            assert n == allowed.get(f, 0), (f, n)


def test_behz_base_beyond_the_lazy_sums_is_refused():
    """A BFV base whose worst-case row sum of the BEHZ base conversions does not fit 128 bits is refused when the
    context is created (exact arithmetic on the moduli at hand: 63 primes of 61 bits + their 64 base primes do not
    fit, 58 primes of 60 bits do) -- not computed wrongly later (ADVICE r3 on redc128)."""
    import heongpu_amd as hg
    with pytest.raises(hg.HEError) as e:
        hg.Context.from_bit_sizes(hg.BFV, 4096, [61] * 63, [61], plain_modulus=65537, sec=hg.SEC_NONE)
    assert e.value.code == hg.E_INVALID and "lazy 128-bit row sum" in str(e.value)
    c = hg.Context.from_bit_sizes(hg.BFV, 4096, [60] * 58, [60], plain_modulus=65537, sec=hg.SEC_NONE)
    assert len(c.table("q_Bsk_merge_modulus")) == 58 + 59

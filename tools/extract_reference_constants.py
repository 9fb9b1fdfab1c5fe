#!/usr/bin/env python3
"""Build-container only: read every constant the reference tree holds for the hot path into
tests/golden/reference_constants.json (data only -- numbers, no source text).

  src/lib/util/defaultmodulus.cpp:12-175        default prime chains, 128 / 192 / 256-bit security
  src/include/heongpu/util/secstdparams.h:22-79 error_std_dev and the max log2(Q*P) tables
  src/lib/host/tfhe/context.cu:23-57            the hard-coded TFHE parameter set
  benchmark/benchmark_{ckks,bfv}.cpp            the parameter sets of the reference's own harness

The oracle (oracle/*.c) and the product's host-side parameter code (csrc/host_params.cpp, csrc/tfhe
context) are tested against this file (tests/test_oracle_golden.py); with no limb-level vectors in the
reference and no way to build it here (CUDA + empty submodules), these constants plus the reference's own
unchanged test programs are all the reference-held data there is to pin against.

usage: python tools/extract_reference_constants.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def default_modulus():
    src = read("src/lib/util/defaultmodulus.cpp")
    out = {}
    for level in ("128", "192", "256"):
        body = src[src.index("get_%sbit_sec_modulus()" % level):]
        body = body[:body.index("return default_modulus_%s" % level)]
        chains = {}
        # {N, { Modulus64(0x...), ... }}
        for m in re.finditer(r"\{\s*(\d+)\s*,\s*\{((?:\s*Modulus64\(0x[0-9a-fA-F]+\)\s*,?)+)\s*\}\s*\}", body):
            chains[m.group(1)] = [int(v, 16) for v in re.findall(r"Modulus64\((0x[0-9a-fA-F]+)\)", m.group(2))]
        assert sorted(chains, key=int) == ["4096", "8192", "16384", "32768", "65536"], (level, list(chains))
        out[level] = chains
    return out


def sec_tables():
    src = read("src/include/heongpu/util/secstdparams.h")
    out = {"error_std_dev": float(re.search(r"error_std_dev\s*=\s*([0-9.]+)", src).group(1)), "max_logq": {}}
    for level in ("128", "192", "256"):
        body = src[src.index("heongpu_%sbit_std_parms" % level):]
        body = body[:body.index("return 0;")]
        out["max_logq"][level] = {n: int(v) for n, v in re.findall(r"case\s+(\d+):\s*return\s+(\d+);", body)}
        assert len(out["max_logq"][level]) == 5
    return out


def tfhe():
    src = read("src/lib/host/tfhe/context.cu")
    g = lambda pat: re.search(pat, src).group(1)
    return {
        "prime": int(g(r"prime_\s*=\s*Modulus64\((\d+)ULL\)")),
        "psi": int(g(r"Data64 psi\s*=\s*(\d+)ULL")),
        "ntt_log_size": int(g(r"compute_ntt_table\(psi, prime_, (\d+)\)")),
        "ks_base_bit": int(g(r"ks_base_bit_\s*=\s*(\d+)")),
        "ks_length": int(g(r"ks_length_\s*=\s*(\d+)")),
        "ks_stdev_times_sqrt_pi_over_2": g(r"ks_stdev_\s*=\s*\(([^)]*)\)"),
        "bk_stdev_times_sqrt_pi_over_2": g(r"bk_stdev_\s*=\s*\(([^)]*)\)"),
        "max_stdev_times_sqrt_pi_over_2": g(r"max_stdev_\s*=\s*\(([^)]*)\)"),
        "n": int(g(r"\bn_\s*=\s*(\d+);")),
        "N": int(g(r"\bN_\s*=\s*(\d+);")),
        "k": int(g(r"\bk_\s*=\s*(\d+);")),
        "bk_l": int(g(r"bk_l_\s*=\s*(\d+);")),
        "bk_bg_bit": int(g(r"bk_bg_bit_\s*=\s*(\d+);")),
    }


def harness_sets():
    """poly degrees / bit sizes / plain moduli of benchmark_ckks.cpp and benchmark_bfv.cpp (uncommented lines)"""
    out = {}
    src = "\n".join(l for l in read("benchmark/benchmark_ckks.cpp").splitlines() if not l.strip().startswith("//"))
    ints = lambda txt: [int(v) for v in re.findall(r"\d+", txt)]
    deg = ints(re.search(r"poly_modulus_degrees\s*=\s*\{([^}]*)\}", src).group(1))
    nested = lambda name: [ints(g) for g in re.findall(r"\{([\d,\s]+)\}", re.search(name + r"\s*=\s*\{((?:\s*\{[\d,\s]+\}\s*,?)+)\s*\}", src).group(1))]
    lq, lp = nested("log_Q_bit_sizes"), nested("log_P_bit_sizes")
    sc = ints(" ".join(re.findall(r"pow\(2\.0,\s*(\d+)\)", re.search(r"scales\s*=\s*\{([^}]*)\}", src).group(1))))
    out["benchmark_ckks"] = [{"n": n, "log_q": q, "log_p": p_, "log_scale": s_} for n, q, p_, s_ in zip(deg, lq, lp, sc)]
    src = "\n".join(l for l in read("benchmark/benchmark_bfv.cpp").splitlines() if not l.strip().startswith("//"))
    m = re.search(r"poly_modulus_degrees\s*=\s*\{([\d,\s]+)\}", src)
    t = re.search(r"plain_modulus\s*=\s*\{([\d,\s]+)\}", src)
    if m and t:
        out["benchmark_bfv"] = {"n": [int(v) for v in m.group(1).split(",") if v.strip()],
                                "plain_modulus": [int(v) for v in t.group(1).split(",") if v.strip()]}
    return out


def main():
    data = {"source": "Alisah-Ozcan/HEonGPU reference tree (extracted by tools/extract_reference_constants.py)",
            "default_modulus": default_modulus()}
    data.update(sec_tables())
    data["tfhe"] = tfhe()
    data["harness"] = harness_sets()
    path = os.path.join(ROOT, "tests", "golden", "reference_constants.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", path, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in data.items()})


if __name__ == "__main__":
    main()

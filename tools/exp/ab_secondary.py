#!/usr/bin/env python3
"""one line for tools/lib_ab.sh: headline + the secondary workloads' rates of the library currently installed"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "20", "--detail", "/tmp/d.json"], capture_output=True, text=True, cwd=ROOT)
d = json.load(open("/tmp/d.json"))
s = d["secondary"]
print("step %.3f ms | bfv14 %.0f | c3 %.0f | c2 b64 %.0f b1 %.1f us | m2 %.0f | c5 %.0f | hoist %.0f | ntt %s" % (
    d["ms_per_step"], s["bfv_n14_multiply"]["multiplications_per_s"], s["c3_bfv_n15_rotate"]["rotations_per_s"],
    s["c2_ckks_n14"]["ops_per_s_batch64"], s["c2_ckks_n14"]["latency_us_batch1"], s["ckks_n16_method_II"]["multiply_relinearize_per_s"],
    s["c5_tfhe_gates"]["gates_per_s"], s["hoisted_rotations"]["by_k"]["8"]["hoisted_rotations_per_s"],
    " ".join("%.3f" % v["forward_frac"] for v in d["ntt_by_degree"].values())))

#!/usr/bin/env python3
"""one line for tools/lib_ab.sh: the C4 step and the stand-alone transform pair of the library currently installed"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-secondary", "--no-cpu-baseline", "--steps", "20", "--detail", "/tmp/d.json"],
                     capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()[-1]
d = json.loads(out)
print("ms %.3f frac %.4f fwd %.0f inv %.0f GB/s" % (d["ms_per_step"], d["roofline"]["frac"], d["ntt"]["forward_GBps"], d["ntt"]["inverse_GBps"]))

"""Blind rotate + sample extraction of `shape` gates per call, both FP64 forms (option "wide_max"): which form from
how many gates.  GPU only; random torus32 boot key (timing does not depend on the key values).
  python tools/tfhe_shapes.py [shape ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import heongpu_amd as hg  # noqa: E402


def main():
    shapes = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 192, 256, 320, 384, 512]
    t = hg.TfheContext()
    rng = np.random.default_rng(3)
    rg = hg.Rng(5)
    lwe, tlwe = t.generate_secret_key(rg)
    bk = t.generate_bootstrapping_key(rg, lwe, tlwe)[0]  # a real key: torus32 coefficients -> the FP64 layout
    prepared = t.prepare_bootkey(bk)
    assert t.prepared_is_fp64(prepared)
    for shape in shapes:
        a = torch.from_numpy(rng.integers(-2**31, 2**31, shape * 512, dtype=np.int64).astype(np.int32)).cuda()
        b = torch.from_numpy(rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)).cuda()
        outs = {}
        line = "%5d gates:" % shape
        for name, wm in (("16 waves/gate", 1 << 30), ("4 waves/gate", 0)):
            t.set_option("wide_max", wm)
            oa = torch.empty(shape * 1024, dtype=torch.int32, device="cuda")
            ob = torch.empty(shape, dtype=torch.int32, device="cuda")
            t.bootstrapping(a, b, prepared, oa, ob, shape)
            torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                t.bootstrapping(a, b, prepared, oa, ob, shape)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            outs[name] = (oa.cpu().numpy(), ob.cpu().numpy())
            line += "  %s %.3f ms" % (name, ms)
        same = all(np.array_equal(x, y) for x, y in zip(outs["16 waves/gate"], outs["4 waves/gate"]))
        print(line + "  equal=%s" % same, flush=True)


if __name__ == "__main__":
    main()

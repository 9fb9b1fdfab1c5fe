"""TFHE key switching alone, gates per call x form (option "ks_batched"; the coefficient loop cut into pieces by launch
size in every form): where the several-gates-per-workgroup forms start to pay.  python tools/tfhe_ks_shapes.py [shape ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heongpu_amd as hg

shapes = [int(a) for a in sys.argv[1:]] or [8, 64, 128, 256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192]
t = hg.TfheContext()
rng = np.random.default_rng(2)
ks_a = torch.from_numpy(rng.integers(-2**31, 2**31, t.int("kskey_a_elems"), dtype=np.int64).astype(np.int32)).cuda()
ks_b = torch.from_numpy(rng.integers(-2**31, 2**31, t.int("kskey_b_elems"), dtype=np.int64).astype(np.int32)).cuda()
for shape in shapes:
    ea = torch.from_numpy(rng.integers(-2**31, 2**31, shape * 1024, dtype=np.int64).astype(np.int32)).cuda()
    eb = torch.from_numpy(rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)).cuda()
    line, outs = "%5d gates:" % shape, []
    for form in (0, 8, 16):
        t.set_option("ks_batched", form)
        ka = torch.empty(shape * 512, dtype=torch.int32, device="cuda")
        kb = torch.empty(shape, dtype=torch.int32, device="cuda")
        t.key_switching(ea, eb, ka, kb, ks_a, ks_b, shape)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t.key_switching(ea, eb, ka, kb, ks_a, ks_b, shape)
        torch.cuda.synchronize()
        line += "  %s %.3f ms" % ("%d gates per workgroup x pieces" % form if form else "one gate per workgroup (x pieces)", (time.perf_counter() - t0) / 5 * 1e3)
        outs.append((ka.cpu().numpy(), kb.cpu().numpy()))
    print(line + "  equal=%s" % all(np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) for o in outs[1:]), flush=True)

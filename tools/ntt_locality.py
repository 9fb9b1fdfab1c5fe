"""Forward / inverse NTT throughput against the size of the working set (L2 32 MiB, Infinity Cache 256 MiB, HBM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import heongpu_amd as hg

N = 65536
c = hg.Context.from_bit_sizes(hg.CKKS, N, [60] + [50] * 15, [60])
c.upload()
rc = 17
stream = torch.cuda.current_stream().cuda_stream
for polys in (17, 34, 68, 136, 272, 544, 1088, 4352, 17408):
    buf = torch.randint(0, 1 << 40, (polys * N,), dtype=torch.int64, device="cuda")
    out = torch.empty_like(buf)
    for inverse in (False, True):
        for _ in range(3):
            c.ntt(buf, out, inverse, polys, rc, stream=stream)
        reps = max(5, 4000 // polys)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            c.ntt(buf, out, inverse, polys, rc, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%5d polys (%7.1f MiB in + out) %s: %8.3f ms  %7.1f ns/poly  %6.0f GB/s (2W per poly)"
              % (polys, 2 * polys * N * 8 / 2**20, "inv" if inverse else "fwd", ms, ms * 1e6 / polys,
                 polys * 2 * N * 8 / (ms * 1e-3) / 1e9))

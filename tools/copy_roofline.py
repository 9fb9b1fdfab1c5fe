"""What a plain device-to-device copy (1:1 read/write stream) sustains on this GPU: the practical
ceiling for the two NTT passes, which read W and write W per polynomial each."""
import torch
for mib in (64, 136, 512, 2048, 8704):
    n = mib * (1 << 20) // 8
    src = torch.empty(n, dtype=torch.int64, device="cuda").random_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    reps = max(5, 20000 // mib)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("copy %5d MiB -> %5d MiB: %7.3f ms  %6.0f GB/s (read + write)" % (mib, mib, ms, 2 * n * 8 / (ms * 1e-3) / 1e9))
# read-only reduction for comparison
for mib in (512, 8704):
    n = mib * (1 << 20) // 8
    src = torch.empty(n, dtype=torch.int64, device="cuda").random_()
    for _ in range(3):
        src.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        src.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("sum  %5d MiB: %7.3f ms  %6.0f GB/s (read only)" % (mib, ms, n * 8 / (ms * 1e-3) / 1e9))
# write-only stream
for mib in (512, 8704):
    n = mib * (1 << 20) // 8
    dst = torch.empty(n, dtype=torch.int64, device="cuda")
    for _ in range(3):
        dst.fill_(7)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.fill_(7)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("fill %5d MiB: %7.3f ms  %6.0f GB/s (write only)" % (mib, ms, n * 8 / (ms * 1e-3) / 1e9))

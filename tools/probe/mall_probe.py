"""Does data just written by one kernel come back faster than data that has been pushed out of the memory-side cache?
Read a buffer of N MB right after writing it, and after streaming 1 GB through in between.   python tools/probe/mall_probe.py"""
import torch
dev = "cuda"
big_a = torch.empty(256 * 1024 * 1024 // 4, device=dev)      # 1 GB of traffic per copy (2 x 256 MB read + write... 512 MB)
big_b = torch.empty_like(big_a)
def timed(fn, n=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(n):
        prep = fn(None)
        ev[0].record()
        fn(prep)
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for mb in (16, 64, 128, 200, 400):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    def hot(prep):
        if prep is None:
            x.fill_(1.0)            # written just before
            return 1
        y.copy_(x)                  # read x (and write y)
    def cold(prep):
        if prep is None:
            x.fill_(1.0)
            big_b.copy_(big_a)      # 512 MB of other traffic in between
            big_a.copy_(big_b)
            return 1
        y.copy_(x)
    th, tc = timed(hot), timed(cold)
    print(f"{mb:4d} MB: copy right after the write {th:7.1f} us ({2 * mb / th * 1e-3 * 1e3:6.0f} GB/s), after 1 GB of other traffic {tc:7.1f} us ({2 * mb / tc * 1e-3 * 1e3:6.0f} GB/s)")

#!/usr/bin/env python3
"""Is a memset NODE of a hipGraph ordered after the kernel node captured before it?
Graph: [fill kernel: 1 GiB buffer <- NaN] -> [hipMemsetAsync: last 256 KiB <- 0] -> [kernel: y = tail + 1].
If the memset may start before the fill kernel has drained, the fill's late writes land on top of the zeros
and y holds NaN.  The same chain is also run eagerly (stream order) as the control."""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
N = 256 * 1024 * 1024          # floats = 1 GiB
TAIL = 64 * 1024               # floats = 256 KiB
big = torch.zeros(N, device=dev)
tail = big[N - TAIL:]
y = torch.zeros(TAIL, device=dev)


def chain():
    big.fill_(float("nan"))
    rc = hip.hipMemsetAsync(tail.data_ptr(), 0, TAIL * 4, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.add(tail, 1.0, out=y)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    bad = 0
    for _ in range(20):
        chain()
        torch.cuda.synchronize()
        bad += int(torch.isnan(y).any())
    print(f"eager (stream order): {bad} of 20 runs left NaN in the tail")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
bad, worst = 0, 0
for _ in range(20):
    g.replay()
    torch.cuda.synchronize()
    n = int(torch.isnan(y).sum())
    bad += int(n > 0)
    worst = max(worst, n)
print(f"hipGraph replay: {bad} of 20 replays left NaN in the tail (worst: {worst} of {TAIL} elements)")

#!/usr/bin/env python3
"""Is a hipMemsetAsync that ANOTHER thread issues into a stream under thread-local capture recorded
into the graph?  (torch runs the backward of a captured step on its autograd worker thread.)
Prints, for the memset issued from the capturing thread and from a second thread:
  after capture (no replay yet): was the buffer cleared eagerly?    after a replay: was it cleared by the graph?"""
import ctypes
import threading

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")


def run(from_thread: bool, mode: str):
    buf = torch.full((1024,), 7.0, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    rc = []
    with torch.cuda.graph(g, capture_error_mode=mode):
        st = torch.cuda.current_stream().cuda_stream

        def work():
            rc.append(hip.hipMemsetAsync(buf.data_ptr(), 0, 4096, st))

        if from_thread:
            t = threading.Thread(target=work)
            t.start()
            t.join()
        else:
            work()
        y = buf + 1.0
    torch.cuda.synchronize()
    eager = float(buf[0])
    buf.fill_(7.0)
    g.replay()
    torch.cuda.synchronize()
    print(f"memset from {'second' if from_thread else 'capturing'} thread, mode {mode}: rc {rc[0]}; buffer after capture "
          f"{eager} (7 = not executed eagerly); after replay buf {float(buf[0])}, buf+1 {float(y[0])} "
          f"(0 / 1 = memset is a graph node; 7 / 8 = it is NOT in the graph)")


for mode in ("thread_local", "global", "relaxed"):
    for ft in (False, True):
        try:
            run(ft, mode)
        except Exception as e:
            print(f"from_thread={ft} mode={mode}: {type(e).__name__}: {str(e)[:200]}")

#!/usr/bin/env python3
"""Does the two-sub-batch schedule depend on WHEN its internal streams were created?  (It did: created after a
graph capture in the same process they overlapped badly.)  Sequence: capture a small plan first, then measure."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
def rate(flags):
    f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    f(); f(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(B * K / float(np.median(ts)))
if len(sys.argv) > 1 and sys.argv[1] == "torch-graph-first":  # a capture that does not involve this library at all
    y = torch.zeros(1024, device="cuda")
    g = torch.cuda.CUDAGraph()
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0): y += 1
    torch.cuda.current_stream().wait_stream(s0)
    with torch.cuda.graph(g): y += 1
    g.replay(); torch.cuda.synchronize()
    print("a plain torch graph was captured first" + (" and deleted" if "delete" in sys.argv else ""))
    if "delete" in sys.argv:
        del g
        import gc; gc.collect(); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "capture-first":
    small = nat.EncodePlan(1, 8192, nat.unit_norm(torch.rand(16, 256, device="cuda")), 8, path=nat.MP_PATH_FFT)
    small(torch.rand(1, 8192, device="cuda")); torch.cuda.synchronize()
    print("a one-segment plan was captured first")
import ctypes
nat.lib().mp_stream_pair_ratio.restype = ctypes.c_float
nat.lib().mp_stream_pair_ratio.argtypes = [ctypes.c_int, ctypes.c_int]
print("streams kept by the pool (seen to run side by side):", nat.init_streams())
print("spin ratio of internal stream pairs (1 = side by side, 2 = one after the other):",
      {f"s{a}{b}": round(float(nat.lib().mp_stream_pair_ratio(a, b)), 2) for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))})
if "ratios-only" in sys.argv:
    sys.exit(0)
print("default (two sub-batches)", rate(0), " one stream", rate(nat.MP_FLAG_NO_OVERLAP), flush=True)
for g in (3, 4, 2):
    nat.tune(nat.MP_TUNE_GROUPS, g)
    print(f"  {g} sub-batches:", rate(nat.MP_FLAG_OVERLAP), flush=True)

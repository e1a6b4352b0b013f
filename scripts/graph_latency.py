#!/usr/bin/env python3
"""Does replaying a captured hipGraph of one whole encode (mp_encode_f32 is capture-safe: no host
synchronisation, fork/join of its internal streams by events) beat issuing its launches one by one?
Small shapes, where launches rather than arithmetic set the time.  Usage: python scripts/graph_latency.py"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth


def med(fn, n=40):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts[5:]))


for (A, L, N, B, K) in ((16, 256, 8192, 1, 8), (512, 512, 32768, 1, 64), (512, 512, 32768, 8, 64), (512, 512, 32768, 64, 64)):
    d = synth.make_dictionary(A, L, seed=100)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=24, seed=101)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    path = nat.MP_PATH_FFT
    ref = nat.encode(x, du, K, path=path)
    t_plain = med(lambda: nat.encode(x, du, K, path=path))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nat.encode(x, du, K, path=path)  # warm-up on the capture stream (function attributes, stream pool)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = nat.encode(x, du, K, path=path)
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    t_graph = med(g.replay)
    print(f"A{A} L{L} N{N} B{B} K{K}: launches one by one {t_plain:8.1f} us, graph replay {t_graph:8.1f} us "
          f"({t_plain / t_graph:.2f}x), identical results: {same}", flush=True)

#!/usr/bin/env python3
"""Segments with 16384 < cells <= 65536: which form of the select step wins (default heuristic vs forced forms)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
NO = nat.MP_FLAG_NO_OVERLAP
for (A, L, N, B, K) in ((1024, 512, 65536, 32, 32), (2048, 256, 32768, 32, 32), (512, 1024, 131072, 32, 32), (1024, 512, 65536, 64, 32)):
    d = synth.make_dictionary(A, L, seed=1000)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=96, seed=1002)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    ref = None
    for name, flags in (("default", 0), ("one_stream", NO), ("fused_one_stream", nat.MP_FLAG_FFT_FUSED | NO),
                        ("unfused_one_stream", nat.MP_FLAG_FFT_UNFUSED | NO)):
        ts = []
        for r in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = nat.encode(x, du, K, path=1, flags=flags, want_residual=False); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        if ref is None: ref = out
        same = all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3]))
        print(f"A{A} L{L} N{N} B{B} cells/seg {(N//64)*(A//32)}: {name:20s} {np.median(ts[1:]):8.3f} ms  same={same} nan={bool(torch.isnan(out[2]).any())}", flush=True)

#!/usr/bin/env python3
"""Timing/parity probe at the BASELINE configs[3] shape (4096x2048 dictionary, 131072-sample segments)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat  # noqa: E402
from mpcore import synth  # noqa: E402

A, L, N = 4096, 2048, 131072
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
d = synth.make_dictionary(A, L, seed=4000)
x = synth.make_segments(B, N, d, n_events=48, seed=4001)
print(f"inputs in {time.time() - t0:.1f}s", flush=True)
xd = torch.from_numpy(x).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
nat.profile_enable(True)
for path, name in ((2, "incremental"), (0, "direct")):
    k = K if path == 2 else 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = nat.encode(xd, du, k, path=path)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = nat.profile_read()
    print(name, f"B{B} K{k}: {dt*1e3:.1f} ms", {q: (round(v[0] / max(v[1], 1), 3), v[1]) for q, v in prof.items()}, flush=True)
    if path == 2:
        inc = out
    else:
        print("direct == incremental:", all(torch.equal(a[:, :k] if a.dim() == 2 and a.shape[1] == K else a, b) for a, b in zip(inc[:3], out[:3])))
atom, lag, gain, res = inc
rec = torch.zeros_like(xd)
nat.scatter(atom, torch.arange(B, device="cuda")[:, None].expand(B, K), lag, gain, du, rec)
print("roundtrip err", (rec + res - xd).abs().max().item(), "gain[0]", gain[0, :4].tolist())
full_flop = 2.0 * A * L * N * B
print("full pass TFLOP/s:", full_flop / (prof["corr_full"][0] / max(prof["corr_full"][1], 1) * 1e-3) / 1e12)

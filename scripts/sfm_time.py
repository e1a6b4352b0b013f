#!/usr/bin/env python3
"""sparse_coding_loss / sparse_feature_map (modules/matchingpursuit.py:68-146) at the headline dictionary:
forward only (events -> COO maps) and with the reference's gradient w.r.t. the reconstruction."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp
from mpcore import synth
A, L, N, B, K = 512, 512, 32768, 4, 16
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=48, seed=1002)).cuda()
y = (x + 0.05 * torch.randn_like(x))
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
with torch.no_grad():
    t_fwd = timed(lambda: mp.sparse_coding_loss(y, x, d, n_steps=K))
def with_grad():
    yy = y.clone().requires_grad_(True)
    mp.sparse_coding_loss(yy, x, d, n_steps=K).backward()
t_bwd = timed(with_grad)
print(f"sparse_coding_loss A{A} L{L} N{N} B{B} K{K}: forward only {t_fwd:.1f} ms, forward + backward {t_bwd:.1f} ms "
      f"({2 * B * K * 2 * A * L * N / (t_bwd * 1e-3) / 1e12:.1f} TFLOP/s of the two dense correlations per step the gradient needs)")

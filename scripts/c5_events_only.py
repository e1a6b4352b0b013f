#!/usr/bin/env python3
"""BASELINE configs[4] (512 x 512 atoms, 8 x 32768 samples, K = 32, STFT(2048, 256) iterative loss, Adam): twelve train steps on
the event form of the loss, for a kernel trace of ONE step's launches (rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import synth
from mpcore.model import MatchingPursuit, train_step
A, L, N, B, K = 512, 512, 32768, 8, 32
torch.manual_seed(0)
model = MatchingPursuit(A, L, N, K).cuda()
d = synth.make_dictionary(A, L, seed=5000)
with torch.no_grad():
    model.atoms.copy_(torch.from_numpy(d)[None].cuda() * 0.05)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=96, seed=5001)).cuda()[:, None, :]
ts = []
for it in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(model, opt, x, ("stft", 2048, 256))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"train step: median {np.median(ts[2:]):.2f} ms (a device synchronisation after every step)", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(20):
    train_step(model, opt, x, ("stft", 2048, 256))
torch.cuda.synchronize()
print(f"train step: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms back to back (one synchronisation after twenty steps)", flush=True)

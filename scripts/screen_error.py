#!/usr/bin/env python3
"""How far is an fp32 FFT correlation from the exact fma-chain feature map, in units of ||window||_2?
(sizing of FFT_TAU; uses torch.fft on the device as a stand-in for the screen's transform)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
for (A, L, N, M) in ((512, 512, 32768, 2048), (256, 2048, 32768, 8192), (64, 8192, 98304, 32768)):
    d = synth.make_dictionary(A, L, seed=1000)
    x = torch.from_numpy(synth.make_segments(4, N, d, n_events=192, seed=1002)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    out = nat.encode(x, du, 32, path=nat.MP_PATH_INCREMENTAL)
    res = out[3]                                   # a mid-run residual
    fm = nat.feature_map(res, du)                  # exact chains [B, A, N]
    worst = 0.0
    for s in range(0, N - M, M // 2):
        win = res[:, s:s + M]
        X = torch.fft.rfft(win, dim=-1)
        D = torch.fft.rfft(torch.nn.functional.pad(du, (0, M - L)), dim=-1)
        approx = torch.fft.irfft(X[:, None, :] * torch.conj(D)[None], n=M, dim=-1)[..., : M - L + 1]
        exact = fm[:, :, s:s + M - L + 1]
        err = (approx - exact).abs().amax(dim=(1, 2)) / win.norm(dim=-1)
        worst = max(worst, float(err.max()))
    print(f"A{A} L{L} M{M}: max |fft - chain| / ||window|| = {worst:.3e}", flush=True)

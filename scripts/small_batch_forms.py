#!/usr/bin/env python3
"""Small batches: the persistent form (with and without the coherence table) against the launch-per-step forms, planted and
noise inputs, k segment-iterations/s.   python scripts/small_batch_forms.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
K = 32
def rate(x, du, flags, co):
    for _ in range(3):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags, coherence=co)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags, coherence=co)
    torch.cuda.synchronize()
    return x.shape[0] * K * 6 / (time.perf_counter() - t0) / 1e3, nat.last_schedule()
for A, L, N in ((1024, 1024, 4096), (1024, 1024, 32768), (512, 1024, 32768), (256, 1024, 16384), (1024, 512, 2048), (1024, 512, 32768), (512, 700, 32768)):
    dn = synth.make_dictionary(A, L, seed=N + A)
    du = nat.unit_norm(torch.from_numpy(dn).cuda())
    mu = nat.coherence_table(du)
    for B in (1, 4, 8, 16, 32, 64):
        for kind in ("planted", "noise"):
            xh = synth.make_segments(B, N, dn, n_events=3 * K, seed=5) if kind == "planted" else np.random.default_rng(B).standard_normal((B, N)).astype(np.float32)
            x = torch.from_numpy(xh).cuda()
            r = [rate(x, du, nat.MP_FLAG_FFT_PERSISTENT, mu), rate(x, du, nat.MP_FLAG_FFT_PERSISTENT, False),
                 rate(x, du, nat.MP_FLAG_FFT_NO_PERSISTENT, False), rate(x, du, nat.MP_FLAG_FFT_FUSED, mu), rate(x, du, 0, None)]
            print(f"{A} x {L}, N {N}, B {B:2d}, {kind:7s}: persistent+table {r[0][0]:7.1f}  persistent {r[1][0]:7.1f}  per step {r[2][0]:7.1f} [{r[2][1]}]  "
                  f"per step fused+table {r[3][0]:7.1f}  default {r[4][0]:7.1f} [{r[4][1]}]", flush=True)

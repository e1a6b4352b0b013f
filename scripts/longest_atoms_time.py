#!/usr/bin/env python3
"""experiments/archive/e_2023_12_18/experiment.py:22-24's dictionary (2048 atoms of 16384 samples, 2^15-sample segments,
32 steps): MP_PATH_FFT (2^16-point transforms as four 2^14-point quarters) against MP_PATH_INCREMENTAL (the MFMA
correlation the default fell back to before the quarters existed); same events checked.  argv: B (default 4), K (32)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N = 2048, 16384, 32768
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
which = sys.argv[3] if len(sys.argv) > 3 else "both"
g = torch.Generator(device="cuda").manual_seed(5)
du = nat.unit_norm(torch.rand(A, L, device="cuda", generator=g) * 2 - 1)   # uniform(-1, 1), as the experiment's
x = torch.randn(B, N, device="cuda", generator=g) * 0.1
idx = torch.randint(0, A, (B, 8), device="cuda", generator=g)
for b in range(B):                                    # a few planted atoms on noise
    for j in range(8):
        p = int(torch.randint(0, N - L, (1,), generator=torch.Generator().manual_seed(b * 8 + j)))
        x[b, p:p + L] += du[idx[b, j]] * (1.0 + j)
outs = {}
for name, path, flags in (("fft (four quarters)", nat.MP_PATH_FFT, 0), ("fft simple screen", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_SIMPLE),
                          ("incremental (MFMA)", nat.MP_PATH_INCREMENTAL, 0)):
    if which != "both" and not name.startswith(which):
        continue
    out = nat.encode(x, du, K, path=path, flags=flags); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=path, flags=flags); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs[name] = [t.cpu() for t in out]
    nan = int(torch.isnan(out[2]).any(dim=1).sum())
    print(f"{A}x{L}, {B} x {N}, K={K}  {name}: {dt*1e3:.1f} ms = {B*K/dt:.0f} seg-it/s  (segments marked overflow: {nan}; schedule {nat.last_schedule()})", flush=True)
names = list(outs)
for n in names[1:]:
    a, b = outs[names[0]], outs[n]
    ok = ~torch.isnan(a[2]).any(dim=1) & ~torch.isnan(b[2]).any(dim=1)
    print(f"{names[0]} == {n}: atoms {bool((a[0][ok] == b[0][ok]).all())}, lags {bool((a[1][ok] == b[1][ok]).all())}, gains {bool((a[2][ok] == b[2][ok]).all())}  ({int(ok.sum())} segments compared)")

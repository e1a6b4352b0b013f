#!/usr/bin/env python3
"""The lazy screen of the launch-per-step form on a mid-size shape (2048 x 512 dictionary, 64 x 32768): time, tiles
skipped and contender cells refined per select against margin and cap."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
shapes = [(2048, 512, 32768, 64, 64), (4096, 2048, 131072, 32, 64)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:6])]
FLAGS = int(os.environ.get("MID_FLAGS", nat.MP_FLAG_NO_OVERLAP))   # 0: the library's choice (sub-batches from 48 segments)
for A, L, N, B, K in shapes:
    d = synth.make_dictionary(A, L, seed=A + L)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=7)).cuda()
    mu = nat.coherence_table(du)
    print(f"{A}x{L}, {B} x {N}, K={K}: coherence min {float(mu.min()):.3f} mean {float(mu.mean()):.3f}", flush=True)
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)
    for margin, reuse, co in ((0, 0, False), (0.5, 4, mu), (0.6, 4, mu), (0.7, 4, mu), (0.7, 1, mu), (0.85, 4, mu), (0.85, 2, mu), (1.0, 4, mu)):
        nat.tune(nat.MP_TUNE_LAZY_MARGIN, margin); nat.tune(nat.MP_TUNE_LAZY_REUSE, reuse)
        f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=FLAGS, coherence=co)
        out = f(); torch.cuda.synchronize(); nat.lazy_stats()
        nat.profile_enable(1); nat.profile_read()
        out = f(); torch.cuda.synchronize()
        p = nat.profile_read(); ls = nat.lazy_stats(); nat.profile_enable(0)
        t0 = time.perf_counter()
        for _ in range(3): out = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print(f"  margin {margin} cap {reuse} table {co is not False}: {dt * 1e3:7.2f} ms; screen {p['corr_inc'][0] / max(p['corr_inc'][1], 1) * 1e3:6.1f} us, "
              f"select {p['select'][0] / max(p['select'][1], 1) * 1e3:6.1f} us; skipped {ls['skipped']}/{ls['decided']}, contender cells per select "
              f"{ls['contender_cells'] / (B * K):.2f}, off: contenders {ls['off_contenders']} floor {ls['off_no_floor']}; schedule {nat.last_schedule()}, identical {same}, marked {int(torch.isnan(out[2]).any(dim=1).sum())}", flush=True)
    nat.tune(nat.MP_TUNE_LAZY_MARGIN, 0); nat.tune(nat.MP_TUNE_LAZY_REUSE, 0)

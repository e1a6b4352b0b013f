#!/usr/bin/env python3
"""Large and awkward batches through the persistent form against the launch-per-step form, bit for bit (not collected by
pytest; run by hand:  python tests/stress_persistent.py).  Many more segments than select workers, long runs (K = 256),
batches whose window records approach the 2 GiB cap (and one beyond it, which must fall back), odd atom counts."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
bad = 0
for A, L, N, B, K in ((512, 512, 32768, 300, 64), (512, 512, 32768, 1000, 32), (100, 256, 9000, 777, 20), (512, 512, 32768, 64, 256),
                      (77, 1024, 20000, 130, 40), (512, 512, 32768, 2100, 64), (33, 300, 4096, 4000, 12), (512, 512, 16384, 25, 200)):
    d = synth.make_dictionary(A, L, seed=7 + A)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 96), seed=11 + B)).cuda()
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sched = nat.last_schedule()
    st = nat.persist_stats() if sched == -1 else None
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_NO_PERSISTENT)
    torch.cuda.synchronize()
    keep = ~(torch.isnan(out[2]).any(dim=1) | torch.isnan(ref[2]).any(dim=1))
    same = all(torch.equal(p[keep], q[keep]) for p, q in zip(out, ref))
    marked = int((~keep).sum())
    ok = same and (st is None or (st["error"] == 0 and st["finished"] == B and st["selects"] == B * (K - 1)))
    bad += not ok
    print(f"A{A} L{L} N{N} B{B} K{K}: schedule {sched} {dt * 1e3:8.1f} ms {B * K / dt / 1e3:7.0f} k seg-it/s, identical {same}, "
          f"marked {marked}, stats {None if st is None else (st['error'], st['finished'], st['selects'])} {'OK' if ok else 'FAIL'}", flush=True)
print("stress:", "OK" if not bad else f"{bad} FAILURES", flush=True)
sys.exit(1 if bad else 0)

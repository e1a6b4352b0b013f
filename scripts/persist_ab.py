#!/usr/bin/env python3
"""A/B of one mp_tune knob on the persistent form at the headline dictionary, alternating the two settings so that box
and clock drift cancel: python scripts/persist_ab.py KEY [B ...]   (KEY: the MP_TUNE_* number; values 0 and 1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
key = int(sys.argv[1])
Bs = [int(v) for v in sys.argv[2:]] or [32, 64, 128]
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
mu = nat.coherence_table(du)
for B in Bs:
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)
    torch.cuda.synchronize()
    for co_name, co in (("lazy", mu), ("plain", False)):
        times = {0: [], 1: []}
        stats = {}
        for rep in range(6):
            for v in (0, 1):
                nat.tune(key, v)
                out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=co); torch.cuda.synchronize()
                assert all(torch.equal(p, q) for p, q in zip(out, ref)), (B, co_name, v)
                t0 = time.perf_counter()
                for _ in range(8):
                    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=co)
                torch.cuda.synchronize()
                times[v].append((time.perf_counter() - t0) / 8)
                st = nat.persist_stats()
                stats[v] = (st["task_ticks"] / max(st["tasks"], 1) / 100.0, st["select_ticks"] / max(st["selects"], 1) / 100.0, st["error"])
        nat.tune(key, 0)
        m0, m1 = float(np.median(times[0])), float(np.median(times[1]))
        print(f"B{B:4d} {co_name:5s}: knob {key} = 0: {m0 * 1e3:6.3f} ms ({B * K / m0 / 1e3:6.0f} k, task {stats[0][0]:.2f} us, select {stats[0][1]:.2f} us) | "
              f"= 1: {m1 * 1e3:6.3f} ms ({B * K / m1 / 1e3:6.0f} k, task {stats[1][0]:.2f} us, select {stats[1][1]:.2f} us) | {100 * (m0 / m1 - 1):+.1f} %; identical; errors {stats[0][2]} {stats[1][2]}", flush=True)

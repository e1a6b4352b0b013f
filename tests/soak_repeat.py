#!/usr/bin/env python3
"""Repeat encodes on every schedule and compare each repetition bit for bit with the first one-stream result (not
collected by pytest; run by hand:  python tests/soak_repeat.py [reps]).  A hazard that depends on timing shows up as a
different pick in a different segment every few repetitions, not in a single run."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for A, L, N, B, K in ((512, 256, 32768, 64, 32), (256, 1024, 32768, 64, 32), (512, 512, 32768, 128, 64), (100, 300, 9000, 50, 20),
                      (64, 2048, 30000, 24, 16), (1024, 512, 16384, 40, 24)):
    d = synth.make_dictionary(A, L, seed=3 + A)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 96), seed=5 + B)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP)
    torch.cuda.synchronize()
    for name, path, flags in (("default", nat.MP_PATH_FFT, 0), ("default again", nat.MP_PATH_FFT, 0), ("one stream", nat.MP_PATH_FFT, nat.MP_FLAG_NO_OVERLAP),
                              ("sub-batches", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_NO_PERSISTENT),
                              ("quarter", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_QUARTER | nat.MP_FLAG_NO_OVERLAP),
                              ("incremental", nat.MP_PATH_INCREMENTAL, 0)):
        n = reps if path == nat.MP_PATH_FFT else max(reps // 10, 2)
        miss = 0
        for _ in range(n):
            out = nat.encode(x, du, K, path=path, flags=flags)
            miss += not all(torch.equal(p, q) for p, q in zip(out, ref))
        sched = nat.last_schedule()
        bad += miss
        print(f"A{A} L{L} N{N} B{B} K{K} {name:13s} (schedule {sched:2d}, lazy screen skipped {nat.persist_stats()['skipped'] if sched == -1 else 0:6d} tasks in the "
              f"last one): {miss} of {n} repetitions differ", flush=True)
print("soak:", "OK" if not bad else f"{bad} MISMATCHES", flush=True)
sys.exit(1 if bad else 0)

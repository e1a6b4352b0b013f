#!/usr/bin/env python3
"""Small batches on the persistent form (headline dictionary, 64 steps): time per step, task and select times inside the
launch -- how much of a step is the screen's span when every task has a workgroup to itself?   python scripts/small_batch_steps.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
if len(sys.argv) > 1:   # 1: one slot per tile quarter (round 2's tasks), 2: two slots (finer tasks), default: by load
    nat.tune(nat.MP_TUNE_PERSIST_FINE, int(sys.argv[1]))
    print("MP_TUNE_PERSIST_FINE", sys.argv[1], flush=True)
dn = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(dn).cuda())
for B in (1, 2, 4, 8, 16, 32):
    x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()
    for _ in range(4):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    nat.tune(nat.MP_TUNE_AUDIT, 2); nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
    st = nat.persist_stats(); nat.tune(nat.MP_TUNE_AUDIT, 0)
    print(f"B {B:3d}: {dt * 1e3:.3f} ms per encode = {dt / K * 1e6:.1f} us per step, {B * K / dt / 1e3:.0f} k seg-it/s; schedule {nat.last_schedule()}; "
          f"task {st['task_ticks'] / max(st['tasks'], 1) / 100:.2f} us x {st['tasks']}, select {st['select_ticks'] / max(st['selects'], 1) / 100:.2f} us, skipped {st['skipped']}"
          if 'task_ticks' in st else f"B {B}: {dt * 1e3:.3f} ms {st}", flush=True)

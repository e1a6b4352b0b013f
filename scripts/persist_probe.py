#!/usr/bin/env python3
"""Persistent schedule: per-task / per-select times for a few batch sizes (debug)."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
A, L, N, K = 512, 512, 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
shard_list = [int(v) for v in sys.argv[1:]] or [0]   # arguments: shard counts (0 = the library's choice)
for B, shards in [(b, sh) for b in (32, 64, 128, 256) for sh in shard_list]:
    nat.tune(nat.MP_TUNE_PERSIST_SHARDS, abs(shards))
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    st = nat.persist_stats()
    print(f"B{B:3d} shards {shards:3d}: one {dt1 * 1e3:6.2f} ms, 8 in a row {dt * 1e3:6.2f} ms = {B * K / dt:9.0f} seg-it/s | per task {st['task_ticks'] / max(st['tasks'], 1) / 100:7.1f} us, per select "
          f"{st['select_ticks'] / max(st['selects'], 1) / 100:7.1f} us {st['select_phase_us']}, between tasks {st['idle_ticks'] / max(st['tasks'], 1) / 100:5.1f} us per task, polls {st['polls']}", flush=True)

#!/usr/bin/env python3
"""The multiband model's bands (1024 atoms of N / 4 samples, 8 segments of N samples, 32 steps), noise input: the persistent
form against the launch-per-step forms, with the persistent form's statistics.   python scripts/band_forms.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
B, K, A = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 32, 1024
for N in (512, 1024, 2048, 4096, 8192):
    L = N // 4
    dn = synth.make_dictionary(A, L, seed=N)
    du = nat.unit_norm(torch.from_numpy(dn).cuda())
    x = torch.from_numpy(np.random.default_rng(N).standard_normal((B, N)).astype(np.float32)).cuda()
    row = []
    for name, flags in (("default", 0), ("persistent", nat.MP_FLAG_FFT_PERSISTENT), ("launch per step", nat.MP_FLAG_FFT_NO_PERSISTENT),
                        ("one stream", nat.MP_FLAG_NO_OVERLAP), ("fused", nat.MP_FLAG_FFT_FUSED)):
        try:
            for _ in range(3):
                out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            row.append(f"{name} {dt * 1e3:.2f} ms [{nat.last_schedule()}]")
        except Exception as e:  # noqa: BLE001
            row.append(f"{name}: {str(e)[:40]}")
    nat.tune(nat.MP_TUNE_AUDIT, 2); nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize(); st = nat.persist_stats(); nat.tune(nat.MP_TUNE_AUDIT, 0)
    print(f"N {N} L {L}: " + "; ".join(row) + f"; task {st['task_ticks'] / max(st['tasks'], 1) / 100:.1f} us x {st['tasks']}, select {st['select_ticks'] / max(st['selects'], 1) / 100:.1f} us, phases {st['select_phase_us']}", flush=True)

#!/usr/bin/env python3
"""Randomised sweep of the lazy screen against the launch-per-step form (itself bit-identical to the oracle: the parity
suite), on runs long enough for widened bounds to matter: many steps, random margins, duplicated atoms (coherence 1
between tiles), periodic trains, events at the segment's end (cropped atoms).  Not collected by pytest:
    python tests/fuzz_lazy.py [n_cases] [seed]
FORM=fused in the environment: the lazy screen of the LAUNCH-PER-STEP form instead (fused whole-cell select + tile mask:
segments of more than 16384 cells, MP_FLAG_FFT_FUSED), against the same form without the table."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
if os.environ.get('LAZY_RADIUS'): nat.tune(nat.MP_TUNE_LAZY_RADIUS, int(os.environ['LAZY_RADIUS']))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad = skipped = run = marks_lazy = marks_ref = 0
FUSED = os.environ.get("FORM") == "fused"
reuse = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nat.tune(nat.MP_TUNE_LAZY_REUSE, reuse)
try:
    for case in range(n_cases):
        A = int(rng.integers(33, 260)); L = int(rng.choice([256, 300, 400, 512, 700, 1000, 1300]))
        N = int(rng.integers(3 * L, 24000)); B = int(rng.choice([24, 25, 31, 40, 64])); K = int(rng.integers(8, 48))
        if FUSED:   # more than 16384 cells per segment
            N = int(rng.integers(64 * (16384 // ((A + 31) // 32) + 2), 64 * (16384 // ((A + 31) // 32) + 2) + 40000)); B = int(rng.choice([3, 5, 8]))
        d = synth.make_dictionary(A, L, seed=100 + case)
        if case % 5 == 1:
            d[A // 2:] = d[: A - A // 2]                      # duplicated atoms
        x = synth.make_segments(B, N, d, n_events=int(rng.integers(4, 3 * K)), seed=200 + case)
        if case % 5 == 2:                                     # a periodic train of one atom
            x = np.zeros((B, N), dtype=np.float32)
            for p0 in range(0, N - L, L):
                x[:, p0:p0 + L] += d[case % A] / np.linalg.norm(d[case % A])
        if case % 5 == 3:                                     # energy at the very end: cropped atoms get selected
            x[:, -L // 2:] += (rng.standard_normal((B, L // 2)) * 3).astype(np.float32)
        du = nat.unit_norm(torch.from_numpy(d).cuda())
        xd = torch.from_numpy(x).cuda()
        if not nat.lib().mp_coherence_workspace_bytes(A, L):
            continue
        mu = nat.coherence_table(du)
        ref = nat.encode(xd, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)
        margin = float(rng.choice([0.7, 0.7, 0.4])) if len(sys.argv) > 3 else float(rng.choice([1.0, 0.7, 0.4]))
        if os.environ.get('LAZY_MARGIN'): margin = float(os.environ['LAZY_MARGIN'])
        nat.tune(nat.MP_TUNE_LAZY_MARGIN, margin)
        if FUSED:
            nat.lazy_stats()
            out = nat.encode(xd, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_FUSED, coherence=mu)
            torch.cuda.synchronize()
            ls = nat.lazy_stats()
            st = dict(error=0, skipped=ls["skipped"], tasks=ls["decided"] - ls["skipped"], lazy=ls)
            keep = ~(torch.isnan(out[2]).any(dim=1) | torch.isnan(ref[2]).any(dim=1))
            same = all(torch.equal(p[keep], q[keep]) for p, q in zip(out, ref)) and nat.last_schedule() == 1 and ls["decided"] > 0
        else:
            out = nat.encode(xd, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT, coherence=mu)
            torch.cuda.synchronize()
            st = nat.persist_stats()
            keep = ~(torch.isnan(out[2]).any(dim=1) | torch.isnan(ref[2]).any(dim=1))
            same = all(torch.equal(p[keep], q[keep]) for p, q in zip(out, ref)) and st["error"] == 0 and nat.last_schedule() == -1
        skipped += st["skipped"]; run += st["tasks"]
        ml, mr = int(torch.isnan(out[2]).any(dim=1).sum()), int(torch.isnan(ref[2]).any(dim=1).sum())
        marks_lazy += ml; marks_ref += mr
        if ml != mr and len(sys.argv) > 4:
            print(f"   case {case} (kind {case % 5}): A{A} L{L} N{N} B{B} K{K} margin {margin}: marked {ml} lazy / {mr} without; skipped {st['skipped']} of {st['skipped'] + st['tasks']}", flush=True)
        if not same:
            bad += 1
            print(f"MISMATCH case {case}: A{A} L{L} N{N} B{B} K{K} margin {margin} stats {st}", flush=True)
        if case % 10 == 9:
            print(f"{case + 1} cases, {bad} mismatches, {skipped} of {skipped + run} tasks skipped so far", flush=True)
finally:
    nat.tune(nat.MP_TUNE_LAZY_MARGIN, 0)
    nat.tune(nat.MP_TUNE_LAZY_REUSE, 0)
print(f"segments marked: {marks_lazy} with the lazy screen (reuse {reuse}), {marks_ref} without", flush=True)
print("lazy fuzz:", "OK" if not bad else f"{bad} MISMATCHES", f"({skipped} of {skipped + run} screen tasks answered without a transform)", flush=True)
sys.exit(1 if bad else 0)

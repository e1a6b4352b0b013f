#!/usr/bin/env python3
"""The persistent schedule (MP_FLAG_FFT_PERSISTENT) against the launch-per-step schedules: bitwise, then timing."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth

def check(A, L, N, B, K, seed=1):
    d = synth.make_dictionary(A, L, seed=seed)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 192), seed=seed + 1)).cuda()
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP)
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_FFT_PERSISTENT)
    torch.cuda.synchronize()
    same = [bool(torch.equal(a, b)) for a, b in zip(out, ref)]
    nan = int(torch.isnan(out[2]).any(dim=1).sum())
    print(f"A{A} L{L} N{N} B{B} K{K}: same {same} marked {nan}", flush=True)
    if not all(same):
        bad = (out[0] != ref[0]) | (out[1] != ref[1])
        idx = torch.nonzero(bad)
        print("  first mismatches (segment, step):", idx[:5].tolist(), flush=True)
    return all(same)

ok = True
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "small"):
    ok &= check(64, 128, 4096, 3, 6)
    ok &= check(40, 300, 6000, 5, 8)
    ok &= check(96, 512, 12000, 8, 12)
    ok &= check(512, 512, 32768, 16, 32)
    ok &= check(200, 200, 9000, 33, 10)
    print("persistent schedule bitwise:", "OK" if ok else "MISMATCH", flush=True)
if mode in ("all", "time") and ok:
    A, L, N, K = 512, 512, 32768, 64
    d = synth.make_dictionary(A, L, seed=1000)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    for B in (64, 128, 256):
        x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
        ref = None
        for name, flags in (("one stream", nat.MP_FLAG_NO_OVERLAP), ("sub-batches (default)", 0), ("persistent", nat.MP_FLAG_FFT_PERSISTENT)):
            f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
            out = f(); f(); torch.cuda.synchronize(); ts = []
            for _ in range(9):
                t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            if ref is None: ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            print(f"B{B:4d} {name:24s} {float(np.median(ts)) * 1e3:8.3f} ms {B * K / float(np.median(ts)):9.0f} seg-it/s  same={same}", flush=True)
            if name == "persistent":
                st = nat.persist_stats()
                print("      ", st, f"| per task {st['task_ticks'] / max(st['tasks'], 1) / 100:.1f} us, per select "
                      f"{st['select_ticks'] / max(st['selects'], 1) / 100:.1f} us, idle per workgroup "
                      f"{st['idle_ticks'] / 2048 / 100:.0f} us", flush=True)
sys.exit(0 if ok else 1)

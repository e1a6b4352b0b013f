#!/usr/bin/env python3
"""GPU bring-up diagnostics: compares every kernel variant with the CPU oracle and prints WHERE
they differ (never asserts), so one GPU call is enough to localise a layout or indexing bug."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import mp_oracle  # noqa: E402
from mpcore import _native as nat  # noqa: E402
from mpcore import synth  # noqa: E402

DEV = "cuda:0"
print("device:", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName, flush=True)


def report_fm(tag, got, want):
    bad = got != want
    n = int(bad.sum())
    print(f"[fm {tag}] shape {got.shape} mismatches {n}/{got.size} max|d| {np.abs(got - want).max():.3e}")
    if n:
        idx = np.argwhere(bad)
        print("   first mismatches (b, atom, lag): got / want")
        for b, a, t in idx[:12]:
            print(f"   ({b},{a},{t}) {got[b, a, t]:+.7e} / {want[b, a, t]:+.7e}")
        print("   mismatching atoms:", np.unique(idx[:, 1])[:40], " lags mod 64:", np.unique(idx[:, 2] % 64)[:64])
        # is it a permutation problem?  check whether got[b,a,t] appears elsewhere in want
        b, a, t = idx[0]
        hits = np.argwhere(want[b] == got[b, a, t])
        print(f"   value at first mismatch found in want at (atom, lag): {hits[:6].tolist()}")


def report_enc(tag, got, want):
    atom, lag, gain, res = got
    ok_i = np.array_equal(atom, want["atom"]) and np.array_equal(lag, want["lag"])
    ok_g = np.array_equal(gain, want["gain"])
    ok_r = np.array_equal(res, want["residual"])
    print(f"[enc {tag}] indices {'OK' if ok_i else 'DIFF'} gains {'bitwise' if ok_g else 'max rel %.2e' % (np.abs(gain - want['gain']).max() / np.abs(want['gain']).max())}"
          f" residual {'bitwise' if ok_r else 'max %.2e' % np.abs(res - want['residual']).max()}")
    if not ok_i:
        bad = np.argwhere((atom != want["atom"]) | (lag != want["lag"]))
        for b, k in bad[:6]:
            print(f"   seg {b} step {k}: got (a{atom[b, k]}, t{lag[b, k]}, g{gain[b, k]:.6f}) want (a{want['atom'][b, k]}, t{want['lag'][b, k]}, g{want['gain'][b, k]:.6f})")


shapes = {
    "tiny": (3, 5, 17, 2, 4, 2, 14),
    "ragged": (24, 100, 1000, 2, 12, 8, 12),
    "mid": (64, 128, 4096, 3, 16, 12, 13),
    "k_chunks": (40, 1100, 3000, 2, 6, 4, 16),
}
for name, (A, L, N, B, K, n_ev, seed) in shapes.items():
    d = synth.make_dictionary(A, L, seed=seed)
    x = synth.make_segments(B, N, d, n_events=n_ev, seed=seed)
    du = mp_oracle.unit_norm(d)
    du_g = nat.unit_norm(torch.from_numpy(d).to(DEV)).cpu().numpy()
    print(f"== {name}: A{A} L{L} N{N} B{B} K{K}; unit_norm bitwise: {np.array_equal(du, du_g)}", flush=True)
    want_fm = mp_oracle.feature_map(x, du)
    got_fm = nat.feature_map(torch.from_numpy(x).to(DEV), torch.from_numpy(du).to(DEV)).cpu().numpy()
    report_fm(name, got_fm, want_fm)
    want = mp_oracle.encode(x, du, K)
    for tag, path, flags in [("naive", nat.MP_PATH_NAIVE, 0), ("direct", nat.MP_PATH_DIRECT, 0),
                             ("direct_nodma", nat.MP_PATH_DIRECT, nat.MP_FLAG_NO_DMA),
                             ("direct_ta64", nat.MP_PATH_DIRECT, nat.MP_FLAG_TA64),
                             ("incremental", nat.MP_PATH_INCREMENTAL, 0),
                             ("incremental_np", nat.MP_PATH_INCREMENTAL, 8),
                             ("incremental_np_nodma_ta64", nat.MP_PATH_INCREMENTAL, 8 | 1 | 4)]:
        try:
            print(f"   ... {name}/{tag}", flush=True)
            out = nat.encode(torch.from_numpy(x).to(DEV), torch.from_numpy(du).to(DEV), K, path=path, flags=flags)
            torch.cuda.synchronize()
            report_enc(f"{name}/{tag}", [t.cpu().numpy() for t in out], want)
        except Exception as e:  # noqa: BLE001
            print(f"[enc {name}/{tag}] EXCEPTION {type(e).__name__}: {e}")
    sys.stdout.flush()

# quick timing at the headline shape, all variants
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).to(DEV)
du = nat.unit_norm(torch.from_numpy(d).to(DEV))
ref = None
for tag, path, flags, k in [("incremental", 2, 0, K), ("incremental_nodma", 2, 1, K), ("incremental_ta64", 2, 4, K),
                            ("incremental_nostagger", 2, 16, K), ("incremental_nostagger_ta64", 2, 20, K),
                            ("incremental_np", 2, 8, K),
                            ("direct", 0, 0, 8), ("direct_nodma", 0, 1, 8), ("direct_ta64", 0, 4, 8),
                            ("direct_nostagger", 0, 16, 8), ("direct_nostagger_ta64", 0, 20, 8),
                            ("direct_np", 0, 8, 8)]:
    print(f"   ... timing {tag}", flush=True)
    nat.encode(x, du, 2, path=path, flags=flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = nat.encode(x, du, k, path=path, flags=flags)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ref is None:
        ref = out
    same = all(torch.equal(p[:, :k] if p.dim() == 2 and p.shape[1] == K else p, q[:, :k] if q.dim() == 2 and q.shape[1] == K else q)
               for p, q in zip(out[:3], ref[:3]))
    print(f"[time {tag}] B{B} K{k}: {dt * 1e3:.2f} ms -> {B * k / dt:.0f} seg-it/s; events == incremental: {same}", flush=True)

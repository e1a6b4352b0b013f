#!/usr/bin/env python3
"""Randomised parity sweep: the default schedule (and a few others) against the CPU oracle, bitwise, on shapes
no fixed test uses -- odd atom counts, atoms longer than the segment's tail, batches that split unevenly
into sub-batches, segments shorter than one transform.  Lives under tests/ because it uses the oracle (test
infrastructure); not collected by pytest -- run it by hand:   python tests/fuzz_parity.py [n_cases] [seed] [audit]
With a third argument the library runs in audit mode (mp_tune(MP_TUNE_AUDIT, 1)): after every FFT screen each
screened cell is recomputed exactly, and the sweep ends with the largest |screen - exact| / eps it saw."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import adversarial as adv
from mpcore import _native as nat
from mpcore import synth
import mp_oracle
mp_oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
AUDIT = len(sys.argv) > 3
if AUDIT:
    nat.tune(nat.MP_TUNE_AUDIT, 1)
    nat.audit_read()
audit_worst = dict(max_ratio=0.0, max_quarter_ratio=0.0, cells=0, over_bound=0)
paths = [("fft", nat.MP_PATH_FFT, 0), ("fft_one_stream", nat.MP_PATH_FFT, nat.MP_FLAG_NO_OVERLAP),
         ("fft_scan_refine", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_NO_QUARTER), ("fft_quarter", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_QUARTER), ("fft_fused", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_FUSED),
         ("fft_persistent", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_PERSISTENT), ("incremental", nat.MP_PATH_INCREMENTAL, 0)]
bad = 0
marked = 0
lazy_skipped = 0
for case in range(n_cases):
    A = int(rng.integers(1, 90)); L = int(rng.choice([5, 16, 33, 64, 100, 128, 250, 300, 512, 700, 1100]))
    N = int(rng.integers(max(L // 2, 40), 9000)); B = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 45, 70])); K = int(rng.integers(1, 10))
    if case % 10 == 1:   # degenerate sizes: one-sample atoms, segments shorter than a block, a single atom
        A = int(rng.choice([1, 2, 33])); L = int(rng.choice([1, 2, 3])); N = int(rng.integers(1, 70)); B = int(rng.choice([1, 3, 40]))
    if case % 10 == 2:   # many more steps than the segment has structure: the tail of the run is rounding noise
        A = int(rng.integers(2, 20)); L = int(rng.choice([8, 32])); N = int(rng.integers(100, 600)); B = 3; K = 40
    if case % 10 == 3:   # more than 16384 cells per segment: up to 65536 the persistent form with the fused whole-cell step 0
                         # (1024- to 4096-point transforms, the default since round 3), the launch-per-step fused select beyond / at short atoms
        A = int(rng.choice([1000, 1024, 1500])); L = int(rng.choice([32, 64, 600])); N = int(rng.integers(36000, 48000)); B = 2; K = 3
    if case % 10 == 4:   # atoms beyond 5398 samples: split transforms (two / four 2^14-point parts per 2^15- / 2^16-point transform)
        A = int(rng.integers(1, 10)); L = int(rng.choice([5399, 8192, 10859, 10860, 16384, 21782])); N = int(rng.integers(L // 2, 30000)); B = int(rng.choice([1, 2, 9])); K = 3
    d = synth.make_dictionary(A, L, seed=1000 + case)
    x = synth.make_segments(B, N, d, n_events=min(3 * K, 12), seed=5000 + case) if N > L else \
        rng.standard_normal((B, N)).astype(np.float32)
    if case % 10 == 6:   # duplicated atoms: exact ties between atoms, settled by the lower flat index
        d[A // 2:] = d[: A - A // 2]
    if case % 10 == 7 and N > 4 * L:   # a periodic train of one atom: many equal or near-equal maxima
        x = np.zeros((B, N), dtype=np.float32)
        for p0 in range(0, N - L, 2 * L):
            x[:, p0:p0 + L] += d[0] / np.linalg.norm(d[0])
    if case % 10 == 8:   # tiny and huge amplitudes
        x = (x * (1e-30 if case % 20 == 8 else 1e18)).astype(np.float32)
    if case % 10 == 5 and N > 2 * L:   # same-sign atoms on a DC offset (the chain's rounding at its most biased) / one 1e3 transient among 1e-4 samples
        if case % 20 == 5:
            d = adv.same_sign_dictionary(A, L, 3000 + case)
            x = adv.dc_offset_segments(B, N, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), 8, 3100 + case,
                                       dc=float(rng.choice([0.3, 30.0])))
        else:
            x = adv.transient_segments(B, N, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), 8, 3200 + case)
    du = mp_oracle.unit_norm(d)
    want = mp_oracle.encode(x, du, K)
    gap = (want["top2"][..., 0] - want["top2"][..., 1]) / np.maximum(np.abs(want["top2"][..., 0]), 1e-30)
    xd = torch.from_numpy(x).cuda(); dud = torch.from_numpy(du).cuda()
    mu = nat.coherence_table(dud) if nat.lib().mp_coherence_workspace_bytes(A, L) else None
    # (with the table: the lazy screen inside the persistent launch, and between launches -- the fused select's tile mask)
    for name, path, flags in paths + ([("fft_lazy", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_PERSISTENT),
                                       ("fft_fused_lazy", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_FUSED)] if mu is not None and K >= 2 else []):
        a, l, g, r = nat.encode(xd, dud, K, path=path, flags=flags, coherence=mu if name.endswith("lazy") else False)
        if name == "fft_lazy":
            lazy_skipped += nat.persist_stats()["skipped"]
        if name == "fft_fused_lazy":
            lazy_skipped += nat.lazy_stats()["skipped"]
        a, l, g, r = a.cpu().numpy(), l.cpu().numpy(), g.cpu().numpy(), r.cpu().numpy()
        nanrow = np.isnan(g).any(axis=1)
        marked += int(nanrow.sum())
        ok = True
        for b in range(B):
            if nanrow[b]:
                continue  # screen overflow, marked in-band: allowed (the caller re-encodes)
            if not (np.array_equal(a[b], want["atom"][b]) and np.array_equal(l[b], want["lag"][b]) and
                    np.array_equal(g[b], want["gain"][b]) and np.array_equal(r[b], want["residual"][b])):
                ok = False
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} {name}: A{A} L{L} N{N} B{B} K{K} min top-2 gap {gap.min():.2e}", flush=True)
    if A * N <= 400000:  # the local-contrast-norm rule (mp_encode_lcn_f32; the oracle's box filter is a plain loop)
        wl = mp_oracle.encode_lcn(x, du, K)
        a, l, g, r = [t.cpu().numpy() for t in nat.encode_lcn(xd, dud, K)]
        if not (np.array_equal(a, wl["atom"]) and np.array_equal(l, wl["lag"]) and np.array_equal(g, wl["gain"]) and
                np.array_equal(r, wl["residual"])):
            bad += 1
            print(f"MISMATCH case {case} lcn: A{A} L{L} N{N} B{B} K{K}", flush=True)
    if AUDIT:
        a = nat.audit_read()
        audit_worst = dict(max_ratio=max(audit_worst["max_ratio"], a["max_ratio"]),
                           max_quarter_ratio=max(audit_worst["max_quarter_ratio"], a["max_quarter_ratio"]),
                           cells=audit_worst["cells"] + a["cells"], over_bound=audit_worst["over_bound"] + a["over_bound"])
        if a["over_bound"] or a["max_ratio"] > 0.25:
            print(f"AUDIT case {case}: A{A} L{L} N{N} B{B} K{K} {a}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases done, {bad} mismatches" + (f", audit so far {audit_worst}" if AUDIT else ""), flush=True)
# the convolution model of mp.py (raw atoms, v^2 update, no oracle): its three schedules must agree with each other
conv_bad = 0
for case in range(max(n_cases // 5, 4)):
    A = int(rng.integers(2, 70)); L = int(rng.choice([8, 32, 100, 128])); N = int(rng.integers(2 * L, 5000))
    B = int(rng.choice([1, 3, 33])); K = int(rng.integers(1, 6))
    atoms = (synth.make_dictionary(A, L, seed=9000 + case) * 0.2).astype(np.float32)
    x = synth.make_segments(B, N, synth.make_dictionary(A, L, seed=9000 + case), n_events=8, seed=9500 + case)
    xd = torch.from_numpy(x).cuda(); ad = torch.from_numpy(atoms).cuda()
    ref = nat.encode(xd, ad, K, path=nat.MP_PATH_DIRECT, conv_model=True)
    for name, path, flags in (("incremental", nat.MP_PATH_INCREMENTAL, 0), ("fft", nat.MP_PATH_FFT, 0),
                              ("fft_quarter", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_QUARTER), ("fft_fused", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_FUSED)):
        out = nat.encode(xd, ad, K, path=path, flags=flags, conv_model=True)
        keep = ~torch.isnan(out[2]).any(dim=1)
        if not all(torch.equal(a[keep], b[keep]) for a, b in zip(out, ref)):
            conv_bad += 1
            print(f"CONV MISMATCH case {case} {name}: A{A} L{L} N{N} B{B} K{K}", flush=True)
print("convolution-model schedules agree:", "OK" if conv_bad == 0 else f"{conv_bad} MISMATCHES", flush=True)
bad += conv_bad
if AUDIT:
    nat.tune(nat.MP_TUNE_AUDIT, 0)
    print("screen audit: largest |screen - exact| / eps", audit_worst, flush=True)
    bad += audit_worst["over_bound"]
print("fuzz parity:", "OK" if bad == 0 else f"{bad} MISMATCHES", f"({marked} segment-runs marked as screen overflow; the lazy screen "
      f"answered {lazy_skipped} tasks without a transform)", flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Interleaved A/B of MP_PATH_FFT variants (and the incremental path) at the headline shape."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
shape = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
A, L, N, B, K = {"c2": (512, 512, 32768, 64, 64), "c4": (4096, 2048, 131072, 16, 16)}[shape]
# (name, path, flags, tau, screen pairs-per-slot override, sub-batches)
NO = nat.MP_FLAG_NO_OVERLAP
variants = [("fft", 1, 0, 2e-5, 0, 2), ("fft_groups3", 1, 0, 2e-5, 0, 3), ("fft_groups4", 1, 0, 2e-5, 0, 4),
            ("fft_one_stream", 1, NO, 2e-5, 0, 2), ("fft_quarter_one_stream", 1, NO | nat.MP_FLAG_FFT_QUARTER, 2e-5, 0, 2), ("fft_scan_refine_one_stream", 1, NO | nat.MP_FLAG_FFT_NO_QUARTER, 2e-5, 0, 2),
            ("fft_pps4", 1, 0, 2e-5, 4, 2), ("fft_pps8", 1, 0, 2e-5, 8, 2),
            ("fft_unfused", 1, nat.MP_FLAG_FFT_UNFUSED, 2e-5, 0, 2), ("fft_fused", 1, nat.MP_FLAG_FFT_FUSED, 2e-5, 0, 2),
            ("fft_fused_one_stream", 1, nat.MP_FLAG_FFT_FUSED | NO, 2e-5, 0, 2),
            ("fft_tau5e-6", 1, 0, 5e-6, 0, 2)]
if shape == "c2":
    variants += [("incremental", 2, 0, 2e-5, 0, 2)]
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=min(3 * K, 192), seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
ref = None
times = {v[0]: [] for v in variants}
profs = {}
nat.profile_enable(16)
for r in range(rounds + 1):
    for name, path, flags, tau, pps, groups in variants:
        nat.tune(nat.MP_TUNE_TAU, tau)
        nat.tune(nat.MP_TUNE_GROUPS, groups)
        nat.tune(nat.MP_TUNE_SCREEN_PPS, pps)
        torch.cuda.synchronize(); nat.profile_read()
        t0 = time.perf_counter()
        out = nat.encode(x, du, K, path=path, flags=flags, want_residual=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        p = nat.profile_read()
        if r > 0:
            times[name].append(dt); profs[name] = p
        if ref is None: ref = out
        elif r == 0:
            print(name, "== first variant:", all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3])), "nan", torch.isnan(out[2]).any().item(), flush=True)
nat.tune(nat.MP_TUNE_TAU, 2e-5); nat.tune(nat.MP_TUNE_SCREEN_PPS, 0); nat.tune(nat.MP_TUNE_GROUPS, 2)
for name, path, flags, tau, pps, groups in variants:
    t = np.array(times[name]); p = profs[name]
    print(f"{name:18s} median {np.median(t):8.3f} ms min {t.min():8.3f} -> {B*K/np.median(t)*1e3:9.0f} seg-it/s | full {p['corr_full'][0]/max(p['corr_full'][1],1):7.3f} ms inc {p['corr_inc'][0]/max(p['corr_inc'][1],1)*1e3:7.1f} us select {p['select'][0]/max(p['select'][1],1)*1e3:6.1f} us", flush=True)

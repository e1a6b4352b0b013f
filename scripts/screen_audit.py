#!/usr/bin/env python3
"""How much of its error bound does the FFT screen use?  (GPU)

Runs MP_PATH_FFT in audit mode (mp_tune(MP_TUNE_AUDIT, 1): after every screen each screened cell is recomputed
exactly, fft_audit_kernel) over random, planted, DC-offset / same-sign, transient and extreme-amplitude inputs for
atom lengths 16 .. 8192 and prints, per case, max |screen - exact| / eps and the same error in units of
2^-24 ||window|| -- the number the constant of the bound's transform term (FFT_TAU_C) was set from."""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from mpcore import _native as nat  # noqa: E402
from mpcore import synth  # noqa: E402
import adversarial as adv  # noqa: E402

U = 2.0 ** -24


def log2m(L):
    M, lg = 256, 8
    while M < 3 * L + 190:
        M, lg = M * 2, lg + 1
    return lg


def bound_per_unit_window(d_raw):
    """eps / ||window|| as the library forms it (mpcore.hip::fft_tau, max_row_norm_kernel), unit-normed atoms."""
    d = d_raw.astype(np.float64)
    d = d / (np.linalg.norm(d, axis=-1, keepdims=True) + 1e-8)
    L = d.shape[1]
    W = np.sqrt(((L - np.arange(L)) ** 2 * d * d).sum(-1))
    return float((1.0001 * (1.001 * W + 4.0 * log2m(L))).max() * U)


def run(name, x, d_raw, K, flags=0):
    du = nat.unit_norm(torch.from_numpy(d_raw).cuda())
    nat.audit_read()
    out = nat.encode(torch.from_numpy(x).cuda(), du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize()
    a = nat.audit_read()
    L = d_raw.shape[1]
    marked = int(torch.isnan(out[2]).any(dim=1).sum())
    bound = bound_per_unit_window(d_raw)
    err_u = a["max_ratio"] * bound / U
    print(f"{name:44s} L={L:5d} log2M={log2m(L):2d} cells={a['cells']:8d} ratio={a['max_ratio']:.4f} "
          f"quarter={a['max_quarter_ratio']:.4f} over={a['over_bound']} err={err_u:8.1f} u*||w||  "
          f"(bound {bound / U:7.0f} u; sqrt(L)={math.sqrt(L):.0f}) marked={marked}", flush=True)
    return a


def main():
    nat.tune(nat.MP_TUNE_AUDIT, 1)
    worst = 0.0
    for L, A, N, B in ((16, 40, 2000, 4), (128, 64, 4096, 4), (512, 96, 12000, 4), (2048, 32, 30000, 2), (8192, 8, 40000, 2)):
        K = 6 if L <= 512 else 4
        d = synth.make_dictionary(A, L, seed=70 + L)
        du = d / (np.linalg.norm(d, axis=-1, keepdims=True) + 1e-8)
        cases = [
            ("planted", synth.make_segments(B, N, d, n_events=12, seed=71 + L), d),
            ("white noise", np.random.default_rng(72 + L).standard_normal((B, N)).astype(np.float32), d),
            ("planted x 1e-30", (synth.make_segments(B, N, d, n_events=12, seed=73 + L) * 1e-30).astype(np.float32), d),
            ("planted x 1e18", (synth.make_segments(B, N, d, n_events=12, seed=74 + L) * 1e18).astype(np.float32), d),
            ("transient 1e3 in 1e-4", adv.transient_segments(B, N, du.astype(np.float32), 8, 75 + L), d),
        ]
        ds = adv.same_sign_dictionary(A, L, 76 + L)
        dsu = (ds / np.linalg.norm(ds, axis=-1, keepdims=True)).astype(np.float32)
        cases += [
            ("DC 0.3 + events, same-sign atoms", adv.dc_offset_segments(B, N, dsu, 8, 77 + L), ds),
            ("DC 30 + events, same-sign atoms", adv.dc_offset_segments(B, N, dsu, 8, 78 + L, dc=30.0), ds),
            ("constant 1, same-sign atoms", np.ones((B, N), dtype=np.float32), ds),
            ("DC 0.3 + events, random atoms", adv.dc_offset_segments(B, N, du.astype(np.float32), 8, 79 + L), d),
        ]
        for name, x, dd in cases:
            for fname, flags in (("", 0), (" [quarter cells]", nat.MP_FLAG_FFT_QUARTER)):
                if fname and not 128 <= L <= 512:
                    continue
                a = run(name + fname, x, dd, K, flags)
                worst = max(worst, a["max_ratio"], a["max_quarter_ratio"])
    nat.tune(nat.MP_TUNE_AUDIT, 0)
    print(f"worst ratio over all cases: {worst:.4f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Bring-up of MP_PATH_FFT: transform accuracy vs numpy, encode parity vs the oracle, timing."""
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
rng = np.random.default_rng(0)
for lg in range(8, 15):
    M = 1 << lg
    x = (rng.standard_normal((3, M)) + 1j * rng.standard_normal((3, M))).astype(np.complex64)
    xd = torch.from_numpy(x).to(DEV)
    f = nat.fft_c2c(xd).cpu().numpy()
    fi = nat.fft_c2c(xd, inverse=True).cpu().numpy()
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    refi = np.fft.ifft(x.astype(np.complex128), axis=-1) * M
    sc = np.abs(ref).max()
    print(f"[fft M={M}] fwd rel err {np.abs(f - ref).max() / sc:.2e}  inv rel err {np.abs(fi - refi).max() / sc:.2e}", flush=True)

shapes = {
    "tiny": (3, 5, 17, 2, 4, 2, 14),
    "ragged": (24, 100, 1000, 2, 12, 8, 12),
    "mid": (64, 128, 4096, 3, 16, 12, 13),
    "k_chunks": (40, 1100, 3000, 2, 6, 4, 16),
    "c1": (16, 256, 8192, 1, 8, 6, 11),
    "many_atoms": (200, 32, 700, 1, 10, 6, 17),
    "atom_longer_than_segment": (5, 64, 50, 2, 3, 2, 15),
}
for name, (A, L, N, B, K, n_ev, seed) in shapes.items():
    d = synth.make_dictionary(A, L, seed=seed)
    x = synth.make_segments(B, N, d, n_events=n_ev, seed=seed)
    du = mp_oracle.unit_norm(d)
    want = mp_oracle.encode(x, du, K)
    print(f"   ... {name}", flush=True)
    try:
        out = nat.encode(torch.from_numpy(x).to(DEV), torch.from_numpy(du).to(DEV), K, path=nat.MP_PATH_FFT)
        torch.cuda.synchronize()
        atom, lag, gain, res = [t.cpu().numpy() for t in out]
        ok_i = np.array_equal(atom, want["atom"]) and np.array_equal(lag, want["lag"])
        print(f"[enc {name}/fft] indices {'OK' if ok_i else 'DIFF'} gains {'bitwise' if np.array_equal(gain, want['gain']) else 'DIFF'}"
              f" residual {'bitwise' if np.array_equal(res, want['residual']) else 'DIFF'} nan={np.isnan(gain).any()}", flush=True)
        if not ok_i:
            bad = np.argwhere((atom != want["atom"]) | (lag != want["lag"]))
            for b, k in bad[:6]:
                print(f"   seg {b} step {k}: got (a{atom[b, k]}, t{lag[b, k]}, g{gain[b, k]:.6f}) want (a{want['atom'][b, k]}, t{want['lag'][b, k]}, g{want['gain'][b, k]:.6f})")
    except Exception as e:  # noqa: BLE001
        print(f"[enc {name}/fft] EXCEPTION {type(e).__name__}: {e}", flush=True)

# headline shape: FFT vs incremental
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).to(DEV)
du = nat.unit_norm(torch.from_numpy(d).to(DEV))
ref = nat.encode(x, du, K, path=nat.MP_PATH_INCREMENTAL)
nat.profile_enable(True)
for rep in range(2):
    print("   ... timing fft c2", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = nat.profile_read()
    same = all(torch.equal(p, q) for p, q in zip(out, ref))
    print(f"[time fft c2] B{B} K{K}: {dt * 1e3:.2f} ms -> {B * K / dt:.0f} seg-it/s; == incremental: {same}; nan={torch.isnan(out[2]).any().item()}",
          {q: (round(v[0] / max(v[1], 1), 4), v[1]) for q, v in prof.items()}, flush=True)

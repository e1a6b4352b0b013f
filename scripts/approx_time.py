#!/usr/bin/env python3
"""fft_convolve(approx=slice): the band-limited residual + mp_feature_map_f32 route against the reference's own
formulation (A x (N+L) spectrum products, B x A inverse transforms on torch.fft), same device.  python scripts/approx_time.py"""
import os, sys, time
import torch
from torch.nn import functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp, synth, _native as nat

def reference_form(signal, atoms, approx):          # conv.py:11-29, 50-53 restated with torch.fft
    B, N = signal.shape[0], signal.shape[-1]
    A, L = atoms.shape
    sig_t = F.pad(signal, (0, L))
    padded = F.pad(atoms, (0, sig_t.shape[-1] - L))
    sig = torch.fft.rfft(sig_t, dim=-1)
    atom = torch.fft.rfft(torch.flip(padded, dims=(-1,)), dim=-1)[None, ...]
    spec = torch.zeros(B, A, sig.shape[-1], device=signal.device, dtype=sig.dtype)
    spec[..., approx] = sig[..., approx] * atom[..., approx]
    return torch.roll(torch.fft.irfft(spec, dim=-1), 1, dims=(-1,))[..., :N]

for A, L, N, B in ((512, 512, 32768, 16), (128, 256, 16384, 8)):
    d = torch.from_numpy(synth.make_dictionary(A, L, seed=3)).cuda()
    du = nat.unit_norm(d)
    x = torch.from_numpy(synth.make_segments(B, N, d.cpu().numpy(), n_events=64, seed=4)).cuda()[:, None, :]
    sl = slice(0, N // 8)
    for name, f in (("reference form (torch.fft)", lambda: reference_form(x, du, sl)), ("band-limited residual + mp_feature_map_f32", lambda: mp.fft_convolve(x, du, approx=sl))):
        out = f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): out = f()
        torch.cuda.synchronize()
        print(f"A{A} L{L} N{N} B{B} {name:45s}: {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms", flush=True)
        if name.startswith("ref"): want = out
    print(f"   max |difference| / max |map| = {float((out - want).abs().max() / want.abs().max()):.2e}", flush=True)

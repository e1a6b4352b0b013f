"""End-to-end timing of the multiband model (modules/multibanddict.py:282-473) at the reference's own
configuration (experiments/archive/e_2023_3_8/experiment.py:346-359: 1024 atoms per band, 7 bands 512 ... 32768,
atoms a quarter of the band, 32 steps): encode, recon (encode + decode + recompose) and learn, through the
drop-in Python surface.  Usage: python scripts/multiband_time.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore.multibanddict import BandSpec, MultibandDictionaryLearning  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, STEPS, A = 32768, 32, 1024
torch.manual_seed(0)
model = MultibandDictionaryLearning(
    [BandSpec(size, A, size // 4, device=DEV, signal_samples=N, is_lowest_band=(size == 512))
     for size in (512, 1024, 2048, 4096, 8192, 16384, 32768)], n_samples=N)
t = torch.linspace(0, 1, N, device=DEV)
batch = sum(torch.sin(2 * torch.pi * f * t[None, :] * (1 + 0.1 * torch.rand(B, 1, device=DEV))) * torch.exp(-3 * t)
            for f in (110, 220, 330, 1760, 5000))[:, None, :] + 0.01 * torch.randn(B, 1, N, device=DEV)
batch = batch / batch.abs().amax(dim=-1, keepdim=True)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


with torch.no_grad():
    t_enc = timed(lambda: model.encode(batch, STEPS))
    t_rec = timed(lambda: model.recon(batch, STEPS))
    t_learn = timed(lambda: model.learn(batch, STEPS))
    rec, events = model.recon(batch, STEPS)
    err = float(torch.norm(rec - batch) / torch.norm(batch))
n_ev = sum(len(e) for e in events.values())
print(f"multiband B{B} N{N} 7 bands x {A} atoms, {STEPS} steps: encode {t_enc * 1e3:.1f} ms, recon {t_rec * 1e3:.1f} ms, "
      f"learn {t_learn * 1e3:.1f} ms; {n_ev} events, relative residual {err:.3f}", flush=True)

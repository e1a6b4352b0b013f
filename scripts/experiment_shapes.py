#!/usr/bin/env python3
"""The drop-in surface at the shapes the reference's own experiments call it with (read from
/root/reference/experiments/archive/*/experiment.py; sizes only):
  e_2023_7_14   sparse_code(flatten=True) + scatter + dictionary_learning_step, 1024 x 512 dictionary, 2^15 samples, 512 steps
  e_2023_3_8    MultibandDictionaryLearning: seven bands 512 .. 32768 samples, 1024 atoms each of band / 4 samples, 32 steps
  e_2023_7_20   sparse_code + dictionary_learning_step, 512 x 512 dictionary, 2^15 samples (the headline dictionary)
Synthetic inputs (events of the dictionary planted on a noise bed, mpcore/synth.py).  Per case: wall time of the call,
segment-iterations/s, the schedule the library chose, segments marked.   python scripts/experiment_shapes.py [batch]"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import matchingpursuit as mp
from mpcore import multibanddict as mb
from mpcore import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SCHED = {-1: "persistent", 1: "launch per step, one stream", 2: "launch per step, 2 sub-batches", 4: "launch per step, 4 sub-batches"}


def timed(fn, reps=3):
    out = None
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, out


def single(name, A, L, N, K, n_events):
    dn = synth.make_dictionary(A, L, seed=77)
    d = torch.from_numpy(dn).cuda()
    x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=n_events, seed=78)).cuda()[:, None, :]
    mp.sparse_code(x, d, n_steps=K, flatten=True)   # (first use: workspaces, streams, the coherence table)
    dt, (events, scatter) = timed(lambda: mp.sparse_code(x, d, n_steps=K, flatten=True))
    sched = nat.last_schedule()
    dt_r, recon = timed(lambda: scatter(x.shape, events))
    dt_e, _ = timed(lambda: nat.encode(x[:, 0, :], nat.unit_norm(d), K, path=nat.MP_PATH_FFT))
    dt_d, _ = timed(lambda: mp.dictionary_learning_step(x, d, n_steps=K))
    err = float(((x - recon) ** 2).sum() / (x ** 2).sum())
    print(f"{name}: {A} x {L}, B {B} x {N}, K {K}: sparse_code(flatten) {dt * 1e3:.2f} ms = {B * K / dt / 1e3:.1f} k seg-it/s "
          f"[{SCHED.get(sched, sched)}]; bare encode {dt_e * 1e3:.2f} ms; scatter {dt_r * 1e3:.2f} ms; "
          f"dictionary_learning_step {dt_d * 1e3:.2f} ms; residual energy {err:.3f}", flush=True)


def multiband(steps=32, n_atoms=1024):
    n_samples = 2 ** 15
    sizes = [512, 1024, 2048, 4096, 8192, 16384, 32768]
    specs = [mb.BandSpec(s, n_atoms, s // 4, device="cuda", signal_samples=n_samples, is_lowest_band=(s == 512)) for s in sizes]
    model = mb.MultibandDictionaryLearning(specs, n_samples=n_samples)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((B, 1, n_samples)).astype(np.float32)).cuda()
    # (a signal with structure in every band: decaying sinusoids at random onsets)
    t = torch.arange(n_samples, device="cuda")[None, None, :]
    for f0 in (60., 250., 900., 2500., 6000.):
        on = int(rng.integers(0, n_samples // 2))
        x = x * 0.97 + 3.0 * torch.sin(2 * np.pi * f0 / 22050. * t) * torch.exp(-(t - on).clamp(min=0) / 3000.) * (t >= on)
    model.encode(x, steps)
    dt, enc = timed(lambda: model.encode(x, steps))
    dt_l, _ = timed(lambda: model.learn(x, steps), reps=2)
    dt_r, _ = timed(lambda: model.recon(x, steps), reps=2)
    print(f"e_2023_3_8 multiband: 7 bands x {n_atoms} atoms, B {B}, {steps} steps per band: encode {dt * 1e3:.1f} ms = "
          f"{7 * B * steps / dt / 1e3:.1f} k band-segment-iterations/s; learn {dt_l * 1e3:.1f} ms; recon {dt_r * 1e3:.1f} ms", flush=True)
    for s, spec in zip(sizes, specs):
        bands = model.shape_dict(B)
        xb = torch.from_numpy(rng.standard_normal((B, 1, s)).astype(np.float32)).cuda()
        spec.encode(xb, steps)
        dtb, _ = timed(lambda: spec.encode(xb, steps))
        print(f"    band {s} ({n_atoms} x {s // 4}): encode {dtb * 1e3:.2f} ms [{SCHED.get(nat.last_schedule(), nat.last_schedule())}]", flush=True)


single("e_2023_7_14", 1024, 512, 2 ** 15, 512, 600)
single("e_2023_7_20", 512, 512, 2 ** 15, 64, 192)
multiband()

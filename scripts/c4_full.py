#!/usr/bin/env python3
"""BASELINE configs[3] at full size: 4096 x 2048 dictionary, 128 x 131072-sample segments, K = 256, on the FFT
schedule; checks the round trip and (for the first segments and steps) equality with the incremental MFMA
schedule."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N, B, K = 4096, 2048, 131072, 128, 256
t0 = time.time()
d = synth.make_dictionary(A, L, seed=4000)
NE = int(os.environ.get("C4_EVENTS", 3 * K))   # E = 3 K planted events per segment
x = synth.make_segments(B, N, d, n_events=NE, seed=4001)
print(f"inputs generated in {time.time() - t0:.1f} s ({NE} planted events per segment)", flush=True)
xd = torch.from_numpy(x).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
nat.encode(xd[:4], du, 2, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
M = 8192
spectra_bytes = (A // 2 + 1) * 8.0 * M
t0 = time.perf_counter()
mu = nat.coherence_table(du)
torch.cuda.synchronize()
print(f"coherence table (mp_coherence_f32, first call): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
t0 = time.perf_counter()
mu = nat.coherence_table(du)
torch.cuda.synchronize()
print(f"coherence table (second call): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
ref = None
if os.environ.get("C4_PPS"):    # atom pairs per screen workgroup (16 = a whole tile)
    nat.tune(nat.MP_TUNE_SCREEN_PPS, int(os.environ["C4_PPS"]))
    print(f"screen pairs per workgroup {os.environ['C4_PPS']}", flush=True)
if os.environ.get("C4_FORCE"):  # TIMING ONLY: random tile masks (1.p per tile, 2.p per segment); events are then wrong
    os.environ["MP_ALLOW_WRONG_RESULTS"] = "1"
    nat.tune(nat.MP_TUNE_LAZY_FORCE, float(os.environ["C4_FORCE"]))
    print(f"forced random skip masks {os.environ['C4_FORCE']} (results invalid)", flush=True)
if os.environ.get("C4_COMPACT"):  # 0: masked screens exit per workgroup instead of running from the compacted work list
    nat.tune(nat.MP_TUNE_LAZY_COMPACT, int(os.environ["C4_COMPACT"]))
    print(f"compacted work list: {os.environ['C4_COMPACT']}", flush=True)
if os.environ.get("C4_TUNE"):   # "margin,reuse": how the screen's time follows the share of tiles skipped
    mg, ru = os.environ["C4_TUNE"].split(",")
    nat.tune(nat.MP_TUNE_LAZY_MARGIN, float(mg)); nat.tune(nat.MP_TUNE_LAZY_REUSE, int(ru))
    print(f"lazy margin {mg}, reuse {ru}", flush=True)
RUNS = (("one stream", nat.MP_FLAG_NO_OVERLAP, 16, False), ("one stream, lazy screen", nat.MP_FLAG_NO_OVERLAP, 16, mu)) if os.environ.get("C4_SHORT") else None
if os.environ.get("C4_SHORT") == "groups":
    RUNS = (("one stream, lazy screen", nat.MP_FLAG_NO_OVERLAP, 0, mu), ("two sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(2), 0, mu),
            ("four sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(4), 0, mu), ("one stream, lazy screen", nat.MP_FLAG_NO_OVERLAP, 0, mu),
            ("two sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(2), 0, mu),
            ("four sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(4), 0, mu),
            ("three sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(3), 0, mu),
            ("one stream, lazy screen", nat.MP_FLAG_NO_OVERLAP, 0, mu),
            ("four sub-batches, lazy screen", nat.MP_FLAG_OVERLAP | nat.flag_groups(4), 0, mu),
            ("four sub-batches, no table", nat.MP_FLAG_OVERLAP | nat.flag_groups(4), 0, False),
            ("one stream, no table", nat.MP_FLAG_NO_OVERLAP, 0, False))
for name, flags, every, co in RUNS or (("one stream, events around every launch", nat.MP_FLAG_NO_OVERLAP, 1, False),
                               ("one stream", nat.MP_FLAG_NO_OVERLAP, 16, False),
                               ("one stream, lazy screen", nat.MP_FLAG_NO_OVERLAP, 16, mu),
                               ("one stream, lazy screen, events around every launch", nat.MP_FLAG_NO_OVERLAP, 1, mu),
                               ("flags = 0, no table", 0, 0, False),
                               ("library default (table from the cache)", 0, 0, None),
                               ("library default again", 0, 0, None),
                               ("library default again", 0, 0, None)):
    nat.profile_enable(every); nat.profile_read(); nat.lazy_stats()
    t0 = time.perf_counter()
    atom, lag, gain, res = nat.encode(xd, du, K, path=nat.MP_PATH_FFT, flags=flags, coherence=co)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = nat.profile_read()
    ls = nat.lazy_stats()
    if ref is None:
        ref = (atom, lag, gain, res)
    print(f"   lazy screen: {ls}; identical to the first run: {all(torch.equal(a, b) for a, b in zip(ref, (atom, lag, gain, res)))}", flush=True)
    print(f"c4 full, {name}: B{B} K{K}: {dt:.3f} s -> {B * K / dt:.0f} seg-it/s; overflow segments: "
          f"{int(torch.isnan(gain).any(dim=1).sum())}", flush=True)
    print("  ", {k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in p.items()}, "ms avg, spans", flush=True)
    if flags and p['corr_inc'][1]:
        inc = p['corr_inc'][0] / p['corr_inc'][1] * 1e-3
        print(f"   screen: algorithmic spectra per incremental launch {B * spectra_bytes / 1e9:.2f} GB -> "
              f"{B * spectra_bytes / inc / 1e12:.2f} TB/s", flush=True)
nat.profile_enable(0)
rec = torch.zeros_like(xd)
nat.scatter(atom, torch.arange(B, device="cuda")[:, None].expand(B, K), lag, gain, du, rec)
print("round trip max err", float((rec + res - xd).abs().max()), "residual dB", float(20 * torch.log10(res.norm() / xd.norm())), flush=True)
ref = nat.encode(xd[:4], du, 16, path=nat.MP_PATH_INCREMENTAL)
print("first 4 segments x 16 steps == incremental MFMA schedule:",
      bool(torch.equal(ref[0], atom[:4, :16]) and torch.equal(ref[1], lag[:4, :16]) and torch.equal(ref[2], gain[:4, :16])), flush=True)

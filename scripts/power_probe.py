#!/usr/bin/env python3
"""Does the headline encode run at the board's power limit?  Encodes back to back for a few seconds while rocm-smi is
sampled from a thread: average socket power, power cap and shader clock.  Usage: python scripts/power_probe.py [seconds]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
samples, stop = [], threading.Event()


def sample():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True,
                                 text=True, timeout=5).stdout
            samples.append([ln.strip() for ln in out.splitlines() if "Power" in ln or "sclk" in ln])
        except Exception as e:  # noqa: BLE001
            samples.append([repr(e)])
        time.sleep(0.5)


for name, flags in (("idle", None), ("default (persistent form)", 0), ("one stream, launch per step", nat.MP_FLAG_NO_OVERLAP)):
    samples.clear(); stop.clear()
    th = threading.Thread(target=sample); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        if flags is None:
            time.sleep(0.1)
        else:
            for _ in range(20):
                nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
            torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print(f"== {name}: {n * B * K / dt / 1e3:.0f} k segment-iterations/s", flush=True)
    for s in samples[2:6]:
        print("   ", " | ".join(s), flush=True)

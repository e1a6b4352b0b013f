#!/usr/bin/env python3
"""Every surface of the drop-in, 24 calls each: median and worst wall time, and the container's throttle counters before and
after (a host path that opens OpenMP regions on every core gets the process throttled under a CPU quota: that shows as a
worst call of 50-130 ms beside a median of a few).   python scripts/stall_scan.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp, _native as nat, synth, streaming, multibanddict as mb
from mpcore import model as mpmodel

def throttled():
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except OSError:
        pass
    return -1

def scan(name, fn, n=24, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    th0 = throttled()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.asarray(ts)
    flag = "  <-- STALLS" if ts.max() > 4 * np.median(ts) + 2 else ""
    print(f"{name:52s} median {np.median(ts):8.2f} ms  worst {ts.max():8.2f} ms  throttled periods +{throttled() - th0}{flag}", flush=True)

print("torch threads", torch.get_num_threads(), "cpus visible", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-", flush=True)
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()[:, None, :]
scan("sparse_code", lambda: mp.sparse_code(x, d, n_steps=K))
scan("sparse_code(flatten=True)", lambda: mp.sparse_code(x, d, n_steps=K, flatten=True))
def sc_scatter():
    ev, sc = mp.sparse_code(x, d, n_steps=K, flatten=True)
    return sc(x.shape, ev)
scan("sparse_code(flatten=True) + scatter", sc_scatter)
def sc_tuples():
    ev, sc = mp.sparse_code(x, d, n_steps=K, flatten=True)
    return [e[0] for e in ev]
scan("sparse_code(flatten=True) + walk the tuples", sc_tuples)
scan("dictionary_learning_step", lambda: mp.dictionary_learning_step(x, d, n_steps=K))
scan("sparse_feature_map (B 8, K 16)", lambda: mp.sparse_feature_map(x[:8], d, n_steps=16))
xr = x[:8].clone().requires_grad_(True)
def loss_bw():
    l = mp.sparse_coding_loss(xr, x[:8] * 0.9, d, n_steps=16)
    l.backward()
scan("sparse_coding_loss + backward (B 8, K 16)", loss_bw)
scan("sparse_code_to_differentiable_key_points (B 8)", lambda: mp.sparse_code_to_differentiable_key_points(x[:8], d, n_steps=16))
audio = x[0, 0].repeat(4)[None, :]
scan("encode_streaming (4 x 32768 samples, hop = window)", lambda: streaming.encode_streaming(audio, d, window=N, n_steps=16))
specs = [mb.BandSpec(s, 128, s // 4, device="cuda", signal_samples=2 ** 15, is_lowest_band=(s == 512)) for s in (512, 1024, 2048, 4096, 8192, 16384, 32768)]
model = mb.MultibandDictionaryLearning(specs, n_samples=2 ** 15)
xa = torch.randn(4, 1, 2 ** 15, device="cuda")
scan("multiband encode (7 bands x 128 atoms, B 4)", lambda: model.encode(xa, 16))
scan("multiband learn", lambda: model.learn(xa, 16), n=12)
scan("multiband recon", lambda: model.recon(xa, 16), n=12)
net = mpmodel.MatchingPursuit(n_atoms=64, atom_samples=256, n_samples=8192, n_iterations=8).cuda()
try:
    ta = torch.randn(4, 1, 8192, device="cuda")
    def fwd_bw():
        out = net(ta)
        (out[0] if isinstance(out, (tuple, list)) else out).square().mean().backward()
    scan("MatchingPursuit model forward + backward", fwd_bw, n=12)
except Exception as e:   # noqa: BLE001 -- a scan, not a test
    print("model scan skipped:", repr(e)[:200], flush=True)

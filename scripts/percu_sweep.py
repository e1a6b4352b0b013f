import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import _native as nat, synth
A, L, N, K = int(os.environ.get('SWEEP_A', 512)), int(os.environ.get('SWEEP_L', 512)), 32768, 64
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for B in [int(v) for v in sys.argv[1:]] or (8, 16, 24, 32, 40, 48, 64):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
    for _ in range(20): nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    values = [0, 512, 768]
    times = {v: [] for v in values}
    for rep in range(5):
        for v in values:
            nat.tune(7, v)
            out = nat.encode(x, du, K, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
            assert all(torch.equal(p, q) for p, q in zip(out, ref))
            t0 = time.perf_counter()
            for _ in range(10): nat.encode(x, du, K, path=nat.MP_PATH_FFT)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 10)
    nat.tune(7, 0)
    print(f"B{B}: default {np.median(times[0])*1e3:.3f} ms | 2 per CU {np.median(times[512])*1e3:.3f} | 3 per CU {np.median(times[768])*1e3:.3f}", flush=True)

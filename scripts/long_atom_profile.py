"""Where a long-atom encode (L > 5398: split transforms) spends its time: event-timed kinds of mp_encode_f32
plus a rocprof-free per-launch view.  Usage: python scripts/long_atom_profile.py [A L N B K]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth  # noqa: E402

A, L, N, B, K = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (1024, 8192, 32768, 16, 32)
d = synth.make_dictionary(A, L, seed=N)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=48, seed=N)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
for name, flags in (("register screen", nat.MP_FLAG_NO_OVERLAP), ("plain screen", nat.MP_FLAG_NO_OVERLAP | nat.MP_FLAG_FFT_SIMPLE),
                    ("register screen, MFMA refine", nat.MP_FLAG_NO_OVERLAP | nat.MP_FLAG_REFINE_MFMA)):
    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nat.profile_enable(1)
    nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags)
    torch.cuda.synchronize()
    prof = nat.profile_read()
    nat.profile_enable(0)
    print(f"{name}: {dt * 1e3:.2f} ms/encode = {B * K / dt:.0f} seg-it/s; per-kind (ms total, launches): "
          + ", ".join(f"{k} {v[0]:.2f}/{v[1]}" for k, v in prof.items()), flush=True)

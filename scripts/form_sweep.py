#!/usr/bin/env python3
"""Which form of MP_PATH_FFT wins where: the persistent form against the launch-per-step forms (the library's sub-batch
choice; the fused select, which has the lazy screen between launches), with and without the coherence table, over
dictionary size, atom length and batch.  k segment-iterations/s, steady state (4 encodes after 2 warm-ups)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth
K = 32
N = int(os.environ.get('SWEEP_N', 32768))
dicts = [(512, 512), (1024, 512), (256, 1024), (1024, 1024), (512, 256), (2048, 256)]
if len(sys.argv) > 1:
    dicts = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for A, L in dicts:
    d = synth.make_dictionary(A, L, seed=A + L)
    du = nat.unit_norm(torch.from_numpy(d).cuda())
    mu = nat.coherence_table(du) if nat.lib().mp_coherence_workspace_bytes(A, L) else None
    nt = ((A + 31) // 32) * (16 // (max(1, 256 // (max(1024, 1 << (3 * L + 189).bit_length()) // 16)) * 4))
    for B in (8, 16, 32, 64, 128, 256):
        x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=7)).cuda()
        ref = nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=False)
        row = f"{A:5d}x{L:<5d} B{B:4d}:"
        for name, flags, co in (("persistent", nat.MP_FLAG_FFT_PERSISTENT, False), ("steps", nat.MP_FLAG_FFT_NO_PERSISTENT, False),
                                ("persistent+lazy", nat.MP_FLAG_FFT_PERSISTENT, mu), ("fused+lazy", nat.MP_FLAG_FFT_FUSED, mu),
                                ("fused+lazy one stream", nat.MP_FLAG_FFT_FUSED | nat.MP_FLAG_NO_OVERLAP, mu),
                                ("DEFAULT", 0, False), ("DEFAULT+table", 0, mu)):
            if co is None:
                continue
            f = lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=flags, coherence=co)
            out = f(); out = f(); torch.cuda.synchronize()
            keep = ~torch.isnan(out[2]).any(dim=1)
            same = all(torch.equal(p[keep], q[keep]) for p, q in zip(out, ref))
            t0 = time.perf_counter()
            for _ in range(4): out = f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            row += f"  {name} {B * K / dt / 1e3:6.0f}{'' if same else ' MISMATCH'}{'' if bool(keep.all()) else ' (marks)'}" + (f" [{nat.last_schedule()}]" if name.startswith("DEFAULT") else "")
        print(row, flush=True)

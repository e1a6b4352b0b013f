import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N = 512, 512, 32768
d = synth.make_dictionary(A, L, seed=1000)
du = nat.unit_norm(torch.from_numpy(d).cuda())
for B in (1, 4, 64):
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
    for path in (2, 1):
        nat.encode(x, du, 4, path=path); torch.cuda.synchronize()
        nat.profile_enable(True); nat.profile_read()
        nat.encode(x, du, 16, path=path); torch.cuda.synchronize()
        p = nat.profile_read()
        print(f"B{B} path{path}", {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in p.items()}, "us", flush=True)

import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import matchingpursuit as mp, _native as nat, synth
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()
def t(fn, n=40):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out
du = nat.unit_norm(d)
print("encode FFT no table:", t(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, want_residual=False, coherence=False)), flush=True)
print("encode FFT default:", t(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, want_residual=False)), flush=True)
def chk():
    a, l, g, r = nat.encode(x, du, K, path=nat.MP_PATH_FFT, want_residual=False)
    bad = torch.isnan(g).any(dim=1)
    return int(bad.sum())
print("encode + isnan:", t(chk), flush=True)
marks = []
def chk2():
    marks.append(chk())
print("marks:", t(chk2), marks, flush=True)
print("encode_checked:", t(lambda: nat.encode_checked(x, du, K, want_residual=False)), flush=True)

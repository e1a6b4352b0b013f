#!/usr/bin/env python3
"""A short FFT-schedule run at the BASELINE configs[3] shape, for rocprofv3 --pmc passes (fabric traffic of
the screen kernel when the pair spectra no longer fit the L2s):  c4_traffic.py [B] [K]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N = 4096, 2048, 131072
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = synth.make_dictionary(A, L, seed=4000)
x = synth.make_segments(B, N, d, n_events=int(os.environ.get("C4_EVENTS", 64)), seed=4001)
xd = torch.from_numpy(x).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
co = False
if os.environ.get("C4_LAZY") or os.environ.get("C4_FORCE"):   # the lazy screen (tile masks); C4_FORCE: random masks, timing only
    co = nat.coherence_table(du)
    if os.environ.get("C4_FORCE"):
        os.environ["MP_ALLOW_WRONG_RESULTS"] = "1"
        nat.tune(nat.MP_TUNE_LAZY_FORCE, float(os.environ["C4_FORCE"]))
    if os.environ.get("C4_COMPACT"):   # 0: masked screens exit per workgroup instead of running from the compacted work list
        nat.tune(nat.MP_TUNE_LAZY_COMPACT, int(os.environ["C4_COMPACT"]))
    if os.environ.get("C4_TUNE"):
        mg, ru = os.environ["C4_TUNE"].split(",")
        nat.tune(nat.MP_TUNE_LAZY_MARGIN, float(mg)); nat.tune(nat.MP_TUNE_LAZY_REUSE, int(ru))
out = nat.encode(xd, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP, coherence=co)
torch.cuda.synchronize()
print("done", int(torch.isnan(out[2]).any()), flush=True)

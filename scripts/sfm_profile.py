"""torch-profiler table of sparse_coding_loss forward + backward (modules/matchingpursuit.py:68-146) at the headline
dictionary, 4 segments x 16 steps: which tensor operations the reference-defined dense gradient spends its time in."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import matchingpursuit as mp
from mpcore import synth
from torch.profiler import profile, ProfilerActivity
A, L, N, B, K = 512, 512, 32768, 4, 16
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=48, seed=1002)).cuda()
y = (x + 0.05 * torch.randn_like(x))
def with_grad():
    yy = y.clone().requires_grad_(True)
    mp.sparse_coding_loss(yy, x, d, n_steps=K).backward()
with_grad(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with_grad(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))

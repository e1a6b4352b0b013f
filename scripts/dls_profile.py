"""torch-profiler table of one dictionary_learning_step (modules/matchingpursuit.py:348-419) at the headline shape:
device time per kernel (the encode's screen / select kernels and the fused dictionary_update_kernel)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "matching-pursuit_amd")
from mpcore import matchingpursuit as mp
from mpcore import synth
from torch.profiler import profile, ProfilerActivity
A, L, N, B, K = 512, 512, 32768, 64, 64
dn = synth.make_dictionary(A, L, seed=1000)
d = torch.from_numpy(dn).cuda()
x = torch.from_numpy(synth.make_segments(B, N, dn, n_events=192, seed=1002)).cuda()[:, None, :]
for _ in range(2): mp.dictionary_learning_step(x, d, n_steps=K)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t0=time.perf_counter(); mp.dictionary_learning_step(x, d, n_steps=K); torch.cuda.synchronize(); print("wall ms", (time.perf_counter()-t0)*1e3)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))

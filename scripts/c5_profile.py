#!/usr/bin/env python3
"""Where a config-5 train step (mpcore/model.py: analysis loop + iterative STFT loss + backward + Adam) spends
its time: torch profiler table, device time per operator."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import synth
from mpcore.model import MatchingPursuit, train_step
from torch.profiler import profile, ProfilerActivity
A, L, N, B, K = 512, 512, 32768, 8, 32
dev = "cuda:0"
torch.manual_seed(0)
model = MatchingPursuit(A, L, N, K).to(dev)
with torch.no_grad():
    model.atoms.copy_(torch.from_numpy(synth.make_dictionary(A, L, seed=5000))[None].to(dev) * 0.05)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=5000), n_events=96, seed=5001)).to(dev)[:, None, :]
win = torch.hann_window(2048, device=dev)
def transform(t):
    b = t.shape[0]
    s = torch.stft(t.reshape(-1, t.shape[-1]), 2048, 256, window=win, return_complex=True, center=True)
    return torch.abs(s).reshape(b, -1, s.shape[-2] * s.shape[-1])
for _ in range(3):
    train_step(model, opt, x, transform)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        train_step(model, opt, x, transform)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))

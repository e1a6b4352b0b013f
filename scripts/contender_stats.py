#!/usr/bin/env python3
"""How many cells does the screen hand to the exact refinement?  Counts, per segment and iteration, the
32-atom x 64-lag cells whose exact maximum lies within 2 * tau * ||window|| of the segment's maximum
(an upper estimate of select-A's contender set) at the headline shape."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat
from mpcore import synth
A, L, N, B = 512, 512, 32768, 16
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).cuda()
du = nat.unit_norm(torch.from_numpy(d).cuda())
for tau in (2e-5, 5e-6):
    counts = []
    for k in (0, 1, 4, 16, 32, 63):
        res = nat.encode(x, du, k, path=nat.MP_PATH_INCREMENTAL)[3] if k else x
        fm = nat.feature_map(res, du)                               # [B, A, N]
        cells = fm.view(B, A // 32, 32, N // 64, 64).amax(dim=(2, 4))   # [B, tiles, blocks]
        top = cells.amax(dim=(1, 2))
        # window norm ~ norm of the ~2048 samples around the block; use a running norm per block
        e = torch.nn.functional.pad(res, (0, 2048)) ** 2
        cs = torch.cumsum(e, dim=-1)
        idx = torch.arange(0, N, 64, device=res.device)
        wn = torch.sqrt(cs[:, (idx + 2047).clamp(max=cs.shape[1] - 1)] - cs[:, idx] + e[:, idx])   # [B, blocks]
        eps = tau * wn[:, None, :]
        lb = (cells - eps).amax(dim=(1, 2))
        n = ((cells + eps) >= lb[:, None, None]).sum(dim=(1, 2))
        counts.append(n.cpu().numpy())
        print(f"tau {tau:g} k {k:2d}: contenders per segment mean {n.float().mean():.2f} max {int(n.max())}", flush=True)

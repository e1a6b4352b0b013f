import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "matching-pursuit_amd"))
from mpcore import _native as nat, synth, encode_packed
A, L, N, B, K = 64, 256, 6000, 30, 12
d = torch.from_numpy(synth.make_dictionary(A, L, seed=5)).to("cuda:0")
x = torch.from_numpy(synth.make_segments(B, N, d.cpu().numpy(), n_events=20, seed=6)).to("cuda:0")
for call in range(5):
    out = encode_packed(x, d, K)
    torch.cuda.synchronize()
    for k, e in nat._coherence_cache.items():
        print(call, k, "pending", e.pending, "flag", bool(e.flag[0]), "volatile", e.volatile, "table", e.table is not None,
              "query", e.event.query(), "equal now", bool((out["dict_unit"] == e.copy).all()), "lazy", nat._tls.lazy,
              "sched", nat.last_schedule(), "skipped", nat.persist_stats()["skipped"])

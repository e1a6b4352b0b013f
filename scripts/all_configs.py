#!/usr/bin/env python3
"""One measurement per BASELINE.json configuration on this GPU, written as JSON (profiles/rNN_configs.json):
  configs[0]  16 x 256, one 8192-sample segment, 8 steps          latency of one encode (plain and hipGraph replay)
  configs[1]  512 x 512, 64 x 32768 samples, 64 steps             the bench.py workload (library default schedule)
  configs[3]  4096 x 2048, 128 x 131072 samples, 256 steps        one full encode, round-trip checked
  configs[4]  512 x 512, 8 x 32768 samples, 32 steps + STFT loss  one train step of the mp.py model
(configs[2] is configs[1] on 8 GPUs: bench.py --gpus 8.)   Usage: python scripts/all_configs.py > out.json"""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
from mpcore import _native as nat, synth
from mpcore.model import MatchingPursuit, train_step

DEV = "cuda:0"
out = {"device": torch.cuda.get_device_name(0)}


def med(fn, n=9, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


# configs[0]
A, L, N, B, K = 16, 256, 8192, 1, 8
d = synth.make_dictionary(A, L, seed=100)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=24, seed=101)).to(DEV)
du = nat.unit_norm(torch.from_numpy(d).to(DEV))
t_plain = med(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT), n=30)
plan = nat.EncodePlan(B, N, du, K, path=nat.MP_PATH_FFT)
t_graph = med(lambda: plan(x), n=30)
out["configs[0]"] = {"shape": "A16 L256 N8192 B1 K8", "encode_us": round(t_plain * 1e6, 1),
                     "encode_us_hipgraph_replay": round(t_graph * 1e6, 1),
                     "segment_iterations_per_s": round(B * K / t_graph)}
# configs[1]
A, L, N, B, K = 512, 512, 32768, 64, 64
d = synth.make_dictionary(A, L, seed=1000)
x = torch.from_numpy(synth.make_segments(B, N, d, n_events=192, seed=1002)).to(DEV)
du = nat.unit_norm(torch.from_numpy(d).to(DEV))
t_def = med(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT))
t_one = med(lambda: nat.encode(x, du, K, path=nat.MP_PATH_FFT, flags=nat.MP_FLAG_NO_OVERLAP))
plan = nat.EncodePlan(B, N, du, K, path=nat.MP_PATH_FFT)
t_graph = med(lambda: plan(x))
out["configs[1]"] = {"shape": "A512 L512 N32768 B64 K64",
                     "segment_iterations_per_s": {"library_default": round(B * K / t_def), "one_stream": round(B * K / t_one),
                                                  "library_default_hipgraph_replay": round(B * K / t_graph)}}
# configs[4]
A, L, N, B, K = 512, 512, 32768, 8, 32
torch.manual_seed(0)
model = MatchingPursuit(A, L, N, K).to(DEV)
with torch.no_grad():
    model.atoms.copy_(torch.from_numpy(synth.make_dictionary(A, L, seed=5000))[None].to(DEV) * 0.05)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
xt = torch.from_numpy(synth.make_segments(B, N, synth.make_dictionary(A, L, seed=5000), n_events=96, seed=5001)).to(DEV)[:, None, :]
from mpcore.model import reference_stft
t_evt = med(lambda: train_step(model, opt, xt, ("stft", 2048, 256)))
t_dense = med(lambda: train_step(model, opt, xt, lambda t: reference_stft(t, 2048, 256).reshape(t.shape[0], t.shape[1], -1)), n=5)
out["configs[4]"] = {"shape": "A512 L512 N32768 B8 K32, STFT(2048, 256) iterative loss, Adam",
                     "train_step_ms": {"event_form_of_the_loss": round(t_evt * 1e3, 2), "generic_iterative_loss": round(t_dense * 1e3, 2)},
                     "segment_iterations_per_s": round(B * K / t_evt)}
del model, opt, xt
torch.cuda.empty_cache()
# configs[3]
A, L, N, B, K = 4096, 2048, 131072, 128, 256
d = synth.make_dictionary(A, L, seed=4000)
xh = synth.make_segments(B, N, d, n_events=3 * K, seed=4001)   # E = 3 K planted events per segment
x = torch.from_numpy(xh).to(DEV)
du = nat.unit_norm(torch.from_numpy(d).to(DEV))
nat.encode(x[:4], du, 2, path=nat.MP_PATH_FFT); torch.cuda.synchronize()
runs = []
for _ in range(3):   # (the first call also builds the dictionary's coherence table, 40 ms)
    t0 = time.perf_counter()
    atom, lag, gain, res = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    runs.append(time.perf_counter() - t0)
t0 = time.perf_counter()
nat.encode(x, du, K, path=nat.MP_PATH_FFT, coherence=False)
torch.cuda.synchronize()
dt_plain = time.perf_counter() - t0
dt = min(runs)
rec = torch.zeros_like(x)
nat.scatter(atom, torch.arange(B, device=DEV)[:, None].expand(B, K), lag, gain, du, rec)
out["configs[3]"] = {"shape": "A4096 L2048 N131072 B128 K256, 768 planted events per segment", "encode_s": round(dt, 3),
                     "encode_s_first_call": round(runs[0], 3), "encode_s_without_the_coherence_table": round(dt_plain, 3),
                     "segment_iterations_per_s": round(B * K / dt),
                     "round_trip_max_abs_error": float((rec + res - x).abs().max()),
                     "residual_db": round(float(20 * torch.log10(res.norm() / x.norm())), 2),
                     "screen_overflow_segments": int(torch.isnan(gain).any(dim=1).sum())}
print(json.dumps(out, indent=1))

"""Time mp_encode_lcn_f32 (the native local-contrast-norm schedule, matchingpursuit.py:284-294) beside the
hook-serving dense loop (mp_feature_map_f32 + torch's avg_pool2d / argmax on the device) it replaced as the
default for local_contrast_norm=True.  Usage: python scripts/lcn_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matching-pursuit_amd"))
from mpcore import _native as nat, synth  # noqa: E402
from mpcore import matchingpursuit as mp  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    return min(t)


for A, L, N, B, K in [(512, 512, 32768, 16, 32), (512, 512, 32768, 64, 64), (512, 128, 8192, 16, 32),
                      (256, 64, 2048, 32, 32)]:
    d = synth.make_dictionary(A, L, seed=1)
    x = torch.from_numpy(synth.make_segments(B, N, d, n_events=3 * K, seed=1)).to(DEV)
    du = nat.unit_norm(torch.from_numpy(d).to(DEV))
    t_nat = timed(lambda: nat.encode_lcn(x, du, K))
    small = min(B, 4)
    t_dense = timed(lambda: mp._sparse_code_dense(x[:small, None, :], du, K, None, None, None, True, None), reps=1)
    a, l, g, r = nat.encode_lcn(x[:small], du, K)
    a2, l2, g2, r2, _ = mp._sparse_code_dense(x[:small, None, :], du, K, None, None, None, True, None)
    same = bool((a == a2).all() and (l == l2).all())
    print(f"A{A} L{L} N{N} B{B} K{K}: native {t_nat * 1e3:8.2f} ms = {B * K / t_nat:9.0f} seg-it/s | "
          f"dense torch loop (B={small}) {small * K / t_dense:8.0f} seg-it/s | same picks as torch box filter: {same}",
          flush=True)

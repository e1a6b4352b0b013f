#!/usr/bin/env python3
"""The training loop of the reference's experiments/archive/e_2023_7_14/experiment.py on synthetic audio, through the drop-in:

    encoded, scatter = sparse_code(batch, d, flatten=True, n_steps=n_steps)     # events of the current dictionary
    recon = scatter(batch.shape, encoded)                                       # what they explain
    d[:] = dictionary_learning_step(batch, d, n_steps=n_steps)                  # the dictionary moves towards the signal

With the reference checked out next to this repository the names come from ITS package under the overlay
(`python -m mpcore.run examples/dictionary_learning_loop.py`, or mpcore.install() as below); without it, from mpcore itself.
Signals: events of a hidden dictionary on a noise bed (mpcore/synth.py).  Prints the share of the signal's energy the
events explain, per iteration -- it rises as the random starting dictionary turns into the hidden one -- and the time per
iteration.   python examples/dictionary_learning_loop.py [iterations] [batch]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
import mpcore  # noqa: E402
from mpcore import synth  # noqa: E402

REFERENCE = os.environ.get("MP_REFERENCE", "/root/reference")
if os.path.isdir(os.path.join(REFERENCE, "modules")):
    sys.path.insert(0, REFERENCE)                   # the reference's own package, hot-path names rebound by the overlay
mode = mpcore.install()                             # "overlay", or "standalone" where no reference checkout is importable
from modules.matchingpursuit import dictionary_learning_step, sparse_code  # noqa: E402  (the reference's import line)
source = f"modules.matchingpursuit ({mode})"


def run(iterations=12, batch=8, d_size=64, kernel_size=256, n_samples=2 ** 14, n_steps=32, device="cuda", log=print):
    hidden = synth.make_dictionary(d_size, kernel_size, seed=11)
    d = torch.zeros(d_size, kernel_size, device=device).uniform_(-1, 1)     # e_2023_7_14/experiment.py:29
    explained = []
    for i in range(iterations):
        x = torch.from_numpy(synth.make_segments(batch, n_samples, hidden, n_events=n_steps, seed=100 + i)).to(device)[:, None, :]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        encoded, scatter = sparse_code(x, d, device=device, flatten=True, n_steps=n_steps)
        recon = scatter(x.shape, encoded)
        d[:] = dictionary_learning_step(x, d, n_steps=n_steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        share = 1.0 - float(((x - recon) ** 2).sum() / (x ** 2).sum())
        explained.append(share)
        log(f"iteration {i:2d}: {len(encoded)} events explain {100 * share:5.1f} % of the energy; {dt * 1e3:.2f} ms")
    return explained


if __name__ == "__main__":
    print("names from", source, flush=True)
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 8)

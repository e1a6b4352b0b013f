#!/usr/bin/env python3
"""dictionary_learning_step across ranks (modules/matchingpursuit.py:348-419 by global dependency levels, one
[atoms in level, L] all-reduce per level): two ranks sharing this box's one MI355X, headline dictionary, 32 segments each,
beside the single-process step over the same 64 segments.  `--backend nccl` needs one GPU per rank (not this box)."""
import os, socket, sys, time
import numpy as np, torch
import torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
A, L, N, B, K = 512, 512, 32768, 64, 64


def worker(rank, world, port, x_full, d, ret):
    sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
    import torch.distributed as dist
    from mpcore import dist as mpdist
    import mpcore.matchingpursuit as mpm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    mpdist.init_from_env(backend="gloo")
    lo, hi = mpdist.shard_range(x_full.shape[0], rank, world)
    shard = torch.from_numpy(x_full[lo:hi]).to("cuda:0")[:, None, :]
    dd = torch.from_numpy(d).to("cuda:0")
    ts = []
    for it in range(4):
        mpdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = mpm.dictionary_learning_step(shard, dd, n_steps=K, process_group=dist.group.WORLD)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ret.put((rank, ts, out.cpu().numpy()))
    mpdist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    from mpcore import synth
    import mpcore.matchingpursuit as mpm
    d = synth.make_dictionary(A, L, seed=1000)
    x = synth.make_segments(B, N, d, n_events=3 * K, seed=1002)
    xs = torch.from_numpy(x).to("cuda:0")[:, None, :]
    dd = torch.from_numpy(d).to("cuda:0")
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        single = mpm.dictionary_learning_step(xs, dd, n_steps=K)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"single process, 64 segments: {min(ts[1:]):.2f} ms per step (runs: {[round(t, 2) for t in ts]})", flush=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, x, d, ret)) for r in range(2)]
    for p in procs: p.start()
    got = dict((r, (t, o)) for r, t, o in (ret.get(timeout=300) for _ in range(2)))
    for p in procs: p.join(60)
    for r in range(2):
        t, o = got[r]
        print(f"rank {r} of 2 (gloo, both on this GPU, 32 segments each): {min(t[1:]):.2f} ms per step (runs: {[round(v, 2) for v in t]}); "
              f"max |d - single| {np.abs(o - single.cpu().numpy()).max():.2e}", flush=True)

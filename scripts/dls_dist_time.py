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
    # for comparison: rounds 1-2's form -- the same step atom by atom, one blocking [L] all-reduce per used atom
    from mpcore import _native as nat
    tp = []
    for it in range(2):
        mpdist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        sig = shard[:, 0, :].contiguous()
        d_work = nat.unit_norm(dd)
        residual = sig.clone()
        atom, lag, gain, _ = nat.encode_checked(sig, d_work, K, want_residual=False)
        rows = d_work[atom] * gain[..., None]
        anorm = torch.norm(rows, dim=-1)
        atom_global, _ = mpdist.gather_batch(atom, dist.group.WORLD)
        order = mpm.first_selection_order(atom_global.cpu())
        perm, counts = mpm.group_events_by_atom(atom.cpu(), order, A)
        perm_d = perm.to("cuda:0")
        ev_batch, ev_lag = perm_d // K, lag.reshape(-1)[perm_d]
        ev_rows, ev_norm = rows.reshape(-1, L)[perm_d], anorm.reshape(-1)[perm_d]
        sparse = torch.empty_like(residual)
        start = 0
        for oi, index in enumerate(order):
            n = counts[oi]; sl = slice(start, start + n); start += n
            if n:
                sparse.zero_(); nat.scatter_rows(ev_rows[sl], ev_batch[sl], ev_lag[sl], sparse); residual += sparse
                acc = nat.gather_sum(residual, ev_batch[sl], ev_lag[sl], L)
            else:
                acc = torch.zeros(L, dtype=torch.float64, device="cuda:0")
            acc = mpdist.all_reduce_sum(acc, dist.group.WORLD)
            new_atom = nat.unit_norm(acc.to(torch.float32).view(1, L))
            d_work[index] = new_atom[0]
            if n:
                sparse.zero_(); nat.scatter_rows(new_atom * ev_norm[sl, None], ev_batch[sl], ev_lag[sl], sparse); residual -= sparse
        old = nat.unit_norm(d_work)
        torch.cuda.synchronize(); tp.append((time.perf_counter() - t0) * 1e3)
    if rank == 0:
        print(f"  (atom by atom, one [L] all-reduce per used atom -- rounds 1-2: {min(tp):.1f} ms per step, {len(order)} atoms; "
              f"max |d - by levels| {float((old - out).abs().max()):.2e})", flush=True)
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

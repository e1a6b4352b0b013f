"""World-size-2 rehearsal of the multi-GPU plumbing on CPU (gloo): sharding, the rank-ordered gather
that fixes dictionary_learning_step's global atom order, the all-reduces, and the benchmark's
max-over-ranks timing reduction.  The data path itself needs no collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, atom_full, ret):
    sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
    from mpcore import dist as mpdist
    from mpcore.matchingpursuit import first_selection_order, group_events_by_atom
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = mpdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and mpdist.is_distributed()
    B = atom_full.shape[0]
    lo, hi = mpdist.shard_range(B, rank, world)  # uneven on purpose (B = 5, world = 2 -> 3 + 2)
    local = atom_full[lo:hi].clone()
    glob, off = mpdist.gather_batch(local)
    assert off == lo and torch.equal(glob, atom_full)
    order = first_selection_order(glob)
    # every rank derives the same global order, and its local groups are slices of the global grouping
    perm, counts = group_events_by_atom(local, order, int(atom_full.max()) + 1)
    tot = mpdist.all_reduce_sum(torch.tensor(counts, dtype=torch.float64))
    gperm, gcounts = group_events_by_atom(atom_full, order, int(atom_full.max()) + 1)
    assert tot.tolist() == [float(c) for c in gcounts]
    # the [L] window-sum all-reduce: sum of per-rank partial sums == single-process sum
    x = torch.arange(8, dtype=torch.float64) * (rank + 1)
    s = mpdist.all_reduce_sum(x.clone())
    assert torch.equal(s, torch.arange(8, dtype=torch.float64) * 3)
    # config 5: the [A, L] dictionary gradient is averaged over ranks with one flat all-reduce
    from mpcore.model import all_reduce_gradients
    p1 = torch.nn.Parameter(torch.zeros(4, 6))
    p2 = torch.nn.Parameter(torch.zeros(3))
    p1.grad = torch.full((4, 6), float(rank + 1))
    p2.grad = torch.arange(3, dtype=torch.float32) * (rank + 1)
    all_reduce_gradients([p1, p2])
    assert torch.equal(p1.grad, torch.full((4, 6), 1.5)) and torch.equal(p2.grad, torch.arange(3.0) * 1.5)
    # bench.py's timing reduction: max over ranks
    t = mpdist.all_reduce_max(torch.tensor([0.5 + rank], dtype=torch.float64))
    assert t.item() == 1.5
    mpdist.barrier()
    if rank == 0:
        ret.put((order, counts))
    dist.destroy_process_group()


def test_gloo_world2_sharding_order_and_reductions():
    rng = np.random.default_rng(5)
    atom_full = torch.from_numpy(rng.integers(0, 7, size=(5, 6)))
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, atom_full, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    order, counts = ret.get(timeout=5)
    # the order equals the reference's dict-insertion order over (step, batch)
    seen = []
    for k in range(atom_full.shape[1]):
        for b in range(atom_full.shape[0]):
            a = int(atom_full[b, k])
            if a not in seen:
                seen.append(a)
    assert order == seen


def test_grouping_matches_reference_flatten_order(golden_dir):
    """flatten_atom_dict order (grouped by atom, first-selection order; matchingpursuit.py:61-65)
    reproduced from packed arrays, against what the real reference returned."""
    from mpcore.matchingpursuit import first_selection_order, group_events_by_atom
    for name in ("encode_mid_64x128_n4096_b3_k16", "encode_ragged_24x100_n1000_b2_k12"):
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        atom = torch.from_numpy(z["atom"])
        lag = torch.from_numpy(z["lag"])
        B, K = atom.shape
        order = first_selection_order(atom)
        perm, counts = group_events_by_atom(atom, order, int(atom.max()) + 1)
        got = torch.stack([atom.reshape(-1)[perm], perm // K, lag.reshape(-1)[perm]], dim=1).numpy()
        assert np.array_equal(got, z["flat_order"])
        assert sum(counts) == B * K


def _bench(*argv):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True,
                          env=env, timeout=600)


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver starts it) must start
    the two ranks itself and relay exactly one JSON line from rank 0 (--dry-run: gloo on CPU, no encode)."""
    import json
    out = _bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["max_over_ranks"] == 2.0            # the max over ranks of (1 + rank)
    assert line["segments_all_ranks"] == 128        # 64 segments per rank, weak scaling


@pytest.mark.skipif(torch.cuda.is_available(), reason="uses the no-GPU failure of a rank")
def test_bench_launcher_exits_nonzero_when_a_rank_fails():
    out = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-variants")
    assert out.returncode != 0
    assert out.stdout.strip() == ""                 # no JSON line from a failed job
    assert "needs a GPU" in out.stderr

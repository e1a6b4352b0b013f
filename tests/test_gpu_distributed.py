"""Two ranks sharing one MI355X (gloo for the collectives, the device for the work): the multi-GPU form of
dictionary_learning_step -- the atom-by-atom loop by global dependency levels, one [atoms in level, L] all-reduce of the
window sums per level -- gives every rank the dictionary a single process computes from the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, x_full, d, n_steps, ret):
    sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
    import torch.distributed as dist
    import mpcore
    from mpcore import dist as mpdist
    mpcore.install()   # (a spawned process has no conftest: the stand-alone `modules` package, as there)
    import modules.matchingpursuit as mpm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")   # both ranks on cuda:0
    mpdist.init_from_env(backend="gloo")
    lo, hi = mpdist.shard_range(x_full.shape[0], rank, world)
    shard = torch.from_numpy(x_full[lo:hi]).to("cuda:0")[:, None, :]
    out = mpm.dictionary_learning_step(shard, torch.from_numpy(d).to("cuda:0"), n_steps=n_steps,
                                       process_group=dist.group.WORLD)
    ret.put((rank, out.cpu().numpy()))
    mpdist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("A,L,N,B,K,ne", [(24, 64, 1500, 5, 6, 10),        # 5 segments: an uneven 3 + 2 split
                                          (96, 128, 3000, 6, 24, 40)])     # crowded segments: many atoms share samples -> many levels
def test_two_rank_dictionary_learning_step_equals_single_process(A, L, N, B, K, ne):
    from mpcore import synth
    import modules.matchingpursuit as mpm
    d = synth.make_dictionary(A, L, seed=61)
    x = synth.make_segments(B, N, d, n_events=ne, seed=62)
    single = mpm.dictionary_learning_step(torch.from_numpy(x).to("cuda:0")[:, None, :], torch.from_numpy(d).to("cuda:0"),
                                          n_steps=K).cpu().numpy()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, d, K, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # fp64 partial sums are added in a different order than the single-process sum: equal to fp32 rounding
    for r in range(2):
        assert np.abs(got[r] - single).max() <= 2e-6
    assert np.array_equal(got[0], got[1])


def _train_worker(rank, world, port, x_full, atoms0, n_steps, ret):
    sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))
    import torch.distributed as dist
    from mpcore import dist as mpdist
    from mpcore.model import MatchingPursuit, all_reduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    mpdist.init_from_env(backend="gloo")
    A, L = atoms0.shape
    model = MatchingPursuit(A, L, x_full.shape[-1], n_steps).to("cuda:0")
    with torch.no_grad():
        model.atoms.copy_(torch.from_numpy(atoms0)[None].to("cuda:0"))
    lo, hi = mpdist.shard_range(x_full.shape[0], rank, world)
    target = torch.from_numpy(x_full[lo:hi]).to("cuda:0")[:, None, :]
    recon = model(target).sum(dim=1, keepdim=True)
    ((recon - target) ** 2).mean().backward()
    all_reduce_gradients(list(model.parameters()), dist.group.WORLD)
    ret.put((rank, model.atoms.grad.cpu().numpy()))
    mpdist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_of_the_model():
    """config 5's data parallelism: every rank ends up with the mean of the per-shard dictionary gradients."""
    from mpcore import synth
    from mpcore.model import MatchingPursuit
    A, L, N, B, K = 12, 32, 512, 4, 4
    atoms0 = (synth.make_dictionary(A, L, seed=71) * 0.22).astype(np.float32)
    x = synth.make_segments(B, N, synth.make_dictionary(A, L, seed=71), n_events=6, seed=72)
    want = []
    for lo, hi in ((0, 2), (2, 4)):
        model = MatchingPursuit(A, L, N, K).to("cuda:0")
        with torch.no_grad():
            model.atoms.copy_(torch.from_numpy(atoms0)[None].to("cuda:0"))
        target = torch.from_numpy(x[lo:hi]).to("cuda:0")[:, None, :]
        recon = model(target).sum(dim=1, keepdim=True)
        ((recon - target) ** 2).mean().backward()
        want.append(model.atoms.grad.cpu().numpy())
    want = (want[0] + want[1]) / 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, x, atoms0, K, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.abs(got[r] - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-12)
    assert np.array_equal(got[0], got[1])


def test_rccl_backend_initialises_and_runs_the_collectives_used():
    """A world-size-1 `nccl` (= RCCL) group in a child process: the backend loads on this box and the three
    collectives bench.py / dictionary_learning_step issue complete on device tensors."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "rccl_single_rank_check.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_internal_streams_run_side_by_side_even_when_created_after_a_graph_capture():
    """The sub-batch streams are chosen by a spin test (stream_pool): in a fresh process that captured a hipGraph
    before the library created its streams -- the order that used to put the first two on one hardware queue and
    made the default schedule 30 % slower than one stream -- every pair of the kept streams overlaps."""
    import ast
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "after_capture.py"), "torch-graph-first",
                          "ratios-only"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("spin ratio")][0]
    ratios = ast.literal_eval(line[line.index("{"):])
    kept = int([l for l in out.stdout.splitlines() if l.startswith("streams kept")][0].rsplit(":", 1)[1])
    # the pool keeps the candidates its own spin test saw overlapping (streams 0 .. kept - 1; the rest of the table are
    # candidates it rejected, never used): at least two, and every pair of the kept ones side by side when measured again
    assert len(ratios) == 6 and kept >= 2, (kept, ratios)
    assert all(0.5 < r < 1.5 for k, r in ratios.items() if int(k[1]) < kept and int(k[2]) < kept), (kept, ratios)

"""CPU-side checks: the C-ABI library loads and exports every symbol include/mpcore.h declares
(no compute calls without a GPU), the host-side mirror fails loudly without a device, and the
pure-host pieces (iterative_loss mirror, sharding helpers) agree with the reference's golden
vectors."""
import ctypes
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from mpcore import _native as nat
from mpcore import dist as mpdist
from mpcore import iterative as mpit

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "mpcore.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(nat.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(nat.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mpcore.h but not exported"
    assert set(nat.EXPORTS) == set(declared)
    assert nat.lib().mp_version() >= 1


def test_workspace_query_is_host_only_and_validates():
    n = nat.workspace_bytes(64, 32768, 512, 512, 64, nat.MP_PATH_INCREMENTAL)
    assert 8 * 2**20 < n < 64 * 2**20  # residual 8.5 MiB + dictionary image 1 MiB + keys
    with pytest.raises(nat.NativeError):
        nat.workspace_bytes(1, 1 << 20, 1 << 14, 64, 1, 0)  # A*N >= 2^32


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_a_gpu():
    import modules.matchingpursuit as mp
    with pytest.raises(nat.NativeError):
        mp.sparse_code(torch.zeros(1, 1, 256), torch.rand(4, 16), n_steps=2)
    with pytest.raises(nat.NativeError):
        mp.dictionary_learning_step(torch.zeros(1, 1, 256), torch.rand(4, 16), n_steps=2)
    with pytest.raises(nat.NativeError):
        nat.encode(torch.zeros(1, 256), torch.rand(4, 16), 2)
    with pytest.raises(ValueError):
        mp.sparse_code(torch.zeros(1, 256), torch.rand(4, 16), n_steps=2)  # non 3-D, like :244


def test_drop_in_surface_names():
    import modules
    import modules.iterative as it
    import modules.matchingpursuit as mp
    for n in ("sparse_code", "dictionary_learning_step", "sparse_feature_map", "build_scatter_segments",
              "flatten_atom_dict", "sparse_coding_loss", "SparseCodingLoss"):
        assert hasattr(mp, n)
    assert hasattr(it, "iterative_loss") and hasattr(it, "sort_channels_descending_norm")
    assert hasattr(modules, "iterative_loss") and hasattr(modules, "fft_convolve")
    sig = inspect.signature(mp.sparse_code)
    assert list(sig.parameters)[:5] == ["signal", "d", "n_steps", "device", "approx"]
    assert sig.parameters["n_steps"].default == 100
    assert "local_constrast_norm" in inspect.signature(mp.dictionary_learning_step).parameters  # sic


def _stft(x, ws=512, step=128):
    # same transform the fixture was generated with (stft(x, 512, 128, pad=True))
    frames = x.shape[-1] // step
    x = torch.nn.functional.pad(x, (0, ws)).unfold(-1, ws, step)
    x = x * torch.hann_window(ws)[None, None, :]
    return torch.abs(torch.fft.rfft(x, norm="ortho"))[:, :, :frames, :]


def test_iterative_loss_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "iterative_loss.npz"))
    target = torch.from_numpy(z["target"])
    chans = torch.from_numpy(z["channels"])
    assert np.abs(_stft(target).numpy() - z["stft_target"]).max() <= 1e-5
    for tag, kw in [("default", {}), ("ratio", {"ratio_loss": True}), ("nosort", {"sort_channels": False})]:
        r, l = mpit.iterative_loss(target, chans, _stft, return_residual=True, **kw)
        assert r.shape == z[f"residual_{tag}"].shape
        assert np.abs(r.numpy() - z[f"residual_{tag}"]).max() <= 2e-5
        assert abs(l.item() - float(z[f"loss_{tag}"])) <= 1e-5 * abs(float(z[f"loss_{tag}"])) + 1e-3
    srt = mpit.sort_channels_descending_norm(chans)
    assert np.array_equal(srt.numpy(), z["sorted_channels"])
    # shape contract of the reference's own test (modules/test_modules.py:41-55)
    r, _ = mpit.iterative_loss(torch.zeros(3, 1, 2048), torch.zeros(3, 4, 2048), _stft, return_residual=True)
    assert r.shape == (3, int(np.prod(_stft(torch.zeros(3, 1, 2048)).shape[1:])))


def test_iterative_loss_gradients_flow():
    target = torch.randn(2, 1, 1024)
    chans = torch.randn(2, 3, 1024, requires_grad=True)
    loss = mpit.iterative_loss(target, chans, _stft)
    loss.backward()
    assert chans.grad is not None and torch.isfinite(chans.grad).all() and chans.grad.abs().sum() > 0


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            spans = [mpdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sparse_helpers_match_reference_golden(golden_dir):
    from mpcore import sparse as mps
    z = np.load(os.path.join(golden_dir, "sparse_helpers.npz"))
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = mps.soft_dirac(x)
    assert np.abs(y.detach().numpy() - z["soft_dirac"]).max() <= 2e-7
    (y * torch.from_numpy(z["w"])).sum().backward()
    assert np.abs(x.grad.numpy() - z["soft_dirac_grad"]).max() <= 1e-6
    sp, packed, onehot = mps.sparsify2(torch.from_numpy(z["x3"]), n_to_keep=4)
    assert np.array_equal(sp.numpy(), z["sparse"])
    assert np.array_equal(packed.numpy(), z["packed"])
    assert np.array_equal(onehot.numpy(), z["one_hot"])


def test_band_split_matches_reference_golden(golden_dir):
    """modules/decompose.py mirror (pure torch.fft: runs anywhere)."""
    from mpcore import decompose as dec
    z = np.load(os.path.join(golden_dir, "multiband.npz"))
    x = torch.from_numpy(z["signal"])[:, None, :]
    split = dec.fft_frequency_decompose(x, 512)
    assert sorted(split.keys()) == z["sizes"].tolist()
    for size, band in split.items():
        assert np.abs(band.numpy() - z[f"band_{size}"]).max() <= 2e-6
    rec = dec.fft_frequency_recompose(split, x.shape[-1])
    assert np.abs(rec.numpy() - z["recompose"]).max() <= 2e-6
    d1 = torch.from_numpy(z["dict_1024"])
    d1 = d1 / (torch.norm(d1, dim=-1, keepdim=True) + 1e-8)
    ra = dec.fft_resample(d1.view(8, 1, 32), 64, False)
    assert np.abs(ra.numpy() - z["resampled_atoms_band1"]).max() <= 2e-6


def test_hot_kernels_keep_their_register_budget():
    """The register screen kernels live at the 128-VGPR edge of four wavefronts per SIMD; a change elsewhere in
    the translation unit can tip them into spilling (it happened: 24 spilled registers in the headline kernel from
    an unrelated template parameter).  Read the built code object's metadata -- no compile, no GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(kr.LIB) or not os.path.exists(os.path.join(kr.LLVM, "clang-offload-bundler")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    res = kr.kernel_resources()
    assert len(res) > 50
    screens = {n: r for n, r in res.items() if "17fft_screen_kernelILi" in n}
    assert len(screens) == 9   # 2^10 .. 2^14, and the |.| variant of 2^10 .. 2^13 (the coherence table's screen)
    for name, r in screens.items():
        assert r["spill"] == 0 and r["vgpr"] <= 128, (name, r)
    persistent = {n: r for n, r in res.items() if "correlate_persistent_kernel" in n}
    assert persistent and all(r["spill"] == 0 for r in persistent.values())


def _kernel_resources_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "scripts", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    if not os.path.exists(kr.LIB) or not os.path.exists(os.path.join(kr.LLVM, "llvm-objdump")):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    return kr


def test_lds_reads_of_the_transform_are_consumed_behind_their_wait():
    """csrc/mpfft.inc issues the transform's LDS reads as `asm volatile("ds_read_b64 ...")` and waits with a separate
    `asm volatile("s_waitcnt lgkmcnt(0)")` -- an order the compiler cannot see: a multiply that consumed a read's
    destination once moved in front of the wait (wrong picks in 31 of 60 encodes, no marker).  The
    wait statement now carries the sixteen values as in/out operands; THIS is the check that it keeps working: in the
    built code object, every ds_read_b64 of every kernel has its destination first read only after an s_waitcnt that
    covers it (scripts/kernel_resources.py::lds_read_hazards walks the disassembly with the counter's semantics)."""
    kr = _kernel_resources_module()
    # the checker itself: a use before the wait is seen, a use behind it is not, lgkmcnt(n) covers the reads but the last n
    bad = [(0, "ds_read_b64", "v[4:5], v1 offset:64"), (8, "v_pk_mul_f32", "v[6:7], v[4:5], v[8:9]"), (16, "s_waitcnt", "lgkmcnt(0)")]
    good = [(0, "ds_read_b64", "v[4:5], v1 offset:64"), (8, "s_waitcnt", "lgkmcnt(0)"), (12, "v_pk_mul_f32", "v[6:7], v[4:5], v[8:9]")]
    part = [(0, "ds_read_b64", "v[4:5], v1"), (8, "ds_read_b64", "v[6:7], v1 offset:8"), (16, "s_waitcnt", "lgkmcnt(1)"),
            (20, "v_pk_add_f32", "v[10:11], v[4:5], v[4:5]"), (28, "v_pk_add_f32", "v[12:13], v[6:7], v[6:7]")]
    store = [(0, "ds_read_b64", "v[4:5], v1"), (8, "ds_write_b64", "v2, v[4:5]")]
    assert len(kr.lds_read_hazards(bad)) == 1 and kr.lds_read_hazards(good) == []
    assert [h[0] for h in kr.lds_read_hazards(part)] == [28] and len(kr.lds_read_hazards(store)) == 1
    reads = 0
    for name in kr.kernel_resources():
        insts = kr.kernel_instructions(name + ">:")
        reads += sum(op == "ds_read_b64" for _, op, _ in insts)
        assert kr.lds_read_hazards(insts) == [], name
    assert reads > 1000   # (the transform is inlined into ~20 kernels, 56+ reads each)


def test_global_loads_are_consumed_behind_their_wait():
    """Round 4: screen_task's sixteen pair-spectrum loads per transform are `asm volatile("global_load_dwordx2 v, v_off,
    s[base]")` (a scalar base per element: no 64-bit vector address arithmetic) waited for by a separate
    `asm volatile("s_waitcnt vmcnt(0)")` that carries the sixteen values.  The same audit as for the LDS reads, over
    every global load of every kernel in the built code object (scripts/kernel_resources.py::vmem_read_hazards): none is
    read before a vmcnt wait that covers it."""
    kr = _kernel_resources_module()
    bad = [(0, "global_load_dwordx2", "v[4:5], v1, s[2:3]"), (8, "v_pk_mul_f32", "v[6:7], v[4:5], v[8:9]"), (16, "s_waitcnt", "vmcnt(0)")]
    good = [(0, "global_load_dwordx2", "v[4:5], v1, s[2:3]"), (8, "s_waitcnt", "vmcnt(0)"), (12, "v_pk_mul_f32", "v[6:7], v[4:5], v[8:9]")]
    part = [(0, "global_load_dwordx2", "v[4:5], v1, s[2:3]"), (8, "global_load_dwordx2", "v[6:7], v1, s[4:5]"), (16, "s_waitcnt", "vmcnt(1)"),
            (20, "v_pk_add_f32", "v[10:11], v[4:5], v[4:5]"), (28, "v_pk_add_f32", "v[12:13], v[6:7], v[6:7]")]
    other = [(0, "global_load_dwordx2", "v[4:5], v1, s[2:3]"), (8, "s_waitcnt", "lgkmcnt(0)"), (12, "v_pk_mul_f32", "v[6:7], v[4:5], v[8:9]")]
    assert len(kr.vmem_read_hazards(bad)) == 1 and kr.vmem_read_hazards(good) == []
    assert [h[0] for h in kr.vmem_read_hazards(part)] == [28] and len(kr.vmem_read_hazards(other)) == 1
    loads = asm_form = 0
    for name in kr.kernel_resources():
        insts = kr.kernel_instructions(name + ">:")
        loads += sum(op.startswith("global_load") for _, op, _ in insts)
        asm_form += sum(op == "global_load_dwordx2" and re.search(r"s\[\d+:\d+\]", args or "") is not None for _, op, args in insts)
        assert kr.vmem_read_hazards(insts) == [], name
    assert loads > 1000 and asm_form >= 16 * 9   # (sixteen scalar-base loads in each of the register screen kernels at least)


def test_scalar_bases_of_memory_instructions_are_not_fresh_from_the_valu():
    """gfx9 wants five wait states between a VALU write of an SGPR (v_readlane: an SGPR spilled into a VGPR lane comes back;
    v_readfirstlane) and a memory instruction that reads it as its base; the compiler pads its own memory instructions but
    does not look inside inline asm.  Round 4 met exactly that: the split screen's hoisted window-spectrum bases were
    spilled into VGPR lanes and read back right in front of the hand-written loads, which went out with stale bases.
    The audit (scripts/kernel_resources.py::sgpr_base_hazards) walks every kernel of the built code object."""
    kr = _kernel_resources_module()
    bad = [(0, "v_readlane_b32", "s10, v126, 15"), (4, "v_readlane_b32", "s11, v126, 16"), (8, "global_load_dwordx2", "v[4:5], v121, s[10:11]")]
    padded = [(0, "v_readlane_b32", "s10, v126, 15"), (4, "v_readlane_b32", "s11, v126, 16"), (8, "s_nop", "4"),
              (12, "global_load_dwordx2", "v[4:5], v121, s[10:11]")]
    salu = [(0, "v_readlane_b32", "s8, v126, 15"), (4, "s_add_u32", "s10, s8, 0x2000"), (8, "s_addc_u32", "s11, s9, 0"),
            (12, "global_load_dwordx2", "v[4:5], v121, s[10:11]")]
    assert len(kr.sgpr_base_hazards(bad)) == 1 and kr.sgpr_base_hazards(padded) == [] and kr.sgpr_base_hazards(salu) == []
    for name in kr.kernel_resources():
        assert kr.sgpr_base_hazards(kr.kernel_instructions(name + ">:")) == [], name


def test_bench_prices_the_screen_with_the_instruction_counts_of_the_built_code():
    """bench.py's VALU roofline multiplies transforms by the VALU instructions one 16-point thread-transform costs; that
    number is READ from the built code object (the pair loop of the kernel that ran: scripts/kernel_resources.py::
    screen_pair_loop) and bench.py's table -- its fallback where the LLVM tools are missing -- must say the same."""
    kr = _kernel_resources_module()
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for (kind, lg), want in bench.VALU_PER_THREAD_TRANSFORM.items():
        got = kr.screen_pair_loop(kind, lg)
        assert got["valu"] == want, (kind, lg, got)
        assert got["barriers"] in (4, 6) and got["packed"] > 0.75 * got["valu"]
    assert bench.valu_per_thread_transform("persistent", 11)[1] == "code object"
    # ... and `roofline.frac` is priced by the ALGORITHM's count (bench.py::min_valu_per_thread_transform): the model's numbers,
    # its relation to the built loops (never above them; the butterflies' packed additions -- which no implementation of this
    # factorisation can avoid -- are exactly the model's: idft16 = 64, idft8 pair = 48, radix-2 / radix-4 tail = 16 / 32)
    assert [bench.min_valu_per_thread_transform(lg)[0] for lg in (10, 11, 12, 13, 14)] == [294, 324, 348, 380, 404]
    for (kind, lg), adds in ((("screen", 11), 48 + 64 + 64), (("persistent", 11), 48 + 64 + 64), (("screen", 13), 64 * 3 + 16)):
        insts = kr.kernel_instructions(kr.SCREEN_KERNELS[(kind, lg)])
        loop = kr.pair_loop_range(insts)
        body = insts[loop[0]:loop[1] + 1]
        assert sum(op == "v_pk_add_f32" for _, op, _ in body) == adds, (kind, lg)
        assert bench.min_valu_per_thread_transform(lg)[0] < kr.screen_pair_loop(kind, lg)["valu"]
    f = bench.algorithmic_fractions(300000, 11, 1e-3, 403)
    assert f["frac"] < f["frac_issue"] and abs(f["frac"] / f["frac_issue"] - 324 / 403) < 5e-3 and f["frac_flops"] < f["frac"]


def test_event_grouping_matches_the_composite_key_sort():
    """group_events_by_atom (the layout of flatten_atom_dict, modules/matchingpursuit.py:61-65: groups in first-selection
    order, each in (step, batch) order) sorts 16-bit group numbers over the step-major layout; the statement it replaces --
    a stable sort of the composite key (group, step, batch) -- must give the same permutation and counts, ragged sizes
    and atoms of another rank (absent from `order`) included."""
    from mpcore import matchingpursuit as mpm
    rng = np.random.default_rng(3)
    for B, K, A in ((64, 64, 512), (3, 5, 7), (1, 1, 1), (7, 0, 4), (5, 9, 70000)):
        a = rng.integers(0, A, (B, K))
        order = mpm.first_selection_order(a) if K else []
        if B == 3:
            order = order[:-1]                      # one atom selected here belongs to another rank's list
        perm, counts = mpm.group_events_by_atom(a, order, A)
        rank_of = np.full(A, len(order), dtype=np.int64)
        if len(order):
            rank_of[np.asarray(order, dtype=np.int64)] = np.arange(len(order))
        r = rank_of[a]
        key = (r * K + np.arange(K)[None, :]) * max(B, 1) + np.arange(B)[:, None]
        assert np.array_equal(perm.numpy(), np.argsort(key.reshape(-1), kind="stable")), (B, K, A)
        assert counts == np.bincount(r.reshape(-1), minlength=len(order) + 1)[: len(order)].tolist()


def test_dictionary_levels_host_helper_matches_brute_force():
    """mp_dictionary_levels_host (host arrays only, no GPU): level[g] = 0 if group g shares no sample with an earlier
    group, else 1 + the highest level among the earlier groups it overlaps; overlap[g] flags two events of the same
    group sharing a sample."""
    rng = np.random.default_rng(12)
    for trial in range(6):
        B, N, L = int(rng.integers(1, 6)), int(rng.integers(200, 3000)), int(rng.choice([1, 16, 100, 257]))
        G = int(rng.integers(1, 40))
        counts = rng.integers(1, 7, G)
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        E = int(offsets[-1])
        ev_batch = rng.integers(0, B, E).astype(np.int64)
        ev_lag = rng.integers(0, N, E).astype(np.int64)
        level, overlap, n_levels = nat.dictionary_levels(offsets, ev_batch, ev_lag, L)
        group = np.repeat(np.arange(G), counts)
        want_level = np.zeros(G, dtype=np.int64)
        want_overlap = np.zeros(G, dtype=np.int64)
        touch = lambda a, b: ev_batch[a] == ev_batch[b] and abs(int(ev_lag[a]) - int(ev_lag[b])) < L
        for g in range(G):
            mine = np.nonzero(group == g)[0]
            for i in mine:
                for j in mine:
                    if i < j and touch(i, j):
                        want_overlap[g] = 1
            lv = 0
            for h in range(g):
                theirs = np.nonzero(group == h)[0]
                if any(touch(i, j) for i in mine for j in theirs):
                    lv = max(lv, want_level[h] + 1)
            want_level[g] = lv
        assert np.array_equal(level, want_level) and np.array_equal(overlap, want_overlap)
        assert n_levels == int(want_level.max()) + 1
    level, overlap, n_levels = nat.dictionary_levels(np.zeros(1, dtype=np.int64), np.zeros(0), np.zeros(0), 8)
    assert n_levels == 0 and level.shape == (0,)


# ---- host sanitizers ("ASan host build of the CPU restatement"; CPU only -- the GPU pool has none) ----
def _sanitizer_runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_levels_helper_under_host_sanitizers(tmp_path):
    """csrc/mplevels.inc -- the host C++ behind mp_dictionary_levels_host, working on caller-supplied index arrays --
    built on its own with -fsanitize=address,undefined and run over 400 random event sets against a brute-force
    restatement (tests/native/levels_sanitize.cpp)."""
    if _sanitizer_runtime("libasan.so") is None:
        pytest.skip("gcc has no libasan here")
    exe = str(tmp_path / "levels_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", exe, os.path.join(REPO, "tests", "native", "levels_sanitize.cpp")])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "levels_sanitize ok" in run.stdout, run.stdout + run.stderr


def test_oracle_under_host_sanitizers():
    """oracle/mp_oracle.c built with -fsanitize=address,undefined (make -C oracle asan) reproduces the reference's
    fixtures with no report: the small encode, LCN, primitive, dictionary-update and loss cases here (the 512 x 512 and
    4096 x 2048 shapes, minutes under ASan, run with `make -C oracle check-asan`)."""
    asan, ubsan = _sanitizer_runtime("libasan.so"), _sanitizer_runtime("libubsan.so")
    if asan is None or ubsan is None:
        pytest.skip("gcc has no sanitizer runtimes here")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "asan"])
    env = dict(os.environ, MP_ORACLE_LIB=os.path.join(REPO, "oracle", "_build", "libmp_oracle_asan.so"),
               LD_PRELOAD=f"{asan} {ubsan}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", "not config3_shape and not c2shape and not headline_depth and not headline_shape"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "passed" in run.stdout and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr


# ---- lazy event lists (mpcore/matchingpursuit.py::EventList / _EventStore: the drop-in surface's return structures) ----
def test_event_lists_are_lazy_and_equal_to_the_eager_structures():
    """sparse_code's `instances` / flattened list hold ranges of the packed arrays and build the reference's tuples
    (modules/matchingpursuit.py:305-321: (atom, batch, lag[1,1], gain * d[atom] [1,1,L])) only when looked at; what
    they build equals the eager construction -- keys in first-selection order (steps outer, batch inner), each list in
    (step, batch) order, flatten_atom_dict grouped by atom (:61-65)."""
    from collections import defaultdict
    import mpcore.matchingpursuit as mpm
    rng = np.random.default_rng(3)
    B, K, A, L = 5, 7, 6, 4
    atom = torch.from_numpy(rng.integers(0, A, size=(B, K)))
    lag = torch.from_numpy(rng.integers(0, 100, size=(B, K)))
    gain = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32))
    du = torch.from_numpy(rng.standard_normal((A, L)).astype(np.float32))
    want = defaultdict(list)
    for k in range(K):
        for b in range(B):
            a = int(atom[b, k])
            want[a].append((a, b, int(lag[b, k]), du[a] * gain[b, k]))
    store = mpm._EventStore(atom, lag, gain, du, torch.device("cpu"), A)
    inst = store.instances()
    assert isinstance(inst, defaultdict) and list(inst.keys()) == list(want.keys())
    assert all(isinstance(v, list) and v._store is store for v in inst.values())      # nothing built yet
    assert [len(v) for v in inst.values()] == [len(v) for v in want.values()] and store._tuples is None
    flat = mpm.flatten_atom_dict(inst)
    assert flat._store is store and len(flat) == B * K and store._tuples is None       # still lazy, as a range of the store
    pk = flat.packed
    assert store._tuples is None and pk["atom"].tolist() == [e[0] for v in want.values() for e in v]
    assert pk["batch"].tolist() == [e[1] for v in want.values() for e in v]
    assert pk["lag"].tolist() == [e[2] for v in want.values() for e in v]
    # looking at one event builds them
    ev = flat[0]
    assert store._tuples is not None and flat._store is None
    assert isinstance(ev, tuple) and isinstance(ev[0], int) and isinstance(ev[1], int)
    assert ev[2].shape == (1, 1) and ev[2].dtype == torch.int64 and ev[3].shape == (1, 1, L)
    eager = [e for v in want.values() for e in v]
    assert len(list(flat)) == len(eager)
    for got, w in zip(flat, eager):
        assert got[0] == w[0] and got[1] == w[1] and int(got[2]) == w[2] and torch.equal(got[3].view(L), w[3])
    for a, v in inst.items():
        assert [(e[0], e[1], int(e[2])) for e in v] == [(e[0], e[1], e[2]) for e in want[a]]
    # list behaviour on a fresh, unmaterialised list: concatenation either way, comparison, copy, membership, slices
    flat2 = store.flat()
    assert ([1] + flat2)[0] == 1 and len([1] + flat2) == B * K + 1
    flat3 = store.flat()
    assert len(flat3 + [2]) == B * K + 1 and (flat3 + [2])[-1] == 2 and flat3.packed is not None
    assert len(store.flat()[2:5]) == 3 and list(reversed(store.flat()))[0][1] == eager[-1][1]
    assert store.flat().copy().__class__ is list or len(store.flat().copy()) == B * K
    assert bool(store.flat()) and not bool(mpm.EventList())
    # a mutated list no longer carries packed arrays (they would not describe it)
    flat4 = store.flat()
    flat4.append(eager[0])
    assert flat4.packed is None and len(flat4) == B * K + 1
    flat5 = store.flat()
    del flat5[0]
    assert flat5.packed is None and len(flat5) == B * K - 1
    # a dict the caller changed is flattened the general way
    inst2 = store.instances()
    first = next(iter(inst2))
    inst2[first] = list(inst2[first])[:1]
    assert len(mpm.flatten_atom_dict(inst2)) == B * K - len(want[first]) + 1
    # an empty encode
    empty = mpm._EventStore(atom[:, :0], lag[:, :0], gain[:, :0], du, torch.device("cpu"), A)
    assert len(empty.instances()) == 0 and len(empty.flat()) == 0 and list(empty.flat()) == []


def test_event_lists_and_c_level_consumers():
    """Consumers of an UNMATERIALISED EventList that are implemented in C, pinned (INTEGRATION.md "Event lists").  On this
    interpreter (CPython 3.10) the exact-storage fast paths are taken for EXACT lists only (PyList_CheckExact), so a list
    subclass goes through __iter__ / __len__ / __getitem__ and everything below sees the tuples: slice assignment from
    the list, extend, tuple(), deque(), numpy.array(dtype=object), heapq.nsmallest, the json encoder, copy / deepcopy.
    The one consumer found that reads the storage of a list SUBCLASS directly is in-place heapq.heapify (PyList_Check +
    ob_item): on an unmaterialised list it orders nothing.  Below EAGER_EVENTS events sparse_code hands out materialised
    lists, for which heapify too behaves as for a plain list."""
    import collections
    import copy
    import heapq
    import json
    import mpcore.matchingpursuit as mpm
    rng = np.random.default_rng(4)
    B, K, A, L = 3, 4, 5, 4
    atom = torch.from_numpy(rng.integers(0, A, size=(B, K)))
    lag = torch.from_numpy(rng.integers(0, 50, size=(B, K)))
    gain = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32))
    du = torch.from_numpy(rng.standard_normal((A, L)).astype(np.float32))
    store = mpm._EventStore(atom, lag, gain, du, torch.device("cpu"), A)
    n = B * K
    target = [None]
    target[0:1] = store.flat()
    assert len(target) == n
    grown = []
    grown.extend(store.flat())
    assert len(grown) == n and len(tuple(store.flat())) == n and len(collections.deque(store.flat())) == n
    assert np.array(store.flat(), dtype=object).shape[0] == n
    assert len(heapq.nsmallest(2, store.flat(), key=lambda e: e[1])) == 2
    assert len(json.loads(json.dumps(store.flat(), default=lambda o: 0))) == n
    assert len(copy.copy(store.flat())) == n and len(copy.deepcopy(store.flat())) == n
    assert len(list(store.flat())) == n and len([] + store.flat()) == n and len(sum([store.flat()], [])) == n
    # the known-unsafe one: heapify works on the C storage, which an unmaterialised list has not filled yet
    by_batch = lambda ev: [(e[1], e[0], int(e[2])) for e in ev]   # noqa: E731  (tensors do not order; keys do)
    lazy = store.flat()
    heapq.heapify(lazy)
    assert lazy._store is not None and list.__len__(lazy) == 0        # nothing was ordered, nothing was built
    # materialised (what sparse_code returns for small encodes): a plain list to everybody
    keyed = mpm.EventList(by_batch(store.flat(eager=True)))
    heapq.heapify(keyed)
    assert list.__len__(keyed) == n and keyed[0] == min(by_batch(store.flat()))
    eager = store.flat(eager=True)
    assert eager._store is None and list.__len__(eager) == n and eager.packed is not None
    inst = store.instances(eager=True)
    assert all(v._store is None and list.__len__(v) == len(v) for v in inst.values())
    assert mpm.EAGER_EVENTS >= 256


def test_the_timing_only_knob_is_refused_without_its_environment_variable(monkeypatch):
    """mp_tune(MP_TUNE_LAZY_FORCE) draws the lazy screen's tile masks at random -- an instrument for timing the screen
    against the share and pattern of skipped workgroups; the events are wrong while it is set.  No product
    path can switch it on: the library refuses it unless the process says MP_ALLOW_WRONG_RESULTS=1."""
    monkeypatch.delenv("MP_ALLOW_WRONG_RESULTS", raising=False)
    with pytest.raises(nat.NativeError):
        nat.tune(nat.MP_TUNE_LAZY_FORCE, 1.5)
    nat.tune(nat.MP_TUNE_LAZY_FORCE, 0)            # switching it OFF is always allowed
    monkeypatch.setenv("MP_ALLOW_WRONG_RESULTS", "1")
    nat.tune(nat.MP_TUNE_LAZY_FORCE, 1.5)
    nat.tune(nat.MP_TUNE_LAZY_FORCE, 0)
    # ... and the same gate on MP_TUNE_CLEAR_MEMSET: hipMemsetAsync clears replayed from a hipGraph leave wrong events on
    # this runtime, so only the repro script (which sets the variable) may switch them on
    monkeypatch.delenv("MP_ALLOW_WRONG_RESULTS", raising=False)
    with pytest.raises(nat.NativeError):
        nat.tune(nat.MP_TUNE_CLEAR_MEMSET, 1)
    nat.tune(nat.MP_TUNE_CLEAR_MEMSET, 0)
    monkeypatch.setenv("MP_ALLOW_WRONG_RESULTS", "1")
    nat.tune(nat.MP_TUNE_CLEAR_MEMSET, 1)
    nat.tune(nat.MP_TUNE_CLEAR_MEMSET, 0)


def test_when_the_mirror_asks_for_the_lazy_screen():
    """_native.lazy_pays: by tile screens per step, not by batch alone (scripts/small_lazy.py)."""
    assert nat.lazy_pays(64, 512, 64) and nat.lazy_pays(24, 512, 64) and not nat.lazy_pays(16, 512, 64)
    assert nat.lazy_pays(1, 1024, 32) and nat.lazy_pays(128, 4096, 256)
    assert not nat.lazy_pays(24, 64, 16) and not nat.lazy_pays(1000, 16, 8) and not nat.lazy_pays(1000, 64, 16)   # one or two tiles: never
    assert not nat.lazy_pays(64, 512, 4)
    # ... and only where the library would look at the table: not at 8192-point transforms below 65536 cells per segment,
    # not for short segments, which run launch per step on the quarter select (csrc/mpcore.hip: short_segments)
    assert nat.lazy_pays(64, 512, 64, 32768, 512) and nat.lazy_pays(128, 4096, 256, 131072, 2048)
    assert not nat.lazy_pays(8, 1024, 32, 8192, 2048) and not nat.lazy_pays(8, 1024, 32, 4096, 1024)
    assert not nat.lazy_pays(8, 1024, 32, 2048, 512) and not nat.lazy_pays(16, 1024, 32, 2048, 512) and nat.lazy_pays(8, 1024, 32, 32768, 1024)

#!/usr/bin/env python3
"""Benchmark of the matching-pursuit hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] per GPU: 512-atom x 512-sample dictionary, 64 segments of
32768 samples, 64 MP iterations per segment.  One "step" = one mp_encode_f32 call over the
batch = 64 x 64 = 4096 segment-iterations.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank encodes its own 64 segments (configs[2] at N = 8: 512 segments), no
collective on the data path, weak scaling; rank 0 prints ONE JSON line.

value = segment-iterations/s over all ranks, inputs resident in HBM, timed between
barrier+synchronize pairs, max over ranks, on the library's default schedule for this shape (flags = 0:
MP_PATH_FFT, FFT screen + exact refinement, events bit-identical to the direct paths, re-checked every run; the
persistent form -- step 0 as separate kernels, steps 1 .. K-1 of the whole batch in ONE launch, csrc/mppersist.inc -- with
its lazy screen: the coherence table it needs, 0.45 ms, is computed by the first encode of a batch this size (from 64
segments one call pays for it: mpcore/_native.py::encode), i.e. in the first warm-up step; with --warmup 0 inside the
timed region).  `roofline` is for the dominant kernel from HIP events recorded inside the timed region on the launch
stream (the persistent launch: one span per encode; launch-per-step forms: sampled every 16th iteration, see
launch_times).  For the FFT schedule the bound is packed-fp32 VALU issue, NOT HBM: the screen's spectra are L2 /
Infinity-Cache resident, and the measured fabric traffic (`traffic`, from the committed PMC passes) over the kernel time is
reported beside it as `hbm_gbs_measured` / `frac_hbm`; the instruction count it multiplies with is READ from the built code
object (valu_per_thread_transform).  `variants` (rank 0, N = 1 only), each with its own figures:
  * the launch-per-step forms of the same schedule (one stream: the form whose per-step screen kernel has its own
    roofline; sub-batches on forked streams), the default replayed from a captured hipGraph (mpcore.EncodePlan), the two
    direct-correlation (MFMA) schedules with their rooflines;
  * what the headline does not time: fft_new_dictionary_every_call (the coherence table rebuilt inside every call),
    product_default_checked (mpcore.encode_packed: unit_norm, the checked entry point with its host synchronisation),
    sparse_code_surface (the reference's call surface: sparse_code(flatten=True) + scatter_segments), unplanted_signal
    and half_planted_signal (signals that are NOT sparse in the dictionary: marks, retries, contenders per select);
  * configs3_full_size: BASELINE configs[3] (4096 x 2048 dictionary, 128 x 131072-sample segments, K = 256) on the
    library default, one warm-up and two timed encodes, with the roofline of fft_screen_kernel<13>.
`cpu_baseline` is the CPU oracle (oracle/mp_oracle.c) timed on this host (rank 0, N = 1 only); `cpu_baseline_torch_ops`
is SURVEY.md 8(d)'s second CPU baseline -- the reference's loop in the torch CPU operators it calls itself (F.conv1d +
torch.max, oracle/mp_oracle_torch.py) at all of the host's cores and at 8 threads.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, before anything touches the
GPU), relays rank 0's JSON line and exits with the child's status; started by torch.distributed.run itself it is
one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "matching-pursuit_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpcore import _native as nat  # noqa: E402
from mpcore import dist as mpdist  # noqa: E402
from mpcore import synth  # noqa: E402

A, L, N, B_PER_GPU, K_ITERS = 512, 512, 32768, 64, 64
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0         # MI355X HBM3E peak (MI355X_MICROARCH.md)
# packed-fp32 VALU issue: 256 CUs x 4 SIMDs, 2.4 GHz, one wave64 v_pk_{fma,mul,add}_f32 per 4 cycles and SIMD
# (= 64 flop / clk / SIMD, the 157.3 TFLOP/s vector peak of MI355X_MICROARCH.md)
N_SIMD, PEAK_CLOCK_HZ, CYCLES_PER_WAVE_INSTR = 1024, 2.4e9, 4
PEAK_VALU_GINSTR = N_SIMD * PEAK_CLOCK_HZ / CYCLES_PER_WAVE_INSTR / 1e9   # 614.4 G wave-instructions/s
# VALU instructions one thread spends per 16-point share of one M-point transform in the screen's loop over atom pairs
# (spectrum product, transform, running maxima), per kernel: READ FROM THE BUILT CODE OBJECT at run time
# (scripts/kernel_resources.py::screen_pair_loop disassembles libmpcore.so and counts the loop body); this table is the
# fallback where the LLVM tools are missing, and tests/test_abi_and_host.py holds it equal to the code object.
VALU_PER_THREAD_TRANSFORM = {("persistent", 11): 403, ("screen", 11): 404, ("screen", 13): 473}
_valu_cache = {}


def valu_per_thread_transform(kind, log_m):
    """-> (VALU instructions per thread and transform, where the number comes from)."""
    key = (kind, log_m)
    if key not in _valu_cache:
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(REPO, "scripts", "kernel_resources.py"))
            kr = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(kr)
            _valu_cache[key] = (kr.screen_pair_loop(kind, log_m)["valu"], "code object")
        except Exception as e:  # noqa: BLE001  (no llvm-objdump here: the table, checked against the code object by the CPU tests)
            _valu_cache[key] = (VALU_PER_THREAD_TRANSFORM[key], f"table ({type(e).__name__})")
    return _valu_cache[key]


def min_valu_per_thread_transform(log_m):
    """The ALGORITHM's packed-fp32 instruction count per thread and transform -- what `roofline.frac` is priced by, instead of
    the built kernel's own count (valu_per_thread_transform, which a fatter kernel would raise).  One thread carries 16 of
    the M points of one atom PAIR's transform (two real correlations per complex transform); a complex add, a complex
    multiply-by-i-and-add is ONE v_pk_add_f32, a complex multiply is TWO (v_pk_mul_f32 + v_pk_fma_f32).  Counted for the
    Stockham factorisation the screen runs (csrc/mpfft.inc::ScreenCfg: M = R0 x 16 x .. x 16 [x RS]), with EVERY twiddle
    factor a ready operand (loads are not VALU; the kernels compute some factors as powers instead -- that is overhead):
      spectrum product X . P          16 complex multiplies                                             = 32
      first pass, radix 8             2 x idft8  = 2 x (2 idft4 x 8 adds + 2 constant multiplies x 2 + 8 adds)   = 56
      first pass, radix 16            idft16 = 4 idft4 x 8 + 8 constant multiplies x 2 + 4 idft4 x 8              = 80
      every later radix-16 pass       15 twiddle multiplies x 2 + idft16                                  = 110
      last pass, radix 4 / radix 2    4 x (3 twiddles x 2 + idft4 8) = 56  /  8 x (1 twiddle x 2 + 2 adds) = 32
      running maximum over the pair   16 v_max3_f32 (both atoms' values of a lag in one instruction)              = 16
    -> (count, breakdown)."""
    rs = 16 if log_m % 4 == 0 else 1 << (log_m % 4)
    small_last = rs < 8
    r0 = 16 if small_last else rs
    n16 = (log_m // 4 - 1) if (small_last or log_m % 4 == 0) else log_m // 4
    parts = {"spectrum_product": 32, "first_pass_radix_%d" % r0: 56 if r0 == 8 else 80,
             "radix_16_passes": [110] * n16, "last_pass_radix_%d" % rs: ({4: 56, 2: 32}[rs] if small_last else 0),
             "running_maximum": 16}
    total = 32 + (56 if r0 == 8 else 80) + 110 * n16 + parts["last_pass_radix_%d" % rs] + 16
    return total, parts


def flops_per_transform(log_m):
    """Textbook flop price of one M-point complex transform with its spectrum product: 5 M log2 M + 6 M."""
    m = 1 << log_m
    return 5.0 * m * log_m + 6.0 * m


def pmc_valu_fraction(kind, which):
    """VALU-busy x measured clock / 2.4 GHz of the dominant kernel from the newest committed counter pass
    (profiles/rNN_<kind>_summary.json: sq_counters_dominant_kernel[which]) -> dict or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", f"r*_{kind}_summary.json")), reverse=True):
        try:
            doc = json.load(open(f))
            c = (doc.get("sq_counters_dominant_kernel") or doc["sq_counters_fft_screen_kernel"])[which]   # (summarize_profiles / summarize_c4)
            return {"value": round(c["valu_busy_frac_of_simd_cycles"] * c["clock_GHz"] / (PEAK_CLOCK_HZ / 1e9), 4),
                    "valu_busy_frac_of_simd_cycles": c["valu_busy_frac_of_simd_cycles"], "clock_GHz": c["clock_GHz"],
                    "avg_us_under_counters": c["avg_us"] if "avg_us" in c else round(c["avg_ms"] * 1e3, 2),
                    "from": "profiles/" + os.path.basename(f)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def algorithmic_fractions(transforms, log_m, seconds, valu_code):
    """The three prices of the same launch time: the algorithm's minimal packed-fp32 instructions (-> frac), the built
    kernel's own instruction count (-> frac_issue), textbook flops (-> frac_flops).  `transforms` pair transforms of
    2^log_m points ran in `seconds`."""
    waves = (1 << log_m) // 16 // 64
    vmin, parts = min_valu_per_thread_transform(log_m)
    g_min = transforms * waves * vmin / seconds / 1e9
    g_code = transforms * waves * valu_code / seconds / 1e9
    tfl = transforms * flops_per_transform(log_m) / seconds / 1e12
    return {
        "achieved": round(g_min, 1), "frac": round(g_min / PEAK_VALU_GINSTR, 4),
        "priced_by": "minimal packed-fp32 instructions of the transform the screen runs (min_valu: derivation in "
                     "bench.py::min_valu_per_thread_transform), NOT the kernel's own count",
        "min_valu": {"per_thread_transform": vmin, "breakdown": parts, "waves_per_transform": waves},
        "frac_issue": round(g_code / PEAK_VALU_GINSTR, 4), "achieved_issue": round(g_code, 1),
        "frac_flops": round(tfl / PEAK_MFMA_F32_TFLOPS, 4), "tflops": round(tfl, 2),
        "flops_per_transform": flops_per_transform(log_m), "peak_tflops_f32_vector": PEAK_MFMA_F32_TFLOPS,
    }


PATHS = {"fft": nat.MP_PATH_FFT, "incremental": nat.MP_PATH_INCREMENTAL, "direct": nat.MP_PATH_DIRECT}


class Shape:
    """One workload: dictionary A x L, B segments of N samples, K steps; M = transform size, V = valid lags per
    transform (csrc/mpfft.inc::make_fft_geom)."""

    def __init__(self, A, L, N, B, K, name):
        self.A, self.L, self.N, self.B, self.K, self.name = A, L, N, B, K, name
        M = 256
        while M < 3 * L + 190:
            M *= 2
        self.M, self.V, self.log_m = M, (M - L + 1) // 64 * 64, M.bit_length() - 1
        self.nw_full = -(-N // self.V)


HEAD = Shape(A, L, N, B_PER_GPU, K_ITERS, "BASELINE configs[1] per GPU: 512x512 dictionary, 64 x 32768-sample segments, K=64")
C3 = Shape(4096, 2048, 131072, 128, 256, "BASELINE configs[3] at full size: 4096x2048 dictionary, 128 x 131072-sample segments, K=256")


def algorithmic_bytes_fft(sh):
    """Bytes the FFT screen kernel must read per encode, by its own algorithm: per (segment, window)
    the window spectrum (8 M bytes) and one pair spectrum per two atoms (A/2 * 8 M bytes), plus one
    8-byte key and 4-byte bound written per cell.  Full pass: ceil(N / V) windows per segment;
    every later step: one window per segment."""
    per_window = 8.0 * sh.M * (sh.A // 2 + 1)
    cells_full = (sh.N // 64) * (sh.A // 32) * 12.0
    cells_inc = ((2 * sh.L - 2) // 64 + 2) * (sh.A // 32) * 12.0
    full = sh.B * (sh.nw_full * per_window + cells_full)
    inc = sh.B * (sh.K - 1) * (per_window + cells_inc)
    return full + inc, full, inc


def algorithmic_flops(lag, path):
    """Algorithmic flop of ONE encode of this rank's batch for the correlate kernel.
    full pass: 2*A*L*N per segment (17.18 GFLOP at this shape);
    incremental launch k>=1: 2*A*L*(lags whose window the previous event touched)."""
    Bn, K = lag.shape
    full = 2.0 * A * L * N * Bn
    if path == nat.MP_PATH_DIRECT:
        return full * K, full * K, 0.0
    p = lag[:, : K - 1].astype(np.int64)
    lo = np.maximum(p - L + 1, 0)
    hi = np.minimum(p + L - 1, N - 1)
    inc = float((2.0 * A * L * (hi - lo + 1)).sum())
    return full + inc, full, inc


def timed_encodes(x, du, steps, warmup, path, flags, group, n_iters=K_ITERS, before_each=None, coherence=None):
    """`before_each`: called ahead of every encode, timed ones included (e.g. nat.clear_caches: a new dictionary every call)."""
    for _ in range(warmup):
        if before_each:
            before_each()
        out = nat.encode(x, du, n_iters, path=path, flags=flags, want_residual=True, coherence=coherence)
    torch.cuda.synchronize()
    nat.profile_read()  # drop warm-up spans
    nat.lazy_stats()
    mpdist.barrier(group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if before_each:
            before_each()
        out = nat.encode(x, du, n_iters, path=path, flags=flags, want_residual=True, coherence=coherence)
    torch.cuda.synchronize()
    mpdist.barrier(group)
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=x.device)
    t = mpdist.all_reduce_max(t, group)
    return float(t.item()), out, nat.profile_read()


def pmc_traffic(kernel="correlate"):
    """HBM (fabric) bytes per launch of the named kernel from the newest committed PMC summary
    (profiles/rNN_*_summary.json: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this
    same command -- scripts/profile_round.sh; counters cannot be read from inside the process).
    None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_summary.json")))
    # the pass that profiled THIS kernel's schedule first (the persistent passes also contain the step-0 screen launches)
    own = {"fft_persistent": "_persist_", "fft_screen": "_fft_", "correlate": "_mfma_", "c3_fft_screen": "_c4_"}.get(kernel, "")
    if kernel == "c3_fft_screen":   # the configs[3]-shape pass (scripts/c4_traffic.py under rocprofv3): its own key
        for f in reversed([f for f in files if own in os.path.basename(f)]):
            try:
                return json.load(open(f)).get("hbm_traffic_bytes_per_incremental_launch_fft_screen_kernel")
            except (OSError, ValueError):
                continue
        return None
    files = [f for f in files if own not in os.path.basename(f)] + [f for f in files if own in os.path.basename(f)]
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if k.startswith("hbm_traffic_bytes_per_launch_" + kernel):
                return v
    return None


PROF_EVERY = 16  # hipEvent spans around the kernels of iterations 0, 16, 32, 48 of every timed encode


def launch_times(prof, steps, n_iters=K_ITERS):
    """Sampled HIP-event spans -> (seconds all launches of the dominant kernel took, per-kind figures).
    An event between two kernels idles the GPU for ~10 us (scripts/prof_overhead.py: 7.5 vs 6.0 ms per
    encode with spans around every launch), so only every PROF_EVERY-th iteration of the timed region
    carries spans; the full pass (k = 0) is always one of them.  Totals are the sampled averages times
    the number of launches the timed region made."""
    ms_full, n_full = prof["corr_full"]
    ms_inc, n_inc = prof["corr_inc"]
    if n_full + n_inc == 0:
        return None
    # the FFT schedule splits the batch into sub-batches on forked streams (one launch per sub-batch and
    # step): the sampled full passes tell how many
    groups = max(1, n_full // steps) if n_inc else 1
    full_launches = steps * groups if n_inc else steps * n_iters   # MP_PATH_DIRECT: every launch is a full pass
    inc_launches = steps * (n_iters - 1) * groups if n_inc else 0
    avg_full = ms_full / max(n_full, 1)
    avg_inc = ms_inc / max(n_inc, 1)
    sec = (avg_full * full_launches + avg_inc * inc_launches) * 1e-3
    launches = full_launches + inc_launches
    return sec, {
        "launches": launches, "avg_launch_ms": round(sec * 1e3 / launches, 5), "sub_batches": groups,
        "timed_with_events": {"full_pass": n_full, "incremental": n_inc,
                              "sampling": f"iterations k % {PROF_EVERY} == 0 of each timed encode"},
        "full_pass_launches": full_launches, "full_pass_avg_ms": round(avg_full, 5),
        "incremental_launches": inc_launches, "incremental_avg_ms": round(avg_inc, 5),
    }


def roofline_fft(prof, sh, steps, lazy=None):
    """Roofline of fft_screen_kernel<log2 M> on the resource that binds it: packed-fp32 VALU issue.

    achieved = ALGORITHMIC wave-instructions (transforms RUN x waves per transform x the VALU instructions of one
    16-point thread-transform, read from the kernel's pair loop in the built code object) / launch durations; peak = 1024
    SIMDs x 2.4 GHz / 4 cycles per wave64 instruction.  With the lazy screen (`lazy` = mp_lazy_stats of the timed region)
    the (segment, tile) screens the select left out are not counted: they ran no transform.  HBM is reported beside it
    from the committed PMC passes: `traffic` bytes per launch / average launch duration = hbm_gbs_measured."""
    lt = launch_times(prof, steps, sh.K)
    if lt is None:
        return None
    sec, per_kind = lt
    valu, valu_src = valu_per_thread_transform("screen", sh.log_m)
    all_transforms = sh.B * (sh.A // 2) * (sh.nw_full + sh.K - 1) * steps
    skipped_transforms = 16 * lazy["skipped"] if lazy else 0     # a tile = 16 atom pairs, one window per incremental launch
    transforms = all_transforms - skipped_transforms
    waves = sh.M // 16 // 64
    wave_instr = transforms * waves * valu
    traffic = pmc_traffic("fft_screen" if sh is HEAD else "c3_fft_screen")
    out = {
        "bound": "valu", "peak": round(PEAK_VALU_GINSTR, 1),
        "unit": "G wave64 packed-fp32 instructions/s", "traffic": traffic,
        "kernel": f"fft_screen_kernel<{sh.log_m}> (radix-16 Stockham, packed fp32)",
        "valu_per_thread_transform": valu, "valu_count_from": valu_src,
    }
    out.update(algorithmic_fractions(transforms, sh.log_m, sec, valu))
    out["frac_from_pmc"] = pmc_valu_fraction("fft" if sh is HEAD else "c4", "incremental")
    out.update(per_kind)
    avg_s = sec / per_kind["launches"]
    if traffic is not None:
        # (the PMC passes measure an incremental launch; the full pass is one launch in K)
        inc_s = per_kind["incremental_avg_ms"] * 1e-3 if per_kind["incremental_launches"] else avg_s
        out["hbm_gbs_measured"] = round(traffic / inc_s / 1e9, 1)
        out["frac_hbm"] = round(traffic / inc_s / 1e9 / PEAK_HBM_GBS, 4)
    total, full, inc = algorithmic_bytes_fft(sh)
    survey_bytes = (8.0 * ((sh.N + sh.L) // 2 + 1) * (sh.A + 1) + 8.0 * sh.N) * sh.B * sh.K
    out.update({
        "other_kernels_avg_ms_per_iteration": round(prof["select"][0] / max(per_kind["timed_with_events"]["full_pass"]
                                                    + per_kind["timed_with_events"]["incremental"], 1), 5) if prof["select"][1] else None,
        "algorithmic_wave_instructions_per_launch": round(wave_instr / per_kind["launches"]),
        "transforms": transforms,
        "valu_floor_ms_per_launch": round(wave_instr / (PEAK_VALU_GINSTR * 1e9) * 1e3 / per_kind["launches"], 5),
        "spectra_streamed_mb_per_launch": round(total * steps / per_kind["launches"] / 1e6, 2),
        "spectra_streamed_gbs": round(total * steps / sec / 1e9, 1),
        "survey_8d_equivalent_gbs": round(survey_bytes * steps / sec / 1e9, 1),
        "note": "HBM does not bind this kernel and the north-star's >= 70 % HBM target does not apply to it: the "
                "spectra it streams (spectra_streamed_*) are L2 / Infinity-Cache hits (pair spectra shared by all "
                "segments); fabric traffic is `traffic` bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, "
                "separate rocprofv3 --pmc passes, profiles/) = hbm_gbs_measured.  By SURVEY.md 8(d)'s own per-unit "
                "bytes (the reference's full-length transform) the run would exceed HBM peak "
                "(survey_8d_equivalent_gbs): the screen does not do that work, it screens one window of dirty lags "
                "and refines exactly (DESIGN.md 4b, 6).",
    })
    if lazy:
        out["lazy_screen"] = {
            "tile_screens_without_it": lazy["decided"], "tile_screens_skipped": lazy["skipped"],
            "transforms_without_it": all_transforms, "transforms_run": transforms,
            "selects_that_decided_nothing": {k: lazy[k] for k in ("off_static", "off_contenders", "off_no_floor", "off_no_bound")},
            "contender_cells_refined": lazy["contender_cells"],
        }
    return out


WARMUP_STEPS = 2   # (main() stores --warmup here: the lazy screen's note says where the table was computed)


def roofline_persistent(prof, sh, steps):
    """Roofline of fft_persistent_kernel (steps 1 .. K-1 of the whole batch in one launch) on packed-fp32 VALU issue.

    achieved = the screens' ALGORITHMIC wave-instructions per launch (segments x A/2 pair transforms x (K-1) steps x
    waves per transform x 410) / the launch's duration (one HIP-event span per encode).  The selects that run inside
    the same launch (bound scan, exact fp32 chains of the contender quarter-cells, subtraction, next window's
    transform) are NOT counted as algorithmic work, so the fraction is the share of the launch the chip spends
    issuing screen arithmetic at peak; the step-0 screen (a separate launch) is reported beside it."""
    ms_full, n_full = prof["corr_full"]
    ms_p, n_p = prof["corr_inc"]
    if n_p == 0:
        return None
    waves = sh.M // 16 // 64
    pst = nat.persist_stats()   # (of the last launch)
    all_transforms = sh.B * (sh.A // 2) * (sh.K - 1)
    pairs_per_task = 8          # two slots of four atom pairs (csrc/mppersist.inc, M = 2048)
    transforms = pst["tasks"] * pairs_per_task   # the ones RUN: the lazy screen answers the rest from widened bounds
    valu, valu_src = valu_per_thread_transform("persistent", sh.log_m)
    valu0, _ = valu_per_thread_transform("screen", sh.log_m)     # (step 0's full pass is fft_screen_kernel)
    wave_instr = transforms * waves * valu
    avg_s = ms_p / n_p * 1e-3
    traffic = pmc_traffic("fft_persistent") if sh is HEAD else None   # (the PMC passes profiled the headline batch)
    t0 = sh.B * (sh.A // 2) * sh.nw_full
    out = {"bound": "valu", "peak": round(PEAK_VALU_GINSTR, 1), "unit": "G wave64 packed-fp32 instructions/s"}
    out.update(algorithmic_fractions(transforms, sh.log_m, avg_s, valu))
    out["frac_from_pmc"] = pmc_valu_fraction("persist", "persistent_launch") if sh is HEAD else None
    out.update({
        "traffic": traffic,
        "kernel": "fft_persistent_kernel<11,2> (queue of radix-16 Stockham screen tasks + select workers, three "
                  "workgroups per CU, one launch for steps 1 .. K-1)",
        "launches": steps, "avg_launch_ms": round(avg_s * 1e3, 5),
        "timed_with_events": {"persistent_launch": n_p, "step0_screen": n_full, "sampling": "every launch"},
        "algorithmic_wave_instructions_per_launch": wave_instr, "transforms_per_launch": transforms,
        "valu_floor_ms_per_launch": round(wave_instr / (PEAK_VALU_GINSTR * 1e9) * 1e3, 5),
        "step0_screen_avg_ms": round(ms_full / max(n_full, 1), 5),
        "step0_screen_frac_valu": round(t0 * waves * valu0 / (ms_full / max(n_full, 1) * 1e-3) / 1e9
                                        / PEAK_VALU_GINSTR, 4) if n_full else None,
        "valu_per_thread_transform": valu, "valu_count_from": valu_src,
        "other_kernels_avg_ms_per_encode": round(prof["select"][0] / max(n_p, 1), 5) if prof["select"][1] else None,
    })
    if traffic is not None:
        out["hbm_gbs_measured"] = round(traffic / avg_s / 1e9, 1)
        out["frac_hbm"] = round(traffic / avg_s / 1e9 / PEAK_HBM_GBS, 4)
    out["lazy_screen"] = {
        "transforms_without_it": all_transforms, "transforms_run": transforms,
        "tasks_answered_without_a_transform": pst["skipped"],
        "note": "the dictionary tensor's coherence table (mp_coherence_f32, ~0.45 ms) is computed by the first encode of a batch this "
                "size (mpcore/_native.py::encode: from 64 segments one call pays for it; smaller batches get it at the second "
                "encode against the same tensor): " +
                ("during warm-up" if WARMUP_STEPS >= 1 else "inside the timed region, in its first step"),
    }
    ls = nat.lazy_stats()
    out["contender_quarters_per_select"] = round(ls["contender_quarters"] / max(pst["selects"], 1) / max(steps, 1), 3)
    out["inside_the_launch"] = {
        "screen_task_us": round(pst["task_ticks"] / max(pst["tasks"], 1) / 100.0, 2), "tasks": pst["tasks"],
        "select_us": round(pst["select_ticks"] / max(pst["selects"], 1) / 100.0, 2), "selects": pst["selects"],
        "select_phase_us": pst["select_phase_us"], "error": pst["error"],
    }
    out["note"] = ("HBM does not bind this kernel and the north-star's >= 70 % HBM target does not apply to it: the spectra "
                   "the screen tasks stream are L2 / Infinity-Cache hits (4 MiB of pair spectra shared by all segments); "
                   "fabric traffic is `traffic` bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc "
                   "passes, profiles/).  The same screen arithmetic launched per step (variants."
                   "fft_launch_per_step_one_stream.roofline) is the per-kernel figure: there the kernel is ONLY the "
                   "screen; here the launch also holds every select of every step and the hand-offs between them "
                   "(DESIGN.md 4c).")
    return out


def ran_persistent():
    """True if the encodes just timed took the persistent form (mp_last_schedule)."""
    return nat.last_schedule() == -1


def roofline_from(prof, flops_one_encode, steps):
    lt = launch_times(prof, steps)
    if lt is None:
        return None
    sec, per_kind = lt
    total, _, _ = flops_one_encode
    achieved = total * steps / sec / 1e12
    out = {
        "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_MFMA_F32_TFLOPS, 4), "traffic": pmc_traffic(),
        "kernel": "correlate_persistent_kernel<32,true> (v_mfma_f32_32x32x2_f32)",
    }
    out.update(per_kind)
    out.update({
        "select_avg_ms": round(prof["select"][0] / max(prof["select"][1], 1), 5) if prof["select"][1] else None,
        "algorithmic_gflop_per_launch": round(total * steps / per_kind["launches"] / 1e9, 3),
    })
    return out


def cpu_baseline(d, x_host, gpu_sample):
    """The oracle (oracle/mp_oracle.c, the CPU restatement of the reference) on this host's cores,
    on a bounded sample of the same workload; also the in-run parity check."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mp_oracle
    mp_oracle.build()
    threads = mp_oracle.cpu_share()
    mp_oracle.set_num_threads(threads)
    du = mp_oracle.unit_norm(d)
    mp_oracle.encode(x_host[:1], du, 1)  # warm the thread pool
    t0 = time.perf_counter()
    mp_oracle.encode(x_host[:2], du, 2)
    per = (time.perf_counter() - t0) / 4
    # about 15 s of CPU work: up to 16 segments, then as many iterations as fit (<= K)
    b_s = int(min(max(round(15.0 / (per * 8)), 1), 16, x_host.shape[0]))
    k_s = int(min(max(round(15.0 / (per * b_s)), 4), K_ITERS))
    t0 = time.perf_counter()
    want = mp_oracle.encode(x_host[:b_s], du, k_s)
    dt = time.perf_counter() - t0
    atom, lag, gain = gpu_sample
    parity = {
        "segments": b_s, "steps": k_s,
        "indices_equal": bool(np.array_equal(atom[:b_s, :k_s], want["atom"]) and
                              np.array_equal(lag[:b_s, :k_s], want["lag"])),
        "gain_max_rel_err": float(np.abs(gain[:b_s, :k_s] - want["gain"]).max() / np.abs(want["gain"]).max()),
    }
    return {
        "value": round(b_s * k_s / dt, 3), "unit": "segment-iterations/s", "cores": threads, "kind": "port",
        "sample": f"{b_s} of the 64 segments x the first {k_s} of 64 iterations, "
                  f"oracle/mp_oracle.c with OpenMP on {threads} threads, {dt:.1f} s",
    }, parity


def cpu_baseline_torch(d, x_host, gpu_sample):
    """SURVEY.md 8(d)'s second CPU baseline: the reference's loop in the torch CPU operators it calls itself (F.conv1d =
    oneDNN, torch.max; oracle/mp_oracle_torch.py, test infrastructure) at all of this host's cores and at 8 threads (the
    build container's count), on the first 8 segments x 4 iterations of the same workload."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import mp_oracle
    import mp_oracle_torch as mot
    du = mot.unit_norm(d).numpy()
    cores = mp_oracle.cpu_share()
    b_s, k_s = min(8, x_host.shape[0]), 4
    atom, lag, gain = gpu_sample
    out = {}
    for label, threads in (("all_cores", cores), ("8_threads", min(8, cores))):
        rate, dt, res = mot.timed_encode(x_host[:b_s], du, k_s, threads)
        out[label] = {"value": round(rate, 3), "unit": "segment-iterations/s", "cores": threads, "kind": "port",
                      "sample": f"{b_s} of the 64 segments x the first {k_s} of 64 iterations, F.conv1d + torch.max on "
                                f"{threads} threads, {dt:.1f} s",
                      "picks_equal_gpu": bool(np.array_equal(res["atom"], atom[:b_s, :k_s]) and
                                              np.array_equal(res["lag"], lag[:b_s, :k_s]))}
    return out


def _rate(fn, steps):
    """Median-free plain timing of `steps` synchronised calls after one warm-up -> (seconds per call, last result)."""
    out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def non_ideal_variants(x, du, d, head_out, steps, group):
    """What the headline does not time: a new dictionary every call, the checked product entry
    point, the reference-shaped surface, and signals that are NOT sparse in the dictionary."""
    import mpcore
    from mpcore import encode_packed
    import mpcore.matchingpursuit as mpm
    B, K = HEAD.B, HEAD.K
    seg = B * K
    v = {}
    nat.profile_enable(0)
    # (b) the coherence table never reused: every call sees a dictionary it has not seen (clear_caches before each encode;
    #     at this batch size one call pays for the table -- mp_coherence_f32 runs inside every timed call)
    dt, out, _ = timed_encodes(x, du, steps, 1, nat.MP_PATH_FFT, 0, group, before_each=nat.clear_caches)
    v["fft_new_dictionary_every_call"] = {
        "value": round(seg * steps / dt, 2), "unit": "segment-iterations/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
        "bit_identical_to_headline": bool(all(torch.equal(p, q) for p, q in zip(out, head_out))),
        "note": "nat.clear_caches() ahead of every encode: the dictionary's coherence table (0.45 ms) is rebuilt inside each call"}
    # (c) the product's checked default (what sparse_code runs): unit_norm of the RAW dictionary on every call, the encode,
    #     the NaN scan of the marks with its host synchronisation, re-encodes of marked segments (none here)
    d_raw = torch.from_numpy(d).to(x.device)
    sec, pk = _rate(lambda: encode_packed(x, d_raw, K), steps)
    v["product_default_checked"] = {
        "value": round(seg / sec, 2), "unit": "segment-iterations/s", "ms_per_step": round(sec * 1e3, 4), "steps": steps,
        "bit_identical_to_headline": bool(torch.equal(pk["atom"], head_out[0]) and torch.equal(pk["gain"], head_out[2])),
        "note": "mpcore.encode_packed(signal, raw dictionary, K): unit_norm + encode_checked (one host sync, NaN scan)"}
    # ... and the reference's own call surface: sparse_code(flatten=True) + scatter_segments(shape, events), as
    #     modules/multibanddict.py:239-266 uses it (event tuples are built only if somebody looks at them: EventList)
    x3 = x[:, None, :]

    def surface():
        ev, scatter = mpm.sparse_code(x3, d_raw, n_steps=K, flatten=True)
        return scatter(x3.shape, ev)
    sec_s, _ = _rate(surface, steps)
    sec_i, inst = _rate(lambda: mpm.sparse_code(x3, d_raw, n_steps=K), steps)
    t0 = time.perf_counter()
    n_ev = sum(len(list(vv)) for vv in inst[0].values())     # (looking at every event: the tuples get built)
    dt_tuples = time.perf_counter() - t0
    v["sparse_code_surface"] = {
        "value": round(seg / sec_s, 2), "unit": "segment-iterations/s", "ms_per_step": round(sec_s * 1e3, 4), "steps": steps,
        "ms_sparse_code_dict_of_events": round(sec_i * 1e3, 4), "ms_materialising_all_event_tuples": round(dt_tuples * 1e3, 3),
        "events": n_ev,
        "note": "sparse_code(signal[B,1,N], d, n_steps, flatten=True) + scatter_segments(shape, events) through mpcore's "
                "mirror of modules.matchingpursuit; ratio to product_default_checked = the surface's own cost"}
    # (d) signals that are not sparse in the dictionary: no planted atoms at all (harmonic bed + noise), and a mix -- half
    #     as many planted atoms as steps -- whose maxima collapse to the bed's level mid-run.  Raw pass first (marks = NaN
    #     gains counted), then the checked entry point whose retries the rate includes.
    for name, n_events in (("unplanted_signal", 0), ("half_planted_signal", K // 2)):
        xs = torch.from_numpy(synth.make_segments(B, HEAD.N, d, n_events=n_events, seed=2002)).to(x.device)
        nat.clear_caches()
        for _ in range(2):
            raw = nat.encode(xs, du, K, path=nat.MP_PATH_FFT)
        torch.cuda.synchronize()
        nat.lazy_stats()
        raw = nat.encode(xs, du, K, path=nat.MP_PATH_FFT)
        torch.cuda.synchronize()
        pst, ls = nat.persist_stats(), nat.lazy_stats()
        marked = int(torch.isnan(raw[2]).any(dim=1).sum())
        sec_raw, _ = _rate(lambda: nat.encode(xs, du, K, path=nat.MP_PATH_FFT), steps)
        sec_chk, chk = _rate(lambda: nat.encode_checked(xs, du, K), steps)
        inc = nat.encode(xs, du, K, path=nat.MP_PATH_INCREMENTAL)
        torch.cuda.synchronize()
        rdb = 20 * torch.log10(chk[3].norm(dim=-1) / xs.norm(dim=-1))
        v[name] = {
            "value": round(seg / sec_chk, 2), "unit": "segment-iterations/s", "ms_per_step": round(sec_chk * 1e3, 4), "steps": steps,
            "planted_events_per_segment": n_events,
            "unchecked_encode_ms": round(sec_raw * 1e3, 4), "segments_marked_by_the_first_pass": marked,
            "share_of_time_in_retries": round(max(0.0, 1.0 - sec_raw / sec_chk), 4),
            "contender_quarters_per_select": round(ls["contender_quarters"] / max(pst["selects"], 1), 3),
            "screen_tasks_skipped_by_the_lazy_screen": pst["skipped"], "screen_tasks_run": pst["tasks"],
            "identical_to_incremental_schedule": bool(all(torch.equal(p, q) for p, q in zip(chk[:3], inc[:3]))),
            "residual_db_mean": round(float(rdb.mean()), 3),
            "note": "encode_checked (the product default): marked segments are re-encoded without the lazy screen, then on "
                    "the incremental schedule; `value` includes those retries"}
    # (e) twice the batch: where the launch is bound by its screen tasks' throughput instead of one segment's chain of steps
    #     (64 segments keep ~77 % of the screen workers busy), with the persistent launch's own roofline
    sh2 = Shape(HEAD.A, HEAD.L, HEAD.N, 2 * HEAD.B, HEAD.K, "the headline dictionary, 128 segments")
    x2 = torch.cat([x, torch.from_numpy(synth.make_segments(HEAD.B, HEAD.N, d, n_events=3 * HEAD.K, seed=1002,
                                                            first_index=100000)).to(x.device)])
    nat.profile_enable(PROF_EVERY)
    dt2, out2, prof2 = timed_encodes(x2, du, steps, 1, nat.MP_PATH_FFT, 0, group)
    v["fft_128_segments"] = {
        "value": round(sh2.B * sh2.K * steps / dt2, 2), "unit": "segment-iterations/s", "ms_per_step": round(dt2 / steps * 1e3, 4),
        "steps": steps, "first_64_segments_bit_identical_to_headline": bool(all(torch.equal(p[:HEAD.B], q) for p, q in zip(out2, head_out))),
        "roofline": roofline_persistent(prof2, sh2, steps) if ran_persistent() else None}
    nat.profile_enable(PROF_EVERY)
    return v


def configs3_variant(dev):
    """BASELINE configs[3] at full size on the library default (FFT screen launch per step, fused whole-cell select with
    block summaries, lazy screen by tile mask): one warm-up encode (which also builds the dictionary's coherence table),
    two timed ones on the default (four sub-batches on forked streams); the roofline of fft_screen_kernel<13> from the HIP
    events of one more encode on one stream, where its launches do not overlap."""
    sh = C3
    t0 = time.perf_counter()
    d = synth.make_dictionary(sh.A, sh.L, seed=4000)
    x = torch.empty(sh.B, sh.N, device=dev)
    for b0 in range(0, sh.B, 32):   # E = 3 K planted events per segment
        x[b0:b0 + 32] = torch.from_numpy(synth.make_segments(32, sh.N, d, n_events=3 * sh.K, seed=4001, first_index=b0)).to(dev)
    du = nat.unit_norm(torch.from_numpy(d).to(dev))
    gen_s = time.perf_counter() - t0
    nat.clear_caches()
    steps = 2
    nat.profile_enable(0)
    dt, out, _ = timed_encodes(x, du, steps, 1, nat.MP_PATH_FFT, 0, None, n_iters=sh.K)
    schedule = nat.last_schedule()
    # the screen kernel's own durations: the same encode once more on ONE stream (the default cuts this batch into four
    # sub-batches whose launches overlap -- their spans would add up to several times the wall clock)
    nat.profile_enable(PROF_EVERY)
    one_dt, one, prof = timed_encodes(x, du, 1, 0, nat.MP_PATH_FFT, nat.MP_FLAG_NO_OVERLAP, None, n_iters=sh.K,
                                      coherence=nat.cached_coherence(du))
    ls = nat.lazy_stats()
    nat.profile_enable(0)
    plain_dt, plain, _ = timed_encodes(x, du, 1, 0, nat.MP_PATH_FFT, 0, None, n_iters=sh.K, coherence=False)
    nat.lazy_stats()
    marked = int(torch.isnan(out[2]).any(dim=1).sum())
    inc = nat.encode(x[:4], du, 16, path=nat.MP_PATH_INCREMENTAL)
    torch.cuda.synchronize()
    rec = torch.zeros_like(x)
    nat.scatter(out[0], torch.arange(sh.B, device=dev)[:, None].expand(sh.B, sh.K), out[1], out[2], du, rec)
    rt = float((rec + out[3] - x).abs().max())
    rdb = 20 * torch.log10(out[3].norm(dim=-1) / x.norm(dim=-1))
    nat.clear_caches()
    res = {
        "workload": sh.name, "value": round(sh.B * sh.K * steps / dt, 2), "unit": "segment-iterations/s",
        "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": 1,
        "schedule": schedule, "segments_marked": marked,
        "one_stream": {"value": round(sh.B * sh.K / one_dt, 2), "ms_per_step": round(one_dt * 1e3, 3),
                       "bit_identical": bool(all(torch.equal(p, q) for p, q in zip(out, one)))},
        "without_the_lazy_screen": {"value": round(sh.B * sh.K / plain_dt, 2), "ms_per_step": round(plain_dt * 1e3, 3),
                                    "bit_identical": bool(all(torch.equal(p, q) for p, q in zip(out, plain)))},
        "first_4_segments_x_16_steps_equal_incremental_mfma": bool(torch.equal(inc[0], out[0][:4, :16]) and
                                                                   torch.equal(inc[1], out[1][:4, :16]) and
                                                                   torch.equal(inc[2], out[2][:4, :16])),
        "round_trip_max_err": rt, "residual_db_mean": round(float(rdb.mean()), 3),
        "inputs_generated_s": round(gen_s, 1),
        "roofline": roofline_fft(prof, sh, 1, lazy=ls),   # (of the one-stream encode above)
        "note": "coherence table (mp_coherence_f32, ~115 ms for 4096 x 2048) built once, in the warm-up encode; SURVEY.md 8(d)'s "
                "HBM roofline for this config prices the reference's full-length transforms (2.18 GB per segment-iteration): "
                "by it this run would move " + f"{2.181e9 * sh.B * sh.K * steps / dt / 1e12:.0f}" + " TB/s -- the screen does not do "
                "that work (one 8192-point window of dirty lags per step, exact refinement); its own bound is VALU issue",
    }
    del x, rec, out, plain
    torch.cuda.empty_cache()
    return res


def next_rows_variants(dev, steps):
    """SURVEY.md 8(f)'s rows on the driver line (they lived in scripts/ only): the multiband model on
    the band table of experiments/archive/e_2023_3_8/experiment.py:351-359, the streaming encode of a 2^20-sample
    recording at a 50 % hop (iterativedecomposition.py:275-319's shape), one training step of the mp.py model at BASELINE
    configs[4]'s shape (512 x 512, 8 x 32768 samples, K = 32, STFT(2048, 256) iterative loss, Adam).  One line each."""
    import mpcore  # noqa: F401
    from mpcore import multibanddict as mb
    from mpcore import streaming
    from mpcore.model import MatchingPursuit, train_step
    v = {}
    reps = max(2, min(steps, 5))
    # ---- multiband: seven bands 512 .. 32768 samples, 1024 atoms of band / 4 samples each, 32 steps per band, B = 8 ----
    Bm, n_atoms, msteps, n = 8, 1024, 32, 2 ** 15
    sizes = [512, 1024, 2048, 4096, 8192, 16384, 32768]
    specs = [mb.BandSpec(sz, n_atoms, sz // 4, device=dev, signal_samples=n, is_lowest_band=(sz == 512)) for sz in sizes]
    model = mb.MultibandDictionaryLearning(specs, n_samples=n)
    rng = np.random.default_rng(5)
    xm = torch.from_numpy(rng.standard_normal((Bm, 1, n)).astype(np.float32)).to(dev)
    t = torch.arange(n, device=dev)[None, None, :]
    for f0 in (60., 250., 900., 2500., 6000.):   # (structure in every band: decaying sinusoids at random onsets)
        on = int(rng.integers(0, n // 2))
        xm = xm * 0.97 + 3.0 * torch.sin(2 * np.pi * f0 / 22050. * t) * torch.exp(-(t - on).clamp(min=0) / 3000.) * (t >= on)
    model.encode(xm, msteps)
    sec_e, _ = _rate(lambda: model.encode(xm, msteps), reps)
    sec_r, rec = _rate(lambda: model.recon(xm, msteps), reps)
    sec_l, _ = _rate(lambda: model.learn(xm, msteps), 2)
    v["multiband_e_2023_3_8"] = {
        "value": round(7 * Bm * msteps / sec_e, 1), "unit": "band-segment-iterations/s", "encode_ms": round(sec_e * 1e3, 2),
        "recon_ms": round(sec_r * 1e3, 2), "learn_ms": round(sec_l * 1e3, 2), "batch": Bm, "steps_per_band": msteps,
        "bands": [[sz, n_atoms, sz // 4] for sz in sizes],
        "residual_energy_share": round(float(((xm - rec[0]) ** 2).sum() / (xm ** 2).sum()), 4),
        "note": "mpcore.multibanddict.MultibandDictionaryLearning.encode / recon / learn (dictionary_learning_step per band) "
                "on noise + decaying sinusoids; atoms of 8192 samples run the split 2^14-point transforms"}
    del model, specs, xm, rec
    # ---- streaming: one 2^20-sample recording per row, windows of 32768 at hop 16384, even / odd batches ----
    Bs, T, window, hop, Ks = 2, 2 ** 20, 32768, 16384, 64
    d = synth.make_dictionary(A, L, seed=1000)
    # (the recording: 32 headline-like segments end to end -- 3 K planted events, a note bed and noise per 32768 samples)
    audio = torch.from_numpy(synth.make_segments(Bs * (T // window), window, d, n_events=3 * Ks, seed=3003).reshape(Bs, T)).to(dev)
    dd = torch.from_numpy(d).to(dev)
    W = streaming.n_windows(T, window, hop)
    sec_s, code = _rate(lambda: streaming.encode_streaming(audio, dd, window, hop, Ks, order="even_odd"), reps)
    # the same number of window-encodes as two plain batches (no residual hand-over, no stacking / scatter of windows)
    seg = torch.stack([audio[:, w * hop:w * hop + window] for w in range(0, W - 1, 2)], dim=1).reshape(-1, window).contiguous()
    du_s = nat.unit_norm(dd)
    sec_plain, _ = _rate(lambda: (nat.encode_checked(seg, du_s, Ks), nat.encode_checked(seg, du_s, Ks)), reps)
    rt = float((streaming.decode_streaming(code) + code.residual - audio).abs().max())
    v["streaming_even_odd"] = {
        "value": round(Bs * W * Ks / sec_s, 1), "unit": "segment-iterations/s", "ms_per_recording_batch": round(sec_s * 1e3, 2),
        "recordings": Bs, "samples": T, "window": window, "hop": hop, "windows": W, "iterations_per_window": Ks,
        "two_plain_batches_of_the_same_size_ms": round(sec_plain * 1e3, 2),
        "share_lost_to_window_hand_over": round(max(0.0, 1.0 - sec_plain / sec_s), 4),
        "round_trip_max_err": rt,
        "note": "mpcore.streaming.encode_streaming(order='even_odd'): windows of equal parity as one batch each, the second on "
                "what the first left; hand-over = stacking the windows, writing residuals back, the odd batch waiting for the even one"}
    del audio, seg, code
    # ---- configs[4]: one training step of the mp.py model at its own shape ----
    A5, L5, N5, B5, K5 = 512, 512, 32768, 8, 32
    torch.manual_seed(0)
    m5 = MatchingPursuit(A5, L5, N5, K5).to(dev)
    d5 = synth.make_dictionary(A5, L5, seed=5000)
    with torch.no_grad():
        m5.atoms.copy_(torch.from_numpy(d5)[None].to(dev) * 0.05)
    opt = torch.optim.Adam(m5.parameters(), lr=1e-3)
    x5 = torch.from_numpy(synth.make_segments(B5, N5, d5, n_events=3 * K5, seed=5001)).to(dev)[:, None, :]
    for _ in range(2):
        train_step(m5, opt, x5, ("stft", 2048, 256))
    sec_t, loss = _rate(lambda: train_step(m5, opt, x5, ("stft", 2048, 256)), max(reps, 4))
    atoms = m5.atoms[0].detach()
    sec_f, _ = _rate(lambda: nat.encode(x5[:, 0], atoms, K5, path=nat.default_path(L5), conv_model=True), max(reps, 4))
    v["config5_train_step"] = {
        "value": round(B5 * K5 / sec_t, 1), "unit": "segment-iterations/s", "ms_per_step": round(sec_t * 1e3, 3),
        "analysis_loop_only_ms": round(sec_f * 1e3, 3), "loss": round(float(loss), 4),
        "shape": {"atoms": A5, "atom_samples": L5, "segments": B5, "samples": N5, "iterations": K5, "stft": [2048, 256]},
        "note": "mpcore.model.train_step: mp_encode_conv_f32 (mp.py:58-65), event form of iterative_loss under "
                "stft(x, 2048, 256, pad=True) (mp.py:68-70, 102-104), mp_conv_model_backward_f32, Adam; one rank: the "
                "gradient all-reduce is the identity (world size 1)"}
    del m5, x5
    torch.cuda.empty_cache()
    return v


def lcn_variant(x, du, steps):
    """The local-contrast-norm schedule (sparse_code(local_contrast_norm=True), modules/matchingpursuit.py:284-294; used
    with 512 steps by experiments/archive/e_2023_7_20/experiment.py:31-41) at the headline shape: 64 x 32768, K = 64."""
    B, K = HEAD.B, HEAD.K
    sec, out = _rate(lambda: nat.encode_lcn(x, du, K), max(1, min(steps, 3)))
    plain = nat.encode(x, du, K, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    # the schedule's own roofline: it keeps the dense A x N map (in cell order) and recomputes, EXACTLY, every map value an
    # event changes -- 2 A L flop per (atom, lag) on the fp32 matrix core; no screen can stand in for the map, because every
    # one of the 81 values under a box enters the selection rule.  Algorithmic flop = full pass + the dirty lags of K - 1 steps.
    total, full, inc = algorithmic_flops(out[1].cpu().numpy(), nat.MP_PATH_INCREMENTAL)
    tfl = total / sec / 1e12
    res = {"value": round(B * K / sec, 1), "unit": "segment-iterations/s", "ms_per_step": round(sec * 1e3, 3),
           "share_of_picks_that_differ_from_the_plain_rule": round(float(((out[0] != plain[0]) | (out[1] != plain[1])).float().mean()), 4),
           "residual_db_mean": round(float((20 * torch.log10(out[3].norm(dim=-1) / x.norm(dim=-1))).mean()), 3),
           "roofline": {"bound": "mfma", "achieved": round(tfl, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tfl / PEAK_MFMA_F32_TFLOPS, 4),
                        "algorithmic_gflop_per_encode": round(total / 1e9, 1), "full_pass_gflop": round(full / 1e9, 1),
                        "incremental_gflop": round(inc / 1e9, 1),
                        "mfma_floor_ms": round(total / (PEAK_MFMA_F32_TFLOPS * 1e12) * 1e3, 2),
                        "box_sum_adds_per_encode": int(81 * B * HEAD.A * (HEAD.N + (K - 1) * (2 * HEAD.L + 128))),
                        "kernels": "correlate_persistent_kernel<32,true,true> (cells into the cell-order map), lcn_keys_kernel, "
                                   "lcn_select_subtract_kernel; per-kernel times: profiles/r04_lcn_kernel_stats.csv",
                        "note": "whole-encode time against the matrix-core floor of the exact map update; the LCN pass (81 "
                                "sequential fp32 additions per map value, packed two sums per instruction) and the select "
                                "run between the correlates, not beside them"}}
    return res


def multi_rank_fields(x, d, dt_max, steps, rank, world, dev, group):
    """N > 1 only: what the first real multi-GPU lease should yield in one run -- how many ranks took
    part and on which devices, every rank's own encode rate (its timed region, before the max over ranks), and the two
    collectives of the surface at their own shapes: dictionary_learning_step with its per-level [atoms in level, L]
    all-reduce over all ranks' segments (configs[2]'s batch: 64 segments per rank), and one training step of the mp.py
    model with its [A, L] gradient all-reduce (configs[4]: 8 segments per rank).  A failure in either is reported as a
    string, never raised: the encode line above stands on its own."""
    import torch.distributed as tdist
    import mpcore.matchingpursuit as mpm
    from mpcore.model import MatchingPursuit, train_step
    out = {}
    mine = torch.tensor([float(rank), float(torch.cuda.current_device()), float(B_PER_GPU * K_ITERS * steps)], dtype=torch.float64, device=dev)
    # (a rank's own elapsed time is not kept by timed_encodes -- its barrier makes them equal; the per-rank rate is from one more
    #  un-barriered encode batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    du = nat.unit_norm(torch.from_numpy(d).to(dev))
    for _ in range(steps):
        nat.encode(x, du, K_ITERS, path=nat.MP_PATH_FFT)
    torch.cuda.synchronize()
    own = torch.tensor([B_PER_GPU * K_ITERS * steps / (time.perf_counter() - t0)], dtype=torch.float64, device=dev)
    try:
        rows, _ = mpdist.gather_batch(torch.cat([mine, own])[None, :], group)
        rows = rows.cpu().tolist()
        out["ranks_seen"] = len({int(r[0]) for r in rows})
        out["devices_seen"] = sorted({int(r[1]) for r in rows})
        out["per_rank_seg_it_s"] = [round(r[3], 1) for r in sorted(rows)]
    except Exception as e:  # noqa: BLE001
        out["ranks_seen"] = f"all_gather failed: {type(e).__name__}: {e}"[:200]
    try:
        d_raw = torch.from_numpy(d).to(dev)
        x3 = x[:, None, :]
        mpm.dictionary_learning_step(x3, d_raw, n_steps=K_ITERS, process_group=tdist.group.WORLD)
        torch.cuda.synchronize()
        mpdist.barrier(group)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            d_new = mpm.dictionary_learning_step(x3, d_raw, n_steps=K_ITERS, process_group=tdist.group.WORLD)
        torch.cuda.synchronize()
        mpdist.barrier(group)
        ms = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev)
        ms = mpdist.all_reduce_max(ms, group)
        chk = torch.tensor([float(d_new.double().sum())], dtype=torch.float64, device=dev)
        hi, lo = mpdist.all_reduce_max(chk.clone(), group), -mpdist.all_reduce_max(-chk.clone(), group)
        out["dls_by_levels_ms"] = round(float(ms.item()), 3)
        out["dls_by_levels"] = {"segments_all_ranks": world * B_PER_GPU, "iterations": K_ITERS,
                                "same_dictionary_on_every_rank": bool(lo.item() == hi.item()),
                                "note": "dictionary_learning_step(process_group=WORLD): encode of the rank's shard + one "
                                        "[atoms in level, L] fp64 all-reduce per dependency level"}
    except Exception as e:  # noqa: BLE001
        out["dls_by_levels_ms"] = f"failed: {type(e).__name__}: {e}"[:300]
    try:
        A5, L5, N5, B5, K5 = 512, 512, 32768, 8, 32
        torch.manual_seed(0)
        m5 = MatchingPursuit(A5, L5, N5, K5).to(dev)
        d5 = synth.make_dictionary(A5, L5, seed=5000)
        with torch.no_grad():
            m5.atoms.copy_(torch.from_numpy(d5)[None].to(dev) * 0.05)
        opt = torch.optim.Adam(m5.parameters(), lr=1e-3)
        x5 = torch.from_numpy(synth.make_segments(B5, N5, d5, n_events=3 * K5, seed=5001, first_index=rank * B5)).to(dev)[:, None, :]
        for _ in range(2):
            train_step(m5, opt, x5, ("stft", 2048, 256), tdist.group.WORLD)
        torch.cuda.synchronize()
        mpdist.barrier(group)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            train_step(m5, opt, x5, ("stft", 2048, 256), tdist.group.WORLD)
        torch.cuda.synchronize()
        mpdist.barrier(group)
        ms = mpdist.all_reduce_max(torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device=dev), group)
        chk = m5.atoms.detach().double().sum().reshape(1)
        hi, lo = mpdist.all_reduce_max(chk.clone(), group), -mpdist.all_reduce_max(-chk.clone(), group)
        out["config5_train_step_ms"] = round(float(ms.item()), 3)
        out["config5_train_step"] = {"segments_all_ranks": world * B5, "seg_it_s_all_ranks": round(world * B5 * K5 / (float(ms.item()) * 1e-3), 1),
                                     "same_atoms_on_every_rank": bool(lo.item() == hi.item()),
                                     "note": "mp.py model at BASELINE configs[4]'s shape, 8 segments per rank; one all-reduce of the "
                                             "[512, 512] gradient per step (mpcore.model.all_reduce_gradients), identical Adam on every rank"}
    except Exception as e:  # noqa: BLE001
        out["config5_train_step_ms"] = f"failed: {type(e).__name__}: {e}"[:300]
    return out


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as ONE child process tree
    (this process has not touched the GPU and never will), relay rank 0's JSON line, return the child's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode
    if len(json_lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(json_lines)}", file=sys.stderr)
        return 1
    print(json_lines[0], flush=True)
    return 0


def dry_run(args):
    """Launcher / rendezvous rehearsal without a GPU (tests/test_distributed_cpu.py): the ranks meet on gloo, go
    through the bench's barrier and max-over-ranks reduction, rank 0 prints one line.  Measures nothing."""
    rank, world, _ = mpdist.init_from_env(backend=args.backend or "gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    mpdist.barrier(None)
    t = mpdist.all_reduce_max(torch.tensor([1.0 + rank], dtype=torch.float64), None)
    lo, hi = mpdist.shard_range(world * B_PER_GPU, rank, world)
    n = mpdist.all_reduce_sum(torch.tensor([float(hi - lo)], dtype=torch.float64), None)
    if rank == 0:
        print(json.dumps({"dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks": float(t.item()), "segments_all_ranks": int(n.item())}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--path", default="fft", choices=list(PATHS))
    ap.add_argument("--flags", type=int, default=0,
                    help="MP_FLAG_* bits for the timed region.  Default 0: whatever the library picks for the shape "
                         "(MP_PATH_FFT at 64 segments: the persistent form); the launch-per-step forms "
                         "(MP_FLAG_NO_OVERLAP = 4096: one stream; MP_FLAG_FFT_NO_PERSISTENT = 131072: sub-batches) are "
                         "reported under variants.")
    ap.add_argument("--no-variants", action="store_true", help="skip the direct-path variant leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-configs3", action="store_true", help="skip the BASELINE configs[3] full-size variant (~10 s)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo on a one-GPU box)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal on CPU: no encode, nothing measured")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: become the launcher.  Nothing above has initialised the GPU (imports only).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)

    if args.share_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = mpdist.init_from_env(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    group = None

    d = synth.make_dictionary(A, L, seed=1000)
    x_host = synth.make_segments(B_PER_GPU, N, d, n_events=3 * K_ITERS, seed=1002,
                                 first_index=rank * B_PER_GPU)
    x = torch.from_numpy(x_host).to(dev)
    du = nat.unit_norm(torch.from_numpy(d).to(dev))
    torch.cuda.synchronize()

    # (the headline's timed region: spans around the correlate / screen launches only -- the dominant kernel's durations
    #  are what the roofline needs; the selects' spans cost four more events per encode, ~1 % of it)
    nat.profile_enable(PROF_EVERY, correlate_only=True)
    path = PATHS[args.path]
    # One-time initialisation, before the W warm-up steps and not counted among them: the first encodes of a process load
    # code objects (the library's, and torch's for the handful of tensor operators on the host path -- the coherence
    # cache's device-side comparison alone costs ~45 ms the first time it runs, at the SECOND encode), create the stream
    # pool and build the dictionary's coherence table.  Without this a run with --warmup 1 times that start-up.
    # ... and bring the GPU to the clocks it holds under this load: timed straight after three encodes the K steps read 1 %
    # (20 steps) to 3 % (5 steps) low against the same schedule measured later in the run (MP_BENCH_INIT_ENCODES=3 shows it).
    init_encodes = int(os.environ.get("MP_BENCH_INIT_ENCODES", "40"))
    for _ in range(init_encodes):
        nat.encode(x, du, K_ITERS, path=path, flags=args.flags, want_residual=True)
    torch.cuda.synchronize()
    global WARMUP_STEPS
    WARMUP_STEPS = args.warmup
    dt, out, prof = timed_encodes(x, du, args.steps, args.warmup, path, args.flags, group)
    atom, lag, gain, residual = [t.cpu().numpy() for t in out]
    seg_its = world * B_PER_GPU * K_ITERS * args.steps
    if path == nat.MP_PATH_FFT:
        roof = (roofline_persistent if ran_persistent() else roofline_fft)(prof, HEAD, args.steps)
    else:
        roof = roofline_from(prof, algorithmic_flops(lag, path), args.steps)
    rdb = 20 * np.log10(np.linalg.norm(residual, axis=-1) / np.linalg.norm(x_host, axis=-1))

    line = {
        "metric": "MP iterations/sec (32768-samp seg, 512x512 dict, K=64)",
        "value": round(seg_its / dt, 2), "unit": "segment-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] per GPU: 512x512 dictionary, 64 x 32768-sample segments, K=64"
                        + (f" (x{world} ranks = configs[2])" if world > 1 else ""),
            "path": args.path, "global_batch": world * B_PER_GPU, "n_samples": N, "n_atoms": A,
            "atom_samples": L, "iterations": K_ITERS, "segment_iterations_per_step": world * B_PER_GPU * K_ITERS,
            "parallelism": f"segments sharded over {world} rank(s), no data-path collective",
            "untimed_initialisation_encodes": init_encodes,
        },
        "roofline": roof,
        "residual_db_mean": round(float(rdb.mean()), 4),
    }

    if rank == 0 and world == 1:
        if not args.no_variants:
            line["variants"] = {}
            for name, other, vflags in (("fft_launch_per_step_one_stream", nat.MP_PATH_FFT, nat.MP_FLAG_NO_OVERLAP),
                                        ("fft_launch_per_step_sub_batches", nat.MP_PATH_FFT, nat.MP_FLAG_FFT_NO_PERSISTENT),
                                        ("fft_library_default", nat.MP_PATH_FFT, 0),
                                        ("incremental_direct_mfma", nat.MP_PATH_INCREMENTAL, nat.MP_FLAG_NO_OVERLAP),
                                        ("direct_full_recompute_mfma", nat.MP_PATH_DIRECT, nat.MP_FLAG_NO_OVERLAP)):
                if other == path and vflags == args.flags:
                    continue
                vsteps = 1 if other == nat.MP_PATH_DIRECT else max(1, min(args.steps, 3))
                if other == nat.MP_PATH_FFT:
                    vsteps = args.steps
                vdt, vout, vprof = timed_encodes(x, du, vsteps, 1, other, vflags, group)
                vlag = vout[1].cpu().numpy()
                same = all(torch.equal(p, q) for p, q in zip(vout, out))
                if other == nat.MP_PATH_FFT:
                    vroof = (roofline_persistent if ran_persistent() else roofline_fft)(vprof, HEAD, vsteps)
                else:
                    vroof = roofline_from(vprof, algorithmic_flops(vlag, other), vsteps)
                line["variants"][name] = {
                    "value": round(B_PER_GPU * K_ITERS * vsteps / vdt, 2), "unit": "segment-iterations/s",
                    "ms_per_step": round(vdt / vsteps * 1e3, 4), "steps": vsteps,
                    "bit_identical_to_headline": bool(same), "roofline": vroof,
                }
            line["variants"].update(non_ideal_variants(x, du, d, out, args.steps, group))
            # the library default once more, replayed from a captured hipGraph (mpcore.EncodePlan)
            nat.profile_enable(0)  # (no timing events inside the captured graph)
            plan = nat.EncodePlan(B_PER_GPU, N, du, K_ITERS, path=nat.MP_PATH_FFT)
            plan(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pout = plan(x)
            torch.cuda.synchronize()
            pdt = time.perf_counter() - t0
            line["variants"]["fft_library_default_hipgraph_replay"] = {
                "value": round(B_PER_GPU * K_ITERS * args.steps / pdt, 2), "unit": "segment-iterations/s",
                "ms_per_step": round(pdt / args.steps * 1e3, 4), "steps": args.steps,
                "bit_identical_to_headline": bool(all(torch.equal(p, q) for p, q in zip(pout, out))),
                "note": "includes the device copy of the batch into the plan's static input",
            }
            if not args.no_configs3:
                nat.profile_enable(PROF_EVERY)
                line["variants"]["configs3_full_size"] = configs3_variant(dev)
            nat.profile_enable(0)
            line["variants"]["lcn_headline"] = lcn_variant(x, du, args.steps)
            line["variants"].update(next_rows_variants(dev, args.steps))
        # what SURVEY.md 8(d) prices -- the DIRECT schedule's 17.18 GFLOP per segment-iteration on the matrix core -- and
        # configs[3] go INTO `roofline` (the part of the line the driver's record keeps): a short direct encode if the
        # variants leg did not run one
        direct = (line.get("variants") or {}).get("direct_full_recompute_mfma")
        if direct is None and path != nat.MP_PATH_DIRECT:
            nat.profile_enable(PROF_EVERY)
            ddt, dout, dprof = timed_encodes(x, du, 1, 0, nat.MP_PATH_DIRECT, nat.MP_FLAG_NO_OVERLAP, group)
            direct = {"value": round(B_PER_GPU * K_ITERS / ddt, 2), "ms_per_step": round(ddt * 1e3, 4), "steps": 1,
                      "bit_identical_to_headline": bool(all(torch.equal(p, q) for p, q in zip(dout, out))),
                      "roofline": roofline_from(dprof, algorithmic_flops(dout[1].cpu().numpy(), nat.MP_PATH_DIRECT), 1)}
            nat.profile_enable(0)
        if direct is not None and roof is not None and direct.get("roofline"):
            dr = direct["roofline"]
            roof["direct_path"] = {
                "seg_it_s": direct["value"], "tflops": dr["achieved"], "frac": dr["frac"], "bound": "mfma",
                "peak_tflops": dr["peak"], "kernel": dr["kernel"], "avg_launch_ms": dr["avg_launch_ms"],
                "algorithmic_gflop_per_segment_iteration": round(2.0 * A * L * N / 1e9, 3),
                "traffic": dr.get("traffic"), "bit_identical_to_headline": direct["bit_identical_to_headline"],
                "note": "MP_PATH_DIRECT: every cell of the A x N plane recomputed every step on v_mfma_f32_32x32x2_f32 -- the "
                        "schedule that does SURVEY 8(d)'s 2 A L N flop per segment-iteration; the headline's FFT screen + exact "
                        "refinement produces the same events bit for bit with ~1/165 of that work"}
        c3 = (line.get("variants") or {}).get("configs3_full_size")
        if c3 is not None and roof is not None and c3.get("roofline"):
            cr = c3["roofline"]
            roof["configs3"] = {
                "workload": c3["workload"], "seg_it_s": c3["value"], "ms_per_encode": c3["ms_per_step"],
                "kernel": cr["kernel"], "bound": "valu", "frac": cr["frac"], "frac_issue": cr["frac_issue"],
                "frac_flops": cr["frac_flops"], "frac_from_pmc": cr.get("frac_from_pmc"), "tflops": cr["tflops"],
                "min_valu_per_thread_transform": cr["min_valu"]["per_thread_transform"],
                "valu_per_thread_transform": cr["valu_per_thread_transform"],
                "avg_launch_ms": cr["avg_launch_ms"], "traffic": cr["traffic"], "frac_hbm": cr.get("frac_hbm"),
                "survey_8d_equivalent_gbs": cr.get("survey_8d_equivalent_gbs"),
                "timed_on": "one-stream encode (the default's four sub-batches overlap their launches)"}
        if not args.no_cpu:
            base, parity = cpu_baseline(d, x_host, (atom, lag, gain))
            line["cpu_baseline"] = base
            line["parity_vs_cpu_oracle"] = parity
            line["speedup_vs_cpu_baseline"] = round(line["value"] / base["value"], 1)
            line["cpu_baseline_torch_ops"] = cpu_baseline_torch(d, x_host, (atom, lag, gain))
    nat.profile_enable(False)
    if world > 1:
        line.update(multi_rank_fields(x, d, dt, args.steps, rank, world, dev, group))
    if rank == 0 and line.get("variants"):
        # (the driver's record keeps the standard keys and the TAIL of stdout: one compact line per variant goes last)
        line["variants_summary"] = {k: [v.get("value"), v.get("unit", "")[:28], v.get("ms_per_step", v.get("encode_ms"))]
                                    for k, v in line["variants"].items() if isinstance(v, dict)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

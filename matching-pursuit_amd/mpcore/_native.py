"""ctypes binding of libmpcore.so (C ABI: include/mpcore.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
PyTorch is used only as the owner of device memory and streams; every pointer handed to the
library is `tensor.data_ptr()`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmpcore.so")

MP_PATH_DIRECT = 0
MP_PATH_FFT = 1
MP_PATH_INCREMENTAL = 2
MP_PATH_NAIVE = 8
MP_FLAG_NO_DMA = 1
MP_FLAG_TA32 = 2
MP_FLAG_TA64 = 4
MP_FLAG_NO_PERSISTENT = 8
MP_FLAG_NO_STAGGER = 16
MP_FLAG_REFINE_MFMA = 32
MP_FLAG_FFT_SIMPLE = 64
MP_TUNE_TAU = 1
MP_TUNE_SCREEN_PPS = 2
MP_TUNE_GROUPS = 3
MP_TUNE_AUDIT = 4
MP_TUNE_PERSIST_SHARDS = 6
MP_TUNE_PERSIST_WORKERS = 7
MP_TUNE_PERSIST_SELECTS = 8
MP_TUNE_LAZY_MARGIN = 10
MP_TUNE_LAZY_REUSE = 12
MP_TUNE_LAZY_RADIUS = 13
MP_TUNE_PERSIST_PRESCAN = 14
MP_TUNE_CLEAR_MEMSET = 15
MP_TUNE_PERSIST_FINE = 18 # persistent form: 1 = one slot per tile quarter, 2 = two slots (finer tasks), 0 = by load
MP_TUNE_LAZY_COMPACT = 17 # launch-per-step lazy screen: masked launches from a compacted work list (1, default) or early exits (0)
MP_TUNE_LAZY_FORCE = 16   # timing experiments only: random tile masks, wrong events
MP_FLAG_GROUPS_SHIFT = 20
MP_FLAG_NO_OVERLAP = 4096
MP_FLAG_FFT_NO_QUARTER = 8192
MP_FLAG_FFT_QUARTER = 16384
MP_FLAG_FFT_UNFUSED = 512
MP_FLAG_FFT_FUSED = 1024
MP_FLAG_OVERLAP = 2048
MP_FLAG_FFT_PERSISTENT = 65536
MP_FLAG_FFT_NO_PERSISTENT = 131072

EXPORTS = (
    "mp_version", "mp_last_error", "mp_workspace_bytes", "mp_unit_norm_f32", "mp_encode_f32",
    "mp_feature_map_f32", "mp_scatter_f32", "mp_scatter_rows_f32", "mp_gather_sum_f32",
    "mp_profile_enable", "mp_profile_read", "mp_fft_c2c_f32", "mp_encode_conv_f32", "mp_tune",
    "mp_dictionary_update_f32", "mp_lcn_workspace_bytes", "mp_encode_lcn_f32", "mp_conv_model_backward_f32",
    "mp_stream_pair_ratio", "mp_init_streams", "mp_audit_read", "mp_dictionary_levels_host",
    "mp_dictionary_update_levels_f32", "mp_persist_stats", "mp_last_schedule", "mp_encode_lazy_f32",
    "mp_coherence_f32", "mp_coherence_workspace_bytes", "mp_lazy_stats", "mp_gather_sum_groups_f32",
    "mp_dictionary_level_addback_sum_f32", "mp_dictionary_level_subtract_f32", "mp_form_table",
)


def flag_groups(n):
    """MP_FLAG_GROUPS(n): this call splits its batch into n (2..4) sub-batches where it splits at all."""
    return (int(n) & 7) << MP_FLAG_GROUPS_SHIFT

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libmpcore.so (once).  Raises NativeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with matching-pursuit_amd/csrc/build.sh "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        i64, vp, fp = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
        L.mp_version.restype = ctypes.c_int
        L.mp_last_error.restype = ctypes.c_char_p
        L.mp_workspace_bytes.restype = ctypes.c_size_t
        L.mp_workspace_bytes.argtypes = [i64, i64, i64, i64, ctypes.c_int, ctypes.c_int]
        L.mp_unit_norm_f32.argtypes = [vp, i64, i64, fp, vp, vp]
        L.mp_conv_model_backward_f32.argtypes = [vp, i64, i64, vp, vp, vp, ctypes.c_int, vp, vp, ctypes.c_int, i64, i64,
                                                 vp, vp, vp, vp]
        L.mp_lcn_workspace_bytes.restype = ctypes.c_size_t
        L.mp_lcn_workspace_bytes.argtypes = [i64, i64, i64, i64, ctypes.c_int]
        L.mp_encode_lcn_f32.argtypes = [vp, i64, i64, vp, i64, i64, ctypes.c_int, vp, vp, vp, vp, vp,
                                        ctypes.c_size_t, vp]
        L.mp_encode_f32.argtypes = [vp, i64, i64, vp, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.mp_encode_conv_f32.argtypes = L.mp_encode_f32.argtypes
        L.mp_coherence_workspace_bytes.restype = ctypes.c_size_t
        L.mp_coherence_workspace_bytes.argtypes = [i64, i64]
        L.mp_coherence_f32.argtypes = [vp, i64, i64, vp, vp, ctypes.c_size_t, vp]
        L.mp_encode_lazy_f32.argtypes = [vp, i64, i64, vp, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp,
                                         vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.mp_feature_map_f32.argtypes = [vp, i64, i64, vp, i64, i64, vp, vp, ctypes.c_size_t, vp]
        L.mp_scatter_f32.argtypes = [vp, vp, vp, vp, i64, vp, i64, i64, vp, i64, i64, vp]
        L.mp_scatter_rows_f32.argtypes = [vp, vp, vp, i64, i64, vp, i64, i64, vp]
        L.mp_gather_sum_f32.argtypes = [vp, i64, i64, vp, vp, i64, i64, vp, vp]
        L.mp_gather_sum_groups_f32.argtypes = [vp, i64, i64, vp, vp, vp, i64, i64, vp, vp]
        L.mp_dictionary_level_addback_sum_f32.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp]
        L.mp_dictionary_level_subtract_f32.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp]
        L.mp_dictionary_update_f32.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, vp, vp, vp, vp,
                                               ctypes.c_float, vp, vp]
        L.mp_fft_c2c_f32.argtypes = [vp, vp, ctypes.c_int, i64, ctypes.c_int, vp, vp]
        L.mp_dictionary_levels_host.argtypes = [vp, i64, vp, vp, i64, i64, vp, vp, ctypes.POINTER(i64)]
        L.mp_dictionary_update_levels_f32.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, vp, i64, vp, vp, vp, vp,
                                                      ctypes.c_float, vp, vp, vp, i64, vp]
        L.mp_form_table.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.mp_init_streams.argtypes = [vp]
        L.mp_audit_read.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i64),
                                    ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i64)]
        L.mp_tune.argtypes = [ctypes.c_int, ctypes.c_double]
        L.mp_stream_pair_ratio.restype = ctypes.c_float
        L.mp_stream_pair_ratio.argtypes = [ctypes.c_int, ctypes.c_int]
        for name in EXPORTS:
            getattr(L, name)
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed (rc={rc}): {lib().mp_last_error().decode()}")


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("libmpcore operates on device memory only: got a CPU tensor")


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def tune(key, value):
    """mp_tune(MP_TUNE_*, value): process-wide tuning / debug knobs (include/mpcore.h)."""
    _check(lib().mp_tune(int(key), float(value)), "mp_tune")


def init_streams(device=None):
    """mp_init_streams: build and test this thread's internal stream pool now (host-synchronising, once) rather
    than inside the first sub-batched encode; -> number of streams seen to run side by side."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        n = lib().mp_init_streams(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if n < 0:
        raise NativeError(f"mp_init_streams failed (rc={n}): {lib().mp_last_error().decode()}")
    return int(n)


def last_schedule():
    """mp_last_schedule: -1 persistent form, 1 one stream, n >= 2 sub-batches (this thread's last encode)."""
    return int(lib().mp_last_schedule())


def persist_stats():
    """mp_persist_stats: dict of the last persistent launch's statistics (ticks of 10 ns summed over workgroups)."""
    buf = (ctypes.c_uint64 * 16)()
    _check(lib().mp_persist_stats(buf), "mp_persist_stats")
    k = ("idle_ticks", "task_ticks", "select_ticks", "tasks", "selects", "polls", "error", "finished")
    out = {n: int(buf[i]) for i, n in enumerate(k)}
    n = max(int(buf[13]), 1)
    out["skipped"] = int(buf[14])
    out["select_phase_us"] = {p: round(int(buf[8 + i]) / 100.0 / n, 2)
                              for i, p in enumerate(("acquire", "scan", "chains", "event_window", "transform_stores"))}
    out["prescans"] = int(buf[15])
    return out


def lazy_stats():
    """mp_lazy_stats: dict(skipped, decided, ...) -- (segment, tile) screens the launch-per-step lazy screen left out /
    decided on since the last read, and the selects that decided nothing, by reason (the persistent form counts its own
    in persist_stats()['skipped']).  Synchronises; resets."""
    buf = (ctypes.c_uint64 * 8)()
    _check(lib().mp_lazy_stats(buf), "mp_lazy_stats")
    return dict(skipped=int(buf[0]), decided=int(buf[1]), off_static=int(buf[2]), off_contenders=int(buf[3]),
                off_no_floor=int(buf[4]), off_no_bound=int(buf[5]), contender_cells=int(buf[6]),
                contender_quarters=int(buf[7]))


def audit_read():
    """mp_audit_read (after tune(MP_TUNE_AUDIT, 1)): dict(max_ratio, cells, max_quarter_ratio, over_bound); resets."""
    r, q = ctypes.c_float(0), ctypes.c_float(0)
    n, o = ctypes.c_int64(0), ctypes.c_int64(0)
    _check(lib().mp_audit_read(ctypes.byref(r), ctypes.byref(n), ctypes.byref(q), ctypes.byref(o)), "mp_audit_read")
    return dict(max_ratio=float(r.value), cells=int(n.value), max_quarter_ratio=float(q.value), over_bound=int(o.value))


def profile_enable(every=1, correlate_only=False):
    """0 / False: off; 1 / True: events around the launches of every iteration; n: every n-th iteration.
    correlate_only: no spans around the selects (every event between two kernels idles the GPU for ~8 us)."""
    _check(lib().mp_profile_enable(int(every) | ((4 << 16) if correlate_only and every else 0)), "mp_profile_enable")


def profile_read():
    """-> dict(kind -> (total_ms, launches)) for kinds corr_full, corr_inc, select; resets."""
    ms = (ctypes.c_double * 3)()
    cnt = (ctypes.c_int64 * 3)()
    _check(lib().mp_profile_read(ms, cnt), "mp_profile_read")
    return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(("corr_full", "corr_inc", "select"))}


def unit_norm(d, eps=1e-8):
    d = _f32(d)
    _require_cuda(d)
    A, L = d.shape
    out = torch.empty_like(d)
    with torch.cuda.device(d.device):
        _check(lib().mp_unit_norm_f32(_ptr(d), A, L, eps, _ptr(out), _stream(d)), "mp_unit_norm_f32")
    return out


def workspace_bytes(B, N, A, L, K, path):
    n = lib().mp_workspace_bytes(B, N, A, L, K, path)
    if n == 0 and B > 0:
        raise NativeError(f"mp_workspace_bytes: {lib().mp_last_error().decode()}")
    return int(n)


FFT_MAX_ATOM = 21782    # longest atom whose 3L+190-point transform fits LDS whole (<= 5398), as two halves (<= 10859) or four quarters
FFT_MAX_BATCH = 65535   # segments per mp_encode_f32 call on MP_PATH_FFT (the wrapper chunks larger batches)


def default_path(n_atom_samples):
    """The fastest exact schedule: the FFT screen + exact refinement where a transform fits LDS,
    the incremental direct (MFMA) schedule otherwise.  All paths select identical events."""
    return MP_PATH_FFT if n_atom_samples <= FFT_MAX_ATOM else MP_PATH_INCREMENTAL


def encode_checked(signal, dict_unit, n_steps, flags=0, want_residual=True):
    """encode() on the default path, plus the one thing the asynchronous call cannot do itself: segments whose FFT
    screen overflowed (marked in-band with gain = NaN, see include/mpcore.h) are encoded again -- first on the same
    schedule without the lazy screen (its stale bounds add contenders on signals whose maxima collapse within the run:
    such marks go away at the FFT schedule's speed), then, if still marked (near-ties: duplicated atoms, periodic
    signals), on the incremental path.  Costs one host synchronisation."""
    path = default_path(dict_unit.shape[1])
    atom, lag, gain, residual = encode(signal, dict_unit, n_steps, path=path, flags=flags,
                                       want_residual=want_residual)
    was_lazy = bool(getattr(_tls, "lazy", False))   # (this thread's call, not the process-wide statistics of the last launch)
    if path == MP_PATH_FFT and gain.numel():
        bad = torch.isnan(gain).any(dim=1)
        if bool(bad.any()):
            idx = torch.nonzero(bad).flatten()
            for retry_path, kw in ((MP_PATH_FFT, {"coherence": False}), (MP_PATH_INCREMENTAL, {})):
                if retry_path == MP_PATH_FFT and not was_lazy:
                    continue                               # (the lazy screen was not in play: the same run again)
                a2, l2, g2, r2 = encode(signal[idx], dict_unit, n_steps, path=retry_path, flags=flags,
                                        want_residual=want_residual, **kw)
                atom[idx], lag[idx], gain[idx] = a2, l2, g2
                if want_residual:
                    residual[idx] = r2
                still = torch.isnan(g2).any(dim=1)
                if not bool(still.any()):
                    break
                idx = idx[still]
    return atom, lag, gain, residual


def coherence_table(dict_unit, chunk=128, exact=False, slack=True):
    """The dictionary's coherence table for the lazy screen (mp_encode_lazy_f32): [A, NAT] f32 on the device,
    entry (a, t) >= max over the 32 atoms b of tile t and all shifts s of |sum_j d_a[j] d_b[j + s]|.
    Default: mp_coherence_f32 -- one full-pass FFT screen of the atoms against the dictionary, |.| maxima plus the
    screen's bound (~0.3 ms at the headline dictionary); where that does not exist (transform sizes the lazy screen
    does not cover) or with exact=True: exact correlations of every atom, placed in a zero row, with the whole dictionary
    (mp_feature_map_f32), plus the fp32 chain's worst-case rounding u L (~5 ms; slack=False leaves that term out: the
    computed correlations themselves, for tests)."""
    dict_unit = _f32(dict_unit)
    _require_cuda(dict_unit)
    A, L = dict_unit.shape
    nbytes = 0 if exact else lib().mp_coherence_workspace_bytes(A, L)
    if nbytes:
        dev = dict_unit.device
        out = torch.empty((A, (A + 31) // 32), dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        with torch.cuda.device(dev):
            rc = lib().mp_coherence_f32(_ptr(dict_unit), A, L, _ptr(out), ctypes.c_void_p(ws.data_ptr() + off), nbytes,
                                        _stream(dict_unit))
        _check(rc, "mp_coherence_f32")
        ws.record_stream(torch.cuda.current_stream(dev))
        return out
    nat_tiles = (A + 31) // 32
    rows = torch.zeros((min(chunk, A), 3 * L - 2), dtype=torch.float32, device=dict_unit.device)
    out = torch.empty((A, nat_tiles), dtype=torch.float32, device=dict_unit.device)
    pad = nat_tiles * 32 - A
    for a0 in range(0, A, chunk):
        n = min(chunk, A - a0)
        rows.zero_()
        rows[:n, L - 1:2 * L - 1] = dict_unit[a0:a0 + n]
        fm = feature_map(rows[:n], dict_unit)[..., :2 * L - 1]          # [n, A, 2L-1]: every shift of every pair
        m = fm.abs().amax(dim=-1)                                       # [n, A]
        if pad:
            m = torch.nn.functional.pad(m, (0, pad))
        out[a0:a0 + n] = m.view(n, nat_tiles, 32).amax(dim=-1)
    if not slack:
        return out
    return out + float(L) * 5.9604645e-8 * float(dict_unit.norm(dim=-1).max()) ** 2


class _RememberedDictionary:
    """One remembered dictionary per (shape, device, stream): a private COPY of the normalised dictionary, the coherence
    table computed from that copy, and the pinned flag a call's device-side comparison lands in (read, without waiting
    for it, by a later call)."""
    __slots__ = ("copy", "table", "flag", "event", "pending", "volatile")

    def __init__(self, dict_unit):
        self.copy = dict_unit.clone()
        self.table = None
        self.flag = torch.zeros(1, dtype=torch.bool).pin_memory()
        self.event = torch.cuda.Event()
        self.pending = False
        self.volatile = False      # the dictionary was seen to change between calls: nothing is kept until it stops


_coherence_cache = {}   # (shape, device, stream) -> _RememberedDictionary
_coherence_lock = __import__("threading").Lock()
COHERENCE_CACHE_ENTRIES = 8


def clear_caches():
    """Forget every remembered dictionary and coherence table (device memory: a copy of the dictionary and an
    [A, A / 32] table per entry, at most COHERENCE_CACHE_ENTRIES entries)."""
    with _coherence_lock:
        _coherence_cache.clear()


def validated_table(dict_unit, copy, table):
    """`table` where `dict_unit` equals `copy` (the dictionary the table was computed from) in every element, +inf
    everywhere otherwise -- decided ON THE DEVICE, in stream order, no host synchronisation: with an infinite coherence
    the lazy screen's widened bound never stays below anything, so no tile is ever skipped and the encode is the plain
    one (mppersist.inc: `val + ev + |g| mu < LB_clean` is false; for |g| = 0 the product is NaN and compares false
    too).  -> (table to hand to mp_encode_lazy_f32, the 0-d bool tensor of the comparison)."""
    same = (dict_unit == copy).all()
    return torch.where(same, table, torch.full_like(table, float("inf"))), same


def cached_coherence(dict_unit, build_now=False):
    """The coherence table for the lazy screen if THIS dictionary -- by content -- has been encoded against before (or,
    with build_now, in any case), else None.

    Nothing here trusts object identity, storage pointers or torch's version counter (writes through `.data`, which is how
    the reference's experiments update their dictionaries -- experiments/archive/e_2023_7_20/experiment.py:269-283 --
    do not bump it).  Per (shape, device, stream) one dictionary is remembered as a private copy, with the table
    computed from that copy.  A remembered table is only ever handed out through validated_table(): the call's dictionary
    is compared with the copy on the device and the kernel gets the table where they are equal and +inf (no tile
    skipped: the plain encode) where they are not -- a stale table cannot reach a decision.  The comparison's result also
    lands in a pinned flag which a LATER call reads once its event has completed (no waiting):
      * equal: a dictionary seen unchanged gets its table built from the copy (mp_coherence_f32, ~0.45 ms at 512 x 512)
        -- the lazy screen from the third encode against a fixed dictionary;
      * different: the entry turns `volatile` -- the copy follows the dictionary call by call and no table is kept --
        until a comparison comes back equal again.  A dictionary that changes every call never pays for a table.
    build_now (batches large enough for one call to pay for the table): where no trusted table exists the table is
    computed from `dict_unit` itself -- valid by construction -- and remembered with it.
    Returns None inside a stream capture (an EncodePlan owns its table) and for shapes the lazy screen does not cover."""
    A, L = dict_unit.shape
    if torch.cuda.is_current_stream_capturing() or lib().mp_coherence_workspace_bytes(A, L) == 0:
        return None
    dev = dict_unit.device
    stream = torch.cuda.current_stream(dev)
    key = (A, L, str(dev), stream.cuda_stream)
    with _coherence_lock:
        ent = _coherence_cache.get(key)
        if ent is None:
            while len(_coherence_cache) >= COHERENCE_CACHE_ENTRIES:
                _coherence_cache.pop(next(iter(_coherence_cache)))
            ent = _coherence_cache[key] = _RememberedDictionary(dict_unit)
            if build_now:
                ent.table = coherence_table(ent.copy)
            return ent.table                                   # (computed from this very dictionary: nothing to validate)
        news = None
        if ent.pending and ent.event.query():
            ent.pending = False
            news = bool(ent.flag[0])
            if news:
                ent.volatile = False
            else:
                ent.volatile, ent.table = True, None
        trusted = ent.table is not None and not ent.volatile
        if trusted:
            table, same = validated_table(dict_unit, ent.copy, ent.table)
        else:
            table, same = None, (dict_unit == ent.copy).all()
        if not ent.pending:
            ent.flag.copy_(same.view(1), non_blocking=True)
            ent.event.record(stream)
            ent.pending = True
        if trusted:
            return table
        if build_now:
            ent.copy.copy_(dict_unit)                          # (after the comparison, in stream order)
            table = coherence_table(ent.copy)
            if not ent.volatile:
                ent.table = table
            return table
        if news and not ent.volatile:                          # seen unchanged: worth a table
            ent.table = coherence_table(ent.copy)
            return validated_table(dict_unit, ent.copy, ent.table)[0]
        if ent.volatile:
            ent.copy.copy_(dict_unit)
        return None


FORM_FIELDS = ("quarter_max_cells", "fused_min_cells", "persist_max_cells", "persist_points_table", "persist_points_table_1024",
               "persist_points_any_batch_1024", "persist_segments_table_1024", "sub_batch_min_segments", "persist_spectra_bytes", "persist_points_fit", "persist_points_nofit", "short_ratio",
               "short_logm_always", "short_logm_small", "short_small_segments", "sub_batches", "persist_two_per_cu_load",
               "persist_fine_tasks_per_cu", "persist_select_workers", "lazy_min_steps", "lazy_min_tiles", "lazy_always_tiles",
               "lazy_batch_tiles", "lazy_margin_persistent", "lazy_margin_persistent_1024", "lazy_margin_steps",
               "persist_fine_max_logm")
_form = None


def form_table():
    """mp_form_table: the library's form-selection thresholds (csrc/mpcore.hip::FormTable) as a dict -- the one place the
    numbers live; lazy_pays below takes its own from here."""
    global _form
    if _form is None:
        buf = (ctypes.c_double * len(FORM_FIELDS))()
        n = lib().mp_form_table(buf, len(FORM_FIELDS))
        if n != len(FORM_FIELDS):
            raise NativeError(f"mp_form_table has {n} entries, this mirror knows {len(FORM_FIELDS)}")
        _form = dict(zip(FORM_FIELDS, [float(v) for v in buf]))
    return _form


def lazy_pays(batch, n_atoms, n_steps, n_samples=None, atom_samples=None):
    """Is the lazy screen worth asking for (scripts/small_lazy.py, persistent form with / without the table)?  What it saves
    is screen tasks, what it costs is ~1.2 us in every select: it pays where a step has many tile screens to shed -- 512 x 512
    (16 tiles): -1 .. -4 % up to 16 segments, +4 % at 24, +30 % at 64; 1024 x 1024 (32 tiles of 4096-point tasks): +7 % at ONE
    segment, +16 .. +32 % from four; dictionaries of one or two tiles (64 x 300, 16 x 256): -5 % at any batch."""
    tiles = (int(n_atoms) + 31) // 32
    if n_samples is not None and atom_samples is not None:
        # (the table is validated against the dictionary call after call -- a pass over both copies and a handful of host
        #  operations, 0.13 ms at 1024 x 2048 -- so it is only asked for where the library would look at it:)
        L, N = int(atom_samples), int(n_samples)
        log_m = max(8, (3 * L + 190 - 1).bit_length())        # csrc/mpfft.inc: make_fft_geom -- M = 2^log_m >= 3 L + 190
        if log_m >= 13 and ((N + 63) // 64) * tiles < form_table()["fused_min_cells"]:
            # transforms of 8192 points and more: no persistent form, and the launch-per-step lazy screen lives in the fused
            # select, which only segments of 65536 cells and more take
            return False
        if form_table()["short_ratio"] * L >= N:
            # short segments (an event dirties half of the lags or more): at 4096-point transforms, and at 2048 points up to
            # 8 segments, they run launch per step on the quarter select (csrc/mpcore.hip: encode_impl, short_segments);
            # in the persistent form the table loses too (scripts/small_batch_forms.py, 1024 x 512 atoms, 2048-sample
            # segments, planted events, with / without: 8 segments 206 / 214 k, 32: 552 / 597 k)
            return False
    f = form_table()
    return int(n_steps) >= f["lazy_min_steps"] and tiles >= f["lazy_min_tiles"] and (
        tiles >= f["lazy_always_tiles"] or int(batch) * tiles >= f["lazy_batch_tiles"])


_tls = __import__("threading").local()   # .lazy: did this thread's last encode() hand the kernel a coherence table?


def encode(signal, dict_unit, n_steps, path=MP_PATH_INCREMENTAL, flags=0, want_residual=True, conv_model=False,
           coherence=None):
    """signal [B,N] f32 cuda, dict_unit [A,L] f32 cuda -> (atom[B,K] i64, lag[B,K] i64,
    gain[B,K] f32, residual[B,N] f32 | None), all on signal.device, asynchronous.
    conv_model=True: mp_encode_conv_f32 (the analysis loop of mp.py's model; `dict_unit` = raw atoms).
    coherence: the dictionary's coherence_table() -> mp_encode_lazy_f32 (MP_PATH_FFT's persistent form skips the
    transforms of tiles an event cannot have lifted into contention; same events).  A table passed explicitly must belong
    to `dict_unit` as it is NOW (a stale one skips tiles that hold the maximum -- wrong events, no mark).  None:
    cached_coherence() decides (a dictionary seen before, compared by content on the device, gets its table); False: never."""
    signal = _f32(signal)
    dict_unit = _f32(dict_unit)
    _require_cuda(signal, dict_unit)
    B, N = signal.shape
    A, L = dict_unit.shape
    K = int(n_steps)
    dev = signal.device
    atom = torch.empty((B, K), dtype=torch.int64, device=dev)
    lag = torch.empty((B, K), dtype=torch.int64, device=dev)
    gain = torch.empty((B, K), dtype=torch.float32, device=dev)
    residual = torch.empty((B, N), dtype=torch.float32, device=dev) if want_residual else None
    _tls.lazy = False
    if B == 0:
        return atom, lag, gain, residual
    if path == MP_PATH_FFT and B > FFT_MAX_BATCH:  # one grid dimension of the screen is the batch: chunk it
        any_lazy = False
        for b0 in range(0, B, FFT_MAX_BATCH):
            sl = slice(b0, min(b0 + FFT_MAX_BATCH, B))
            a, l, g, r = encode(signal[sl], dict_unit, K, path=path, flags=flags, want_residual=want_residual,
                                conv_model=conv_model, coherence=coherence)
            atom[sl], lag[sl], gain[sl] = a, l, g
            any_lazy = any_lazy or _tls.lazy
            if want_residual:
                residual[sl] = r
        _tls.lazy = any_lazy
        return atom, lag, gain, residual
    if coherence is None and path == MP_PATH_FFT and not conv_model and lazy_pays(B, A, K, N, L) and \
            not (int(flags) & ~MP_FLAG_FFT_PERSISTENT):
        # a batch large enough for the table to pay for itself within this one call gets it at once, new dictionary or not:
        # the table is A * A / 2 transforms, the launch it trims B (K - 1) A / 2, of which a third go (512 atoms, 64 steps:
        # 64 segments 3.75 -> 2.93 + 0.45 ms, 128 segments 7.3 -> 5.4 + 0.45 ms; 48 segments 3.21 -> 2.82 + 0.45: not yet)
        coherence = cached_coherence(dict_unit, build_now=B >= 64 and B * (K - 1) >= 6 * A)
    nbytes = workspace_bytes(B, N, A, L, K, path)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        if coherence is not None and coherence is not False and not conv_model and path == MP_PATH_FFT:
            coherence = _f32(coherence)
            assert coherence.shape == (A, (A + 31) // 32) and coherence.device == dev
            rc = lib().mp_encode_lazy_f32(_ptr(signal), B, N, _ptr(dict_unit), A, L, K, int(path), int(flags), _ptr(coherence),
                                          _ptr(atom), _ptr(lag), _ptr(gain), _ptr(residual),
                                          ctypes.c_void_p(ws.data_ptr() + off), nbytes, _stream(signal))
            coherence.record_stream(torch.cuda.current_stream(dev))
            _tls.lazy = True
        else:
            fn = lib().mp_encode_conv_f32 if conv_model else lib().mp_encode_f32
            rc = fn(_ptr(signal), B, N, _ptr(dict_unit), A, L, K, int(path), int(flags),
                    _ptr(atom), _ptr(lag), _ptr(gain), _ptr(residual),
                    ctypes.c_void_p(ws.data_ptr() + off), nbytes, _stream(signal))
    _check(rc, "mp_encode_conv_f32" if conv_model else "mp_encode_f32")
    # the workspace must outlive the asynchronous kernels: tie it to the stream
    ws.record_stream(torch.cuda.current_stream(dev))
    return atom, lag, gain, residual


def conv_model_backward(atoms, atom_idx, time_idx, value, residual_final, grad_channels, windowed=False):
    """mp_conv_model_backward_f32: -> (grad_audio [B, N], grad_rows [B, K, L]).  grad_channels is [B, K, N], or
    [B, K, L] (each event's own support only) with windowed=True."""
    atoms, value, residual_final = _f32(atoms), _f32(value), _f32(residual_final)
    grad_channels = _f32(grad_channels)
    atom_idx, time_idx = atom_idx.contiguous(), time_idx.contiguous()
    _require_cuda(atoms, atom_idx, time_idx, value, residual_final, grad_channels)
    A, L = atoms.shape
    B, K = atom_idx.shape
    N = residual_final.shape[1]
    assert grad_channels.shape == (B, K, L if windowed else N)
    assert atom_idx.dtype == torch.int64 and time_idx.dtype == torch.int64
    dev = atoms.device
    grad_audio = torch.empty((B, N), dtype=torch.float32, device=dev)
    grad_rows = torch.empty((B, K, L), dtype=torch.float32, device=dev)
    scratch = torch.empty((B, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mp_conv_model_backward_f32(_ptr(atoms), A, L, _ptr(atom_idx), _ptr(time_idx), _ptr(value), K,
                                              _ptr(residual_final), _ptr(grad_channels), int(bool(windowed)), B, N,
                                              _ptr(grad_audio),
                                              _ptr(grad_rows), _ptr(scratch), _stream(atoms))
    _check(rc, "mp_conv_model_backward_f32")
    scratch.record_stream(torch.cuda.current_stream(dev))
    return grad_audio, grad_rows


class EncodePlan:
    """One whole encode (all K steps, every launch, the fork / join of the internal streams) captured ONCE as a
    hipGraph and replayed per batch -- for callers that encode many batches of one shape (a training loop, a
    streaming encoder).  mp_encode_f32 is capture-safe (no host synchronisation, caller-supplied workspace), so
    the capture is the ordinary call under torch.cuda.graph.  Measured (scripts/graph_latency.py,
    scripts/graph_groups.py): 1.15x at BASELINE configs[0] (one segment, 8 steps: 226 -> 198 us); at the headline
    shape a replay (848 k segment-iterations/s) beats the one-stream schedule (819 k) but not the plain launches of
    the four-sub-batch default (877 k): a graph's parallel branches run on the runtime's own streams, which share
    hardware queues beyond two -- so a plan is captured with two sub-batches unless the shape takes the persistent
    form (one launch for steps 1 .. K-1: no branches), which is captured as it is.

        plan = EncodePlan(B, N, dict_unit, n_steps)        # captures; the dictionary is read at replay time
        atom, lag, gain, residual = plan(signal)             # [B, N] -> views of the plan's static outputs

    The outputs are overwritten by the next call (clone what must outlive it).  As with encode(), a segment
    whose FFT screen overflowed is marked with gain = NaN: encode_checked() is the checked, un-captured form.

    The dictionary and the lazy screen.  `dict_unit` (when it is fp32 and contiguous, the caller's own tensor) is read
    by every replay, so a training loop may update it in place between replays.  The lazy screen's coherence table is a
    function of the dictionary, so the plan OWNS one (`lazy=True`, the default where the shape takes the persistent form)
    together with a private copy of the dictionary it was computed from, and the captured graph begins with the
    device-side comparison of _native.validated_table(): a replay whose dictionary differs from that copy runs with an
    infinite table -- every tile screened, the plain encode, same events as any other schedule -- until
    `plan.refresh_dictionary()` recomputes copy and table (mp_coherence_f32, ~0.45 ms at 512 x 512) into the same
    buffers.  Nothing the graph reads belongs to a cache that could evict it."""

    def __init__(self, batch, n_samples, dict_unit, n_steps, path=None, flags=0, want_residual=True, sub_batches=None,
                 lazy=None):
        dict_unit = _f32(dict_unit)
        _require_cuda(dict_unit)
        self.path = default_path(dict_unit.shape[1]) if path is None else path
        self.dict_unit = dict_unit
        dev = dict_unit.device
        A, L = dict_unit.shape
        self.signal = torch.zeros((int(batch), int(n_samples)), dtype=torch.float32, device=dev)
        # sub_batches=None: the library's choice for the shape (the persistent form where it applies, else two
        # sub-batches); a number: that many.  A per-call flag: nothing process-wide is touched while other threads encode
        if sub_batches is None:
            encode(self.signal, dict_unit, n_steps, path=self.path, flags=int(flags), want_residual=False, coherence=False)
            groups = 0 if last_schedule() == -1 else 2
        else:
            groups = max(2, min(4, int(sub_batches)))
        if lazy is None:
            lazy = (groups == 0 and self.path == MP_PATH_FFT and lazy_pays(batch, A, n_steps, n_samples, L) and
                    not (int(flags) & ~MP_FLAG_FFT_PERSISTENT))
        self.lazy = bool(lazy) and self.path == MP_PATH_FFT and lib().mp_coherence_workspace_bytes(A, L) > 0
        self._dict_copy = self._table = None
        if self.lazy:
            self._dict_copy = torch.empty_like(dict_unit)
            self._table = torch.empty((A, (A + 31) // 32), dtype=torch.float32, device=dev)
            self.refresh_dictionary()
        self._args = dict(path=self.path, flags=int(flags) | (flag_groups(groups) if groups else 0),
                          want_residual=want_residual)
        with torch.cuda.device(dev):
            init_streams(dev)  # the pool's self-test synchronises with the host: before the capture, not inside it
        self._capture(dev, n_steps)

    def refresh_dictionary(self):
        """After an in-place update of the dictionary: recompute the plan's copy and coherence table (into the buffers
        the captured graph reads), so that later replays skip transforms again.  Replays in between are exact without
        it (every tile screened).  No-op for a plan without the lazy screen."""
        if self.lazy:
            self._dict_copy.copy_(self.dict_unit)
            self._table.copy_(coherence_table(self._dict_copy))

    def _encode(self, n_steps):
        if not self.lazy:
            return encode(self.signal, self.dict_unit, n_steps, coherence=False, **self._args)
        table, _ = validated_table(self.dict_unit, self._dict_copy, self._table)
        return encode(self.signal, self.dict_unit, n_steps, coherence=table, **self._args)

    def _capture(self, dev, n_steps):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up outside the capture: kernel attributes, the stream pool
            self._encode(n_steps)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked while capturing -- another host thread may be encoding,
        # synchronising or allocating meanwhile (the default "global" mode invalidates the capture when it does)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = self._encode(n_steps)

    def __call__(self, signal):
        if signal.shape != self.signal.shape:
            raise NativeError(f"EncodePlan was captured for {tuple(self.signal.shape)}, got {tuple(signal.shape)}")
        self.signal.copy_(signal)
        self.graph.replay()
        return self.outputs


LCN_MAP_BYTES = 16 << 30  # the dense [B, A, N] map one mp_encode_lcn_f32 call may hold; larger batches are chunked


def encode_lcn(signal, dict_unit, n_steps, want_residual=True):
    """sparse_code(local_contrast_norm=True) (modules/matchingpursuit.py:284-294) -> as encode()."""
    signal = _f32(signal)
    dict_unit = _f32(dict_unit)
    _require_cuda(signal, dict_unit)
    B, N = signal.shape
    A, L = dict_unit.shape
    K = int(n_steps)
    dev = signal.device
    atom = torch.empty((B, K), dtype=torch.int64, device=dev)
    lag = torch.empty((B, K), dtype=torch.int64, device=dev)
    gain = torch.empty((B, K), dtype=torch.float32, device=dev)
    residual = torch.empty((B, N), dtype=torch.float32, device=dev) if want_residual else None
    if B == 0:
        return atom, lag, gain, residual
    chunk = max(1, min(FFT_MAX_BATCH, LCN_MAP_BYTES // (4 * A * N)))
    if B > chunk:
        for b0 in range(0, B, chunk):
            sl = slice(b0, min(b0 + chunk, B))
            a, l, g, r = encode_lcn(signal[sl], dict_unit, K, want_residual=want_residual)
            atom[sl], lag[sl], gain[sl] = a, l, g
            if want_residual:
                residual[sl] = r
        return atom, lag, gain, residual
    nbytes = int(lib().mp_lcn_workspace_bytes(B, N, A, L, K))
    if nbytes == 0:
        raise NativeError(f"mp_lcn_workspace_bytes failed: {lib().mp_last_error().decode()}")
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        rc = lib().mp_encode_lcn_f32(_ptr(signal), B, N, _ptr(dict_unit), A, L, K, _ptr(atom), _ptr(lag),
                                     _ptr(gain), _ptr(residual), ctypes.c_void_p(ws.data_ptr() + off), nbytes,
                                     _stream(signal))
    _check(rc, "mp_encode_lcn_f32")
    ws.record_stream(torch.cuda.current_stream(dev))
    return atom, lag, gain, residual


def feature_map(residual, dict_unit):
    residual = _f32(residual)
    dict_unit = _f32(dict_unit)
    _require_cuda(residual, dict_unit)
    B, N = residual.shape
    A, L = dict_unit.shape
    dev = residual.device
    fm = torch.empty((B, A, N), dtype=torch.float32, device=dev)
    if B == 0:
        return fm
    nbytes = workspace_bytes(B, N, A, L, 0, MP_PATH_DIRECT)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(dev):
        rc = lib().mp_feature_map_f32(_ptr(residual), B, N, _ptr(dict_unit), A, L, _ptr(fm),
                                      ctypes.c_void_p(ws.data_ptr() + off), nbytes, _stream(residual))
    _check(rc, "mp_feature_map_f32")
    ws.record_stream(torch.cuda.current_stream(dev))
    return fm


def fft_c2c(x, inverse=False):
    """Batched complex FFT test hook: x complex64 [batch, 2^k] on the device -> same shape, unscaled.
    inverse: False / True, or 2 for the inverse through the screen kernel's register transform."""
    _require_cuda(x)
    assert x.dtype == torch.complex64 and x.dim() == 2 and x.is_contiguous()
    batch, M = x.shape
    lg = M.bit_length() - 1
    assert 1 << lg == M
    out = torch.empty_like(x)
    ws = torch.empty(8 * M, dtype=torch.uint8, device=x.device)
    xr, outr = torch.view_as_real(x), torch.view_as_real(out)
    with torch.cuda.device(x.device):
        rc = lib().mp_fft_c2c_f32(_ptr(xr), _ptr(outr), lg, batch, int(inverse), _ptr(ws), _stream(x))
    _check(rc, "mp_fft_c2c_f32")
    ws.record_stream(torch.cuda.current_stream(x.device))
    return out


def _i64(t, dev):
    return t.detach().to(device=dev, dtype=torch.int64).contiguous().view(-1)


def scatter(atom, batch, lag, gain, dict_unit, out):
    """out[B,N] += sum of gain * dict_unit[atom] at lag (in place, event order per segment)."""
    _require_cuda(out, dict_unit)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 2
    dev = out.device
    dict_unit = _f32(dict_unit)
    atom, batch, lag = _i64(atom, dev), _i64(batch, dev), _i64(lag, dev)
    gain = gain.detach().to(device=dev, dtype=torch.float32).contiguous().view(-1)
    B, N = out.shape
    A, L = dict_unit.shape
    with torch.cuda.device(dev):
        rc = lib().mp_scatter_f32(_ptr(atom), _ptr(batch), _ptr(lag), _ptr(gain), atom.numel(),
                                  _ptr(dict_unit), A, L, _ptr(out), B, N, _stream(out))
    _check(rc, "mp_scatter_f32")
    return out


def scatter_rows(rows, batch, lag, out):
    _require_cuda(out, rows)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 2
    dev = out.device
    rows = _f32(rows)
    n, L = rows.shape
    batch, lag = _i64(batch, dev), _i64(lag, dev)
    B, N = out.shape
    with torch.cuda.device(dev):
        rc = lib().mp_scatter_rows_f32(_ptr(rows), _ptr(batch), _ptr(lag), n, L, _ptr(out), B, N,
                                       _stream(out))
    _check(rc, "mp_scatter_rows_f32")
    return out


def gather_sum(x, batch, lag, L):
    """sum over events of x[batch, lag:lag+L] (zero beyond N) -> float64 [L]."""
    x = _f32(x)
    _require_cuda(x)
    dev = x.device
    B, N = x.shape
    batch, lag = _i64(batch, dev), _i64(lag, dev)
    out = torch.empty((L,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mp_gather_sum_f32(_ptr(x), B, N, _ptr(batch), _ptr(lag), batch.numel(), L,
                                     _ptr(out), _stream(x))
    _check(rc, "mp_gather_sum_f32")
    return out


def gather_sum_groups(x, batch, lag, offsets, L):
    """Per group g of events [offsets[g] - offsets[0], offsets[g + 1] - offsets[0]): sum over its events of
    x[batch, lag:lag+L] (zero beyond N) -> float64 [G, L].  `offsets`: int64 device tensor [G + 1] (a slice is fine)."""
    x = _f32(x)
    _require_cuda(x)
    dev = x.device
    B, N = x.shape
    batch, lag = _i64(batch, dev), _i64(lag, dev)
    offsets = _i64(offsets, dev)
    G = offsets.numel() - 1
    out = torch.empty((max(G, 0), L), dtype=torch.float64, device=dev)
    if G > 0:
        with torch.cuda.device(dev):
            rc = lib().mp_gather_sum_groups_f32(_ptr(x), B, N, _ptr(batch), _ptr(lag), _ptr(offsets), G, L, _ptr(out), _stream(x))
        _check(rc, "mp_gather_sum_groups_f32")
    return out


def level_addback_sum(residual, sparse, ev_batch, ev_lag, ev_rows, offsets, overlap, L):
    """mp_dictionary_level_addback_sum_f32: phase A of one dependency level of the multi-rank dictionary update.
    `offsets` int64 device [G + 1] (a slice of a longer table), `overlap` int32 device [G] or None -> acc float64 [G, L];
    `residual` is updated in place, `sparse` ([B, N] zeros) is scratch and zero again on return."""
    dev = residual.device
    B, N = residual.shape
    G = offsets.numel() - 1
    acc = torch.empty((max(G, 0), L), dtype=torch.float64, device=dev)
    if G > 0:
        with torch.cuda.device(dev):
            rc = lib().mp_dictionary_level_addback_sum_f32(_ptr(residual), _ptr(sparse), B, N, L, _ptr(offsets), G, _ptr(ev_batch),
                                                           _ptr(ev_lag), _ptr(ev_rows), _ptr(overlap), _ptr(acc), _stream(residual))
        _check(rc, "mp_dictionary_level_addback_sum_f32")
    return acc


def level_subtract(residual, sparse, ev_batch, ev_lag, ev_norm, offsets, overlap, new_atoms, L):
    """mp_dictionary_level_subtract_f32: phase B -- residual -= new_atoms[g] * ||row_e|| for every event of every group."""
    dev = residual.device
    B, N = residual.shape
    G = offsets.numel() - 1
    if G > 0:
        with torch.cuda.device(dev):
            rc = lib().mp_dictionary_level_subtract_f32(_ptr(residual), _ptr(sparse), B, N, L, _ptr(offsets), G, _ptr(ev_batch),
                                                        _ptr(ev_lag), _ptr(ev_norm), _ptr(overlap), _ptr(new_atoms), _stream(residual))
        _check(rc, "mp_dictionary_level_subtract_f32")


def dictionary_levels(offsets, ev_batch, ev_lag, L):
    """mp_dictionary_levels_host on HOST arrays (numpy / CPU tensors, int64): -> (level [G] int32, overlap [G] int32,
    n_levels).  level[g] = 0 if group g overlaps no earlier group, else 1 + the highest level among those it overlaps."""
    import numpy as np
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    ev_batch = np.ascontiguousarray(ev_batch, dtype=np.int64)
    ev_lag = np.ascontiguousarray(ev_lag, dtype=np.int64)
    G = int(offsets.shape[0]) - 1
    level = np.zeros(max(G, 1), dtype=np.int32)
    overlap = np.zeros(max(G, 1), dtype=np.int32)
    n_levels = ctypes.c_int64(0)
    _check(lib().mp_dictionary_levels_host(offsets.ctypes.data, G, ev_batch.ctypes.data, ev_lag.ctypes.data,
                                           int(ev_batch.shape[0]), int(L), level.ctypes.data, overlap.ctypes.data,
                                           ctypes.byref(n_levels)), "mp_dictionary_levels_host")
    return level[:G], overlap[:G], int(n_levels.value)


def dictionary_update(residual, d_work, order, offsets, ev_batch, ev_lag, ev_rows, ev_norm, eps=1e-8,
                      one_by_one=False, host_events=None):
    """The atom-by-atom loop of dictionary_learning_step; residual [B, N] and d_work [A, L] are updated in place.
    host_events = (offsets, ev_batch, ev_lag) as host arrays: the loop is spread over the chip, one launch per
    dependency level and one workgroup per atom (mp_dictionary_update_levels_f32); without them, or with
    one_by_one (tests), it runs as one launch of one workgroup (mp_dictionary_update_f32).  Bit-identical."""
    _require_cuda(residual, d_work)
    assert residual.dtype == torch.float32 and residual.is_contiguous() and d_work.is_contiguous()
    dev = residual.device
    B, N = residual.shape
    A, L = d_work.shape
    order, offsets = _i64(order, dev), _i64(offsets, dev)
    ev_batch, ev_lag = _i64(ev_batch, dev), _i64(ev_lag, dev)
    ev_rows, ev_norm = _f32(ev_rows), _f32(ev_norm)
    sparse = torch.zeros_like(residual)
    n_groups, n_events = order.numel(), ev_batch.numel()
    if host_events is not None and not one_by_one and n_groups > 0:
        import numpy as np
        level, overlap_h, n_levels = dictionary_levels(*host_events, L)
        glist = np.argsort(level, kind="stable").astype(np.int64)          # groups by level, ascending inside a level
        level_off = np.zeros(n_levels + 1, dtype=np.int64)
        level_off[1:] = np.cumsum(np.bincount(level, minlength=n_levels))
        packed = torch.from_numpy(np.concatenate([glist, overlap_h.astype(np.int64)])).to(dev)   # one small H2D copy
        glist_d = packed[:n_groups]
        overlap = packed[n_groups:].to(torch.int32)
        with torch.cuda.device(dev):
            rc = lib().mp_dictionary_update_levels_f32(
                _ptr(residual), _ptr(sparse), B, N, _ptr(d_work), A, L, _ptr(order), _ptr(offsets), n_groups,
                _ptr(ev_batch), _ptr(ev_lag), _ptr(ev_rows), _ptr(ev_norm), ctypes.c_float(eps), _ptr(overlap),
                _ptr(glist_d), level_off.ctypes.data, n_levels, _stream(residual))
        _check(rc, "mp_dictionary_update_levels_f32")
        for t in (sparse, packed, overlap):
            t.record_stream(torch.cuda.current_stream(dev))
        return n_levels
    # which atoms have two events sharing a sample (same segment, lags < L apart)?  Sort the events by
    # (group, segment, lag) and look at neighbours -- a few small device operations, no synchronisation
    overlap = torch.zeros(max(n_groups, 1), dtype=torch.int32, device=dev)
    if one_by_one:  # (tests) every atom's events one after another, as if they all overlapped
        overlap.fill_(1)
    elif n_events > 1:
        group = torch.searchsorted(offsets[1:].contiguous(), torch.arange(n_events, device=dev), right=True)
        key, _ = torch.sort((group * B + ev_batch) * N + ev_lag)
        near = ((key[1:] // N) == (key[:-1] // N)) & ((key[1:] - key[:-1]) < L)
        overlap.scatter_reduce_(0, key[1:] // (N * B), near.to(torch.int32), "amax")
    with torch.cuda.device(dev):
        rc = lib().mp_dictionary_update_f32(_ptr(residual), _ptr(sparse), B, N, _ptr(d_work), A, L, _ptr(order),
                                            _ptr(offsets), order.numel(), _ptr(ev_batch), _ptr(ev_lag), _ptr(ev_rows),
                                            _ptr(ev_norm), ctypes.c_float(eps), _ptr(overlap), _stream(residual))
    _check(rc, "mp_dictionary_update_f32")
    sparse.record_stream(torch.cuda.current_stream(dev))

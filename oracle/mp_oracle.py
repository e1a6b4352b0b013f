"""ctypes wrapper + float64 arbiter for the CPU oracle (oracle/mp_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of mp_oracle.c.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product never imports it.

The C library restates /root/reference/modules/matchingpursuit.py (file:line citations are
in mp_oracle.c).  `arbiter_feature_map` below is an independent float64 numpy restatement of
the same correlation (modules/conv.py:4-9) used to arbitrate near-ties in fixtures.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MP_ORACLE_LIB: another build of the same source (`make -C oracle asan`: the host-sanitizer build, tests only)
_LIB_PATH = os.environ.get("MP_ORACLE_LIB") or os.path.join(_HERE, "_build", "libmp_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with the Makefile beside this file (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mp_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if os.environ.get("MP_ORACLE_LIB") else []))
    return _LIB_PATH


def cpu_share():
    """CPUs this process may use: the affinity mask capped by the cgroup quota (a GPU box shows 256
    logical CPUs but grants 16; 256 OpenMP threads on 16 cores run several times slower)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.mpo_num_threads.restype = ctypes.c_int
        _lib.mpo_set_num_threads(ctypes.c_int(cpu_share()))
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_ip)


def num_threads():
    return int(lib().mpo_num_threads())


def set_num_threads(n):
    lib().mpo_set_num_threads(ctypes.c_int(int(n)))


def unit_norm(d, eps=1e-8):
    d, dp = _f(d)
    A, L = d.shape
    out = np.empty_like(d)
    rc = lib().mpo_unit_norm(dp, _i64(A), _i64(L), ctypes.c_float(eps), out.ctypes.data_as(_fp))
    assert rc == 0
    return out


def feature_map(residual, dict_unit):
    r, rp = _f(residual)
    du, dup = _f(dict_unit)
    B, N = r.shape
    A, L = du.shape
    fm = np.empty((B, A, N), dtype=np.float32)
    rc = lib().mpo_feature_map(rp, _i64(B), _i64(N), dup, _i64(A), _i64(L), fm.ctypes.data_as(_fp))
    assert rc == 0
    return fm


def encode(signal, dict_unit, n_steps):
    """-> dict(atom[B,K] i64, lag[B,K] i64, gain[B,K] f32, residual[B,N] f32, top2[B,K,2] f32)."""
    s, sp = _f(signal)
    du, dup = _f(dict_unit)
    B, N = s.shape
    A, L = du.shape
    K = int(n_steps)
    atom = np.zeros((B, K), dtype=np.int64)
    lag = np.zeros((B, K), dtype=np.int64)
    gain = np.zeros((B, K), dtype=np.float32)
    residual = np.zeros((B, N), dtype=np.float32)
    top2 = np.zeros((B, K, 2), dtype=np.float32)
    rc = lib().mpo_encode(sp, _i64(B), _i64(N), dup, _i64(A), _i64(L), ctypes.c_int(K),
                          atom.ctypes.data_as(_ip), lag.ctypes.data_as(_ip),
                          gain.ctypes.data_as(_fp), residual.ctypes.data_as(_fp),
                          top2.ctypes.data_as(_fp))
    if rc != 0:
        raise RuntimeError(f"mpo_encode failed rc={rc}")
    return dict(atom=atom, lag=lag, gain=gain, residual=residual, top2=top2)


def encode_lcn(signal, dict_unit, n_steps):
    """sparse_code(local_contrast_norm=True), matchingpursuit.py:284-294 -> dict(atom, lag, gain, residual)."""
    s, sp = _f(signal)
    du, dup = _f(dict_unit)
    B, N = s.shape
    A, L = du.shape
    K = int(n_steps)
    atom = np.zeros((B, K), dtype=np.int64)
    lag = np.zeros((B, K), dtype=np.int64)
    gain = np.zeros((B, K), dtype=np.float32)
    residual = np.zeros((B, N), dtype=np.float32)
    rc = lib().mpo_encode_lcn(sp, _i64(B), _i64(N), dup, _i64(A), _i64(L), ctypes.c_int(K),
                              atom.ctypes.data_as(_ip), lag.ctypes.data_as(_ip),
                              gain.ctypes.data_as(_fp), residual.ctypes.data_as(_fp))
    if rc != 0:
        raise RuntimeError(f"mpo_encode_lcn failed rc={rc}")
    return dict(atom=atom, lag=lag, gain=gain, residual=residual)


def scatter(atom, batch, lag, gain, dict_unit, B, N):
    a, ap = _i(np.ravel(atom))
    b, bp = _i(np.ravel(batch))
    p, pp = _i(np.ravel(lag))
    g, gp = _f(np.ravel(gain))
    du, dup = _f(dict_unit)
    A, L = du.shape
    out = np.zeros((B, N), dtype=np.float32)
    rc = lib().mpo_scatter(ap, bp, pp, gp, _i64(a.size), dup, _i64(A), _i64(L),
                           out.ctypes.data_as(_fp), _i64(B), _i64(N))
    if rc != 0:
        raise RuntimeError(f"mpo_scatter failed rc={rc}")
    return out


def scatter_rows(rows, batch, lag, B, N):
    r, rp = _f(rows)
    b, bp = _i(np.ravel(batch))
    p, pp = _i(np.ravel(lag))
    n, L = r.shape
    out = np.zeros((B, N), dtype=np.float32)
    rc = lib().mpo_scatter_rows(rp, bp, pp, _i64(n), _i64(L), out.ctypes.data_as(_fp), _i64(B), _i64(N))
    if rc != 0:
        raise RuntimeError(f"mpo_scatter_rows failed rc={rc}")
    return out


def dictionary_learning_step(signal, d_raw, n_steps):
    s, sp = _f(signal)
    d, dp = _f(d_raw)
    B, N = s.shape
    A, L = d.shape
    out = np.empty_like(d)
    rc = lib().mpo_dictionary_learning_step(sp, _i64(B), _i64(N), dp, _i64(A), _i64(L),
                                            ctypes.c_int(int(n_steps)), out.ctypes.data_as(_fp))
    if rc != 0:
        raise RuntimeError(f"mpo_dictionary_learning_step failed rc={rc}")
    return out


# ----------------------------------------------------------------------------------------------
# numpy restatements on top of the C loop (small inputs only)
# ----------------------------------------------------------------------------------------------
def sparse_feature_map(signal, d_raw, n_steps):
    """modules/matchingpursuit.py:68-125, forward values: the loop is sparse_code's (direct correlation :88-91,
    signed first-max argmax :103, cropped subtraction :115-120) and `fm += soft_dirac(f) * f` (:100-101) adds the
    map's value at the argmax, soft_dirac's forward being the argmax's one-hot (modules/sparse.py:29-43).
    -> (fm [B, A, N] float32, residual [B, N])."""
    s = np.ascontiguousarray(signal, dtype=np.float32).reshape(len(signal), -1)
    du = unit_norm(d_raw)                                                   # :80
    out = encode(s, du, n_steps)
    B, N = s.shape
    fm = np.zeros((B, du.shape[0], N), dtype=np.float32)
    for k in range(int(n_steps)):                                           # one float32 add per step, as :101
        fm[np.arange(B), out["atom"][:, k], out["lag"][:, k]] += out["gain"][:, k]
    return fm, out["residual"]


def sparse_coding_loss(recon, target, d_raw, n_steps):
    """modules/matchingpursuit.py:128-146: binary cross entropy (mean over all B*A*N cells, logs clamped at
    -100 as torch.nn.functional.binary_cross_entropy does) between the two maps divided by their joint maximum."""
    r_map, _ = sparse_feature_map(recon, d_raw, n_steps)
    t_map, _ = sparse_feature_map(target, d_raw, n_steps)
    mx = max(float(r_map.max()), float(t_map.max()))                        # :139-141
    r = (r_map / np.float32(mx)).astype(np.float64)
    t = (t_map / np.float32(mx)).astype(np.float64)
    with np.errstate(divide="ignore"):
        log_r = np.maximum(np.log(r), -100.0)
        log_1r = np.maximum(np.log1p(-r), -100.0)
    return float(-(t * log_r + (1.0 - t) * log_1r).mean())


def fft_convolve(signal, dict_unit, approx=None):
    """modules/conv.py:11-53 with numpy's FFT (float64 inside): signal [B, 1, N] -> fm [B, A, N].
    approx=slice keeps the slice's bins (:24-29); approx=int < N keeps, per segment, the `approx` largest-magnitude
    bins of the SIGNAL spectrum -- and, the index tensor being [B, 1, k], only atom 0's row of the product is ever
    filled (:30-47: gather and scatter both follow the index's shape); anything else is the exact map (:48-49)."""
    s = np.asarray(signal, dtype=np.float64)
    d = np.asarray(dict_unit, dtype=np.float64)
    B, _, N = s.shape
    A, L = d.shape
    M = N + L
    sig = np.fft.rfft(np.pad(s, ((0, 0), (0, 0), (0, L))), axis=-1)         # [B, 1, F]
    atom = np.fft.rfft(np.pad(d, ((0, 0), (0, M - L)))[:, ::-1], axis=-1)[None]  # [1, A, F]
    if isinstance(approx, slice):
        spec = np.zeros((B, A, sig.shape[-1]), dtype=np.complex128)
        spec[..., approx] = sig[..., approx] * atom[..., approx]
    elif isinstance(approx, int) and approx < N:
        spec = np.zeros((B, A, sig.shape[-1]), dtype=np.complex128)
        idx = np.argsort(-np.abs(sig[:, 0, :]), axis=-1, kind="stable")[:, :approx]     # [B, k]
        rows = np.arange(B)[:, None]
        spec[rows, 0, idx] = sig[rows, 0, idx] * atom[0, 0, idx]
    else:
        spec = sig * atom
    fm = np.roll(np.fft.irfft(spec, n=M, axis=-1), 1, axis=-1)
    return fm[..., :N].astype(np.float32)


# ----------------------------------------------------------------------------------------------
# float64 arbiter (pure numpy; small inputs only)
# ----------------------------------------------------------------------------------------------
def arbiter_feature_map(residual, dict_unit):
    """fm[b,a,t] = sum_k r[b,t+k] d[a,k] in float64 (modules/conv.py:4-9), r zero beyond N."""
    r = np.asarray(residual, dtype=np.float64)
    d = np.asarray(dict_unit, dtype=np.float64)
    B, N = r.shape
    A, L = d.shape
    rp = np.concatenate([r, np.zeros((B, L))], axis=1)
    win = np.lib.stride_tricks.sliding_window_view(rp, L, axis=1)[:, :N, :]  # [B,N,L]
    return np.einsum("bnl,al->ban", win, d, optimize=True)

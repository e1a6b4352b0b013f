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
_LIB_PATH = os.path.join(_HERE, "_build", "libmp_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with the Makefile beside this file (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mp_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
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

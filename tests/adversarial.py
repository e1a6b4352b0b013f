"""Adversarial inputs for the FFT screen's error bound (test infrastructure; used by tests/test_gpu_parity.py,
tests/fuzz_parity.py and scripts/screen_audit.py).

The bound must cover |screen - fp32 fma chain|.  The chain's own rounding is largest where its partial sums grow
monotonically and the roundings share a sign: SAME-SIGN atoms on a DC-OFFSET signal.  The transforms' rounding is
relative to the window's energy: one HUGE TRANSIENT inside a window of tiny samples makes every other cell of that
window small against the bound."""
import numpy as np


def same_sign_dictionary(A, L, seed):
    """Positive, smooth-ish atoms (|noise| under a raised-cosine envelope), rows NOT normalised."""
    rng = np.random.default_rng(seed)
    env = 0.55 - 0.45 * np.cos(2 * np.pi * (np.arange(L) + 0.5) / L)
    d = np.abs(rng.standard_normal((A, L))) * env[None, :] + 0.05
    return d.astype(np.float32)


def dc_offset_segments(B, N, d_unit, n_events, seed, dc=0.3):
    """A constant offset plus planted events with well separated gains (so the first n_events steps have an
    unambiguous maximum on top of the large common pedestal dc * sum(atom))."""
    rng = np.random.default_rng(seed)
    A, L = d_unit.shape
    x = np.full((B, N), dc, dtype=np.float64)
    for b in range(B):
        gains = np.linspace(2.0, 0.6, n_events) * rng.uniform(0.97, 1.03, n_events)
        for g in gains:
            a = int(rng.integers(0, A))
            p = int(rng.integers(0, max(N - L, 1)))
            n = min(L, N - p)
            x[b, p:p + n] += g * d_unit[a, :n]
    return x.astype(np.float32)


def transient_segments(B, N, d_unit, n_events, seed, quiet=1e-4, loud=1e3):
    """Tiny samples (planted events and noise at `quiet`) with ONE transient of amplitude `loud` (a scaled atom)."""
    rng = np.random.default_rng(seed)
    A, L = d_unit.shape
    x = quiet * 0.05 * rng.standard_normal((B, N))
    for b in range(B):
        for g in np.linspace(2.0, 0.6, n_events) * rng.uniform(0.97, 1.03, n_events):
            a = int(rng.integers(0, A))
            p = int(rng.integers(0, max(N - L, 1)))
            n = min(L, N - p)
            x[b, p:p + n] += quiet * g * d_unit[a, :n]
        a = int(rng.integers(0, A))
        p = int(rng.integers(0, max(N - L, 1)))
        n = min(L, N - p)
        x[b, p:p + n] += loud * d_unit[a, :n]
    return x.astype(np.float32)

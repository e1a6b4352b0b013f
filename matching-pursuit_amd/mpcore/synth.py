"""Deterministic synthetic inputs for parity tests and benchmarks.

There is no network and no audio corpus on the build or GPU boxes, so every measured or
tested workload is generated here from `numpy.random.Generator(PCG64(seed))`, whose stream
is stable across numpy versions.

Dictionary: A x L i.i.d. U(-1, 1) -- the initialisation the reference's experiments use
(/root/reference/experiments/archive/e_2023_3_19/experiment.py:30-31,
 /root/reference/modules/matchingpursuit.py:439).  The encoder normalises it itself.

Segments ("MusicNet-shaped": 22050 Hz mono, peak-normalised like
/root/reference/data/datastore.py:152-153): a sum of planted unit atoms at random lags and
gains, a bed of decaying harmonic notes at -12 dB, white noise at -40 dB, divided by the
peak.  Pure white noise is a useless MP input (nothing to find), hence the structure.
"""
import numpy as np

SAMPLE_RATE = 22050


def make_dictionary(n_atoms, atom_size, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(-1.0, 1.0, size=(n_atoms, atom_size)).astype(np.float32)


def _unit_rows(d):
    d64 = d.astype(np.float64)
    return d64 / (np.linalg.norm(d64, axis=-1, keepdims=True) + 1e-8)


def make_segments(batch, n_samples, dictionary, n_events=24, seed=0, first_index=0):
    """-> float32 [batch, n_samples].  Segment i depends only on (seed, first_index + i),
    so a rank that owns a shard of a larger batch generates exactly its own rows."""
    n_atoms, atom_size = dictionary.shape
    du = _unit_rows(dictionary)
    out = np.zeros((batch, n_samples), dtype=np.float64)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64([seed, first_index + i]))
        x = np.zeros(n_samples, dtype=np.float64)
        hi = max(1, n_samples - atom_size + 1)
        for _ in range(n_events):
            a = int(rng.integers(0, n_atoms))
            p = int(rng.integers(0, hi))
            g = float(rng.uniform(0.5, 2.0))
            seg = du[a][: max(0, min(atom_size, n_samples - p))]
            x[p:p + seg.shape[0]] += g * seg
        bed = np.zeros(n_samples, dtype=np.float64)
        for _ in range(4):
            midi = float(rng.integers(40, 89))
            f0 = 440.0 * 2.0 ** ((midi - 69.0) / 12.0)
            onset = float(rng.uniform(0.0, 0.6)) * n_samples / SAMPLE_RATE
            decay = float(rng.uniform(0.2, 1.0))
            env = np.where(t >= onset, np.exp(-(t - onset) / decay), 0.0)
            for h in range(1, 9):
                if f0 * h < SAMPLE_RATE / 2:
                    bed += env * np.sin(2 * np.pi * f0 * h * t + float(rng.uniform(0, 2 * np.pi))) / h
        peak_x = np.max(np.abs(x)) + 1e-12
        bed *= (10 ** (-12 / 20)) * peak_x / (np.max(np.abs(bed)) + 1e-12)
        noise = rng.standard_normal(n_samples) * (10 ** (-40 / 20)) * peak_x
        x = x + bed + noise
        out[i] = x / (np.max(x) + 1e-12)
    return out.astype(np.float32)

"""Multiband matching pursuit (/root/reference/modules/multibanddict.py:53-473): one dictionary per octave
band, each band encoded / learned by the native `sparse_code` / `dictionary_learning_step`.

Host-side composition only -- the per-band work is the hot path already built.  Same public names, method
signatures, event wire formats and on-disk dictionary format (`band_{size}.dat`, a `torch.save`d [A, L]
tensor) as the reference; `samplerate` is accepted and stored but not interpreted (the reference only uses it
as a default argument, multibanddict.py:63).
"""
import os
from collections import Counter, defaultdict
from concurrent.futures import ThreadPoolExecutor
from hashlib import sha256
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .decompose import fft_frequency_decompose, fft_frequency_recompose, fft_resample
from .matchingpursuit import build_scatter_segments, dictionary_learning_step, sparse_code, unit_norm

LocalEventTuple = Tuple[int, int, int, torch.Tensor]      # (atom, batch, sample position, scaled atom)
GlobalEventTuple = Tuple[int, int, float, float]          # (global atom, batch, unit time, amplitude)
BandEncodingPackage = Tuple[List[LocalEventTuple], Callable, Tuple]


class BandSpec(object):
    """One band: its length `size`, its dictionary `d` [n_atoms, atom_size] (multibanddict.py:53-279)."""

    def __init__(self, size: int, n_atoms: int, atom_size: int, slce: Optional[slice] = None, device=None,
                 signal_samples: int = 0, samplerate=None, local_contrast_norm: bool = False,
                 is_lowest_band: bool = False):
        super().__init__()
        self.size, self.n_atoms, self.atom_size = size, n_atoms, atom_size
        self.slce, self.device = slce, device
        self.signal_samples, self.samplerate = signal_samples, samplerate
        self.local_contrast_norm, self.is_lowest_band = local_contrast_norm, is_lowest_band
        fresh = torch.zeros(n_atoms, atom_size, requires_grad=False).uniform_(-1, 1)
        self.d = unit_norm(fresh.to(device) if device is not None else fresh)

    def __hash__(self):
        return hash(sha256(self.d.data.cpu().numpy()).hexdigest())

    # ---- geometry ---------------------------------------------------------------------------------------
    @property
    def n_samples_at_native_rate(self):
        return self.atom_size * (self.signal_samples // self.size)

    def resampled_atoms(self) -> torch.Tensor:
        return fft_resample(self.d.view(self.n_atoms, 1, self.atom_size), self.n_samples_at_native_rate,
                            self.is_lowest_band)

    def shape(self, batch_size):
        return (batch_size, 1, self.size)

    @property
    def filename(self):
        return f"band_{self.size}.dat"

    @property
    def scatter_func(self):
        return build_scatter_segments(self.size, self.atom_size)

    def get_atom(self, index: int, norm: float):
        return self.d[index] * norm

    # ---- persistence ------------------------------------------------------------------------------------
    def load(self):
        try:
            self.d = torch.load(self.filename)
        except IOError:
            pass

    def store(self):
        torch.save(self.d, self.filename)

    # ---- the hot path -----------------------------------------------------------------------------------
    def learn(self, batch, steps=16):
        d = dictionary_learning_step(batch, self.d, steps, device=self.device, approx=self.slce,
                                     local_constrast_norm=self.local_contrast_norm)
        self.d = unit_norm(d)
        return d

    def encode(self, batch, steps=16, extract_embeddings=None) -> BandEncodingPackage:
        out = sparse_code(batch, self.d, steps, device=self.device, approx=self.slce, flatten=True,
                          extract_atom_embedding=extract_embeddings,
                          local_contrast_norm=self.local_contrast_norm)
        if extract_embeddings:
            return out                       # (embeddings, residual)
        events, scatter = out
        return events, scatter, batch.shape

    def decode(self, shape, all_instances, scatter):
        return scatter(shape, all_instances)

    def recon(self, batch, steps=16):
        events, scatter, shape = self.encode(batch, steps)
        return self.decode(shape, events, scatter), events, scatter

    # ---- event wire formats (multibanddict.py:189-235) ---------------------------------------------------
    def to_global_atom_index(self, index: int, offset: int) -> int:
        return offset + index

    def to_local_atom_index(self, index: int, offset: int) -> int:
        return index - offset

    def to_unit_time(self, sample_position) -> float:
        return sample_position / self.size

    def to_sample_time(self, unit_time: float) -> int:
        return int(unit_time * self.size)

    def to_amplitude(self, scaled_atom: torch.Tensor):
        return torch.norm(scaled_atom)

    def to_global_tuple(self, event: LocalEventTuple, offset: int) -> GlobalEventTuple:
        atom_index, batch, sample_pos, atom = event
        return (self.to_global_atom_index(atom_index, offset), batch, self.to_unit_time(sample_pos),
                self.to_amplitude(atom))

    def to_local_tuple(self, event: GlobalEventTuple, offset: int) -> LocalEventTuple:
        global_index, batch, unit_time, amplitude = event
        local = self.to_local_atom_index(global_index, offset)
        return (local, batch, self.to_sample_time(unit_time), self.get_atom(local, amplitude))


class MultibandDictionaryLearning(object):
    """All bands of a signal of `n_samples` samples (multibanddict.py:282-473)."""

    def __init__(self, specs: List[BandSpec], n_samples: int):
        super().__init__()
        self.bands = {spec.size: spec for spec in specs}
        self.min_size = min(spec.size for spec in specs)
        self.n_samples = n_samples
        counts = {spec.n_atoms for spec in specs}
        if len(counts) > 1:
            raise ValueError("Only specs with equal atom counts is currently allowed")
        self.n_atoms = counts.pop()

    def __len__(self):
        return len(self.bands)

    def __hash__(self):
        return hash(tuple(hash(b) for b in self.bands.values()))

    def event_count(self, iterations: int) -> int:
        return len(self) * iterations

    # ---- lookups ----------------------------------------------------------------------------------------
    @property
    def total_atoms(self):
        return sum(b.n_atoms for b in self.bands.values())

    @property
    def band_dicts(self):
        return {size: band.d for size, band in self.bands.items()}

    @property
    def band_sizes(self):
        return list(self.bands.keys())

    def get_atom(self, size, index, norm):
        return self.bands[size].get_atom(index, norm)

    def size_at_index(self, index):
        return self.band_sizes[index]

    def index_of_size(self, band_size):
        return [b.size for b in self.bands.values()].index(band_size)

    def index_of_dict_size(self, size):
        for i, d in enumerate(self.band_dicts.values()):
            if size == d.shape[-1]:
                return i
        raise IndexError(f"{size} not found in {self.shape_dict(1)}")

    def shape_dict(self, batch_size):
        return {size: band.shape(batch_size) for size, band in self.bands.items()}

    def partial_decoding_dict(self, batch_size):
        return {size: (build_scatter_segments(size, band.atom_size), (batch_size, 1, size))
                for size, band in self.bands.items()}

    def get_band_from_global_atom_index(self, index: int):
        band_index = index // self.n_atoms
        return band_index, list(self.bands.values())[band_index]

    def atom_embeddings(self):
        return torch.eye(self.total_atoms, device=next(iter(self.bands.values())).d.device)

    def event_embeddings(self, batch_size: int, events: List[GlobalEventTuple], atom_embeddings) -> torch.Tensor:
        with torch.no_grad():
            per_item = len(events) // batch_size
            out = torch.zeros(batch_size, per_item, atom_embeddings.shape[-1], device=atom_embeddings.device)
            seen = defaultdict(Counter)
            for global_index, batch, unit_time, amplitude in events:
                band_index, _ = self.get_band_from_global_atom_index(global_index)
                slot = seen[batch][band_index]
                seen[batch][band_index] += 1
                out[batch, slot, :] = atom_embeddings[global_index] * amplitude.view(1)
            return out

    # ---- persistence ------------------------------------------------------------------------------------
    def store(self):
        for band in self.bands.values():
            band.store()

    def load(self):
        for band in self.bands.values():
            band.load()

    # ---- the hot path -----------------------------------------------------------------------------------
    # The bands are independent encodes of very different sizes, and every one of them is a chain of steps that cannot fill
    # the GPU by itself at the batch sizes this model runs with (e_2023_3_8: 8 .. 16 segments; the 4096-sample band's
    # select is one workgroup per SEGMENT).  The reference walks them one after the other (multibanddict.py:318-330);
    # here each band runs on a stream of its own, driven by a host thread of its own (the per-band calls synchronise with
    # the host -- the overflow check, the event order -- so one thread cannot keep seven streams fed; the library is
    # built for concurrent encodes from several threads, tests/test_gpu_parity.py::
    # test_two_threads_run_the_persistent_form_concurrently).  Same results, band by band: nothing is shared.
    # MP_BANDS_SEQUENTIAL=1: the reference's loop.
    _pool = None

    def _map_bands(self, fn, split):
        sizes = list(self.bands)
        first = split[sizes[0]]
        if (len(sizes) < 2 or not torch.is_tensor(first) or first.device.type != "cuda"
                or os.environ.get("MP_BANDS_SEQUENTIAL") == "1"):
            return {size: fn(self.bands[size], split[size]) for size in sizes}
        dev = first.device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_streams", None) is None or self._streams[0] != dev or len(self._streams[1]) != len(sizes):
            self._streams = (dev, [torch.cuda.Stream(dev) for _ in sizes])
        if MultibandDictionaryLearning._pool is None:
            MultibandDictionaryLearning._pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="mp-band")
        grad = torch.is_grad_enabled()

        def run(i):
            stream = self._streams[1][i]
            with torch.cuda.device(dev), torch.cuda.stream(stream), torch.set_grad_enabled(grad):
                stream.wait_stream(cur)        # the band's signal was made on the caller's stream
                return fn(self.bands[sizes[i]], split[sizes[i]])

        # longest band first: it is the critical path the others hide behind
        order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
        futures = {i: MultibandDictionaryLearning._pool.submit(run, i) for i in order}
        out = {sizes[i]: futures[i].result() for i in range(len(sizes))}
        for stream in self._streams[1]:
            cur.wait_stream(stream)            # the caller's stream sees every band's results
        return out

    def learn(self, batch, steps=16):
        self._map_bands(lambda band, x: band.learn(x, steps), fft_frequency_decompose(batch, self.min_size))

    def encode(self, batch, steps, extract_embeddings=None) -> Dict[int, BandEncodingPackage]:
        split = fft_frequency_decompose(batch, self.min_size)
        return self._map_bands(lambda band, x: band.encode(x, steps, extract_embeddings), split)

    def decode(self, d, shapes=None):
        per_band = {}
        for size, pack in d.items():
            if shapes is not None:
                events, scatter, shape = pack, self.bands[size].scatter_func, shapes[size]
            else:
                events, scatter, shape = pack
            per_band[size] = self.bands[size].decode(shape, events, scatter)
        return fft_frequency_recompose(per_band, self.n_samples)

    def recon(self, batch, steps=16):
        split = fft_frequency_decompose(batch, self.min_size)
        per_band = self._map_bands(lambda band, x: band.recon(x, steps), split)
        recon_bands = {size: per_band[size][0] for size in self.bands}
        events = {size: per_band[size][1] for size in self.bands}
        return fft_frequency_recompose(recon_bands, batch.shape[-1]), events

    # ---- global <-> per-band event lists (multibanddict.py:404-441) ---------------------------------------
    def flattened_event_tuples(self, encoding: Dict[int, BandEncodingPackage]) -> List[GlobalEventTuple]:
        out, offset = [], 0
        for size, (events, _, _) in encoding.items():
            band = self.bands[size]
            out.extend(band.to_global_tuple(e, offset) for e in events)
            offset += band.n_atoms
        return out

    def hierarchical_event_tuples(self, encoding: List[GlobalEventTuple],
                                  original: Dict[int, BandEncodingPackage]) -> Dict[int, BandEncodingPackage]:
        per_band = defaultdict(list)
        for event in encoding:
            index, band = self.get_band_from_global_atom_index(event[0])
            per_band[band.size].append(band.to_local_tuple(event, index * self.n_atoms))
        return {size: (events, original[size][1], original[size][2]) for size, events in per_band.items()}

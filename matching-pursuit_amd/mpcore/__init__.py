"""mpcore -- MI355X-native greedy matching pursuit."""
from . import _native  # noqa: F401
from ._native import EncodePlan, NativeError  # noqa: F401
from .matchingpursuit import (  # noqa: F401
    build_scatter_segments, flatten_atom_dict, sparse_code, dictionary_learning_step,
    sparse_feature_map, sparse_coding_loss, SparseCodingLoss, unit_norm, torch_conv, fft_convolve,
    EventList, encode_packed, sparse_feature_map_coo)
from .iterative import iterative_loss, sort_channels_descending_norm  # noqa: F401
from .streaming import encode_streaming, decode_streaming, StreamCode  # noqa: F401
from .overlay import install, uninstall  # noqa: F401  (drop-in overlay over the reference's `modules` package)

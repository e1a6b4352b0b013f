"""Drop-in `modules` package: the subset of the reference's `modules` surface that lies on the
matching-pursuit hot path, backed by libmpcore (HIP, gfx950).  Put `matching-pursuit_amd/` on
sys.path ahead of the reference checkout and

    from modules.matchingpursuit import sparse_code, dictionary_learning_step   # mp.py:17
    from modules import iterative_loss                                           # iterativedecomposition.py:12

resolve here unchanged.  Names mirror /root/reference/modules/__init__.py:18-21.
"""
from .normalization import unit_norm  # noqa: F401
from .conv import fft_convolve  # noqa: F401
from .matchingpursuit import (  # noqa: F401
    build_scatter_segments, dictionary_learning_step, flatten_atom_dict, sparse_coding_loss,
    sparse_feature_map, SparseCodingLoss)
from .iterative import iterative_loss, sort_channels_descending_norm  # noqa: F401
from .sparse import soft_dirac, sparsify2  # noqa: F401

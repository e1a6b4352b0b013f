"""`modules.matchingpursuit` drop-in (/root/reference/modules/matchingpursuit.py)."""
from mpcore.matchingpursuit import (  # noqa: F401
    build_scatter_segments, flatten_atom_dict, sparse_code, dictionary_learning_step,
    sparse_feature_map, sparse_coding_loss, SparseCodingLoss, sparse_code_to_differentiable_key_points)

"""`modules.sparse` drop-in for the helpers on the matching-pursuit surface
(/root/reference/modules/sparse.py:29-89)."""
from mpcore.sparse import soft_dirac, sparsify2  # noqa: F401

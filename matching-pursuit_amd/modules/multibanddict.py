"""`modules.multibanddict` drop-in (/root/reference/modules/multibanddict.py:53-473)."""
from mpcore.multibanddict import (  # noqa: F401
    BandEncodingPackage, BandSpec, GlobalEventTuple, LocalEventTuple, MultibandDictionaryLearning)

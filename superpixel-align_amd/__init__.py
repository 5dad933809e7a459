"""spalign-mi355x: MI355X-native superpixel-align label generation.

The directory name carries a hyphen (it is fixed by the project layout), so import it with
``importlib.import_module('superpixel-align_amd')`` (the root-level scripts batch_spalign_kmeans.py,
direct_clustering.py, superpixel_overlaps.py and utils/apply_spalign_kmeans.py do exactly that).
"""
from . import synth  # noqa: F401
from . import _lib  # noqa: F401
from ._lib import SpalignError  # noqa: F401

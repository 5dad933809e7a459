"""spalign-mi355x: MI355X-native superpixel-align label generation.

The directory name carries a hyphen (it is fixed by the project layout), so import it with
``importlib.import_module('superpixel-align_amd')`` or through the ``spalign`` shim at the
repository root.
"""
from . import synth  # noqa: F401
from . import _lib  # noqa: F401
from ._lib import SpalignError  # noqa: F401

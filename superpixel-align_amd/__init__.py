"""spalign-mi355x: MI355X-native superpixel-align label generation."""
from . import synth  # noqa: F401

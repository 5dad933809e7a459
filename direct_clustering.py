#!/usr/bin/env python
"""Baseline: weighted k-means over all feature pixels (same CLI and outputs as the reference
script of this name), on the MI355X kernels.  See superpixel-align_amd/baselines.py."""
import importlib
import sys

if __name__ == '__main__':
    sys.exit(importlib.import_module('superpixel-align_amd.cli').main_direct())

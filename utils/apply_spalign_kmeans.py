#!/usr/bin/env python
"""Label-free driver (same CLI and outputs as the reference script of this name): writes one
{0,1} PNG mask per input image.  See superpixel-align_amd/cli.py."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    sys.exit(importlib.import_module('superpixel-align_amd.cli').main_labelfree())

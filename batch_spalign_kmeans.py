#!/usr/bin/env python
"""Labelled label-generation driver (same CLI and outputs as the reference script of this name),
running on the MI355X pipeline.  See superpixel-align_amd/cli.py."""
import importlib
import sys

if __name__ == '__main__':
    sys.exit(importlib.import_module('superpixel-align_amd.cli').main_labelled())

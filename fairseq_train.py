#!/usr/bin/env python3
"""fairseq-train entry point of this build (chimera/scripts/*.sh call `fairseq-train <data> ...`)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    importlib.import_module("chimera-st_amd.cli").train_main()

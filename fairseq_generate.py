#!/usr/bin/env python3
"""fairseq-generate entry point of this build (chimera/generate/*.sh call `fairseq-generate <data> --path ...`)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    importlib.import_module("chimera-st_amd.cli").generate_main()

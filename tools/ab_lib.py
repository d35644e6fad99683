#!/usr/bin/env python3
"""Run a tool script against another build of the library:  CST_AB_LIB=<path to .so> python tools/ab_lib.py tools/<script>.py [args]
(same-box A/B of two builds: box-to-box spread on the pool is +-4 %, larger than most kernel changes)."""
import importlib, os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("chimera-st_amd.lib")
L.LIB_PATH = os.environ["CST_AB_LIB"]
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

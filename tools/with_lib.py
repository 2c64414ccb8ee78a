"""Run another tool with a variant build of the library: python tools/with_lib.py <lib.so in the package dir> tools/<tool>.py [args]"""
import os, runpy, sys
sys.path.insert(0, os.getcwd())
from gpu_fluid_simulation_amd import _abi
_abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", sys.argv[1]))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

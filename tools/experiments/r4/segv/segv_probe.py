"""Runs bench.py's main() with segv_probe.so's handler installed (after torch and the profiler's tool library have installed theirs):
    rocprofv3 --pmc SQ_INSTS_VALU -d ... -- python3 tools/experiments/r4/segv/segv_probe.py <report file> <bench.py args...>"""
import ctypes
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
report = sys.argv[1]
import torch  # noqa: E402,F401  (its handlers first)
probe = ctypes.CDLL(os.path.join(HERE, "segv_probe.so"))
probe.segv_probe_install.argtypes = [ctypes.c_char_p]
assert probe.segv_probe_install(report.encode()) == 0
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
sys.path.insert(0, ROOT)
runpy.run_path(sys.argv[0], run_name="__main__")

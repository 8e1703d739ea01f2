"""A/B of two builds of libnsx.so on one box: runs bench.py with `nersemble_amd._lib.SO_PATH` pointed at another library.
    python tools/ab_lib.py nersemble_amd/csrc/libnsx_prev.so --workload static_h1 --steps 20 ...   (bench.py's arguments)"""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import nersemble_amd._lib as L  # noqa: E402

L.SO_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

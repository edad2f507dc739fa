"""Experiment: bench.py with the aggregate-before-project form of layer 0 switched off (project first, 3 x 250-wide gathers).
Single GPU: equal within noise (19.76-19.82 vs 19.80-19.99 ms); the aggregate-first form is kept for its 4.5x narrower halo rows."""
import os
import sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd.nn import fused
fused.AGG_FIRST = False
sys.argv = ["bench.py", "--cpu-baseline", "off"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")

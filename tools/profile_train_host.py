#!/usr/bin/env python3
"""Developer tool: host-side (Python) profile of the training step of bench.py c4."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "c4", "--steps", "40", "--warmup", "8", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])

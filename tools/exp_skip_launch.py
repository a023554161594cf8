import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/dfa-nerf_amd")
sys.argv = ["bench.py", "--workload", "c4", "--steps", "600", "--warmup", "50", "--no-cpu-baseline", "--no-extra", "--sustain-seconds", "0"]
from dfanerf import training
skip = os.environ.get("SKIP", "").split(",")
class P:
    def __init__(s, l): s._l = l
    def __getattr__(s, k):
        f = getattr(s._l, k)
        if k in skip:
            return lambda *a: 0
        return f
training.lib = P(training.lib)
import bench
bench.main()

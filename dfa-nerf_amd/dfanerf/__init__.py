"""dfanerf - MI355X-native rendering path of DFA-NeRF.

`dfanerf.engine` drives the HIP library (libdfanerf.so) through its C ABI; `dfanerf.synth` generates the
deterministic synthetic weights/scene used by tests and the bench.  Importing `dfanerf.engine` fails loudly
if the library has not been built: there is no CPU fallback."""
__version__ = "0.2"

import os as _os

# One process per GPU under a launcher (WORLD_SIZE > 1): the collective backend brings a stream of its own next to the
# training step's four (main, weight gradients, two conditioning-network chains), and the HIP runtime multiplexes streams
# onto FOUR hardware queues by default - the weight-gradient stream then shares the main stream's queue and the overlap of
# the step is gone (measured through RCCL on one MI355X: 1.36 -> 1.32 ms per step with eight queues; a single-rank process
# is 2 % slower with eight, so it keeps the default).  Read by the runtime when it initialises, i.e. at the first GPU call:
# this import has to come before it (bench.py and run_nerf.py import the package before they touch the device).
# The functional modes that put EVERY rank on one GPU (DFN_ONE_GPU / DFN_BENCH_ONE_GPU: gloo, tests) share the device's queues:
# eight processes x eight queues oversubscribe it - the CLI at world 8 then died in five runs of six with a GPU memory fault in
# whatever kernel touched freshly allocated memory (an ATen fill / copy), and in none of three with two queues per process
# (round 5, profiles/r05e_world8_queues.txt) - so they get 16 / world queues each, at least two.
_world = int(_os.environ.get("WORLD_SIZE", "1") or 1)
if _world > 1 or _os.environ.get("DFN_BENCH_RCCL_WORLD1"):
    _shared = bool(_os.environ.get("DFN_ONE_GPU") or _os.environ.get("DFN_BENCH_ONE_GPU"))
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(2, 16 // _world)) if _shared else "8")

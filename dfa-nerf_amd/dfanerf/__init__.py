"""dfanerf - MI355X-native rendering path of DFA-NeRF.

`dfanerf.engine` drives the HIP library (libdfanerf.so) through its C ABI; `dfanerf.synth` generates the
deterministic synthetic weights/scene used by tests and the bench.  Importing `dfanerf.engine` fails loudly
if the library has not been built: there is no CPU fallback."""
__version__ = "0.1"

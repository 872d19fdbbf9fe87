"""dr4sr_amd — MI355X-native hot path for DR4SR's target-model training loop (SASRec first).

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (libdr4sr_hip.so, include/dr4sr_hip.h)
  _lib.py    ctypes binding (no fallback: raises if the library is missing)
  engine.py  flat-parameter training engine on top of the C ABI (buffers, plan, hipGraph replay)
  model/ data/ utils/ quickstart/   host-side mirror of the reference's RecStudio-style interface
"""
__version__ = "0.1.0"

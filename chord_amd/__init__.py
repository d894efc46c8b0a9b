"""chord_amd — MI355X-native visibility hot path of qiutang98/chord behind a C ABI.

Sub-modules:
  records   numpy/ctypes mirrors of include/chordvis_types.h
  lib       ctypes binding of libchordvis.so (raises if the HIP library is not built)
  renderer  host-side mirror of the reference's pass surface over the C ABI
  scenes    deterministic procedural meshlet scenes (BASELINE configs)
  build     in-tree hipcc build
"""
__all__ = ["records", "scenes", "build"]

"""MI355X-native UV-dose hot path (generate / extend / accumulate / reset / shade).

The product is two shared libraries built in this directory:
  libuvrt_hip.so   HIP kernels for gfx950 behind the C ABI of include/uvrt.h
  libuvrt_host.so  C++ mirror of the reference's RayTracer / Mesh / BVH surface (host/)
`capi` and `host` are thin ctypes bindings used by tests/ and bench.py.  The directory name
is not a Python identifier; load it with `__graft_entry__.load_package()`.
"""
from . import capi  # noqa: F401

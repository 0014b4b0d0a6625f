"""ctypes binding of the C++ host layer (libuvrt_host.so: Mesh / BVH / RayTracer mirrors of
the reference's classes).  Method and field names are the reference's (raytracer.h:13-59)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuvrt_host.so")
_LIB = None

NODE_DT = np.dtype([("minx", "<f4"), ("miny", "<f4"), ("minz", "<f4"), ("leftFirst", "<i4"),
                    ("maxx", "<f4"), ("maxy", "<f4"), ("maxz", "<f4"), ("triCount", "<i4")])

VIEW_DOSAGE, VIEW_MAXPOWER, VIEW_TEXTURE = 0, 1, 2

_vp = C.c_void_p
SYMBOLS = [
    ("uvrt_host_mesh_load", _vp, [C.c_char_p]),
    ("uvrt_host_mesh_from_tris", _vp, [_vp, C.c_int]),
    ("uvrt_host_mesh_free", None, [_vp]),
    ("uvrt_host_mesh_tri_count", C.c_int, [_vp]),
    ("uvrt_host_mesh_floor_height", C.c_float, [_vp]),
    ("uvrt_host_mesh_tris", _vp, [_vp]),
    ("uvrt_host_mesh_nodes", _vp, [_vp]),
    ("uvrt_host_mesh_nodes_used", C.c_uint, [_vp]),
    ("uvrt_host_mesh_tri_idx", _vp, [_vp]),
    ("uvrt_host_mesh_rebuild_bvh", None, [_vp]),
    ("uvrt_host_rt_new", _vp, []),
    ("uvrt_host_rt_free", None, [_vp]),
    ("uvrt_host_rt_set_route_dir", None, [_vp, C.c_char_p]),
    ("uvrt_host_rt_set_default_route", None, [_vp, C.c_char_p]),
    ("uvrt_host_rt_set_device", None, [_vp, C.c_int]),
    ("uvrt_host_rt_set_auto_save", None, [_vp, C.c_int]),
    ("uvrt_host_rt_init", None, [_vp, _vp]),
    ("uvrt_host_rt_load_route", None, [_vp, C.c_char_p]),
    ("uvrt_host_rt_save_route", None, [_vp, C.c_char_p]),
    ("uvrt_host_rt_update_photons_per_light", None, [_vp]),
    ("uvrt_host_rt_reset_dosage_map", None, [_vp]),
    ("uvrt_host_rt_clear_buffers", None, [_vp, C.c_int]),
    ("uvrt_host_rt_compute_dosage_map", None, [_vp]),
    ("uvrt_host_rt_compute_single", None, [_vp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]),
    ("uvrt_host_rt_shade", None, [_vp]),
    ("uvrt_host_rt_add_lamp", None, [_vp]),
    ("uvrt_host_rt_calibrate", None, [_vp, C.c_float, C.c_float, C.c_float]),
    ("uvrt_host_rt_sync", None, [_vp]),
    ("uvrt_host_rt_read_dosage", None, [_vp, _vp, C.c_int, C.c_int]),
    ("uvrt_host_rt_ctx", _vp, [_vp]),
    ("uvrt_host_rt_set_shard", None, [_vp, C.c_int, C.c_int]),
    ("uvrt_host_rt_compute_batched", None, [_vp, C.c_int]),
    ("uvrt_host_rt_compute_batched_group", None, [C.POINTER(_vp), C.c_int, C.c_int]),
    ("uvrt_host_rt_set_ray_range", None, [_vp, C.c_int, C.c_int]),
    ("uvrt_host_rt_set_reduce_over_comm", None, [_vp, C.c_int]),
    ("uvrt_host_rt_lamp_count", C.c_int, [_vp]),
    ("uvrt_host_rt_get_lamp", None, [_vp, C.c_int, C.POINTER(C.c_float)]),
    ("uvrt_host_rt_set_lamps", None, [_vp, C.POINTER(C.c_float), C.c_int]),
    ("uvrt_host_rt_get", C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    ("uvrt_host_rt_set", C.c_int, [_vp, C.c_char_p, C.c_double]),
]

_FIELDS = {"lightLength", "lightHeight", "maxPhotonCount", "photonCount", "maxIterations",
           "currIterations", "lightIntensity", "minDosage", "minPower", "photonsPerLight", "compTime",
           "progress", "finishedComputation", "thresholdView", "startedComputation", "calibratedPower",
           "photonMapSize", "viewMode"}
_INT_FIELDS = {"maxPhotonCount", "photonCount", "maxIterations", "currIterations", "photonsPerLight",
               "photonMapSize", "viewMode"}
_BOOL_FIELDS = {"finishedComputation", "thresholdView", "startedComputation"}


def lib():
    global _LIB
    if _LIB is None:
        capi.lib()   # the device library first: fails loudly when the HIP extension is missing
        if not os.path.exists(LIB_PATH):
            raise capi.UvrtError("host library missing: %s (run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class Mesh:
    """Tmpl8::Mesh: GLB -> Tri[] -> floorHeight -> BVH (mesh.cpp:5-136, bvh.cpp)."""

    def __init__(self, glb_path=None, tris=None):
        L = lib()
        if glb_path is not None:
            self._h = L.uvrt_host_mesh_load(os.fsencode(glb_path))
            if not self._h:
                raise capi.UvrtError("cannot load %s" % glb_path)
        else:
            t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
            self._h = L.uvrt_host_mesh_from_tris(t.ctypes.data_as(_vp), t.shape[0])
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.uvrt_host_mesh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def triangleCount(self):
        return int(self._L.uvrt_host_mesh_tri_count(self._h))

    @property
    def floorHeight(self):
        return float(self._L.uvrt_host_mesh_floor_height(self._h))

    @property
    def nodesUsed(self):
        return int(self._L.uvrt_host_mesh_nodes_used(self._h))

    def _view(self, ptr, nbytes):
        return (C.c_char * nbytes).from_address(ptr)

    def tris(self):
        T = self.triangleCount
        return np.frombuffer(self._view(self._L.uvrt_host_mesh_tris(self._h), T * 64),
                             dtype=np.float32).reshape(T, 16).copy()

    def nodes(self):
        n = self.nodesUsed
        return np.frombuffer(self._view(self._L.uvrt_host_mesh_nodes(self._h), n * 32), dtype=NODE_DT).copy()

    def triIdx(self):
        T = self.triangleCount
        return np.frombuffer(self._view(self._L.uvrt_host_mesh_tri_idx(self._h), T * 4), dtype=np.uint32).copy()

    def rebuild_bvh(self):
        self._L.uvrt_host_mesh_rebuild_bvh(self._h)


class RayTracer:
    """Tmpl8::RayTracer over the HIP C ABI.  `glb` / `route_xml` are conveniences of the
    binding: the route file is loaded from its own directory under its own name."""

    def __init__(self, glb=None, route_xml=None, device=0, mesh=None, init=True):
        L = lib()
        self._L = L
        if not init:          # route / parameter handling only: no device context is created
            self.mesh = None
            self._h = L.uvrt_host_rt_new()
            L.uvrt_host_rt_set_auto_save(self._h, 0)
            self.ctx = None
            return
        self.mesh = mesh if mesh is not None else Mesh(glb)
        self._h = L.uvrt_host_rt_new()
        L.uvrt_host_rt_set_device(self._h, int(device))
        L.uvrt_host_rt_set_auto_save(self._h, 0)
        if route_xml is not None:
            d, name = os.path.split(os.path.abspath(route_xml))
            L.uvrt_host_rt_set_route_dir(self._h, os.fsencode(d + os.sep))
            L.uvrt_host_rt_set_default_route(self._h, os.fsencode(os.path.splitext(name)[0]))
        else:
            L.uvrt_host_rt_set_route_dir(self._h, b"/nonexistent/")
        L.uvrt_host_rt_init(self._h, self.mesh._h)
        self.ctx = _BorrowedCtx(L.uvrt_host_rt_ctx(self._h), self.mesh.triangleCount)

    def close(self):
        if getattr(self, "_h", None):
            self._L.uvrt_host_rt_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name):
        if name in _FIELDS:
            v = C.c_double()
            self._L.uvrt_host_rt_get(self._h, name.encode(), C.byref(v))
            if name in _BOOL_FIELDS:
                return bool(v.value)
            return int(v.value) if name in _INT_FIELDS else float(v.value)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _FIELDS:
            self._L.uvrt_host_rt_set(self._h, name.encode(), float(value))
            if name == "photonCount":
                self._L.uvrt_host_rt_update_photons_per_light(self._h)
        else:
            object.__setattr__(self, name, value)

    # reference method names
    def UpdatePhotonsPerLight(self): self._L.uvrt_host_rt_update_photons_per_light(self._h)
    def ComputeDosageMap(self): self._L.uvrt_host_rt_compute_dosage_map(self._h)
    def ComputeSingleLightDosageMap(self, lamp, photonsPerLight, triangleCount):
        self._L.uvrt_host_rt_compute_single(self._h, lamp[0], lamp[1], lamp[2], int(photonsPerLight),
                                            int(triangleCount))
    def Shade(self): self._L.uvrt_host_rt_shade(self._h)
    def ResetDosageMap(self): self._L.uvrt_host_rt_reset_dosage_map(self._h)
    def ClearBuffers(self, resetColor): self._L.uvrt_host_rt_clear_buffers(self._h, int(bool(resetColor)))
    def AddLamp(self): self._L.uvrt_host_rt_add_lamp(self._h)
    def CalibratePower(self, measurePower, measureHeight, measureDist):
        self._L.uvrt_host_rt_calibrate(self._h, measurePower, measureHeight, measureDist)
    def SaveRoute(self, name): self._L.uvrt_host_rt_save_route(self._h, os.fsencode(name))
    def LoadRoute(self, name): self._L.uvrt_host_rt_load_route(self._h, os.fsencode(name))
    def set_route_dir(self, d): self._L.uvrt_host_rt_set_route_dir(self._h, os.fsencode(d))

    # headless additions
    def Sync(self): self._L.uvrt_host_rt_sync(self._h)
    def set_shard(self, rank, world): self._L.uvrt_host_rt_set_shard(self._h, int(rank), int(world))
    def ComputeIterationsBatched(self, iterations): self._L.uvrt_host_rt_compute_batched(self._h, int(iterations))
    def SetRayRange(self, rank, world): self._L.uvrt_host_rt_set_ray_range(self._h, int(rank), int(world))
    def set_reduce_over_comm(self, on): self._L.uvrt_host_rt_set_reduce_over_comm(self._h, int(bool(on)))

    def lamps(self):
        out = []
        buf = (C.c_float * 3)()
        for i in range(self._L.uvrt_host_rt_lamp_count(self._h)):
            self._L.uvrt_host_rt_get_lamp(self._h, i, buf)
            out.append((float(buf[0]), float(buf[1]), float(buf[2])))
        return out

    def set_lamps(self, lamps):
        flat = (C.c_float * (3 * len(lamps)))(*[float(np.float32(v)) for l in lamps for v in l])
        self._L.uvrt_host_rt_set_lamps(self._h, flat, len(lamps))

    def read_dosage(self, first=0, count=None):
        T = self.mesh.triangleCount
        count = T - first if count is None else count
        out = np.empty(count, dtype=np.float32)
        self._L.uvrt_host_rt_read_dosage(self._h, out.ctypes.data_as(_vp), first, count)
        return out


def compute_iterations_batched_group(rts, iterations):
    """RayTracer::ComputeIterationsBatched over instances that share every launch by ray range (one process)."""
    arr = (C.c_void_p * len(rts))(*[rt._h for rt in rts])
    lib().uvrt_host_rt_compute_batched_group(arr, len(rts), int(iterations))


class _BorrowedCtx(capi.Ctx):
    """The uvrt_ctx owned by a C++ RayTracer, exposed with the capi.Ctx methods (read-backs,
    knobs, device pointers).  Never destroyed from Python."""

    def __init__(self, handle, T):
        self._L = capi.lib()
        self._h = C.c_void_p(handle)
        self.T = T

    def close(self):
        self._h = None

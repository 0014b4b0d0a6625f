"""ctypes/numpy front end of the CPU oracle (oracle/uvrt_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product.  Scene loading (GLB, route XML) is restated here in
numpy/stdlib; the arithmetic lives in the C file.  Reference citations are file:line relative
to the reference checkout.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RAY_DT = np.dtype([("dirx", "<f4"), ("diry", "<f4"), ("dirz", "<f4"),
                   ("origx", "<f4"), ("origy", "<f4"), ("origz", "<f4"),
                   ("dist", "<f4"), ("triID", "<u4")])                    # cl/tools.cl:8-14
NODE_DT = np.dtype([("minx", "<f4"), ("miny", "<f4"), ("minz", "<f4"), ("leftFirst", "<i4"),
                    ("maxx", "<f4"), ("maxy", "<f4"), ("maxz", "<f4"), ("triCount", "<i4")])
assert RAY_DT.itemsize == 32 and NODE_DT.itemsize == 32


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("aabb_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("hits", C.c_uint64), ("max_stack", C.c_uint32),
                ("visit_hist", C.c_void_p)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "visit_hist"}


def build():
    """Compile liboracle.so (and the gfx950 build of the reference .cl files when the
    reference checkout is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/cl"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    if os.path.exists(os.path.join(_HERE, "_ref", "ref_extend.co")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libref_gpu.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        L.orc_wang_hash.restype = C.c_uint32
        L.orc_wang_hash.argtypes = [C.c_uint32]
        L.orc_seed_of.restype = C.c_uint32
        L.orc_seed_of.argtypes = [C.c_int32, fp, C.c_uint32]
        L.orc_generate_one.restype = C.c_uint32
        L.orc_generate_one.argtypes = [C.c_void_p, C.c_int32, fp, C.c_float, C.c_uint32]
        L.orc_generate_fixed_seed.restype = C.c_uint32
        L.orc_generate_fixed_seed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, fp, C.c_float, C.c_uint32, C.c_int]
        L.orc_generate.restype = None
        L.orc_generate.argtypes = [C.c_void_p, C.c_int64, C.c_int64, fp, C.c_float,
                                   C.POINTER(C.c_uint32)]
        L.orc_extend.restype = None
        L.orc_extend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_accumulate.restype = None
        L.orc_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32]
        L.orc_reset.restype = None
        L.orc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_int32]
        L.orc_compute_dosage.restype = None
        L.orc_compute_dosage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_float, C.c_int32]
        L.orc_dosage_to_color.restype = None
        L.orc_dosage_to_color.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                          C.c_int32]
        L.orc_floor_height.restype = C.c_float
        L.orc_floor_height.argtypes = [C.c_void_p, C.c_int32]
        L.orc_set_flavour.restype = None
        L.orc_set_flavour.argtypes = [C.c_int]
        L.orc_set_rcp_table.restype = None
        L.orc_set_rcp_table.argtypes = [C.c_void_p]
        L.orc_have_rcp_table.restype = C.c_int
        L.orc_rcp_model.restype = C.c_float
        L.orc_rcp_model.argtypes = [C.c_float]
        L.orc_extend_steps.restype = None
        L.orc_extend_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_extend_visit_hist.restype = None
        L.orc_extend_visit_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bvh_build.restype = C.c_int32
        L.orc_bvh_build.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


# --------------------------------------------------------------------------- scene loading

def load_glb(path):
    """mesh.cpp:5-71: binary glTF, meshes[0].primitives[0] only, POSITION + u16/u32 indices,
    no node transforms, no byteStride.  Returns float32 [T,16] in the 64-byte Tri layout
    (mesh.h:6-13); pads and centroid are zero (the reference leaves them uninitialised)."""
    b = open(path, "rb").read()
    magic, _ver, _length = struct.unpack_from("<4sII", b, 0)
    if magic != b"glTF":
        raise ValueError("not a GLB file")
    clen, ctype = struct.unpack_from("<I4s", b, 12)
    if ctype != b"JSON":
        raise ValueError("first chunk is not JSON")
    doc = json.loads(b[20:20 + clen])
    off = 20 + clen
    blen, btype = struct.unpack_from("<I4s", b, off)
    if btype != b"BIN\x00":
        raise ValueError("second chunk is not BIN")
    binbuf = b[off + 8: off + 8 + blen]
    prim = doc["meshes"][0]["primitives"][0]
    pacc = doc["accessors"][prim["attributes"]["POSITION"]]
    iacc = doc["accessors"][prim["indices"]]
    pview = doc["bufferViews"][pacc["bufferView"]]
    iview = doc["bufferViews"][iacc["bufferView"]]
    poff = pview.get("byteOffset", 0) + pacc.get("byteOffset", 0)
    ioff = iview.get("byteOffset", 0) + iacc.get("byteOffset", 0)
    pos = np.frombuffer(binbuf, dtype="<f4", count=pacc["count"] * 3, offset=poff).reshape(-1, 3)
    if iacc["componentType"] == 5123:
        idx = np.frombuffer(binbuf, dtype="<u2", count=iacc["count"], offset=ioff)
    elif iacc["componentType"] == 5125:
        idx = np.frombuffer(binbuf, dtype="<u4", count=iacc["count"], offset=ioff)
    else:
        raise ValueError("indices must be u16 or u32 (mesh.cpp:44-51)")
    T = iacc["count"] // 3
    idx = idx[:T * 3].astype(np.int64).reshape(T, 3)
    tris = np.zeros((T, 16), dtype=np.float32)
    tris[:, 0:3] = pos[idx[:, 0]]
    tris[:, 4:7] = pos[idx[:, 1]]
    tris[:, 8:11] = pos[idx[:, 2]]
    return np.ascontiguousarray(tris)


def floor_height(tris):
    """mesh.cpp:100-136 over the duplicated vertex list (3 vertices per triangle)."""
    y = np.ascontiguousarray(tris[:, [1, 5, 9]].reshape(-1))
    return float(lib().orc_floor_height(_p(y), y.size))


def build_bvh(tris):
    """bvh.cpp:5-44.  Mutates tris[:,12:15] (centroids).  Returns (nodes[extent], triIdx)."""
    T = tris.shape[0]
    cap = 2 * T + 64
    nodes = np.zeros(cap, dtype=NODE_DT)
    triIdx = np.zeros(T, dtype=np.uint32)
    ext = lib().orc_bvh_build(_p(tris), T, _p(nodes), cap, _p(triIdx))
    if ext < 0:
        raise RuntimeError("orc_bvh_build failed")
    return nodes[:ext].copy(), triIdx


def load_route(path):
    """raytracer.cpp:261-300 (Dutch tags).  Missing tags keep the class defaults of
    raytracer.h:28-37."""
    r = {"photonCount": 1 << 25, "maxIterations": 10, "lightIntensity": 450.0,
         "minDosage": 100.0, "minPower": 1500.0, "lightLength": 1.0, "lightHeight": 0.8,
         "lamps": []}
    root = ET.parse(path).getroot()
    tag = {"aantal_fotonen": ("photonCount", int), "aantal_iteraties": ("maxIterations", int),
           "lamp_sterkte": ("lightIntensity", float), "minimale_dosis": ("minDosage", float),
           "minimale_bestralingssterkte": ("minPower", float),
           "lamp_lengte": ("lightLength", float), "lamp_hoogte": ("lightHeight", float)}
    for k, (name, conv) in tag.items():
        e = root.find(k)
        if e is not None and e.text is not None:
            r[name] = conv(e.text)
    route = root.find("route")
    if route is not None:
        i = 0
        while True:
            e = route.find("lamp_positie_%d" % i)
            if e is None:
                break
            r["lamps"].append((float(e.get("positie_x")), float(e.get("positie_y")),
                               float(e.get("duration"))))
            i += 1
    for k in ("lightIntensity", "minDosage", "minPower", "lightLength", "lightHeight"):
        r[k] = float(np.float32(r[k]))
    return r


# ------------------------------------------------------------------------------- kernels

def seed_of(tid, lp, SEED):
    return int(lib().orc_seed_of(int(tid), _f3(lp), int(SEED)))


def generate(first, n, lp, lightLength, SEED):
    """Returns (rays[n], SEED_k)."""
    rays = np.zeros(n, dtype=RAY_DT)
    s = C.c_uint32(int(SEED))
    lib().orc_generate(_p(rays), int(first), int(n), _f3(lp), float(np.float32(lightLength)),
                       C.byref(s))
    return rays, int(s.value)


def generate_fixed_seed(first, n, lp, lightLength, SEED, saturate=False):
    """Every work-item reads the same SEED (uvrt_oracle.h orc_generate_fixed_seed).  Returns
    (rays[n], final RNG state of gid 0 or 0)."""
    rays = np.zeros(n, dtype=RAY_DT)
    s0 = lib().orc_generate_fixed_seed(_p(rays), int(first), int(n), _f3(lp), float(np.float32(lightLength)),
                                       int(SEED), int(bool(saturate)))
    return rays, int(s0)


_RCP_TABLE = None


def set_flavour(flavour):
    """0 = canonical strict arithmetic (default), 1 = "ocl-amd", 2 = "shipped flags" (uvrt_oracle.h).  Flavour 2
    evaluates v_rcp_f32 through a table read from the GPU (refgpu_rcp_table): it exists only on a GPU box."""
    if int(flavour) == 2 and not lib().orc_have_rcp_table():
        set_rcp_table(refgpu_rcp_table())
    lib().orc_set_flavour(int(flavour))


def set_rcp_table(table):
    """2^23 uint32: bits of v_rcp_f32(1.m) per significand m (kept alive here: the C side keeps the pointer)"""
    global _RCP_TABLE
    table = np.ascontiguousarray(table, dtype=np.uint32)
    assert table.size == 1 << 23
    _RCP_TABLE = table
    lib().orc_set_rcp_table(_p(table))


def rcp_model(x):
    return float(lib().orc_rcp_model(float(np.float32(x))))


def extend(temp, tris, rays, nodes, triIdx, nthreads=0):
    st = Stats()
    lib().orc_extend(_p(temp), _p(tris), _p(rays), rays.size, _p(nodes), _p(triIdx),
                     C.byref(st), int(nthreads))
    return st.as_dict()


def extend_steps(tris, rays, nodes, triIdx):
    """node visits per ray (analysis helper)"""
    steps = np.zeros(rays.size, dtype=np.uint16)
    lib().orc_extend_steps(_p(tris), _p(rays), rays.size, _p(nodes), _p(triIdx), _p(steps))
    return steps


def extend_visit_hist(tris, rays, nodes, triIdx):
    """visits per node index (analysis helper)"""
    hist = np.zeros(nodes.size, dtype=np.uint32)
    lib().orc_extend_visit_hist(_p(tris), _p(rays), rays.size, _p(nodes), _p(triIdx), _p(hist))
    return hist


def accumulate(photonMap, maxPhotonMap, temp, timeStep):
    lib().orc_accumulate(_p(photonMap), _p(maxPhotonMap), _p(temp),
                         float(np.float32(timeStep)), temp.size)


def reset(photonMap, maxPhotonMap, temp, color=None):
    lib().orc_reset(_p(photonMap), _p(maxPhotonMap), _p(temp),
                    _p(color) if color is not None else None, 0 if color is None else 1,
                    temp.size)


def compute_dosage(pmap, tris, photonsPerLight, scaledPower):
    dose = np.zeros(pmap.size, dtype=np.float32)
    lib().orc_compute_dosage(_p(pmap), _p(dose), _p(tris), int(photonsPerLight),
                             float(np.float32(scaledPower)), pmap.size)
    return dose


def dosage_to_color(dose, minValue, thresholdView):
    color = np.zeros((dose.size, 9), dtype=np.float32)
    lib().orc_dosage_to_color(_p(dose), _p(color), float(np.float32(minValue)),
                              int(bool(thresholdView)), dose.size)
    return color


def algorithmic_bytes_per_ray(st):
    """B_extend = 32 + 8 + 32*(1+A) + (4+64)*K + 4*H with the reference record
    sizes (32-B ray, 32-B node, 64-B triangle, 4-B index)."""
    n = float(st["rays"])
    A = st["aabb_tests"] / n
    K = st["tri_tests"] / n
    H = st["hits"] / n
    return 32.0 + 8.0 + 32.0 * (1.0 + A) + 68.0 * K + 4.0 * H


class Scene:
    """What MyApp::Init hands to RayTracer::Init (myapp.cpp:36-39): Tri[], floorHeight, BVH."""

    def __init__(self, glb_path):
        self.tris = load_glb(glb_path)
        self.T = self.tris.shape[0]
        self.floorHeight = floor_height(self.tris)
        self.nodes, self.triIdx = build_bvh(self.tris)


class Computation:
    """Host sequence of raytracer.cpp:66-143 + myapp.cpp:156-175 on the oracle kernels."""

    def __init__(self, scene, lamps, photonCount, lightHeight, lightLength, lightIntensity,
                 nthreads=0):
        self.s = scene
        self.lamps = list(lamps)
        self.photonCount = int(photonCount)
        self.photonsPerLight = (self.photonCount // len(self.lamps)) & ~1   # raytracer.cpp:63
        self.lightHeight = np.float32(lightHeight)
        self.lightLength = np.float32(lightLength)
        self.lightIntensity = np.float32(lightIntensity)
        self.nthreads = nthreads
        T = scene.T
        self.photonMap = np.zeros(T, dtype=np.float64)
        self.maxPhotonMap = np.zeros(T, dtype=np.float64)
        self.temp = np.zeros(T, dtype=np.int32)
        self.SEED = 0                      # fresh Init: program-scope SEED is zero
        self.photonMapSize = 0
        self.stats = []
        self.last_rays = None

    def lamp_world_pos(self, lamp):
        # raytracer.cpp:77 -- f32 add
        y = np.float32(np.float32(self.s.floorHeight) + self.lightHeight)
        return (np.float32(lamp[0]), y, np.float32(lamp[1]))

    def reset(self):                                       # raytracer.cpp:122-143
        self.photonMapSize = 0
        reset(self.photonMap, self.maxPhotonMap, self.temp)

    def single_light(self, lamp, first=0, n=None):         # raytracer.cpp:75-88
        n = self.photonsPerLight if n is None else n
        lp = self.lamp_world_pos(lamp)
        rays, self.SEED = generate(first, n, lp, self.lightLength, self.SEED)
        st = extend(self.temp, self.s.tris, rays, self.s.nodes, self.s.triIdx, self.nthreads)
        self.stats.append(st)
        self.last_rays = rays
        accumulate(self.photonMap, self.maxPhotonMap, self.temp, lamp[2])
        self.photonMapSize += n

    def iteration(self):                                   # raytracer.cpp:66-72
        for lamp in self.lamps:
            self.single_light(lamp)

    def dose(self):                                        # raytracer.cpp:106-118
        return compute_dosage(self.photonMap, self.s.tris,
                              self.photonMapSize // len(self.lamps),
                              np.float32(self.lightIntensity * np.float32(0.1)))

    def max_power(self):                                   # raytracer.cpp:96-104
        return compute_dosage(self.maxPhotonMap, self.s.tris, self.photonsPerLight,
                              np.float32(self.lightIntensity * np.float32(100.0)))


# ------------------------------------------------ the reference's own kernels on the GPU

_REFGPU = None


def refgpu():
    """oracle/_ref/ref_*.co (the reference's cl/*.cl compiled unmodified for gfx950) behind
    oracle/ref_gpu.cpp.  Returns None when the code objects were not built (they are built by
    `make -C oracle ref` where /root/reference exists and travel to the GPU box prebuilt)."""
    global _REFGPU
    if _REFGPU is None:
        d = os.path.join(_HERE, "_ref")
        so = os.path.join(_HERE, "libref_gpu.so")
        if not os.path.exists(os.path.join(d, "ref_extend.co")):
            return None
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libref_gpu.so"])
        L = C.CDLL(so)
        L.refgpu_last_error.restype = C.c_char_p
        L.refgpu_load.argtypes = [C.c_char_p]
        L.refgpu_extend.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.refgpu_generate.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_float,
                                      C.POINTER(C.c_double)]
        L.refgpu_extend_shipped.argtypes = L.refgpu_extend.argtypes
        L.refgpu_generate_shipped.argtypes = L.refgpu_generate.argtypes
        L.refgpu_rcp_table.argtypes = [C.c_void_p]
        L.refgpu_rcp_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
        L.refgpu_shade.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
        L.refgpu_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.refgpu_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32]
        L.refgpu_compute_dosage.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
        if L.refgpu_load(os.fsencode(d)) != 0:
            raise RuntimeError("refgpu_load: " + L.refgpu_last_error().decode())
        _REFGPU = L
    return _REFGPU


def refgpu_extend(rays, tris, nodes, triIdx, reps=1, shipped=False):
    """extend.cl:render of the reference on the GPU.  rays (RAY_DT, len % 256 == 0) is updated in
    place; returns (counts, kernel_ms).  shipped = the build with the reference's own flags (ref_extend_fast.co)."""
    L = refgpu()
    assert rays.size % 256 == 0
    counts = np.zeros(tris.shape[0], dtype=np.int32)
    ms = C.c_double()
    fn = L.refgpu_extend_shipped if shipped else L.refgpu_extend
    if fn(_p(rays), rays.size, _p(tris), tris.shape[0], _p(nodes), nodes.shape[0], _p(triIdx),
          _p(counts), C.byref(ms), int(reps)) != 0:
        raise RuntimeError("refgpu_extend: " + L.refgpu_last_error().decode())
    return counts, float(ms.value)


def refgpu_have_shipped():
    L = refgpu()
    return L is not None and bool(L.refgpu_have_shipped())


def refgpu_rcp_table():
    """bits of v_rcp_f32(1.m) for all 2^23 significands, read from the GPU (oracle/rcp_probe.hip)"""
    L = refgpu()
    if L is None:
        raise RuntimeError("flavour 2 needs the GPU-side probe (oracle/libref_gpu.so) and a gfx950")
    table = np.empty(1 << 23, dtype=np.uint32)
    if L.refgpu_rcp_table(_p(table)) != 0:
        raise RuntimeError("refgpu_rcp_table failed")
    return table


def refgpu_rcp_check(table, cap=64):
    """oracle/rcp_model.h (built from `table`) against v_rcp_f32 on all 2^32 inputs: (mismatches, first few (x, hw, model))"""
    L = refgpu()
    n = C.c_uint64()
    bad = np.zeros((cap, 3), dtype=np.uint32)
    table = np.ascontiguousarray(table, dtype=np.uint32)
    if L.refgpu_rcp_check(_p(table), C.byref(n), _p(bad), cap) != 0:
        raise RuntimeError("refgpu_rcp_check failed")
    return int(n.value), bad[:min(cap, int(n.value))]


def refgpu_reload():
    """Fresh modules: program-scope SEED = 0 again (generate.cl:6), as after RayTracer::Init."""
    if refgpu().refgpu_reload() != 0:
        raise RuntimeError("refgpu_reload: " + refgpu().refgpu_last_error().decode())


def refgpu_generate(n, lp, lightLength, shipped=False):
    """generate.cl:render of the reference on the GPU over n work-items (n % 256 == 0), with whatever
    SEED the loaded module holds.  Returns (rays, kernel_ms).  shipped = the build with the reference's own flags."""
    L = refgpu()
    assert n % 256 == 0
    rays = np.zeros(n, dtype=RAY_DT)
    ms = C.c_double()
    fn = L.refgpu_generate_shipped if shipped else L.refgpu_generate
    if fn(_p(rays), int(n), _f3(lp), float(np.float32(lightLength)), C.byref(ms)) != 0:
        raise RuntimeError("refgpu_generate: " + L.refgpu_last_error().decode())
    return rays, float(ms.value)


def refgpu_reset(photonMap, maxPhotonMap, temp, color, resetColor):
    """reset.cl:render of the reference on the GPU; the arrays are updated in place."""
    L = refgpu()
    assert color.dtype == np.float32 and color.size == 9 * temp.size
    if L.refgpu_reset(_p(photonMap), _p(maxPhotonMap), _p(temp), _p(color), temp.size, int(bool(resetColor))) != 0:
        raise RuntimeError("refgpu_reset: " + L.refgpu_last_error().decode())


def refgpu_accumulate(photonMap, maxPhotonMap, temp, timeStep):
    """accumulate.cl:render of the reference on the GPU; the arrays are updated in place."""
    L = refgpu()
    if L.refgpu_accumulate(_p(photonMap), _p(maxPhotonMap), _p(temp), float(np.float32(timeStep)), temp.size) != 0:
        raise RuntimeError("refgpu_accumulate: " + L.refgpu_last_error().decode())


def refgpu_compute_dosage(pmap, tris, photonsPerLight, scaledPower):
    """shade.cl:computeDosage of the reference on the GPU."""
    L = refgpu()
    dose = np.zeros(pmap.size, dtype=np.float32)
    if L.refgpu_compute_dosage(_p(pmap), _p(tris), pmap.size, int(photonsPerLight),
                               float(np.float32(scaledPower)), _p(dose)) != 0:
        raise RuntimeError("refgpu_compute_dosage: " + L.refgpu_last_error().decode())
    return dose


def refgpu_generate_ms(n, lp, lightLength):
    L = refgpu()
    ms = C.c_double()
    if L.refgpu_generate(None, int(n), _f3(lp), float(np.float32(lightLength)), C.byref(ms)) != 0:
        raise RuntimeError("refgpu_generate: " + L.refgpu_last_error().decode())
    return float(ms.value)


def refgpu_shade(photonMap, maxPhotonMap, temp, timeStep, tris, photonsPerLight, scaledPower, minValue,
                 thresholdView):
    """accumulate.cl, shade.cl:computeDosage, shade.cl:dosageToColor of the reference on the GPU.
    Arrays are padded to a multiple of 256 triangles internally.  Returns (dose, color)."""
    L = refgpu()
    T = temp.size
    Tp = (T + 255) // 256 * 256
    pm = np.zeros(Tp); pm[:T] = photonMap
    mm = np.zeros(Tp); mm[:T] = maxPhotonMap
    tc = np.zeros(Tp, dtype=np.int32); tc[:T] = temp
    tt = np.zeros((Tp, 16), dtype=np.float32); tt[:T] = tris
    tt[T:, 0] = 1.0; tt[T:, 5] = 1.0          # unit right triangles in the padding (non-zero area)
    dose = np.zeros(Tp, dtype=np.float32)
    col = np.zeros((Tp, 9), dtype=np.float32)
    if L.refgpu_shade(_p(pm), _p(mm), _p(tc), float(np.float32(timeStep)), _p(tt), Tp, int(photonsPerLight),
                      float(np.float32(scaledPower)), float(np.float32(minValue)), int(bool(thresholdView)),
                      _p(dose), _p(col)) != 0:
        raise RuntimeError("refgpu_shade: " + L.refgpu_last_error().decode())
    photonMap[:] = pm[:T]; maxPhotonMap[:] = mm[:T]; temp[:] = tc[:T]
    return dose[:T].copy(), col[:T].copy()

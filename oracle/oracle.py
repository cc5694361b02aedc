"""ctypes bindings for the CPU oracle (oracle/librt_oracle.so) and, when present,
the unmodified-reference validation libraries (oracle/_ref/libref*.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package (raytracer_amd) never imports
this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref.so")
REF_O0_PATH = os.path.join(_HERE, "_ref", "libref_O0.so")

SPHERE, PLANE, TRIANGLE = 0, 1, 2


class Material(C.Structure):
    _fields_ = [("reflect", C.c_float), ("diffuse", C.c_float), ("spec", C.c_float),
                ("color", C.c_float * 3)]


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int), ("is_light", C.c_int), ("mat", Material),
                ("a", C.c_float * 3), ("b", C.c_float * 3), ("c", C.c_float * 3),
                ("n", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("pitch", C.c_float), ("off_x", C.c_float),
                ("off_y", C.c_float), ("plane_z", C.c_float)]


class Counts(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("reflect", C.c_uint64),
                ("prim_tests", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        f3 = C.POINTER(C.c_float)
        L.rto_make_sphere.argtypes = [C.POINTER(Prim), f3, C.c_float]
        L.rto_make_plane.argtypes = [C.POINTER(Prim), f3, C.c_float]
        L.rto_make_triangle.argtypes = [C.POINTER(Prim), f3, f3, f3]
        L.rto_make_mesh.argtypes = [C.POINTER(Prim), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Material), C.c_void_p]
        L.rto_default_camera.argtypes = [C.POINTER(Camera)]
        L.rto_scene_create.restype = C.c_void_p
        L.rto_scene_create.argtypes = [C.POINTER(Prim), C.c_int]
        L.rto_scene_destroy.argtypes = [C.c_void_p]
        L.rto_scene_build_bvh.argtypes = [C.c_void_p]
        L.rto_scene_set_corner_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rto_render.argtypes = [C.c_void_p, C.POINTER(Camera)] + [C.c_int] * 9 + [C.c_void_p] * 5
        L.rto_raytrace.argtypes = [C.c_void_p, f3, f3, C.c_int, C.c_int, C.c_int, f3,
                                   C.POINTER(C.c_int32), f3, C.c_void_p]
        for n in ("rto_tri_raycast", "rto_sphere_raycast", "rto_plane_raycast"):
            getattr(L, n).argtypes = [C.POINTER(Prim), f3, f3, f3]
        L.rto_quantise.restype = C.c_uint8
        L.rto_quantise.argtypes = [C.c_float]
        L.rto_load_obj.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int),
                                   C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
        L.rto_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def material(reflect=0.0, diffuse=1.0, spec=1.0, color=(1.0, 1.0, 1.0)):
    m = Material()
    m.reflect, m.diffuse, m.spec = reflect, diffuse, spec
    m.color[:] = list(color)
    return m


def sphere(pos, radius, mat=None, is_light=False):
    p = Prim()
    lib().rto_make_sphere(C.byref(p), _f3(pos), radius)
    if mat is not None:
        p.mat = mat
    p.is_light = int(is_light)
    return p


def plane(normal, offset, mat=None):
    p = Prim()
    lib().rto_make_plane(C.byref(p), _f3(normal), offset)
    if mat is not None:
        p.mat = mat
    return p


def triangle(v1, v2, v3, mat=None):
    p = Prim()
    lib().rto_make_triangle(C.byref(p), _f3(v1), _f3(v2), _f3(v3))
    if mat is not None:
        p.mat = mat
    return p


def camera(origin=(0.0, 0.0, -5.0), pitch=0.01, off_x=-4.0, off_y=-4.0, plane_z=0.0):
    c = Camera()
    c.origin[:] = [float(x) for x in origin]
    c.pitch, c.off_x, c.off_y, c.plane_z = pitch, off_x, off_y, plane_z
    return c


def triangles_from_mesh(verts, indices, mats=None, tri_mat=None):
    """The flatten of main.cpp:228-231: one Triangle per index triple.  mats: list of
    Material (tri_mat indexes it; None = mats[0] everywhere; mats None = default)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
    n = idx.size // 3
    out = (Prim * max(n, 1))()
    marr = (Material * len(mats))(*mats) if mats else None
    tm = np.ascontiguousarray(tri_mat, dtype=np.int32) if tri_mat is not None else None
    lib().rto_make_mesh(out, verts.ctypes.data, idx.ctypes.data, n, marr, tm.ctypes.data if tm is not None else None)
    return out, n


class Scene:
    def __init__(self, prims, bvh=False, mesh=None):
        """prims: list of Prim in scene order; mesh: optional (Prim array, count) from
        triangles_from_mesh(), appended after them (as LoadModel does, main.cpp:257)."""
        n0 = len(prims)
        n1 = mesh[1] if mesh is not None else 0
        self.n = n0 + n1
        arr = (Prim * max(self.n, 1))()
        for k, p in enumerate(prims):
            arr[k] = p
        if n1:
            C.memmove(C.addressof(arr) + n0 * C.sizeof(Prim), mesh[0], n1 * C.sizeof(Prim))
        self._h = lib().rto_scene_create(arr, self.n)
        self.has_bvh = False
        if bvh:
            self.build_bvh()

    def build_bvh(self):
        if lib().rto_scene_build_bvh(self._h) != 0:
            raise RuntimeError("rto_scene_build_bvh failed")
        self.has_bvh = True

    def set_corner_normals(self, vn):
        """Opt-in shading normals, (n_tris, 3, 3) in scene order; None = the reference's geometric normals."""
        if vn is None:
            lib().rto_scene_set_corner_normals(self._h, None, 0)
            return
        a = np.ascontiguousarray(vn, dtype=np.float32).reshape(-1, 9)
        if lib().rto_scene_set_corner_normals(self._h, a.ctypes.data, a.shape[0]) != 0:
            raise RuntimeError("rto_scene_set_corner_normals: count mismatch")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rto_scene_destroy(self._h)
            self._h = None

    def render(self, W, H, y0=0, y1=None, max_depth=3, cam=None, use_bvh=False, threads=1,
               x_step=1, y_step=1, want=("rgb8", "rgb32", "hit_id", "hit_t")):
        y1 = H if y1 is None else y1
        cam = cam or camera()
        rows = y1 - y0
        out = {}
        rgb8 = np.zeros((rows, W, 3), np.uint8) if "rgb8" in want else None
        rgb32 = np.zeros((rows, W, 3), np.float32) if "rgb32" in want else None
        hid = np.full((rows, W), -2, np.int32) if "hit_id" in want else None
        ht = np.zeros((rows, W), np.float32) if "hit_t" in want else None
        cn = Counts()
        ptr = lambda a: a.ctypes.data if a is not None else None
        rc = lib().rto_render(self._h, C.byref(cam), W, H, y0, y1, max_depth, int(use_bvh),
                              threads, x_step, y_step, ptr(rgb8), ptr(rgb32), ptr(hid), ptr(ht),
                              C.addressof(cn))
        if rc != 0:
            raise RuntimeError(f"rto_render rc={rc}")
        out.update(rgb8=rgb8, rgb32=rgb32, hit_id=hid, hit_t=ht,
                   counts={k: getattr(cn, k) for k, _ in Counts._fields_})
        return out


def load_obj(path):
    v = C.POINTER(C.c_float)()
    i = C.POINTER(C.c_int)()
    nv, ni = C.c_int(), C.c_int()
    rc = lib().rto_load_obj(os.fsencode(path), C.byref(v), C.byref(nv), C.byref(i), C.byref(ni))
    verts = np.ctypeslib.as_array(v, (nv.value * 3,)).copy().reshape(-1, 3) if nv.value else np.zeros((0, 3), np.float32)
    inds = np.ctypeslib.as_array(i, (ni.value,)).copy() if ni.value else np.zeros((0,), np.int32)
    lib().rto_free(v)
    lib().rto_free(i)
    return rc, verts.astype(np.float32), inds.astype(np.int32)


def quantise(x):
    return lib().rto_quantise(float(x))


# ------------------------------------------------------------------ unmodified reference (oracle/_ref)
class Reference:
    """The reference's own object code (built by oracle/build_ref.sh)."""

    def __init__(self, o0=False):
        path = REF_O0_PATH if o0 else REF_PATH
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        f3 = C.POINTER(C.c_float)
        L.ref_stock_frame.argtypes = [C.c_void_p, C.c_int]
        L.ref_add_sphere.argtypes = [f3, C.c_float, C.POINTER(C.c_float), C.c_int]
        L.ref_add_plane.argtypes = [f3, C.c_float, C.POINTER(C.c_float)]
        L.ref_add_triangle.argtypes = [f3, f3, f3, C.POINTER(C.c_float)]
        L.ref_load_model.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
        L.ref_parse_obj.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_get_triangle.argtypes = [C.c_int, C.POINTER(C.c_float)]
        L.ref_raytrace.argtypes = [f3, f3, C.c_int, f3]
        L.ref_render.argtypes = [f3, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_int] * 7 + [C.c_void_p] * 4
        self.L = L

    @staticmethod
    def available():
        return os.path.exists(REF_PATH) and os.path.exists(REF_O0_PATH)

    @staticmethod
    def _mat(m):
        return (C.c_float * 6)(m.reflect, m.diffuse, m.spec, m.color[0], m.color[1], m.color[2])

    def stock_frame(self):
        buf = np.zeros(800 * 600 * 3, np.uint8)
        rc = self.L.ref_stock_frame(buf.ctypes.data, buf.size)
        if rc < 0:
            raise RuntimeError("ref_stock_frame failed")
        return buf.reshape(rc & 0xFFFF, rc >> 16, 3)

    def clear(self):
        self.L.ref_scene_clear()

    def add_prims(self, prims):
        """Mirror a list of oracle Prim records into the reference's `scene` vector.
        Triangles are re-created from vertices v1, v1+e1, v1+e2 only when those sums are
        exact; callers that need bit-faithful triangles use add_triangle/load_model."""
        for p in prims:
            if p.kind == SPHERE:
                self.L.ref_add_sphere(_f3(p.a), p.b[0], self._mat(p.mat), p.is_light)
            elif p.kind == PLANE:
                self.L.ref_add_plane(_f3(p.a), p.b[0], self._mat(p.mat))
            else:
                raise ValueError("use add_triangle for triangles")

    def add_triangle(self, v1, v2, v3, mat):
        return self.L.ref_add_triangle(_f3(v1), _f3(v2), _f3(v3), self._mat(mat))

    def load_model(self, path, mat=None):
        return self.L.ref_load_model(os.fsencode(path), self._mat(mat) if mat is not None else None)

    def parse_obj(self, path, cap=1 << 22):
        v = np.zeros(cap, np.float32)
        i = np.zeros(cap, np.int32)
        nv, ni = C.c_int(), C.c_int()
        rc = self.L.ref_parse_obj(os.fsencode(path), v.ctypes.data, cap, i.ctypes.data, cap,
                                  C.byref(nv), C.byref(ni))
        if rc != 0:
            raise RuntimeError("ref_parse_obj: capacity")
        return v[: nv.value * 3].reshape(-1, 3).copy(), i[: ni.value].copy()

    def get_triangle(self, pid):
        out = (C.c_float * 12)()
        if self.L.ref_get_triangle(pid, out) != 0:
            raise ValueError("not a triangle")
        return np.array(out[:], np.float32).reshape(4, 3)

    def render(self, W, H, y0=0, y1=None, bounces=3, cam=None, x_step=1, y_step=1):
        y1 = H if y1 is None else y1
        cam = cam or camera()
        rows = y1 - y0
        rgb8 = np.zeros((rows, W, 3), np.uint8)
        rgb32 = np.zeros((rows, W, 3), np.float32)
        hid = np.full((rows, W), -2, np.int32)
        ht = np.zeros((rows, W), np.float32)
        self.L.ref_render(_f3(cam.origin), cam.pitch, cam.off_x, cam.off_y, cam.plane_z, W, H, y0, y1,
                          bounces, x_step, y_step, rgb8.ctypes.data, rgb32.ctypes.data,
                          hid.ctypes.data, ht.ctypes.data)
        return dict(rgb8=rgb8, rgb32=rgb32, hit_id=hid, hit_t=ht)


def stock_prims():
    """The scene of main.cpp:240-255."""
    return [
        sphere((0.0, 0.0, 0.0), 2.5, material(reflect=1.0, diffuse=0.0)),
        sphere((2.0, 5.0, 1.0), 0.1, material(color=(0.7, 0.7, 0.9)), is_light=True),
        sphere((-2.0, 5.0, -3.0), 0.1, material(color=(0.9, 0.9, 0.4)), is_light=True),
        plane((0.0, 1.0, 0.0), -4.0),
    ]


def scene_from_workload(wl, bvh=True):
    """Mirror a raytracer_amd.workloads.Workload (duck-typed: analytic records with
    kind/is_light/mat/p/s, a mesh and a material table) into an oracle Scene, in the same
    scene order: analytic primitives first, then one triangle per index triple."""
    def m(x):
        return material(x.reflect, x.diffuse, x.spec, tuple(x.color))
    prims = []
    for a in wl.analytic:
        if a.kind == SPHERE:
            prims.append(sphere(tuple(a.p), a.s, m(a.mat), bool(a.is_light)))
        else:
            prims.append(plane(tuple(a.p), a.s, m(a.mat)))
    mesh = triangles_from_mesh(wl.verts, wl.indices, [m(x) for x in wl.mats], wl.tri_material)
    return Scene(prims, bvh=bvh, mesh=mesh)


def camera_from(cam):
    return camera(tuple(cam.origin), cam.pitch, cam.off_x, cam.off_y, cam.plane_z)

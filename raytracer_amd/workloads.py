"""The benchmark configurations, as data the HIP path (and, separately, the checker) can load.

Each workload is the reference's stock lights (Raytracer/main.cpp:245-253) plus geometry:

  cube         cube.obj geometry + floor plane, mirror sphere removed so the cube is visible
  sponza       stand-in for the absent sponza.obj: the procedural atrium (~262k triangles)
               with a mirror ball on the courtyard floor
  soup         4x4 instanced atrium (~4.2M triangles)

Mesh triangles get reflect = 0.5 (the reference gives meshes reflect = 0, main.cpp:34 + 230,
so bounces on meshes need this declared material extension).
"""
import ctypes as C
import os

import numpy as np

from . import Camera, Material, Plane, Sphere, camera_for_width, scenes

ATRIUM_DETAIL = 0.905   # ~262k triangles

_ATRIUM_MATS = [
    (0.5, 1.0, 1.0, (0.85, 0.82, 0.75)),   # stone
    (0.5, 1.0, 1.0, (0.7, 0.7, 0.72)),     # floor
    (0.5, 1.0, 1.0, (0.8, 0.15, 0.12)),    # red cloth
    (0.5, 1.0, 1.0, (0.15, 0.25, 0.8)),    # blue cloth
    (0.5, 0.6, 1.0, (0.9, 0.7, 0.3)),      # brass
]


def _lights():
    return [Sphere((2.0, 5.0, 1.0), 0.1, Material(color=(0.7, 0.7, 0.9)), is_light=True),
            Sphere((-2.0, 5.0, -3.0), 0.1, Material(color=(0.9, 0.9, 0.4)), is_light=True)]


def _mirror():
    """A mirror ball standing on the courtyard floor (material of main.cpp:240-242)."""
    return Sphere((0.0, -3.0, 6.0), 1.0, Material(reflect=1.0, diffuse=0.0))


def _courtyard_camera(W):
    """Eye in front of the colonnade at standing height, image plane at z = 0 as in
    main.cpp:267-270; pitch 8/W keeps the reference's horizontal field of view."""
    pitch = float(np.float32(8.0) / np.float32(W))
    return Camera(origin=(0.0, -1.5, -6.0), pitch=pitch, off_x=-4.0, off_y=-3.4, plane_z=0.0)


class Workload:
    def __init__(self, name, analytic, verts, indices, mats, tri_material, W, H, max_depth, camera):
        self.name, self.analytic = name, analytic
        self.verts, self.indices = verts, indices
        self.mats, self.tri_material = mats, tri_material
        self.W, self.H, self.max_depth, self.camera = W, H, max_depth, camera

    @property
    def n_tris(self):
        return len(self.indices) // 3

    def load(self, ctx):
        ctx.set_analytic(self.analytic)
        ctx.set_mesh(self.verts, self.indices, self.mats[0])
        if len(self.mats) > 1:
            ctx.set_triangle_materials(self.mats, self.tri_material)
        ctx.set_camera(self.camera)
        return ctx


def cube(W=1920, H=1080, max_depth=4, reflect=0.5):
    v, i = scenes.cube_mesh()
    return Workload("cube.obj %dx%d depth %d" % (W, H, max_depth), _lights() + [Plane((0.0, 1.0, 0.0), -4.0)], v, i,
                    [Material(reflect=reflect)], np.zeros(len(i) // 3, np.int32), W, H, max_depth, camera_for_width(W))


def load_obj(path, relaxed=True):
    """An OBJ file through the host's own loader (include/rt_objhost.h over raytracer_amd/host/objloader.cpp,
    the restatement of Raytracer/objloader.cpp:18-105).  Returns verts (n,3) f32, indices (3m) i32,
    tri_material (m) i32, materials (k,6) f32 = (reflect, diffuse, spec, r, g, b), dropped face count."""
    lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libobjhost.so")
    if not os.path.exists(lib_path):
        raise RuntimeError("%s not built: run `python raytracer_amd/build.py`" % lib_path)
    L = C.CDLL(lib_path)
    L.rt_obj_load.restype = C.c_void_p
    L.rt_obj_load.argtypes = [C.c_char_p, C.c_int]
    L.rt_obj_free.argtypes = [C.c_void_p]
    L.rt_obj_counts.argtypes = [C.c_void_p, C.c_void_p]
    L.rt_obj_copy.argtypes = [C.c_void_p] * 5
    h = L.rt_obj_load(os.fsencode(path), 1 if relaxed else 0)
    if not h:
        raise MemoryError("rt_obj_load")
    try:
        cnt = np.zeros(4, np.int32)
        L.rt_obj_counts(h, cnt.ctypes.data)
        nv, ni, nm, dropped = (int(x) for x in cnt)
        v = np.zeros((nv, 3), np.float32); i = np.zeros(ni, np.int32)
        tm = np.zeros(ni // 3, np.int32); mats = np.zeros((nm, 6), np.float32)
        L.rt_obj_copy(h, v.ctypes.data, i.ctypes.data, tm.ctypes.data, mats.ctypes.data)
    finally:
        L.rt_obj_free(h)
    return v, i, tm, mats, dropped


def fit_to_courtyard(verts):
    """Place an arbitrary mesh where the courtyard camera and the reference lights (main.cpp:245-253) expect the
    atrium: the longer horizontal axis becomes z, the mesh is scaled uniformly so that axis spans 22 units
    (z in [-4, 18]), centred in x, floor at y = -4.  Rotation by a quarter turn about y keeps the handedness."""
    v = np.asarray(verts, np.float64).copy()
    lo, hi = v.min(0), v.max(0)
    if (hi[0] - lo[0]) > (hi[2] - lo[2]):
        v = np.stack([-v[:, 2], v[:, 1], v[:, 0]], 1)
        lo, hi = v.min(0), v.max(0)
    scale = 22.0 / max(hi[2] - lo[2], 1e-30)
    v = (v - np.array([0.5 * (lo[0] + hi[0]), lo[1], lo[2]])) * scale + np.array([0.0, -4.0, -4.0])
    return v.astype(np.float32)


def sponza_obj(path, W=1920, H=1080, max_depth=4, reflect=0.5):
    """The mesh of main.cpp:257 (`LoadModel(".../sponza.obj")`) from a supplied file, fitted to the courtyard
    camera.  The reference gives mesh triangles reflect = 0 (main.cpp:34 + 230), so bounces need a declared
    material extension: materials of a .mtl that set a reflection coefficient keep it, every other material
    gets `reflect` (the stand-in's 0.5)."""
    v, i, tm, mats, dropped = load_obj(path, relaxed=True)
    if len(i) < 3:
        raise RuntimeError("%s holds no triangles" % path)
    ms = [Material(float(m[0]) if m[0] > 0 else reflect, float(m[1]), float(m[2]), tuple(float(c) for c in m[3:6])) for m in mats]
    name = "%s(%d tris) %dx%d depth %d" % (os.path.basename(path), len(i) // 3, W, H, max_depth)
    return Workload(name, [_mirror()] + _lights(), fit_to_courtyard(v), i, ms, tm, W, H, max_depth, _courtyard_camera(W))


def sponza(W=1920, H=1080, max_depth=4, detail=ATRIUM_DETAIL, obj=None):
    """sponza.obj when a file is supplied (argument, else the RT_SPONZA_OBJ environment variable) and exists;
    otherwise the procedural stand-in: the asset is not part of the reference tree."""
    obj = obj or os.environ.get("RT_SPONZA_OBJ", "")
    if obj and os.path.exists(obj):
        return sponza_obj(obj, W, H, max_depth)
    v, i, m = scenes.atrium_mesh(detail=detail)
    mats = [Material(*t) for t in _ATRIUM_MATS]
    return Workload("sponza-standin(atrium %d tris) %dx%d depth %d" % (len(i) // 3, W, H, max_depth),
                    [_mirror()] + _lights(), v, i, mats, m, W, H, max_depth, _courtyard_camera(W))


def atrium2(W=1920, H=1080, max_depth=4):
    """A second stand-in with the irregularities of the real asset (scenes.atrium2_mesh): two-triangle walls
    next to dense ornaments, long thin slivers, overlapping cloth."""
    v, i, m = scenes.atrium2_mesh()
    mats = [Material(*t) for t in _ATRIUM_MATS]
    return Workload("sponza-standin-v2(atrium2 %d tris) %dx%d depth %d" % (len(i) // 3, W, H, max_depth),
                    [_mirror()] + _lights(), v, i, mats, m, W, H, max_depth, _courtyard_camera(W))


def soup(W=7680, H=4320, max_depth=8, detail=ATRIUM_DETAIL, nx=4, nz=4):
    v, i, m = scenes.atrium_mesh(detail=detail)
    v, i, m = scenes.instance_grid(v, i, m, nx, nz, 15.0, 23.0)
    v = v - np.array([0.5 * (nx - 1) * 15.0, 0.0, 0.0], np.float32)
    mats = [Material(*t) for t in _ATRIUM_MATS]
    return Workload("soup(%d tris) %dx%d depth %d" % (len(i) // 3, W, H, max_depth),
                    [_mirror()] + _lights(), v.astype(np.float32), i, mats, m, W, H, max_depth, _courtyard_camera(W))


def by_name(name, **kw):
    return {"cube": cube, "sponza": sponza, "soup": soup, "atrium2": atrium2}[name](**kw)

"""Shared scene descriptions for the tests: one neutral description that is handed both
to the CPU oracle (oracle/) and to the HIP path (raytracer_amd), so both trace the very
same primitives in the very same scene order."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import raytracer_amd as rt                      # noqa: E402
from raytracer_amd import scenes                # noqa: E402
from oracle import oracle as O                  # noqa: E402  (tests are allowed to use the checker)

GOLDEN = os.path.join(ROOT, "tests", "golden")


class SceneSpec:
    """analytic: list of dicts {kind, p, s, mat=(reflect, diffuse, spec, (r,g,b)), is_light};
    mesh: verts (n,3) + indices (3m); mesh_mat: one material tuple, or (mats, tri_material)."""

    def __init__(self, analytic, verts=None, indices=None, mesh_mat=None, mats=None, tri_material=None):
        self.analytic = analytic
        self.verts = None if verts is None else np.asarray(verts, np.float32).reshape(-1, 3)
        self.indices = None if indices is None else np.asarray(indices, np.int32).reshape(-1)
        self.mesh_mat = mesh_mat
        self.mats = mats
        self.tri_material = tri_material

    @property
    def n_tris(self):
        return 0 if self.indices is None else len(self.indices) // 3


DEFAULT_MAT = (0.0, 1.0, 1.0, (1.0, 1.0, 1.0))


def sphere(p, r, mat=DEFAULT_MAT, is_light=False):
    return dict(kind="sphere", p=p, s=r, mat=mat, is_light=is_light)


def plane(n, off, mat=DEFAULT_MAT):
    return dict(kind="plane", p=n, s=off, mat=mat, is_light=False)


def stock_analytic(mirror=True, floor=True):
    """main.cpp:240-255; mirror=False drops the big sphere (the cube is hidden inside it)."""
    a = []
    if mirror:
        a.append(sphere((0.0, 0.0, 0.0), 2.5, (1.0, 0.0, 1.0, (1.0, 1.0, 1.0))))
    a.append(sphere((2.0, 5.0, 1.0), 0.1, (0.0, 1.0, 1.0, (0.7, 0.7, 0.9)), True))
    a.append(sphere((-2.0, 5.0, -3.0), 0.1, (0.0, 1.0, 1.0, (0.9, 0.9, 0.4)), True))
    if floor:
        a.append(plane((0.0, 1.0, 0.0), -4.0))
    return a


def stock_spec():
    return SceneSpec(stock_analytic())


def cube_spec(mirror=False, mesh_reflect=0.0):
    v, i = scenes.cube_mesh()
    return SceneSpec(stock_analytic(mirror=mirror), v, i, (mesh_reflect, 1.0, 1.0, (1.0, 1.0, 1.0)))


ATRIUM_MATS = [
    (0.5, 1.0, 1.0, (0.85, 0.82, 0.75)),   # stone
    (0.5, 1.0, 1.0, (0.7, 0.7, 0.72)),     # floor
    (0.5, 1.0, 1.0, (0.8, 0.15, 0.12)),    # red cloth
    (0.5, 1.0, 1.0, (0.15, 0.25, 0.8)),    # blue cloth
    (0.5, 0.6, 1.0, (0.9, 0.7, 0.3)),      # brass
]


def atrium_spec(detail=1.0, per_tri_materials=True, mirror=True):
    v, i, m = scenes.atrium_mesh(detail=detail)
    a = stock_analytic(mirror=mirror, floor=False)
    if per_tri_materials:
        return SceneSpec(a, v, i, mats=ATRIUM_MATS, tri_material=m)
    return SceneSpec(a, v, i, (0.5, 1.0, 1.0, (1.0, 1.0, 1.0)))


# ------------------------------------------------------------------ to the oracle
def _omat(m):
    return O.material(m[0], m[1], m[2], m[3])


def to_oracle(spec, bvh=False):
    prims = []
    for a in spec.analytic:
        if a["kind"] == "sphere":
            prims.append(O.sphere(a["p"], a["s"], _omat(a["mat"]), a["is_light"]))
        else:
            prims.append(O.plane(a["p"], a["s"], _omat(a["mat"])))
    mesh = None
    if spec.n_tris:
        if spec.mats is not None:
            mesh = O.triangles_from_mesh(spec.verts, spec.indices, [_omat(m) for m in spec.mats], spec.tri_material)
        else:
            mesh = O.triangles_from_mesh(spec.verts, spec.indices, [_omat(spec.mesh_mat or DEFAULT_MAT)])
    return O.Scene(prims, bvh=bvh, mesh=mesh)


def ocam(cam):
    return O.camera(tuple(cam.origin), cam.pitch, cam.off_x, cam.off_y, cam.plane_z)


# ------------------------------------------------------------------ to the HIP path
def _pmat(m):
    return rt.Material(m[0], m[1], m[2], m[3])


def to_product(ctx, spec, flat=False):
    prims = []
    for a in spec.analytic:
        if a["kind"] == "sphere":
            prims.append(rt.Sphere(a["p"], a["s"], _pmat(a["mat"]), a["is_light"]))
        else:
            prims.append(rt.Plane(a["p"], a["s"], _pmat(a["mat"])))
    ctx.set_analytic(prims)
    if spec.n_tris:
        mm = _pmat(spec.mesh_mat) if spec.mesh_mat is not None else None
        if flat:
            ctx.set_triangles(scenes.flatten(spec.verts, spec.indices), mm)
        else:
            ctx.set_mesh(spec.verts, spec.indices, mm)
        if spec.mats is not None:
            ctx.set_triangle_materials([_pmat(m) for m in spec.mats], spec.tri_material)
    else:
        ctx.set_triangles(np.zeros((0, 9), np.float32))
    return ctx


def compare(gpu, ref, rgb_tol=1):
    """Parity summary between a product frame and an oracle frame (dicts of arrays)."""
    out = {}
    n = ref["hit_id"].size
    out["id_mismatch"] = int((gpu["hit_id"] != ref["hit_id"]).sum())
    d8 = np.abs(gpu["rgb8"].astype(np.int32) - ref["rgb8"].astype(np.int32)).max(axis=-1)
    out["rgb8_exact"] = int((d8 == 0).sum())
    out["rgb8_over_tol"] = int((d8 > rgb_tol).sum())
    out["rgb8_max"] = int(d8.max()) if d8.size else 0
    a, b = gpu["rgb32"].view(np.uint32), ref["rgb32"].view(np.uint32)
    out["rgb32_bit_mismatch_px"] = int((a != b).any(axis=-1).sum())
    with np.errstate(invalid="ignore"):
        out["rgb32_max_abs"] = float(np.nanmax(np.abs(gpu["rgb32"] - ref["rgb32"]))) if n else 0.0
    out["pixels"] = int(n)
    return out


# ------------------------------------------------------------------ golden cases
def lcg_floats(n, seed):
    """Deterministic uniform [0,1) floats that do not depend on numpy's generators."""
    out = np.empty(n, np.float64)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (1664525 * s + 1013904223) & 0xFFFFFFFF
        out[i] = s / 4294967296.0
    return out


def soup_spec(n_tris=40, seed=7):
    """Random triangles in front of the reference camera, four materials (one a mirror), plus
    a reflective floor plane (exercises the plane's any-t>0 self-hit rule, main.cpp:91) and
    a sphere that contains the camera (negative-t hits, main.cpp:66-67)."""
    r = lcg_floats(n_tris * 12, seed).reshape(n_tris, 12)
    c = np.stack([r[:, 0] * 8 - 4, r[:, 1] * 6 - 4, r[:, 2] * 8 - 1], 1)
    tri = c[:, None, :] + (r[:, 3:12].reshape(n_tris, 3, 3) - 0.5) * 2.5
    verts = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(n_tris * 3, dtype=np.int32)
    mats = [(0.0, 1.0, 1.0, (1.0, 1.0, 1.0)), (0.6, 0.7, 0.5, (0.9, 0.3, 0.2)), (1.0, 0.0, 1.0, (1.0, 1.0, 1.0)),
            (0.3, 1.0, 0.0, (0.2, 0.8, 0.4))]
    tm = (np.arange(n_tris) % 4).astype(np.int32)
    a = [sphere((2.0, 5.0, 1.0), 0.1, (0.0, 1.0, 1.0, (0.7, 0.7, 0.9)), True),
         plane((0.0, 1.0, 0.0), -4.0, (0.4, 1.0, 1.0, (0.8, 0.8, 0.8))),
         sphere((-2.0, 5.0, -3.0), 0.1, (0.0, 1.0, 1.0, (0.9, 0.9, 0.4)), True),
         sphere((0.0, 0.0, 3.0), 9.0, (0.2, 0.5, 1.0, (0.3, 0.3, 0.5)))]
    return SceneSpec(a, verts, idx, mats=mats, tri_material=tm)


def golden_cases():
    """name -> (spec, W, H, bounces, camera).  Small enough for the unmodified reference
    (brute force, scalar) to render in seconds."""
    cam160 = rt.Camera(pitch=0.05)
    cam200 = rt.Camera(pitch=0.04)
    return {
        "stock_160": (stock_spec(), 160, 120, 3, cam160),
        "cube_hidden_160": (cube_spec(mirror=True), 160, 120, 3, cam160),
        "cube_d1_200": (cube_spec(mirror=False), 200, 150, 1, cam200),
        "cube_refl_d4_200": (cube_spec(mirror=False, mesh_reflect=0.5), 200, 150, 4, cam200),
        "soup_d3_160": (soup_spec(), 160, 120, 3, cam160),
        "soup_d8_160": (soup_spec(n_tris=24, seed=99), 160, 120, 8, rt.Camera(origin=(0.5, -1.0, -6.0), pitch=0.05, off_x=-4.2, off_y=-3.5, plane_z=0.25)),
    }

"""The benchmark configurations, as data the HIP path (and, separately, the checker) can load.

Each workload is the reference's stock lights (Raytracer/main.cpp:245-253) plus geometry:

  cube         cube.obj geometry + floor plane, mirror sphere removed so the cube is visible
  sponza       stand-in for the absent sponza.obj: the procedural atrium (~262k triangles)
               with a mirror ball on the courtyard floor
  soup         4x4 instanced atrium (~4.2M triangles)

Mesh triangles get reflect = 0.5 (the reference gives meshes reflect = 0, main.cpp:34 + 230,
so bounces on meshes need this declared material extension).
"""
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


def sponza(W=1920, H=1080, max_depth=4, detail=ATRIUM_DETAIL):
    v, i, m = scenes.atrium_mesh(detail=detail)
    mats = [Material(*t) for t in _ATRIUM_MATS]
    return Workload("sponza-standin(atrium %d tris) %dx%d depth %d" % (len(i) // 3, W, H, max_depth),
                    [_mirror()] + _lights(), v, i, mats, m, W, H, max_depth, _courtyard_camera(W))


def soup(W=7680, H=4320, max_depth=8, detail=ATRIUM_DETAIL, nx=4, nz=4):
    v, i, m = scenes.atrium_mesh(detail=detail)
    v, i, m = scenes.instance_grid(v, i, m, nx, nz, 15.0, 23.0)
    v = v - np.array([0.5 * (nx - 1) * 15.0, 0.0, 0.0], np.float32)
    mats = [Material(*t) for t in _ATRIUM_MATS]
    return Workload("soup(%d tris) %dx%d depth %d" % (len(i) // 3, W, H, max_depth),
                    [_mirror()] + _lights(), v.astype(np.float32), i, mats, m, W, H, max_depth, _courtyard_camera(W))


def by_name(name, **kw):
    return {"cube": cube, "sponza": sponza, "soup": soup}[name](**kw)

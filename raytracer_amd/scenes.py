"""Deterministic synthetic geometry for tests and benchmarks.

The reference ships one mesh (Raytracer/cube.obj: a unit cube on y in [0,1], six quad
faces) and its headline asset sponza.obj is absent from the reference tree.  This
module provides:

  cube_mesh()         the cube.obj geometry, index order as the reference's loader
                      produces it (objloader.cpp:90-99: quad -> (0,1,2) + (0,2,3))
  atrium_mesh()       a procedural open-roof colonnaded courtyard of ~262k triangles,
                      the stand-in used wherever a config names sponza.obj
  instance_grid()     translate-and-repeat a mesh on an nx x nz grid (the 4M-triangle soup)
  write_obj()         emit a mesh as OBJ text the reference's loader can read
                      (single spaces, position indices only)

Everything is numpy, float32, seeded, and produces identical bytes on every run.
"""
import numpy as np

# cube.obj: vertices in file order, faces (1-based in the file) as quads.
_CUBE_VERTS = np.array([
    [-0.5, 0.0, -0.5], [0.5, 0.0, -0.5], [-0.5, 1.0, -0.5], [0.5, 1.0, -0.5],
    [-0.5, 0.0, 0.5], [0.5, 0.0, 0.5], [-0.5, 1.0, 0.5], [0.5, 1.0, 0.5]], dtype=np.float32)
_CUBE_QUADS = np.array([[4, 2, 1, 3], [6, 8, 7, 5], [6, 5, 1, 2], [8, 6, 2, 4], [7, 8, 4, 3], [5, 7, 3, 1]],
                       dtype=np.int32) - 1


def quads_to_indices(quads):
    """Fan split of objloader.cpp:90-99."""
    q = np.asarray(quads, dtype=np.int32).reshape(-1, 4)
    return np.stack([q[:, 0], q[:, 1], q[:, 2], q[:, 0], q[:, 2], q[:, 3]], axis=1).reshape(-1)


def cube_mesh():
    return _CUBE_VERTS.copy(), quads_to_indices(_CUBE_QUADS)


def cube_quads():
    return _CUBE_VERTS.copy(), _CUBE_QUADS.copy()


class _Builder:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.verts = []
        self.faces = []
        self.mats = []
        self.nv = 0

    def add(self, v, f, mat):
        v = np.asarray(v, dtype=np.float32).reshape(-1, 3)
        f = np.asarray(f, dtype=np.int64).reshape(-1, 3)
        self.verts.append(v)
        self.faces.append(f + self.nv)
        self.mats.append(np.full(len(f), mat, np.int32))
        self.nv += len(v)

    def grid(self, P, mat, wrap_u=False):
        """Triangulate a (nu+1) x (nv+1) lattice of points P[u, v, 3] (wrap_u closes the u direction)."""
        nu, nv = P.shape[0] - (0 if wrap_u else 1), P.shape[1] - 1
        U = P.shape[0]
        iu = np.arange(nu)[:, None]
        iv = np.arange(nv)[None, :]
        a = iu * (nv + 1) + iv
        b = ((iu + 1) % U) * (nv + 1) + iv
        c = b + 1
        d = a + 1
        f = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], 2).reshape(-1, 3)
        self.add(P.reshape(-1, 3), f, mat)

    def patch(self, p0, du, dv, nu, nv, mat, bump=0.0):
        """Flat parallelogram p0 + s*du + t*dv, nu x nv cells, with seeded displacement along the normal."""
        s = np.linspace(0.0, 1.0, nu + 1)[:, None, None]
        t = np.linspace(0.0, 1.0, nv + 1)[None, :, None]
        p0, du, dv = (np.asarray(x, dtype=np.float64) for x in (p0, du, dv))
        P = p0 + s * du + t * dv
        if bump:
            n = np.cross(du, dv)
            n /= np.linalg.norm(n)
            P = P + n * (self.rng.random((nu + 1, nv + 1, 1)) - 0.5) * bump
        self.grid(P, mat)

    def column(self, cx, cz, y0, y1, r, nseg, nh, mat, flutes=12):
        th = np.linspace(0.0, 2 * np.pi, nseg, endpoint=False)[:, None]
        h = np.linspace(0.0, 1.0, nh + 1)[None, :]
        rr = r * (1.0 - 0.12 * h ** 2) * (1.0 + 0.04 * np.cos(flutes * th))
        P = np.stack([cx + rr * np.cos(th), y0 + (y1 - y0) * h + 0 * th, cz + rr * np.sin(th)], -1)
        self.grid(P, mat, wrap_u=True)

    def arch(self, x, z0, z1, y, tube, nu, nv, mat):
        """Half torus in the plane x = const, spanning z0..z1, springing from height y."""
        R = 0.5 * (z1 - z0)
        zc = 0.5 * (z0 + z1)
        a = np.linspace(0.0, np.pi, nu + 1)[:, None]
        b = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)[None, :]
        rad = R + tube * np.cos(b)
        P = np.stack([x + tube * np.sin(b) + 0 * a, y + rad * np.sin(a), zc - rad * np.cos(a)], -1)
        # grid() wraps its first axis: put the tube angle there
        self.grid(np.transpose(P, (1, 0, 2)).copy(), mat, wrap_u=True)

    def ball(self, c, r, nu, nv, mat):
        th = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)[:, None]
        ph = np.linspace(0.02, np.pi - 0.02, nv + 1)[None, :]
        P = np.stack([c[0] + r * np.cos(th) * np.sin(ph), c[1] + r * np.cos(ph) + 0 * th,
                      c[2] + r * np.sin(th) * np.sin(ph)], -1)
        self.grid(P, mat, wrap_u=True)

    def curtain(self, x, z0, z1, y0, y1, nu, nv, mat, depth=0.25, waves=7):
        s = np.linspace(0.0, 1.0, nu + 1)[:, None]
        t = np.linspace(0.0, 1.0, nv + 1)[None, :]
        P = np.stack([x + depth * np.sin(2 * np.pi * waves * s) * (0.3 + 0.7 * t), y1 + (y0 - y1) * t + 0 * s,
                      z0 + (z1 - z0) * s + 0 * t], -1)
        self.grid(P, mat)

    def result(self):
        return (np.concatenate(self.verts).astype(np.float32), np.concatenate(self.faces).astype(np.int32).reshape(-1),
                np.concatenate(self.mats))


# material slots of atrium_mesh(): 0 stone, 1 floor, 2 red cloth, 3 blue cloth, 4 brass
ATRIUM_MATERIALS = 5


def atrium_mesh(detail=1.0, seed=0):
    """Open-roof courtyard: tiled floor, two colonnades with arches, gallery slabs, outer and
    back walls, hanging cloths and brass finials.  Sized for the reference camera at (0,0,-5)
    looking down +z and the reference lights at (2,5,1), (-2,5,-3).  detail = 1.0 gives about
    262k triangles.  Returns (verts[n,3] f32, indices[3m] i32, tri_material[m] i32)."""
    B = _Builder(seed)
    k = lambda n: max(2, int(round(n * np.sqrt(detail))))
    x_col, x_wall, z_front, z_back = 3.2, 7.0, -4.0, 18.0
    y_floor, y_cap, y_gallery, y_top = -4.0, 1.0, 3.4, 7.0
    # floor
    B.patch((-x_wall, y_floor, z_front), (0, 0, z_back - z_front), (2 * x_wall, 0, 0), k(150), k(100), 1, bump=0.01)
    # outer walls (normals face the courtyard) and back wall
    B.patch((-x_wall, y_floor, z_front), (0, y_top - y_floor, 0), (0, 0, z_back - z_front), k(70), k(140), 0, bump=0.03)
    B.patch((x_wall, y_floor, z_front), (0, 0, z_back - z_front), (0, y_top - y_floor, 0), k(140), k(70), 0, bump=0.03)
    B.patch((-x_wall, y_floor, z_back), (2 * x_wall, 0, 0), (0, y_top - y_floor, 0), k(100), k(70), 0, bump=0.03)
    zs = np.arange(0.0, 16.1, 4.0)
    for side in (-1.0, 1.0):
        x = side * x_col
        for z in zs:
            B.column(x, z, y_floor, y_cap, 0.38, k(56), k(60), 0)
            B.ball((x, y_cap + 0.25, z), 0.42, k(40), k(24), 4)
        for z0, z1 in zip(zs[:-1], zs[1:]):
            B.arch(x, z0, z1, y_cap + 0.45, 0.22, k(72), k(28), 0)
        # gallery slab between colonnade and wall: top, bottom and inner edge
        xa, xb = (x, side * x_wall) if side > 0 else (side * x_wall, x)
        B.patch((xa, y_gallery, zs[0] - 1), (0, 0, zs[-1] - zs[0] + 2), (xb - xa, 0, 0), k(120), k(30), 0, bump=0.01)
        B.patch((xa, y_gallery - 0.3, zs[0] - 1), (xb - xa, 0, 0), (0, 0, zs[-1] - zs[0] + 2), k(30), k(120), 0, bump=0.01)
        B.patch((x, y_gallery - 0.3, zs[0] - 1), (0, 0.3, 0), (0, 0, zs[-1] - zs[0] + 2), k(4), k(120), 0)
        # cloths hanging from the gallery edge between columns
        for j, (z0, z1) in enumerate(zip(zs[:-1], zs[1:])):
            B.curtain(x - side * 0.05, z0 + 0.6, z1 - 0.6, y_cap - 1.5, y_gallery - 0.3, k(90), k(40), 2 + (j & 1))
    return B.result()


def atrium2_mesh(seed=0):
    """The courtyard again, built the way a modelled asset is rather than as even grids (about 270k triangles):
    floor, walls and gallery slabs are single quads of two huge triangles; columns are 64 flutes of one
    full-height quad each (slivers of aspect > 100:1); thin rails run the length of the galleries and
    diagonal wires cross the courtyard (long, thin and not axis-aligned); next to them sit ornaments of
    ~50k triangles inside half a unit, dense capitals, and cloths hanging in three overlapping layers a
    few centimetres apart.  Same footprint, camera and lights as atrium_mesh().
    Returns (verts[n,3] f32, indices[3m] i32, tri_material[m] i32)."""
    B = _Builder(seed)
    x_col, x_wall, z_front, z_back = 3.2, 7.0, -4.0, 18.0
    y_floor, y_cap, y_gallery, y_top = -4.0, 1.0, 3.4, 7.0
    # two-triangle floor and walls
    B.patch((-x_wall, y_floor, z_front), (0, 0, z_back - z_front), (2 * x_wall, 0, 0), 1, 1, 1)
    B.patch((-x_wall, y_floor, z_front), (0, y_top - y_floor, 0), (0, 0, z_back - z_front), 1, 1, 0)
    B.patch((x_wall, y_floor, z_front), (0, 0, z_back - z_front), (0, y_top - y_floor, 0), 1, 1, 0)
    B.patch((-x_wall, y_floor, z_back), (2 * x_wall, 0, 0), (0, y_top - y_floor, 0), 1, 1, 0)
    zs = np.arange(0.0, 16.1, 4.0)
    for side in (-1.0, 1.0):
        x = side * x_col
        for z in zs:
            B.column(x, z, y_floor, y_cap, 0.38, 64, 1, 0)                    # full-height slivers
            B.ball((x, y_cap + 0.25, z), 0.42, 72, 36, 4)                      # dense capital
        for z0, z1 in zip(zs[:-1], zs[1:]):
            B.arch(x, z0, z1, y_cap + 0.45, 0.22, 72, 28, 0)
        xa, xb = (x, side * x_wall) if side > 0 else (side * x_wall, x)
        B.patch((xa, y_gallery, zs[0] - 1), (0, 0, zs[-1] - zs[0] + 2), (xb - xa, 0, 0), 1, 1, 0)
        B.patch((xa, y_gallery - 0.3, zs[0] - 1), (xb - xa, 0, 0), (0, 0, zs[-1] - zs[0] + 2), 1, 1, 0)
        B.patch((x, y_gallery - 0.3, zs[0] - 1), (0, 0.3, 0), (0, 0, zs[-1] - zs[0] + 2), 1, 1, 0)
        # rails: 18 m long, 2 cm wide
        for j in range(10):
            B.patch((x - side * 0.1, y_gallery + 0.1 * (j + 1), zs[0] - 1), (0, 0, zs[-1] - zs[0] + 2), (0, 0.02, 0), 1, 1, 4)
        # three overlapping cloth layers per bay
        for j, (z0, z1) in enumerate(zip(zs[:-1], zs[1:])):
            for layer in range(3):
                B.curtain(x - side * (0.05 + 0.03 * layer), z0 + 0.6 - 0.1 * layer, z1 - 0.6 + 0.1 * layer, y_cap - 1.5 + 0.2 * layer,
                          y_gallery - 0.3, 60, 30, 2 + ((j + layer) & 1), depth=0.25 - 0.05 * layer, waves=7 - layer)
    # wires across the courtyard, gallery edge to gallery edge, none parallel to an axis
    for j in range(24):
        za, zb = -1.0 + 0.75 * j, 17.0 - 0.75 * j
        B.patch((-x_col, y_gallery + 0.2, za), (2 * x_col, 0.5 + 0.05 * j, zb - za), (0, 0.015, 0), 1, 1, 4)
    # two ornaments of ~50k triangles each, on pedestals of two-triangle faces
    for cx in (-1.6, 1.6):
        B.ball((cx, y_floor + 1.3, 3.0), 0.5, 224, 112, 4)
        for dx_, dz_, nx_, nz_ in ((-0.4, -0.4, 0.8, 0.0), (0.4, -0.4, 0.0, 0.8), (0.4, 0.4, -0.8, 0.0), (-0.4, 0.4, 0.0, -0.8)):
            B.patch((cx + dx_, y_floor, 3.0 + dz_), (0, 0.8, 0), (nx_, 0, nz_), 1, 1, 0)
        B.patch((cx - 0.4, y_floor + 0.8, 2.6), (0, 0, 0.8), (0.8, 0, 0), 1, 1, 0)
    return B.result()


def instance_grid(verts, indices, tri_mat, nx, nz, dx, dz):
    """nx x nz translated copies (the instanced 'soup'); instance (i,j) is shifted by (i*dx, 0, j*dz)."""
    V, I, M = [], [], []
    n = len(verts)
    c = 0
    for i in range(nx):
        for j in range(nz):
            V.append(verts + np.array([i * dx, 0.0, j * dz], np.float32))
            I.append(indices + c * n)
            M.append(tri_mat)
            c += 1
    return np.concatenate(V).astype(np.float32), np.concatenate(I).astype(np.int32), np.concatenate(M)


def uv_sphere(center=(0.0, 0.0, 0.0), radius=1.0, n_lat=12, n_lon=24):
    """A latitude/longitude sphere: verts (n,3), indices (3m,), unit vertex normals (n,3)."""
    pts = [(0.0, 1.0, 0.0)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2.0 * np.pi * j / n_lon
            pts.append((np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)))
    pts.append((0.0, -1.0, 0.0))
    nrm = np.asarray(pts, np.float64)
    ring = lambda i, j: 1 + (i - 1) * n_lon + j % n_lon
    idx = []
    for j in range(n_lon):
        idx += [0, ring(1, j + 1), ring(1, j)]
        idx += [len(pts) - 1, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)]
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j + 1), ring(i + 1, j)
            idx += [a, b, c, a, c, d]
    verts = (nrm * radius + np.asarray(center, np.float64)).astype(np.float32)
    return verts, np.asarray(idx, np.int32), nrm.astype(np.float32)


def corner_normals(vertex_normals, indices):
    """(n_tris, 3, 3): the vertex normal at each triangle corner, the layout of rt_set_triangle_normals."""
    return np.asarray(vertex_normals, np.float32)[np.asarray(indices, np.int64)].reshape(-1, 3, 3)


def flatten(verts, indices):
    """main.cpp:228-231 on the host: (n_tris, 9) array of v1,v2,v3."""
    return np.asarray(verts, np.float32)[np.asarray(indices).reshape(-1, 3)].reshape(-1, 9)


def write_obj(path, verts, faces):
    """faces: (m,3) or (m,4) 0-based.  Single-space tokens, 6 decimals like cube.obj."""
    faces = np.asarray(faces)
    with open(path, "w") as f:
        f.write("o mesh\n")
        for v in np.asarray(verts, np.float32):
            f.write("v %.6f %.6f %.6f\n" % (v[0], v[1], v[2]))
        for q in faces:
            f.write("f " + " ".join(str(int(i) + 1) for i in q) + "\n")

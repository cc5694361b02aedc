"""raytracer_amd — MI355X-native trace path of zanders3/raytracer behind a C ABI.

This package is a thin ctypes binding of librt_amd.so (include/rt_shim.h) plus a
Python mirror of the reference's scene-building seam (Raytracer/main.cpp:240-257):
the same primitive kinds, material fields and defaults, in the same scene order.
There is no CPU fallback: if the HIP library is missing, or no HIP device is
usable, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")   # RT_AMD_LIB: A/B-test another build

RT_SPHERE, RT_PLANE = 0, 1
OPT_ACCEL, OPT_COUNT_TESTS, OPT_LEAF_MAX, OPT_TIMING, OPT_PIPELINE = 1, 2, 3, 4, 5
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2
OPT_BUILDER = 6
OPT_GRAPH = 7
OPT_SPLIT = 8
OPT_CLAIM_ORDER = 9
BUILDER_AUTO, BUILDER_LBVH, BUILDER_PLOC, BUILDER_SAH = 0, 1, 2, 3
ACCEL_AUTO, ACCEL_LINEAR, ACCEL_LBVH = 0, 1, 2
RT_MAX_ANALYTIC, RT_MAX_DEPTH, RT_BRUTE_MAX = 1024, 64, 48

# every symbol include/rt_shim.h declares
ABI_SYMBOLS = (
    "rt_abi_version", "rt_last_error", "rt_create", "rt_destroy", "rt_set_option", "rt_set_analytic",
    "rt_set_mesh", "rt_set_triangles", "rt_set_triangle_materials", "rt_set_camera", "rt_render_frame",
    "rt_render_rows_device", "rt_deinterleave_bands", "rt_get_stats", "rt_synchronize", "rt_render_frame_planes",
    "rt_update_triangles", "rt_update_mesh_vertices", "rt_set_triangle_normals", "rt_set_triangle_texcoords",
    "rt_render_rows_device_noalloc", "rt_render_rows_device_slot", "rt_get_stats_sized",
    "rt_alloc_host_buffer", "rt_free_host_buffer",
)
ABI_VERSION = 2     # RT_ABI_VERSION of include/rt_shim.h this binding mirrors
MAX_FRAME_SLOTS = 8
NOALLOC_SCRATCH_BYTES = 64 << 20


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


class Material(C.Structure):
    """struct Material, main.cpp:29-36 (defaults reflect 0, diffuse 1, spec 1, white)."""
    _fields_ = [("reflect", C.c_float), ("diffuse", C.c_float), ("spec", C.c_float), ("color", C.c_float * 3)]

    def __init__(self, reflect=0.0, diffuse=1.0, spec=1.0, color=(1.0, 1.0, 1.0)):
        super().__init__()
        self.reflect, self.diffuse, self.spec = reflect, diffuse, spec
        self.color[:] = [float(c) for c in color]


class Analytic(C.Structure):
    _fields_ = [("kind", C.c_int32), ("is_light", C.c_int32), ("mat", Material), ("p", C.c_float * 3), ("s", C.c_float)]


class Camera(C.Structure):
    """The literals of main.cpp:267-270."""
    _fields_ = [("origin", C.c_float * 3), ("pitch", C.c_float), ("off_x", C.c_float), ("off_y", C.c_float),
                ("plane_z", C.c_float)]

    def __init__(self, origin=(0.0, 0.0, -5.0), pitch=0.01, off_x=-4.0, off_y=-4.0, plane_z=0.0):
        super().__init__()
        self.origin[:] = [float(c) for c in origin]
        self.pitch, self.off_x, self.off_y, self.plane_z = pitch, off_x, off_y, plane_z


class RowSet(C.Structure):
    _fields_ = [("first_band", C.c_int32), ("band_rows", C.c_int32), ("band_stride", C.c_int32), ("n_rows", C.c_int32)]


class FramePlanes(C.Structure):
    _fields_ = [("rgb8", C.c_void_p), ("hit_id", C.c_void_p), ("rgb32", C.c_void_p), ("hit_t", C.c_void_p), ("normal", C.c_void_p), ("uv", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("analytic_tests", C.c_uint64),
                ("ms_frame", C.c_float), ("ms_trace", C.c_float), ("ms_resolve", C.c_float),
                ("n_launches", C.c_int32), ("n_analytic", C.c_int32), ("n_triangles", C.c_int32),
                ("bvh_nodes", C.c_int32), ("bvh_depth", C.c_int32), ("bvh_leaf_max", C.c_int32),
                ("accel", C.c_int32), ("ms_build", C.c_float), ("ms_traverse", C.c_float),
                ("bvh_sah_cost", C.c_float), ("bvh_refs", C.c_int32),
                ("box_tests_occupied", C.c_uint64), ("staging_ms", C.c_float), ("reserved0", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def Sphere(pos, radius, mat=None, is_light=False):
    """struct Sphere, main.cpp:48-74."""
    a = Analytic()
    a.kind, a.is_light = RT_SPHERE, int(bool(is_light))
    a.mat = mat if mat is not None else Material()
    a.p[:] = [float(c) for c in pos]
    a.s = radius
    return a


def Plane(normal, offset, mat=None):
    """struct Plane, main.cpp:76-98."""
    a = Analytic()
    a.kind, a.is_light = RT_PLANE, 0
    a.mat = mat if mat is not None else Material()
    a.p[:] = [float(c) for c in normal]
    a.s = offset
    return a


def stock_scene():
    """The scene of main.cpp:240-255: mirror sphere, two light spheres, floor plane."""
    return [
        Sphere((0.0, 0.0, 0.0), 2.5, Material(reflect=1.0, diffuse=0.0)),
        Sphere((2.0, 5.0, 1.0), 0.1, Material(color=(0.7, 0.7, 0.9)), is_light=True),
        Sphere((-2.0, 5.0, -3.0), 0.1, Material(color=(0.9, 0.9, 0.4)), is_light=True),
        Plane((0.0, 1.0, 0.0), -4.0),
    ]


def camera_for_width(W):
    """Reference camera generalised to another width: the 0.01 pixel pitch of
    main.cpp:268-269 is 8/800, so pitch = float32(8)/float32(W) spans the same
    x range [-4, 4); at W = 800 it is bit-identical to the literal 0.01f."""
    return Camera(pitch=float(np.float32(8.0) / np.float32(W)))


_lib = None


def _share_torch_hip_runtime():
    """A PyTorch-ROCm wheel carries its own libamdhip64.so (soname libamdhip64.so.7, the same
    as /opt/rocm's).  Two HIP runtimes in one process cannot both own the GPU: whichever
    initialises second finds no devices.  Loading the wheel's copy first makes the dynamic
    loader resolve this library's libamdhip64.so.7 dependency to it, so torch tensors and this
    library's kernels share one runtime whatever the import order.  Without torch installed the
    system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)


def load_library():
    """Load librt_amd.so and declare its prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `python raytracer_amd/build.py` "
                           "(there is no CPU fallback for the trace path)")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.rt_abi_version.restype = i32
    L.rt_last_error.restype = C.c_char_p
    L.rt_create.restype = vp
    L.rt_create.argtypes = [i32]
    L.rt_destroy.argtypes = [vp]
    L.rt_set_option.argtypes = [vp, i32, i32]
    L.rt_set_analytic.argtypes = [vp, C.POINTER(Analytic), i32]
    L.rt_set_mesh.argtypes = [vp, vp, i32, vp, i32, C.POINTER(Material)]
    L.rt_set_triangles.argtypes = [vp, vp, i32, C.POINTER(Material)]
    L.rt_set_triangle_materials.argtypes = [vp, C.POINTER(Material), i32, vp, i32]
    L.rt_set_triangle_normals.argtypes = [vp, vp, i32]
    L.rt_set_triangle_texcoords.argtypes = [vp, vp, i32]
    L.rt_update_triangles.argtypes = [vp, vp, i32]
    L.rt_update_mesh_vertices.argtypes = [vp, vp, i32]
    L.rt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.rt_render_frame.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp]
    L.rt_render_frame_planes.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(FramePlanes)]
    L.rt_render_rows_device.argtypes = [vp, i32, i32, C.POINTER(RowSet), i32, vp, vp, vp, vp, vp]
    L.rt_render_rows_device_noalloc.argtypes = [vp, i32, i32, C.POINTER(RowSet), i32, vp, vp, vp, vp, vp]
    L.rt_render_rows_device_slot.argtypes = [vp, i32, i32, i32, C.POINTER(RowSet), i32, vp, vp, vp, vp, vp]
    L.rt_deinterleave_bands.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_get_stats_sized.argtypes = [vp, vp, C.c_uint64]
    L.rt_alloc_host_buffer.restype = vp
    L.rt_alloc_host_buffer.argtypes = [vp, C.c_uint64]
    L.rt_free_host_buffer.argtypes = [vp, vp]
    if L.rt_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} speaks ABI version {L.rt_abi_version()}, this binding {ABI_VERSION}: rebuild the library")
    L.rt_synchronize.argtypes = [vp]
    _lib = L
    return L


class Context:
    """One trace context on one GPU (rt_ctx)."""

    def __init__(self, device=0):
        self.L = load_library()
        self._h = self.L.rt_create(device)
        if not self._h:
            raise RtError(-2, self.L.rt_last_error().decode())
        self.device = device
        self.n_analytic = 0
        self.n_tris = 0

    def close(self):
        if getattr(self, "_h", None):
            self.L.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self.L.rt_last_error().decode())

    def set_option(self, opt, value):
        self._check(self.L.rt_set_option(self._h, opt, value))

    def set_analytic(self, prims):
        n = len(prims)
        arr = (Analytic * max(n, 1))(*prims)
        self._check(self.L.rt_set_analytic(self._h, arr, n))
        self.n_analytic = n

    def set_mesh(self, verts, indices, mat=None):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        self._check(self.L.rt_set_mesh(self._h, v.ctypes.data, v.shape[0], i.ctypes.data, i.shape[0],
                                       C.byref(mat) if mat is not None else None))
        self.n_tris = i.shape[0] // 3

    def set_triangles(self, tri_verts, mat=None):
        t = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        self._check(self.L.rt_set_triangles(self._h, t.ctypes.data, t.shape[0],
                                            C.byref(mat) if mat is not None else None))
        self.n_tris = t.shape[0]

    def set_triangle_materials(self, mats, tri_material):
        tm = np.ascontiguousarray(tri_material, dtype=np.int32).reshape(-1)
        arr = (Material * len(mats))(*mats)
        self._check(self.L.rt_set_triangle_materials(self._h, arr, len(mats), tm.ctypes.data, tm.shape[0]))

    def set_triangle_normals(self, corner_normals):
        """rt_set_triangle_normals: (n_tris, 3, 3) shading normals at v1, v2, v3; None = geometric normals."""
        if corner_normals is None:
            self._check(self.L.rt_set_triangle_normals(self._h, None, 0))
            return
        v = np.ascontiguousarray(corner_normals, dtype=np.float32).reshape(-1, 9)
        self._check(self.L.rt_set_triangle_normals(self._h, v.ctypes.data, v.shape[0]))

    def set_triangle_texcoords(self, corner_uv):
        """rt_set_triangle_texcoords: (n_tris, 3, 2) texture coordinates at v1, v2, v3; None drops them."""
        if corner_uv is None:
            self._check(self.L.rt_set_triangle_texcoords(self._h, None, 0))
            return
        v = np.ascontiguousarray(corner_uv, dtype=np.float32).reshape(-1, 6)
        self._check(self.L.rt_set_triangle_texcoords(self._h, v.ctypes.data, v.shape[0]))

    def update_triangles(self, tri_verts):
        """rt_update_triangles: moved vertices for the loaded triangles (same count and order)."""
        v = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        self._check(self.L.rt_update_triangles(self._h, v.ctypes.data, v.shape[0]))

    def update_mesh_vertices(self, verts):
        """rt_update_mesh_vertices: moved vertices under the indices given to set_mesh."""
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self._check(self.L.rt_update_mesh_vertices(self._h, v.ctypes.data, v.shape[0]))

    def set_camera(self, cam=None):
        self._check(self.L.rt_set_camera(self._h, C.byref(cam) if cam is not None else None))

    def render_frame(self, W, H, y0=0, y1=None, max_depth=3, want=("rgb8", "hit_id", "rgb32")):
        """rt_render_frame: host numpy outputs for rows [y0, y1)."""
        y1 = H if y1 is None else y1
        rows = max(y1 - y0, 0)
        rgb8 = np.zeros((rows, W, 3), np.uint8) if "rgb8" in want else None
        hid = np.full((rows, W), -2, np.int32) if "hit_id" in want else None
        rgb32 = np.zeros((rows, W, 3), np.float32) if "rgb32" in want else None
        ptr = lambda a: a.ctypes.data if a is not None and a.size else None
        self._check(self.L.rt_render_frame(self._h, W, H, y0, y1, max_depth, ptr(rgb8), ptr(hid), ptr(rgb32)))
        return dict(rgb8=rgb8, hit_id=hid, rgb32=rgb32)

    def render_frame_planes(self, W, H, y0=0, y1=None, max_depth=3):
        """rt_render_frame_planes: colour, id, float colour, nearest t and normal of the primary hit."""
        y1 = H if y1 is None else y1
        rows = max(y1 - y0, 0)
        out = dict(rgb8=np.zeros((rows, W, 3), np.uint8), hit_id=np.full((rows, W), -2, np.int32),
                   rgb32=np.zeros((rows, W, 3), np.float32), hit_t=np.zeros((rows, W), np.float32),
                   normal=np.zeros((rows, W, 3), np.float32), uv=np.zeros((rows, W, 2), np.float32))
        fp = FramePlanes(*[out[k].ctypes.data if out[k].size else None for k in ("rgb8", "hit_id", "rgb32", "hit_t", "normal", "uv")])
        self._check(self.L.rt_render_frame_planes(self._h, W, H, y0, y1, max_depth, C.byref(fp)))
        return out

    def render_rows_device(self, W, H, rowset, max_depth, d_rgb8=None, d_hit_id=None, d_rgb32=None, d_hit_t=None,
                           stream=None, slot=0):
        """rt_render_rows_device (slot 0) / rt_render_rows_device_slot: raw device pointers (ints), asynchronous on
        `stream`.  Frames of different slots may be in flight at the same time (on different streams)."""
        if slot:
            self._check(self.L.rt_render_rows_device_slot(self._h, slot, W, H, C.byref(rowset), max_depth, d_rgb8, d_hit_id,
                                                          d_rgb32, d_hit_t, stream))
        else:
            self._check(self.L.rt_render_rows_device(self._h, W, H, C.byref(rowset), max_depth, d_rgb8, d_hit_id, d_rgb32,
                                                     d_hit_t, stream))

    def render_rows_device_noalloc(self, W, H, rowset, max_depth, d_rgb8=None, d_hit_id=None, d_rgb32=None, d_hit_t=None,
                                   stream=None):
        """rt_render_rows_device_noalloc: the same frame without any device allocation (static scratch, size-limited)."""
        self._check(self.L.rt_render_rows_device_noalloc(self._h, W, H, C.byref(rowset), max_depth, d_rgb8, d_hit_id,
                                                         d_rgb32, d_hit_t, stream))

    def deinterleave_bands(self, d_src, d_dst, W, H, band_rows, n_parts, rows_per_part, bytes_per_pixel, stream=None):
        self._check(self.L.rt_deinterleave_bands(self._h, d_src, d_dst, W, H, band_rows, n_parts, rows_per_part,
                                                 bytes_per_pixel, stream))

    def alloc_host_buffer(self, shape, dtype=np.uint8):
        """rt_alloc_host_buffer as a numpy array: pinned memory the device writes frames into directly.  The array is a
        view; it dies with free_host_buffer() or the context."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.L.rt_alloc_host_buffer(self._h, n)
        if not p:
            raise RtError(-3, self.L.rt_last_error().decode())
        return np.frombuffer((C.c_ubyte * n).from_address(p), dtype=dtype).reshape(shape)

    def free_host_buffer(self, array):
        self._check(self.L.rt_free_host_buffer(self._h, array.ctypes.data))

    def render_frame_into(self, W, H, rgb8, max_depth=3, y0=0, y1=None):
        """rt_render_frame into a caller-owned uint8 array (rows [y0, y1) x W x 3), the reference's `image` (main.cpp:259)."""
        y1 = H if y1 is None else y1
        assert rgb8.dtype == np.uint8 and rgb8.flags["C_CONTIGUOUS"] and rgb8.size == (y1 - y0) * W * 3
        self._check(self.L.rt_render_frame(self._h, W, H, y0, y1, max_depth, rgb8.ctypes.data, None, None))

    def synchronize(self):
        self._check(self.L.rt_synchronize(self._h))

    def stats(self):
        s = Stats()
        self._check(self.L.rt_get_stats(self._h, C.byref(s)))
        return s.as_dict()


def band_rowset(H, band_rows, part, n_parts):
    """Rows of part `part` when the image is dealt out in `band_rows`-row bands round-robin."""
    n_bands = (H + band_rows - 1) // band_rows
    mine = range(part, n_bands, n_parts)
    n_rows = sum(min(band_rows, H - b * band_rows) for b in mine)
    return RowSet(part, band_rows, n_parts, n_rows)


def rows_per_part(H, band_rows, n_parts):
    """Row capacity every part's buffer needs so that all parts gather at equal size."""
    n_bands = (H + band_rows - 1) // band_rows
    return ((n_bands + n_parts - 1) // n_parts) * band_rows

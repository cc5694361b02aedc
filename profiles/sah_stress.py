"""profiles/sah_stress.py [cases]: many random meshes (uniform, clustered, collinear, duplicated, huge and tiny extents) through the
SAH builder.  The builder checks its own tree (parents, ranges, order array) and fails the call otherwise; every mesh is also
rendered at 24x16 against the linear scan."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as rt

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(12345)
W, H = 24, 16
cam = rt.Camera(pitch=8.0 / W)
bad = 0
for case in range(cases):
    n = int(rng.choice([49, 50, 63, 64, 65, 66, 127, 128, 129, 130, 257, 1000, 4097, 20000, int(rng.integers(49, 30000))]))
    kind = case % 6
    if kind == 0:
        c = rng.uniform(-4, 4, (n, 3))
    elif kind == 1:                       # a few tight clusters
        c = rng.normal(0, 0.02, (n, 3)) + rng.uniform(-4, 4, (8, 3))[rng.integers(0, 8, n)]
    elif kind == 2:                       # on a line, spacing growing geometrically
        c = np.zeros((n, 3)); c[:, 0] = np.cumsum(1.0002 ** np.arange(n)) * 1e-3 - 4
    elif kind == 3:                       # many exact duplicates
        c = rng.uniform(-4, 4, (max(n // 50, 2), 3))[rng.integers(0, max(n // 50, 2), n)]
    elif kind == 4:                       # one far outlier stretches the scene
        c = rng.uniform(-1, 1, (n, 3)); c[0] = (3e4, -2e4, 1e4)
    else:                                 # flat sheet
        c = rng.uniform(-4, 4, (n, 3)); c[:, 1] = -2.0
    size = 0.05 if kind != 3 else 0.3
    tri = c[:, None, :] + rng.uniform(-size, size, (n, 3, 3))
    if kind == 3:
        tri = np.repeat(tri[: max(n // 50, 2)], n // max(n // 50, 2) + 1, axis=0)[:n]
    tri[:, :, 2] += 4.0                   # in front of the camera
    verts = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(n * 3, dtype=np.int32)
    outs = []
    for accel in (rt.ACCEL_LINEAR, rt.ACCEL_LBVH):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.OPT_ACCEL, accel)
            ctx.set_option(rt.OPT_BUILDER, rt.BUILDER_SAH)
            ctx.set_analytic([rt.Sphere((2.0, 5.0, 1.0), 0.1, None, True)])
            ctx.set_mesh(verts, idx, rt.Material(reflect=0.3))
            ctx.set_camera(cam)
            g = ctx.render_frame(W, H, max_depth=2, want=("hit_id", "rgb32"))
            st = ctx.stats()
        outs.append((g, st))
    same = np.array_equal(outs[0][0]["hit_id"], outs[1][0]["hit_id"]) and np.array_equal(outs[0][0]["rgb32"].view(np.uint32), outs[1][0]["rgb32"].view(np.uint32))
    if not same or outs[1][1]["bvh_depth"] > 64:
        bad += 1
        print("case %d kind %d n %d: MISMATCH (depth %d)" % (case, kind, n, outs[1][1]["bvh_depth"]), flush=True)
    elif case % 20 == 0:
        print("case %d kind %d n %d ok, depth %d, SAH %.1f" % (case, kind, n, outs[1][1]["bvh_depth"], outs[1][1]["bvh_sah_cost"]), flush=True)
print("%d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)

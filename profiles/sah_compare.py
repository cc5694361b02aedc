"""profiles/sah_compare.py [workload]: is the on-device hierarchy as good as a top-down SAH build?
Prints the surface-area-heuristic cost (node cost 1, triangle cost 1, leaves of <= 2 triangles; rt_stats.bvh_sah_cost)
of the GPU's PLOC and Karras trees next to the cost of a binned (32 bins, 3 axes) top-down SAH tree built here on the
host over the same triangle boxes.  Needs a GPU for the first two numbers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import raytracer_amd as rt
from raytracer_amd import workloads


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return 2.0 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])


def host_sah_cost(lo, hi, leaf_max=2, bins=32):
    n = len(lo)
    cen = 0.5 * (lo + hi)
    root_area = area(lo.min(0), hi.max(0))
    total = 0.0
    stack = [np.arange(n)]
    done, t_last = 0, time.time()
    while stack:
        idx = stack.pop()
        done += 1
        if time.time() - t_last > 30:
            print("  ... %d nodes, %d subtrees waiting" % (done, len(stack)), flush=True)
            t_last = time.time()
        blo, bhi = lo[idx].min(0), hi[idx].max(0)
        a = area(blo, bhi)
        if len(idx) <= leaf_max:
            total += a * len(idx)
            continue
        total += a
        best = (np.inf, None)
        clo, chi = cen[idx].min(0), cen[idx].max(0)
        for ax in range(3):
            ext = chi[ax] - clo[ax]
            if ext <= 0:
                continue
            b = np.minimum(((cen[idx, ax] - clo[ax]) / ext * bins).astype(np.int64), bins - 1)
            cnt = np.bincount(b, minlength=bins)
            l = np.full((bins, 3), np.inf); h = np.full((bins, 3), -np.inf)
            np.minimum.at(l, b, lo[idx]); np.maximum.at(h, b, hi[idx])
            ll = np.minimum.accumulate(l, 0); lh = np.maximum.accumulate(h, 0)
            rl = np.minimum.accumulate(l[::-1], 0)[::-1]; rh = np.maximum.accumulate(h[::-1], 0)[::-1]
            cl = np.cumsum(cnt)
            for s in range(bins - 1):
                nl, nr = cl[s], len(idx) - cl[s]
                if nl == 0 or nr == 0:
                    continue
                c = area(ll[s], lh[s]) * nl + area(rl[s + 1], rh[s + 1]) * nr
                if c < best[0]:
                    best = (c, (ax, s, b))
        if best[1] is None:          # all centroids equal: split in the middle
            m = len(idx) // 2
            stack.append(idx[:m]); stack.append(idx[m:])
            continue
        ax, s, b = best[1]
        left = b <= s
        stack.append(idx[left]); stack.append(idx[~left])
    return total / root_area


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    wl = workloads.by_name(name, W=64, H=36)
    for builder, label in ((rt.BUILDER_SAH, "binned SAH (default)"), (rt.BUILDER_PLOC, "PLOC"), (rt.BUILDER_LBVH, "Karras LBVH")):
        for split in (1, 0):
            with rt.Context(0) as ctx:
                ctx.set_option(rt.OPT_BUILDER, builder)
                ctx.set_option(rt.OPT_SPLIT, split)
                wl.load(ctx)
                st = ctx.stats()
            print("GPU %-16s split %-3s references %8d  SAH cost %8.2f  build %.2f ms" % (label, "off" if split == 1 else "on", st["bvh_refs"], st["bvh_sah_cost"], st["ms_build"]))
    tri = wl.verts[wl.indices.reshape(-1, 3)].astype(np.float64)
    lo, hi = tri.min(1), tri.max(1)
    t0 = time.time()
    c = host_sah_cost(lo, hi)
    print("host binned top-down SAH (32 bins, whole triangles)        SAH cost %8.2f  (%.0f s)" % (c, time.time() - t0))


if __name__ == "__main__":
    main()

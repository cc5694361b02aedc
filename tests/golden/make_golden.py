"""Generate the golden fixtures from the UNMODIFIED reference (oracle/_ref, built by
oracle/build_ref.sh from /root/reference).  Run where the reference tree is mounted:

    python tests/golden/make_golden.py

Outputs (committed): stock_800x600_d3.npz, ref_scenes.npz, obj_cases.json.
Fixtures are data only: inputs and the reference's outputs.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from common import O, golden_cases, ocam   # noqa: E402

OBJ_CASES = {
    "quads_trailing_space": "o q\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1/1/1 2/2/2 3/3/3 4/4/4 \n",
    "tris_plain": "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4\nf 2 4 3\n",
    "tri_trailing_space": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 3 2 1 \n",
    "double_slash": "v 0 0 0\nv 2 0 0\nv 0 2 0\nv 2 2 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 2//1 4//1 3//1\n",
    "comments_groups_mtl": "# comment\nmtllib x.mtl\no thing\ng grp\nusemtl m\nv 0.5 -1.25 3e-1\nv 1 2 3\nv -1 -2 -3\ns off\nf 1 2 3\n",
    "crlf": "v 0 0 0\r\nv 1 0 0\r\nv 0 1 0.5\r\nf 1 2 3\r\n",
    "short_vertex": "v 1 2\nv 4 5 6\nv 7 8 9\nf 1 2 3\n",
    "pentagon_truncated": "v 0 0 0\nv 1 0 0\nv 2 1 0\nv 1 2 0\nv 0 1 0\nf 1 2 3 4 5\n",
    "no_newline_at_end": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3",
    "vt_vn_ignored": "v 0 0 0\nvt 0.1 0.2\nvn 0 1 0\nv 1 0 0\nvt 0.3 0.4\nv 0 1 0\nf 1/1/1 2/2/1 3/2/1\n",
    "tab_is_not_a_separator": "v\t1 2 3\nv 1 2 3\nv 4 5 6\nv 7 8 9\nf 1 2 3\n",
    "leading_space_line": " v 9 9 9\nv 1 2 3\nv 4 5 6\nv 7 8 9\nf 3 2 1\n",
}


def main():
    if not O.Reference.available():
        raise SystemExit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")
    # (i) the reference's own frame: main() -> setup(), -O0 like the Xcode Debug scheme
    ref0 = O.Reference(o0=True)
    img = ref0.stock_frame()
    md5 = hashlib.md5(b"P6\n800 600\n255\n" + img.tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, "stock_800x600_d3.npz"), rgb8=img, ppm_md5=np.array(md5))
    print("stock frame md5", md5)

    # (ii) the reference's raytrace()/Raycast() on the shared golden scenes
    ref = O.Reference()
    out = {}
    for name, (spec, W, H, bounces, cam) in golden_cases().items():
        ref.clear()
        for a in spec.analytic:
            m = O.material(*a["mat"][:3], a["mat"][3])
            if a["kind"] == "sphere":
                ref.L.ref_add_sphere(O._f3(a["p"]), a["s"], ref._mat(m), int(a["is_light"]))
            else:
                ref.L.ref_add_plane(O._f3(a["p"]), a["s"], ref._mat(m))
        if spec.n_tris:
            idx = spec.indices.reshape(-1, 3)
            for k, (i0, i1, i2) in enumerate(idx):
                mt = spec.mats[int(spec.tri_material[k])] if spec.mats is not None else spec.mesh_mat
                ref.add_triangle(spec.verts[i0], spec.verts[i1], spec.verts[i2], O.material(*mt[:3], mt[3]))
        r = ref.render(W, H, bounces=bounces, cam=ocam(cam))
        for k in ("rgb8", "rgb32", "hit_id", "hit_t"):
            out[name + "/" + k] = r[k]
        print(name, "hits", int((r["hit_id"] >= 0).sum()), "of", W * H)
    ref.clear()
    np.savez_compressed(os.path.join(HERE, "ref_scenes.npz"), **out)

    # (iii) the reference's OBJ parser on small texts (and on its own cube.obj)
    cases = {}
    with tempfile.TemporaryDirectory() as td:
        for name, text in OBJ_CASES.items():
            p = os.path.join(td, name + ".obj")
            with open(p, "wb") as f:
                f.write(text.encode())
            v, i = ref.parse_obj(p)
            cases[name] = {"text": text, "verts": [float(x) for x in v.reshape(-1)], "indices": [int(x) for x in i]}
        v, i = ref.parse_obj(os.path.join(td, "missing.obj"))
        cases["__missing_file__"] = {"text": None, "verts": [float(x) for x in v.reshape(-1)], "indices": [int(x) for x in i]}
    v, i = ref.parse_obj("/root/reference/Raytracer/cube.obj")
    cases["__reference_cube__"] = {"text": None, "verts": [float(x) for x in v.reshape(-1)], "indices": [int(x) for x in i]}
    with open(os.path.join(HERE, "obj_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()

"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared
symbol, the C++ host's OBJ loader reproduces the reference parser's outputs, the band
partition covers the frame exactly once, and the synthetic scenes are deterministic."""
import ctypes
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from common import GOLDEN, ROOT, rt, scenes

HOST_BIN = os.environ.get("RT_HOST_BIN") or os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")      # (RT_HOST_BIN: another build of the host, e.g. a sanitized one)


def test_library_exports_every_declared_symbol():
    L = rt.load_library()
    header = open(os.path.join(ROOT, "include", "rt_shim.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header))
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.rt_abi_version() == rt.ABI_VERSION == 2
    version = int(re.search(r"#define RT_ABI_VERSION (\d+)", header).group(1))
    assert version == rt.ABI_VERSION


def test_build_entry_point_accepts_the_libraries_in_the_tree():
    # the part of __graft_entry__.build() after the compile: both libraries load, export every entry point and report
    # the ABI version of the Python host
    import __graft_entry__ as entry
    entry.check_libraries()


def test_struct_layouts_match_the_header():
    # sizes the C side assumes (include/rt_shim.h); a mismatch would corrupt every call
    assert ctypes.sizeof(rt.Material) == 24
    assert ctypes.sizeof(rt.Analytic) == 8 + 24 + 12 + 4
    assert ctypes.sizeof(rt.Camera) == 28
    assert ctypes.sizeof(rt.RowSet) == 16
    assert ctypes.sizeof(rt.Stats) == 120  # 6 x u64 + 3 x f32 + 7 x i32 + 3 x f32 + 1 x i32, then (ABI 2) u64 + f32 + i32
    # the fields of ABI version 1 keep their offsets: fields are only ever appended
    assert rt.Stats.bvh_refs.offset == 100 and rt.Stats.box_tests_occupied.offset == 104


def test_null_context_is_rejected_without_touching_a_gpu():
    L = rt.load_library()
    assert L.rt_set_option(None, rt.OPT_ACCEL, 0) == -1
    assert b"null" in L.rt_last_error()
    assert L.rt_render_frame(None, 8, 8, 0, 8, 1, None, None, None) == -1
    rs = rt.RowSet(0, 1, 1, 8)
    assert L.rt_render_rows_device_slot(None, 1, 8, 8, rs, 1, None, None, None, None, None) == -1
    assert L.rt_render_rows_device_noalloc(None, 8, 8, rs, 1, None, None, None, None, None) == -1
    s = rt.Stats()
    assert L.rt_get_stats_sized(None, ctypes.addressof(s), ctypes.sizeof(s)) == -1
    assert not L.rt_alloc_host_buffer(None, 64)
    assert L.rt_free_host_buffer(None, ctypes.addressof(s)) == -1


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers must compile as strict C99 (a C translation unit that calls through them,
    warnings as errors)."""
    src = tmp_path / "abi.c"
    src.write_text('#include "rt_shim.h"\n#include "rt_objhost.h"\n'
                   'int main(void) { rt_rowset r = { 0, 1, 1, 8 };\n'
                   '  return rt_abi_version() > 0 && rt_render_rows_device_slot(0, RT_MAX_FRAME_SLOTS - 1, 8, 8, &r, 1, 0, 0, 0, 0, 0) == RT_E_INVALID ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)], check=True)


def _dump(path):
    out = subprocess.run([HOST_BIN, "--dump-obj", path], check=True, capture_output=True, text=True).stdout.splitlines()
    head = dict(zip(out[0].split()[0::2], map(int, out[0].split()[1::2])))
    v = np.array([[float(x) for x in l.split()[1:]] for l in out if l.startswith("v ")], np.float32).reshape(-1)
    i = [int(l.split()[1]) for l in out if l.startswith("i ")]
    return head, v, i


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_obj_loader_matches_reference_parser_outputs():
    cases = json.load(open(os.path.join(GOLDEN, "obj_cases.json")))
    with tempfile.TemporaryDirectory() as td:
        for name, c in cases.items():
            if c["text"] is None:
                continue
            p = os.path.join(td, "c.obj")
            with open(p, "wb") as f:
                f.write(c["text"].encode())
            head, v, i = _dump(p)
            assert head["malformed"] == 0, name
            assert np.array_equal(v, np.array(c["verts"], np.float32)), name
            assert i == c["indices"], name
        head, v, i = _dump(os.path.join(td, "missing.obj"))      # silent on a missing file (objloader.cpp:22,104)
        assert head["verts"] == 0 and head["indices"] == 0
        # a face with an empty corner is what the reference asserts on (objloader.cpp:83)
        p = os.path.join(td, "bad.obj")
        open(p, "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1  2 3\n")
        head, v, i = _dump(p)
        assert head["malformed"] == 1


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_host_loader_on_written_cube_equals_reference_cube():
    ref = json.load(open(os.path.join(GOLDEN, "obj_cases.json")))["__reference_cube__"]
    v, q = scenes.cube_quads()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "cube.obj")
        scenes.write_obj(p, v, q)
        head, hv, hi = _dump(p)
    assert np.array_equal(hv, np.array(ref["verts"], np.float32))
    assert hi == ref["indices"]
    cv, ci = scenes.cube_mesh()
    assert list(ci) == ref["indices"] and np.array_equal(cv.reshape(-1), np.array(ref["verts"], np.float32))


@pytest.mark.parametrize("H,band,parts", [(1080, 16, 8), (1080, 16, 3), (600, 16, 4), (37, 8, 2), (5, 16, 4), (64, 1, 7)])
def test_band_partition_covers_every_row_once(H, band, parts):
    from raytracer_amd.dist import local_rows, deinterleave_numpy
    seen = np.zeros(H, int)
    cap = rt.rows_per_part(H, band, parts)
    gathered = np.full((parts, cap, 3), -1, np.int64)
    for p in range(parts):
        ys = local_rows(H, band, p, parts)
        rs = rt.band_rowset(H, band, p, parts)
        assert len(ys) == rs.n_rows <= cap
        assert (ys < H).all() and (np.diff(ys) > 0).all()
        seen[ys] += 1
        gathered[p, : len(ys), :] = ys[:, None]
    assert (seen == 1).all()
    full = deinterleave_numpy(gathered, H, band)
    assert np.array_equal(full[:, 0], np.arange(H))


def test_scenes_are_deterministic_and_sized():
    v1, i1, m1 = scenes.atrium_mesh(detail=0.05)
    v2, i2, m2 = scenes.atrium_mesh(detail=0.05)
    assert v1.dtype == np.float32 and i1.dtype == np.int32
    assert np.array_equal(v1, v2) and np.array_equal(i1, i2) and np.array_equal(m1, m2)
    assert i1.max() < len(v1) and i1.min() >= 0 and len(i1) % 3 == 0 and len(m1) == len(i1) // 3
    assert m1.max() < scenes.ATRIUM_MATERIALS
    from raytracer_amd import workloads
    wl = workloads.sponza()
    assert 255_000 <= wl.n_tris <= 270_000
    vi, ii, mi = scenes.instance_grid(v1, i1, m1, 2, 3, 15.0, 23.0)
    assert len(ii) == 6 * len(i1) and ii.max() == len(vi) - 1


def test_camera_pitch_at_800_is_the_reference_literal():
    assert np.float32(rt.camera_for_width(800).pitch).view(np.uint32) == np.float32(0.01).view(np.uint32)


# ------------------------------------------------------------------ tolerant OBJ + MTL loader (beyond the reference)
def _dump_relaxed(path):
    out = subprocess.run([HOST_BIN, "--dump-obj-relaxed", path], check=True, capture_output=True, text=True).stdout.splitlines()
    head = dict(zip(out[0].split()[0::2], map(int, out[0].split()[1::2])))
    v = np.array([[float(x) for x in l.split()[1:]] for l in out if l.startswith("v ")], np.float32).reshape(-1, 3)
    i = [int(l.split()[1]) for l in out if l.startswith("i ")]
    t = [int(l.split()[1]) for l in out if l.startswith("t ")]
    m = []
    for l in out:
        if l.startswith("m "):
            name, vals = l[2:].split("|")
            m.append((name, [float(x) for x in vals.split()]))
    return head, v, i, t, m


RELAXED_OBJ = (
    "# a comment\r\n"
    "mtllib  a.mtl   b.mtl\r\n"
    "v  0 0 0\r\n"
    "v\t1   0 0\r\n"
    "v 1 1 0\r\n"
    "v 0 1 0  # trailing comment\r\n"
    "v 0.5 1.5 0\r\n"
    "vn 0 0 1\r\n"
    "vt 0.5 0.5\r\n"
    "usemtl red\r\n"
    "f 1/1/1  2/1/1   3/1/1\r\n"            # runs of spaces, full corner syntax
    "usemtl mirror\r\n"
    "f -5 -4 -3 -2 -1\r\n"                  # relative indices, five corners -> 3 triangles
    "usemtl nosuch\r\n"
    "f 1//1 3//1 6//1\r\n"                  # forward reference to vertex 6, defined below
    "f 1 2 99\r\n"                          # out of range -> dropped
    "f 1 2\r\n"                             # two corners -> dropped
    "v 2 2 2\r\n"
)
A_MTL = "newmtl red\nKd 0.8 0.1 0.1\nKs 0.5 0.25 0.1\nillum 2\n"
B_MTL = "# second library\r\nnewmtl mirror\r\nKd 1 1 1\r\nKs 0.9 0.9 0.9\r\nillum 3\r\nnewmtl glass\r\nillum 5\r\nreflect 0.25\r\n"


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_relaxed_loader_handles_what_the_reference_parser_cannot():
    with tempfile.TemporaryDirectory() as td:
        for name, text in (("m.obj", RELAXED_OBJ), ("a.mtl", A_MTL), ("b.mtl", B_MTL)):
            with open(os.path.join(td, name), "wb") as f:
                f.write(text.encode())
        head, v, i, t, m = _dump_relaxed(os.path.join(td, "m.obj"))
    assert head["verts"] == 6 and head["normals"] == 1 and head["texcoords"] == 1
    assert np.array_equal(v[1], [1, 0, 0]) and np.array_equal(v[5], [2, 2, 2])
    assert i == [0, 1, 2,  0, 1, 2, 0, 2, 3, 0, 3, 4,  0, 2, 5]
    assert head["dropped"] == 2 and head["missing"] == 1
    names = [n for n, _ in m]
    assert names == ["", "red", "mirror", "glass"]
    assert t == [1, 2, 2, 2, 0]                                    # unknown material falls back to the default
    assert m[0][1] == [0.0, 1.0, 1.0, 1.0, 1.0, 1.0]               # Material(), main.cpp:34
    assert np.allclose(m[1][1], [0.0, 1.0, 0.5, 0.8, 0.1, 0.1])    # illum 2: no reflection, spec = max Ks
    assert np.allclose(m[2][1], [0.9, 1.0, 0.9, 1.0, 1.0, 1.0])    # illum 3: reflect = max Ks
    assert np.allclose(m[3][1], [0.25, 1.0, 1.0, 1.0, 1.0, 1.0])   # explicit `reflect` wins


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_relaxed_loader_equals_the_strict_one_on_a_clean_file():
    v, q = scenes.cube_quads()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "cube.obj")
        scenes.write_obj(p, v, q)
        head, sv, si = _dump(p)
        rhead, rv, ri, rt_, rm = _dump_relaxed(p)
    assert np.array_equal(rv.reshape(-1), sv) and ri == si
    assert rt_ == [0] * (len(si) // 3) and rhead["dropped"] == 0 and len(rm) == 1


@pytest.mark.skipif(not os.path.exists(HOST_BIN), reason="host binary not built")
def test_relaxed_loader_keeps_the_normal_and_texcoord_index_of_every_corner():
    """`f v/vt/vn` (the reference keeps only v, objloader.cpp:75-78): per triangle corner the 0-based vn
    and vt index, -1 where a corner names none, relative (negative) and forward references resolved,
    an index beyond the file turned into -1; fan-split polygons carry their corners' indices along."""
    text = ("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\n"
            "vn 0 0 1\nvn 0 1 0\nvt 0 0\nvt 1 1\n"
            "f 1/1/1 2/2/2 3/1/1 4/2/2\n"          # quad -> (0,1,2) (0,2,3)
            "f 1//-1 2//-2 3//3\n"                 # relative normals; normal 3 is defined below
            "f 1/2 2/9 3\n"                        # texcoord only; 9 does not exist; bare corner
            "vn 1 0 0\n")
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "n.obj")
        open(p, "w").write(text)
        out = subprocess.run([HOST_BIN, "--dump-obj-relaxed", p], check=True, capture_output=True, text=True).stdout.splitlines()
    pairs = [tuple(int(x) for x in l.split()[1:]) for l in out if l.startswith("n ")]
    assert pairs == [(0, 0), (1, 1), (0, 0),  (0, 0), (0, 0), (1, 1),
                     (1, -1), (0, -1), (2, -1),
                     (-1, 1), (-1, -1), (-1, -1)]


# ------------------------------------------------------------------ OBJ loader behind its C ABI, bench workloads
def test_objhost_library_exports_every_declared_symbol_and_loads_a_mesh(tmp_path):
    from raytracer_amd import workloads
    header = open(os.path.join(ROOT, "include", "rt_objhost.h")).read()
    declared = set(re.findall(r"\b(rt_obj_[a-z_]+)\s*\(", header))
    L = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "libobjhost.so"))
    assert declared == {"rt_obj_load", "rt_obj_free", "rt_obj_counts", "rt_obj_copy"}
    for s in declared:
        assert hasattr(L, s), s
    v, q = scenes.cube_quads()
    path = str(tmp_path / "cube_quads.obj")
    scenes.write_obj(path, v, q)
    for relaxed in (False, True):
        lv, li, tm, mats, dropped = workloads.load_obj(path, relaxed=relaxed)
        assert dropped == 0 and np.array_equal(lv, v) and np.array_equal(li, scenes.cube_mesh()[1])    # quads fan-split, objloader.cpp:90-99
        assert len(tm) == 12 and mats.shape == (1, 6) and tuple(mats[0]) == (0.0, 1.0, 1.0, 1.0, 1.0, 1.0)   # Material(), main.cpp:34
    # a missing file is an empty mesh (objloader.cpp:22,104), not an error
    lv, li, tm, mats, dropped = workloads.load_obj(str(tmp_path / "absent.obj"))
    assert len(lv) == 0 and len(li) == 0


def test_sponza_workload_takes_a_supplied_obj_and_falls_back_to_the_stand_in(tmp_path, monkeypatch):
    from raytracer_amd import workloads
    v, i, m = scenes.atrium_mesh(detail=0.002)
    # a mesh lying along x, in other units: it must come out along z inside the courtyard box
    big = np.stack([v[:, 2] * 100.0, v[:, 1] * 100.0 + 3.0, v[:, 0] * 100.0], 1)
    path = str(tmp_path / "sponza.obj")
    scenes.write_obj(path, big, i.reshape(-1, 3))
    wl = workloads.sponza(W=64, H=36, obj=path)
    assert wl.name.startswith("sponza.obj(") and wl.n_tris == len(i) // 3
    lo, hi = wl.verts.min(0), wl.verts.max(0)
    assert abs(lo[1] + 4.0) < 1e-4 and abs(lo[2] + 4.0) < 1e-4 and abs(hi[2] - 18.0) < 1e-3 and abs(lo[0] + hi[0]) < 1e-3
    assert all(mm.reflect == 0.5 for mm in wl.mats)           # no .mtl: the declared mesh-reflection extension
    monkeypatch.setenv("RT_SPONZA_OBJ", path)
    assert workloads.sponza(W=64, H=36).name.startswith("sponza.obj(")
    monkeypatch.setenv("RT_SPONZA_OBJ", str(tmp_path / "nothing.obj"))
    assert workloads.sponza(W=64, H=36, detail=0.002).name.startswith("sponza-standin(")


def test_second_stand_in_has_the_irregularities_it_is_for():
    v, i, m = scenes.atrium2_mesh()
    assert 250_000 < len(i) // 3 < 300_000
    t = v[i.reshape(-1, 3)].astype(np.float64)
    e = np.stack([np.linalg.norm(t[:, 1] - t[:, 0], axis=1), np.linalg.norm(t[:, 2] - t[:, 1], axis=1),
                  np.linalg.norm(t[:, 0] - t[:, 2], axis=1)], 1)
    area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)
    assert ((e.max(1) ** 2 / (2 * area + 1e-30)) > 100).sum() > 1000        # slivers of aspect > 100:1
    assert area.max() / np.median(area) > 1e5                                # two-triangle walls beside dense ornaments
    v2, i2, m2 = scenes.atrium2_mesh()
    assert np.array_equal(v, v2) and np.array_equal(i, i2)                  # deterministic


# ------------------------------------------------------------------ guarded build covers every device-derived index
def test_every_device_derived_index_of_the_queue_kernels_is_guarded():
    """In the staged pipeline's kernels, an index that comes out of device memory (queue positions from region
    counts, node and triangle references, pixels carried by rays, stack spill slots) must pass through RTG(),
    which a -DRT_GUARD=1 build turns into a range check.  This scans the sources for the array accesses that
    take such indices and fails on one that bypasses the macro."""
    src = open(os.path.join(ROOT, "raytracer_amd", "csrc", "wavefront.hip")).read()
    body = src[src.index("void k_trace(TraceArgs A)"):src.index("int wavefront_trace_grid")]
    guarded_arrays = {
        r"A\.hits\[": "pixel", r"A\.occ\[": "pixel", r"A\.q_o\[|A\.q_d\[": "queue", r"A\.nq_o\[|A\.nq_d\[": "queue",
        r"A\.sq_o\[|A\.sq_d\[": "queue",
    }
    bad = []
    for line_no, line in enumerate(body.splitlines(), 1):
        code = line.split("//")[0]
        for pat in guarded_arrays:
            for mm in re.finditer(pat, code):
                rest = code[mm.end():]
                idx = rest[:rest.index("]")] if "]" in rest else rest
                # the index is either an RTG(...) expression or a local that was assigned from one on an earlier line
                if "RTG(" in idx or re.fullmatch(r"\s*(q|dst|r\.pixel)\s*", idx):
                    continue
                bad.append((line_no, line.strip()))
    assert not bad, bad
    # the locals accepted above are themselves produced by RTG
    assert re.search(r"const size_t q = \(size_t\)RTG\(", body) and re.search(r"const size_t dst = \(size_t\)RTG\(", body)
    assert "r.pixel = (int)RTG(r.pixel, A.n_pixels)" in body
    # node, triangle and spill-slot references
    assert "RTG(node, A.n_wnodes)" in body and "RTG(tri_i, sc.n_refs)" in body and body.count("RTG(sp - WF_LDS_STACK, WF_OVF)") == 2
    assert os.path.exists(os.path.join(ROOT, "raytracer_amd", "variants", "guard.so")), "build() must also produce the guarded variant"


def test_bench_reports_counter_traffic_only_for_the_sources_and_frame_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed counter run; it is only valid for the kernel sources, workload and
    resolution of that run.  bench.measured_traffic must say null (with the reason) for anything else."""
    import importlib.util
    import json
    from raytracer_amd.build import source_hash
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    (tmp_path / "profiles").mkdir()
    rec = {"source_hash": source_hash(), "sponza": {"resolution": [1920, 1080], "bytes_per_frame": 123.0, "bytes_per_frame_as_counted": 77.0, "source": "x.txt"}}
    (tmp_path / "profiles" / "hbm_traffic.json").write_text(json.dumps(rec))
    valu = {"source_hash": source_hash(), "sponza": {"resolution": [1920, 1080], "source": "x", "kernels": {"k_trace<2, false, 0>": {"active_lanes": 40.0}}}}
    (tmp_path / "profiles" / "valu_counters.json").write_text(json.dumps(valu))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_traffic("sponza", 1, (1920, 1080))[0] == (123.0, 77.0)     # FETCH_SIZE doubled, and as counted
    assert bench.measured_traffic("sponza", 1, (3840, 2160))[0] == (None, None)      # another frame size
    assert bench.measured_traffic("soup", 1, (1920, 1080))[0] == (None, None)        # another workload
    assert bench.measured_traffic("sponza", 2, (1920, 1080))[0] == (None, None)      # a partitioned run
    assert bench.measured_valu("sponza", 1, (1920, 1080))[0]["k_trace<2, false, 0>"]["active_lanes"] == 40.0
    assert bench.measured_valu("sponza", 1, (1280, 720))[0] is None
    rec["source_hash"] = "0" * 16
    (tmp_path / "profiles" / "hbm_traffic.json").write_text(json.dumps(rec))
    valu["source_hash"] = "0" * 16
    (tmp_path / "profiles" / "valu_counters.json").write_text(json.dumps(valu))
    got, why = bench.measured_traffic("sponza", 1, (1920, 1080))
    assert got == (None, None) and "stale" in why
    got, why = bench.measured_valu("sponza", 1, (1920, 1080))
    assert got is None and "stale" in why
    # the committed file itself is either valid for the committed sources or reported as stale, never silently wrong
    committed = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert "source_hash" in committed and "sponza" in committed and committed["sponza"]["resolution"] == [1920, 1080]


def test_counter_derivation_reads_the_newest_collection_and_only_the_plain_traversal_kernels(tmp_path, capsys):
    """profiles/derive.py on a made-up collection directory: frames are the shading launches minus the full counting frames,
    only the <.., 0> traversal kernels are summed, a camera-ray kernel that ran a handful of times (the timing runs of the
    kernel choice) is left out, and of two collections merged into one directory only the newer counts."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("derive", os.path.join(ROOT, "profiles", "derive.py"))
    derive = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(derive)
    d = tmp_path / "prof"
    (d / "trace" / "vm").mkdir(parents=True)
    (d / "pmc1" / "vm").mkdir(parents=True)
    ns = "void (anonymous namespace)::"

    def stats(path, scale):
        rows = [(ns + "k_trace<2, false, 0>(TraceArgs)", 40, 400000 * scale), (ns + "k_trace<2, false, 1>(TraceArgs)", 4, 1300000),
                (ns + "k_trace<1, false, 0>(TraceArgs)", 10, 200000 * scale), (ns + "k_trace_packet<0>(TraceArgs)", 12, 150000 * scale),
                (ns + "k_trace_packet<1>(TraceArgs)", 1, 500000), (ns + "k_trace<0, true, 0>(TraceArgs)", 3, 200000),
                (ns + "k_trace<2, false, 2>(TraceArgs)", 4, 480000), (ns + "k_shade_all<5>(ShadeAllArgs)", 11, 125000)]
        with open(path, "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
            for name, calls, avg in rows:
                f.write('"%s",%d,%d,%d,1.0\n' % (name, calls, calls * avg, avg))

    stats(d / "trace" / "vm" / "1_kernel_stats.csv", 7)          # an older collection in the same directory
    time.sleep(0.02)
    stats(d / "trace" / "vm" / "2_kernel_stats.csv", 1)
    os.utime(d / "trace" / "vm" / "1_kernel_stats.csv", (1, 1))
    with open(d / "pmc1" / "vm" / "9_counter_collection.csv", "w") as f:
        f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
        for i in range(40):
            f.write('%d,"%sk_trace<2, false, 0>(TraceArgs)","SQ_INSTS_VALU",%d\n' % (i + 1, ns, 200000000))
    derive.main([str(d), "sponza", "1920", "1080"])
    out = capsys.readouterr().out
    assert "frames with plain kernels in the trace pass: 10 (of 11 frames)" in out
    assert "traversal per frame: 1.950 ms" in out          # 4 x 400 + 200 + 150 us: the newer file, plain kernels only
    assert "k_trace<0, true, 0>" in out and "timing runs" in out
    assert "k_trace<2, false, 1>" not in out and "k_trace<2, false, 2>" not in out


def test_obj_loader_survives_made_up_files_under_the_sanitizers(tmp_path):
    """tests/objfuzz.cpp: 3000 made-up OBJ files (with a made-up .mtl beside some of them) through both tokenizers of the
    host's loader, compiled with -fsanitize=address,undefined: no crash, no report, and no index handed out that points
    outside what was loaded.  (The reference's parser asserts or reads out of bounds on such input, objloader.cpp:83.)"""
    exe = str(tmp_path / "objfuzz")
    host = os.path.join(ROOT, "raytracer_amd", "host")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "objfuzz.cpp"),
                        os.path.join(host, "objhost_c.cpp"), os.path.join(host, "objloader.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower():
        pytest.skip("no sanitizer runtime for g++ here")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "verify_asan_link_order=0"      # (another preloaded library may come before the sanitizer's runtime)
    r = subprocess.run([exe, str(tmp_path), "3000"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert "no crash, no index out of range" in r.stdout

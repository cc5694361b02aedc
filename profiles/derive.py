"""profiles/derive.py <collect dir> [workload [W H]] [--write]: what the counters of a profiles/collect.sh run say about the
traversal kernels, per kernel and per frame:
  * VALU issue: SIMD-cycles per VALU instruction (1024 SIMDs x duration x 2.4 GHz / SQ_INSTS_VALU) and the share of the
    issue slots that is at the 4.1 cycles per instruction most of the node step's instructions take
    (profiles/valu_ceiling.txt);
  * mean active lanes per VALU instruction: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU;
  * the share of VALU instructions that are f32 FMA / MUL / ADD;
  * fabric-side traffic: FETCH_SIZE + WRITE_SIZE (KiB), as counted and with FETCH_SIZE doubled.  The doubling is the
    gfx950 correction calibrated on wide coalesced streaming reads (128-byte requests tallied at 64 bytes); the
    traversal fetches scattered 64-byte nodes and 48-byte triangles, for which it is NOT calibrated: the true figure
    lies between the two.  Both include Infinity-Cache hits.
With --write the figures go to profiles/valu_counters.json and profiles/hbm_traffic.json, keyed by the hash of the kernel
sources, which is how bench.py knows whether they describe the kernels it is running."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.build import source_hash  # noqa: E402

N_SIMD, CLOCK_GHZ, CYCLES_4 = 1024, 2.4, 4.1


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def newest(files):
    # a directory merged from several collections holds one file per collection: only the last one counts
    return sorted(files, key=os.path.getmtime)[-1:]


def load(d):
    times = {}
    for f in newest(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            times[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
    counters = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
        if not os.path.isdir(p):
            continue
        for f in newest(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    per_launch = {k: {c: v / max(len(launches[k][c]), 1) for c, v in cs.items()} for k, cs in counters.items()}
    return times, per_launch


def is_plain_traversal(k):
    # the non-counting variants: k_trace<mode, primary, 0>, k_trace_packet<0>
    return (k.startswith("k_trace<") and k.endswith(", 0>")) or k.startswith("k_trace_packet<0>")


def main(argv):
    write = "--write" in argv
    argv = [a for a in argv if a != "--write"]
    d = argv[0]
    workload = argv[1] if len(argv) > 1 else "sponza"
    W, H = (int(argv[2]), int(argv[3])) if len(argv) > 3 else (1920, 1080)
    times, pl = load(d)
    # frames = launches of the shading kernel; the counting frames among them (bench.py's own counting frame) run the
    # counting variants <.., 1> of the camera-ray kernels (the pass that makes the claim order, <.., 2>, does not shade)
    shade_calls = sum(c for k, (c, _) in times.items() if k.startswith("k_shade_all"))
    counting = sum(c for k, (c, _) in times.items() if (k.startswith("k_trace_packet<1>") or k.startswith("k_trace<0, true, 1>")))
    frames = max(shade_calls - counting, 1)
    print("frames with plain kernels in the trace pass: %d (of %d frames)" % (frames, shade_calls))
    kernels = {}
    fetch = wr = 0.0
    print("%-34s %6s %9s %12s %10s %8s %9s" % ("kernel", "per fr", "avg us", "cyc/VALUinst", "issue@4.1", "lanes", "fma share"))
    for k in sorted(times):
        if not is_plain_traversal(k):
            continue
        calls, avg_ns = times[k]
        if 2 * calls < frames:
            print("%-34s %d launches: the timing runs of the camera-ray kernel choice, not part of a frame" % (k, calls))
            continue
        if k.startswith("k_trace_packet") or k.startswith("k_trace<0, true"):
            calls = min(calls, frames)          # one camera-ray launch per frame; any more are the timing runs of the kernel choice
        c = pl.get(k, {})
        insts = c.get("SQ_INSTS_VALU", 0.0)
        rec = {"launches_per_frame": calls / frames, "avg_us": avg_ns / 1e3}
        if insts:
            cyc = N_SIMD * avg_ns * CLOCK_GHZ / insts
            rec["simd_cycles_per_valu_inst"] = cyc
            rec["issue_share_at_4.1_cycles"] = min(1.0, CYCLES_4 / cyc)
        if c.get("SQ_ACTIVE_INST_VALU"):
            rec["active_lanes"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"]
        if insts:
            rec["fma_mul_add_share"] = (c.get("SQ_INSTS_VALU_FMA_F32", 0.0) + c.get("SQ_INSTS_VALU_MUL_F32", 0.0) + c.get("SQ_INSTS_VALU_ADD_F32", 0.0)) / insts
        kernels[k] = rec
        fetch += c.get("FETCH_SIZE", 0.0) * calls / frames
        wr += c.get("WRITE_SIZE", 0.0) * calls / frames
        print("%-34s %6.2f %9.1f %12.2f %10.2f %8.1f %9.3f" % (k, rec["launches_per_frame"], rec["avg_us"], rec.get("simd_cycles_per_valu_inst", 0),
                                                         rec.get("issue_share_at_4.1_cycles", 0), rec.get("active_lanes", 0), rec.get("fma_mul_add_share", 0)))
    trav_ms = sum(r["launches_per_frame"] * r["avg_us"] for r in kernels.values()) / 1e3
    print("traversal per frame: %.3f ms; fabric-side traffic per frame: %.2f GB as counted, %.2f GB with FETCH_SIZE doubled" % (
        trav_ms, (fetch + wr) * 1024 / 1e9, (2 * fetch + wr) * 1024 / 1e9))
    if write:
        src = os.path.basename(os.path.normpath(d))
        h = source_hash()
        json.dump({"source_hash": h, workload: {"resolution": [W, H], "source": src, "traversal_ms_per_frame": trav_ms, "kernels": kernels}},
                  open(os.path.join(ROOT, "profiles", "valu_counters.json"), "w"), indent=1)
        json.dump({"source_hash": h, "source_hash_covers": "the traversal kernels and the hierarchy builder only (raytracer_amd/build.py TRAVERSAL_SOURCES)",
                   workload: {"resolution": [W, H], "bytes_per_frame": (2 * fetch + wr) * 1024, "bytes_per_frame_as_counted": (fetch + wr) * 1024,
                              "fetch_KiB_raw_per_frame": fetch, "write_KiB_per_frame": wr, "kernel": "k_trace + k_trace_packet, all launches of one frame",
                              "source": src,
                              "note": "bytes_per_frame = 2 x FETCH_SIZE + WRITE_SIZE: the x2 is the gfx950 request-size correction calibrated on wide "
                                      "coalesced streaming reads; the traversal's scattered 64-byte node and 48-byte triangle fetches are not what it "
                                      "was calibrated on, so the truth lies between bytes_per_frame_as_counted and bytes_per_frame; both include "
                                      "Infinity-Cache hits"}},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
        print("wrote profiles/valu_counters.json and profiles/hbm_traffic.json for source hash", h)


if __name__ == "__main__":
    main(sys.argv[1:])

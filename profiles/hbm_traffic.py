"""profiles/hbm_traffic.py <summary.txt> <frames> [workload [W H]]: HBM-side bytes per frame of the traversal kernel
from the FETCH_SIZE / WRITE_SIZE passes of a profiles/collect.sh summary (counter unit: KiB).
FETCH_SIZE is doubled: on gfx950 it tallies 128-byte requests at 64 bytes (calibrated on wide
streaming reads; the traversal's 16-32 byte gathers are uncalibrated, so treat the figure as
an estimate).  Only the non-counting k_trace variants (<..., false>) are summed."""
import json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd.build import source_hash

def main(path, frames, workload="sponza", W=1920, H=1080):
    fetch = write = 0.0
    for line in open(path):
        m = re.match(r"\s+(k_trace<[^>]*false>)\s+launches\s+(\d+)\s+(.*)", line)
        if not m:
            continue
        n = int(m.group(2))
        for c, v in re.findall(r"(\w+)=([0-9.e+]+)", m.group(3)):
            if c == "FETCH_SIZE": fetch += float(v) * n
            if c == "WRITE_SIZE": write += float(v) * n
    per_frame = (2.0 * fetch + write) * 1024.0 / frames
    out = {"source_hash": source_hash(),
           workload: {"resolution": [int(W), int(H)], "bytes_per_frame": per_frame, "fetch_KiB_raw_per_frame": fetch / frames, "write_KiB_per_frame": write / frames,
                      "kernel": "k_trace, all launches of one frame", "source": path.split("/")[-1],
                      "note": "FETCH_SIZE x2 (gfx950 request-size correction) + WRITE_SIZE; includes Infinity-Cache hits"}}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), *(sys.argv[3:6]))

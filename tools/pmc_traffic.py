"""Aggregate rocprofv3 --pmc counter_collection.csv files into per-kernel HBM traffic per launch.

usage: pmc_traffic.py FETCH_dir WRITE_dir out.json
FETCH_SIZE / WRITE_SIZE are reported in KiB... checked against a known byte count below; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
                a = acc[name]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    n = fetch.get(k, write.get(k))[0]
    fk = fetch[k][1] / max(fetch[k][0], 1) if k in fetch else None
    wk = write[k][1] / max(write[k][0], 1) if k in write else None
    out[k] = {"launches": n, "FETCH_SIZE_raw_per_launch": fk, "WRITE_SIZE_raw_per_launch": wk,
              # raw unit is KiB; x2 gfx950 correction on the read side
              "hbm_read_bytes_per_launch": None if fk is None else fk * 1024 * 2,
              "hbm_write_bytes_per_launch": None if wk is None else wk * 1024}
    r = out[k]
    if r["hbm_read_bytes_per_launch"] is not None and r["hbm_write_bytes_per_launch"] is not None:
        r["hbm_bytes_per_launch"] = r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]
# frames the profiled command processed = launches of the per-frame blend kernel (bench.py: step_model.hbm_util)
out["_frames"] = sum(v["launches"] for k, v in out.items() if k.startswith("blend_kernel"))
import subprocess
try:
    out["_git_sha"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True,
                                     cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or os.environ.get("IRM_GIT_SHA")
except OSError:
    out["_git_sha"] = os.environ.get("IRM_GIT_SHA")
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if not k.startswith('_')), key=lambda kv: -(kv[1].get("hbm_bytes_per_launch") or 0) * kv[1]["launches"])[:14]:
    print(f"{v['launches']:6d} {(v.get('hbm_read_bytes_per_launch') or 0)/1e6:10.1f} MB rd {(v.get('hbm_write_bytes_per_launch') or 0)/1e6:10.1f} MB wr  {k[:80]}")

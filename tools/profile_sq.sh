#!/bin/bash
# SQ counters of the headline command, two separate --pmc passes (no tracing domains besides the kernel trace), run ON
# THE GPU BOX from the repo root:  tools/profile_sq.sh NAME  -> gpurun_out/profiles/NAME_sq_counters.txt
# The table prints per-launch averages and the derived ratios the DESIGN quotes.
set -e -o pipefail
NAME=$1
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles/$(dirname $NAME); mkdir -p $OUT
BASE=$(basename $NAME)
export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline"
cd /tmp
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/sq1 -- $CMD > /dev/null 2> /tmp/sq1.err
echo "pass 1 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/sq2 -- $CMD > /dev/null 2> /tmp/sq2.err
echo "pass 2 done"
cd $ROOT
python3 - "$OUT/${BASE}_sq_counters.txt" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("/tmp/sq1", "/tmp/sq2"):
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if name.startswith(("at::", "__amd", "rocblas")):
                continue
            a = acc[name][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[1], "w") as f:
    f.write("per-launch averages (GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = gui_cycles / 8); ratios: mfma = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) = share of the SIMD cycles with a matrix instruction executing, "
            "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, lds_conf = "
            "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n")
    for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0.0])[1]):
        g = {c: v / n for c, (n, v) in cs.items()}
        n = max(x[0] for x in cs.values())
        gui = g.get("GRBM_GUI_ACTIVE", 0.0)
        mf = g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * gui / 8.0) if gui else float("nan")
        wc = g.get("SQ_WAVE_CYCLES", 0.0)
        f.write(f"{k[:90]:90s} n={n:5d} gui_cycles={gui:12.0f} mfma={mf:6.3f} wait={g.get('SQ_WAIT_ANY', 0) / wc if wc else 0:6.3f} "
                f"valu={g.get('SQ_ACTIVE_INST_VALU', 0) / wc if wc else 0:6.3f} lds_conf="
                f"{g.get('SQ_LDS_BANK_CONFLICT', 0) / g['SQ_LDS_IDX_ACTIVE'] if g.get('SQ_LDS_IDX_ACTIVE') else 0:6.3f} "
                f"insts_mfma={g.get('SQ_INSTS_MFMA', 0):12.0f} insts_valu={g.get('SQ_INSTS_VALU', 0):12.0f}\n")
print(open(sys.argv[1]).read()[:3000])
PY

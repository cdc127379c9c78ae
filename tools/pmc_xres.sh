# SQ counters of the input-resident emulated GEMM (M510 K96 512x512 B6)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_x1 -o a -- python tools/bench_split.py > gpurun_out/pmc_x1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_x2 -o b -- python tools/bench_split.py > gpurun_out/pmc_x2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_x1", "gpurun_out/pmc_x2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "gemm_xres" in r["Kernel_Name"] and r["Grid_Size"] in ("2211840",):   # 6144 blocks x 360? filter below
                acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            elif "gemm_xres" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:50] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k)
        for c, vals in v.items():
            print(f"   {c:28s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
PY

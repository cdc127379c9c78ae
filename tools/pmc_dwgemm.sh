# SQ counters of the fused kernel on its largest shape (one pass = at most 8 SQ counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export IRM_BENCH_ONE=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_dw1 -o a -- python tools/bench_dwgemm.py > gpurun_out/pmc_dw1.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_dw2 -o b -- python tools/bench_dwgemm.py > gpurun_out/pmc_dw2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_dw1", "gpurun_out/pmc_dw2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "dwgemm" in r["Kernel_Name"] or "dwconv" in r["Kernel_Name"] or "gemm_ring" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k)
        for c, vals in v.items():
            print(f"   {c:28s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
PY

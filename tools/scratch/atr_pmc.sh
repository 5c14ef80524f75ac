export TMPDIR=/tmp
rm -rf gpurun_out/atr_pmc; mkdir -p gpurun_out/atr_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/atr_pmc -o p -- python tools/scratch/atr_ab.py > gpurun_out/atr_pmc/log.txt 2>&1
F=$(ls gpurun_out/atr_pmc/*/p_counter_collection.csv gpurun_out/atr_pmc/p_counter_collection.csv 2>/dev/null | head -1)
python - "$F" <<'PY' > gpurun_out/atr_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_bf16_nt_kernel" not in k: continue
    if int(r["Grid_Size"]) < 100000: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
for k, v in acc.items():
    print(k[:90], n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()}, "conflict share", round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), 3))
PY
rm -rf gpurun_out/atr_pmc

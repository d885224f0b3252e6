# dev: SQ counters of single bf16 GEMM shapes (rocprofv3 PMC, two passes): bash tools/pmc_gemm16.sh "M N K flags" ... -> stdout
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd $R
for shp in "$@"; do
  for pass in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc1
    timeout -k 10 200 rocprofv3 --pmc $pass -d /tmp/pmc1 -o x --output-format csv -- python3 tools/one_gemm16.py $shp > /tmp/pmc1.log 2>&1
    F=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
    python3 - "$F" "$shp" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    print(sys.argv[2], k, " ".join(f"{a}={v / n[(k, a)]:.4g}" for a, v in c.items()))
PY
  done
done

# SQ counters of ONE fp32 flash-attention shape (two passes: issue / wait counters, instruction mix).  GPU box: bash tools/pmc_attn.sh s2 both <tag>
S=${1:-s2}; W=${2:-both}; TAG=${3:-attn}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
for pass in A B; do
  if [ $pass = A ]; then CNT="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"
  else CNT="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; fi
  rm -rf gpurun_out/pmc_${TAG}_$pass
  timeout -k 10 200 rocprofv3 --pmc $CNT -d $R/gpurun_out/pmc_${TAG}_$pass -o s --output-format csv -- python3 tools/bench_one_attn_f32.py $S $W > gpurun_out/pmc_${TAG}_$pass.log 2>&1 || { tail -5 gpurun_out/pmc_${TAG}_$pass.log; continue; }
  F=$(find gpurun_out/pmc_${TAG}_$pass -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    if "flash" not in k: continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in tot.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / n[(k, name)]:16.0f} per launch")
PY
  rm -rf gpurun_out/pmc_${TAG}_$pass
done

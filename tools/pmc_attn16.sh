# dev: SQ counters of the bf16 window-attention kernels at one stage (two passes).  GPU box: bash tools/pmc_attn16.sh s2
S=${1:-s2}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
for pass in A B; do
  if [ $pass = A ]; then CNT="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"
  else CNT="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"; fi
  rm -rf /tmp/pmc_a16
  timeout -k 10 200 rocprofv3 --pmc $CNT -d /tmp/pmc_a16 -o s --output-format csv -- python3 tools/bench_one_attn.py $S > /tmp/pmc_a16.log 2>&1 || { tail -5 /tmp/pmc_a16.log; continue; }
  F=$(find /tmp/pmc_a16 -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    if "attn" not in k: continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in tot.items():
    print(k, " ".join(f"{name}={v / n[(k, name)]:.4g}" for name, v in sorted(c.items())))
PY
done

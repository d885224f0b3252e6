# SQ counters of the split-product kernels (GEMM NT / TN, window attention) at the bench shapes, two PMC passes (8 SQ slots each); no tracing options with --pmc.
# GPU box: bash tools/pmc_split.sh [tag] [kernel filter for tools/run_split_kernels.py ...] -> gpurun_out/<tag>_split_sq_counters.txt
TAG=${1:-r06}; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
OUT=gpurun_out/${TAG}_split_sq_counters.txt
: > $OUT
for pass in A B; do
  if [ $pass = A ]; then CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
  else CNT="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; fi
  rm -rf gpurun_out/pmc_split_$pass
  timeout -k 10 300 rocprofv3 --pmc $CNT -d $R/gpurun_out/pmc_split_$pass -o s --output-format csv -- python3 tools/run_split_kernels.py "$@" > gpurun_out/pmc_split_$pass.log 2>&1 || { echo "pass $pass failed" >> $OUT; tail -5 gpurun_out/pmc_split_$pass.log >> $OUT; continue; }
  F=$(find gpurun_out/pmc_split_$pass -name "*counter_collection.csv" | head -1)
  python3 - "$F" $pass >> $OUT <<'PY'
import csv, sys, collections
tot = collections.defaultdict(collections.Counter); n = collections.Counter(); grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    if "split" not in k or k.startswith("split3_kernel"): continue
    k = f"{k} grid={r.get('Grid_Size', '?')}"
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print(f"== pass {sys.argv[2]} (per launch)")
for k, c in sorted(tot.items()):
    print(k)
    v = {name: c[name] / n[(k, name)] for name in c}
    for name in sorted(v): print(f"   {name:28s} {v[name]:18.0f}")
    if "SQ_BUSY_CYCLES" in v:
        wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
        print(f"   -> mfma_busy/busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(v['SQ_BUSY_CYCLES'], 1):.3f}  wait_any/wave {v.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst/wave {v.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  "
              f"active/wave {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  wait_lds/wave {v.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}")
    if "SQ_LDS_IDX_ACTIVE" in v:
        print(f"   -> lds bank-conflict cycles / lds active cycles {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v['SQ_LDS_IDX_ACTIVE'], 1):.3f}")
PY
  rm -rf gpurun_out/pmc_split_$pass
done
cat $OUT

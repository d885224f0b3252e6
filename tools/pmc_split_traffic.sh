# HBM fetch / write bytes per launch of the split-product kernels at the bench shapes (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes; gfx950: FETCH_SIZE x 2,
# both in KiB).  GPU box: [GG_DEV_SWITCHES=1 GG_SPLIT3A_ABL=16] bash tools/pmc_split_traffic.sh <tag> [kernel filters of tools/run_split_kernels.py ...]
TAG=${1:-traffic}; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd $R
OUT=gpurun_out/${TAG}_split_traffic.txt
: > $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_st
  REP=3 timeout -k 10 300 rocprofv3 --pmc $c -d /tmp/pmc_st -o x --output-format csv -- python3 tools/run_split_kernels.py "$@" > /tmp/pmc_st.log 2>&1 || { echo "$c pass failed" >> $OUT; tail -3 /tmp/pmc_st.log >> $OUT; continue; }
  F=$(find /tmp/pmc_st -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$c" >> $OUT <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    if "split" not in k or k.startswith("split3_kernel"): continue
    k = f"{k} grid={r.get('Grid_Size', '?')}"
    tot[k] += float(r["Counter_Value"]); n[k] += 1
mult = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
for k in sorted(tot): print(f"{sys.argv[2]:10s} {k:84s} launches {n[k]}  {mult * tot[k] / n[k] * 1024 / 1e9:8.3f} GB/launch")
PY
done
cat $OUT

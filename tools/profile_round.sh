# Round profile: rocprofv3 kernel-trace summary + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domains) of bench.py,
# once per precision.  Run on the GPU box:  GG_GIT_HEAD=<commit> bash tools/profile_round.sh <fp32|bf16> <tag>   -> gpurun_out/<tag>_*
# (the box has no .git: pass the commit in; the JSON also records the sha256 of the kernel sources, which bench.py checks)
set -x
P=${1:-fp32}
TAG=${2:-r05_$P}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --precision $P --dump-launches gpurun_out/${TAG}_launches > gpurun_out/${TAG}_trace.log 2>&1 || exit 1
tail -1 gpurun_out/${TAG}_trace.log | cut -c1-300
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --precision $P > gpurun_out/${TAG}_fetch.log 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --precision $P > gpurun_out/${TAG}_write.log 2>&1
F=$(find gpurun_out/${TAG}_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/${TAG}_write -name "*counter_collection.csv" | head -1)
GG_PMC_COMMAND="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --precision $P" python3 tools/pmc_traffic.py $F $W 3 gpurun_out/${TAG}_hbm_traffic_pmc.json gpurun_out/${TAG}_launches.$P.json gpurun_out/${TAG}_gemm_forms_traffic.txt
cp gpurun_out/${TAG}_hbm_traffic_pmc.json gpurun_out/${TAG%%_*}_hbm_traffic_pmc_$P.json      # the name bench.py's `roofline.traffic` reads under profiles/
S=$(find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/${TAG}_kernel_stats.csv
# keep the merge-back small: drop the raw per-dispatch tables
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
ls -la gpurun_out | grep ${TAG}

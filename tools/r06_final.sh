# round-6 final evidence run, part $1: a = full GPU test suite + smoke; b = bench line; c = profile set of the headline mode; d = profile sets of the fp32 / bf16 modes
cd $GRAFT_REPO_ROOT
( while true; do sleep 50; echo "[r06_final] running"; done ) &
HB=$!
case "$1" in
a) python -m pytest tests -m gpu -q > gpurun_out/r06_gputests.log 2>&1; echo "pytest rc=$?" > gpurun_out/r06_final_a.rc
   python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_final_a.rc; cat gpurun_out/r06_final_a.rc; tail -3 gpurun_out/r06_gputests.log; tail -4 gpurun_out/r06_smoke.log ;;
b) python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r06_bench_line.json ;;
c) bash tools/profile_round.sh fp32_split r06_fp32_split > gpurun_out/r06_prof_fp32_split.log 2>&1
   bash tools/pmc_split.sh r06 > gpurun_out/r06_pmc_split.log 2>&1
   bash tools/pmc_split_traffic.sh r06 > gpurun_out/r06_pmc_split_traffic.log 2>&1 ;;
d) bash tools/profile_round.sh fp32 r06_fp32 > gpurun_out/r06_prof_fp32.log 2>&1
   bash tools/profile_round.sh bf16 r06_bf16 > gpurun_out/r06_prof_bf16.log 2>&1 ;;
esac
kill $HB

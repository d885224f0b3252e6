# round-6 profile set, part $1 (a: the headline mode's kernel stats + PMC traffic + SQ counters of the split kernels; b: fp32 and bf16 modes)
cd $GRAFT_REPO_ROOT
( while true; do sleep 50; echo "[r06_profiles] running"; done ) &
HB=$!
if [ "$1" = a ]; then
  bash tools/profile_round.sh fp32_split r06_fp32_split > gpurun_out/r06_prof_fp32_split.log 2>&1
  bash tools/pmc_split.sh r06 > gpurun_out/r06_pmc_split.log 2>&1
  bash tools/pmc_split_traffic.sh r06 > gpurun_out/r06_pmc_split_traffic.log 2>&1
else
  bash tools/profile_round.sh fp32 r06_fp32 > gpurun_out/r06_prof_fp32.log 2>&1
  bash tools/profile_round.sh bf16 r06_bf16 > gpurun_out/r06_prof_bf16.log 2>&1
fi
kill $HB
ls -la gpurun_out | grep r06_

# Round profile set (run on the GPU box): GG_GIT_HEAD=<commit> bash tools/round_profiles.sh <tag>   ->  gpurun_out/<tag>_*
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
( while true; do sleep 50; echo "[round_profiles] running"; done ) &
HB=$!
bash tools/profile_round.sh fp32 ${TAG}_fp32 > gpurun_out/${TAG}_prof_fp32.log 2>&1
bash tools/profile_round.sh bf16 ${TAG}_bf16 > gpurun_out/${TAG}_prof_bf16.log 2>&1
python tools/prof_step.py --fp32 > gpurun_out/${TAG}_fp32_per_shape.txt 2>/dev/null
python tools/prof_step.py > gpurun_out/${TAG}_bf16_per_shape.txt 2>/dev/null
bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt gpurun_out/${TAG}_fp32_sq_counters.txt
python tools/bench_secondary.py > gpurun_out/${TAG}_secondary.jsonl 2>gpurun_out/${TAG}_secondary.err
bash tools/pmc_one_gemm.sh "3211264 384 96" "3211264 96 384" "200704 1152 384" "200704 1536 384" "802816 576 192" > gpurun_out/${TAG}_gemm_shape_traffic.txt 2>&1
kill $HB
ls -la gpurun_out | grep ${TAG}

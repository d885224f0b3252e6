cd $GRAFT_REPO_ROOT
for rep in 1 2; do
 for D in 0 8; do
  echo "GG_GEMM_DEBUG=$D" >> gpurun_out/c4_nt.log
  GG_DEV_SWITCHES=1 GG_GEMM_DEBUG=$D python tools/bench_c4.py fp16 2>&1 | tail -1 >> gpurun_out/c4_nt.log
  GG_DEV_SWITCHES=1 GG_GEMM_DEBUG=$D python tools/bench_c4.py bf16 2>&1 | tail -1 >> gpurun_out/c4_nt.log
 done
done
cat gpurun_out/c4_nt.log

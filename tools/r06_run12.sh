cd $GRAFT_REPO_ROOT
for rep in 1 2; do
 for D in 0 32; do
  echo "GG_GEMM_DEBUG=$D (0 = non-temporal result stores, 32 = default policy)" >> gpurun_out/bf16_nt.log
  GG_DEV_SWITCHES=1 GG_GEMM_DEBUG=$D python bench.py --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/bf16_nt.log
 done
done
cat gpurun_out/bf16_nt.log

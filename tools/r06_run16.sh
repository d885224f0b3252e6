cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "prologue" > gpurun_out/t16.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t16.log
python tools/bench_split3_pro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/split3_pro.log
for rep in 1 2; do for V in 0 1; do
  echo "GG_SPLIT3_NO_PRO=$V" >> gpurun_out/step_pro_ab.log
  if [ $V = 1 ]; then export GG_DEV_SWITCHES=1 GG_SPLIT3_NO_PRO=1; else unset GG_DEV_SWITCHES GG_SPLIT3_NO_PRO; fi
  python bench.py --precision fp32_split --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/step_pro_ab.log
done; done
cat gpurun_out/step_pro_ab.log

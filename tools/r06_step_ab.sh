# the training step with the round-6 kernel changes switched off (dev switches) against the default, alternating on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "round-6 default" >> gpurun_out/r06_step_ab.txt
  python bench.py --precision fp32_split --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/r06_step_ab.txt
  echo "round-5 forms (GG_SPLIT3_SWZ=0 GG_SPLIT3A_ABL=256 GG_SPLIT3_NO_EC=1 GG_SPLIT3_TN=1 GG_SPLIT3_NO_PRO=1)" >> gpurun_out/r06_step_ab.txt
  GG_DEV_SWITCHES=1 GG_SPLIT3_SWZ=0 GG_SPLIT3A_ABL=256 GG_SPLIT3_NO_EC=1 GG_SPLIT3_TN=1 GG_SPLIT3_NO_PRO=1 python bench.py --precision fp32_split --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/r06_step_ab.txt
done
cat gpurun_out/r06_step_ab.txt

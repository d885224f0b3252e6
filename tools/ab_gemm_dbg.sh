cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py -q -x -k "stride2 or train_step or fp32 or f32" 2>&1 | tail -3
for sw in 1 0; do
  echo "== step, s2 fused f32 $( [ $sw = 1 ] && echo on || echo off )"
  if [ $sw = 1 ]; then timeout -k 10 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200;
  else GG_DEV_SWITCHES=1 GG_NO_FUSE_BNBWD_EPI=1 timeout -k 10 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200; fi
done

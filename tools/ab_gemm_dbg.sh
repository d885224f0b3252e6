cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_precision.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -2
for dbg in 128 0; do
  echo "== DEBUG=$dbg (128: old 64-byte-run stores)"
  GG_GEMM_F32_DEBUG=$dbg timeout -k 10 300 python tools/bench_gemm_f32.py 2>&1 | grep "gelu+pre\|s1.fc1\|s3.fc1 "
done
for dbg in 128 0; do
  echo "== step DEBUG=$dbg"
  GG_GEMM_F32_DEBUG=$dbg timeout -k 10 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_precision.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -2
for dbg in 128 0; do
  echo "== DEBUG=$dbg (128: old 64-byte-run stores and loads)"
  GG_DEV_SWITCHES=1 GG_GEMM_F32_DEBUG=$dbg timeout -k 10 300 python tools/bench_gemm_f32.py 2>&1 | grep "dgelu\| res"
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_precision.py tests/test_gpu_kernels.py -m gpu -q -k "f32_gemm or f32_mfma or fp32_mode_train or gemm_f32 or small_m or splitk or bn_gemm or prologue or two_source or fusion" > gpurun_out/t24.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t24.log
for rep in 1 2; do for V in 1 0; do
  echo "GG_GEMM_F32_SWZ=$V (1 = permuted rows, 0 = plain)"
  GG_DEV_SWITCHES=1 GG_GEMM_F32_SWZ=$V python tools/bench_gemm_f32.py 2>&1 | grep -v amdgpu | cut -c1-120
done; done

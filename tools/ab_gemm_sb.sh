# dev A/B: small-K (4-5 workgroups per CU) ring form of gg_gemm_nt_f32 against the default, on the model's shapes and in the whole step
cd $GRAFT_REPO_ROOT
for sb in 0 100000; do
  echo "== GG_GEMM_F32_SB=$sb"
  GG_GEMM_F32_SB=$sb timeout -k 10 300 python tools/bench_gemm_f32.py 2>&1 | grep -v "wgrad\|head\|4096\|pro \|2src\|amdgpu.ids"
done
for sb in 0 384 100000; do
  echo "== step GG_GEMM_F32_SB=$sb"
  GG_GEMM_F32_SB=$sb timeout -k 10 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done

cd $GRAFT_REPO_ROOT
for dbg in 0 256 2; do
  echo "== SB=100000 DEBUG=$dbg (256: stores into a 4096-row window, 2: no stores)"
  GG_GEMM_F32_SB=100000 GG_GEMM_F32_DEBUG=$dbg timeout -k 10 300 python tools/bench_gemm_f32.py 2>&1 | grep "s0.conv1 plain\|s2.qkv\|s1.qkv"
done

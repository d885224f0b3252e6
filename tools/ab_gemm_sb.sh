# dev A/B (run on the GPU box, inside ONE gpurun call: box-to-box spread is larger than most effects): forms of gg_gemm_nt_f32 on the model's shapes.
#   GG_GEMM_F32_SB=0        the double-buffered 3-workgroups-per-CU ring instead of the single-buffer 4-per-CU form
#   GG_GEMM_F32_DEBUG=128   64-byte-run epilogue stores / loads instead of the paired whole-line ones
#   GG_GEMM_F32_PRO_RING=0  register-staged prologue GEMMs
cd $GRAFT_REPO_ROOT
export GG_DEV_SWITCHES=1
for cfg in "" "GG_GEMM_F32_SB=0" "GG_GEMM_F32_DEBUG=128" "GG_GEMM_F32_PRO_RING=0"; do
  echo "== ${cfg:-default}"
  env $cfg timeout -k 10 300 python tools/bench_gemm_f32.py 2>&1 | grep -v "amdgpu.ids\|4096"
done
for cfg in "" "GG_GEMM_F32_SB=0" "GG_GEMM_F32_DEBUG=128"; do
  echo "== step ${cfg:-default}"
  env $cfg timeout -k 10 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 | cut -c1-200
done

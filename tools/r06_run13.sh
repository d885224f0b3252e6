cd $GRAFT_REPO_ROOT
for rep in 1 2; do for V in 0 1; do echo "GG_DW_F32_NT=$V" >> gpurun_out/dw_nt.log; GG_DEV_SWITCHES=1 GG_DW_F32_NT=$V python tools/bench_dw_f32.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> gpurun_out/dw_nt.log; done; done
cat gpurun_out/dw_nt.log

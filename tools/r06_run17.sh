cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "prologue" > gpurun_out/t17.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t17.log
for rep in 1 2; do
echo "scalar GELU"; python tools/bench_split3_pro.py 2>&1 | grep -v amdgpu.ids | sed 's/(  *[0-9]* GB\/s)//g'
echo "packed GELU"; GG_DEV_SWITCHES=1 GG_SPLIT3_PRO_PK=1 python tools/bench_split3_pro.py 2>&1 | grep -v amdgpu.ids | sed 's/(  *[0-9]* GB\/s)//g'
done

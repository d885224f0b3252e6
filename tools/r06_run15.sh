cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split3" > gpurun_out/t15.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t15.log
python tools/bench_split3_pro.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/split3_pro.log
python -m pytest tests/test_gpu_precision.py -m gpu -q -k "every_split_route_forced or gate_at_64" > gpurun_out/t15b.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t15b.log

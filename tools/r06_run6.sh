cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split3" > gpurun_out/t6.log 2>&1; echo "pytest rc=$?" > gpurun_out/t6.rc
for Z in 0 1 A small 0; do echo "ZERO=$Z" >> gpurun_out/split3_zero.log; ZERO=$Z python tools/bench_split3a.py s2.fc2 s2.fc1 s1.fc1 2>&1 | grep "^s" | cut -c1-150 >> gpurun_out/split3_zero.log; done
for E in gelu dact res; do
  for X in 0 1; do
    echo "EPI=$E NO_EC=$X" >> gpurun_out/split3_ec.log
    if [ $X = 1 ]; then export GG_DEV_SWITCHES=1 GG_SPLIT3_NO_EC=1; else unset GG_DEV_SWITCHES GG_SPLIT3_NO_EC; fi
    EPI=$E python tools/bench_split3a.py s1.fc1 s1.fc2 s2.fc1 s2.fc2 s3.fc1 s3.fc2 2>&1 | grep "^s" | cut -c100-150 >> gpurun_out/split3_ec.log
  done
done
cat gpurun_out/t6.rc; tail -2 gpurun_out/t6.log; cat gpurun_out/split3_zero.log gpurun_out/split3_ec.log

# round-6 GPU run 3: the 32x32x16 form of the split NT kernel (A/B in one process order: wide default, then GG_SPLIT3A_MFMA=16), TN remap/skew A/B needs git stash -> skip
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split3" > gpurun_out/t3.log 2>&1; echo "pytest rc=$?" > gpurun_out/t3.rc
python tools/bench_split3a.py s2 s3 edge > gpurun_out/split3w.log 2>&1
GG_DEV_SWITCHES=1 GG_SPLIT3A_MFMA=32 python tools/bench_split3a.py s2 s3 edge > gpurun_out/split3w_old16.log 2>&1
GG_DEV_SWITCHES=1 GG_SPLIT3A_ABL=256 python tools/bench_split3a.py s2.fc1 s2.fc2 s3.fc1 > gpurun_out/split3w_nohint.log 2>&1
bash tools/pmc_split.sh r06c nt.s2fc2 nt.s2fc1 > gpurun_out/pmc3.log 2>&1
cat gpurun_out/t3.rc; tail -2 gpurun_out/t3.log; cut -c1-150 gpurun_out/split3w.log; cut -c1-150 gpurun_out/split3w_old16.log

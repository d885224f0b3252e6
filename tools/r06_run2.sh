# round-6 GPU run 2: split kernels after the LDS swizzle fix / TN remap + skew; cache-policy variants (timing + HBM traffic)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split3 or split" > gpurun_out/t2.log 2>&1; echo "pytest rc=$?" > gpurun_out/t2.rc
python tools/bench_split3a.py > gpurun_out/split3a_swz.log 2>&1
python tools/bench_split3_tn.py > gpurun_out/split3_tn_remap.log 2>&1
bash tools/pmc_split.sh r06b nt tn > gpurun_out/pmc2.log 2>&1
bash tools/pmc_split_traffic.sh r06b_abl0 nt.s2fc1 nt.s3fc1 tn > /dev/null 2>&1
for A in 16 32 48 112; do
  echo "ABL=$A" >> gpurun_out/split3a_abl.log
  GG_DEV_SWITCHES=1 GG_SPLIT3A_ABL=$A python tools/bench_split3a.py s2.fc1 s2.fc2 s3.fc1 2>&1 | grep "^s" | cut -c1-150 >> gpurun_out/split3a_abl.log
  GG_DEV_SWITCHES=1 GG_SPLIT3A_ABL=$A bash tools/pmc_split_traffic.sh r06b_abl$A nt.s2fc1 nt.s3fc1 > /dev/null 2>&1
done
cat gpurun_out/t2.rc; tail -2 gpurun_out/t2.log

cd $GRAFT_REPO_ROOT
for V in 0 1 2 3 0 3; do echo "GG_SPLIT3_TN=$V" >> gpurun_out/tn_ab.log; GG_DEV_SWITCHES=1 GG_SPLIT3_TN=$V python tools/bench_split3_tn.py 2>&1 | grep "^s3" | cut -c1-140 >> gpurun_out/tn_ab.log; done
cat gpurun_out/tn_ab.log

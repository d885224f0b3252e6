cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py -q -x -k "stride2 or train_step" 2>&1 | tail -3
python tools/prof_step.py --fp32 2>/dev/null | grep "^dwconv" | head -8

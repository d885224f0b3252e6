cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision.py -q -x 2>&1 | tail -2
python tools/prof_step.py --fp32 2>/dev/null | sed -n 2,18p | cut -c1-120

set -x
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/prof_v8 gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v8 -o v8 --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof_v8.log 2>&1
tail -1 gpurun_out/prof_v8.log | cut -c1-200
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
find gpurun_out/prof_v8 gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head -20
du -sh gpurun_out/prof_v8 gpurun_out/pmc_fetch gpurun_out/pmc_write

cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for V in "SWZ=1 ABL=16" "SWZ=0 ABL=16" "SWZ=1 ABL=256" "SWZ=0 ABL=256"; do
  set -- $V; S=${1#SWZ=}; A=${2#ABL=}
  echo "swizzle=$S (1 = conflict-free permutation, 0 = round-5 XOR)  stores=$A (16 = non-temporal, 256 = default policy)" >> gpurun_out/nt_ab.log
  GG_DEV_SWITCHES=1 GG_SPLIT3_SWZ=$S GG_SPLIT3A_ABL=$A python tools/bench_split3a.py s1.fc1 s2.qkv s2.fc1 s2.fc2 s3.fc1 s3.fc2 2>&1 | grep "^s" | cut -c1-8,95-130 >> gpurun_out/nt_ab.log
done
done
cat gpurun_out/nt_ab.log

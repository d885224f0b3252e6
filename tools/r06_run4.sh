# round-6 GPU run 4: epilogue-heavy shapes, 256 x 128 (one workgroup per CU) against 128 x 128 (two per CU)
cd $GRAFT_REPO_ROOT
for E in gelu dact res; do
  for T in 256 128; do
    echo "EPI=$E TILE=$T" >> gpurun_out/split3_epi_tiles.log
    EPI=$E GG_DEV_SWITCHES=1 GG_SPLIT3A_TILE=$T python tools/bench_split3a.py s1.fc1 s2.qkv s2.fc1 s2.fc2 s3.fc1 s3.fc2 2>&1 | grep "^s" | cut -c100-150 >> gpurun_out/split3_epi_tiles.log
  done
done
cat gpurun_out/split3_epi_tiles.log

cd $GRAFT_REPO_ROOT
for A in 16 4 2 6 1; do echo "TILE=256 ABL=$A (16 = normal, 4 = no A path, 2 = no fragment reads, 6 = neither, 1 = no MFMA)"; GG_DEV_SWITCHES=1 GG_SPLIT3A_TILE=256 GG_SPLIT3A_ABL=$A python tools/bench_split3a.py s1.fc1 s1.qkv 2>&1 | grep "^s" | cut -c1-8,95-130; done
echo "TILE=128 normal"; python tools/bench_split3a.py s1.fc1 s1.qkv 2>&1 | grep "^s" | cut -c1-8,95-130

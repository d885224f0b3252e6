# dev: what bounds gemm_nt_split3a_kernel -- the same shapes with one ingredient removed at a time (GG_SPLIT3A_ABL; results are garbage, only the time counts)
for A in 0 1 2 4 8 6; do echo "ABL=$A"; GG_DEV_SWITCHES=1 GG_SPLIT3A_TILE=256 GG_SPLIT3A_ABL=$A python tools/bench_split3a.py s2.fc1 s2.fc2 s3.fc2 2>&1 | grep "^s" | cut -c1-150; done

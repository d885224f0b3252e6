# dev: HBM fetch / write bytes of single fp32 GEMM shapes (rocprofv3 PMC, separate passes): bash tools/pmc_one_gemm.sh "M N K" ...
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd $R
for shp in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc1
    timeout -k 10 200 rocprofv3 --pmc $c -d /tmp/pmc1 -o x --output-format csv -- python3 tools/one_gemm_f32.py $shp > /tmp/pmc1.log 2>&1
    F=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
    python3 - "$F" "$c" "$shp" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2] and "gemm_nt_f32" in r["Kernel_Name"]]
v = [float(r["Counter_Value"]) for r in rows]
M, N, K = (int(x) for x in sys.argv[3].split())
mult = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0
alg = 4.0 * (M * K + N * K) if sys.argv[2] == "FETCH_SIZE" else 4.0 * M * N
print(f"{sys.argv[3]:24s} {sys.argv[2]:10s} launches {len(v)}  measured {mult * sum(v) / len(v) * 1024 / 1e9:8.3f} GB/launch   algorithmic {alg / 1e9:8.3f} GB   ratio {mult * sum(v) / len(v) * 1024 / alg:.2f}")
PY
  done
done

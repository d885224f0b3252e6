# dev: SQ issue / wait counters per kernel for one fp32 step (run on the GPU box): bash tools/pmc_sq.sh -> gpurun_out/pmc_sq.txt
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/pmc_sq
( while true; do sleep 60; echo "[pmc_sq] still running" ; done ) &
HB=$!
timeout -k 10 380 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_sq -o s --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --precision fp32 > gpurun_out/pmc_sq.log 2>&1
kill $HB
F=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$F" > gpurun_out/pmc_sq.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:25]
for k, c in rows:
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:72s} n={n[k]:3d} wave_cyc={wc:.3g} mfma_busy/busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(c.get('SQ_BUSY_CYCLES',1),1):.3f} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} wait_lds={c.get('SQ_WAIT_INST_LDS',0)/wc:.2f} valu_insts={c.get('SQ_INSTS_VALU',0):.3g}")
PY
rm -rf gpurun_out/pmc_sq
cat gpurun_out/pmc_sq.txt

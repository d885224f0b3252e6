# dev: A/B of two builds on the model's fp32 GEMM shapes in ONE gpurun call (box-to-box variance is 2-3 %):
#   cp geoguessr-ai_amd/lib/libgg.so tools/bin/libgg_prev.so; <change + rebuild>; gpurun -- bash tools/ab_lib.sh
cd $GRAFT_REPO_ROOT
python tools/bench_gemm_f32.py > /dev/null 2>&1      # warm the box
GG_LIB=$PWD/tools/bin/libgg_prev.so python tools/bench_gemm_f32.py 2>/dev/null | grep " us " > gpurun_out/ab_prev.txt
python tools/bench_gemm_f32.py 2>/dev/null | grep " us " > gpurun_out/ab_new.txt
GG_LIB=$PWD/tools/bin/libgg_prev.so python tools/bench_gemm_f32.py 2>/dev/null | grep " us " > gpurun_out/ab_prev2.txt
python tools/bench_gemm_f32.py 2>/dev/null | grep " us " > gpurun_out/ab_new2.txt
python - <<'P'
import re
def rd(f):
    d={}
    for l in open(f):
        m=re.match(r"(.*?)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) us", l)
        if m: d[(m.group(1).strip(), m.group(2), m.group(3), m.group(4))]=float(m.group(5))
    return d
a,b,a2,b2=rd("gpurun_out/ab_prev.txt"),rd("gpurun_out/ab_new.txt"),rd("gpurun_out/ab_prev2.txt"),rd("gpurun_out/ab_new2.txt")
for k in a:
    p, n = min(a[k], a2.get(k, 1e9)), min(b.get(k, 1e9), b2.get(k, 1e9))
    print(f"{k[0]:22s} M={k[1]:>8s} N={k[2]:>5s} K={k[3]:>5s}  prev {p:8.1f} us  new {n:8.1f} us  {100.0 * (n - p) / p:+5.1f} %")
P

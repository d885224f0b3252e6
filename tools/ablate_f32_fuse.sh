# fp32-mode schedule ablation: step time with each BatchNorm fusion family switched off (run on the GPU box)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --precision fp32"
run() { echo "== $1"; env $2 timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_breakdown']
print(d['ms_per_step'], {n:round(v['ms_per_step'],1) for n,v in k.items() if isinstance(v,dict)})"; }
run all_on GG_X=1
run no_fuse GG_F32_NO_FUSE=1
run no_pro GG_NO_PRO=1
run no_bngemm GG_NO_BNGEMM=1
run no_bnbwd_epi GG_NO_FUSE_BNBWD_EPI=1
run no_bnbwd GG_NO_FUSE_BNBWD=1
run no_dw_s1 GG_NO_FUSE_DW_S1=1
run no_dw_s2 GG_NO_FUSE_DW_S2=1

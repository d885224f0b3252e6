cd $GRAFT_REPO_ROOT
python tools/bench_attn_f32.py > /dev/null 2>&1
for i in 1 2; do
echo prev; GG_LIB=$PWD/tools/bin/libgg_prev.so python tools/bench_attn_f32.py 2>/dev/null | grep ws=
echo new; python tools/bench_attn_f32.py 2>/dev/null | grep ws=
done

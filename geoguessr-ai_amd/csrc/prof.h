// RAII HIP-event bracket around a kernel launch (active only after gg_prof_enable(1)).
#pragma once
void gg_set_error(const char* fmt, ...);
enum { GG_CAT_GEMM = 0, GG_CAT_ATTN = 1, GG_CAT_DWCONV = 2, GG_CAT_NORM = 3, GG_CAT_HEAD = 4, GG_CAT_OPTIM = 5, GG_CAT_MOVE = 6, GG_NUM_CATS = 7,
       GG_CAT_SPLIT_FLAG = 16 };      // or-ed into GG_CAT_GEMM by the split-bf16 GEMMs: bench.py prices them against the bf16 matrix peak / 6 (category = value & 15)
struct GgProfScope {
    GgProfScope(int cat, double flops, double bytes, void* stream);
    ~GgProfScope();
    int idx_; void* stream_;
};
#define GG_PROF(cat, flops, bytes, stream) GgProfScope gg_prof_scope_((cat), (double)(flops), (double)(bytes), (stream))

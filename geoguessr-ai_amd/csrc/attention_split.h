// fp32 window attention on the bf16 matrix pipe (included by attention_flash.hip, inside its anonymous namespace, after the fl_* helpers).
//
// The fp32 flash kernels compute every product on v_mfma_f32_16x16x4_f32 -- exact f32, but at the VECTOR ALU's rate (1/16 of the bf16 MFMA) and sharing the
// SIMD's vector issue with the softmax arithmetic: a 16 x 16 x 32 tile product is 8 instructions = 256 matrix-pipe cycles.  Here every f32 operand is carried as
// THREE bf16 planes (x = x1 + x2 + x3: 24 significand bits, the same split as gg_gemm_nt_split3 / gg_split3_bf16) and a product is six bf16 MFMAs
//     (a1 b3 + a2 b2 + a3 b1) + (a1 b2 + a2 b1) + a1 b1        (small terms first; f32 accumulation; the dropped terms are < 2^-24 of the product)
// = 96 cycles for the same tile on v_mfma_f32_16x16x32_bf16 (every contraction is 32 deep: the 16-deep bf16 MFMA costs the same 16 cycles, so products over
// tokens pair two 16-token tiles per instruction), and the bf16 MFMA does not share its issue with the vector ALU the way the f32 MFMA does.  Error against an
// fp64 reference (tools/check_attn_split.py): below the f32-MFMA kernels' on all six forward / backward shapes of TinyViT-21M-224 (about half
// of it in the backward, where -lse joins the exponent after the product instead of seeding the score accumulator); the kernels pass the fp32 mode's own
// gates at unchanged tolerances (tests/test_gpu_kernels.py::test_flash_attention_forward_backward, tests/test_gpu_precision.py).
// Operands that are reused -- K / V (forward), Q / dO (backward) of the whole window -- are split ONCE while they are staged into LDS (three bf16 images
// of 64-byte rows, 16-byte chunks XOR-swizzled with (-(row >> 2)) & 3: conflict-free fragment ds_read_b128); per-wave strips are split once in registers;
// probabilities / dS tiles are split where they are formed (5 vector instructions per score).
// Layout conventions are those of the f32 kernels (swapped scores in the forward: a lane owns a query column, so softmax statistics are lane-local).
typedef short sp_s4 __attribute__((ext_vector_type(4)));

// NPL planes: 3 for f32 storage (x = x1 + x2 + x3), 1 for bf16 storage (the value itself: one MFMA per product, no split arithmetic -- the same kernels are
// the bf16 mode's single-pass window attention)
template <int NPL> struct Sp8T { bf16x8 p[NPL]; };          // 8 values (the 32-deep contraction slots of a lane)
template <int NPL> struct Sp4T { sp_s4 p[NPL]; };           // 4 values (half of them: one transposing read), raw bf16 bits

struct Sp1 { bf16 a, b, c; };
__device__ __forceinline__ Sp1 sp_split1(float x) {
    Sp1 o;
    o.a = (bf16)x;
    const float r1 = x - (float)o.a;
    o.b = (bf16)r1;
    o.c = (bf16)(r1 - (float)o.b);
    return o;
}
template <int NPL> __device__ __forceinline__ Sp8T<NPL> sp_split8(const f32x4& lo, const f32x4& hi) {
    Sp8T<NPL> o;
    if constexpr (NPL == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { o.p[0][j] = (bf16)lo[j]; o.p[0][4 + j] = (bf16)hi[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Sp1 u = sp_split1(lo[j]), v = sp_split1(hi[j]);
            o.p[0][j] = u.a; o.p[1][j] = u.b; o.p[2][j] = u.c;
            o.p[0][4 + j] = v.a; o.p[1][4 + j] = v.b; o.p[2][4 + j] = v.c;
        }
    }
    return o;
}
// acc += A . B over a 32-deep contraction (three planes: six products, small terms first)
template <int NPL> __device__ __forceinline__ f32x4 sp_mma32(const Sp8T<NPL>& a, const Sp8T<NPL>& b, f32x4 acc) {
    if constexpr (NPL == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
    else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[2], b.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
        return acc;
    }
}
// 8 consecutive elements at byte offset `off` of a descriptor (FL_OOB: zeros), widened to f32
template <typename T> struct SpLd8;
template <> struct SpLd8<float> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int off, f32x4& lo, f32x4& hi) {
        lo = Bld<float>::load(rs, off); hi = Bld<float>::load(rs, off == FL_OOB ? FL_OOB : off + 16);
    }
};
template <> struct SpLd8<bf16> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int off, f32x4& lo, f32x4& hi) {
        const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo[j] = (float)v[j]; hi[j] = (float)v[4 + j]; }
    }
};
// plane images: [NPL][R][32] bf16, rows of 64 bytes; element offset of 16-byte chunk ch (8 elements) of row `row` within a plane
__device__ __forceinline__ int sp_off(int row, int ch) { return row * 32 + ((ch ^ ((-(row >> 2)) & 3)) << 3); }
// all R rows of one head's 32-column block -> plane images (zero rows beyond N); colb = byte offset of the block in a row
template <typename T, int NPL>
__device__ __forceinline__ void sp_stage(const FlashParams& p, __amdgpu_buffer_rsrc_t rs, int ldb, int colb, bf16* X, int R) {
    const int PL = R * 32;
    for (int id = threadIdx.x; id < R * 4; id += blockDim.x) {
        const int row = id >> 2, ch = id & 3;
        const int off = row < p.N ? fl_tokrel(p, row) * ldb + colb + ch * 8 * (int)sizeof(T) : FL_OOB;
        f32x4 lo, hi;
        SpLd8<T>::load(rs, off, lo, hi);
        const Sp8T<NPL> s = sp_split8<NPL>(lo, hi);
        const int o = sp_off(row, ch);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<bf16x8*>(X + pl * PL + o) = s.p[pl];
    }
}
// row-contiguous fragment (A / B operand of a product over d): row `row`, this lane's 8 contraction slots d = 8 lg .. 8 lg + 7
template <int NPL> __device__ __forceinline__ Sp8T<NPL> sp_frag(const bf16* X, int PL, int row, int lg) {
    Sp8T<NPL> f;
    const int o = sp_off(row, lg);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) f.p[pl] = *reinterpret_cast<const bf16x8*>(X + pl * PL + o);
    return f;
}
// token-major ("transposed") fragment (A operand of a product over the 16 tokens of tile rows row0 ..): lane (lr, lg) receives X[row0 + 4 lg + j][16 c + lr],
// j = 0 .. 3, through the transposing LDS read (within a 16-lane group, lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3; EXEC all ones)
template <int NPL> __device__ __forceinline__ Sp4T<NPL> sp_frag_t(const bf16* X, int PL, int row0, int c, int lr, int lg) {
    const int row = row0 + 4 * lg + (lr >> 2), col = 16 * c + 4 * (lr & 3);
    const int o = sp_off(row, col >> 3) + (col & 7);
    Sp4T<NPL> f;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) f.p[pl] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sp_s4*)(X + pl * PL + o));
    return f;
}
// two token tiles (rows row0 .. and row1 ..) as one 32-deep token-major fragment: slot j <-> row (j < 4 ? row0 : row1) + 4 lg + (j & 3)
template <int NPL> __device__ __forceinline__ Sp8T<NPL> sp_frag_t2(const bf16* X, int PL, int row0, int row1, int c, int lr, int lg) {
    const Sp4T<NPL> a = sp_frag_t<NPL>(X, PL, row0, c, lr, lg), b = sp_frag_t<NPL>(X, PL, row1, c, lr, lg);
    Sp8T<NPL> f;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        union { sp_s4 s[2]; bf16x8 v; } u;
        u.s[0] = a.p[pl]; u.s[1] = b.p[pl];
        f.p[pl] = u.v;
    }
    return f;
}

// ------------------------------------------------------------------------------------------- forward
// One workgroup per (window, head), one wave per 16-query strip; K and V of the window are staged once as plane images.  Exact (two-sweep) softmax: the NT
// score tiles of a strip stay in registers (NT <= 16: windows of at most 256 tokens), so nothing is rescaled.
template <typename T, int NPL, int NT>
__global__ __launch_bounds__(64 * NT) void flash_fwd_split_kernel(FlashParams p) {
    constexpr int D = 32, R = 16 * NT, PL = R * 32, ES = (int)sizeof(T);
    typedef Sp8T<NPL> Sp8;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    bf16* Kp = reinterpret_cast<bf16*>(fsm);
    bf16* Vp = Kp + NPL * PL;
    float* btab = reinterpret_cast<float*>(Vp + NPL * PL);
    int* klin = reinterpret_cast<int*>(btab + p.nbpad);
    const int h = blockIdx.x % p.nh, w = blockIdx.x / p.nh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    const int span = fl_span(p);
    const int ldb = (int)p.ld * ES, ldob = (int)p.ldo * ES;
    const __amdgpu_buffer_rsrc_t rsQKV = fl_rsrc(reinterpret_cast<const T*>(p.qkv) + origin * p.ld, span * ldb);
    const __amdgpu_buffer_rsrc_t rsOUT = fl_rsrc(reinterpret_cast<T*>(p.out) + origin * p.ldo, span * ldob);
    const __amdgpu_buffer_rsrc_t rsLSE = fl_rsrc(p.lse ? p.lse + origin * p.nh : nullptr, p.lse ? span * p.nh * 4 : 0);
    // every global load of the workgroup is issued before anything waits: the K / V rows (R * 4 eight-column chunks over 64 NT threads: one each) and the
    // wave's query strip; the bias table and the coordinates are staged under their latency
    const int srow = threadIdx.x >> 2, sch = threadIdx.x & 3;                     // (R * 4 == 64 * NT: one chunk per thread)
    const int soff = srow < p.N ? fl_tokrel(p, srow) * ldb + (hc + sch * 8) * ES : FL_OOB;
    f32x4 klo, khi, vlo, vhi, qlo, qhi;
    SpLd8<T>::load(rsQKV, soff == FL_OOB ? FL_OOB : soff + p.k_off * ES, klo, khi);
    SpLd8<T>::load(rsQKV, soff == FL_OOB ? FL_OOB : soff + p.v_off * ES, vlo, vhi);
    const int strip = wave;
    // this wave's query strip: 8 contraction slots d = 8 lg .. of query row lr, split once
    const int qi = strip * 16 + lr;
    const bool qok = qi < p.N;
    const int qrel = qok ? fl_tokrel(p, qi) : 0;
    SpLd8<T>::load(rsQKV, qok ? qrel * ldb + (p.q_off + hc + 8 * lg) * ES : FL_OOB, qlo, qhi);
    if (has_bias) fl_stage_bias(p, h, btab);
    if (has_bias) fl_stage_coords(p, 0, klin, R);
    {
        const Sp8 k3 = sp_split8<NPL>(klo, khi), v3 = sp_split8<NPL>(vlo, vhi);
        const int o = sp_off(srow, sch);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            *reinterpret_cast<bf16x8*>(Kp + pl * PL + o) = k3.p[pl];
            *reinterpret_cast<bf16x8*>(Vp + pl * PL + o) = v3.p[pl];
        }
    }
    const Sp8 q3 = sp_split8<NPL>(qlo, qhi);
    const int qlin = has_bias ? fl_lin4(p, min(qi, p.N - 1)) + 4 * (p.ws - 1) * 2 * p.ws : 0;
    const float sc2 = p.scale * 1.4426950408889634f;
    __syncthreads();
    if (strip * 16 >= p.N) return;                                  // (a strip of padding only: nothing to do; no barrier follows)
    // S^T[key][q] tiles: lane holds keys 16 kt + 4 lg + r of query lr
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (16 * kt + 15 >= p.N) {                                  // (wave-uniform: the window's last tile) padded keys are masked through the initial value
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = (16 * kt + 4 * lg + r < p.N) ? 0.f : -INFINITY;
        }
        s = sp_mma32<NPL>(sp_frag<NPL>(Kp, PL, 16 * kt + lr, lg), q3, s);
        f32x4 bia = {0.f, 0.f, 0.f, 0.f};
        if (has_bias) {
            const i32x4 kl4 = *reinterpret_cast<const i32x4*>(klin + 16 * kt + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) bia[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + (qlin - kl4[r]));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r] = fmaf(s[r], sc2, bia[r]); mx = fmaxf(mx, s[r]); }       // exp2 domain: s * scale * log2 e + bias * log2 e
        st[kt] = s;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    f32x4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // O^T[d][q] += sum over keys of V^T[d][key] P[key][q], two key tiles (32 keys) per MFMA: v_mfma_f32_16x16x16_bf16 takes the cycles of the 16x16x32 form
    // (tools/mfma_rate16.hip: 17 cycles either way), so a 16-deep contraction would run the pipe half empty.  Slot j of lane group lg <-> key
    // 16 (kt + (j >> 2)) + 4 lg + (j & 3): the accumulator layout of two score tiles on the P side, two transposing reads on the V side
#pragma unroll
    for (int kt = 0; kt < NT; kt += 2) {
        f32x4 e0, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) { e0[r] = __builtin_amdgcn_exp2f(st[kt][r] - mx); l += e0[r]; }
        if (kt + 1 < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { e1[r] = __builtin_amdgcn_exp2f(st[kt + 1 < NT ? kt + 1 : kt][r] - mx); l += e1[r]; }
        }
        const Sp8 p3 = sp_split8<NPL>(e0, e1);
#pragma unroll
        for (int c = 0; c < 2; ++c) oacc[c] = sp_mma32<NPL>(sp_frag_t2<NPL>(Vp, PL, 16 * kt, kt + 1 < NT ? 16 * (kt + 1) : 16 * kt, c, lr, lg), p3, oacc[c]);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < 2; ++c) Bld<T>::store(rsOUT, qok ? qrel * ldob + (h * D + 16 * c + 4 * lg) * ES : FL_OOB, oacc[c] * inv);
    if (p.lse) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx * 0.6931471805599453f + __logf(l)), rsLSE,
                                                     (qok && lg == 0) ? (qrel * p.nh + h) * 4 : FL_OOB, 0, 0);
}
size_t sp_lds_fwd(const FlashParams& p, int npl) { return (size_t)2 * npl * p.npad * 32 * 2 + ((size_t)p.nbpad + p.npad) * 4; }

// ------------------------------------------------------------------------------------------- backward, single pass
// The structure of flash_bwd_fused_kernel (one workgroup per (window, head); Q and dO of the window in LDS, here as plane images; the dQ image in LDS is the
// accumulator of the dQ product; waves walk the query tiles staggered, one barrier per step), re-tiled so that EVERY product contracts over 32:
// a wave owns TWO key strips (32 keys) and a step covers TWO query tiles (32 queries).  Per step and wave
//   S, dP            4 tile products over d (32)                                      48 MFMAs
//   dV, dK           2 strips x 2 column blocks over the 32 queries of the step       48        (P / dS of the two query tiles side by side in the 8 slots)
//   dQ               2 query tiles x 2 column blocks over the wave's 32 keys          24        (dS crosses a 2 x 1.25 KB slot to reach the B layout)
// = 30 per 16 x 16 score tile, all v_mfma_f32_16x16x32_bf16 (against 40 x 2 passes of the f32 form).  K / V strips are split once into registers (B operands
// of S / dP; K once more key-major as the A operand of dQ), P / dS where they are formed, dS once more after the transposing slot.
template <typename T, int NPL, bool DBIAS, int NT>
__global__ __launch_bounds__(64 * ((NT + 1) / 2)) void flash_bwd_split_kernel(FlashParams p) {
    constexpr int D = 32, NP = (NT + 1) / 2, R = 32 * NP, PL = R * 32, RS = D + 4, TS = 20, ES = (int)sizeof(T);
    typedef Sp8T<NPL> Sp8;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    bf16* Qp = reinterpret_cast<bf16*>(fsm);
    bf16* Op = Qp + NPL * PL;                                        // dO planes
    float* dQs = reinterpret_cast<float*>(Op + NPL * PL);            // dQ accumulator image (unscaled)
    float* btab = dQs + R * RS;
    float* dbt = btab + p.nbpad;
    float* lse_s = dbt + (DBIAS ? p.nbpad : 0);
    float* del_s = lse_s + R;
    int* qlin = reinterpret_cast<int*>(del_s + R);
    float* Gt = reinterpret_cast<float*>(qlin + R);                // per wave: one dS slot [16 q][TS] (the two key strips take turns)
    const int wh = blockIdx.x;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    const int nb = p.ws * p.ws;
    const int w2 = 2 * p.ws - 1;
    const int span = fl_span(p);
    const int ldb = (int)p.ld * ES, lddob = (int)p.lddo * ES, ldob = (int)p.ldo * ES;
    const __amdgpu_buffer_rsrc_t rsQKV = fl_rsrc(reinterpret_cast<const T*>(p.qkv) + origin * p.ld, span * ldb);
    const __amdgpu_buffer_rsrc_t rsDO = fl_rsrc(reinterpret_cast<const T*>(p.dout) + origin * p.lddo, span * lddob);
    const __amdgpu_buffer_rsrc_t rsO = fl_rsrc(reinterpret_cast<const T*>(p.out) + origin * p.ldo, span * ldob);
    const __amdgpu_buffer_rsrc_t rsLSE = fl_rsrc(p.lse + origin * p.nh, span * p.nh * 4);
    const __amdgpu_buffer_rsrc_t rsDQKV = fl_rsrc(reinterpret_cast<T*>(p.dqkv) + origin * p.ld, span * ldb);
    // every global load of the workgroup is issued before anything waits: Q, dO and O rows (two 8-column chunks per thread: R * 4 chunks over 64 NP threads),
    // lse, then (below) the K / V strips; the LDS-only setup runs under their latency
    f32x4 qv[2][2], gv[2][2], ov[2][2];
    float lsev[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = threadIdx.x + i * (64 * NP), row = id >> 2, ch = id & 3;
        const bool ok = row < p.N;
        const int rel = ok ? fl_tokrel(p, row) : 0;
        const int oq = ok ? rel * ldb + (p.q_off + hc + ch * 8) * ES : FL_OOB, og = ok ? rel * lddob + (h * D + ch * 8) * ES : FL_OOB, oo = ok ? rel * ldob + (h * D + ch * 8) * ES : FL_OOB;
        SpLd8<T>::load(rsQKV, oq, qv[i][0], qv[i][1]);
        SpLd8<T>::load(rsDO, og, gv[i][0], gv[i][1]);
        SpLd8<T>::load(rsO, oo, ov[i][0], ov[i][1]);
        lsev[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsLSE, (ok && ch == 0) ? (rel * p.nh + h) * 4 : FL_OOB, 0, 0));
    }
    // the wave's two key strips: rows as B operands (lane: key lr of the strip, slots d = 8 lg ..), K once more key-major (lane: d = 16 c + lr, slots
    // j < 4: key 4 lg + j of strip 0, j >= 4: of strip 1 -- the order the dS fragment below uses)
    Sp8 k3[2], v3[2], kT3[2];
    int klin[2], krow[2];
    bool kok[2];
    f32x4 kv[2][2], vv[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ki = 32 * wave + 16 * s + lr;
        kok[s] = ki < p.N;
        krow[s] = kok[s] ? fl_tokrel(p, ki) * ldb : FL_OOB;
        SpLd8<T>::load(rsQKV, kok[s] ? krow[s] + (p.k_off + hc + 8 * lg) * ES : FL_OOB, kv[s][0], kv[s][1]);
        SpLd8<T>::load(rsQKV, kok[s] ? krow[s] + (p.v_off + hc + 8 * lg) * ES : FL_OOB, vv[s][0], vv[s][1]);
        klin[s] = has_bias ? fl_lin4(p, min(ki, p.N - 1)) - 4 * (p.ws - 1) * 2 * p.ws : 0;      // (a dead second strip of the last wave is computed like a live one: zero K / V rows, results dropped)
    }
    // (without a bias table: one zero entry and zero coordinates -- the step below is ONE basic block, no wave-uniform branches, so that the scheduler can
    // run the vector work of one product under the matrix work of another)
    if (has_bias) {
        fl_stage_bias(p, h, btab);
        if (NPL == 1 && p.round_bias) {
            // bf16 storage, forward by attention.hip (expanded table: the bias enters the score accumulator as bf16(bias / scale)); P is recomputed from that
            // forward's lse, so the same rounded value goes here
            __syncthreads();
            for (int i = threadIdx.x; i < w2 * w2; i += blockDim.x) {
                const float b = btab[i] * 0.6931471805599453f;                             // fl_stage_bias stores bias * log2 e
                btab[i] = (float)(bf16)(b * -p.neg_inv_scale) * (p.scale * 1.4426950408889634f);
            }
        }
    } else if (threadIdx.x < 4) btab[threadIdx.x] = 0.f;
    if (DBIAS) for (int i = threadIdx.x; i < w2 * w2; i += blockDim.x) dbt[i] = 0.f;
    for (int i = threadIdx.x; i < R * RS / 4; i += blockDim.x) reinterpret_cast<f32x4*>(dQs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_bias) fl_stage_coords(p, 0, qlin, R); else for (int i = threadIdx.x; i < R; i += blockDim.x) qlin[i] = 0;
    const float sc2 = p.scale * 1.4426950408889634f;
    float* slot = Gt + wave * (16 * TS);
    float* tslot = reinterpret_cast<float*>(Qp) + wave * (2 * 16 * TS);      // K transposition scratch: the plane area, before the planes are written (barrier below)
    {
        f32x4 kt[2][2];                                            // [c][strip]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            k3[s] = sp_split8<NPL>(kv[s][0], kv[s][1]);
            v3[s] = sp_split8<NPL>(vv[s][0], vv[s][1]);
            // transposed through the wave's slots, one per 16-column block: [key = lr][16 d] at stride TS (lane groups 2 c, 2 c + 1 hold the columns of block c)
            *reinterpret_cast<f32x4*>(tslot + (lg >> 1) * (16 * TS) + lr * TS + 8 * (lg & 1)) = kv[s][0];
            *reinterpret_cast<f32x4*>(tslot + (lg >> 1) * (16 * TS) + lr * TS + 8 * (lg & 1) + 4) = kv[s][1];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[c][s][r] = tslot[c * (16 * TS) + (4 * lg + r) * TS + lr];           // K[key = 4 lg + r][d = 16 c + lr]
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) kT3[c] = sp_split8<NPL>(kt[c][0], kt[c][1]);
    }
    f32x4 dk[2][2], dv[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) dk[s][c] = dv[s][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                               // every wave is done with its transposition scratch
    // Q / dO rows -> plane images; -delta = -sum_d dO O (4 threads per row) and -lse / scale -> row scalars
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = threadIdx.x + i * (64 * NP), row = id >> 2, ch = id & 3;
        const Sp8 q3 = sp_split8<NPL>(qv[i][0], qv[i][1]), g3 = sp_split8<NPL>(gv[i][0], gv[i][1]);
        const int o = sp_off(row, ch);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            *reinterpret_cast<bf16x8*>(Qp + pl * PL + o) = q3.p[pl];
            *reinterpret_cast<bf16x8*>(Op + pl * PL + o) = g3.p[pl];
        }
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) dsum = fmaf(gv[i][0][j], ov[i][0][j], dsum);
#pragma unroll
        for (int j = 0; j < 4; ++j) dsum = fmaf(gv[i][1][j], ov[i][1][j], dsum);
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        if (ch == 0) {
            del_s[row] = -dsum;
            lse_s[row] = row < p.N ? -1.4426950408889634f * lsev[i] : -INFINITY;      // -lse in the exp2 domain: joins the bias AFTER the product (below)
        }
    }
    __syncthreads();

    int pair = wave;
    for (int t = 0; t < NP; ++t) {
        f32x4 pr[2][2], ds[2][2];                                  // [query tile of the pair][strip]; lane holds [q = 16 tile + 4 lg + r][key = lr]
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int tile = 2 * pair + a, q4 = tile * 16 + 4 * lg;
            const Sp8 qf = sp_frag<NPL>(Qp, PL, tile * 16 + lr, lg), of = sp_frag<NPL>(Op, PL, tile * 16 + lr, lg);
            // S starts from zero and -lse is added with the bias: an accumulator that starts at -lse / scale (magnitude ~170 against scores of ~30) rounds every one
            // of the six partial products to the ulp of 170 -- that was the 15 % by which this kernel's gradients missed the f32-MFMA kernel's error
            const f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, nl4 = *reinterpret_cast<const f32x4*>(lse_s + q4), dp0 = *reinterpret_cast<const f32x4*>(del_s + q4);      // dP - delta
            const i32x4 ql4 = *reinterpret_cast<const i32x4*>(qlin + q4);
            f32x4 dT[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 st = sp_mma32<NPL>(qf, k3[s], st0), dp = sp_mma32<NPL>(of, v3[s], dp0);
                const i32x4 boff4 = ql4 - klin[s];
                f32x4 bia;
#pragma unroll
                for (int r = 0; r < 4; ++r) bia[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + boff4[r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(st[r], sc2, bia[r] + nl4[r]));     // -inf for padded queries -> 0
                    const float g = e * dp[r];
                    pr[a][s][r] = e;
                    ds[a][s][r] = g;                                                    // the softmax scale is applied once, to dK and dQ
                    slot[(4 * lg + r) * TS + lr] = g;                                   // (a padded key's column is finite and meets zero K rows in the dQ product)
                    if (DBIAS && kok[s] && q4 + r < p.N) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(dbt) + boff4[r]), g);
                }
                // dS[q = lr][keys 4 lg .. of this strip] back from the slot (LDS operations of a wave complete in order: the next strip's writes follow this read)
                dT[s] = *reinterpret_cast<const f32x4*>(slot + lr * TS + 4 * lg);
            }
            // dQ^T[d][q] += sum over the wave's keys of K^T[d][key] dS^T[key][q]
            const f32x4 d0 = dT[0], d1 = dT[1];
            const Sp8 dst = sp_split8<NPL>(d0, d1);
            float* dQt = dQs + (tile * 16 + lr) * RS + 4 * lg;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 dq = sp_mma32<NPL>(kT3[c], dst, *reinterpret_cast<const f32x4*>(dQt + 16 * c));
                *reinterpret_cast<f32x4*>(dQt + 16 * c) = dq;
            }
        }
        // dV^T[d][key] += sum over the 32 queries of dO^T[d][q] P[q][key], dK likewise with Q and dS
        Sp8 ot[2], qt[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ot[c] = sp_frag_t2<NPL>(Op, PL, 32 * pair, 32 * pair + 16, c, lr, lg);
            qt[c] = sp_frag_t2<NPL>(Qp, PL, 32 * pair, 32 * pair + 16, c, lr, lg);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Sp8 p3 = sp_split8<NPL>(pr[0][s], pr[1][s]), g3 = sp_split8<NPL>(ds[0][s], ds[1][s]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                dv[s][c] = sp_mma32<NPL>(ot[c], p3, dv[s][c]);
                dk[s][c] = sp_mma32<NPL>(qt[c], g3, dk[s][c]);
            }
        }
        __syncthreads();                                                       // the pair's next writer (the wave one down) reads after this
        pair = pair + 1 == NP ? 0 : pair + 1;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {           // (rows beyond N carry FL_OOB: dropped by the range check)
            Bld<T>::store(rsDQKV, kok[s] ? krow[s] + (p.k_off + hc + 16 * c + 4 * lg) * ES : FL_OOB, dk[s][c] * p.scale);
            Bld<T>::store(rsDQKV, kok[s] ? krow[s] + (p.v_off + hc + 16 * c + 4 * lg) * ES : FL_OOB, dv[s][c]);
        }
    constexpr int CH = D / 4;
    for (int id = threadIdx.x; id < p.N * CH; id += blockDim.x) {
        const int row = id / CH, ch = id % CH;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dQs + row * RS + ch * 4);
        Bld<T>::store(rsDQKV, fl_tokrel(p, row) * ldb + (p.q_off + hc + ch * 4) * ES, v * p.scale);
    }
    if (DBIAS) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {      // fold the signed offsets back onto attention_biases[|dy| * ws + |dx|]
            const int ady = fl_div(i, p.rcp_ws), adx = i - ady * p.ws, c0 = (p.ws - 1) * w2 + (p.ws - 1);
            float tt = dbt[c0 + ady * w2 + adx];
            if (ady) tt += dbt[c0 - ady * w2 + adx];
            if (adx) tt += dbt[c0 + ady * w2 - adx];
            if (ady && adx) tt += dbt[c0 - ady * w2 - adx];
            if (p.dbias_part) p.dbias_part[((int64_t)w * p.nh + h) * nb + i] = tt;
            else atomicAdd(&p.dbias[h * nb + i], tt);
        }
    }
}
size_t sp_lds_bwd(const FlashParams& p, bool dbias, int npl) {
    const size_t np = (p.npad / 16 + 1) / 2, R = 32 * np;
    return 2 * npl * R * 64 + (R * 36 + (size_t)p.nbpad * (dbias ? 2 : 1) + 3 * R + np * 320) * 4;
}

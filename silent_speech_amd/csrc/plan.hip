// plan.hip -- the execution plan of the transduction model as NATIVE code: ONE C call enqueues the whole forward pass
// (reference architecture.py:61-84, :29-40, transformer.py:43-60,87-112) and ONE the whole hand-derived backward pass
// (what loss.backward(), transduction_model.py:209, runs through autograd in the reference): ~450 kernel launches per
// training step that used to cost ~25 us of Python + ctypes each (12.7 ms of host time per 20 ms step on the bench box).
//
// The plan owns no memory.  The caller binds device pointers to NAMED slots once (parameters, GEMM-ready weight copies,
// gradient buffers: ss_plan_slot_name / ss_plan_bind), and hands a workspace to every forward call; activations, saved
// tensors and backward temporaries are carved from it by a bump allocator whose sequence is identical in "dry" mode
// (ss_plan_workspace_bytes) and in the real run.  The saved-for-backward pointers travel in a small host struct the caller
// keeps between forward and backward (ss_plan_ctx_bytes).
//
// Canonical activation layout: (B, T, C) row-major == a flat [B*T][C] matrix; convolution inputs live in (B, T+2, C)
// buffers with a zero halo row at both ends of every sequence, so a k=3 window is one contiguous 3C-wide row and
// conv == GEMM with an overlapping-row ss_rowmap (csrc/gemm.hip).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>
#include <stdlib.h>
#include <string>
#include <vector>

namespace {

struct BnP { float *gamma = 0, *beta = 0, *rmean = 0, *rvar = 0, *dgamma = 0, *dbeta = 0; long long* nbt = 0; };
struct PermB { void* jobs = 0; void* blocks = 0; long long total = 0, all_f32 = 0; };
struct BlockP {
    void *w1f = 0, *w2f = 0, *wr = 0, *wrT = 0, *w2b = 0, *w1b_even = 0, *w1b_odd = 0;
    float *b1 = 0, *br = 0, *b2 = 0;
    BnP bn1, bn2, bnr;
    float *c2_stage = 0, *c1_stage = 0, *wr_grad = 0;
    PermB unpack;
    int O = 0, I = 0;
};
struct LayerP {
    void *wqkv = 0, *wqkvT = 0, *wo = 0, *woT = 0, *E = 0, *ET = 0, *EF = 0, *w1 = 0, *w2 = 0, *w1T = 0, *w2T = 0;
    float *b1 = 0, *b2 = 0, *g1 = 0, *be1 = 0, *g2 = 0, *be2 = 0;
    float *dg1 = 0, *dbe1 = 0, *dg2 = 0, *dbe2 = 0, *dw1 = 0, *db1 = 0, *dw2 = 0, *db2 = 0, *wo_stage = 0, *wqkv_stage = 0;
    PermB unpack;          // this layer's re-laid-out gradients (w_o, w_q / w_k / w_v) -> .grad arena
};

// pointers saved by forward for backward (all into the caller's workspace)
struct BlockCtx { void *xin, *c1, *cr, *h1, *c2, *y; float *m1, *i1, *m2, *i2, *mr, *ir, *scratch; int Tin, Cin, Tout, O, pad_y; };
struct LayerCtx { void *x, *qkv, *qkvT, *o, *z1, *y1, *hid, *z2, *pimg; float *lse, *mean1, *rstd1, *mean2, *rstd2; unsigned char* hid_sign; };
constexpr int MAX_LAYERS = 16;
// hi / lo bf16 planes of f32 buffers (parity-grade mode on the 8-wave kernels): every GEMM operand is split ONCE per step, on first use, into
// [2][elems] bf16 carved from the workspace (hi plane, then lo); forward activations keep theirs for the weight-gradient GEMMs of the backward.
constexpr int MAX_PLANES = 384;
struct PlaneCache { int n; const void* key[MAX_PLANES]; void* hi[MAX_PLANES]; long long elems[MAX_PLANES]; };
struct Ctx {
    int B, T0, T, M, Tp, need_T, n_layers;
    float p_drop, scale; unsigned long long seed;
    BlockCtx blk[3];
    LayerCtx layer[MAX_LAYERS];
    void *conv_out, *x_final;
    PlaneCache planes;
    unsigned long long ws_used;
    char* ws; unsigned long long ws_bytes;
};

struct Exec {      // bump allocator + streams; dry = size pass (no launches)
    bool dry; char* base; size_t off, cap; void* stream; void* side;
    void* alloc(size_t bytes) { off = (off + 255) & ~(size_t)255; char* p = base + off; off += bytes; return p; }
};

static ss_rowmap RM(long long row_stride, int rows_per_batch = 0, long long batch_stride = 0, long long base = 0) {
    ss_rowmap m; m.base = base; m.batch_stride = batch_stride; m.row_stride = row_stride; m.rows_per_batch = rows_per_batch; return m;
}
static ss_gemm_epilogue EPI() { ss_gemm_epilogue e; memset(&e, 0, sizeof(e)); e.alpha = 1.f; e.gate_scale = 1.f; return e; }
static int round_up(int x, int m) { return (x + m - 1) / m * m; }

typedef double (*reduce_fn)(void* user, float* sums, int n_floats, double n_local, void* stream);
typedef void (*event_fn)(void* user, int what, void* stream);

struct Plan {
    ss_model_dims D;
    std::vector<std::string> names; std::vector<void**> targets;
    BlockP blk[3]; std::vector<LayerP> layers;
    void *w_raw_in = 0, *w_raw_in_T = 0, *w_head = 0, *w_head_T = 0;
    float *b_raw_in = 0, *dw_raw_in = 0, *db_raw_in = 0, *b_head = 0, *head_w_stage = 0, *head_b_stage = 0, *stage_arena = 0;
    long long stage_arena_bytes = 0;
    PermB unpack_enc, unpack_all;          // heads only | heads + every encoder layer (one launch: runs without the per-layer events)
    reduce_fn hook = 0; void* hook_user = 0;
    event_fn on_event = 0; void* event_user = 0;
    int side_enabled = 1, dw_grouped = 1, side_blocks = 2, fuse_stats = 1, regate_on = 1, f32_x3 = 0;
    int x3_attn = 0;         // option 8: the bound EF tables are the [hi | lo] tables of ss_relpos_attention_x3_prepare_tables -> an f32_x3 plan with planes runs the attention on them
    int sign_gate = 1;       // option 10: bf16 training plans keep the sign of the FFN hidden activation as one bit per element (written by linear1's epilogue) and the FFN input gradient gates from it
    int x3_emit = 1;         // option 9: plane GEMMs whose consumers take planes emit them from their epilogue (off = a split pass per consumer-side first use, the first form of round 6)
    int x3_planes = 1;       // option 7: an f32_x3 plan runs every GEMM the 8-wave kernel can take on hi / lo bf16 planes (off = operands split in registers on the 128 x 128 kernels, round 4)
    Ctx* cur = nullptr;      // the context of the call in flight (plane cache)
    bool use_planes() const { return D.dtype == SS_F32 && f32_x3 && x3_planes; }
    // the attention of the parity-grade mode on hi / lo planes (attention_t.hip x3 kernels): its outputs exist ONLY as planes, so every consumer must be a plane GEMM
    bool use_x3_attention(int T) const { return use_planes() && x3_attn && dw_grouped && ss_relpos_attention_x3_supported(T, D.dp, D.max_rel); }
    // a buffer that already holds [hi plane | lo plane] (written by a kernel that emits planes): consumers find it in the cache under `key`
    void adopt_planes(const void* key, void* hi, long long n) {
        PlaneCache& pc = cur->planes;
        if (key && pc.n < MAX_PLANES) { pc.key[pc.n] = key; pc.hi[pc.n] = hi; pc.elems[pc.n] = n; ++pc.n; }
    }
    int keep_input = 0;      // option 6: leave x_raw as it is (the shifted signal is only handed out in `shifted`; a functional caller copies it back itself)
    hipEvent_t ev_fork = 0, ev_join = 0;
    size_t esz() const { return D.dtype == SS_BF16 ? 2 : 4; }

    // ---- per-launch timing (HIP events on the launch stream) for the roofline report of bench.py
    struct ProfRec { const char* tag; int sub; double flops, bytes; hipEvent_t e0, e1; };
    bool profiling = false;
    std::vector<ProfRec> recs; std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    hipEvent_t prof_event() {
#if !defined(SS_EMU)
        if (ev_used == ev_pool.size()) { hipEvent_t e; hipEventCreate(&e); ev_pool.push_back(e); }
        return ev_pool[ev_used++];
#else
        return 0;
#endif
    }
    template <class F>
    int timed(Exec& X, const char* tag, double flops, double bytes, void* stream, F&& f, bool gemm_sub = false) {
        if (X.dry) return f();
#if !defined(SS_EMU)
        if (profiling) {
            ProfRec r{tag, 0, flops, bytes, prof_event(), prof_event()};
            hipEventRecord(r.e0, (hipStream_t)stream);
            const int rc = f();
            hipEventRecord(r.e1, (hipStream_t)stream);
            if (gemm_sub) r.sub = ss_gemm_last_kernel();
            recs.push_back(r);
            return rc;
        }
#endif
        return f();
    }

    void slot(const std::string& n, void** t) { names.push_back(n); targets.push_back(t); }
    void slot_bn(const std::string& p, BnP& b) {
        slot(p + ".weight", (void**)&b.gamma); slot(p + ".bias", (void**)&b.beta); slot(p + ".running_mean", (void**)&b.rmean); slot(p + ".running_var", (void**)&b.rvar);
        slot(p + ".weight.grad", (void**)&b.dgamma); slot(p + ".bias.grad", (void**)&b.dbeta); slot(p + ".num_batches_tracked", (void**)&b.nbt);
    }
    void slot_perm(const std::string& p, PermB& b) { slot(p + ".jobs", &b.jobs); slot(p + ".blocks", &b.blocks); slot(p + ".total", (void**)&b.total); slot(p + ".all_f32", (void**)&b.all_f32); }

    explicit Plan(const ss_model_dims& d) : D(d), layers(d.n_layers) {
        for (int i = 0; i < 3; ++i) {
            const std::string p = "conv_blocks." + std::to_string(i) + ".";
            BlockP& b = blk[i];
            b.O = D.d_model; b.I = i == 0 ? 8 : D.d_model;
            slot(p + "w1f", &b.w1f); slot(p + "w2f", &b.w2f); slot(p + "wr", &b.wr); slot(p + "wrT", &b.wrT); slot(p + "w2b", &b.w2b);
            if (i > 0) { slot(p + "w1b_even", &b.w1b_even); slot(p + "w1b_odd", &b.w1b_odd); }
            slot(p + "conv1.bias", (void**)&b.b1); slot(p + "residual_path.bias", (void**)&b.br); slot(p + "conv2.bias", (void**)&b.b2);
            slot_bn(p + "bn1", b.bn1); slot_bn(p + "bn2", b.bn2); slot_bn(p + "res_norm", b.bnr);
            slot(p + "conv2.weight.stage", (void**)&b.c2_stage); slot(p + "conv1.weight.stage", (void**)&b.c1_stage); slot(p + "residual_path.weight.grad", (void**)&b.wr_grad);
            slot_perm(p + "unpack", b.unpack);
        }
        slot("w_raw_in", &w_raw_in); slot("w_raw_in_T", &w_raw_in_T); slot("w_raw_in.bias", (void**)&b_raw_in);
        slot("w_raw_in.weight.grad", (void**)&dw_raw_in); slot("w_raw_in.bias.grad", (void**)&db_raw_in);
        for (int l = 0; l < D.n_layers; ++l) {
            const std::string p = "transformer.layers." + std::to_string(l) + ".";
            LayerP& L = layers[l];
            slot(p + "wqkv", &L.wqkv); slot(p + "wqkvT", &L.wqkvT); slot(p + "wo", &L.wo); slot(p + "woT", &L.woT); slot(p + "E", &L.E); slot(p + "ET", &L.ET); slot(p + "EF", &L.EF);
            slot(p + "w1", &L.w1); slot(p + "w2", &L.w2); slot(p + "w1T", &L.w1T); slot(p + "w2T", &L.w2T);
            slot(p + "linear1.bias", (void**)&L.b1); slot(p + "linear2.bias", (void**)&L.b2);
            slot(p + "norm1.weight", (void**)&L.g1); slot(p + "norm1.bias", (void**)&L.be1); slot(p + "norm2.weight", (void**)&L.g2); slot(p + "norm2.bias", (void**)&L.be2);
            slot(p + "norm1.weight.grad", (void**)&L.dg1); slot(p + "norm1.bias.grad", (void**)&L.dbe1); slot(p + "norm2.weight.grad", (void**)&L.dg2); slot(p + "norm2.bias.grad", (void**)&L.dbe2);
            slot(p + "linear1.weight.grad", (void**)&L.dw1); slot(p + "linear1.bias.grad", (void**)&L.db1); slot(p + "linear2.weight.grad", (void**)&L.dw2); slot(p + "linear2.bias.grad", (void**)&L.db2);
            slot(p + "w_o.stage", (void**)&L.wo_stage); slot(p + "w_qkv.stage", (void**)&L.wqkv_stage);
            slot_perm(p + "unpack", L.unpack);
        }
        slot("w_head", &w_head); slot("w_head_T", &w_head_T); slot("b_head", (void**)&b_head);
        slot("head_w.stage", (void**)&head_w_stage); slot("head_b.stage", (void**)&head_b_stage);
        slot("stage_arena", (void**)&stage_arena); slot("stage_arena.bytes", (void**)&stage_arena_bytes);
        slot_perm("unpack_encoder", unpack_enc); slot_perm("unpack_encoder_all", unpack_all);
    }

    // ---------------------------------------------------------------- small launch helpers (all return non-zero on error)
    // elements an operand's row map spans (rows x row_len window)
    static long long extent(const ss_rowmap& m, int rows, int row_len) {
        const long long nb = m.rows_per_batch > 0 ? (rows - 1) / m.rows_per_batch : 0, rr = m.rows_per_batch > 0 ? (rows < m.rows_per_batch ? rows : m.rows_per_batch) - 1 : rows - 1;
        return m.base + nb * m.batch_stride + rr * m.row_stride + row_len;
    }
    // hi plane of the f32 buffer [src, src + n) (lo plane n elements behind it); split on first use, on the MAIN stream (every consumer -- side-stream
    // launches included -- is ordered behind it by the fork that precedes them).  key: what identifies the buffer (its address; a weight: its slot)
    struct Pl { void* hi; void* lo; };
    // planes a producer kernel will write for the dense f32 buffer `key` of n elements (registered in the cache; nullptr pair when the mode is off)
    Pl new_planes(Exec& X, const void* key, long long n) {
        if (!(use_planes() && x3_emit)) return Pl{nullptr, nullptr};
        n = (n + 7) & ~7LL;
        void* hi = X.alloc((size_t)n * 4);
        adopt_planes(key, hi, n);
        return Pl{hi, (char*)hi + n * 2};
    }
    Pl planes(Exec& X, const void* key, const void* src, long long n) {
        PlaneCache& pc = cur->planes;
        // (a null key is never reused: in the sizing pass the first workspace buffer and the caller's dhead both have address 0)
        if (key) for (int i = 0; i < pc.n; ++i) if (pc.key[i] == key && pc.elems[i] >= n) return Pl{pc.hi[i], (char*)pc.hi[i] + pc.elems[i] * 2};
        n = (n + 7) & ~7LL;
        void* hi = X.alloc((size_t)n * 4);
        if (key && pc.n < MAX_PLANES) { pc.key[pc.n] = key; pc.hi[pc.n] = hi; pc.elems[pc.n] = n; ++pc.n; }
        if (!X.dry) {
            if (timed(X, "split_planes", 0, (double)n * 8, X.stream, [&] { return ss_split_planes((const float*)src, hi, (char*)hi + n * 2, n, X.stream); })) return Pl{nullptr, nullptr};
        }
        return Pl{hi, (char*)hi + n * 2};
    }

    int gemm(Exec& X, int dt_out, const void* A, const void* B, void* C, int M, int N, int K, ss_rowmap am, ss_rowmap bm, ss_rowmap cm, const ss_gemm_epilogue* e = 0,
             int a_mode = SS_OP_KC, int b_mode = SS_OP_KC, int split = 1, void* stream = 0, const void* wkey = nullptr, int emit = 0) {
        // emit (plane GEMMs only): 1 = the result also leaves as hi / lo planes, written by the epilogue and found by its consumers in the plane cache under C
        // (no split pass over C: 12 -> 8 bytes of traffic per element); 2 = ... and the f32 C is not written at all (every consumer takes planes: 12 -> 4)
        void* st = stream ? stream : X.stream;
        if (use_planes() && st == X.stream && a_mode == SS_OP_KC && b_mode == SS_OP_KC && dt_out == SS_F32 && split == 1 &&
            ss_gemm_planes_supported(SS_F32, C, M, N, K, &am, &bm, &cm, e)) {
            const Pl a = planes(X, A, A, extent(am, M, K)), b = planes(X, wkey ? wkey : B, B, extent(bm, N, K));
            ss_gemm_epilogue ee = e ? *e : EPI();
            if (emit && x3_emit) {
                const long long n = (extent(cm, M, N) + 7) & ~7LL;
                void* hi = X.alloc((size_t)n * 4);
                adopt_planes(C, hi, n);
                ee.planes_hi = hi; ee.planes_lo = (char*)hi + n * 2; ee.planes_only = emit == 2;
            }
            if (X.dry) return 0;
            if (!a.hi || !b.hi) return 1;
            return timed(X, "gemm", 2.0 * M * N * K, ((double)M * K + (double)N * K) * 4 + (double)M * N * 4.0, st,
                         [&] { return ss_gemm_planes(SS_F32, a.hi, a.lo, b.hi, b.lo, C, M, N, K, &am, &bm, &cm, &ee, st); }, true);
        }
        if (X.dry) return 0;
        const double ob = dt_out == SS_BF16 ? 2.0 : 4.0;
        return timed(X, a_mode == SS_OP_OC && b_mode == SS_OP_OC ? "gemm_dw" : "gemm", 2.0 * M * N * K, ((double)M * K + (double)N * K) * esz() + (double)M * N * ob, st,
                     [&] { return ss_gemm(D.dtype == SS_F32 && f32_x3 ? SS_F32X3 : D.dtype, dt_out, a_mode, b_mode, A, B, C, M, N, K, &am, &bm, &cm, e, split, st); }, true);
    }
    int colsum(Exec& X, const void* x, int rows, int C, float* out, void* stream) {
        float* scratch = (float*)X.alloc((size_t)ss_colsum_scratch_floats(rows, C) * 4);
        if (X.dry) return 0;
        return timed(X, "colsum", 0, (double)rows * C * esz(), stream, [&] { return ss_colsum(D.dtype, x, rows, C, C, scratch, out, stream); });
    }
    int permute_batch(Exec& X, const PermB& b, void* stream) {
        if (X.dry || !b.jobs || b.total <= 0) return 0;
        return timed(X, "grad_unlayout", 0, 0, stream, [&] { return ss_permute3d_batch(b.jobs, (const int32_t*)b.blocks, (int)b.total, (int)b.all_f32, stream); });
    }
    // fused != null: [2][C] sums already accumulated by the producing GEMM's epilogue (shift = the running mean before this update)
    // reduced_n > 0: the sums were already all-reduced by the caller (two BatchNorms of a block in ONE collective), reduced_n = global row count
    int bn_stats(Exec& X, const void* x, int B, int T, int C, float* scratch, BnP& bn, bool training, float** mean, float** invstd, float* fused = nullptr, double reduced_n = 0.0) {
        *mean = (float*)X.alloc((size_t)C * 4); *invstd = (float*)X.alloc((size_t)C * 4);
        float* sums = (training && !fused) ? (float*)X.alloc((size_t)3 * C * 4) : fused;
        if (X.dry) return 0;
        double n_total = (double)B * T;
        if (training) {
            if (!fused && timed(X, "bn_stats", 0, (double)B * T * C * esz(), X.stream, [&] { return ss_bn_stats_sums(D.dtype, x, B, T, C, 0, scratch, hook ? bn.rmean : nullptr, sums, X.stream); })) return 1;
            if (reduced_n > 0.0) n_total = reduced_n;
            else if (hook) n_total = hook(hook_user, sums, 2 * C, n_total, X.stream);
        }
        if (fused) return timed(X, "bn_finalize", 0, 0, X.stream, [&] { return ss_bn_finalize_shift(sums, bn.rmean, n_total, C, *mean, *invstd, bn.rmean, bn.rvar, 0.1f, 1e-5f, 1, X.stream); });
        return timed(X, "bn_finalize", 0, 0, X.stream, [&] { return ss_bn_finalize(sums, n_total, C, *mean, *invstd, bn.rmean, bn.rvar, 0.1f, 1e-5f, training ? 1 : 0, X.stream); });
    }
    // a forward convolution whose output feeds a training-mode BatchNorm: the batch statistics leave with the GEMM epilogue when the
    // 8-wave kernel runs this shape (returns the [2][C] sums then, else null -> the stand-alone statistics pass)
    float* conv_gemm(Exec& X, const void* A, const void* Bw, void* C, int M, int N, int K, ss_rowmap am, ss_rowmap bm, const float* bias, BnP& bn, float* stat_slot, int* rc) {
        ss_gemm_epilogue e = EPI(); e.bias = bias;
        ss_rowmap cm = RM(N);
        float* fused = nullptr;
        if (!X.dry && stat_slot && fuse_stats) {
            // the 8-wave kernel accumulates the statistics in its epilogue: bf16 plans, and the plane form of the parity-grade mode (same kernel, f32 out)
            ss_gemm_epilogue es = e; es.col_sum = stat_slot; es.col_sumsq = stat_slot + N; es.col_shift = bn.rmean;
            const bool ok = use_planes() ? ss_gemm_planes_supported(SS_F32, C, M, N, K, &am, &bm, &cm, &es) != 0
                                         : ss_gemm_fuses_column_stats(D.dtype, D.dtype, SS_OP_KC, SS_OP_KC, C, M, N, K, &am, &bm, &cm, &e, 1) != 0;
            if (ok) { fused = stat_slot; e = es; }
        }
        *rc = gemm(X, D.dtype, A, Bw, C, M, N, K, am, bm, cm, &e);
        return fused;
    }
    // side stream: the weight-gradient GEMMs, bias column sums and gradient un-layouts do not feed the backward chain
    void fork(Exec& X) {
#if !defined(SS_EMU)
        if (X.dry || !X.side || X.side == X.stream) return;
        hipEventRecord(ev_fork, (hipStream_t)X.stream); hipStreamWaitEvent((hipStream_t)X.side, ev_fork, 0);
#endif
    }
    void join(Exec& X) {
#if !defined(SS_EMU)
        if (X.dry || !X.side || X.side == X.stream) return;
        hipEventRecord(ev_join, (hipStream_t)X.side); hipStreamWaitEvent((hipStream_t)X.stream, ev_join, 0);
#endif
    }
    int split_k(int M, int N, int K, bool overlapped) const {   // 128-wide transposing-read kernel (f32 mode / grouping off): see the measurements in engine.py history
        const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        int s = (overlapped ? 400 : 512) / (tiles > 0 ? tiles : 1); s = s < 1 ? 1 : (s > 64 ? 64 : s);
        const int cap = K >= 512 ? K / 512 : 1;
        return s < cap ? s : (cap < 1 ? 1 : cap);
    }

    struct DwGroup {
        Plan* P; Exec* X; bool grouped; int n = 0; ss_dw_job jobs[24];
        int add(const void* dy, const void* x, float* grad, int N, int K, int rows, ss_rowmap am, ss_rowmap bm, void* stream) {
            if (grouped && P->use_planes()) {
                // hi / lo planes of both operands: dW = dY_lo^T X_hi + dY_hi^T X_lo + dY_hi^T X_hi as three jobs that accumulate into the same gradient
                if (n + 3 > 24 && launch(stream)) return 1;
                const Pl a = P->planes(*X, dy, dy, extent(am, rows, N)), b = P->planes(*X, x, x, extent(bm, rows, K));
                if (X->dry) return 0;
                if (!a.hi || !b.hi) return 1;
                const void* as[3] = {a.lo, a.hi, a.hi}; const void* bs[3] = {b.hi, b.lo, b.hi};
                for (int t = 0; t < 3; ++t) {
                    ss_dw_job& j = jobs[n++]; memset(&j, 0, sizeof(j));
                    j.A = as[t]; j.B = bs[t]; j.C = grad; j.amap = am; j.bmap = bm; j.ldc = K; j.M = N; j.N = K; j.K = rows; j.flags = 1;
                }
                return 0;
            }
            if (grouped) {
                if (n == 8 && launch(stream)) return 1;
                ss_dw_job& j = jobs[n++]; memset(&j, 0, sizeof(j));
                j.A = dy; j.B = x; j.C = grad; j.amap = am; j.bmap = bm; j.ldc = K; j.M = N; j.N = K; j.K = rows;
                return 0;
            }
            // exact-f32 mode / grouping off: one 128-wide split-K GEMM per weight, queued on the side stream behind everything the main
            // stream has produced so far
            ss_gemm_epilogue e = EPI(); e.mode = 2;
            const bool ov = X->side && X->side != X->stream;
            P->fork(*X);
            if (ov) ss_gemm_set_blocks_per_cu(P->side_blocks);
            const int rc = P->gemm(*X, SS_F32, dy, x, grad, N, K, rows, am, bm, RM(K), &e, SS_OP_OC, SS_OP_OC, P->split_k(N, K, rows, ov), stream);
            if (ov) ss_gemm_set_blocks_per_cu(2);
            return rc;
        }
        int launch(void* stream) {
            int rc = 0;
            if (n && !X->dry) {
                double fl = 0, by = 0;
                const double pl = P->use_planes() ? 3.0 : 1.0;      // plane form: three jobs per gradient -> useful flops = a third
                for (int i = 0; i < n; ++i) { fl += 2.0 * jobs[i].M * jobs[i].N * jobs[i].K / pl; by += ((double)jobs[i].M + jobs[i].N) * jobs[i].K * 2.0 + (double)jobs[i].M * jobs[i].N * 4.0 / pl; }
                rc = P->timed(*X, "gemm_dw_grouped", fl, by, stream, [&] { return ss_gemm_dw_grouped(n, jobs, stream); });
            }
            n = 0; return rc;
        }
    };

    int forward(Exec& X, const float* x_raw, float* shifted, int B, int T0, int training, int shift_r, float p_drop, unsigned long long seed, float* head, Ctx* c);
    int backward(Exec& X, Ctx* c, const float* dhead);
};

#define L_(call) do { if (call) return 1; } while (0)

// (query, key) pairs inside the relative-position band |k - q| <= D - 1 of one sequence: the attention kernels' algorithmic work is
// 2 dp flops per pair and product -- 3 products forward (QK, QE, PV), 5 backward (dP, dS K, dR E, P^T dO, dS^T Q)
static double band_pairs(int T, int D) { double n = 0; for (int q = 0; q < T; ++q) { const int lo = q - (D - 1) < 0 ? 0 : q - (D - 1), hi = q + (D - 1) > T - 1 ? T - 1 : q + (D - 1); n += hi - lo + 1; } return n; }

int Plan::forward(Exec& X, const float* x_raw, float* shifted, int B, int T0, int training, int shift_r, float p_drop, unsigned long long seed, float* head, Ctx* c)
{
    const int dt = D.dtype, d = D.d_model; const size_t es = esz();
    const int Cin0 = 8;
    SS_CHECK(T0 % 8 == 0, "raw EMG length %d must be a multiple of 8 (three stride-2 convolutions)", T0);
    SS_CHECK(D.n_layers <= MAX_LAYERS, "at most %d encoder layers", MAX_LAYERS);
    void* stream = X.stream;
    cur = c; c->planes.n = 0;
    if (!training) p_drop = 0.f;
    c->B = B; c->T0 = T0; c->p_drop = p_drop; c->seed = seed; c->n_layers = D.n_layers;

    void* xin = X.alloc((size_t)B * (T0 + 2) * Cin0 * es);
    if (!X.dry) {
        L_(timed(X, "emg_prepare", 0, (double)B * T0 * Cin0 * (4 + es), stream, [&] { return ss_emg_prepare(dt, x_raw, xin, (training && shift_r > 0) ? shifted : nullptr, B, T0, Cin0, training ? shift_r : 0, stream); }));
        if (training && shift_r > 0 && shifted && !keep_input)      // the reference mutates its input in place (architecture.py:67-68)
            if (!ss_copy_d2d_async((void*)x_raw, shifted, (size_t)B * T0 * Cin0 * 4, stream)) { ss_set_error("forward: input write-back failed"); return 1; }
    }
    // [9][2][C] per-channel sums of the nine BatchNorms, zeroed once: filled by the conv GEMM epilogues where the 8-wave kernel runs
    float* bnsums = training ? (float*)X.alloc((size_t)9 * 2 * d * 4) : nullptr;
    if (bnsums && !X.dry) {
        if (!ss_memset_async(bnsums, 0, (size_t)9 * 2 * d * 4, stream)) { ss_set_error("forward: memset failed"); return 1; }
    }
    int Tin = T0, Cin = Cin0;
    for (int i = 0; i < 3; ++i) {
        BlockP& w = blk[i]; BlockCtx& s = c->blk[i];
        const int O = w.O, Tout = Tin / 2, rows = B * Tout;
        s.xin = xin; s.Tin = Tin; s.Cin = Cin; s.Tout = Tout; s.O = O;
        s.scratch = (float*)X.alloc((size_t)ss_bn_scratch_floats(B, Tout, O) * 4);
        const long long in_bs = (long long)(Tin + 2) * Cin;
        int rc = 0;
        float* slot = bnsums ? bnsums + (size_t)i * 3 * 2 * d : nullptr;
        void* c1 = X.alloc((size_t)rows * O * es);
        float* f1 = conv_gemm(X, xin, w.w1f, c1, rows, O, 3 * Cin, RM(2 * Cin, Tout, in_bs), RM(3 * Cin), w.b1, w.bn1, slot, &rc); L_(rc);
        void* cr = X.alloc((size_t)rows * O * es);
        float* fr = conv_gemm(X, xin, w.wr, cr, rows, O, Cin, RM(2 * Cin, Tout, in_bs, Cin), RM(Cin), w.br, w.bnr, slot ? slot + 2 * d : nullptr, &rc); L_(rc);
        // data parallel: bn1 and res_norm both normalise a function of the block input, their epilogue sums sit side by side in the
        // slot -> ONE all-reduce of [4][O] floats instead of two of [2][O] (these collectives are latency-bound)
        double nred = 0.0;
        if (training && hook && f1 && fr == f1 + 2 * d && O == d && !X.dry) nred = hook(hook_user, f1, 4 * O, (double)B * Tout, stream);
        L_(bn_stats(X, c1, B, Tout, O, s.scratch, w.bn1, training, &s.m1, &s.i1, f1, nred));
        void* h1 = X.alloc((size_t)B * (Tout + 2) * O * es);
        if (!X.dry) L_(timed(X, "bn_apply", 0, (double)B * Tout * O * es * 2, stream, [&] { return ss_bn_apply(dt, c1, s.m1, s.i1, w.bn1.gamma, w.bn1.beta, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, h1, 1, B, Tout, O, 1, stream); }));
        void* c2 = X.alloc((size_t)rows * O * es);
        float* f2 = conv_gemm(X, h1, w.w2f, c2, rows, O, 3 * O, RM(O, Tout, (long long)(Tout + 2) * O), RM(3 * O), w.b2, w.bn2, slot ? slot + 4 * d : nullptr, &rc); L_(rc);
        L_(bn_stats(X, c2, B, Tout, O, s.scratch, w.bn2, training, &s.m2, &s.i2, f2));
        L_(bn_stats(X, cr, B, Tout, O, s.scratch, w.bnr, training, &s.mr, &s.ir, fr, nred));
        const int pad_y = i == 2 ? 0 : 1;
        void* y = X.alloc((size_t)B * (Tout + 2 * pad_y) * O * es);
        if (!X.dry) L_(timed(X, "bn_apply", 0, (double)B * Tout * O * es * 3, stream, [&] { return ss_bn_apply(dt, c2, s.m2, s.i2, w.bn2.gamma, w.bn2.beta, 0, cr, s.mr, s.ir, w.bnr.gamma, w.bnr.beta, 0, y, pad_y, B, Tout, O, 1, stream); }));
        s.c1 = c1; s.cr = cr; s.h1 = h1; s.c2 = c2; s.y = y; s.pad_y = pad_y;
        xin = y; Tin = Tout; Cin = O;
    }
    if (training && !X.dry) {
        long long* ctrs[9];
        for (int i = 0; i < 3; ++i) { ctrs[3 * i] = blk[i].bn1.nbt; ctrs[3 * i + 1] = blk[i].bn2.nbt; ctrs[3 * i + 2] = blk[i].bnr.nbt; }
        L_(timed(X, "counters", 0, 0, stream, [&] { return ss_counters_add(9, (int64_t**)ctrs, 1, stream); }));
    }
    const int T = Tin, M = B * T;
    c->T = T; c->M = M; c->conv_out = xin;
    void* x = X.alloc((size_t)M * d * es);
    { ss_gemm_epilogue e = EPI(); e.bias = b_raw_in; L_(gemm(X, dt, xin, w_raw_in, x, M, d, d, RM(d), RM(d), RM(d), &e, SS_OP_KC, SS_OP_KC, 1, 0, nullptr, 1)); }

    const int H = D.n_head, dp = D.dp, Dr = D.max_rel, ff = D.ff, HD = H * dp;
    const int Tp = round_up(T, 8);
    const float scale = 1.0f / sqrtf((float)D.d_qkv);
    c->Tp = Tp; c->scale = scale;
    const bool x3att = use_x3_attention(T);
    c->need_T = x3att ? 0 : ss_relpos_attention_needs_transposed(dt, T, dp, Dr);
    for (int l = 0; l < D.n_layers; ++l) {
        LayerP& w = layers[l]; LayerCtx& s = c->layer[l];
        s.x = x;
        void* qkv = X.alloc((size_t)M * 3 * HD * es);
        void* qkvT = nullptr;
        if (c->need_T) {      // the per-tile attention kernels read a per-sequence transposed copy, written by the same GEMM epilogue
            qkvT = X.alloc((size_t)B * 3 * HD * Tp * es);
            ss_gemm_epilogue e = EPI(); e.c2 = qkvT; e.cmap2 = RM(1, T, (long long)3 * HD * Tp); e.col_stride2 = Tp;
            L_(gemm(X, dt, x, w.wqkv, qkv, M, 3 * HD, d, RM(d), RM(d), RM(3 * HD), &e));
        } else L_(gemm(X, dt, x, w.wqkv, qkv, M, 3 * HD, d, RM(d), RM(d), RM(3 * HD), nullptr, SS_OP_KC, SS_OP_KC, 1, 0, nullptr, x3att ? 2 : 0));      // x3 attention: qkv exists as planes only
        void* o = X.alloc((size_t)M * HD * es);
        float* lse = (float*)X.alloc((size_t)B * H * T * 4);
        // training: the resident forward leaves its probabilities for the backward kernels (no recomputation of the logits there)
        const size_t pimg_bytes = training ? (size_t)(x3att ? ss_relpos_attention_x3_saved_bytes(B, H, T, dp, Dr) : ss_relpos_attention_saved_bytes(dt, B, H, T, dp, Dr)) : 0;
        void* pimg = pimg_bytes ? X.alloc(pimg_bytes) : nullptr;
        if (x3att) {
            // qkv is split once (the backward finds its planes in the cache); o leaves as planes IN the f32-sized buffer: [hi M x HD | lo M x HD] bf16
            const Pl q = planes(X, qkv, qkv, (long long)M * 3 * HD);
            adopt_planes(o, o, (long long)M * HD);
            if (!X.dry) {
                if (!q.hi) return 1;
                L_(timed(X, "attn_fwd", 2.0 * B * H * band_pairs(T, Dr) * dp * 3.0, (double)M * 4 * HD * es, stream, [&] { return ss_relpos_attention_x3_forward(q.hi, q.lo, w.EF, o, (char*)o + (size_t)M * HD * 2, lse, pimg, B, H, T, dp, Dr, scale, p_drop, seed, 4 * l, stream); }));
            }
        } else
        if (!X.dry) L_(timed(X, "attn_fwd", 2.0 * B * H * band_pairs(T, Dr) * dp * 3.0, (double)M * 4 * HD * es, stream, [&] { return ss_relpos_attention_forward_p(dt == SS_F32 && f32_x3 ? SS_F32X3 : dt, qkv, qkvT, w.E, w.EF, o, lse, pimg, B, H, T, Tp, dp, Dr, scale, p_drop, seed, 4 * l, stream); }));
        s.pimg = pimg;
        void* a = X.alloc((size_t)M * d * es);
        L_(gemm(X, dt, o, w.wo, a, M, d, HD, RM(HD), RM(HD), RM(d)));
        void* y1 = X.alloc((size_t)M * d * es);
        float* mean1 = (float*)X.alloc((size_t)M * 4); float* rstd1 = (float*)X.alloc((size_t)M * 4);
        const Pl y1p = new_planes(X, y1, (long long)M * d);      // (parity-grade mode: linear1 and dW1 take y1 as planes)
        if (!X.dry) L_(timed(X, "add_dropout_ln_fwd", 0, (double)M * d * es * 5, stream, [&] { return ss_add_dropout_layernorm_forward_planes(dt, x, a, w.g1, w.be1, y1, y1p.hi, y1p.lo, mean1, rstd1, M, d, D.ln_eps, p_drop, seed, 4 * l + 1, stream); }));
        void* hid = X.alloc((size_t)M * ff * es);
        unsigned char* hid_sign = nullptr;
        { ss_gemm_epilogue e = EPI(); e.bias = w.b1; e.relu = 1; e.dropout_p = p_drop; e.seed = seed; e.rng_stream = 4 * l + 2;
          ss_rowmap am_ = RM(d), bm_ = RM(d), cm_ = RM(ff);
          // training, bf16: the epilogue also writes [hid > 0] as one bit per element (M x ff / 8 bytes): the backward's ReLU / dropout gate reads those instead of
          // the 16 x larger tensor, all bytes of a tile in one request (transformer.py:57)
          if (training && dt == SS_BF16 && sign_gate && ff % 8 == 0 && ss_gemm_sign_bits_supported(dt, dt, SS_OP_KC, SS_OP_KC, hid, M, ff, d, &am_, &bm_, &cm_, &e, 1)) {
              hid_sign = (unsigned char*)X.alloc((size_t)M * (ff / 8));
              e.sign_out = hid_sign; e.sign_pitch = ff / 8;
          }
          L_(gemm(X, dt, y1, w.w1, hid, M, ff, d, am_, bm_, cm_, &e, SS_OP_KC, SS_OP_KC, 1, 0, nullptr, 1)); }      // planes for linear2 and dW2; the f32 copy is the backward's gate
        void* f = X.alloc((size_t)M * d * es);
        { ss_gemm_epilogue e = EPI(); e.bias = w.b2; L_(gemm(X, dt, hid, w.w2, f, M, d, ff, RM(ff), RM(ff), RM(d), &e)); }
        void* y2 = X.alloc((size_t)M * d * es);
        float* mean2 = (float*)X.alloc((size_t)M * 4); float* rstd2 = (float*)X.alloc((size_t)M * 4);
        const Pl y2p = new_planes(X, y2, (long long)M * d);      // (the next layer's qkv GEMM / dWqkv, or the output heads)
        if (!X.dry) L_(timed(X, "add_dropout_ln_fwd", 0, (double)M * d * es * 5, stream, [&] { return ss_add_dropout_layernorm_forward_planes(dt, y1, f, w.g2, w.be2, y2, y2p.hi, y2p.lo, mean2, rstd2, M, d, D.ln_eps, p_drop, seed, 4 * l + 3, stream); }));
        s.qkv = qkv; s.qkvT = qkvT; s.o = o; s.lse = lse; s.z1 = a; s.mean1 = mean1; s.rstd1 = rstd1; s.y1 = y1;
        s.hid = hid; s.hid_sign = hid_sign; s.z2 = f; s.mean2 = mean2; s.rstd2 = rstd2;
        x = y2;
    }
    c->x_final = x;
    const int nh = D.n_head_cols;
    { ss_gemm_epilogue e = EPI(); e.bias = b_head; L_(gemm(X, SS_F32, x, w_head, head, M, nh, d, RM(d), RM(d), RM(nh), &e)); }
    c->ws_used = X.off;
    return 0;
}

int Plan::backward(Exec& X, Ctx* c, const float* dhead)
{
    const int dt = D.dtype, d = D.d_model; const size_t es = esz();
    const int B = c->B, T = c->T, M = c->M, H = D.n_head, dp = D.dp, Dr = D.max_rel, ff = D.ff, HD = H * dp, Tp = c->Tp, nh = D.n_head_cols;
    const float p_drop = c->p_drop, keep_scale = 1.0f / (1.0f - p_drop);
    const unsigned long long seed = c->seed;
    void* stream = X.stream;
    void* side = (side_enabled && X.side) ? X.side : X.stream;
    const bool overlapped = side != stream;
    X.side = side;                      // (side stream switched off: nothing forks, the DwGroups see one stream)
    Exec& XS = X;                       // the DwGroups allocate (operand planes) from the same bump allocator
    const bool grouped = (dt == SS_BF16 || use_planes()) && dw_grouped;
    cur = c;
    const bool x3att = use_x3_attention(T);

    // ---- heads (architecture.py:82)
    const void* dh_t = dhead;
    if (dt != SS_F32) { void* t = X.alloc((size_t)M * nh * es); if (!X.dry) L_(timed(X, "cast", 0, (double)M * nh * (4 + es), stream, [&] { return ss_cast_f32(dhead, t, dt, (long long)M * nh, stream); })); dh_t = t; }
    if (!X.dry && stage_arena && stage_arena_bytes > 0) {        // staging buffers of the re-laid-out weight gradients (one memset, main stream)
        if (!ss_memset_async(stage_arena, 0, (size_t)stage_arena_bytes, stream)) { ss_set_error("backward: memset failed"); return 1; }
    }
    const int old_blocks = ss_gemm_set_blocks_per_cu(overlapped ? side_blocks : 2);
    struct Restore { int v; ~Restore() { ss_gemm_set_blocks_per_cu(v); } } restore{old_blocks};
    ss_gemm_set_blocks_per_cu(2);
    // launches on the side stream run with the reduced persistent grid (room for the main-stream kernels on every CU)
#define SIDE_BEGIN() do { fork(X); if (overlapped) ss_gemm_set_blocks_per_cu(side_blocks); } while (0)
#define SIDE_END() do { if (overlapped) ss_gemm_set_blocks_per_cu(2); } while (0)

    DwGroup grp{this, &XS, grouped};                 // head + last encoder layer travel together
    L_(grp.add(dh_t, c->x_final, head_w_stage, nh, d, M, RM(nh), RM(d), side));
    SIDE_BEGIN(); L_(colsum(X, dh_t, M, nh, head_b_stage, side)); SIDE_END();
    void* G = X.alloc((size_t)M * d * es);
    L_(gemm(X, dt, dh_t, w_head_T, G, M, d, nh, RM(nh), RM(nh), RM(d)));

    // scratch of the LayerNorm backward (per-workgroup column sums): one buffer, the launches are stream-ordered
    const int64_t ln_floats = ss_layernorm_backward_scratch_floats(M, d);
    float* ln_scratch = ln_floats ? (float*)X.alloc((size_t)ln_floats * 4) : nullptr;
    // per-layer gradient un-layout (7 small launches instead of 1) only when someone listens to the per-layer events (data parallel) or no combined table is bound
    const bool per_layer = on_event != nullptr || !unpack_all.jobs;
    // ---- encoder layers, last to first (transformer.py:54-59)
    for (int l = c->n_layers - 1; l >= 0; --l) {
        LayerP& w = layers[l]; LayerCtx& s = c->layer[l];
        // every layer gets its own temporaries (~340 MB at the reference batch): the side stream reads dF / dHid / dA / dqkv of a layer
        // (its grouped weight-gradient launch) while the main stream is already inside the next layer, so nothing is recycled before join()
        void* dF = X.alloc((size_t)M * d * es);
        // linear2.bias.grad = column sums of dF: accumulated by the LayerNorm backward kernel itself
        const Pl dFp = ln_floats ? new_planes(X, dF, (long long)M * d) : Pl{nullptr, nullptr};      // (parity-grade mode: dW2 and the FFN input-gradient GEMM take dF as planes)
        if (!X.dry) L_(timed(X, "ln_bwd", 0, (double)M * d * es * 4, stream, [&] { return ss_layernorm_backward_ws_planes(dt, G, s.z2, s.mean2, s.rstd2, w.g2, G, dF, dFp.hi, dFp.lo, w.dg2, w.dbe2, w.db2, ln_scratch, ln_floats, M, d, p_drop, seed, 4 * l + 3, stream); }));
        L_(grp.add(dF, s.hid, w.dw2, d, ff, M, RM(d), RM(ff), side));
        void* dHid = X.alloc((size_t)M * ff * es);
        bool db1_fused = false;
        { ss_gemm_epilogue e = EPI(); e.gate = s.hid; e.gate_scale = keep_scale;
          ss_rowmap am_ = RM(d), bm_ = RM(d), cm_ = RM(ff);
          // linear1.bias.grad = column sums of dHid: accumulated by this GEMM's epilogue when the 8-wave kernel runs the shape
          if (!X.dry && fuse_stats) {
              ss_gemm_epilogue es = e; es.col_sum = w.db1;
              if (s.hid_sign && !use_planes()) {          // the gate from the sign bits linear1's epilogue left (rides the column-sum epilogue of the 8-wave kernel)
                  ss_gemm_epilogue eb = es; eb.gate = nullptr; eb.gate_bits = s.hid_sign; eb.gate_bits_pitch = ff / 8;
                  if (ss_gemm_fuses_column_stats(dt, dt, SS_OP_KC, SS_OP_KC, dHid, M, ff, d, &am_, &bm_, &cm_, &eb, 1) != 0) { e = eb; db1_fused = true; }
              }
              if (db1_fused) {}
              else if (use_planes() ? ss_gemm_planes_supported(SS_F32, dHid, M, ff, d, &am_, &bm_, &cm_, &es) != 0 : ss_gemm_fuses_column_stats(dt, dt, SS_OP_KC, SS_OP_KC, dHid, M, ff, d, &am_, &bm_, &cm_, &e, 1) != 0) { e = es; db1_fused = true; }
          }
          L_(gemm(X, dt, dF, w.w2T, dHid, M, ff, d, am_, bm_, cm_, &e, SS_OP_KC, SS_OP_KC, 1, 0, nullptr, db1_fused ? 2 : 1)); }      // consumers: dW1 and the dX GEMM (planes); the f32 copy only feeds an unfused bias gradient
        L_(grp.add(dHid, s.y1, w.dw1, ff, d, M, RM(ff), RM(d), side));
        if (!db1_fused) { SIDE_BEGIN(); L_(colsum(X, dHid, M, ff, w.db1, side)); SIDE_END(); }
        { ss_gemm_epilogue e = EPI(); e.mode = 1; L_(gemm(X, dt, dHid, w.w1T, G, M, d, ff, RM(ff), RM(ff), RM(d), &e)); }
        void* dA = X.alloc((size_t)M * d * es);
        const Pl dAp = ln_floats ? new_planes(X, dA, (long long)M * d) : Pl{nullptr, nullptr};      // (dWo and the dO GEMM)
        if (!X.dry) L_(timed(X, "ln_bwd", 0, (double)M * d * es * 4, stream, [&] { return ss_layernorm_backward_ws_planes(dt, G, s.z1, s.mean1, s.rstd1, w.g1, G, dA, dAp.hi, dAp.lo, w.dg1, w.dbe1, nullptr, ln_scratch, ln_floats, M, d, p_drop, seed, 4 * l + 1, stream); }));
        // output projection  out[t,b,f] = sum_{h,a} o[b,h,t,a] w_o[h,a,f]   (transformer.py:111)
        L_(grp.add(dA, s.o, w.wo_stage, d, HD, M, RM(d), RM(HD), side));
        void* dO = X.alloc((size_t)M * HD * es);
        void* dOT = nullptr;
        if (c->need_T) {
            dOT = X.alloc((size_t)B * HD * Tp * es);
            ss_gemm_epilogue e = EPI(); e.c2 = dOT; e.cmap2 = RM(1, T, (long long)HD * Tp); e.col_stride2 = Tp;
            L_(gemm(X, dt, dA, w.woT, dO, M, HD, d, RM(d), RM(d), RM(HD), &e));
        } else L_(gemm(X, dt, dA, w.woT, dO, M, HD, d, RM(d), RM(d), RM(HD), nullptr, SS_OP_KC, SS_OP_KC, 1, 0, nullptr, x3att ? 2 : 0));      // x3 attention backward: dO as planes only
        void* dqkv = X.alloc((size_t)M * 3 * HD * es);
        float* dsc = (float*)X.alloc((size_t)B * H * T * 4);
        if (x3att) {
            const Pl q = planes(X, s.qkv, s.qkv, (long long)M * 3 * HD), ol = planes(X, s.o, s.o, (long long)M * HD), dop = planes(X, dO, dO, (long long)M * HD);
            adopt_planes(dqkv, dqkv, (long long)M * 3 * HD);
            if (!X.dry) {
                if (!q.hi || !ol.hi || !dop.hi) return 1;
                L_(timed(X, "attn_bwd", 2.0 * B * H * band_pairs(T, Dr) * dp * 5.0, (double)M * 8 * HD * es, stream, [&] { return ss_relpos_attention_x3_backward(q.hi, q.lo, w.EF, ol.hi, ol.lo, dop.hi, dop.lo, dsc, dqkv, (char*)dqkv + (size_t)M * 3 * HD * 2, s.pimg, B, H, T, dp, Dr, c->scale, p_drop, seed, 4 * l, stream); }));
            }
        } else
        if (!X.dry) L_(timed(X, "attn_bwd", 2.0 * B * H * band_pairs(T, Dr) * dp * 5.0, (double)M * 8 * HD * es, stream, [&] { return ss_relpos_attention_backward_p(dt == SS_F32 && f32_x3 ? SS_F32X3 : dt, s.qkv, s.qkvT, w.E, w.ET, w.EF, s.o, s.lse, dO, dOT, dsc, dqkv, s.pimg, B, H, T, Tp, dp, Dr, c->scale, p_drop, seed, 4 * l, stream); }));
        L_(grp.add(dqkv, s.x, w.wqkv_stage, 3 * HD, d, M, RM(3 * HD), RM(d), side));
        { ss_gemm_epilogue e = EPI(); e.mode = 1; L_(gemm(X, dt, dqkv, w.wqkvT, G, M, d, 3 * HD, RM(3 * HD), RM(3 * HD), RM(d), &e)); }
        if (l > 0) {                                                             // layer 0's group waits for w_raw_in's gradient
            // the layer's weight gradients, their un-layout into the .grad arena, and -- data-parallel runs -- the event that lets the caller
            // start THIS layer's all-reduce now (what = 4 + l): 6 collectives of ~28 MB spread over the encoder backward instead of one of
            // 170 MB after it.  (The fork makes `side` wait for everything the main stream produced for this layer: bias / LayerNorm gradients.)
            SIDE_BEGIN(); L_(grp.launch(side)); if (per_layer) L_(permute_batch(X, w.unpack, side)); SIDE_END();
            if (on_event && !X.dry) on_event(event_user, 4 + l, side);
        }
    }
    // ---- w_raw_in (architecture.py:73)
    L_(grp.add(G, c->conv_out, dw_raw_in, d, d, M, RM(d), RM(d), side));
    SIDE_BEGIN(); L_(grp.launch(side)); L_(colsum(X, G, M, d, db_raw_in, side));
    if (per_layer) {
        if (c->n_layers > 0) L_(permute_batch(X, layers[0].unpack, side));
        L_(permute_batch(X, unpack_enc, side));                                  // the fused heads: under the conv backward
    } else L_(permute_batch(X, unpack_all, side));                               // no listener for per-layer events: heads + all layers in ONE launch
    SIDE_END();
    if (on_event && !X.dry) {
        if (c->n_layers > 0) on_event(event_user, 4, side);                      // encoder layer 0
        on_event(event_user, 0, side);                                          // heads + w_raw_in: every non-convolutional gradient is final on `side`
    }
    void* dy = X.alloc((size_t)M * d * es);
    L_(gemm(X, dt, G, w_raw_in_T, dy, M, d, d, RM(d), RM(d), RM(d)));

    // ---- ResBlocks, last to first (architecture.py:29-40).  regate: the BatchNorm backward passes recompute the ReLU gate from the convolution
    // outputs they read anyway instead of reading the saved block output as well (option 4, SS_AMD_BN_REGATE)
    const bool regate = regate_on != 0;
    for (int i = 2; i >= 0; --i) {
        BlockP& w = blk[i]; BlockCtx& s = c->blk[i];
        const int O = s.O, Cin = s.Cin, Tin = s.Tin, Tout = s.Tout, rows = B * Tout;
        const long long pbs = (long long)(Tout + 2) * O;           // batch stride of a padded (B, Tout+2, O) buffer
        void* dc2 = X.alloc((size_t)B * (Tout + 2) * O * es);
        void* dcr = X.alloc((size_t)rows * O * es);
        float* sums = (float*)X.alloc((size_t)3 * O * 4);
        if (!X.dry) {
            L_(timed(X, "bn_bwd_sums", 0, (double)B * Tout * O * es * (regate ? 3 : 4), stream, [&] { return ss_bn_backward_sums(dt, dy, 0, s.y, s.pad_y, s.c2, 0, s.m2, s.i2, s.cr, 0, s.mr, s.ir, w.bn2.dgamma, w.bn2.dbeta, w.bnr.dgamma, w.bnr.dbeta, s.scratch, sums, B, Tout, O, 1, regate ? w.bn2.gamma : nullptr, regate ? w.bn2.beta : nullptr, regate ? w.bnr.gamma : nullptr, regate ? w.bnr.beta : nullptr, stream); }));
            double n_total = (double)B * Tout;
            if (hook) n_total = hook(hook_user, sums, 3 * O, n_total, stream);
            L_(timed(X, "bn_bwd_apply", 0, (double)B * Tout * O * es * (regate ? 5 : 6), stream, [&] { return ss_bn_backward_apply(dt, dy, 0, s.y, s.pad_y, s.c2, 0, s.m2, s.i2, w.bn2.gamma, s.cr, 0, s.mr, s.ir, w.bnr.gamma, sums, n_total, dc2, 1, dcr, 0, B, Tout, O, 1, regate ? w.bn2.beta : nullptr, regate ? w.bnr.beta : nullptr, stream); }));
        }
        // conv2 (k3, stride 1): weight and input gradients.  d/d(bias) of a conv feeding training-mode BatchNorm is identically 0
        DwGroup cg{this, &XS, grouped};
        L_(cg.add(dc2, s.h1, w.c2_stage, O, 3 * O, rows, RM(O, Tout, pbs, O), RM(O, Tout, pbs), side));
        void* dh1 = X.alloc((size_t)rows * O * es);
        L_(gemm(X, dt, dc2, w.w2b, dh1, rows, O, 3 * O, RM(O, Tout, pbs), RM(3 * O), RM(O)));
        void* dc1 = X.alloc((size_t)B * (Tout + 2) * O * es);
        float* sums1 = (float*)X.alloc((size_t)3 * O * 4);
        if (!X.dry) {
            L_(timed(X, "bn_bwd_sums", 0, (double)B * Tout * O * es * (regate ? 2 : 3), stream, [&] { return ss_bn_backward_sums(dt, dh1, 0, s.h1, 1, s.c1, 0, s.m1, s.i1, nullptr, 0, nullptr, nullptr, w.bn1.dgamma, w.bn1.dbeta, nullptr, nullptr, s.scratch, sums1, B, Tout, O, 1, regate ? w.bn1.gamma : nullptr, regate ? w.bn1.beta : nullptr, nullptr, nullptr, stream); }));
            double n_total = (double)B * Tout;
            if (hook) n_total = hook(hook_user, sums1, 3 * O, n_total, stream);
            L_(timed(X, "bn_bwd_apply", 0, (double)B * Tout * O * es * (regate ? 3 : 4), stream, [&] { return ss_bn_backward_apply(dt, dh1, 0, s.h1, 1, s.c1, 0, s.m1, s.i1, w.bn1.gamma, nullptr, 0, nullptr, nullptr, nullptr, sums1, n_total, dc1, 1, nullptr, 0, B, Tout, O, 1, regate ? w.bn1.beta : nullptr, nullptr, stream); }));
        }
        // conv1 (k3, stride 2) and the 1x1 stride-2 residual path
        const long long in_bs = (long long)(Tin + 2) * Cin;
        L_(cg.add(dc1, s.xin, w.c1_stage, O, 3 * Cin, rows, RM(O, Tout, pbs, O), RM(2 * Cin, Tout, in_bs), side));
        L_(cg.add(dcr, s.xin, w.wr_grad, O, Cin, rows, RM(O, Tout, (long long)Tout * O), RM(2 * Cin, Tout, in_bs, Cin), side));
        SIDE_BEGIN(); L_(cg.launch(side)); L_(permute_batch(X, w.unpack, side)); SIDE_END();      // this block's conv gradients -> parameter layout
        if (on_event && !X.dry) on_event(event_user, 1 + (2 - i), side);
        if (i > 0) {
            void* dx = X.alloc((size_t)B * Tin * Cin * es);
            const ss_rowmap out_even = RM(2 * Cin, Tout, (long long)Tin * Cin), out_odd = RM(2 * Cin, Tout, (long long)Tin * Cin, Cin);
            L_(gemm(X, dt, dc1, w.w1b_even, dx, rows, Cin, O, RM(O, Tout, pbs, O), RM(O), out_even));
            { ss_gemm_epilogue e = EPI(); e.mode = 1; L_(gemm(X, dt, dcr, w.wrT, dx, rows, Cin, O, RM(O), RM(O), out_even, &e)); }
            L_(gemm(X, dt, dc1, w.w1b_odd, dx, rows, Cin, 2 * O, RM(O, Tout, pbs, O), RM(2 * O), out_odd));
            dy = dx;
        }
    }
    join(X);
#undef SIDE_BEGIN
#undef SIDE_END
    c->ws_used = X.off;
    return 0;
}

}  // namespace

// ================================================================ C ABI
struct ss_plan { Plan* p; };

extern "C" ss_plan* ss_plan_create(const ss_model_dims* dims)
{
    if (!dims || dims->n_layers < 0 || dims->n_layers > MAX_LAYERS || dims->d_model % 8 || (dims->dtype != SS_F32 && dims->dtype != SS_BF16)) { ss_set_error("ss_plan_create: bad model dimensions"); return nullptr; }
    ss_plan* h = new ss_plan; h->p = new Plan(*dims);
#if !defined(SS_EMU)
    hipEventCreateWithFlags(&h->p->ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&h->p->ev_join, hipEventDisableTiming);
#endif
    return h;
}
extern "C" void ss_plan_destroy(ss_plan* h)
{
    if (!h) return;
#if !defined(SS_EMU)
    if (h->p->ev_fork) hipEventDestroy(h->p->ev_fork);
    if (h->p->ev_join) hipEventDestroy(h->p->ev_join);
#endif
    delete h->p; delete h;
}
extern "C" int ss_plan_slot_count(const ss_plan* h) { return h ? (int)h->p->names.size() : 0; }
extern "C" const char* ss_plan_slot_name(const ss_plan* h, int i) { return (h && i >= 0 && i < (int)h->p->names.size()) ? h->p->names[i].c_str() : nullptr; }
extern "C" int ss_plan_bind(ss_plan* h, int i, void* ptr)
{
    SS_CHECK(h && i >= 0 && i < (int)h->p->names.size(), "ss_plan_bind: bad slot %d", i);
    *h->p->targets[i] = ptr;
    return 0;
}
extern "C" int ss_plan_set_option(ss_plan* h, int what, int value)
{
    SS_CHECK(h, "ss_plan_set_option: null plan");
    int old = -1;
    if (what == 0) { old = h->p->side_enabled; h->p->side_enabled = value; }
    else if (what == 1) { old = h->p->dw_grouped; h->p->dw_grouped = value; }
    else if (what == 2) { old = h->p->side_blocks; h->p->side_blocks = value >= 1 && value <= 2 ? value : 2; }
    else if (what == 3) { old = h->p->fuse_stats; h->p->fuse_stats = value; }
    else if (what == 4) { old = h->p->regate_on; h->p->regate_on = value; }
    else if (what == 5) { old = h->p->f32_x3; h->p->f32_x3 = value != 0; }
    else if (what == 6) { old = h->p->keep_input; h->p->keep_input = value != 0; }
    else if (what == 7) { old = h->p->x3_planes; h->p->x3_planes = value != 0; }
    else if (what == 9) { old = h->p->x3_emit; h->p->x3_emit = value != 0; }
    else if (what == 10) { old = h->p->sign_gate; h->p->sign_gate = value != 0; }
    else if (what == 8) { old = h->p->x3_attn; h->p->x3_attn = value != 0; }
    return old;
}
extern "C" int ss_plan_set_reduce_hook(ss_plan* h, ss_reduce_hook fn, void* user) { SS_CHECK(h, "null plan"); h->p->hook = (reduce_fn)fn; h->p->hook_user = user; return 0; }
extern "C" int ss_plan_set_event_hook(ss_plan* h, ss_event_hook fn, void* user) { SS_CHECK(h, "null plan"); h->p->on_event = (event_fn)fn; h->p->event_user = user; return 0; }
extern "C" int64_t ss_plan_ctx_bytes(void) { return (int64_t)sizeof(Ctx); }

extern "C" int ss_plan_profile(ss_plan* h, int enable)
{
    SS_CHECK(h, "ss_plan_profile: null plan");
    const int old = h->p->profiling ? 1 : 0;
    h->p->profiling = enable != 0;
    return old;
}
// Aggregates (and clears) the per-launch records gathered since the last read; synchronises with the recorded events.
extern "C" int ss_plan_profile_read(ss_plan* h, ss_profile_row* rows, int max_rows)
{
    SS_CHECK(h && rows && max_rows > 0, "ss_plan_profile_read: bad arguments");
    int n = 0;
#if !defined(SS_EMU)
    Plan* P = h->p;
    static const char* gemm_names[5] = {"gemm_kernel (128x128, register-staged)", "gemm_glds_kernel (128x128)", "gemm_w2_kernel (128|144 x 128)", "gemm8_kc_kernel (256x256)", "gemm8_kc_kernel (288x256)"};
    for (const Plan::ProfRec& r : P->recs) {
        hipEventSynchronize(r.e1);
        float ms = 0.f; hipEventElapsedTime(&ms, r.e0, r.e1);
        const char* name = r.tag;
        double flops = r.flops;
        if (!strcmp(r.tag, "gemm") && r.sub >= 0 && r.sub < 5) name = gemm_names[r.sub];
        else if (!strcmp(r.tag, "gemm") && r.sub == 5) { name = "gemm_smallk_kernel (K <= 32, first conv)"; flops = 0; }      // an HBM-bound write of C: reported by its bytes
        else if (!strcmp(r.tag, "gemm_dw")) name = "gemm_kernel<OC,OC> (128x128 dW, split-K atomics)";
        int k = 0;
        for (; k < n; ++k) if (!strcmp(rows[k].name, name)) break;
        if (k == n) { if (n == max_rows) continue; memset(&rows[n], 0, sizeof(rows[n])); strncpy(rows[n].name, name, sizeof(rows[n].name) - 1); ++n; }
        rows[k].calls += 1; rows[k].seconds += ms * 1e-3; rows[k].flops += flops; rows[k].bytes += r.bytes;
    }
    P->recs.clear(); P->ev_used = 0;
#else
    (void)h; (void)rows; (void)max_rows;
#endif
    return n;
}

extern "C" int64_t ss_plan_workspace_bytes(ss_plan* h, int B, int T0, int training)
{
    if (!h) return -1;
    Exec X{true, nullptr, 0, 0, nullptr, nullptr};
    Ctx c; memset(&c, 0, sizeof(c));
    if (h->p->forward(X, nullptr, nullptr, B, T0, training, 0, 0.f, 0, nullptr, &c)) return -1;
    if (training && h->p->backward(X, &c, nullptr)) return -1;
    return (int64_t)((X.off + 255) & ~(size_t)255) + 256;
}

extern "C" int ss_plan_forward(ss_plan* h, const float* x_raw, float* shifted_scratch, void* workspace, int64_t workspace_bytes, int B, int T0, int training, int shift_r,
                               float dropout_p, uint64_t seed, float* head, void* ctx_out, void* stream)
{
    SS_CHECK(h && x_raw && workspace && head && ctx_out, "ss_plan_forward: null pointer");
    for (size_t i = 0; i < h->p->targets.size(); ++i) {
        const std::string& n = h->p->names[i];
        const bool optional = n.find(".grad") != std::string::npos || n.find(".stage") != std::string::npos || n.find("unpack") != std::string::npos || n.find("stage_arena") != std::string::npos ||
                              n.find("num_batches_tracked") != std::string::npos;
        SS_CHECK(*h->p->targets[i] || optional, "ss_plan_forward: slot '%s' is not bound", n.c_str());
    }
    Exec X{false, (char*)workspace, 0, (size_t)workspace_bytes, stream, nullptr};
    Ctx* c = (Ctx*)ctx_out; memset(c, 0, sizeof(Ctx));
    c->ws = (char*)workspace; c->ws_bytes = (unsigned long long)workspace_bytes;
    const int64_t need = ss_plan_workspace_bytes(h, B, T0, training);
    SS_CHECK(need >= 0 && need <= workspace_bytes, "ss_plan_forward: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, (long long)need);
    return h->p->forward(X, x_raw, shifted_scratch, B, T0, training, shift_r, dropout_p, seed, head, c);
}

extern "C" int ss_plan_backward(ss_plan* h, void* ctx, const float* dhead, void* stream, void* side_stream)
{
    SS_CHECK(h && ctx && dhead, "ss_plan_backward: null pointer");
    Ctx* c = (Ctx*)ctx;
    SS_CHECK(c->ws && c->M > 0, "ss_plan_backward: context of an eval-mode or failed forward");
    for (size_t i = 0; i < h->p->targets.size(); ++i) SS_CHECK(*h->p->targets[i] || h->p->names[i].find("w1b_") != std::string::npos, "ss_plan_backward: slot '%s' is not bound", h->p->names[i].c_str());
    Exec X{false, c->ws, (size_t)c->ws_used, (size_t)c->ws_bytes, stream, side_stream};
    return h->p->backward(X, c, dhead);
}

// Sub-layer executor (host code only): one C-ABI call enqueues every kernel of an attention sub-layer, a feed-forward
// sub-layer or a stack of DETR decoder layers - forward or backward - on the caller's streams.
//
// Why: a training step is ~1 150 launches.  Issued one by one from Python they cost ~17 us each (20 ms of host time per
// 26 ms step), so the chains of small kernels (decoders, language layers, hyper-graph encoder) run at the
// host's pace.  Here a launch costs the ~4 us of the HIP runtime and nothing else.
//
// The arithmetic is exactly the entry points of include/shg_vqa.h in the order the reference's modules apply them:
//   attention sub-layer  BertAttention + BertAttOutput (modeling_capsbert.py:384-435), nn.MultiheadAttention + dropout + norm
//                        of the DETR decoder layer (transformer.py:216-227)
//   feed-forward         BertIntermediate + BertOutput (modeling_capsbert.py:463-489), linear1/ReLU/dropout/linear2/norm3
//                        (transformer.py:230-232)
//   decoder              TransformerDecoder.forward over forward_post layers (transformer.py:86-124, :212-233)
// Weight gradients go to run->wgrad_stream behind an event recorded on the main stream right after their operands were
// enqueued there (dW only feeds the optimiser; the input-gradient chain does not wait for it).
#include <algorithm>
#include <vector>

#include "common.h"

namespace shg {

struct ColsumJob {
    const void* x;
    int64_t rows, cols, ld;
    float* out;
};
struct FinishJob {
    const float* parts[3];
    float* outs[3];
    int n, n_part, cols;
};
struct Exec {
    std::vector<FinishJob> fq;
    std::vector<hipEvent_t> events;
    size_t cursor = 0;
    // deferred weight gradients (shg_run_t.defer_wgrad): problems, bias column sums, and the streams their operands come from
    std::vector<shg_wgrad_problem_t> wq;
    std::vector<ColsumJob> cq;
    std::vector<void*> producers;
    int64_t pending_tiles = 0;
    int q_dtype = -1;          // storage type of the queued operands (-1: queues empty); one type per flush
};

static inline int64_t al256(int64_t n) { return (n + 255) / 256 * 256; }
static inline int esize(int dtype) { return dtype == SHG_BF16 ? 2 : 4; }

// bump allocator over a caller-owned buffer (offsets only: the same walk sizes the buffer and hands out the pieces)
struct Carve {
    char* base;
    int64_t off = 0;
    explicit Carve(void* p) : base((char*)p) {}
    void* take(int64_t bytes) {
        void* p = base ? base + off : nullptr;
        off += al256(bytes);
        return p;
    }
};

#define CK(call)                \
    do {                        \
        if (int e_ = (call)) return e_; \
    } while (0)

static int run_check(const shg_run_t* R) {
    if (!R) return fail_arg("executor: null run context");
    if (R->dtype != SHG_F32 && R->dtype != SHG_BF16) return fail_arg("executor: bad dtype");
    if (R->wgrad_stream && !R->exec) return fail_arg("executor: a weight-gradient stream needs an shg_exec_t");
    return 0;
}

// orders the weight-gradient stream behind everything enqueued on the main stream so far; returns the stream to launch on
static int fork_wgrad(const shg_run_t* R, void** out) {
    if (!R->wgrad_stream || R->wgrad_stream == R->stream) {
        *out = R->stream;
        return 0;
    }
    Exec* ex = reinterpret_cast<Exec*>(R->exec);
    if (ex->events.empty()) return fail_arg("executor: shg_exec_t has no events");
    hipEvent_t ev = ex->events[ex->cursor];
    ex->cursor = (ex->cursor + 1) % ex->events.size();
    hipError_t e = hipEventRecord(ev, (hipStream_t)R->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)R->wgrad_stream, ev, 0);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return (int)e;
    }
    *out = R->wgrad_stream;
    return 0;
}

static void note_producer(Exec* ex, void* stream) {
    for (void* p : ex->producers)
        if (p == stream) return;
    ex->producers.push_back(stream);
}
// a queue entry carries no type of its own: the queues hold ONE storage type between two flushes
static int queue_dtype(Exec* ex, int dtype) {
    if (ex->q_dtype >= 0 && ex->q_dtype != dtype)
        return fail_arg("executor: operand dtype changed while weight gradients are queued (flush them first)");
    ex->q_dtype = dtype;
    return 0;
}

// (shg_colsum_accumulate takes at most 512 16-byte chunks per row: wider outputs go in column slices)
static int colsum_sliced(const void* dy, int dtype, int64_t rows, int64_t n_out, int64_t ldy, float* gb, void* st) {
    const int64_t es = esize(dtype), max_cols = 512 * (16 / es);
    for (int64_t c0 = 0; c0 < n_out; c0 += max_cols) {
        const int64_t nc = n_out - c0 < max_cols ? n_out - c0 : max_cols;
        CK(shg_colsum_accumulate((const char*)dy + c0 * es, dtype, rows, (int)nc, ldy, gb + c0, st));
    }
    return 0;
}

// dW += dy^T x ; db += colsum(dy).   dy [rows, n_out] (row stride ldy), x [rows, n_in] (row stride ldx)
static int wgrad(const shg_run_t* R, const shg_linear_t& lin, const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t rows,
                 int64_t n_out, int64_t n_in, bool with_bias) {
    const bool want_b = with_bias && lin.gb;
    // the calling stream is a producer of gradient slices even when nothing is queued here (a bias gradient written by the
    // input-gradient GEMM's epilogue on this stream, with the weight frozen): the flush - and the data-parallel reducer
    // behind it - must be ordered after it
    if (R->defer_wgrad && R->wgrad_stream && R->exec) note_producer(reinterpret_cast<Exec*>(R->exec), R->stream);
    if (!lin.gw && !want_b) return 0;
    if (R->defer_wgrad && R->wgrad_stream && R->exec) {          // queued: shg_exec_flush_wgrads issues them grouped
        Exec* ex = reinterpret_cast<Exec*>(R->exec);
        CK(queue_dtype(ex, R->dtype));
        if (lin.gw) {
            ex->wq.push_back(shg_wgrad_problem_t{dy, x, lin.gw, rows, n_out, n_in, ldy, ldx});
            ex->pending_tiles += ((n_out + 255) / 256) * ((n_in + 255) / 256);
        }
        if (want_b) ex->cq.push_back(ColsumJob{dy, rows, n_out, ldy, lin.gb});
        note_producer(ex, R->stream);
        return 0;
    }
    void* st = nullptr;
    CK(fork_wgrad(R, &st));
    if (lin.gw) CK(shg_gemm(dy, x, lin.gw, nullptr, R->dtype, SHG_F32, n_out, n_in, rows, ldy, ldx, n_in, 0, 0, 1, st));
    if (want_b) CK(colsum_sliced(dy, R->dtype, rows, n_out, ldy, lin.gb, st));
    return 0;
}

// ------------------------------------------------------------------------------------------------ attention sub-layer
struct AttnSaved {
    void *qkv, *kv, *o, *z, *t;
    float *lse, *mean, *rstd;
    uint64_t* keep;            // the forward's dropout lane masks (shg_attention_keep_mask_bytes)
    int64_t bytes;
};
static AttnSaved attn_saved(void* base, int mode, int dtype, int B, int Sq, int Sk, int heads) {
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Sq, rk = (int64_t)B * Sk, es = esize(dtype);
    const bool fused3 = mode == SHG_ATTN_SELF || mode == SHG_ATTN_DEC_SELF;
    Carve c(base);
    AttnSaved s{};
    s.qkv = c.take(rq * (fused3 ? 3 * H : H) * es);
    s.kv = fused3 ? nullptr : c.take(rk * 2 * H * es);
    s.o = c.take(rq * H * es);
    s.z = c.take(rq * H * es);
    s.t = c.take(rq * H * es);                       // forward temporary (out-projection before the LayerNorm)
    s.lse = (float*)c.take((int64_t)B * heads * Sq * 4);
    s.mean = (float*)c.take(rq * 4);
    s.rstd = (float*)c.take(rq * 4);
    s.keep = (uint64_t*)c.take(shg_attention_keep_mask_bytes(B, heads, Sq, Sk));
    s.bytes = c.off;
    return s;
}
struct AttnScratch {
    void *dt, *d_o, *dqkv, *dkv;
    float *delta, *parts;
    int n_part;
    int64_t bytes;
};
static AttnScratch attn_scratch(void* base, int mode, int dtype, int B, int Sq, int Sk, int heads) {
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Sq, rk = (int64_t)B * Sk, es = esize(dtype);
    const bool fused3 = mode == SHG_ATTN_SELF || mode == SHG_ATTN_DEC_SELF;
    Carve c(base);
    AttnScratch s{};
    s.dt = c.take(rq * H * es);
    s.d_o = c.take(rq * H * es);
    s.dqkv = c.take(rq * (fused3 ? 3 * H : H) * es);
    s.dkv = fused3 ? nullptr : c.take(rk * 2 * H * es);
    s.delta = (float*)c.take((int64_t)B * heads * Sq * 4);
    s.n_part = shg_colsum_partials(rq);
    s.parts = (float*)c.take((int64_t)3 * s.n_part * H * 4);
    s.bytes = c.off;
    return s;
}

static int attn_args_ok(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk) {
    CK(run_check(R));
    if (!L) return fail_arg("attn_sublayer: null descriptor");
    if (L->mode < 0 || L->mode > 3) return fail_arg("attn_sublayer: bad mode");
    if (B < 1 || Sq < 1 || Sk < 1 || L->heads < 1) return fail_arg("attn_sublayer: bad sizes");
    if ((L->mode == SHG_ATTN_SELF || L->mode == SHG_ATTN_DEC_SELF) && Sk != Sq) return fail_arg("attn_sublayer: self-attention needs Sk == Sq");
    if (!L->a.w || !L->o.w || !L->ln.gamma || !L->ln.beta) return fail_arg("attn_sublayer: missing parameter");
    if (L->mode != SHG_ATTN_SELF && !L->b.w) return fail_arg("attn_sublayer: missing second projection");
    return 0;
}

// LayerNorm gamma / beta (+ bias) gradients: second stage of the column sums.  Only the optimiser reads them, so the launch
// goes to the weight-gradient stream (or its queue) and the dependent chain on the main stream does not wait for it.
static int finish_ln_grads(const shg_run_t* R, const shg_norm_t& ln, float* g_bias, const float* parts, int n_part, int cols) {
    FinishJob f{};
    if (ln.g_gamma) {
        f.parts[f.n] = parts; f.outs[f.n++] = ln.g_gamma;
        f.parts[f.n] = parts + (int64_t)n_part * cols; f.outs[f.n++] = ln.g_beta;
    }
    if (g_bias) { f.parts[f.n] = parts + (int64_t)2 * n_part * cols; f.outs[f.n++] = g_bias; }
    if (!f.n) return 0;
    f.n_part = n_part;
    f.cols = cols;
    if (R->defer_wgrad && R->wgrad_stream && R->exec) {
        Exec* ex = reinterpret_cast<Exec*>(R->exec);
        note_producer(ex, R->stream);
        ex->fq.push_back(f);                         // (fp32 partial sums: no operand type involved)
        return 0;
    }
    void* st = nullptr;
    CK(fork_wgrad(R, &st));
    return shg_colsum_finish_multi(f.parts, f.outs, f.n, f.n_part, f.cols, st);
}

static int attn_fwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x, const void* xpos,
                    const void* mem, void* y, const void* pos, void* y_pos, void* saved, uint64_t sid, bool kv_done = false) {
    const int mode = L->mode, dt = R->dtype, heads = L->heads;
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Sq, rk = (int64_t)B * Sk, es = esize(dt);
    const float pa = R->training ? L->p_attn : 0.f, po = R->training ? L->p_out : 0.f;
    void* st = R->stream;
    AttnSaved s = attn_saved(saved, mode, dt, B, Sq, Sk, heads);
    const char *q, *k, *v;
    int64_t qb, qs, kb, ks;
    if (mode == SHG_ATTN_SELF) {
        CK(shg_gemm(x, L->a.w, s.qkv, L->a.bias, dt, dt, rq, 3 * H, H, H, H, 3 * H, 1, 1, 0, st));
    } else if (mode == SHG_ATTN_DEC_SELF) {
        CK(shg_gemm(xpos, L->a.w, s.qkv, L->a.bias, dt, dt, rq, 2 * H, H, H, H, 3 * H, 1, 1, 0, st));
        CK(shg_gemm(x, L->b.w, (char*)s.qkv + 2 * H * es, L->b.bias, dt, dt, rq, H, H, H, H, 3 * H, 1, 1, 0, st));
    } else {
        CK(shg_gemm(mode == SHG_ATTN_DEC_CROSS ? xpos : x, L->a.w, s.qkv, L->a.bias, dt, dt, rq, H, H, H, H, H, 1, 1, 0, st));
        if (!kv_done) CK(shg_gemm(mem, L->b.w, s.kv, L->b.bias, dt, dt, rk, 2 * H, H, H, H, 2 * H, 1, 1, 0, st));
    }
    if (mode == SHG_ATTN_SELF || mode == SHG_ATTN_DEC_SELF) {
        q = (const char*)s.qkv; k = q + H * es; v = q + 2 * H * es;
        qb = (int64_t)Sq * 3 * H; qs = 3 * H; kb = qb; ks = qs;
    } else {
        q = (const char*)s.qkv; k = (const char*)s.kv; v = k + H * es;
        qb = (int64_t)Sq * H; qs = H; kb = (int64_t)Sk * 2 * H; ks = 2 * H;
    }
    CK(shg_attention_fwd(q, k, v, s.o, s.lse, dt, B, heads, Sq, Sk, qb, qs, kb, ks, kb, ks, L->mask_kind, L->mask, L->scale, pa,
                         R->seed_state, sid, s.keep, st));
    CK(shg_gemm(s.o, L->o.w, s.t, nullptr, dt, dt, rq, H, H, H, H, H, 1, 1, 0, st));
    CK(shg_bias_act_drop_res_ln_fwd_pos(s.t, L->o.bias, x, L->ln.gamma, L->ln.beta, y, s.z, s.mean, s.rstd, pos, y_pos, dt, rq, (int)H,
                                        SHG_ACT_NONE, L->ln.eps, po, R->seed_state, sid + 1, st));
    return 0;
}

static int attn_bwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x, const void* xpos,
                    const void* mem, const void* saved, const void* dy, void* dx, void* dxpos, void* dmem, int dmem_acc,
                    void* scratch, uint64_t sid) {
    const int mode = L->mode, dt = R->dtype, heads = L->heads;
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Sq, rk = (int64_t)B * Sk, es = esize(dt);
    const float pa = R->training ? L->p_attn : 0.f, po = R->training ? L->p_out : 0.f;
    void* st = R->stream;
    const AttnSaved s = attn_saved(const_cast<void*>(saved), mode, dt, B, Sq, Sk, heads);
    const AttnScratch w = attn_scratch(scratch, mode, dt, B, Sq, Sk, heads);
    float* dbi = L->o.gb ? w.parts + (int64_t)2 * w.n_part * H : nullptr;
    // LayerNorm backward: dt = gradient of the out-projection's output (dropout applied), dx = residual gradient
    CK(shg_bias_act_drop_res_ln_bwd(dy, s.z, nullptr, L->o.bias, L->ln.gamma, s.mean, s.rstd, w.dt, dx, w.parts,
                                    w.parts + (int64_t)w.n_part * H, dbi, w.n_part, dt, rq, (int)H, SHG_ACT_NONE, po, R->seed_state,
                                    sid + 1, st));
    CK(finish_ln_grads(R, L->ln, L->o.gb, w.parts, w.n_part, (int)H));
    CK(wgrad(R, L->o, w.dt, H, s.o, H, rq, H, H, false));
    CK(shg_gemm(w.dt, L->o.w, w.d_o, nullptr, dt, dt, rq, H, H, H, H, H, 1, 0, 0, st));
    if (mode == SHG_ATTN_SELF || mode == SHG_ATTN_DEC_SELF) {
        const char* q = (const char*)s.qkv;
        char* dq = (char*)w.dqkv;
        const int64_t qb = (int64_t)Sq * 3 * H, qs = 3 * H;
        // the bias gradients of the projections are column sums of dq / dk / dv: the attention backward kernels add them
        // (self: a = [q; k; v]; decoder self: a = [q; k] of tgt + pos, b = v of tgt)
        float* gbq = L->a.gb;
        float* gbk = L->a.gb ? L->a.gb + H : nullptr;
        float* gbv = mode == SHG_ATTN_SELF ? (L->a.gb ? L->a.gb + 2 * H : nullptr) : L->b.gb;
        CK(shg_attention_bwd(q, q + H * es, q + 2 * H * es, s.o, w.d_o, s.lse, w.delta, dq, dq + H * es, dq + 2 * H * es, dt, B, heads,
                             Sq, Sk, qb, qs, qb, qs, qb, qs, qb, qs, qb, qs, qb, qs, L->mask_kind, L->mask, L->scale, pa,
                             R->seed_state, sid, s.keep, gbq, gbk, gbv, st));
        if (mode == SHG_ATTN_SELF) {
            CK(wgrad(R, L->a, dq, 3 * H, x, H, rq, 3 * H, H, false));
            if (dx) CK(shg_gemm(dq, L->a.w, dx, nullptr, dt, dt, rq, H, 3 * H, 3 * H, H, H, 1, 0, 1, st));
        } else {
            const char* dv = dq + 2 * H * es;
            CK(wgrad(R, L->a, dq, 3 * H, xpos, H, rq, 2 * H, H, false));
            CK(wgrad(R, L->b, dv, 3 * H, x, H, rq, H, H, false));
            if (dxpos) CK(shg_gemm(dq, L->a.w, dxpos, nullptr, dt, dt, rq, H, 2 * H, 3 * H, H, H, 1, 0, 0, st));
            if (dx) CK(shg_gemm(dv, L->b.w, dx, nullptr, dt, dt, rq, H, H, 3 * H, H, H, 1, 0, 1, st));
        }
    } else {
        const char* k = (const char*)s.kv;
        char* dk = (char*)w.dkv;
        const int64_t qb = (int64_t)Sq * H, qs = H, kb = (int64_t)Sk * 2 * H, ks = 2 * H;
        // (cross: a = q of the queries, b = [k; v] of the memory)
        CK(shg_attention_bwd(s.qkv, k, k + H * es, s.o, w.d_o, s.lse, w.delta, w.dqkv, dk, dk + H * es, dt, B, heads, Sq, Sk, qb, qs,
                             kb, ks, kb, ks, qb, qs, kb, ks, kb, ks, L->mask_kind, L->mask, L->scale, pa, R->seed_state, sid, s.keep,
                             L->a.gb, L->b.gb, L->b.gb ? L->b.gb + H : nullptr, st));
        CK(wgrad(R, L->a, w.dqkv, H, mode == SHG_ATTN_DEC_CROSS ? xpos : x, H, rq, H, H, false));
        CK(wgrad(R, L->b, w.dkv, 2 * H, mem, H, rk, 2 * H, H, false));
        if (mode == SHG_ATTN_CROSS) {
            if (dx) CK(shg_gemm(w.dqkv, L->a.w, dx, nullptr, dt, dt, rq, H, H, H, H, H, 1, 0, 1, st));
        } else if (dxpos) {
            CK(shg_gemm(w.dqkv, L->a.w, dxpos, nullptr, dt, dt, rq, H, H, H, H, H, 1, 0, 0, st));
        }
        if (dmem) CK(shg_gemm(w.dkv, L->b.w, dmem, nullptr, dt, dt, rk, H, 2 * H, 2 * H, H, H, 1, 0, dmem_acc ? 1 : 0, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ feed-forward sub-layer
struct FfnSaved {
    void *pre, *h, *z, *t;
    float *mean, *rstd;
    int64_t bytes;
};
static FfnSaved ffn_saved(void* base, int dtype, int64_t rows, int H, int F) {
    const int64_t es = esize(dtype);
    Carve c(base);
    FfnSaved s{};
    s.pre = c.take(rows * F * es);
    s.h = c.take(rows * F * es);
    s.z = c.take(rows * H * es);
    s.t = c.take(rows * H * es);
    s.mean = (float*)c.take(rows * 4);
    s.rstd = (float*)c.take(rows * 4);
    s.bytes = c.off;
    return s;
}
struct FfnScratch {
    void *dt, *dpre;
    float* parts;
    int n_part;
    int64_t bytes;
};
static FfnScratch ffn_scratch(void* base, int dtype, int64_t rows, int H, int F) {
    const int64_t es = esize(dtype);
    Carve c(base);
    FfnScratch s{};
    s.dt = c.take(rows * H * es);
    s.dpre = c.take(rows * F * es);
    s.n_part = shg_colsum_partials(rows);
    s.parts = (float*)c.take((int64_t)3 * s.n_part * H * 4);
    s.bytes = c.off;
    return s;
}

static int ffn_args_ok(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F) {
    CK(run_check(R));
    if (!L) return fail_arg("ffn_sublayer: null descriptor");
    if (rows < 1 || H < 8 || F < 8) return fail_arg("ffn_sublayer: bad sizes");
    if (!L->l1.w || !L->l2.w || !L->ln.gamma || !L->ln.beta) return fail_arg("ffn_sublayer: missing parameter");
    return 0;
}

// (forward and backward of a call pair must agree: training / dtype / dropout setting do not change in between)
static bool ffn_saves_grad(const shg_ffn_sublayer_t* L, const shg_run_t* R) {
    return R->dtype == SHG_BF16 && L->act == SHG_ACT_GELU && !(R->training && L->p_inner > 0.f);
}

static int ffn_fwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x, void* y,
                   const void* pos, void* y_pos, void* saved, uint64_t sid) {
    const int dt = R->dtype;
    const float pi = R->training ? L->p_inner : 0.f, po = R->training ? L->p_out : 0.f;
    void* st = R->stream;
    FfnSaved s = ffn_saved(saved, dt, rows, H, F);
    // bf16, GELU, no inner dropout (the BERT blocks): `pre` keeps the GELU's derivative, computed beside the GELU (ffn_bwd below)
    const int act_fwd = ffn_saves_grad(L, R) ? (L->act | SHG_ACT_SAVE_GRAD) : L->act;
    CK(shg_gemm_act(x, L->l1.w, s.h, L->l1.bias, dt, dt, rows, F, H, H, H, F, 1, 1, act_fwd, s.pre, pi, R->seed_state, sid, st));
    CK(shg_gemm(s.h, L->l2.w, s.t, nullptr, dt, dt, rows, H, F, F, F, H, 1, 1, 0, st));
    CK(shg_bias_act_drop_res_ln_fwd_pos(s.t, L->l2.bias, x, L->ln.gamma, L->ln.beta, y, s.z, s.mean, s.rstd, pos, y_pos, dt, rows, H,
                                        SHG_ACT_NONE, L->ln.eps, po, R->seed_state, sid + 1, st));
    return 0;
}

static int ffn_bwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x, const void* saved,
                   const void* dy, void* dx, void* scratch, uint64_t sid) {
    const int dt = R->dtype;
    const float pi = R->training ? L->p_inner : 0.f, po = R->training ? L->p_out : 0.f;
    void* st = R->stream;
    const FfnSaved s = ffn_saved(const_cast<void*>(saved), dt, rows, H, F);
    const FfnScratch w = ffn_scratch(scratch, dt, rows, H, F);
    float* dbi = L->l2.gb ? w.parts + (int64_t)2 * w.n_part * H : nullptr;
    CK(shg_bias_act_drop_res_ln_bwd(dy, s.z, nullptr, L->l2.bias, L->ln.gamma, s.mean, s.rstd, w.dt, dx, w.parts,
                                    w.parts + (int64_t)w.n_part * H, dbi, w.n_part, dt, rows, H, SHG_ACT_NONE, po, R->seed_state, sid + 1,
                                    st));
    CK(finish_ln_grads(R, L->ln, L->l2.gb, w.parts, w.n_part, H));
    CK(wgrad(R, L->l2, w.dt, H, s.h, F, rows, H, F, false));
    // activation (and inner dropout) backward + linear1's bias gradient in the input-gradient GEMM's epilogue
    CK(shg_gemm_dact(w.dt, L->l2.w, w.dpre, s.pre, L->l1.gb, dt, rows, F, H, H, F, F, ffn_saves_grad(L, R) ? SHG_ACT_SAVED_GRAD : L->act, pi,
                     R->seed_state, sid, st));
    CK(wgrad(R, L->l1, w.dpre, F, x, H, rows, F, H, false));
    if (dx) CK(shg_gemm(w.dpre, L->l1.w, dx, nullptr, dt, dt, rows, H, F, F, H, H, 1, 0, 1, st));      // dx = residual gradient + dpre W1
    return 0;
}

// ------------------------------------------------------------------------------------------------ decoder stack
struct DecLayerBufs {
    void *s_self, *s_cross, *s_ffn;     // saved areas of the three sub-layers
    void *y1, *y1p, *y2, *y3, *y3p;     // sub-layer outputs (+ pos where the next projection wants it)
};
struct DecSaved {
    void *zero, *xp0;
    std::vector<DecLayerBufs> layer;
    int64_t bytes;
};
static DecSaved dec_saved(void* base, int n_layers, int dtype, int B, int Q, int S, int heads, int F) {
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Q, es = esize(dtype);
    Carve c(base);
    DecSaved d;
    d.zero = c.take(rq * H * es);        // (both always carved: the layout must not depend on the arguments of one call)
    d.xp0 = c.take(rq * H * es);
    d.layer.resize(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        DecLayerBufs& b = d.layer[i];
        b.s_self = c.take(attn_saved(nullptr, SHG_ATTN_DEC_SELF, dtype, B, Q, Q, heads).bytes);
        b.s_cross = c.take(attn_saved(nullptr, SHG_ATTN_DEC_CROSS, dtype, B, Q, S, heads).bytes);
        b.s_ffn = c.take(ffn_saved(nullptr, dtype, rq, (int)H, F).bytes);
        b.y1 = c.take(rq * H * es);
        b.y1p = c.take(rq * H * es);
        b.y2 = c.take(rq * H * es);
        b.y3 = c.take(rq * H * es);
        b.y3p = c.take(rq * H * es);
    }
    d.bytes = c.off;
    return d;
}
struct DecScratchLayer {
    void *w_self, *w_cross, *w_ffn;
    void *d2, *d1, *d0, *dxp_a, *dxp_b;  // gradients w.r.t. y2, y1, the layer input; the two `x + pos` gradients
};
struct DecScratch {
    std::vector<DecScratchLayer> layer;
    int64_t bytes;
};
static DecScratch dec_scratch(void* base, int n_layers, int dtype, int B, int Q, int S, int heads, int F) {
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Q, es = esize(dtype);
    Carve c(base);
    DecScratch d;
    d.layer.resize(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        DecScratchLayer& b = d.layer[i];
        b.w_self = c.take(attn_scratch(nullptr, SHG_ATTN_DEC_SELF, dtype, B, Q, Q, heads).bytes);
        b.w_cross = c.take(attn_scratch(nullptr, SHG_ATTN_DEC_CROSS, dtype, B, Q, S, heads).bytes);
        b.w_ffn = c.take(ffn_scratch(nullptr, dtype, rq, (int)H, F).bytes);
        b.d2 = c.take(rq * H * es);
        b.d1 = c.take(rq * H * es);
        b.d0 = c.take(rq * H * es);
        b.dxp_a = c.take(rq * H * es);
        b.dxp_b = c.take(rq * H * es);
    }
    d.bytes = c.off;
    return d;
}

static int dec_args_ok(const shg_decoder_layer_t* layers, int n_layers, const shg_run_t* R, int B, int Q, int S, int F) {
    CK(run_check(R));
    if (!layers || n_layers < 1 || n_layers > 64) return fail_arg("decoder: bad layer table");
    if (B < 1 || Q < 1 || S < 1 || F < 8) return fail_arg("decoder: bad sizes");
    for (int i = 0; i < n_layers; ++i) {
        if (layers[i].self_attn.mode != SHG_ATTN_DEC_SELF || layers[i].cross_attn.mode != SHG_ATTN_DEC_CROSS)
            return fail_arg("decoder: layer descriptors must use the decoder attention modes");
        if (layers[i].self_attn.heads != layers[0].self_attn.heads || layers[i].cross_attn.heads != layers[0].self_attn.heads)
            return fail_arg("decoder: all layers must share one width");
        CK(attn_args_ok(&layers[i].self_attn, R, B, Q, Q));
        CK(attn_args_ok(&layers[i].cross_attn, R, B, Q, S));
        CK(ffn_args_ok(&layers[i].ffn, R, (int64_t)B * Q, layers[0].self_attn.heads * 64, F));
    }
    return 0;
}

}  // namespace shg

using namespace shg;

extern "C" int shg_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(shg_run_t);
        case 1: return (int)sizeof(shg_linear_t);
        case 2: return (int)sizeof(shg_norm_t);
        case 3: return (int)sizeof(shg_attn_sublayer_t);
        case 4: return (int)sizeof(shg_ffn_sublayer_t);
        case 5: return (int)sizeof(shg_decoder_layer_t);
        default: return -1;
    }
}

extern "C" shg_exec_t* shg_exec_create(int n_events) {
    if (n_events < 1 || n_events > 4096) {
        set_error("exec_create: n_events must be in [1, 4096]");
        return nullptr;
    }
    Exec* ex = new Exec();
    ex->events.resize(n_events);
    for (int i = 0; i < n_events; ++i) {
        if (hipEventCreateWithFlags(&ex->events[i], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(ex->events[j]);
            delete ex;
            set_error("exec_create: hipEventCreateWithFlags failed");
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return reinterpret_cast<shg_exec_t*>(ex);
}

extern "C" int64_t shg_exec_pending_tiles(const shg_exec_t* h) {
    const Exec* ex = reinterpret_cast<const Exec*>(h);
    return ex ? ex->pending_tiles : 0;
}

extern "C" int shg_exec_flush_wgrads(shg_exec_t* h, int dtype, void* wgrad_stream) {
    Exec* ex = reinterpret_cast<Exec*>(h);
    if (!ex) return fail_arg("exec_flush_wgrads: null handle");
    if (ex->wq.empty() && ex->cq.empty() && ex->fq.empty()) {
        ex->producers.clear();
        ex->q_dtype = -1;
        return 0;
    }
    // whatever happens below, the queues are emptied before this call returns: their entries point into buffers the caller
    // only keeps alive until the flush
    int rc = 0;
    if (!wgrad_stream) rc = fail_arg("exec_flush_wgrads: deferred weight gradients need the weight-gradient stream");
    else if (ex->events.empty()) rc = fail_arg("exec_flush_wgrads: shg_exec_t has no events");
    else if (ex->q_dtype >= 0 && ex->q_dtype != dtype) rc = fail_arg("exec_flush_wgrads: dtype differs from the queued operands' type");
    // sort: problems with equal row counts next to each other (a group shares its K length best), largest first
    std::stable_sort(ex->wq.begin(), ex->wq.end(), [](const shg_wgrad_problem_t& a, const shg_wgrad_problem_t& b) { return a.rows > b.rows; });
    for (size_t i = 0; rc == 0 && i < ex->producers.size(); ++i) {
        void* p = ex->producers[i];
        if (p == wgrad_stream) continue;
        hipEvent_t ev = ex->events[ex->cursor];
        ex->cursor = (ex->cursor + 1) % ex->events.size();
        hipError_t e = hipEventRecord(ev, (hipStream_t)p);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)wgrad_stream, ev, 0);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); rc = (int)e; }
    }
    if (rc == 0) rc = shg_wgrad_group(ex->wq.data(), (int)ex->wq.size(), dtype, wgrad_stream);
    for (size_t i = 0; rc == 0 && i < ex->cq.size(); ++i) {
        const ColsumJob& c = ex->cq[i];
        rc = colsum_sliced(c.x, dtype, c.rows, c.cols, c.ld, c.out, wgrad_stream);
    }
    for (size_t i = 0; rc == 0 && i < ex->fq.size(); ++i) {
        const FinishJob& f = ex->fq[i];
        rc = shg_colsum_finish_multi(f.parts, f.outs, f.n, f.n_part, f.cols, wgrad_stream);
    }
    ex->fq.clear();
    ex->wq.clear();
    ex->cq.clear();
    ex->producers.clear();
    ex->pending_tiles = 0;
    ex->q_dtype = -1;
    return rc;
}

extern "C" void shg_exec_destroy(shg_exec_t* h) {
    Exec* ex = reinterpret_cast<Exec*>(h);
    if (!ex) return;
    for (hipEvent_t e : ex->events) (void)hipEventDestroy(e);
    delete ex;
}

extern "C" int64_t shg_attn_sublayer_saved_bytes(int mode, int dtype, int B, int Sq, int Sk, int heads) {
    if (mode < 0 || mode > 3 || B < 1 || Sq < 1 || Sk < 1 || heads < 1) return -1;
    return attn_saved(nullptr, mode, dtype, B, Sq, Sk, heads).bytes;
}
extern "C" int64_t shg_attn_sublayer_scratch_bytes(int mode, int dtype, int B, int Sq, int Sk, int heads) {
    if (mode < 0 || mode > 3 || B < 1 || Sq < 1 || Sk < 1 || heads < 1) return -1;
    return attn_scratch(nullptr, mode, dtype, B, Sq, Sk, heads).bytes;
}

extern "C" int shg_attn_sublayer_fwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x,
                                     const void* xpos, const void* mem, void* y, const void* pos, void* y_pos, void* saved,
                                     uint64_t sid) {
    CK(attn_args_ok(L, R, B, Sq, Sk));
    if (!x || !y || !saved) return fail_arg("attn_sublayer_fwd: null pointer");
    if ((L->mode == SHG_ATTN_DEC_SELF || L->mode == SHG_ATTN_DEC_CROSS) && !xpos) return fail_arg("attn_sublayer_fwd: decoder modes need xpos");
    if ((L->mode == SHG_ATTN_CROSS || L->mode == SHG_ATTN_DEC_CROSS) && !mem) return fail_arg("attn_sublayer_fwd: cross modes need mem");
    return attn_fwd(L, R, B, Sq, Sk, x, xpos, mem, y, pos, y_pos, saved, sid);
}

extern "C" int shg_attn_sublayer_bwd(const shg_attn_sublayer_t* L, const shg_run_t* R, int B, int Sq, int Sk, const void* x,
                                     const void* xpos, const void* mem, const void* saved, const void* dy, void* dx, void* dxpos,
                                     void* dmem, int dmem_accumulate, void* scratch, uint64_t sid) {
    CK(attn_args_ok(L, R, B, Sq, Sk));
    if (!x || !saved || !dy || !scratch) return fail_arg("attn_sublayer_bwd: null pointer");
    if ((L->mode == SHG_ATTN_DEC_SELF || L->mode == SHG_ATTN_DEC_CROSS) && !xpos) return fail_arg("attn_sublayer_bwd: decoder modes need xpos");
    if ((L->mode == SHG_ATTN_CROSS || L->mode == SHG_ATTN_DEC_CROSS) && !mem) return fail_arg("attn_sublayer_bwd: cross modes need mem");
    return attn_bwd(L, R, B, Sq, Sk, x, xpos, mem, saved, dy, dx, dxpos, dmem, dmem_accumulate, scratch, sid);
}

extern "C" int64_t shg_ffn_sublayer_saved_bytes(int dtype, int64_t rows, int H, int F) {
    if (rows < 1 || H < 1 || F < 1) return -1;
    return ffn_saved(nullptr, dtype, rows, H, F).bytes;
}
extern "C" int64_t shg_ffn_sublayer_scratch_bytes(int dtype, int64_t rows, int H, int F) {
    if (rows < 1 || H < 1 || F < 1) return -1;
    return ffn_scratch(nullptr, dtype, rows, H, F).bytes;
}

extern "C" int shg_ffn_sublayer_fwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x,
                                    void* y, const void* pos, void* y_pos, void* saved, uint64_t sid) {
    CK(ffn_args_ok(L, R, rows, H, F));
    if (!x || !y || !saved) return fail_arg("ffn_sublayer_fwd: null pointer");
    return ffn_fwd(L, R, rows, H, F, x, y, pos, y_pos, saved, sid);
}

extern "C" int shg_ffn_sublayer_bwd(const shg_ffn_sublayer_t* L, const shg_run_t* R, int64_t rows, int H, int F, const void* x,
                                    const void* saved, const void* dy, void* dx, void* scratch, uint64_t sid) {
    CK(ffn_args_ok(L, R, rows, H, F));
    if (!x || !saved || !dy || !scratch) return fail_arg("ffn_sublayer_bwd: null pointer");
    return ffn_bwd(L, R, rows, H, F, x, saved, dy, dx, scratch, sid);
}

extern "C" int64_t shg_decoder_saved_bytes(int n_layers, int dtype, int B, int Q, int S, int heads, int F) {
    if (n_layers < 1 || B < 1 || Q < 1 || S < 1 || heads < 1 || F < 1) return -1;
    return dec_saved(nullptr, n_layers, dtype, B, Q, S, heads, F).bytes;
}
extern "C" int64_t shg_decoder_scratch_bytes(int n_layers, int dtype, int B, int Q, int S, int heads, int F) {
    if (n_layers < 1 || B < 1 || Q < 1 || S < 1 || heads < 1 || F < 1) return -1;
    return dec_scratch(nullptr, n_layers, dtype, B, Q, S, heads, F).bytes;
}

extern "C" int shg_decoder_fwd(const shg_decoder_layer_t* layers, int n_layers, const shg_run_t* R, int B, int Q, int S, int F,
                               const void* tgt, const void* memory, const void* query_pos, void* out, void* saved, uint64_t sid) {
    CK(dec_args_ok(layers, n_layers, R, B, Q, S, F));
    if (!memory || !query_pos || !out || !saved) return fail_arg("decoder_fwd: null pointer");
    const int dt = R->dtype, heads = layers[0].self_attn.heads;
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Q, es = esize(dt);
    DecSaved d = dec_saved(saved, n_layers, dt, B, Q, S, heads, F);
    const void *x, *xp;
    if (!tgt) {                                   // tgt = zeros (agqa_model.py:234): x + pos is pos itself
        hipError_t e = hipMemsetAsync(d.zero, 0, (size_t)(rq * H * es), (hipStream_t)R->stream);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return (int)e; }
        x = d.zero;
        xp = query_pos;
    } else {
        CK(shg_add(tgt, query_pos, d.xp0, dt, rq * H, R->stream));
        x = tgt;
        xp = d.xp0;
    }
    // shg_run_t.kv_ahead: the layers' key / value projections of `memory` do not depend on the decoder's state.  They go to the
    // weight-gradient stream (idle in a forward pass) in layer order, one event each; the chain on R->stream waits for layer i's
    // event in front of its cross-attention instead of running a 12 576-row GEMM between two 4 096-row kernels.
    const bool ahead = R->kv_ahead && R->wgrad_stream && R->wgrad_stream != R->stream && R->exec;
    std::vector<hipEvent_t> kv_ready;
    if (ahead) {
        Exec* ex = reinterpret_cast<Exec*>(R->exec);
        if ((int)ex->events.size() < n_layers + 2) return fail_arg("decoder_fwd: shg_exec_t has too few events for kv_ahead");
        void* ws = nullptr;
        CK(fork_wgrad(R, &ws));
        const int64_t rk = (int64_t)B * S;
        for (int i = 0; i < n_layers; ++i) {
            const shg_attn_sublayer_t& C = layers[i].cross_attn;
            AttnSaved sv = attn_saved(d.layer[i].s_cross, SHG_ATTN_DEC_CROSS, dt, B, Q, S, heads);
            CK(shg_gemm(memory, C.b.w, sv.kv, C.b.bias, dt, dt, rk, 2 * H, H, H, H, 2 * H, 1, 1, 0, ws));
            hipEvent_t ev = ex->events[ex->cursor];
            ex->cursor = (ex->cursor + 1) % ex->events.size();
            hipError_t e = hipEventRecord(ev, (hipStream_t)ws);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); return (int)e; }
            kv_ready.push_back(ev);
        }
    }
    for (int i = 0; i < n_layers; ++i) {
        const shg_decoder_layer_t& L = layers[i];
        DecLayerBufs& b = d.layer[i];
        const bool last = i == n_layers - 1;
        void* y3 = last ? out : b.y3;
        CK(attn_fwd(&L.self_attn, R, B, Q, Q, x, xp, nullptr, b.y1, query_pos, b.y1p, b.s_self, sid + 6 * i));
        if (ahead) {
            hipError_t e = hipStreamWaitEvent((hipStream_t)R->stream, kv_ready[i], 0);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); return (int)e; }
        }
        CK(attn_fwd(&L.cross_attn, R, B, Q, S, b.y1, b.y1p, memory, b.y2, nullptr, nullptr, b.s_cross, sid + 6 * i + 2, ahead));
        CK(ffn_fwd(&L.ffn, R, rq, (int)H, F, b.y2, y3, last ? nullptr : query_pos, last ? nullptr : b.y3p, b.s_ffn, sid + 6 * i + 4));
        x = y3;
        xp = b.y3p;
    }
    return 0;
}

extern "C" int shg_decoder_bwd(const shg_decoder_layer_t* layers, int n_layers, const shg_run_t* R, int B, int Q, int S, int F,
                               const void* tgt, const void* memory, const void* query_pos, const void* saved, const void* d_out,
                               void* d_tgt, void* d_query_pos, void* d_memory, void* scratch, uint64_t sid) {
    CK(dec_args_ok(layers, n_layers, R, B, Q, S, F));
    if (!memory || !query_pos || !saved || !d_out || !scratch) return fail_arg("decoder_bwd: null pointer");
    if (d_tgt && !tgt) return fail_arg("decoder_bwd: d_tgt without tgt");
    const int dt = R->dtype, heads = layers[0].self_attn.heads;
    const int64_t H = (int64_t)heads * 64, rq = (int64_t)B * Q, n = rq * H;
    const DecSaved d = dec_saved(const_cast<void*>(saved), n_layers, dt, B, Q, S, heads, F);
    const DecScratch w = dec_scratch(scratch, n_layers, dt, B, Q, S, heads, F);
    void* st = R->stream;
    const void* dy = d_out;
    bool pos_init = true, mem_init = true;
    // d_memory = sum over the layers of dK/dV_l . W_kv_l.  The layers' dK/dV live in their own scratch blocks, the weights in the
    // parameter arena - both at a constant stride from layer to layer - so the sum is ONE GEMM over K = layers x 2 H after the loop
    // (shg_gemm_kseg) instead of one accumulating 12 576 x 768 x 1 536 GEMM per layer in the middle of the chain.
    const int64_t es_ = esize(dt), rk_ = (int64_t)B * S;
    bool kseg = d_memory && n_layers >= 2 && dt == SHG_BF16 && tuning(TUNE_DECODER_KSEG);
    int64_t a_stride = 0, b_stride = 0;
    if (kseg) {
        const char* a0 = (const char*)attn_scratch(w.layer[0].w_cross, SHG_ATTN_DEC_CROSS, dt, B, Q, S, heads).dkv;
        const char* a1 = (const char*)attn_scratch(w.layer[1].w_cross, SHG_ATTN_DEC_CROSS, dt, B, Q, S, heads).dkv;
        const char* b0 = (const char*)layers[0].cross_attn.b.w;
        const char* b1 = (const char*)layers[1].cross_attn.b.w;
        a_stride = a1 - a0;
        b_stride = b1 - b0;
        for (int i = 0; kseg && i < n_layers; ++i) {
            const char* ai = (const char*)attn_scratch(w.layer[i].w_cross, SHG_ATTN_DEC_CROSS, dt, B, Q, S, heads).dkv;
            kseg = ai == a0 + i * a_stride && (const char*)layers[i].cross_attn.b.w == b0 + i * b_stride && layers[i].cross_attn.b.w != nullptr;
        }
        kseg = kseg && a_stride % 16 == 0 && b_stride % 16 == 0;
    }
    for (int i = n_layers - 1; i >= 0; --i) {
        const shg_decoder_layer_t& L = layers[i];
        const DecLayerBufs& b = d.layer[i];
        const DecScratchLayer& s = w.layer[i];
        const void* x = i == 0 ? (tgt ? tgt : d.zero) : d.layer[i - 1].y3;
        const void* xp = i == 0 ? (tgt ? d.xp0 : query_pos) : d.layer[i - 1].y3p;
        // gradient w.r.t. the layer input: only when something upstream wants it
        const bool need_x = i > 0 || d_tgt != nullptr;
        void* d0 = need_x ? (i == 0 ? d_tgt : s.d0) : nullptr;
        CK(ffn_bwd(&L.ffn, R, rq, (int)H, F, b.y2, b.s_ffn, dy, s.d2, s.w_ffn, sid + 6 * i + 4));
        CK(attn_bwd(&L.cross_attn, R, B, Q, S, b.y1, b.y1p, memory, b.s_cross, s.d2, s.d1, s.dxp_b, kseg ? nullptr : d_memory, mem_init ? 0 : 1,
                    s.w_cross, sid + 6 * i + 2));
        mem_init = false;
        // y1p = y1 + pos: its gradient goes to y1 and to pos
        CK(shg_add2_accumulate(s.d1, d_query_pos, s.dxp_b, pos_init ? 1 : 0, dt, n, st));
        pos_init = false;
        const bool want_xp = d_query_pos || need_x;
        CK(attn_bwd(&L.self_attn, R, B, Q, Q, x, xp, nullptr, b.s_self, s.d1, d0, want_xp ? s.dxp_a : nullptr, nullptr, 0, s.w_self,
                    sid + 6 * i));
        if (want_xp) CK(shg_add2_accumulate(d0, d_query_pos, s.dxp_a, 0, dt, n, st));     // xp = x + pos likewise
        dy = s.d0;
    }
    if (kseg) {
        const void* a0 = attn_scratch(w.layer[0].w_cross, SHG_ATTN_DEC_CROSS, dt, B, Q, S, heads).dkv;
        CK(shg_gemm_kseg(a0, layers[0].cross_attn.b.w, d_memory, dt, rk_, H, 2 * H, n_layers, 2 * H, H, H, a_stride / es_, b_stride / es_, 0, st));
    }
    return 0;
}

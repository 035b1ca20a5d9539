// Fused multi-head attention (head dim 64) for gfx950: forward, dQ and dK/dV kernels.
//
// Layout idea: every product is arranged so that the softmax row lives on a LANE.
//   forward, per wave 16 queries:   S^T = K . Q^T   (keys on accumulator rows, query = lane & 15)
//                                   O^T = V^T . P^T (P^T taken straight from the S^T accumulators)
// The running max / sum / rescale of the online softmax are therefore lane-local (two xor-shuffles
// join the four 16-lane groups), P never goes through LDS, and V^T is read from the row-major V
// tile with the transposing LDS read.  K and V tiles (64 keys x 64) are staged in LDS once per
// workgroup (4 waves = 64 queries), double-buffered: the next tile streams global -> LDS directly
// (global_load_lds, XOR swizzle on the source address) while the current one feeds the MFMAs.
// The [B,H,Sq,Sk] score tensor of the reference (modeling_capsbert.py:394-418) is never formed.
//
// Backward recomputes P from the saved log-sum-exp:
//   dQ kernel  (per wave 16 queries, loops over key tiles):   dS^T = P^T o (dP^T - delta),  dQ^T = K^T . dS^T
//   dKV kernel (per wave 16 keys, loops over query tiles):     dV^T = dO^T . P,  dK^T = Q^T . dS
// Both are deterministic (no atomics).
#include <math.h>

#include "mma.h"

namespace shg {

struct AttnParams {
    const void *q, *k, *v;
    int B, H, Sq, Sk;
    int64_t q_bs, q_ss, k_bs, k_ss, v_bs, v_ss;
    const float* mask;
    float scale;
    uint32_t drop_thr;
    float drop_scale;
    const uint64_t* seed_state;
    uint64_t stream_id;
};

template <int MASK>
__device__ __forceinline__ float mask_value(const float* mask, int b, int qrow, int key, int Sk) {
    if (MASK == SHG_MASK_KEY) return mask[(int64_t)b * Sk + key];
    if (MASK == SHG_MASK_FULL) return mask[(int64_t)qrow * Sk + key];
    return 0.f;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int MASK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams P, T* __restrict__ o, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    // two stages of (K tile, V tile): the next key tile streams into LDS while this one is consumed
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qidx = blockIdx.x * 64 + wave * 16 + li;
    const int qrow = min(qidx, P.Sq - 1);
    const T* qptr = (const T*)P.q + (int64_t)b * P.q_bs + (int64_t)qrow * P.q_ss + h * 64;
    const T* kbase = (const T*)P.k + (int64_t)b * P.k_bs + h * 64;
    const T* vbase = (const T*)P.v + (int64_t)b * P.v_bs + h * 64;
    load_tile64_async<T>(smem, kbase, P.k_ss, min(64, P.Sk), tid);
    load_tile64_async<T>(smem + TL::BYTES, vbase, P.v_ss, min(64, P.Sk), tid);
    const Frag<T> qf0 = glb_row_frag(qptr, 0, g), qf1 = glb_row_frag(qptr, 32, g);
    const uint64_t seed = P.drop_thr ? dropout_seed(P.seed_state, P.stream_id) : 0;
    const uint64_t drop_row = ((uint64_t)(b * P.H + h) * P.Sq + qrow) * (uint64_t)((P.Sk + 1) & ~1);   // even row pitch

    f32x4 acc_o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (int kb = 0; kb < P.Sk; kb += 64) {
        const char* ldsK = smem + cur * 2 * TL::BYTES;
        const char* ldsV = ldsK + TL::BYTES;
        if (kb + 64 < P.Sk) {
            char* nxt = smem + (cur ^ 1) * 2 * TL::BYTES;
            const int valid = min(64, P.Sk - kb - 64);
            load_tile64_async<T>(nxt, kbase + (int64_t)(kb + 64) * P.k_ss, P.k_ss, valid, tid);
            load_tile64_async<T>(nxt + TL::BYTES, vbase + (int64_t)(kb + 64) * P.v_ss, P.v_ss, valid, tid);
        }

        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(s[kt], lds_row_frag<T>(ldsK, 16 * kt + li, 0, g), qf0);
            mma(s[kt], lds_row_frag<T>(ldsK, 16 * kt + li, 32, g), qf1);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + 16 * kt + 4 * g + r;
                float val = s[kt][r] * P.scale;
                if (MASK != SHG_MASK_NONE) val += mask_value<MASK>(P.mask, b, qrow, min(key, P.Sk - 1), P.Sk);
                if (key >= P.Sk) val = -INFINITY;
                s[kt][r] = val;
                mx = fmaxf(mx, val);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __expf(s[kt][r] - m_use);
                rs += p;
                if (P.drop_thr) {
                    const int key = kb + 16 * kt + 4 * g + r;
                    p = dropout_keep_run(seed, (drop_row + (uint64_t)(kb + 16 * kt + 4 * g)) >> 1, r, P.drop_thr) ? p * P.drop_scale : 0.f;
                }
                s[kt][r] = p;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc_o[d] *= alpha;
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const Frag<T> pf = acc_frag<T>(s[2 * sx], s[2 * sx + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) mma(acc_o[d], lds_col_frag<T>(ldsV, 32 * sx, 16 * d, lane), pf);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    if (qidx < P.Sq) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        T* optr = o + ((int64_t)b * P.Sq + qidx) * (P.H * 64) + h * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) optr[16 * d + 4 * g + r] = from_f32<T>(acc_o[d][r] * inv);
        if (g == 0) lse[((int64_t)b * P.H + h) * P.Sq + qidx] = m_run + logf(l_run);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, dQ (also produces delta = rowsum(dO o O))
// ------------------------------------------------------------------------------------------------
template <typename T, int MASK>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams P, const T* __restrict__ o, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          T* __restrict__ dq, int64_t dq_bs, int64_t dq_ss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qidx = blockIdx.x * 64 + wave * 16 + li;
    const int qrow = min(qidx, P.Sq - 1);
    const T* qptr = (const T*)P.q + (int64_t)b * P.q_bs + (int64_t)qrow * P.q_ss + h * 64;
    const int64_t orow = ((int64_t)b * P.Sq + qrow) * (P.H * 64) + h * 64;
    const T* kbase = (const T*)P.k + (int64_t)b * P.k_bs + h * 64;
    const T* vbase = (const T*)P.v + (int64_t)b * P.v_bs + h * 64;
    const Frag<T> qf0 = glb_row_frag(qptr, 0, g), qf1 = glb_row_frag(qptr, 32, g);
    const Frag<T> df0 = glb_row_frag(d_o + orow, 0, g), df1 = glb_row_frag(d_o + orow, 32, g);
    const int64_t stat = ((int64_t)b * P.H + h) * P.Sq + qrow;
    float dl = 0.f;
    {
        const Frag<T> of0 = glb_row_frag(o + orow, 0, g), of1 = glb_row_frag(o + orow, 32, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += to_f32(of0.v[j]) * to_f32(df0.v[j]) + to_f32(of1.v[j]) * to_f32(df1.v[j]);
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0 && qidx < P.Sq) delta[stat] = dl;
    }
    const float lse_q = lse[stat];
    const uint64_t seed = P.drop_thr ? dropout_seed(P.seed_state, P.stream_id) : 0;
    const uint64_t drop_row = ((uint64_t)(b * P.H + h) * P.Sq + qrow) * (uint64_t)((P.Sk + 1) & ~1);   // even row pitch

    f32x4 acc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (all ordinary global loads above are consumed before the first direct-to-LDS load is issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    load_tile64_async<T>(smem, kbase, P.k_ss, min(64, P.Sk), tid);
    load_tile64_async<T>(smem + TL::BYTES, vbase, P.v_ss, min(64, P.Sk), tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (int kb = 0; kb < P.Sk; kb += 64) {
        const char* ldsK = smem + cur * 2 * TL::BYTES;
        const char* ldsV = ldsK + TL::BYTES;
        if (kb + 64 < P.Sk) {
            char* nxt = smem + (cur ^ 1) * 2 * TL::BYTES;
            const int valid = min(64, P.Sk - kb - 64);
            load_tile64_async<T>(nxt, kbase + (int64_t)(kb + 64) * P.k_ss, P.k_ss, valid, tid);
            load_tile64_async<T>(nxt + TL::BYTES, vbase + (int64_t)(kb + 64) * P.v_ss, P.v_ss, valid, tid);
        }
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(s[kt], lds_row_frag<T>(ldsK, 16 * kt + li, 0, g), qf0);
            mma(s[kt], lds_row_frag<T>(ldsK, 16 * kt + li, 32, g), qf1);
            mma(dp[kt], lds_row_frag<T>(ldsV, 16 * kt + li, 0, g), df0);
            mma(dp[kt], lds_row_frag<T>(ldsV, 16 * kt + li, 32, g), df1);
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + 16 * kt + 4 * g + r;
                float val = s[kt][r] * P.scale;
                if (MASK != SHG_MASK_NONE) val += mask_value<MASK>(P.mask, b, qrow, min(key, P.Sk - 1), P.Sk);
                float p = (key < P.Sk) ? __expf(val - lse_q) : 0.f;
                float dpe = dp[kt][r];
                if (P.drop_thr) dpe = dropout_keep_run(seed, (drop_row + (uint64_t)(kb + 16 * kt + 4 * g)) >> 1, r, P.drop_thr) ? dpe * P.drop_scale : 0.f;
                s[kt][r] = p * (dpe - dl);
            }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const Frag<T> dsf = acc_frag<T>(s[2 * sx], s[2 * sx + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) mma(acc[d], lds_col_frag<T>(ldsK, 32 * sx, 16 * d, lane), dsf);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    if (qidx < P.Sq) {
        T* out = dq + (int64_t)b * dq_bs + (int64_t)qidx * dq_ss + h * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[16 * d + 4 * g + r] = from_f32<T>(acc[d][r] * P.scale);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, dK and dV
// ------------------------------------------------------------------------------------------------
template <typename T, int MASK>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams P, const T* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dk, int64_t dk_bs, int64_t dk_ss,
                                                           T* __restrict__ dv, int64_t dv_bs, int64_t dv_ss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    // two stages of (Q tile, dO tile, lse[64], delta[64])
    constexpr int STG = 2 * TL::BYTES + 512;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kidx = blockIdx.x * 64 + wave * 16 + li;
    const int krow = min(kidx, P.Sk - 1);
    const T* kptr = (const T*)P.k + (int64_t)b * P.k_bs + (int64_t)krow * P.k_ss + h * 64;
    const T* vptr = (const T*)P.v + (int64_t)b * P.v_bs + (int64_t)krow * P.v_ss + h * 64;
    const Frag<T> kf0 = glb_row_frag(kptr, 0, g), kf1 = glb_row_frag(kptr, 32, g);
    const Frag<T> vf0 = glb_row_frag(vptr, 0, g), vf1 = glb_row_frag(vptr, 32, g);
    const T* qbase = (const T*)P.q + (int64_t)b * P.q_bs + h * 64;
    const T* dbase = d_o + (int64_t)b * P.Sq * (P.H * 64) + h * 64;
    const int64_t stat0 = ((int64_t)b * P.H + h) * P.Sq;
    const uint64_t seed = P.drop_thr ? dropout_seed(P.seed_state, P.stream_id) : 0;
    const uint64_t drop_bh = (uint64_t)(b * P.H + h) * P.Sq;
    const float kmask = (MASK == SHG_MASK_KEY) ? P.mask[(int64_t)b * P.Sk + krow] : 0.f;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    auto stage = [&](int buf, int qb) {
        char* base = smem + buf * STG;
        const int valid = min(64, P.Sq - qb);
        load_tile64_async<T>(base, qbase + (int64_t)qb * P.q_ss, P.q_ss, valid, tid);
        load_tile64_async<T>(base + TL::BYTES, dbase + (int64_t)qb * (P.H * 64), P.H * 64, valid, tid);
        // per-query statistics: 64 floats each, one 4-byte direct-to-LDS load per lane (waves 0 and 1)
        const int qq = min(qb + lane, P.Sq - 1);
        if (wave_u == 0) __builtin_amdgcn_global_load_lds((glb_ptr)(lse + stat0 + qq), (lds_ptr)(base + 2 * TL::BYTES), 4, 0, 0);
        if (wave_u == 1) __builtin_amdgcn_global_load_lds((glb_ptr)(delta + stat0 + qq), (lds_ptr)(base + 2 * TL::BYTES + 256), 4, 0, 0);
    };

    f32x4 acc_k[4], acc_v[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { acc_k[d] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the register loads above are complete
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (int qb = 0; qb < P.Sq; qb += 64) {
        const char* ldsQ = smem + cur * STG;
        const char* ldsD = ldsQ + TL::BYTES;
        const float* ldsLse = reinterpret_cast<const float*>(ldsQ + 2 * TL::BYTES);
        const float* ldsDelta = ldsLse + 64;
        if (qb + 64 < P.Sq) stage(cur ^ 1, qb + 64);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            s[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma(s[qt], lds_row_frag<T>(ldsQ, 16 * qt + li, 0, g), kf0);
            mma(s[qt], lds_row_frag<T>(ldsQ, 16 * qt + li, 32, g), kf1);
            mma(dp[qt], lds_row_frag<T>(ldsD, 16 * qt + li, 0, g), vf0);
            mma(dp[qt], lds_row_frag<T>(ldsD, 16 * qt + li, 32, g), vf1);
        }
        // lane: key = li (kidx), query = qb + 16 qt + 4 g + r
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * qt + 4 * g + r;
                const int query = qb + ql;
                float val = s[qt][r] * P.scale;
                if (MASK == SHG_MASK_KEY) val += kmask;
                if (MASK == SHG_MASK_FULL) val += P.mask[(int64_t)min(query, P.Sq - 1) * P.Sk + krow];
                float p = (query < P.Sq && kidx < P.Sk) ? __expf(val - ldsLse[ql]) : 0.f;
                float dpe = dp[qt][r];
                float pd = p;
                if (P.drop_thr) {
                    const bool keep = dropout_keep(seed, (drop_bh + (uint64_t)min(query, P.Sq - 1)) * (uint64_t)((P.Sk + 1) & ~1) + (uint64_t)krow, P.drop_thr);
                    dpe = keep ? dpe * P.drop_scale : 0.f;
                    pd = keep ? p * P.drop_scale : 0.f;
                }
                s[qt][r] = p * (dpe - ldsDelta[ql]);   // dS
                dp[qt][r] = pd;                        // dropped P
            }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            const Frag<T> pf = acc_frag<T>(dp[2 * sx], dp[2 * sx + 1]);
            const Frag<T> dsf = acc_frag<T>(s[2 * sx], s[2 * sx + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                mma(acc_v[d], lds_col_frag<T>(ldsD, 32 * sx, 16 * d, lane), pf);
                mma(acc_k[d], lds_col_frag<T>(ldsQ, 32 * sx, 16 * d, lane), dsf);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    if (kidx < P.Sk) {
        T* ok = dk + (int64_t)b * dk_bs + (int64_t)kidx * dk_ss + h * 64;
        T* ov = dv + (int64_t)b * dv_bs + (int64_t)kidx * dv_ss + h * 64;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ok[16 * d + 4 * g + r] = from_f32<T>(acc_k[d][r] * P.scale);
                ov[16 * d + 4 * g + r] = from_f32<T>(acc_v[d][r]);
            }
    }
}

static int attn_check(const AttnParams& P, int dtype, int mask_kind, float p_drop) {
    if (!P.q || !P.k || !P.v) return fail_arg("attention: null pointer");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("attention: bad dtype");
    if (P.B < 1 || P.H < 1 || P.Sq < 1 || P.Sk < 1 || P.B > 65535 || P.H > 65535) return fail_arg("attention: bad sizes");
    const int a = dtype == SHG_BF16 ? 8 : 4;
    if ((P.q_bs | P.q_ss | P.k_bs | P.k_ss | P.v_bs | P.v_ss) % a) return fail_arg("attention: strides must keep rows 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(P.q) | reinterpret_cast<uintptr_t>(P.k) | reinterpret_cast<uintptr_t>(P.v)) & 15)
        return fail_arg("attention: q/k/v must be 16-byte aligned");
    if (mask_kind < 0 || mask_kind > 2 || (mask_kind != SHG_MASK_NONE && !P.mask)) return fail_arg("attention: bad mask");
    if (p_drop < 0.f || p_drop >= 1.f) return fail_arg("attention: bad p_drop");
    return 0;
}

}  // namespace shg

using namespace shg;

#define ATTN_DISPATCH(KERNEL, T, LDS, ...)                                                                        \
    do {                                                                                                          \
        if (mask_kind == SHG_MASK_NONE) hipLaunchKernelGGL((KERNEL<T, SHG_MASK_NONE>), grid, block, LDS, st, __VA_ARGS__); \
        else if (mask_kind == SHG_MASK_KEY) hipLaunchKernelGGL((KERNEL<T, SHG_MASK_KEY>), grid, block, LDS, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, SHG_MASK_FULL>), grid, block, LDS, st, __VA_ARGS__);                   \
    } while (0)

extern "C" int shg_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B,
                                 int H, int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride,
                                 int64_t k_sstride, int64_t v_bstride, int64_t v_sstride, int mask_kind,
                                 const float* mask, float scale, float p_drop, const uint64_t* seed_state,
                                 uint64_t stream_id, void* stream) {
    AttnParams P{q, k, v, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride, v_bstride, v_sstride, mask, scale,
                 dropout_threshold(p_drop), p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed_state, stream_id};
    if (int e = attn_check(P, dtype, mask_kind, p_drop)) return e;
    if (!o || !lse) return fail_arg("attention_fwd: null output");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((Sq + 63) / 64, H, B), block(256);
    if (dtype == SHG_F32) ATTN_DISPATCH(attn_fwd_kernel, float, 4 * Tile64<float>::BYTES, P, (float*)o, lse);
    else ATTN_DISPATCH(attn_fwd_kernel, bf16_t, 4 * Tile64<bf16_t>::BYTES, P, (bf16_t*)o, lse);
    return check_launch("attention_fwd");
}

extern "C" int shg_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                 const float* lse, float* delta, void* dq, void* dk, void* dv, int dtype, int B, int H,
                                 int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride,
                                 int64_t k_sstride, int64_t v_bstride, int64_t v_sstride, int64_t dq_bstride,
                                 int64_t dq_sstride, int64_t dk_bstride, int64_t dk_sstride, int64_t dv_bstride,
                                 int64_t dv_sstride, int mask_kind, const float* mask, float scale, float p_drop,
                                 const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    AttnParams P{q, k, v, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride, v_bstride, v_sstride, mask, scale,
                 dropout_threshold(p_drop), p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed_state, stream_id};
    if (int e = attn_check(P, dtype, mask_kind, p_drop)) return e;
    if (!o || !d_o || !lse || !delta || !dq || !dk || !dv) return fail_arg("attention_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    {
        dim3 grid((Sq + 63) / 64, H, B);
        if (dtype == SHG_F32)
            ATTN_DISPATCH(attn_bwd_dq_kernel, float, 4 * Tile64<float>::BYTES, P, (const float*)o, (const float*)d_o, lse, delta, (float*)dq, dq_bstride, dq_sstride);
        else
            ATTN_DISPATCH(attn_bwd_dq_kernel, bf16_t, 4 * Tile64<bf16_t>::BYTES, P, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, dq_bstride, dq_sstride);
    }
    {
        dim3 grid((Sk + 63) / 64, H, B);
        if (dtype == SHG_F32)
            ATTN_DISPATCH(attn_bwd_dkv_kernel, float, 2 * (2 * Tile64<float>::BYTES + 512), P, (const float*)d_o, lse, delta, (float*)dk, dk_bstride, dk_sstride, (float*)dv, dv_bstride, dv_sstride);
        else
            ATTN_DISPATCH(attn_bwd_dkv_kernel, bf16_t, 2 * (2 * Tile64<bf16_t>::BYTES + 512), P, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk, dk_bstride, dk_sstride, (bf16_t*)dv, dv_bstride, dv_sstride);
    }
    return check_launch("attention_bwd");
}

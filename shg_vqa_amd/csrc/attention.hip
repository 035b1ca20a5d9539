// Fused multi-head attention (head dim 64) for gfx950: forward, dQ and dK/dV kernels.
//
// Layout idea: every product is arranged so that the softmax row lives on a LANE.
//   forward, per wave 16 * NB queries:  S^T = K . Q^T   (keys on accumulator rows, query = lane & 15)
//                                       O^T = V^T . P^T (P^T taken straight from the S^T accumulators)
// The running max / sum / rescale of the online softmax are therefore lane-local (two xor-shuffles
// join the four 16-lane groups), P never goes through LDS, and V^T is read from the row-major V
// tile with the transposing LDS read.  K and V tiles (64 keys x 64) are staged in LDS once per
// workgroup (4 waves), double-buffered: the next tile streams global -> LDS directly
// (global_load_lds, XOR swizzle on the source address) while the current one feeds the MFMAs.
// The [B,H,Sq,Sk] score tensor of the reference (modeling_capsbert.py:394-418) is never formed.
//
// What bounds these kernels is the softmax arithmetic, not the matrix cores (a 393 x 393 head is 77 M score
// elements per launch against 15 GFLOP of MFMA work: ~8 us of matrix time, ~20 us of VALU time at one exponential
// per element), so the second version of the kernels is built around the VALU budget per score element:
//   * scores live in the log2 domain: t = s * (scale * log2 e) + mask * log2 e is ONE fma, p = exp2(t - m) one
//     subtraction and one v_exp_f32 (the natural exponential costs an extra multiply per element);
//   * a wave owns NB = 2 blocks of 16 queries (forward, dQ) or keys (dK/dV): every K / V / Q / dO fragment read
//     from LDS feeds two MFMAs, a workgroup covers 128 rows per barrier pair instead of 64, and the key-tile loop
//     runs half as often per query;
//   * the additive key mask of the BERT blocks is staged in LDS with the key tile (one 4-byte direct-to-LDS load per
//     lane) instead of one global load per score element; the tail test (key >= Sk) only exists in the last tile;
//   * the output accumulators are only rescaled when some running maximum actually moved (wave-uniform test).
//
// Dropout: the forward kernel draws the keep decisions (counter-based hash, common.h) and WRITES them out as lane masks: for a
// wave's block of 16 queries and a tile of 64 keys, sixteen 64-bit words - word e = 4 kt + r is the ballot of "keep" over the
// wave for accumulator element (kt, r), i.e. bit 16 g + li <-> (query li of the block, key 16 kt + 4 g + r of the tile): the
// compare result the forward select uses anyway, 128 bytes per wave and key tile (393 x 393 x 384 heads: 8.6 MB per call).  The
// backward kernels never hash: dQ has the forward's lane <-> element mapping, loads the sixteen words with scalar loads and uses
// each as the lane mask of its select; dK/dV holds one KEY per lane and four queries per accumulator block, so a lane fetches the
// one word of its key (e = 4 kt(key) + r(key)) per 16-query block and tests bits 16 g(key) + 4 g + r.  In round 2 both backward
// kernels re-hashed every element (+36 us of 170 at 393 x 393).
//
// Backward recomputes P from the saved log-sum-exp:
//   dQ kernel  (per wave 16 NB queries, loops over key tiles):   dS^T = P^T o (dP^T - delta),  dQ^T = K^T . dS^T
//   dKV kernel (per wave 16 NB keys, loops over query tiles):     dV^T = dO^T . P,  dK^T = Q^T . dS
// Both are deterministic (no atomics).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

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
    uint64_t* keep;            // dropout lane masks [B*H][ceil(Sq/16)][ceil(Sk/64)][16] (written by forward, read by backward)
    float *dbq, *dbk, *dbv;    // backward: bias gradients of the q / k / v projections, fp32 [H * 64] each, or null
};

// Column sums of a head's gradient block, added into the projection's bias gradient: acc[n][d][r] is the gradient of feature
// 16 d + 4 g + r for the lane's row (query or key li of block n).  The rows of a wave are summed over the 16 lanes of a DPP row,
// the waves of the workgroup through LDS, and 64 lanes add the head's 64 features with one atomic instruction.  (The bias
// gradients of the q / k / v projections used to be separate column-sum kernels over the [rows, 3 H] gradient - 59 launches per
// step which, issued twice, cost +0.97 ms per step: tools/family_cost.py.)
template <int NB>
__device__ __forceinline__ void head_colsum(const f32x4 (&acc)[NB][4], const bool (&valid)[NB], float mul, float* dst /* + head offset */,
                                            float* red /* LDS [4][64] */, int tid) {
    const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n) v += valid[n] ? acc[n][d][r] : 0.f;
            v += dpp_take<0xB1, 0xF>(v);         // quad_perm [1,0,3,2]
            v += dpp_take<0x4E, 0xF>(v);         // quad_perm [2,3,0,1]
            v += dpp_take<0x141, 0xF>(v);        // row_half_mirror
            v += dpp_take<0x140, 0xF>(v);        // row_mirror: every lane of the row of 16 holds the row's sum
            if (li == 0) red[wave * 64 + 16 * d + 4 * g + r] = v * mul;
        }
    __syncthreads();
    if (tid < 64) atomicAdd(dst + tid, red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid]);
    __syncthreads();
}

// first of the 16 mask words of (head bh, 16-query block qblk, 64-key tile ktile)
__device__ __forceinline__ int64_t keep_word0(const AttnParams& P, int bh, int qblk, int ktile) {
    const int nq16 = (P.Sq + 15) >> 4, nkt = (P.Sk + 63) >> 6;
    return (((int64_t)bh * nq16 + qblk) * nkt + ktile) * 16;
}
// The sixteen mask words of a (query block, key tile) go to memory through the SCALAR unit (s_store_dwordx2: they are compare
// results, i.e. SGPR pairs already - moving them into a vector register first cost 32 v_writelane per key tile in a kernel that is
// bound by its vector instructions).  The masks were written by VALU compares: the scalar memory instruction must not read them in
// the next few cycles, and the compiler's hazard recogniser does not look into inline assembly - hence the s_nop in front (the
// first version of this path moved the words with v_writelane directly behind the compare and stored stale ones).  The scalar
// data cache is written back at the end of the kernel (attn_fwd_kernel: s_dcache_wb).
__device__ __forceinline__ void store_keep_masks(uint64_t* dst, const uint64_t (&m)[16]) {
    asm volatile("s_nop 4\n\t"
                 "s_store_dwordx2 %1, %0, 0x0\n\ts_store_dwordx2 %2, %0, 0x8\n\ts_store_dwordx2 %3, %0, 0x10\n\ts_store_dwordx2 %4, %0, 0x18\n\t"
                 "s_store_dwordx2 %5, %0, 0x20\n\ts_store_dwordx2 %6, %0, 0x28\n\ts_store_dwordx2 %7, %0, 0x30\n\ts_store_dwordx2 %8, %0, 0x38\n\t"
                 "s_store_dwordx2 %9, %0, 0x40\n\ts_store_dwordx2 %10, %0, 0x48\n\ts_store_dwordx2 %11, %0, 0x50\n\ts_store_dwordx2 %12, %0, 0x58\n\t"
                 "s_store_dwordx2 %13, %0, 0x60\n\ts_store_dwordx2 %14, %0, 0x68\n\ts_store_dwordx2 %15, %0, 0x70\n\ts_store_dwordx2 %16, %0, 0x78"
                 :: "s"(dst), "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(m[4]), "s"(m[5]), "s"(m[6]), "s"(m[7]), "s"(m[8]), "s"(m[9]),
                    "s"(m[10]), "s"(m[11]), "s"(m[12]), "s"(m[13]), "s"(m[14]), "s"(m[15]) : "memory");
}
// x where the lane's bit of `mask` is set, else 0: the mask is an SGPR pair and goes straight into the select
__device__ __forceinline__ float select_by_lane_mask(float x, uint64_t mask) {
    float r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// 64 floats (one per lane of ONE wave) global -> LDS, clamped at n - 1
__device__ __forceinline__ void load_row64_async(char* lds, const float* src, int first, int n, int lane) {
    const int i = min(first + lane, n - 1);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + i), (lds_ptr_t)lds, 4, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// LDS per stage: K tile, V tile, 64 mask floats (MASK_KEY)
template <typename T> constexpr int fwd_stage_bytes() { return 2 * Tile64<T>::BYTES + 256; }

template <typename T, int MASK, int NB, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams P, T* __restrict__ o, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    constexpr int STG = fwd_stage_bytes<T>();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int b = blockIdx.z, h = blockIdx.y;
    const T* kbase = (const T*)P.k + (int64_t)b * P.k_bs + h * 64;
    const T* vbase = (const T*)P.v + (int64_t)b * P.v_bs + h * 64;
    const float* mrow = (MASK == SHG_MASK_KEY) ? P.mask + (int64_t)b * P.Sk : nullptr;
    const TileLaneOffsets<T> koff = tile_lane_offsets<T>(P.k_ss, tid), voff = tile_lane_offsets<T>(P.v_ss, tid);
    auto stage = [&](int buf, int kb) {
        char* base = smem + buf * STG;
        const int valid = min(64, P.Sk - kb);
        if (valid == 64) {                          // whole tile: scalar base + the lane offsets computed once
            load_tile64_async_full<T>(base, kbase + (int64_t)kb * P.k_ss, koff, tid);
            load_tile64_async_full<T>(base + TL::BYTES, vbase + (int64_t)kb * P.v_ss, voff, tid);
        } else {
            load_tile64_async<T>(base, kbase + (int64_t)kb * P.k_ss, P.k_ss, valid, tid);
            load_tile64_async<T>(base + TL::BYTES, vbase + (int64_t)kb * P.v_ss, P.v_ss, valid, tid);
        }
        if (MASK == SHG_MASK_KEY && wave_u == 0) load_row64_async(base + 2 * TL::BYTES, mrow, kb, P.Sk, lane);
    };
    stage(0, 0);
    int qidx[NB], qrow[NB];
    Frag<T> qf[NB][2];
    uint64_t drop_row[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        qidx[n] = blockIdx.x * (64 * NB) + wave * (16 * NB) + 16 * n + li;
        qrow[n] = min(qidx[n], P.Sq - 1);
        const T* qptr = (const T*)P.q + (int64_t)b * P.q_bs + (int64_t)qrow[n] * P.q_ss + h * 64;
        qf[n][0] = glb_row_frag(qptr, 0, g);
        qf[n][1] = glb_row_frag(qptr, 32, g);
        drop_row[n] = ((uint64_t)(b * P.H + h) * P.Sq + qrow[n]) * (uint64_t)((P.Sk + 3) & ~3);   // row pitch: a multiple of 4 (quads)
    }
    const uint64_t seed = DROP ? dropout_seed(P.seed_state, P.stream_id) : 0;
    const float c2 = P.scale * LOG2E;
    // a wave whose rows all lie past the end only helps staging the tiles (wave-uniform; it keeps every barrier)
    const bool active = (int)(blockIdx.x * (64 * NB) + wave_u * (16 * NB)) < P.Sq;

    f32x4 acc_o[NB][4];
    float m_run[NB], l_run[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        m_run[n] = -INFINITY;
        l_run[n] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc_o[n][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (int kb = 0; kb < P.Sk; kb += 64) {
        const char* ldsK = smem + cur * STG;
        const char* ldsV = ldsK + TL::BYTES;
        const float* ldsM = reinterpret_cast<const float*>(ldsK + 2 * TL::BYTES);
        if (kb + 64 < P.Sk) stage(cur ^ 1, kb + 64);

        if (active) {
        f32x4 s[NB][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const Frag<T> kf0 = lds_row_frag<T>(ldsK, 16 * kt + li, 0, g), kf1 = lds_row_frag<T>(ldsK, 16 * kt + li, 32, g);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                s[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma(s[n][kt], kf0, qf[n][0]);
                mma(s[n][kt], kf1, qf[n][1]);
            }
        }
        // log2-domain scores: t = s * scale * log2(e) + mask * log2(e).  The tail test only exists in the last key tile
        // (TAIL is a compile-time constant inside `softmax`: per-element run-time tests would become per-element branches).
        f32x4 mk[4];
        if (MASK == SHG_MASK_KEY) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) mk[kt] = *reinterpret_cast<const f32x4*>(ldsM + 16 * kt + 4 * g) * LOG2E;
        }
        bool moved = false;
        float alpha[NB];
        auto softmax = [&](auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t;
                        if (MASK == SHG_MASK_KEY) t = fmaf(s[n][kt][r], c2, mk[kt][r]);
                        else if (MASK == SHG_MASK_FULL)
                            t = fmaf(s[n][kt][r], c2, P.mask[(int64_t)qrow[n] * P.Sk + min(kb + 16 * kt + 4 * g + r, P.Sk - 1)] * LOG2E);
                        else t = s[n][kt][r] * c2;
                        if (TAIL) t = (kb + 16 * kt + 4 * g + r >= P.Sk) ? -INFINITY : t;
                        s[n][kt][r] = t;
                        mx = fmaxf(mx, t);
                    }
                mx = xrow_max(mx);
                const float m_new = fmaxf(m_run[n], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                alpha[n] = fast_exp2(m_run[n] - m_use);
                moved = moved || (m_new != m_run[n]);
                float rs = 0.f;
                uint64_t km[16];                           // mask word e = 4 kt + r: the ballot of "keep" for accumulator element (kt, r)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = fast_exp2(s[n][kt][r] - m_use);
                        rs += p;
                        if (DROP) {          // (the 1 / (1 - p) scale is applied once, to the normaliser at the end of the kernel)
                            const bool keep = dropout_keep_run(seed, drop_row[n] + (uint64_t)(kb + 16 * kt + 4 * g), r, P.drop_thr);
                            km[4 * kt + r] = __builtin_amdgcn_ballot_w64(keep);       // (the compare result itself: no extra instruction)
                            p = keep ? p : 0.f;
                        }
                        s[n][kt][r] = p;
                    }
                if (DROP) {
                    const int qblk = (int)blockIdx.x * (4 * NB) + wave_u * NB + n;
                    store_keep_masks(P.keep + keep_word0(P, b * P.H + h, qblk, kb >> 6), km);
                }
                rs = xrow_sum(rs);
                l_run[n] = l_run[n] * alpha[n] + rs;
                m_run[n] = m_new;
            }
        };
        if (kb + 64 > P.Sk) softmax(std::true_type{}); else softmax(std::false_type{});
        if (__builtin_amdgcn_ballot_w64(moved)) {       // some running maximum moved: rescale the output accumulators
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc_o[n][d] *= alpha[n];
        }
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            Frag<T> pf[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) pf[n] = acc_frag<T>(s[n][2 * sx], s[n][2 * sx + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const Frag<T> vf = lds_col_frag<T>(ldsV, 32 * sx, 16 * d, lane);
#pragma unroll
                for (int n = 0; n < NB; ++n) mma(acc_o[n][d], vf, pf[n]);
            }
        }
        }   // active
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (qidx[n] < P.Sq) {
            const float inv = l_run[n] > 0.f ? (DROP ? P.drop_scale : 1.f) / l_run[n] : 0.f;
            T* optr = o + ((int64_t)b * P.Sq + qidx[n]) * (P.H * 64) + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) optr[16 * d + 4 * g + r] = from_f32<T>(acc_o[n][d][r] * inv);
            if (g == 0) lse[((int64_t)b * P.H + h) * P.Sq + qidx[n]] = (m_run[n] + log2f(l_run[n])) * LN2;      // natural log
        }
    }
    if (DROP) asm volatile("s_dcache_wb" ::: "memory");    // the keep masks left through the scalar data cache
}

// ------------------------------------------------------------------------------------------------
// backward, dQ (also produces delta = rowsum(dO o O))
// ------------------------------------------------------------------------------------------------
// (second launch bound: waves per SIMD the register allocation must leave room for - the bf16 single-block key-mask kernel of the
//  relation layers sits six registers above the 128 of four waves per SIMD without it: 142 -> 139.8 us for dQ + dK/dV at 393 x 393;
//  the same bound on the dK/dV kernel - 196 -> 168 registers, three waves - spills inside its loops: 233 us)
template <typename T, int MASK, int NB, bool DROP>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && NB == 1 && MASK == SHG_MASK_KEY) ? 4 : 1) void attn_bwd_dq_kernel(AttnParams P, const T* __restrict__ o, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          T* __restrict__ dq, int64_t dq_bs, int64_t dq_ss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    constexpr int STG = fwd_stage_bytes<T>();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int b = blockIdx.z, h = blockIdx.y;
    const T* kbase = (const T*)P.k + (int64_t)b * P.k_bs + h * 64;
    const T* vbase = (const T*)P.v + (int64_t)b * P.v_bs + h * 64;
    const float* mrow = (MASK == SHG_MASK_KEY) ? P.mask + (int64_t)b * P.Sk : nullptr;
    int qidx[NB], qrow[NB];
    Frag<T> qf[NB][2], df[NB][2];
    float dl[NB], lse2[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        qidx[n] = blockIdx.x * (64 * NB) + wave * (16 * NB) + 16 * n + li;
        qrow[n] = min(qidx[n], P.Sq - 1);
        const T* qptr = (const T*)P.q + (int64_t)b * P.q_bs + (int64_t)qrow[n] * P.q_ss + h * 64;
        const int64_t orow = ((int64_t)b * P.Sq + qrow[n]) * (P.H * 64) + h * 64;
        qf[n][0] = glb_row_frag(qptr, 0, g);
        qf[n][1] = glb_row_frag(qptr, 32, g);
        df[n][0] = glb_row_frag(d_o + orow, 0, g);
        df[n][1] = glb_row_frag(d_o + orow, 32, g);
        const Frag<T> of0 = glb_row_frag(o + orow, 0, g), of1 = glb_row_frag(o + orow, 32, g);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += to_f32(of0.v[j]) * to_f32(df[n][0].v[j]) + to_f32(of1.v[j]) * to_f32(df[n][1].v[j]);
        d = xrow_sum(d);
        const int64_t stat = ((int64_t)b * P.H + h) * P.Sq + qrow[n];
        if (g == 0 && qidx[n] < P.Sq) delta[stat] = d;
        dl[n] = d;
        lse2[n] = lse[stat] * LOG2E;
    }
    const float c2 = P.scale * LOG2E;
    const bool active = (int)(blockIdx.x * (64 * NB) + wave_u * (16 * NB)) < P.Sq;       // wave-uniform, see the forward kernel

    f32x4 acc[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[n][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    const TileLaneOffsets<T> koff = tile_lane_offsets<T>(P.k_ss, tid), voff = tile_lane_offsets<T>(P.v_ss, tid);
    auto stage = [&](int buf, int kb) {
        char* base = smem + buf * STG;
        const int valid = min(64, P.Sk - kb);
        if (valid == 64) {
            load_tile64_async_full<T>(base, kbase + (int64_t)kb * P.k_ss, koff, tid);
            load_tile64_async_full<T>(base + TL::BYTES, vbase + (int64_t)kb * P.v_ss, voff, tid);
        } else {
            load_tile64_async<T>(base, kbase + (int64_t)kb * P.k_ss, P.k_ss, valid, tid);
            load_tile64_async<T>(base + TL::BYTES, vbase + (int64_t)kb * P.v_ss, P.v_ss, valid, tid);
        }
        if (MASK == SHG_MASK_KEY && wave_u == 0) load_row64_async(base + 2 * TL::BYTES, mrow, kb, P.Sk, lane);
    };
    // (all ordinary global loads above are consumed before the first direct-to-LDS load is issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;

    for (int kb = 0; kb < P.Sk; kb += 64) {
        const char* ldsK = smem + cur * STG;
        const char* ldsV = ldsK + TL::BYTES;
        const float* ldsM = reinterpret_cast<const float*>(ldsK + 2 * TL::BYTES);
        if (kb + 64 < P.Sk) stage(cur ^ 1, kb + 64);
        if (active) {
        // the forward's keep decisions of this (query block, key tile): sixteen lane masks per block, fetched by the scalar unit
        // (wave-uniform address) while the matrix products below run; word e = 4 kt + r is the select mask of element (kt, r)
        u32x8 km[NB][4];
        if (DROP) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const uint64_t* mp = P.keep + keep_word0(P, b * P.H + h, (int)blockIdx.x * (4 * NB) + wave_u * NB + n, kb >> 6);
                asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60"
                             : "=&s"(km[n][0]), "=&s"(km[n][1]), "=&s"(km[n][2]), "=&s"(km[n][3]) : "s"(mp) : "memory");
            }
        }
        f32x4 s[NB][4], dp[NB][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const Frag<T> kf0 = lds_row_frag<T>(ldsK, 16 * kt + li, 0, g), kf1 = lds_row_frag<T>(ldsK, 16 * kt + li, 32, g);
            const Frag<T> vf0 = lds_row_frag<T>(ldsV, 16 * kt + li, 0, g), vf1 = lds_row_frag<T>(ldsV, 16 * kt + li, 32, g);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                s[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[n][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                mma(s[n][kt], kf0, qf[n][0]);
                mma(s[n][kt], kf1, qf[n][1]);
                mma(dp[n][kt], vf0, df[n][0]);
                mma(dp[n][kt], vf1, df[n][1]);
            }
        }
        f32x4 mk[4];
        if (MASK == SHG_MASK_KEY) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) mk[kt] = *reinterpret_cast<const f32x4*>(ldsM + 16 * kt + 4 * g) * LOG2E;
        }
        if (DROP) {                                  // the scalar loads have landed (the wait also names the registers they wrote)
#pragma unroll
            for (int n = 0; n < NB; ++n)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(km[n][0]), "+s"(km[n][1]), "+s"(km[n][2]), "+s"(km[n][3])::"memory");
        }
        auto elems = [&](auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t;
                        if (MASK == SHG_MASK_KEY) t = fmaf(s[n][kt][r], c2, mk[kt][r]);
                        else if (MASK == SHG_MASK_FULL)
                            t = fmaf(s[n][kt][r], c2, P.mask[(int64_t)qrow[n] * P.Sk + min(kb + 16 * kt + 4 * g + r, P.Sk - 1)] * LOG2E);
                        else t = s[n][kt][r] * c2;
                        float p = fast_exp2(t - lse2[n]);
                        if (TAIL) p = (kb + 16 * kt + 4 * g + r >= P.Sk) ? 0.f : p;
                        float dpe = dp[n][kt][r];
                        if (DROP) {
                            const u32x8 w8 = km[n][(4 * kt + r) >> 2];
                            const uint64_t m = ((uint64_t)w8[2 * ((4 * kt + r) & 3) + 1] << 32) | w8[2 * ((4 * kt + r) & 3)];
                            dpe = select_by_lane_mask(dpe * P.drop_scale, m);
                        }
                        s[n][kt][r] = p * (dpe - dl[n]);
                    }
        };
        if (kb + 64 > P.Sk) elems(std::true_type{}); else elems(std::false_type{});
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            Frag<T> dsf[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) dsf[n] = acc_frag<T>(s[n][2 * sx], s[n][2 * sx + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const Frag<T> kc = lds_col_frag<T>(ldsK, 32 * sx, 16 * d, lane);
#pragma unroll
                for (int n = 0; n < NB; ++n) mma(acc[n][d], kc, dsf[n]);
            }
        }
        }   // active
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (qidx[n] < P.Sq) {
            T* out = dq + (int64_t)b * dq_bs + (int64_t)qidx[n] * dq_ss + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[16 * d + 4 * g + r] = from_f32<T>(acc[n][d][r] * P.scale);
        }
    }
    if (P.dbq) {                                       // (wave-uniform; the operand buffers are free after the loop's last barrier)
        bool valid[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) valid[n] = qidx[n] < P.Sq;
        head_colsum<NB>(acc, valid, P.scale, P.dbq + h * 64, reinterpret_cast<float*>(smem), tid);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, dK and dV
// ------------------------------------------------------------------------------------------------
template <typename T> constexpr int dkv_stage_bytes() { return 2 * Tile64<T>::BYTES + 512 + 1024; }

template <typename T, int MASK, int NB, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnParams P, const T* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           T* __restrict__ dk, int64_t dk_bs, int64_t dk_ss,
                                                           T* __restrict__ dv, int64_t dv_bs, int64_t dv_ss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    // two stages of (Q tile, dO tile, lse[64], delta[64], keep words: per wave and 16-key block 4 query blocks x 4 words = 128 bytes,
    // 256 bytes per wave = what one 4-byte-per-lane direct-to-LDS instruction writes)
    constexpr int STG = dkv_stage_bytes<T>();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    int kidx[NB], krow[NB];
    Frag<T> kf[NB][2], vf[NB][2];
    float kmask2[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        kidx[n] = blockIdx.x * (64 * NB) + wave * (16 * NB) + 16 * n + li;
        krow[n] = min(kidx[n], P.Sk - 1);
        const T* kptr = (const T*)P.k + (int64_t)b * P.k_bs + (int64_t)krow[n] * P.k_ss + h * 64;
        const T* vptr = (const T*)P.v + (int64_t)b * P.v_bs + (int64_t)krow[n] * P.v_ss + h * 64;
        kf[n][0] = glb_row_frag(kptr, 0, g);
        kf[n][1] = glb_row_frag(kptr, 32, g);
        vf[n][0] = glb_row_frag(vptr, 0, g);
        vf[n][1] = glb_row_frag(vptr, 32, g);
        kmask2[n] = (MASK == SHG_MASK_KEY) ? P.mask[(int64_t)b * P.Sk + krow[n]] * LOG2E : 0.f;
    }
    const T* qbase = (const T*)P.q + (int64_t)b * P.q_bs + h * 64;
    const T* dbase = d_o + (int64_t)b * P.Sq * (P.H * 64) + h * 64;
    const int64_t stat0 = ((int64_t)b * P.H + h) * P.Sq;
    const float c2 = P.scale * LOG2E;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // dropout: the forward's lane masks (see the head of the file).  This lane's key sits in tile krow >> 6 at (kt, g, r) =
    // ((krow >> 4) & 3, (krow >> 2) & 3, krow & 3): it reads word e = 4 kt + r of every 16-query block and tests bits 16 g + ...
    const int nq16 = (P.Sq + 15) >> 4;
    int64_t keep_tile0[2] = {0, 0};                                                // per 16-key block of the wave: + r, + qblk * keep_qstep
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int key0c = min((int)blockIdx.x * (64 * NB) + __builtin_amdgcn_readfirstlane(wave) * (16 * NB) + 16 * n, P.Sk - 1);
        keep_tile0[n] = keep_word0(P, b * P.H + h, 0, key0c >> 6) + 4 * ((key0c >> 4) & 3);
    }
    const int64_t keep_qstep = (int64_t)((P.Sk + 63) >> 6) * 16;                  // words per 16-query block
    const int keep_shift = 16 * (li >> 2) + 4 * g;                                 // bit of (query 4 g + r, key li) is keep_shift + r
    const bool active = (int)(blockIdx.x * (64 * NB) + wave_u * (16 * NB)) < P.Sk;       // wave-uniform, see the forward kernel
    const bool key_tail = (int)(blockIdx.x * (64 * NB) + (wave_u + 1) * (16 * NB)) > P.Sk;   // this wave holds keys past Sk
    const TileLaneOffsets<T> qoff = tile_lane_offsets<T>(P.q_ss, tid), doff = tile_lane_offsets<T>(P.H * 64, tid);
    auto stage = [&](int buf, int qb) {
        char* base = smem + buf * STG;
        const int valid = min(64, P.Sq - qb);
        if (valid == 64) {
            load_tile64_async_full<T>(base, qbase + (int64_t)qb * P.q_ss, qoff, tid);
            load_tile64_async_full<T>(base + TL::BYTES, dbase + (int64_t)qb * (P.H * 64), doff, tid);
        } else {
            load_tile64_async<T>(base, qbase + (int64_t)qb * P.q_ss, P.q_ss, valid, tid);
            load_tile64_async<T>(base + TL::BYTES, dbase + (int64_t)qb * (P.H * 64), P.H * 64, valid, tid);
        }
        // per-query statistics: 64 floats each, one 4-byte direct-to-LDS load per lane (waves 0 and 1)
        if (wave_u == 0) load_row64_async(base + 2 * TL::BYTES, lse + stat0, qb, P.Sq, lane);
        if (wave_u == 1) load_row64_async(base + 2 * TL::BYTES + 256, delta + stat0, qb, P.Sq, lane);
        if (DROP) {
            // the forward's keep words this wave's 16 keys need for the tile's four 16-query blocks: word (qblk, r) = mask word
            // e = 4 kt(key block) + r of that block, as two dwords - lanes 0..31: (lane >> 3) = query block, (lane >> 1) & 3 = r
            // (lanes 32..63: the wave's second key block when NB = 2, a duplicate of the first otherwise)
            const int l32 = lane & 31;
            const int qblk = min((qb >> 4) + (l32 >> 3), nq16 - 1);                   // (blocks past Sq: don't-care, p is zeroed)
            const uint32_t* src = reinterpret_cast<const uint32_t*>(P.keep + keep_tile0[NB == 2 ? (lane >> 5) : 0] + (int64_t)qblk * keep_qstep + ((l32 >> 1) & 3)) + (l32 & 1);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(base + 2 * TL::BYTES + 512 + 256 * wave_u), 4, 0, 0);
        }
    };

    f32x4 acc_k[NB][4], acc_v[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int d = 0; d < 4; ++d) { acc_k[n][d] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[n][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
        if (active) {
        const char* ldsKeep = ldsQ + 2 * TL::BYTES + 512 + 256 * wave_u;     // [key block n][query block 0..3][r 0..3] 64-bit words of this wave
        // The 64 queries of the tile are worked in two halves of 32 (sx = the pair of 16-query blocks that forms one contraction
        // step of the dK / dV products): scores, elementwise part and products of a half are complete before the next half
        // starts, so only two of the four score / dP blocks are live at a time.  (Register count 196 -> 176-192; forcing the 168
        // of three waves per SIMD with a launch bound gains where it fits without spills - 131 -> 121 us for dQ + dK/dV at
        // 393 x 393 without dropout - and loses where it spills: the dropout instantiations 140 -> 139-199 us, the full-mask
        // one 36 -> 47 us.  No bound.)
        auto half = [&](auto sx_c, auto tail_c) {
            constexpr int sx = decltype(sx_c)::value;
            constexpr bool TAIL = decltype(tail_c)::value;     // last query tile, or a key block that reaches past Sk
            f32x4 s[NB][2], dp[NB][2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qt = 2 * sx + q2;
                const Frag<T> qf0 = lds_row_frag<T>(ldsQ, 16 * qt + li, 0, g), qf1 = lds_row_frag<T>(ldsQ, 16 * qt + li, 32, g);
                const Frag<T> df0 = lds_row_frag<T>(ldsD, 16 * qt + li, 0, g), df1 = lds_row_frag<T>(ldsD, 16 * qt + li, 32, g);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    s[n][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dp[n][q2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma(s[n][q2], qf0, kf[n][0]);
                    mma(s[n][q2], qf1, kf[n][1]);
                    mma(dp[n][q2], df0, vf[n][0]);
                    mma(dp[n][q2], df1, vf[n][1]);
                }
            }
            // lane: key = li of block n, query = qb + 16 qt + 4 g + r
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qt = 2 * sx + q2;
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(ldsLse + 16 * qt + 4 * g) * LOG2E;
                const f32x4 dlt = *reinterpret_cast<const f32x4*>(ldsDelta + 16 * qt + 4 * g);
                uint32_t kb4[NB];                              // keep bits of (queries 16 qt + 4 g + 0..3, this lane's key) in bits 0..3
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    kb4[n] = DROP ? (uint32_t)(*reinterpret_cast<const uint64_t*>(ldsKeep + 128 * n + 32 * qt + 8 * (li & 3)) >> keep_shift) : 0u;
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int query = qb + 16 * qt + 4 * g + r;
                        float t;
                        if (MASK == SHG_MASK_KEY) t = fmaf(s[n][q2][r], c2, kmask2[n]);
                        else if (MASK == SHG_MASK_FULL)
                            t = fmaf(s[n][q2][r], c2, P.mask[(int64_t)min(query, P.Sq - 1) * P.Sk + krow[n]] * LOG2E);
                        else t = s[n][q2][r] * c2;
                        float p = fast_exp2(t - l2[r]);
                        if (TAIL) p = (query >= P.Sq || kidx[n] >= P.Sk) ? 0.f : p;
                        float dpe = dp[n][q2][r];
                        float pd = p;
                        if (DROP) {                              // dS = P_dropped dP - P delta with P_dropped = keep ? P / (1 - p) : 0
                            pd = ((kb4[n] >> r) & 1u) ? p * P.drop_scale : 0.f;
                            s[n][q2][r] = fmaf(pd, dpe, -p * dlt[r]);
                        } else {
                            s[n][q2][r] = p * (dpe - dlt[r]);
                        }
                        dp[n][q2][r] = pd;                       // dropped P
                    }
            }
            Frag<T> pf[NB], dsf[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                pf[n] = acc_frag<T>(dp[n][0], dp[n][1]);
                dsf[n] = acc_frag<T>(s[n][0], s[n][1]);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const Frag<T> dc = lds_col_frag<T>(ldsD, 32 * sx, 16 * d, lane), qc = lds_col_frag<T>(ldsQ, 32 * sx, 16 * d, lane);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    mma(acc_v[n][d], dc, pf[n]);
                    mma(acc_k[n][d], qc, dsf[n]);
                }
            }
        };
        if (qb + 64 > P.Sq || key_tail) {
            half(std::integral_constant<int, 0>{}, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            half(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            half(std::integral_constant<int, 0>{}, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            half(std::integral_constant<int, 1>{}, std::false_type{});
        }
        }   // active
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (kidx[n] < P.Sk) {
            T* ok = dk + (int64_t)b * dk_bs + (int64_t)kidx[n] * dk_ss + h * 64;
            T* ov = dv + (int64_t)b * dv_bs + (int64_t)kidx[n] * dv_ss + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok[16 * d + 4 * g + r] = from_f32<T>(acc_k[n][d][r] * P.scale);
                    ov[16 * d + 4 * g + r] = from_f32<T>(acc_v[n][d][r]);
                }
        }
    }
    if (P.dbk || P.dbv) {
        bool valid[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) valid[n] = kidx[n] < P.Sk;
        if (P.dbk) head_colsum<NB>(acc_k, valid, P.scale, P.dbk + h * 64, reinterpret_cast<float*>(smem), tid);
        if (P.dbv) head_colsum<NB>(acc_v, valid, 1.f, P.dbv + h * 64, reinterpret_cast<float*>(smem), tid);
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
    if (p_drop > 0.f && !P.seed_state) return fail_arg("attention: dropout needs seed_state");
    if (p_drop > 0.f && !P.keep) return fail_arg("attention: dropout needs the keep-mask buffer (shg_attention_keep_mask_bytes)");
    if (P.keep && (reinterpret_cast<uintptr_t>(P.keep) & 127)) return fail_arg("attention: keep_mask must be 128-byte aligned");
    return 0;
}

// rows per wave: two 16-row blocks once the row count fills a 128-row workgroup reasonably (393 rows: 4 workgroups
// of 128 with 13 of 16 waves busy, against 7 workgroups of 64); short sequences (the 40-token questions, the 48 action
// queries) keep one block so that more workgroups exist
static int blocks_per_wave(int rows, int dtype) {
    const int nb = (int)tuning(TUNE_ATTN_NB);
    return (nb == 2 && dtype == SHG_BF16 && rows >= 96) ? 2 : 1;
}

}  // namespace shg

using namespace shg;

// (more than 64 KiB of dynamic LDS - the fp32 parity instantiations - needs the per-function limit raised first: once per
// instantiation and device, common.h raise_lds_limit)
#define ATTN_LAUNCH(KERN, LDS, ...)                                                                                   \
    do {                                                                                                              \
        if ((LDS) > 64 * 1024) {                                                                                      \
            static std::atomic<uint64_t> raised{0};                                                                   \
            raise_lds_limit(raised, reinterpret_cast<const void*>(KERN), (int)(LDS));                                 \
        }                                                                                                             \
        hipLaunchKernelGGL(KERN, grid, block, LDS, st, __VA_ARGS__);                                                  \
    } while (0)
#define ATTN_DISPATCH_D(KERNEL, T, NB, D, LDS, ...)                                                                   \
    do {                                                                                                              \
        if (mask_kind == SHG_MASK_NONE) ATTN_LAUNCH((KERNEL<T, SHG_MASK_NONE, NB, D>), LDS, __VA_ARGS__);            \
        else if (mask_kind == SHG_MASK_KEY) ATTN_LAUNCH((KERNEL<T, SHG_MASK_KEY, NB, D>), LDS, __VA_ARGS__);         \
        else ATTN_LAUNCH((KERNEL<T, SHG_MASK_FULL, NB, D>), LDS, __VA_ARGS__);                                       \
    } while (0)
// (the dropout test is a template parameter: a run-time test per score element turns into a branch per element)
#define ATTN_DISPATCH(KERNEL, T, NB, LDS, ...)                                                                        \
    do {                                                                                                              \
        if (P.drop_thr) ATTN_DISPATCH_D(KERNEL, T, NB, true, LDS, __VA_ARGS__);                                       \
        else ATTN_DISPATCH_D(KERNEL, T, NB, false, LDS, __VA_ARGS__);                                                 \
    } while (0)

extern "C" int64_t shg_attention_keep_mask_bytes(int B, int H, int Sq, int Sk) {
    if (B < 1 || H < 1 || Sq < 1 || Sk < 1) return -1;
    return (int64_t)B * H * ((Sq + 15) / 16) * ((Sk + 63) / 64) * 16 * 8;
}

extern "C" int shg_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int dtype, int B,
                                 int H, int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride,
                                 int64_t k_sstride, int64_t v_bstride, int64_t v_sstride, int mask_kind,
                                 const float* mask, float scale, float p_drop, const uint64_t* seed_state,
                                 uint64_t stream_id, uint64_t* keep_mask, void* stream) {
    SHG_REPEAT(1, shg_attention_fwd(q, k, v, o, lse, dtype, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride, v_bstride, v_sstride,
                                    mask_kind, mask, scale, p_drop, seed_state, stream_id, keep_mask, stream));
    AttnParams P{q, k, v, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride, v_bstride, v_sstride, mask, scale,
                 dropout_threshold(p_drop), p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed_state, stream_id, keep_mask,
                 nullptr, nullptr, nullptr};
    if (int e = attn_check(P, dtype, mask_kind, p_drop)) return e;
    if (!o || !lse) return fail_arg("attention_fwd: null output");
    hipStream_t st = (hipStream_t)stream;
    // two query blocks per wave pay when nothing but the softmax competes for registers (no dropout: the inference /
    // forward-only pass, 54 -> 49 us at 393 x 393); with the mask hash in the loop one block keeps three waves per SIMD
    const int nb = (p_drop > 0.f || Sq < 256) ? 1 : blocks_per_wave(Sq, dtype);
    dim3 grid((Sq + 64 * nb - 1) / (64 * nb), H, B), block(256);
    if (dtype == SHG_F32) ATTN_DISPATCH(attn_fwd_kernel, float, 1, 2 * fwd_stage_bytes<float>(), P, (float*)o, lse);
    else if (nb == 2) ATTN_DISPATCH(attn_fwd_kernel, bf16_t, 2, 2 * fwd_stage_bytes<bf16_t>(), P, (bf16_t*)o, lse);
    else ATTN_DISPATCH(attn_fwd_kernel, bf16_t, 1, 2 * fwd_stage_bytes<bf16_t>(), P, (bf16_t*)o, lse);
    return check_launch("attention_fwd");
}

extern "C" int shg_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                 const float* lse, float* delta, void* dq, void* dk, void* dv, int dtype, int B, int H,
                                 int Sq, int Sk, int64_t q_bstride, int64_t q_sstride, int64_t k_bstride,
                                 int64_t k_sstride, int64_t v_bstride, int64_t v_sstride, int64_t dq_bstride,
                                 int64_t dq_sstride, int64_t dk_bstride, int64_t dk_sstride, int64_t dv_bstride,
                                 int64_t dv_sstride, int mask_kind, const float* mask, float scale, float p_drop,
                                 const uint64_t* seed_state, uint64_t stream_id, const uint64_t* keep_mask, float* dbias_q,
                                 float* dbias_k, float* dbias_v, void* stream) {
    // (the repeated launch of the diagnostic switch leaves the bias gradients out: they accumulate)
    SHG_REPEAT(2, shg_attention_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, dtype, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride,
                                    v_bstride, v_sstride, dq_bstride, dq_sstride, dk_bstride, dk_sstride, dv_bstride, dv_sstride, mask_kind,
                                    mask, scale, p_drop, seed_state, stream_id, keep_mask, nullptr, nullptr, nullptr, stream));
    AttnParams P{q, k, v, B, H, Sq, Sk, q_bstride, q_sstride, k_bstride, k_sstride, v_bstride, v_sstride, mask, scale,
                 dropout_threshold(p_drop), p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed_state, stream_id,
                 const_cast<uint64_t*>(keep_mask), dbias_q, dbias_k, dbias_v};
    if (int e = attn_check(P, dtype, mask_kind, p_drop)) return e;
    if (!o || !d_o || !lse || !delta || !dq || !dk || !dv) return fail_arg("attention_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    {
        const int nb_dq = (int)tuning(TUNE_ATTN_NB_DQ);   // measured: 141 / 197 us (1) vs 144 / 203 us (2)
        const int nb = nb_dq == 2 ? blocks_per_wave(Sq, dtype) : 1;
        dim3 grid((Sq + 64 * nb - 1) / (64 * nb), H, B);
        if (dtype == SHG_F32)
            ATTN_DISPATCH(attn_bwd_dq_kernel, float, 1, 2 * fwd_stage_bytes<float>(), P, (const float*)o, (const float*)d_o, lse, delta, (float*)dq, dq_bstride, dq_sstride);
        else if (nb == 2)
            ATTN_DISPATCH(attn_bwd_dq_kernel, bf16_t, 2, 2 * fwd_stage_bytes<bf16_t>(), P, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, dq_bstride, dq_sstride);
        else
            ATTN_DISPATCH(attn_bwd_dq_kernel, bf16_t, 1, 2 * fwd_stage_bytes<bf16_t>(), P, (const bf16_t*)o, (const bf16_t*)d_o, lse, delta, (bf16_t*)dq, dq_bstride, dq_sstride);
    }
    {
        const int nb_dkv = (int)tuning(TUNE_ATTN_NB_DKV);
        const int nb = nb_dkv == 2 ? blocks_per_wave(Sk, dtype) : 1;
        dim3 grid((Sk + 64 * nb - 1) / (64 * nb), H, B);
        if (dtype == SHG_F32)
            ATTN_DISPATCH(attn_bwd_dkv_kernel, float, 1, 2 * dkv_stage_bytes<float>(), P, (const float*)d_o, lse, delta, (float*)dk, dk_bstride, dk_sstride, (float*)dv, dv_bstride, dv_sstride);
        else if (nb == 2)
            ATTN_DISPATCH(attn_bwd_dkv_kernel, bf16_t, 2, 2 * dkv_stage_bytes<bf16_t>(), P, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk, dk_bstride, dk_sstride, (bf16_t*)dv, dv_bstride, dv_sstride);
        else
            ATTN_DISPATCH(attn_bwd_dkv_kernel, bf16_t, 1, 2 * dkv_stage_bytes<bf16_t>(), P, (const bf16_t*)d_o, lse, delta, (bf16_t*)dk, dk_bstride, dk_sstride, (bf16_t*)dv, dv_bstride, dv_sstride);
    }
    return check_launch("attention_bwd");
}

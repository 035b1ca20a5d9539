// Shared device helpers for the SHG-VQA gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/shg_vqa.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;

#define SHG_WAVE 64

namespace shg {

// ------------------------------------------------------------------ error plumbing
void set_error(const char* msg);
int fail_arg(const char* msg);           // records msg, returns SHG_ERR_INVALID
int check_launch(const char* what);      // hipGetLastError -> 0 or positive hipError_t

// ------------------------------------------------------------------ tuning switches
// The library reads no environment variables: every switch is an entry of one table with its measured-best default,
// changed through shg_set_tuning(name, value) (include/shg_vqa.h).  The host binding applies SHG_* environment variables
// once at load time; tests flip single switches.  Launch paths read the table with one relaxed atomic load.
enum Tune {
    TUNE_ATTN_NB = 0,          // attention forward: 16-row blocks per wave without dropout (2) - "attn_nb"
    TUNE_ATTN_NB_DQ,           // dQ kernel (1; measured 141 / 197 us vs 144 / 203 us with 2) - "attn_nb_dq"
    TUNE_ATTN_NB_DKV,          // dK/dV kernel (1) - "attn_nb_dkv"
    TUNE_TILE_ORDER,           // 0 auto, 1 N fastest, 2 M fastest - "tile_order"
    TUNE_GEMM4_MAX_TILES,      // 128 x 128 ring kernel up to this many tiles (256) - "gemm4_max_tiles"
    TUNE_STREAMK_SIGMA,        // stream-K head share in percent (112) - "streamk_sigma"
    TUNE_STREAMK,              // bit 0 conv forward, 1 input gradient, 2 weight gradient (1) - "streamk"
    TUNE_GEMM8,                // 0: no 8-phase kernel (1) - "gemm8"
    TUNE_GEMM8_MIN_TILES,      // 8-phase kernel from this many 256 x 256 tiles (120) - "gemm8_min_tiles"
    TUNE_SPLITK_TARGET,        // weight-gradient split-K: workgroups aimed for (384) - "splitk_target"
    TUNE_SPLITK_MIN_STEPS,     // ... and the fewest K steps per split (8) - "splitk_min_steps"
    TUNE_LARGE_MIN_K,          // 256 x 256 tile of the generic kernel from this K (128) - "large_min_k"
    TUNE_WGRAD_GROUP,          // grouped weight gradients: bit 0 K % 64 == 0 groups, bit 1 ragged K, bit 2 launches of at most
                               // 256 workgroups = one round of the CUs (7) - "wgrad_group"
    TUNE_CONV_WGRAD_REMAINDER, // conv weight gradient: remainder column blocks as a split launch (1) - "conv_wgrad_remainder"
    TUNE_BERTADAM_MODE,        // bit 0: two vectors per lane, 3: four, 1: non-temporal stores of shadow / zeroed gradient too,
                               // 2: no non-temporal accesses (3; isolated at 289 M parameters: 1.575 ms = 6.24 TB/s against
                               // 1.677 ms for the round-2 setting 1 / 16 384 blocks, tools/bertadam_bench.py) - "bertadam_mode"
    TUNE_BERTADAM_BLOCKS,      // grid cap (65536) - "bertadam_blocks"
    TUNE_GEMM8_TILE_M,         // rows of the 8-phase tile: 256 (default), 192, or 0 = the one with fewer rounds x work per tile.
                               // Isolated, 0 wins on the 12 576-row problems with 768 / 1 536 columns (26.1 -> 23.0, 68.2 -> 58.6,
                               // 45.5 -> 39.8 us); in the step, where the other streams' kernels use the CUs a 150-tile launch
                               // leaves idle, the smaller tile's extra CU-time costs 0.4-2 % (three interleaved pairs) - "gemm8_tile_m"
    TUNE_ATTN_BWD_FUSED,       // 1: one backward kernel for dQ / dK / dV where available - "attn_bwd_fused"
    TUNE_EPILOGUE_SIDE,        // 1: GEMM row writers issue the loads of `C +=` / activation-backward forms up front - "epilogue_side"
    TUNE_CONV_K_ORDER,         // convolutions on the 8-phase kernel.  Bits 0-1, forward / input gradient: n > 0 = K-tiles in blocks of
                               // 64 * 2^(n-1) channels, all 45 taps of a block before the next (the 45 uses of a 128-byte input line
                               // fall within 45-180 consecutive K-tiles: L2 hits; conv1 6.27 -> 2.49 GB of L2 misses per launch), 0 =
                               // storage order (tap-major).  Bit 2, weight gradient: 256-column blocks channel-block-major, so that the
                               // tiles an XCD runs together gather the same input lines.  Bit 3, weight gradient with position-major
                               // rows (shg_conv3d_k533_wgrad_ex, row_order 1): skip the K-tiles of positions where the tile's tap reads
                               // the zero border.  Bit 4: with that, the taps of a channel block longest first.  Bit 5, forward with position-major
                               // rows (shg_conv3d_k533_fwd_rows, row_order 1): a tile leaves out the taps that read only the zero border
                               // for all its rows; the stream-K launch then uses the weighted plan.  Bit 6, input gradient with frame-major rows
                               // (shg_conv3d_k533_dgrad_rows, row_order 2): a tile leaves out the temporal taps that read only padding
                               // frames, stream-K with the weighted plan (126) - "conv_k_order"
    TUNE_LN_HALF_VEC,          // LayerNorm kernels on 768-column bf16 rows: eight-byte vectors, three per lane (1) - "ln_half_vec"
    TUNE_DECODER_KSEG,         // decoder backward: the gradient w.r.t. the memory as ONE GEMM over all layers' dK/dV (1) - "decoder_kseg"
    TUNE_WGRAD_GROUP_CAP,      // grouped weight gradients: workgroups per launch (256 = one round of the CUs) - "wgrad_group_cap"
    TUNE_WGRAD_GROUP_SPLIT,    // ... and parts of every problem's contraction (1) - "wgrad_group_split"
    TUNE_REPEAT_FAMILY,        // DIAGNOSTIC (0): bit mask of kernel families whose every launch is issued TWICE (all idempotent:
                               // 1 attention forward, 2 attention backward, 4 LayerNorm forward, 8 LayerNorm backward,
                               // 16 non-accumulating GEMMs of >= 120 tiles of 256 x 256, 32 smaller non-accumulating GEMMs,
                               // 64 convolution forward; and, changing the gradients (timing runs only): 128 grouped weight
                               // gradients, 256 convolution weight gradients, 512 convolution input gradient (idempotent),
                               // 1024 GEMM + activation backward, 2048 accumulating bf16 GEMMs, 4096 bias column sums): the
                               // step time it adds is what the family costs IN the step, next to the other streams' kernels
                               // (tools/family_cost.py) - "repeat_family"
    TUNE_COUNT
};
int64_t tuning(int key);
extern thread_local int g_in_repeat;
// first statement of an exported launch function: re-enters it once when its family's bit is set
#define SHG_REPEAT(BIT, CALL)                                                    \
    do {                                                                         \
        if ((shg::tuning(shg::TUNE_REPEAT_FAMILY) & (BIT)) && !shg::g_in_repeat) { \
            shg::g_in_repeat = 1;                                                \
            const int e_ = (CALL);                                               \
            shg::g_in_repeat = 0;                                                \
            if (e_) return e_;                                                   \
        }                                                                        \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one bit per device in a per-instantiation mask
// (true: the attribute has already been set for the current device)
inline bool lds_limit_raised(std::atomic<uint64_t>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    return (mask.load(std::memory_order_acquire) & ((uint64_t)1 << dev)) != 0;
}
// raises the limit of `func` on the current device unless `mask` says it has been done; the bit is published only AFTER
// hipFuncSetAttribute has returned, so a second host thread either sees the bit (attribute set) or sets the attribute itself
// (setting it twice is harmless)
inline void raise_lds_limit(std::atomic<uint64_t>& mask, const void* func, int bytes) {
    if (lds_limit_raised(mask)) return;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev <= 63) mask.fetch_or((uint64_t)1 << dev, std::memory_order_release);
}

// ------------------------------------------------------------------ scalar conversion
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// VB-byte vector of T (16: Vec16; 8: half of it - a row of 768 bf16 values is 96 sixteen-byte chunks, i.e. one and a HALF per lane
// of a 64-lane wave, but exactly three eight-byte chunks per lane)
template <typename T, int VB> struct VecB;
template <typename T> struct VecB<T, 16> : Vec16<T> {};
template <> struct VecB<bf16_t, 8> {
    static constexpr int N = 4;
    bf16x4 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <int VB, typename T> __device__ __forceinline__ VecB<T, VB> loadv(const T* p) {
    VecB<T, VB> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T, int VB> __device__ __forceinline__ void storev(T* p, const VecB<T, VB>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// ------------------------------------------------------------------ wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum on the DPP network instead of six ds_bpermute round trips through the LDS crossbar (~100 cycles each, dependent):
// quad swaps, row_half_mirror and row_mirror leave every lane with the sum of its row of 16; row_bcast:15 / :31 (gfx9 / CDNA
// only) carry the row sums into the last row; lane 63 is broadcast.  The association order differs from wave_sum's butterfly:
// kernels whose bits are pinned by a golden (the matcher's softmax) keep wave_sum.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_take(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_take<0xB1, 0xF>(v);         // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);         // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);        // row_half_mirror
    v += dpp_take<0x140, 0xF>(v);        // row_mirror
    v += dpp_take<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xC>(v);        // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// reductions over the four rows of 16 lanes (lanes l, l ^ 16, l ^ 32, l ^ 48 - the owners of one attention-score row), result
// in all four: v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows / the upper half with the other operand's
// even rows / lower half, so op(vdst, src) is the xor-16 / xor-32 step of a butterfly - a VALU instruction each instead of
// a ds_bpermute round trip; the association order is the butterfly's, so the bits are those of the __shfl_xor form.
__device__ __forceinline__ float xrow_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float xrow_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ counter-based dropout RNG
// keep(element) is a pure function of (seed, stream, element index), so a backward kernel can regenerate the forward mask (the
// attention kernels store theirs instead, attention.hip).  One hash serves the aligned QUAD of elements 4q .. 4q+3: 64 random
// bits, 16 per element, compared with the upper 16 bits of the threshold (p is resolved to 1/65536).  The hash is three
// multiply-folds: a 32 x 32 -> 64-bit product (one v_mad_u64_u32, measured 5.4 cycles per wave against 3.4 for an xor and 5.1
// for a 32-bit multiply, tools/native/valu_rates.hip) whose halves are xor-ed - every output bit then depends on every input
// bit - once on the counter and twice, with different constants, on the result: 3 products + ~6 single-issue instructions per
// quad against two rounds of xorshift-multiply (12 instructions) per PAIR in rounds 1-2.  Statistics of the four fields
// (tools/hash_quality.py: rate, chi-square of the 16-bit fields, correlation within a quad, serial correlation at lags 1, 2, 3,
// one row, one head, 2^16, 2^20, and between consecutive seeds) are those of the old hash.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t mulfold(uint32_t x, uint32_t c) {
    const uint64_t p = (uint64_t)x * c;
    return (uint32_t)p ^ (uint32_t)(p >> 32);
}
struct DropQuad { uint32_t w0, w1; };                // elements 0, 1 in the halves of w0, elements 2, 3 in the halves of w1
__device__ __forceinline__ DropQuad dropout_quad_words(uint64_t seed, uint64_t quad) {
    const uint32_t lo = (uint32_t)quad, hi = (uint32_t)(quad >> 32);
    const uint32_t y = mulfold(lo + (uint32_t)seed + ((hi << 16) | (hi >> 16)), 0x9E3779B1u) ^ (uint32_t)(seed >> 32);
    return DropQuad{mulfold(y, 0x85EBCA77u), mulfold(y, 0xC2B2AE3Du)};
}
__device__ __forceinline__ bool dropout_keep_field(const DropQuad& q, int k, uint32_t threshold) {
    return ((((k & 2) ? q.w1 : q.w0) >> ((k & 1) ? 16 : 0)) & 0xFFFFu) >= (threshold >> 16);
}
// threshold = (uint32_t)(p * 2^32); keep iff the element's 16 random bits >= threshold >> 16
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t threshold) {
    return dropout_keep_field(dropout_quad_words(seed, idx >> 2), (int)(idx & 3), threshold);
}
// element j of a run that starts at the element index `base`, a multiple of 4 (j a compile-time constant after unrolling: the
// four elements of a quad then share one hash)
__device__ __forceinline__ bool dropout_keep_run(uint64_t seed, uint64_t base, int j, uint32_t threshold) {
    return dropout_keep_field(dropout_quad_words(seed, (base >> 2) + (uint64_t)(j >> 2)), j & 3, threshold);
}
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (uint32_t)t;
}

// The per-step dropout seed lives in device memory so that a captured hipGraph replays with fresh
// masks: kernels read seed_state[0] (seed) + seed_state[1] (step counter), and `stream_id`
// separates the call sites inside one step.
__device__ __forceinline__ uint64_t dropout_seed(const uint64_t* seed_state, uint64_t stream_id) {
    uint64_t s = seed_state ? (seed_state[0] + 0x632BE59BD9B4E019ull * (seed_state[1] + 1)) : 0x1234567ull;
    s ^= stream_id * 0xD1342543DE82EF95ull;
    // both halves diffused (once per thread): the per-element hash only adds the low word to the counter
    const uint32_t lo = hash32((uint32_t)s ^ hash32((uint32_t)(s >> 32))), hi = hash32((uint32_t)(s >> 32) + 0x9E3779B9u ^ lo);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 storage: the same functions with erf from the Abramowitz-Stegun 7.1.26 rational form (|error| <= 1.5e-7, far
// below the 2^-9 rounding of the stored result).  libm's erff costs ~3x the instructions, and the activation passes over
// [rows, 3072] are instruction-bound, not HBM-bound, with it.  One exponential serves the cdf AND the density of the
// gradient; the negative branch uses 1 + erf(x) = poly * exp(-x^2/2) directly (no cancellation in the tail).
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __expf(-0.5f * x * x);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float half_tail = 0.5f * poly * e;              // = 0.5 * erfc(|x| / sqrt 2)
    cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_fast(float x) {
    float cdf, e;
    gelu_fast_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_fast_grad(float x) {
    float cdf, e;
    gelu_fast_parts(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

}  // namespace shg

// 16x16x32 "macro-MMA" on the gfx950 matrix cores, for bf16 and for exact-fp32 operands, plus
// the fragment loaders shared by the attention and GEMM kernels.
//
// One macro-MMA contracts 32 values of k.  A lane (li = lane & 15, g = lane >> 4) holds 8 operand
// elements, "slot" j = 0..7 of its group g:
//   bf16: one v_mfma_f32_16x16x32_bf16; slot (g, j) is k = 8g + j of the instruction.
//   fp32: eight v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, 1/16 of the bf16 rate - parity mode);
//         step j contracts the four slots (g, j), g = 0..3.
// A sum over k does not care which k sits in which slot as long as A and B agree, so the same 8
// values per lane serve both types.  C/D layout (both): col = lane & 15, row = 4 * (lane >> 4) + reg.
//
// LDS tiles are 64 rows x 64 elements, row-major, with the 16-byte chunk index XOR-ed with (row & 7)
// so that ds_read_b128 row-fragment reads are bank-conflict free (128-byte bf16 rows would otherwise
// be 8-way conflicting).
#pragma once
#include "common.h"

namespace shg {

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

template <typename T> struct Tile64 {
    static constexpr int EPC = 16 / (int)sizeof(T);     // elements per 16-byte chunk
    static constexpr int CH = 64 / EPC;                 // chunks per row
    static constexpr int ROWB = 64 * (int)sizeof(T);    // bytes per row
    static constexpr int BYTES = 64 * ROWB;
    __device__ __forceinline__ static int chunk_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }
    __device__ __forceinline__ static int elem_off(int row, int col) {
        return chunk_off(row, col / EPC) + (col % EPC) * (int)sizeof(T);
    }
    // Swizzle for tiles whose ROWS are the contraction index and that are only read with the
    // transposing LDS read (GEMM operands stored "contraction strided").  One ds_read_b64_tr_b16 of a
    // 32-lane half touches rows {8g+q : g in 2 groups, q = 0..3} x two adjacent 16-byte chunks; rows of the
    // same parity share a 128-byte half of the 256-byte bank window, so the mask must differ in chunk bits
    // 1..2 between rows r, r+2 (same group) and r, r+8 (the other group): mask = ((r>>1)&1)*2 + ((r>>3)&1)*4.
    // With (row & 7) instead, rows r and r+8 collide (2-way conflict on every transposed read).
    __device__ __forceinline__ static int swz_ks(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }
    __device__ __forceinline__ static int chunk_off_ks(int row, int chunk) { return row * ROWB + ((chunk ^ swz_ks(row)) << 4); }
    __device__ __forceinline__ static int elem_off_ks(int row, int col) {
        return chunk_off_ks(row, col / EPC) + (col % EPC) * (int)sizeof(T);
    }
    template <bool KMAJ> __device__ __forceinline__ static int swz(int row) { return KMAJ ? (row & 7) : swz_ks(row); }
};

// Stages a 64 x 64 tile global -> LDS directly (global_load_lds, 16 B per lane; one wave instruction
// fills 64 consecutive 16-byte slots, the swizzle is applied to the per-lane SOURCE address).  Rows
// beyond rows_valid re-read the last valid row (finite garbage the caller masks out).  The data is only
// visible after the issuing waves' s_waitcnt vmcnt(0) and a workgroup barrier.
template <typename T>
__device__ __forceinline__ void load_tile64_async(char* lds, const T* src, int64_t row_stride, int rows_valid, int tid) {
    using TL = Tile64<T>;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int i = 0; i < 64 * TL::CH / 256; ++i) {
        const int c = tid + 256 * i;
        const int row = c / TL::CH, ch = (c % TL::CH) ^ (row & 7);
        const int rsrc = min(row, rows_valid - 1);
        __builtin_amdgcn_global_load_lds((glb_ptr)(src + (int64_t)rsrc * row_stride + ch * TL::EPC),
                                         (lds_ptr)(lds + (256 * i + 64 * wave_u) * 16), 16, 0, 0);
    }
}

// The same staging with the per-lane part of the source address computed ONCE (tile_lane_offsets, for whole tiles) instead of per
// tile: a tile's address is then a wave-uniform base plus a 32-bit lane offset, which the load takes as scalar base + vector
// offset - no vector arithmetic per tile (the form above spends ~30 VALU instructions per staged tile pair on 64-bit multiplies
// and adds, 10 % of the attention kernels' vector instructions per key tile).
template <typename T> struct TileLaneOffsets { uint32_t v[64 * Tile64<T>::CH / 256]; };
template <typename T> __device__ __forceinline__ TileLaneOffsets<T> tile_lane_offsets(int64_t row_stride, int tid) {
    using TL = Tile64<T>;
    TileLaneOffsets<T> o;
#pragma unroll
    for (int i = 0; i < 64 * TL::CH / 256; ++i) {
        const int c = tid + 256 * i;
        const int row = c / TL::CH, ch = (c % TL::CH) ^ (row & 7);
        o.v[i] = (uint32_t)(((int64_t)row * row_stride + ch * TL::EPC) * (int64_t)sizeof(T));
    }
    return o;
}
template <typename T>
__device__ __forceinline__ void load_tile64_async_full(char* lds, const T* tile /* wave-uniform */, const TileLaneOffsets<T>& off, int tid) {
    using TL = Tile64<T>;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = reinterpret_cast<const char*>(tile);
#pragma unroll
    for (int i = 0; i < 64 * TL::CH / 256; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr)(base + off.v[i]), (lds_ptr)(lds + (256 * i + 64 * wave_u) * 16), 16, 0, 0);
}

// slot (g, j) <-> element k0 + 8g + j of `row` (contraction index contiguous in memory)
__device__ __forceinline__ Frag<bf16_t> lds_row_frag(const char* lds, int row, int k0, int g, bf16_t*) {
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const bf16x8*>(lds + Tile64<bf16_t>::chunk_off(row, (k0 >> 3) + g));
    return f;
}
__device__ __forceinline__ Frag<float> lds_row_frag(const char* lds, int row, int k0, int g, float*) {
    Frag<float> f;
    const int c0 = ((k0 >> 3) + g) * 2;
    const f32x4 a = *reinterpret_cast<const f32x4*>(lds + Tile64<float>::chunk_off(row, c0));
    const f32x4 b = *reinterpret_cast<const f32x4*>(lds + Tile64<float>::chunk_off(row, c0 + 1));
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
    return f;
}
template <typename T> __device__ __forceinline__ Frag<T> lds_row_frag(const char* lds, int row, int k0, int g) {
    return lds_row_frag(lds, row, k0, g, (T*)nullptr);
}

// the same fragment straight from global memory: 8 contiguous elements at rowptr[k0 + 8g ...]
__device__ __forceinline__ Frag<bf16_t> glb_row_frag(const bf16_t* rowptr, int k0, int g) {
    Frag<bf16_t> f;
    f.v = *reinterpret_cast<const bf16x8*>(rowptr + k0 + 8 * g);
    return f;
}
__device__ __forceinline__ Frag<float> glb_row_frag(const float* rowptr, int k0, int g) {
    Frag<float> f;
    const f32x4 a = *reinterpret_cast<const f32x4*>(rowptr + k0 + 8 * g);
    const f32x4 b = *reinterpret_cast<const f32x4*>(rowptr + k0 + 8 * g + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
    return f;
}

// "Accumulator order" of the contraction index: slot (g, j) <-> kappa = 16 * (j >> 2) + 4g + (j & 3),
// i.e. exactly where two stacked 16x16 C/D blocks (rows 0..15 and 16..31) keep their rows.
// acc_frag turns two such accumulator blocks into an operand without any lane movement...
template <typename T> __device__ __forceinline__ Frag<T> acc_frag(const f32x4& lo, const f32x4& hi);
template <> __device__ __forceinline__ Frag<bf16_t> acc_frag<bf16_t>(const f32x4& lo, const f32x4& hi) {
    Frag<bf16_t> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = (bf16_t)lo[j]; f.v[4 + j] = (bf16_t)hi[j]; }
    return f;
}
template <> __device__ __forceinline__ Frag<float> acc_frag<float>(const f32x4& lo, const f32x4& hi) {
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
    return f;
}

// ... and lds_col_frag reads the matching operand from a tile whose ROWS are the contraction index:
// slot (g, j) <-> tile[kb + kappa(g, j)][col0 + li].  bf16 uses the hardware transposing read
// (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, delivered column-major);
// EXEC must be all ones here - callers keep control flow wave-uniform and clamp instead of branching.
__device__ __forceinline__ Frag<bf16_t> lds_col_frag(const char* lds, int kb, int col0, int lane, bf16_t*) {
    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = kb + 4 * g + q, r1 = r0 + 16;
    const int col = col0 + 4 * p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + Tile64<bf16_t>::elem_off(r0, col)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + Tile64<bf16_t>::elem_off(r1, col)));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a;
    u.s[1] = b;
    Frag<bf16_t> f;
    f.v = u.v;
    return f;
}
__device__ __forceinline__ Frag<float> lds_col_frag(const char* lds, int kb, int col0, int lane, float*) {
    const int g = lane >> 4, col = col0 + (lane & 15);
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = kb + 16 * (j >> 2) + 4 * g + (j & 3);
        f.v[j] = *reinterpret_cast<const float*>(lds + Tile64<float>::elem_off(row, col));
    }
    return f;
}
template <typename T> __device__ __forceinline__ Frag<T> lds_col_frag(const char* lds, int kb, int col0, int lane) {
    return lds_col_frag(lds, kb, col0, lane, (T*)nullptr);
}

// Natural-order variant: slot (g, j) <-> tile[kb + 8g + j][col0 + li], i.e. the same k a row fragment
// of the other operand holds in that slot (GEMM operands stored with the contraction index strided).
// These tiles use the transposed-read swizzle (Tile64::swz_ks).
__device__ __forceinline__ Frag<bf16_t> lds_col_frag_nat(const char* lds, int kb, int col0, int lane, bf16_t*) {
    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = kb + 8 * g + q, r1 = r0 + 4;
    const int col = col0 + 4 * p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + Tile64<bf16_t>::elem_off_ks(r0, col)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + Tile64<bf16_t>::elem_off_ks(r1, col)));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a;
    u.s[1] = b;
    Frag<bf16_t> f;
    f.v = u.v;
    return f;
}
__device__ __forceinline__ Frag<float> lds_col_frag_nat(const char* lds, int kb, int col0, int lane, float*) {
    const int g = lane >> 4, col = col0 + (lane & 15);
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        f.v[j] = *reinterpret_cast<const float*>(lds + Tile64<float>::elem_off_ks(kb + 8 * g + j, col));
    return f;
}
template <typename T> __device__ __forceinline__ Frag<T> lds_col_frag_nat(const char* lds, int kb, int col0, int lane) {
    return lds_col_frag_nat(lds, kb, col0, lane, (T*)nullptr);
}

}  // namespace shg

// Matrix-core GEMM family for gfx950: plain GEMM (all four operand layouts) and the implicit-GEMM
// (5,3,3) Conv3d forward / weight-gradient kernels over a channels-last, spatially padded input.
//
// Structure (one kernel template, specialised by operand "sources"):
//   * workgroup = 256 threads (4 waves, 2 x 2), output tile 128 x 128, K step 64;
//   * both operands are staged global -> registers -> LDS with 16-byte accesses (issue the loads of
//     tile t+1 before the MFMAs of tile t, write them to LDS after: one LDS buffer, global latency
//     hidden behind the matrix work) into XOR-swizzled 64 x 64 sub-tiles (mma.h);
//   * an operand stored with the contraction index contiguous is read with ds_read_b128 row
//     fragments; one stored with the contraction index strided (dgrad / wgrad operands) is read
//     with the transposing LDS read, so no transposed copy of weights or activations is ever made;
//   * each wave owns 64 x 64 of the output as 4 x 4 accumulators of 16 x 16; the MFMA is issued
//     as (B-fragment, A-fragment) so that a lane ends up with 4 consecutive n of one output row and
//     stores them as one 8/16-byte vector;
//   * fp32 accumulation always; bf16 or exact-fp32 operands (mma.h).
// The convolution never materialises im2col: row r of the A tile is output position (b,t,h,w), the
// K index runs over (tap, channel) and the gather is a per-row base offset (precomputed table) plus
// a per-K-step tap offset - every 128-byte row piece is a contiguous, coalesced read of the
// channels-last tensor.
#include <math.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "mma.h"

namespace shg {

constexpr int BK = 64;

template <typename T> struct Stage {
    // 16-byte chunks per thread per operand and K-step; both tile configurations keep 128 operand
    // rows per 256 threads, so the count does not depend on the configuration
    static constexpr int NCH = (128 * 64 * (int)sizeof(T)) / 16 / 256;
};

// A thread's i-th staged chunk occupies the 16-byte LDS slot c = tid + NTHR i of the operand's
// stacked Tile64s (slot order = LDS address order, which is what a direct-to-LDS load needs: one wave
// instruction fills 64 consecutive slots).  Returns sub-tile, row and the LOGICAL chunk stored in that
// slot under the operand's XOR swizzle (contraction-contiguous tiles: p ^ (row & 7); contraction-strided
// tiles, read with the transposing LDS read: p ^ swz_ks(row)).
template <typename T, int NTHR, bool KMAJ> __device__ __forceinline__ void chunk_coord(int tid, int i, int& t, int& row, int& ch) {
    using TL = Tile64<T>;
    const int c = tid + NTHR * i;
    t = c / (64 * TL::CH);
    const int w = c % (64 * TL::CH);
    row = w / TL::CH;
    ch = (w % TL::CH) ^ TL::template swz<KMAJ>(row);
}

// ---------------------------------------------------------------------------------------------
// operand sources
// ---------------------------------------------------------------------------------------------
// Plain matrix.  KMAJOR: stored [R][K] (row stride ld); otherwise stored [K][R].
// 16 zero bytes in device memory: where the 8-phase kernel points the direct-to-LDS loads of K rows past the end of a
// contraction-strided operand (ragged last K-tile of the grouped weight gradients)
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

template <typename T, bool KMAJ> struct PlainSrc {
    static constexpr bool KMAJOR = KMAJ;
    static constexpr bool KPERM = false;
    static constexpr bool NPERM = false;
    static constexpr bool SKIP = false;
    __device__ __forceinline__ int64_t k_of_tile(int64_t kt) const { return kt * 64; }
    const T* p;
    int64_t ld, r0, R, K;
    // contraction-strided layout: K row (0..63) inside a K-tile that chunk i of this thread holds
    __device__ __forceinline__ int k_row(int tid, int i, int nthr) const {
        return ((tid + nthr * i) % (64 * Tile64<T>::CH)) / Tile64<T>::CH;
    }
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void prefetch(int, int64_t) {}
    __device__ __forceinline__ const T* addr(int, int t, int row, int ch, int64_t k0, bool& ok) const {
        constexpr int EPC = Tile64<T>::EPC;
        if (KMAJ) {
            const int64_t gr = r0 + 64 * t + row, k = k0 + ch * EPC;
            ok = gr < R && k < K;
            return p + gr * ld + k;
        } else {
            const int64_t k = k0 + row, c = r0 + 64 * t + ch * EPC;
            ok = k < K && c < R;
            return p + k * ld + c;
        }
    }
    // split addressing of the 8-phase kernel: per-lane byte offset of chunk i (row / column clamped into the
    // matrix) + a wave-uniform base for the K offset
    static constexpr bool DYN = false;
    __device__ __forceinline__ uint32_t lane_off(int tid, int i, int nthr) const {
        int t, row, ch;
        const int c = tid + nthr * i, w = c % (64 * Tile64<T>::CH);
        t = c / (64 * Tile64<T>::CH);
        row = w / Tile64<T>::CH;
        ch = (w % Tile64<T>::CH) ^ Tile64<T>::template swz<KMAJ>(row);
        if (KMAJ) {
            const int64_t gr = min(r0 + 64 * t + row, R - 1);
            return (uint32_t)((gr * ld + ch * Tile64<T>::EPC) * (int64_t)sizeof(T));
        }
        int64_t col = r0 + 64 * t + ch * Tile64<T>::EPC;
        if (col >= R) col = 0;
        return (uint32_t)((row * ld + col) * (int64_t)sizeof(T));
    }
    __device__ __forceinline__ const char* k_base(int, int64_t k0) const {
        return reinterpret_cast<const char*>(KMAJ ? p + k0 : p + k0 * ld);
    }
    // always-valid address for a K-step that lies fully inside K: rows / columns beyond R are clamped
    // (they only feed output rows / columns that are never stored)
    __device__ __forceinline__ const T* gaddr(int, int t, int row, int ch, int64_t k0) const {
        constexpr int EPC = Tile64<T>::EPC;
        if (KMAJ) {
            const int64_t gr = min(r0 + 64 * t + row, R - 1);
            return p + gr * ld + k0 + ch * EPC;
        } else {
            int64_t c = r0 + 64 * t + ch * EPC;
            if (c >= R) c = 0;
            return p + (k0 + row) * ld + c;
        }
    }
};

// An operand whose contraction index runs over n_seg separate matrices of seg_tiles K-tiles each, seg_stride elements apart
// (shg_gemm_kseg: C = sum_s A_s . B_s as ONE launch - the decoders' gradient w.r.t. their memory, five `C +=` GEMMs before)
template <typename T, bool KMAJ> struct SegSrc : PlainSrc<T, KMAJ> {
    int seg_tiles;
    uint32_t seg_magic;       // (1 << 20) / seg_tiles + 1: K-tile / seg_tiles for K-tile < 4 095
    int64_t seg_stride;
    __device__ __forceinline__ const char* k_base(int, int64_t k0) const {
        const uint32_t kt = (uint32_t)(k0 >> 6), q = (kt * seg_magic) >> 20;
        const int64_t k = k0 - (int64_t)q * seg_tiles * 64;
        const T* base = this->p + (int64_t)q * seg_stride;
        return reinterpret_cast<const char*>(KMAJ ? base + k : base + k * this->ld);
    }
};

struct ConvGeom {
    int Cin, Hp, Wp;          // padded input plane
    uint32_t inv_cin;         // floor(2^32 / Cin) + 1: k / Cin == umulhi(k, inv_cin) for k < 2^32 / Cin
    int kperm;                // 8-phase kernel: visit the K-tiles channel-block-major (all 45 taps of 64 channels, then the next 64)
    int rpp;                  // forward in position-major row order: rows per spatial position (B To), 0 = every tap for every tile
    int rpt, tv_lo, tv_hi;    // input gradient in frame-major row order: rows per output frame (B H W; 0 = off) and the frames of the
};                            // padded gradient that hold data, [tv_lo, tv_hi): tap kt of output frame t reads frame t + kt
static int conv_k_order(int Cin) {                     // 0: storage order; n: blocks of 64 * 2^(n-1) channels, n <= 3
    const int n = (int)tuning(TUNE_CONV_K_ORDER) & 3;
    return (n >= 1 && n <= 3 && Cin % (64 << (n - 1)) == 0) ? n : 0;
}
static ConvGeom conv_geom(int Cin, int Hp, int Wp) {
    return ConvGeom{Cin, Hp, Wp, (uint32_t)(0x100000000ull / (uint32_t)Cin) + 1u, conv_k_order(Cin), 0, 0, 0, 0};
}
// tap < 45 -> (kt, kh, kw) by multiply-shift (exact on that range): the wave-uniform address math of the
// direct-to-LDS loads sits in the instruction stream of the load phase, where an integer division costs ~30 instructions
__device__ __forceinline__ int64_t tap_offset(const ConvGeom& g, int tap) {
    const int kt = (tap * 57) >> 9, r = tap - 9 * kt, kh = (r * 11) >> 5, kw = r - 3 * kh;
    return ((int64_t)(kt * g.Hp + kh) * g.Wp + kw) * g.Cin;
}
__device__ __forceinline__ uint32_t div_cin(const ConvGeom& g, uint32_t k) { return __umulhi(k, g.inv_cin); }

// (kh, kw) taps that stay inside an H x W grid for position p = h W + w (bit kh * 3 + kw), and their union over the positions
// that rows [m0, m0 + 255] of a position-major problem (rpp rows per position, M rows) belong to
__host__ __device__ __forceinline__ uint32_t conv_valid9(int p, int H, int W) {
    const int h = p / W, w = p - h * W;
    uint32_t m = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            if (h + kh - 1 >= 0 && h + kh - 1 < H && w + kw - 1 >= 0 && w + kw - 1 < W) m |= 1u << (kh * 3 + kw);
    return m;
}
__host__ __device__ __forceinline__ uint32_t conv_tile_mask9(int64_t m0, int64_t M, int rpp, int H, int W) {
    const int64_t m1 = (m0 + 255 < M ? m0 + 255 : M - 1);
    uint32_t m = 0;
    for (int p = (int)(m0 / rpp); p <= (int)(m1 / rpp); ++p) m |= conv_valid9(p, H, W);
    return m;
}

// temporal taps kt (0 .. 4) that read a data frame for ANY of the output frames of rows [m0, m0 + 255] of a frame-major problem:
// a contiguous range [lo, hi] (the union of the frames' ranges [tv_lo - t, tv_hi - 1 - t])
__host__ __device__ __forceinline__ void conv_tile_kt_range(int64_t m0, int64_t M, int rpt, int tv_lo, int tv_hi, int& lo, int& hi) {
    const int64_t m1 = (m0 + 255 < M ? m0 + 255 : M - 1);
    const int t0 = (int)(m0 / rpt), t1 = (int)(m1 / rpt);
    lo = tv_lo - t1;
    hi = tv_hi - 1 - t0;
    if (lo < 0) lo = 0;
    if (hi > 4) hi = 4;
    if (hi < lo) hi = lo;                            // (cannot happen for a frame inside the problem; keeps the list non-empty)
}
// K-tiles a 256-row tile keeps (host: the weighted stream-K plan's table; device: ConvRowSrc::set_tile computes the same)
static int conv_tile_nk(const ConvGeom& g, int64_t m0, int64_t M) {
    if (g.rpt) {
        int lo, hi;
        conv_tile_kt_range(m0, M, g.rpt, g.tv_lo, g.tv_hi, lo, hi);
        return 9 * (hi - lo + 1) * (g.Cin / 64);
    }
    if (g.rpp) return 5 * __builtin_popcount(conv_tile_mask9(m0, M, g.rpp, g.Hp - 2, g.Wp - 2)) * (g.Cin / 64);
    return 45 * (g.Cin / 64);
}

// A operand of the conv forward: rows = output positions (gathered), K = (tap, channel), K contiguous.
template <typename T, int NTHR, bool SKIP_ = false> struct ConvRowSrc {
    static constexpr bool KMAJOR = true;
    static constexpr bool SKIP = SKIP_;      // the instantiation with per-tile tap lists (position-major forward); the plain one keeps
                                             // its registers: 92 against 156 bytes of scratch per lane in the stream-K kernel
    // K = (tap, channel) is a sum: its 64-wide tiles may be visited in any order as long as BOTH operands follow it.  Tap-major (the
    // storage order) re-reads every input line 45 times with a whole sweep over the channels (an XCD's ~10 row panels x Cin x 2 B,
    // 10 MB at 2 048 channels) between two uses: each of them misses the 4 MB L2.  Channel-block-major (g.kperm) keeps the 45 uses
    // of a 128-byte line within 45 consecutive K-tiles.
    static constexpr bool KPERM = true;
    static constexpr bool NPERM = false;
    // Position-major rows (g.rpp rows per spatial position): a tile whose rows all sit at border positions drops the taps that read
    // only the zero border for every one of them.  set_tile: the tile's list of taps - n9 of the 9 (kh, kw), their numbers as
    // nibbles - and the K-tiles it leaves: 5 n9 taps x Cin / 64.  g.rpp == 0 (any row order): all 45.
    __device__ __forceinline__ int set_tile(int64_t m0) {
        if (!SKIP || (!g.rpp && !g.rpt)) return 45 * (g.Cin >> 6);
        if (g.rpt) {                                 // frame-major rows (input gradient): all nine (kh, kw), the temporal taps that read data
            int lo, hi;
            conv_tile_kt_range(m0, M, g.rpt, g.tv_lo, g.tv_hi, lo, hi);
            kt_lo = (uint32_t)lo;
            nib = 0x876543210ull; n9 = 9; ntaps = 9u * (uint32_t)(hi - lo + 1);
            ntaps_magic = (1u << 20) / ntaps + 1;
            n9_magic = (1u << 20) / 9 + 1;
            return (int)(ntaps * (uint32_t)(g.Cin >> 6));
        }
        const uint32_t mask = conv_tile_mask9(m0, M, g.rpp, g.Hp - 2, g.Wp - 2);
        uint64_t nb = 0;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t r = 0; r < 9; ++r)
            if ((mask >> r) & 1u) { nb |= (uint64_t)r << (4 * c); ++c; }
        nib = nb; n9 = c; ntaps = 5 * c;
        ntaps_magic = (1u << 20) / ntaps + 1;
        n9_magic = (1u << 20) / c + 1;
        return (int)(ntaps * (uint32_t)(g.Cin >> 6));
    }
    __device__ __forceinline__ int64_t k_of_tile(int64_t kt) const {
        // blocks of nb = 2^(kperm-1) K-tiles (64 nb channels): tile kt = (block, tap, tile in block).  Branch-free (a branch inside the
        // 8-phase loop splits its phases into basic blocks the scheduler cannot interleave across: +8-24 % per launch, measured)
        const uint32_t sh = (uint32_t)max(g.kperm - 1, 0), k = (uint32_t)kt, q = k >> sh, sub = k - (q << sh);
        if constexpr (!SKIP) {
            const uint32_t blk = (q * 11651u) >> 19, tap = q - blk * 45u;             // q / 45 for q < 20 000
            const uint32_t kp = tap * (uint32_t)g.Cin + (((blk << sh) + sub) << 6);
            return (int64_t)(g.kperm ? kp : k << 6);
        }
        // q / ntaps, then the tap's place in the tile's list -> its number (x / d == (x * ((1 << 20) / d + 1)) >> 20 for x < 4 095)
        const uint32_t blk = (q * ntaps_magic) >> 20, j = q - blk * ntaps;
        const uint32_t k3 = (j * n9_magic) >> 20, s9 = j - k3 * n9;
        const uint32_t tap = 9u * (kt_lo + k3) + ((uint32_t)(nib >> (4u * s9)) & 15u);
        const uint32_t kp = tap * (uint32_t)g.Cin + (((blk << sh) + sub) << 6);
        return (int64_t)(g.kperm ? kp : k << 6);
    }
    const T* x;
    const int32_t* pos;       // [M] padded-input position index of output position m (tap 0)
    int64_t r0, M;
    ConvGeom g;
    int64_t base[Stage<T>::NCH];
    bool okr[Stage<T>::NCH];
    uint32_t n9 = 9, ntaps = 45, ntaps_magic = (1u << 20) / 45 + 1, n9_magic = (1u << 20) / 9 + 1;      // set_tile()
    uint64_t nib = 0x876543210ull;
    uint32_t kt_lo = 0;
    __device__ __forceinline__ void prepare(int tid) {
#pragma unroll
        for (int i = 0; i < Stage<T>::NCH; ++i) {
            int t, row, ch;
            chunk_coord<T, NTHR, true>(tid, i, t, row, ch);
            const int64_t m = r0 + 64 * t + row;
            okr[i] = m < M;
            base[i] = (int64_t)pos[okr[i] ? m : 0] * g.Cin + ch * Tile64<T>::EPC;
        }
    }
    __device__ __forceinline__ void prefetch(int, int64_t) {}
    __device__ __forceinline__ const T* addr(int i, int, int, int, int64_t k0, bool& ok) const {
        const int tap = (int)(k0 / g.Cin);
        ok = okr[i];
        return x + base[i] + tap_offset(g, tap) + (k0 % g.Cin);
    }
    static constexpr bool DYN = false;
    __device__ __forceinline__ uint32_t lane_off(int tid, int i, int nthr) const {
        const int c = tid + nthr * i, w = c % (64 * Tile64<T>::CH);
        const int t = c / (64 * Tile64<T>::CH), row = w / Tile64<T>::CH, ch = (w % Tile64<T>::CH) ^ (row & 7);
        const int64_t m = r0 + 64 * t + row;
        return (uint32_t)(((int64_t)pos[m < M ? m : 0] * g.Cin + ch * Tile64<T>::EPC) * (int64_t)sizeof(T));
    }
    __device__ __forceinline__ const char* k_base(int, int64_t k0) const {
        const uint32_t k = (uint32_t)k0, tap = div_cin(g, k);
        return reinterpret_cast<const char*>(x + tap_offset(g, (int)tap) + (k - tap * (uint32_t)g.Cin));
    }
    __device__ __forceinline__ const T* gaddr(int i, int, int, int, int64_t k0) const {
        const uint32_t k = (uint32_t)k0, tap = div_cin(g, k);   // wave-uniform 32-bit scalar math
        return x + base[i] + tap_offset(g, (int)tap) + (k - tap * (uint32_t)g.Cin);   // rows beyond M read position 0 (valid)
    }
};

// B operand of the conv weight gradient: rows = reduction index (output positions, gathered),
// columns = (tap, channel).  Stored with the contraction index strided.
template <typename T, int NTHR> struct ConvColSrc {
    static constexpr bool KMAJOR = false;
    const T* x;
    const int32_t* pos;
    int64_t r0, R, K;         // r0: first (tap, channel) column of this block; R = 45 * Cin; K = M
    ConvGeom g;
    int32_t posreg[Stage<T>::NCH];      // gather positions of the NEXT K-step, fetched one step ahead
    int64_t cbase = 0;                  // a launch over the column blocks [cbase, cbase + N) of the problem: r0 counts from cbase
    // 8-phase kernel: the launch's 256-column blocks in channel-block-major order (nblk = Cin / 256 > 0): block L = lblk0 + bn of the
    // launch is (channel block L / 45, tap L % 45), i.e. columns 256 (tap nblk + channel block) of the problem.  Neighbouring tiles
    // - the ones an XCD runs at the same time - then gather the SAME input lines shifted by a tap, instead of eight different
    // 512-byte pieces of every position.
    static constexpr bool NPERM = true;
    int nblk = 0, lblk0 = 0;
    // lpt: the taps of a channel block in the order centre (all 49 positions), edges (42), corners (36) - with the zero-border
    // skipping (tpp) the tiles of a launch differ in length by up to 27 %, and the workgroups are handed out in this order: longest
    // first keeps the last round short (position-major rows without it: 2.25 -> 2.08 ms for conv1, with it see DESIGN.md 9 (10))
    int lpt = 0;
    __device__ __forceinline__ int64_t col_of_lblock(int64_t lb) const {            // lb: logical block of the whole problem
        if (!nblk) return lb * 256;
        const uint32_t L = (uint32_t)lb, c = (L * 11651u) >> 19;
        uint32_t t = L - c * 45u;
        if (lpt) {
            const uint32_t j = t < 25u ? t - 5u : t - 25u, k3 = j >> 2, e = j & 3u;
            t = t < 5u ? 9u * t + 4u : (t < 25u ? 9u * k3 + 1u + 2u * e : 9u * k3 + (e & 1u) * 2u + (e >> 1) * 6u);
        }
        return (int64_t)(t * (uint32_t)nblk + c) * 256;
    }
    __device__ __forceinline__ int64_t col_of_block(int64_t bn) const { return col_of_lblock(lblk0 + bn); }
    // overwrite mode of the weight gradient (shg_conv3d_k533_wgrad_sumsq): the launch over the whole rounds of column blocks also
    // zeroes the column blocks [zero_blk0, zero_blk0 + zero_blks) of the `zero_rows` output rows, which the split launch behind it
    // then adds into with atomics
    float* zero_base = nullptr;
    int zero_blk0 = 0, zero_blks = 0, zero_rows = 0;
    // Rows (the contraction index) in POSITION-MAJOR order (row = ((h W + w) B + b) To + t, shg_conv3d_k533_prepare_ex order 1) with
    // tpp = B To / 64 whole K-tiles per spatial position: a tap that reaches into the zero border for position (h, w) gathers only
    // zeros there, so a tile (one tap) contracts over the nh x nw positions its tap stays inside for and skips the rest - 4 of 9
    // (kh, kw) taps lose a row and a column of the 7 x 7 grid, 4 lose one: 18 % of the products of a padded 3 x 3 window on 7 x 7
    // are with zeros.  The remaining terms are summed in the same order: bit-identical results.
    int tpp = 0, tpp_magic = 0, Hin = 0, Win = 0;          // tpp == 0: every K-tile (any row order)
    int nw_ = 1, nw_magic = (1 << 20) + 1, h_lo = 0, w_lo = 0;      // per tile, set_tap(); magics: x / d == (x * ((1 << 20) / d + 1)) >> 20 for x < 4 095
    __device__ __forceinline__ int set_tap(int tap) {       // -> K-tiles of a tile of this tap
        const int kt3 = (tap * 57) >> 9, r = tap - 9 * kt3, kh = (r * 11) >> 5, kw = r - 3 * kh;
        const int nh = Hin - (kh != 1 ? 1 : 0);
        nw_ = Win - (kw != 1 ? 1 : 0);
        nw_magic = (1 << 20) / nw_ + 1;
        h_lo = kh == 0 ? 1 : 0;
        w_lo = kw == 0 ? 1 : 0;
        return nh * nw_ * tpp;
    }
    __device__ __forceinline__ int64_t k_of_tile(int64_t kt) const {       // kt: index into the tile's list of K-tiles -> row offset
        if (!tpp) return kt * 64;
        const uint32_t k = (uint32_t)kt, q = (k * (uint32_t)tpp_magic) >> 20, sub = k - q * (uint32_t)tpp;     // position in the list, tile in it
        const uint32_t hh = (q * (uint32_t)nw_magic) >> 20, ww = q - hh * (uint32_t)nw_;
        const uint32_t pos_i = ((uint32_t)h_lo + hh) * (uint32_t)Win + (uint32_t)w_lo + ww;
        return (int64_t)((pos_i * (uint32_t)tpp + sub) << 6);
    }
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void prefetch(int tid, int64_t k0) {
#pragma unroll
        for (int i = 0; i < Stage<T>::NCH; ++i) {
            int t, row, ch;
            chunk_coord<T, NTHR, false>(tid, i, t, row, ch);
            posreg[i] = pos[min(k0 + row, K - 1)];
        }
    }
    __device__ __forceinline__ const T* addr(int, int t, int row, int ch, int64_t k0, bool& ok) const {
        const int64_t m = k0 + row;
        const int64_t c = cbase + r0 + 64 * t + ch * Tile64<T>::EPC;
        ok = m < K && c < R;
        const int64_t mm = ok ? m : 0;
        const int tap = (int)(c / g.Cin);
        return x + (int64_t)pos[mm] * g.Cin + tap_offset(g, tap) + (c % g.Cin);
    }
    // 8-phase kernel (512 threads): every chunk of a thread sits in row tid >> 3 of its Tile64, so one gather
    // position per thread and K-tile serves all of them; the column block (tap, channel base) is wave-uniform.
    static constexpr bool DYN = true;
    __device__ __forceinline__ int dyn_row(int tid) const { return tid >> 3; }
    __device__ __forceinline__ const int32_t* dyn_ptr(int tid, int64_t k0) const { return pos + min(k0 + (tid >> 3), K - 1); }
    __device__ __forceinline__ uint32_t dyn_off(int tid, int32_t p) const {
        const int row = tid >> 3, ch = (tid & 7) ^ Tile64<T>::swz_ks(row);
        return (uint32_t)(((int64_t)p * g.Cin + ch * Tile64<T>::EPC) * (int64_t)sizeof(T));
    }
    __device__ __forceinline__ uint32_t lane_off(int, int, int) const { return 0; }
    __device__ __forceinline__ const char* k_base(int i, int64_t) const {
        uint32_t cb = (uint32_t)(cbase + r0 + 64 * i);
        if (cb >= (uint32_t)R) cb = 0;
        const uint32_t tap = div_cin(g, cb), c0 = cb - tap * (uint32_t)g.Cin;
        return reinterpret_cast<const char*>(x + tap_offset(g, (int)tap) + c0);
    }
    __device__ __forceinline__ const T* gaddr(int i, int t, int, int ch, int64_t) const {
        // a 64-wide column block never straddles a tap (Cin % 64 == 0): tap and channel base are uniform
        uint32_t cb = (uint32_t)(cbase + r0 + 64 * t);
        if (cb >= (uint32_t)R) cb = 0;
        const uint32_t tap = div_cin(g, cb), c0 = cb - tap * (uint32_t)g.Cin;
        return x + (int64_t)posreg[i] * g.Cin + tap_offset(g, (int)tap) + c0 + ch * Tile64<T>::EPC;
    }
};

// B operand of the conv INPUT gradient, read straight from the forward weight W[co][tap][ci]:
// contraction index k = (tap', co) with tap' the flipped tap (44 - tap), columns n = ci.  For a fixed
// tap' this is a [co][ci] matrix with row stride 45*Cin - "contraction strided" - so the flipped,
// transposed weight copy of a textbook conv-transpose is never materialised.
template <typename T> struct ConvWeightColSrc {
    static constexpr bool KMAJOR = false;
    static constexpr bool NPERM = false;
    const T* w;
    int64_t r0, R, K;         // R = Cin (columns), K = 45 * Cout
    int Cin, Cout;
    uint32_t inv_cout;        // floor(2^32 / Cout) + 1
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void prefetch(int, int64_t) {}
    __device__ __forceinline__ const T* addr(int, int t, int row, int ch, int64_t k0, bool& ok) const {
        const int64_t k = k0 + row;
        const int64_t c = r0 + 64 * t + ch * Tile64<T>::EPC;
        ok = k < K && c < R;
        const int64_t kk = ok ? k : 0;
        const int tapf = 44 - (int)(kk / Cout);
        const int64_t co = kk % Cout;
        return w + (co * 45 + tapf) * (int64_t)Cin + c;
    }
    static constexpr bool DYN = false;
    __device__ __forceinline__ uint32_t lane_off(int tid, int i, int nthr) const {
        const int c = tid + nthr * i, w_ = c % (64 * Tile64<T>::CH);
        const int t = c / (64 * Tile64<T>::CH), row = w_ / Tile64<T>::CH, ch = (w_ % Tile64<T>::CH) ^ Tile64<T>::swz_ks(row);
        int64_t col = r0 + 64 * t + ch * Tile64<T>::EPC;
        if (col >= R) col = 0;
        return (uint32_t)(((int64_t)row * 45 * Cin + col) * (int64_t)sizeof(T));
    }
    __device__ __forceinline__ const char* k_base(int, int64_t k0) const {
        const uint32_t k = (uint32_t)k0, tap = __umulhi(k, inv_cout), co0 = k - tap * (uint32_t)Cout;
        return reinterpret_cast<const char*>(w + ((int64_t)co0 * 45 + (44 - (int)tap)) * (int64_t)Cin);
    }
    __device__ __forceinline__ const T* gaddr(int, int t, int row, int ch, int64_t k0) const {
        // Cout % 64 == 0: a K-step stays inside one tap
        const uint32_t k = (uint32_t)k0, tap = __umulhi(k, inv_cout), co0 = k - tap * (uint32_t)Cout;
        int64_t c = r0 + 64 * t + ch * Tile64<T>::EPC;
        if (c >= R) c = 0;
        return w + ((int64_t)(co0 + row) * 45 + (44 - (int)tap)) * (int64_t)Cin + c;
    }
};

// ---------------------------------------------------------------------------------------------
// epilogue description
// ---------------------------------------------------------------------------------------------
template <typename TC> struct Epilogue {
    TC* c;
    int64_t ldc;
    const float* bias;        // [N] or null
    const int32_t* crow;      // optional output-row remap (conv into a padded buffer): row m -> crow[m]
    int act;
    int accumulate;
    int vec_ok;               // ldc and base pointer allow vector stores of 4 elements
    TC* pre;                  // optional second output [M, N] (ld = N): the value BEFORE the activation
    int atomic;               // split-K: fp32 atomic adds into C (C must hold the running sum already)
    // activation backward fused into an input-gradient GEMM: C = (A . B) * act'(gpre[m, n]) with gpre [M, N]
    // (ld = N) the saved pre-activation, and csum[n] += sum_m C[m, n] (the bias gradient), by atomics
    const TC* gpre;
    float* csum;
    // dropout on the activation's output (forward) / on the incoming gradient (gpre form), same counter-based masks
    // as the stand-alone bias_act kernels: element index = m * N + n
    uint32_t drop_thr;
    float drop_scale;
    const uint64_t* seed_state;
    uint64_t stream_id;
    int no_side;              // 1: the row writers' up-front-load forms are switched off ("epilogue_side" tuning switch, A/B runs)
    int save_grad;            // forward: `pre` receives act'(u) instead of u (SHG_ACT_SAVE_GRAD)
    double* sumsq;            // conv weight gradient on the 8-phase kernel, plain stores of whole tiles: *sumsq += sum of C[m, n]^2
    const int32_t* prow;      // optional row remap of `pre`: row m of the second output goes to row prow[m]
};

__device__ __forceinline__ float act_grad_rt(float u, int act, bool fast) {
    if (act == SHG_ACT_GELU) return fast ? gelu_fast_grad(u) : gelu_erf_grad(u);
    if (act == SHG_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    return 1.f;
}

constexpr int STG_LD = 68;                         // fp32 row stride of the epilogue staging tile (64 + 4 pad)
constexpr int STG_BYTES = 4 * 64 * STG_LD * 4;     // one 64 x 64 staging tile per wave

// Compile-time forms for the row writers: with the run-time `act` inside their unrolled element loops the compiler emitted a
// scalar compare + branch per ELEMENT (292 compares / ~600 branches in the epilogue of one instantiation - 8 us of the 14 us an
// epilogue cost per round of tiles, against 2 us for its global stores); RowWriter::run now switches ONCE per call.
template <int ACT, bool FAST> __device__ __forceinline__ float act_ct(float x) {
    if (ACT == SHG_ACT_GELU) return FAST ? gelu_fast(x) : gelu_erf(x);
    if (ACT == SHG_ACT_RELU) return fmaxf(x, 0.f);
    return x;
}
template <int ACT, bool FAST> __device__ __forceinline__ float act_grad_ct(float u) {
    if (ACT == SHG_ACT_GELU) return FAST ? gelu_fast_grad(u) : gelu_erf_grad(u);
    if (ACT == SHG_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (ACT == SHG_ACT_SAVED_GRAD) return u;          // the forward stored the derivative itself
    return 1.f;
}
// activation and its derivative together (forward epilogue with SHG_ACT_SAVE_GRAD): the fast GELU's parts serve both
template <int ACT> __device__ __forceinline__ void act_and_grad_fast(float x, float& y, float& g) {
    if (ACT == SHG_ACT_GELU) {
        float cdf, e;
        gelu_fast_parts(x, cdf, e);
        y = x * cdf;
        g = fmaf(x * 0.39894228040143267794f, e, cdf);
    } else if (ACT == SHG_ACT_RELU) {
        y = fmaxf(x, 0.f);
        g = x > 0.f ? 1.f : 0.f;
    } else {
        y = x;
        g = 1.f;
    }
}
template <bool FAST = false> __device__ __forceinline__ float apply_act(float x, int act) {
    if (act == SHG_ACT_GELU) return FAST ? gelu_fast(x) : gelu_erf(x);
    if (act == SHG_ACT_RELU) return fmaxf(x, 0.f);
    return x;
}

// Writes one wave's 64 x 64 result (staged in LDS as fp32 [64][STG_LD]) to global memory in whole
// 128/256-byte row segments.
template <typename TC> struct RowWriter;
template <> struct RowWriter<bf16_t> {
    // csum_carry (8 floats per lane, zero-initialised by the caller): the column sums of this piece are added to it instead
    // of going to memory; the caller flushes them once with flush_csum (fewer atomics on the shared bias-gradient vector)
    // SIDE: compile the whole-tile forms that read a second operand (accumulate / activation backward) with their loads up front
    // (off in the stream-K convolution kernel, which has no register to spare and never uses them)
    // ROWS: rows of the staged piece (64, or 48 for the 192-row tile of the 8-phase kernel)
    template <bool SIDE = true, int ROWS = 64>
    __device__ __forceinline__ static void run(const float* stage, const Epilogue<bf16_t>& ep, int64_t mbase, int64_t nbase, int64_t M,
                               int64_t N, int lane, int gap = 0, float* csum_carry = nullptr) {
        if (ep.act == SHG_ACT_GELU) run_act<SHG_ACT_GELU, SIDE, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap, csum_carry);
        else if (ep.act == SHG_ACT_RELU) run_act<SHG_ACT_RELU, SIDE, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap, csum_carry);
        else if (SIDE && ep.act == SHG_ACT_SAVED_GRAD) run_act<SHG_ACT_SAVED_GRAD, SIDE, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap, csum_carry);
        else run_act<SHG_ACT_NONE, SIDE, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap, csum_carry);
    }
    template <int ACT, bool SIDE, int ROWS>
    __device__ __forceinline__ static void run_act(const float* stage, const Epilogue<bf16_t>& ep, int64_t mbase, int64_t nbase, int64_t M,
                                   int64_t N, int lane, int gap, float* csum_carry) {
        float csum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool drop = ep.drop_thr != 0;
        const uint64_t dseed = ep.drop_thr ? dropout_seed(ep.seed_state, ep.stream_id) : 0;
        // a lane keeps its 8 columns through all 8 row passes: their bias is loaded once, ahead of the loop (per-element
        // loads inside it were 8 dependent L2 round trips per pass: 15 us of a 50 us 8192 x 2048 x 768 launch)
        const int col = (lane & 7) * 8;
        const int64_t n = nbase + col + (col >= 32 ? gap : 0);
        const int nv = (int)max((int64_t)0, min((int64_t)8, N - n));
        float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ep.bias) {
            if (nv == 8 && (reinterpret_cast<uintptr_t>(ep.bias + n) & 15) == 0) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + n), b1 = *reinterpret_cast<const f32x4*>(ep.bias + n + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { bias8[r] = b0[r]; bias8[4 + r] = b1[r]; }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < nv) bias8[r] = ep.bias[n + r];
            }
        }
        // the plain case (bias + activation, whole 16-byte vectors, all 64 rows inside the matrix): the eight passes' LDS reads
        // are issued together and nothing but the activation sits between them and the stores
        constexpr int NP = ROWS / 8;                     // row passes: a wave writes 8 rows per pass
        if (!ep.gpre && !ep.csum && !ep.accumulate && !drop && !ep.pre && !ep.crow && ep.vec_ok && nv == 8 && mbase + ROWS <= M) {
            f32x4 va[NP], vb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int row = 8 * p + (lane >> 3);
                va[p] = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col);
                vb[p] = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col + 4);
            }
            bf16_t* dst0 = ep.c + (mbase + (lane >> 3)) * ep.ldc + n;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (bf16_t)act_ct<ACT, true>(va[p][r] + bias8[r]);
                    o[4 + r] = (bf16_t)act_ct<ACT, true>(vb[p][r] + bias8[4 + r]);
                }
                *reinterpret_cast<bf16x8*>(dst0 + (int64_t)(8 * p) * ep.ldc) = o;
            }
            return;
        }
        // the two input-gradient forms of the BERT blocks on whole tiles: `C += A.B` (the residual gradient is already in C) and
        // `C = (A.B) * act'(gpre)` with the bias-gradient column sums.  Their global READS (old C / the saved pre-activation) are
        // all issued before the first row pass - in the general loop below each pass waits for its own load, eight dependent L2
        // round trips per call - with constant indices only, so that the eight vectors stay in registers (32 VGPRs)
        if (SIDE && !ep.no_side && (ep.accumulate || ep.gpre) && !(ep.accumulate && ep.gpre) && !drop && !ep.pre && !ep.crow && ep.vec_ok && nv == 8 && mbase + ROWS <= M) {
            bf16_t* dst0 = ep.c + (mbase + (lane >> 3)) * ep.ldc + n;
            const bf16_t* src0 = ep.gpre ? ep.gpre + (mbase + (lane >> 3)) * N + n : dst0;
            const int64_t lds_ = ep.gpre ? N : ep.ldc;
            const bool dact = ep.gpre != nullptr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // half of the row passes at a time: 12-16 VGPRs of loaded vectors in flight
                constexpr int PH = NP / 2;                                   // 4 (64 rows) or 3 (48 rows)
                bf16x8 s0 = *reinterpret_cast<const bf16x8*>(src0 + (int64_t)(8 * PH * h) * lds_);
                bf16x8 s1 = *reinterpret_cast<const bf16x8*>(src0 + (int64_t)(8 * PH * h + 8) * lds_);
                bf16x8 s2 = *reinterpret_cast<const bf16x8*>(src0 + (int64_t)(8 * PH * h + 16) * lds_);
                bf16x8 s3 = s2;
                if (PH == 4) s3 = *reinterpret_cast<const bf16x8*>(src0 + (int64_t)(8 * PH * h + 24) * lds_);
                auto pass = [&](int p, const bf16x8& side) {
                    const int row = 8 * p + (lane >> 3);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col + 4);
                    const float u[8] = {a[0] + bias8[0], a[1] + bias8[1], a[2] + bias8[2], a[3] + bias8[3],
                                        b[0] + bias8[4], b[1] + bias8[5], b[2] + bias8[6], b[3] + bias8[7]};
                    bf16x8 o;
                    if (dact) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            o[r] = (bf16_t)(u[r] * act_grad_ct<ACT, true>((float)side[r]));
                            csum8[r] += (float)o[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 8; ++r) o[r] = (bf16_t)(act_ct<ACT, true>(u[r]) + (float)side[r]);
                    }
                    *reinterpret_cast<bf16x8*>(dst0 + (int64_t)(8 * p) * ep.ldc) = o;
                };
                // (scheduling fences: left alone, the scheduler gathers the LDS reads of all passes at the top - 64 more live
                //  registers beside the second half's accumulators - and the kernel spills 528 bytes per lane)
                __builtin_amdgcn_sched_barrier(0);
                pass(PH * h, s0);
                __builtin_amdgcn_sched_barrier(0);
                pass(PH * h + 1, s1);
                __builtin_amdgcn_sched_barrier(0);
                pass(PH * h + 2, s2);
                __builtin_amdgcn_sched_barrier(0);
                if (PH == 4) pass(PH * h + 3, s3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ep.csum) {
                if (csum_carry) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) csum_carry[r] += csum8[r];
                } else {
                    flush_csum(csum8, ep, nbase, N, lane, gap);
                }
            }
            return;
        }
#pragma unroll 2
        for (int p = 0; p < NP; ++p) {
            const int row = 8 * p + (lane >> 3);
            const int64_t m = mbase + row;
            if (m >= M || n >= N) continue;
            const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col);
            const f32x4 b = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col + 4);
            float u[8] = {a[0] + bias8[0], a[1] + bias8[1], a[2] + bias8[2], a[3] + bias8[3],
                          b[0] + bias8[4], b[1] + bias8[5], b[2] + bias8[6], b[3] + bias8[7]};
            const int64_t crow = ep.crow ? (int64_t)ep.crow[m] : m;
            bf16_t* dst = ep.c + crow * ep.ldc + n;
            bf16_t* pre = ep.pre ? ep.pre + (ep.prow ? (int64_t)ep.prow[m] : m) * N + n : nullptr;
            if (ep.gpre) {                               // (nv == 8: N % 8 == 0 is checked on the host)
                const bf16x8 gp = *reinterpret_cast<const bf16x8*>(ep.gpre + m * N + n);
                bf16x8 o;
                if (drop) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) u[r] = dropout_keep_run(dseed, (uint64_t)(m * N + n), r, ep.drop_thr) ? u[r] * ep.drop_scale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    u[r] *= act_grad_ct<ACT, true>((float)gp[r]);
                    o[r] = (bf16_t)u[r];
                    csum8[r] += (float)o[r];
                }
                *reinterpret_cast<bf16x8*>(dst) = o;
                continue;
            }
            if (nv == 8 && ep.vec_ok) {
                bf16x8 o, q;
                if (ep.accumulate) {                       // C += A.B in fp32, rounded once (residual-gradient sums)
                    const bf16x8 old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16_t)(act_ct<ACT, true>(u[r]) + (float)old[r]);
                } else if (drop) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float a = act_ct<ACT, true>(u[r]);
                        o[r] = (bf16_t)(dropout_keep_run(dseed, (uint64_t)(m * N + n), r, ep.drop_thr) ? a * ep.drop_scale : 0.f);
                    }
                } else if (pre && ep.save_grad) {             // activation and derivative from shared parts; `pre` keeps the derivative
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float y, gd;
                        act_and_grad_fast<ACT>(u[r], y, gd);
                        o[r] = (bf16_t)y;
                        u[r] = gd;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16_t)act_ct<ACT, true>(u[r]);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) q[r] = (bf16_t)u[r];
                *reinterpret_cast<bf16x8*>(dst) = o;
                if (pre) *reinterpret_cast<bf16x8*>(pre) = q;
            } else {
                for (int r = 0; r < nv; ++r) {
                    const float x = act_ct<ACT, true>(u[r]);
                    dst[r] = (bf16_t)(ep.accumulate ? x + (float)dst[r] : x);
                    if (pre) pre[r] = (bf16_t)u[r];
                }
            }
        }
        if (ep.csum) {
            if (csum_carry) {
#pragma unroll
                for (int r = 0; r < 8; ++r) csum_carry[r] += csum8[r];
            } else {
                flush_csum(csum8, ep, nbase, N, lane, gap);
            }
        }
    }
    // lanes l, l+8, .. l+56 hold partial sums of the same 8 columns: after the butterfly every lane has the totals of its
    // column group, and lane L adds element L >> 3 of it - ONE atomic instruction over 64 consecutive floats per wave (eight
    // instructions of 8 lanes with a 32-byte stride cost 25 us on a 150-tile launch: the L2 serialises atomics per line)
    __device__ static void flush_csum(float* csum8, const Epilogue<bf16_t>& ep, int64_t nbase, int64_t N, int lane, int gap = 0) {
        const int col = (lane & 7) * 8, sel = lane >> 3;
        const int64_t n = nbase + col + (col >= 32 ? gap : 0) + sel;
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float v = csum8[r];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            mine = sel == r ? v : mine;
        }
        if (n < N) atomicAdd(ep.csum + n, mine);
    }
};
template <> struct RowWriter<float> {
    __device__ static void flush_csum(float*, const Epilogue<float>&, int64_t, int64_t, int, int = 0) {}
    template <bool SIDE = true, int ROWS = 64>
    __device__ __forceinline__ static void run(const float* stage, const Epilogue<float>& ep, int64_t mbase, int64_t nbase, int64_t M,
                               int64_t N, int lane, int gap = 0, float* /*csum_carry: fp32 outputs flush per piece*/ = nullptr) {
        if (ep.act == SHG_ACT_GELU) run_act<SHG_ACT_GELU, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap);
        else if (ep.act == SHG_ACT_RELU) run_act<SHG_ACT_RELU, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap);
        else run_act<SHG_ACT_NONE, ROWS>(stage, ep, mbase, nbase, M, N, lane, gap);
    }
    template <int ACT, int ROWS>
    __device__ __forceinline__ static void run_act(const float* stage, const Epilogue<float>& ep, int64_t mbase, int64_t nbase, int64_t M,
                                   int64_t N, int lane, int gap) {
        const bool drop = ep.drop_thr != 0;
        if (ep.atomic) {                               // one 256-byte row segment per wave instruction
            const int64_t n = nbase + lane + (lane >= 32 ? gap : 0);
            for (int row = 0; row < ROWS; ++row) {
                const int64_t m = mbase + row;
                if (m < M && n < N) atomicAdd(ep.c + m * ep.ldc + n, stage[row * STG_LD + lane]);
            }
            return;
        }
        float csum4[4] = {0.f, 0.f, 0.f, 0.f};
        const uint64_t dseed = ep.drop_thr ? dropout_seed(ep.seed_state, ep.stream_id) : 0;
        const int col = (lane & 15) * 4;                // (the lane's columns and their bias: the same in every row pass)
        const int64_t n = nbase + col + (col >= 32 ? gap : 0);
        const int nv = (int)max((int64_t)0, min((int64_t)4, N - n));
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (ep.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nv) bias4[r] = ep.bias[n + r];
        }
#pragma unroll 2
        for (int p = 0; p < ROWS / 4; ++p) {
            const int row = 4 * p + (lane >> 4);
            const int64_t m = mbase + row;
            if (m >= M || n >= N) continue;
            const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * STG_LD + col);
            float u[4] = {a[0] + bias4[0], a[1] + bias4[1], a[2] + bias4[2], a[3] + bias4[3]};
            const int64_t crow = ep.crow ? (int64_t)ep.crow[m] : m;
            float* dst = ep.c + crow * ep.ldc + n;
            float* pre = ep.pre ? ep.pre + (ep.prow ? (int64_t)ep.prow[m] : m) * N + n : nullptr;
            if (ep.gpre) {                               // (N % 8 == 0 is checked on the host)
                const f32x4 gp = *reinterpret_cast<const f32x4*>(ep.gpre + m * N + n);
                if (drop) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) u[r] = dropout_keep_run(dseed, (uint64_t)(m * N + n), r, ep.drop_thr) ? u[r] * ep.drop_scale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u[r] *= act_grad_ct<ACT, false>(gp[r]);
                    csum4[r] += u[r];
                }
                *reinterpret_cast<f32x4*>(dst) = f32x4{u[0], u[1], u[2], u[3]};
                continue;
            }
            if (nv == 4 && ep.vec_ok) {
                f32x4 o = {act_ct<ACT, false>(u[0]), act_ct<ACT, false>(u[1]), act_ct<ACT, false>(u[2]), act_ct<ACT, false>(u[3])};
                if (drop) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = dropout_keep_run(dseed, (uint64_t)(m * N + n), r, ep.drop_thr) ? o[r] * ep.drop_scale : 0.f;
                }
                if (ep.accumulate) o += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = o;
                if (pre && (N % 4) == 0) *reinterpret_cast<f32x4*>(pre) = f32x4{u[0], u[1], u[2], u[3]};
                else if (pre) for (int r = 0; r < 4; ++r) pre[r] = u[r];
            } else {
                for (int r = 0; r < nv; ++r) {
                    const float x = act_ct<ACT, false>(u[r]);
                    dst[r] = ep.accumulate ? dst[r] + x : x;
                    if (pre) pre[r] = u[r];
                }
            }
        }
        if (ep.csum) {                                   // lanes l, l+16, l+32, l+48 hold the same 4 columns
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = csum4[r];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 16 && n + r < N) atomicAdd(ep.csum + n + r, v);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// TM x TN: operand tiles of 64 rows stacked per workgroup (block tile = 64 TM x 64 TN);
// WM x WN: wave grid.  Two configurations are instantiated: 2x2 tiles / 2x2 waves (128 x 128, 256
// threads, 2 workgroups per CU) and, for bf16 problems with enough 256 x 256 tiles to fill the
// chip, 4x4 tiles / 2x4 waves (each wave 128 x 64; half the operand bytes per flop).
template <typename T, typename TC, typename SrcA, typename SrcB, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N,
                                                            int64_t K, int grid_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TL = Tile64<T>;
    constexpr int NTHR = 64 * WM * WN;
    constexpr int NCH = Stage<T>::NCH;
    static_assert(TM * 64 * TL::CH / NTHR == NCH && TN * 64 * TL::CH / NTHR == NCH, "unsupported tile / thread ratio");
    constexpr int IM = TM * 64 / WM / 16, JN = TN * 64 / WN / 16;      // 16 x 16 accumulator blocks per wave
    static_assert(JN == 4 && IM % 4 == 0, "the epilogue stages 64 x 64 pieces");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wr = wave / WN, wc = wave % WN;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2),
    // so give each XCD a CONTIGUOUS range of tile ids and walk N fastest inside it: the gn tiles that
    // re-read one A row-panel (for the convolution: the gathered activations, re-read once per
    // 128 output channels) run back-to-back on one XCD and hit its L2 instead of the Infinity Cache;
    // the weight panels are small per K-step and stay L2-resident on every XCD.  Bijective for any grid.
    // grid_m < 0: walk M fastest instead (few row tiles, many column tiles - the weight-gradient shapes - so
    // that the tiles sharing a B column-panel sit next to each other)
    const int64_t n_tiles = (int64_t)gridDim.x, gm_t = grid_m < 0 ? -grid_m : grid_m, gn_t = n_tiles / gm_t;
    const int64_t xq = n_tiles / 8, xr = n_tiles % 8, xcd = blockIdx.x % 8;
    const int64_t tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + blockIdx.x / 8;
    const int64_t bm = grid_m < 0 ? tile % gm_t : tile / gn_t, bn = grid_m < 0 ? tile / gm_t : tile % gn_t;
    const int64_t m0 = bm * (TM * 64), n0 = bn * (TN * 64);
    sa.r0 = m0;
    sb.r0 = n0;
    sa.prepare(tid);
    sb.prepare(tid);

    f32x4 acc[IM][JN];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- operand staging -------------------------------------------------------------------------
    // Two LDS stages of (A: 2 x Tile64, B: 2 x Tile64).  Full K-steps go global -> LDS directly
    // (global_load_lds, 16 B per lane, one wave instruction = 64 consecutive slots; the XOR swizzle
    // is applied to the per-lane SOURCE address); a ragged last K-step goes through registers with
    // zero fill.  One barrier per K-step: the loads of step t+1 are in flight during the MFMAs of t.
    constexpr int STAGE_BYTES = (TM + TN) * TL::BYTES;
    constexpr int B_OFF = TM * TL::BYTES;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // source address = wave-uniform base of the K-step + a per-lane byte offset computed once (the address math
    // of 8 loads per K-step otherwise costs as many VALU cycles as the step's MFMAs)
    uint32_t offa[NCH], offb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        offa[i] = sa.lane_off(tid, i, NTHR);
        offb[i] = sb.lane_off(tid, i, NTHR);
    }
    auto stage_async = [&](int buf, int64_t k0) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            char* dst = base + (NTHR * i + 64 * wave_u) * 16;         // wave-uniform; the hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((glb_ptr)(sa.k_base(i, k0) + offa[i]), (lds_ptr)dst, 16, 0, 0);
            if constexpr (SrcB::DYN) {                                // per-K-step gather positions (conv weight gradient)
                int t, row, chb;
                chunk_coord<T, NTHR, SrcB::KMAJOR>(tid, i, t, row, chb);
                __builtin_amdgcn_global_load_lds((glb_ptr)sb.gaddr(i, t, row, chb, k0), (lds_ptr)(dst + B_OFF), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((glb_ptr)(sb.k_base(i, k0) + offb[i]), (lds_ptr)(dst + B_OFF), 16, 0, 0);
            }
        }
    };
    auto stage_ragged = [&](int buf, int64_t k0) {                     // predicated, zero-filled (K tail)
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int t, row, cha, chb;
            chunk_coord<T, NTHR, SrcA::KMAJOR>(tid, i, t, row, cha);
            chunk_coord<T, NTHR, SrcB::KMAJOR>(tid, i, t, row, chb);
            bool oka, okb;
            const T* pa = sa.addr(i, t, row, cha, k0, oka);
            const T* pb = sb.addr(i, t, row, chb, k0, okb);
            const uint4 va = oka ? *reinterpret_cast<const uint4*>(pa) : make_uint4(0, 0, 0, 0);
            const uint4 vb = okb ? *reinterpret_cast<const uint4*>(pb) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(base + (tid + NTHR * i) * 16) = va;
            *reinterpret_cast<uint4*>(base + B_OFF + (tid + NTHR * i) * 16) = vb;
        }
    };
    auto stage = [&](int buf, int64_t kt) {
        const int64_t k0 = kt * BK;
        if (k0 + BK <= K) stage_async(buf, k0); else stage_ragged(buf, k0);
    };

    const int64_t nk_all = (K + BK - 1) / BK;
    const int64_t per_split = (nk_all + gridDim.y - 1) / gridDim.y;
    const int64_t kt_begin = (int64_t)blockIdx.y * per_split;
    const int64_t nk = min(nk_all, kt_begin + per_split);
    if (kt_begin >= nk) return;                       // whole block: nothing to add
    sa.prefetch(tid, kt_begin * BK);
    sb.prefetch(tid, kt_begin * BK);
    stage(0, kt_begin);
    if (kt_begin + 1 < nk) { sa.prefetch(tid, (kt_begin + 1) * BK); sb.prefetch(tid, (kt_begin + 1) * BK); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    // 8-wave configuration: waves w and w+4 share a SIMD.  If both run the identical sequence they issue their
    // direct-to-LDS loads (each a multi-ten-cycle issue) and their MFMAs in lockstep and compete; the second
    // half of the workgroup therefore issues the next stage's loads BETWEEN its two MFMA batches, so that on
    // every SIMD one wave is in a matrix phase while its partner is in a load phase.
    // (measured: helps the layouts with transposed LDS reads - dgrad / wgrad forms, 8-15 % - and is neutral to
    // slightly negative when both operands are contraction-contiguous, so it is enabled per layout)
    const bool late_loader = (WM * WN == 8) && (!SrcA::KMAJOR || !SrcB::KMAJOR) && (wave_u >= 4);
    for (int64_t kt = kt_begin; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next && !late_loader) {
            stage(cur ^ 1, kt + 1);                   // lands while this step computes
            if (kt + 2 < nk) { sa.prefetch(tid, (kt + 2) * BK); sb.prefetch(tid, (kt + 2) * BK); }
        }
        // wave (wr, wc) owns rows wr*16*IM .. and columns wc*64 .. of the block tile
        const char* tA = smem + cur * STAGE_BYTES + (wr * IM / 4) * TL::BYTES;
        const char* tB = smem + cur * STAGE_BYTES + B_OFF + wc * TL::BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> fa[IM], fb[JN];
#pragma unroll
            for (int i = 0; i < IM; ++i) {
                const char* ta = tA + (i / 4) * TL::BYTES;               // 4 blocks of 16 rows per Tile64
                fa[i] = SrcA::KMAJOR ? lds_row_frag<T>(ta, 16 * (i % 4) + li, 32 * ks, g) : lds_col_frag_nat<T>(ta, 32 * ks, 16 * (i % 4), lane);
            }
#pragma unroll
            for (int j = 0; j < JN; ++j)
                fb[j] = SrcB::KMAJOR ? lds_row_frag<T>(tB, 16 * j + li, 32 * ks, g) : lds_col_frag_nat<T>(tB, 32 * ks, 16 * j, lane);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) mma(acc[i][j], fb[j], fa[i]);   // rows = n, cols = m
            if (ks == 0 && has_next && late_loader) {
                __builtin_amdgcn_sched_barrier(0);
                stage(cur ^ 1, kt + 1);
                if (kt + 2 < nk) { sa.prefetch(tid, (kt + 2) * BK); sb.prefetch(tid, (kt + 2) * BK); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next stage has landed ...
        __syncthreads();                                      // ... for every wave, and this one is free again
        cur ^= 1;
    }

    // epilogue: stage 64 x 64 fp32 pieces of the wave's result in LDS (the operand tiles are dead after
    // the last barrier of the loop) and write whole row segments
    float* stg = reinterpret_cast<float*>(smem) + wave * (64 * STG_LD);
    // (one 64-row piece per call with a compile-time piece index: a `#pragma unroll` loop over the pieces is left rolled once
    //  the row writer is large, and acc[4 h + i] with a run-time h puts the accumulators into scratch)
    auto piece = [&](auto h_c) {
        constexpr int h = decltype(h_c)::value;
        if (h) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)      // acc[4h+i][j]: row (m) = 16 i + li, columns (n) = 16 j + 4 g + r
                *reinterpret_cast<f32x4*>(stg + (16 * i + li) * STG_LD + 16 * j + 4 * g) = acc[4 * h + i][j];
        __syncthreads();
        RowWriter<TC>::run(stg, ep, m0 + wr * (16 * IM) + 64 * h, n0 + 64 * wc, M, N, lane);
    };
    static_assert(IM == 4 || IM == 8, "one or two 64-row pieces per wave");
    piece(std::integral_constant<int, 0>{});
    if constexpr (IM == 8) piece(std::integral_constant<int, 1>{});
}


// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves, four phases per K-tile ("8-phase" schedule of the CDNA4 guide):
// both operands contraction-contiguous (forward GEMMs, conv forward, and dgrad against a transposed
// weight copy), bf16, K % 64 == 0.
//   * the K-tile lives in LDS as four half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255), two
//     K-tiles deep (128 KiB); a wave owns 64 rows of EACH A half and 32 columns of EACH B half, so the
//     four phases are the (A-half, B-half) quadrants of its 128 x 64 result: 16 MFMAs each;
//   * every phase issues the LDS reads of its quadrant and ONE half-tile of direct-to-LDS prefetch
//     (2 wave instructions), then barrier / MFMAs / barrier.  The two wave groups (wr = 0 / 1, the two
//     waves of each SIMD) run one barrier apart: while one is in its MFMA block the other issues its LDS
//     reads and loads;
//   * the prefetch is 7 phases ahead of its first use and is only waited for once per K-tile with a
//     counted vmcnt (three half-tiles stay in flight across the wait); raw s_barrier, never __syncthreads
//     (which would drain the loads).
// Buffer hazards (X = buffer of the K-tile being computed, Y = the other one), per K-tile t:
//   phase  LDS reads (X)     prefetch                      restage distance after the last read
//   1      B0 (4), A0 (8)    A1 of t+1 -> Y                A1(Y) was read in phase 3 of t-1: 2 phases
//   2      B1 (4)            B0 of t+2 -> X                1 phase: phase 1 retires its B reads (lgkmcnt) BEFORE its barrier
//   3      A1 (8)            A0 of t+2 -> X                2 phases
//   4      -                 B1 of t+2 -> X, vmcnt(6)      2 phases; the wait retires everything of tile t+1
// ---------------------------------------------------------------------------------------------
// tile walk inside an XCD's range: the smaller grid dimension fastest (its tiles share the other operand's panel,
// which then comes out of the XCD's L2 instead of the Infinity Cache); encoded in the sign of grid_m
static int tile_order(int64_t gm, int64_t gn) {
    const int mode = (int)tuning(TUNE_TILE_ORDER);   // 1: N fastest, 2: M fastest
    const bool m_fast = mode == 2 || (mode == 0 && gm < gn);
    return m_fast ? -(int)gm : (int)gm;
}

// Transposing LDS read of a contraction-strided tile as inline assembly: through the builtin the compiler
// cannot tell the read from the in-flight direct-to-LDS writes and drains them (s_waitcnt vmcnt(0)) before
// every group of reads, which serialises the prefetch.  The result is only valid after the kernel's own
// s_waitcnt lgkmcnt + sched_barrier.  The per-lane part of the address (row 8g+q, the XOR-swizzled column
// chunk) is loop-invariant and kept in a register; K sub-step, row pair and half-tile go into the immediate.
template <int OFF> __device__ __forceinline__ s16x4 tr_read(uint32_t addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// fragment (natural slot order, as lds_col_frag_nat) of columns col0.. of a Tile64 at byte offset TOFF from `base`,
// K sub-step KS; `base` = LDS address of the lane's element (row 8g+q, column col0 + 4p) of the first tile
template <int TOFF, int KS> __device__ __forceinline__ Frag<bf16_t> tr_frag(uint32_t base) {
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = tr_read<TOFF + KS * 32 * 128>(base);
    u.s[1] = tr_read<TOFF + KS * 32 * 128 + 4 * 128>(base);
    Frag<bf16_t> f;
    f.v = u.v;
    return f;
}
__device__ __forceinline__ uint32_t tr_lane_base(const char* tile, int col0, int lane) {
    typedef __attribute__((address_space(3))) const char* lds_cptr;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    return (uint32_t)(uintptr_t)(lds_cptr)tile + Tile64<bf16_t>::elem_off_ks(8 * g + q, col0 + 4 * p);
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 x 64 tiles for SMALL problems (at most ~one workgroup per CU: the decoder / language / hyper-graph
// projections, 72-320 tiles, 12-48 K-steps), built like the 8-phase kernel in miniature.
// With one workgroup of four waves per CU the two-stage loop above runs ONE wave per SIMD: the issue of a K-step's
// eight direct-to-LDS loads (~60 cycles each), its LDS reads and its 32 MFMAs are strictly serial
// (measured 0.77 us per K-step against 0.21 us of MFMA work).  Here the tile is split over EIGHT waves (64 x 32
// each: 16 MFMAs, 12 LDS reads, 4 loads per K-step) in two groups that run one barrier apart - the two waves of a
// SIMD alternate between their load phase and their MFMA phase - over a four-buffer ring: a K-step stages K-tile
// t+2 and retires K-tile t+1 with a counted vmcnt (one tile stays in flight across the raw barriers).
//   step t of a group:  read tile t (12 ds_read) | stage tile t+2 -> buffer (t+2)%4 | vmcnt(4) | barrier |
//                       lgkmcnt(0) | 16 MFMA | barrier
//   RAW: a wave's share of tile t+1 is retired before the barrier that ends its load phase; both groups have passed
//        such a barrier before either reads tile t+1.  WAR: buffer (t+2)%4 was last read two steps earlier.
// bf16, both operands contraction-contiguous, K % 64 == 0, no split-K.  128 KiB of LDS.
// ---------------------------------------------------------------------------------------------
template <typename TC, typename SrcA, typename SrcB>
__global__ __launch_bounds__(512) void gemm4_kernel(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K,
                                                    int grid_m) {
    static_assert(SrcA::KMAJOR && !SrcA::DYN && !SrcB::DYN, "A contraction-contiguous; B either layout (forward / input gradient)");
    using T = bf16_t;
    using TL = Tile64<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 512, NCHK = 2, NS = 4;                  // 2 chunks per thread and operand per K-tile
    constexpr bool BKM = SrcB::KMAJOR;
    constexpr int STAGE_BYTES = 4 * TL::BYTES, B_OFF = 2 * TL::BYTES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int grp = wave_u >> 2;                                  // waves w and w+4 share a SIMD: one from each group
    const int wq = wave_u & 3, wr = wq >> 1;                      // group-local wave: rows wr*64.., columns (2*(wq&1)+grp)*32..
    const int wc32 = 2 * (wq & 1) + grp;
    const int64_t n_tiles = (int64_t)gridDim.x, gm_t = grid_m < 0 ? -grid_m : grid_m, gn_t = n_tiles / gm_t;
    const int64_t xq = n_tiles / 8, xr = n_tiles % 8, xcd = blockIdx.x % 8;
    const int64_t tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + blockIdx.x / 8;
    const int64_t bm = grid_m < 0 ? tile % gm_t : tile / gn_t, bn = grid_m < 0 ? tile / gm_t : tile % gn_t;
    const int64_t m0 = bm * 128, n0 = bn * 128;
    sa.r0 = m0;
    sb.r0 = n0;
    uint32_t offa[NCHK], offb[NCHK];
#pragma unroll
    for (int i = 0; i < NCHK; ++i) {
        offa[i] = sa.lane_off(tid, i, NTHR);
        offb[i] = sb.lane_off(tid, i, NTHR);
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int64_t nk = K / BK;
    auto stage = [&](int buf, int64_t kt) {                     // 4 wave instructions per thread
        if (kt >= nk) return;
        char* base = smem + buf * STAGE_BYTES + 64 * wave_u * 16;
        const char* ka = sa.k_base(0, kt * BK);
        const char* kb = sb.k_base(0, kt * BK);
#pragma unroll
        for (int i = 0; i < NCHK; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr)(ka + offa[i]), (lds_ptr)(base + NTHR * i * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(kb + offb[i]), (lds_ptr)(base + B_OFF + NTHR * i * 16), 16, 0, 0);
        }
    };
    stage(0, 0);
    stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                  // second group runs one barrier behind

    Frag<T> fa[4][2], fb[2][2];
    uint32_t trb[2] = {0, 0};                                     // contraction-strided B: per-lane LDS addresses in buffer 0
    if constexpr (!BKM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) trb[j] = tr_lane_base(smem + B_OFF + (wc32 >> 1) * TL::BYTES, (wc32 & 1) * 32 + 16 * j, lane);
    }
    int cur = 0;
    for (int64_t kt = 0; kt < nk; ++kt) {
        const char* tA = smem + cur * STAGE_BYTES + wr * TL::BYTES;
        const char* tB = smem + cur * STAGE_BYTES + B_OFF + (wc32 >> 1) * TL::BYTES;
        const int brow = (wc32 & 1) * 32 + li;
        if constexpr (BKM) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[j][ks] = lds_row_frag<T>(tB, brow + 16 * j, 32 * ks, g);
        } else {
            const uint32_t xo = (uint32_t)cur * STAGE_BYTES;
            fb[0][0] = tr_frag<0, 0>(trb[0] + xo); fb[0][1] = tr_frag<0, 1>(trb[0] + xo);
            fb[1][0] = tr_frag<0, 0>(trb[1] + xo); fb[1][1] = tr_frag<0, 1>(trb[1] + xo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = lds_row_frag<T>(tA, 16 * i + li, 32 * ks, g);
        stage((cur + 2) & (NS - 1), kt + 2);
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile kt+1 has landed (this wave's share)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma(acc[i][j], fb[j][ks], fa[i][ks]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        cur = (cur + 1) & (NS - 1);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                  // re-align the groups
    __syncthreads();
    // epilogue: the two waves that share 64 columns (same wr, wc32 = 2c and 2c+1) fill one 64 x 64 staging piece;
    // the even one writes it out
    const int pair = wr * 2 + (wc32 >> 1);
    float* stg = reinterpret_cast<float*>(smem) + pair * (64 * STG_LD);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<f32x4*>(stg + (16 * i + li) * STG_LD + 32 * (wc32 & 1) + 16 * j + 4 * g) = acc[i][j];
    __syncthreads();
    if ((wc32 & 1) == 0) RowWriter<TC>::run(stg, ep, m0 + wr * 64, n0 + 64 * (wc32 >> 1), M, N, lane);
}

template <typename TC, typename SrcA, typename SrcB>
static int launch4(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, hipStream_t st, const char* what) {
    const int64_t gm = (M + 127) / 128, gn = (N + 127) / 128;
    const size_t lds = 4 * 4 * Tile64<bf16_t>::BYTES;
    auto kern = gemm4_kernel<TC, SrcA, SrcB>;
    static std::atomic<uint64_t> raised{0};          // per instantiation, one bit per device
    raise_lds_limit(raised, reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(gm * gn)), dim3(512), lds, st, sa, sb, ep, M, N, K, tile_order(gm, gn));
    return check_launch(what);
}

// small bf16 NT problems (at most `max_tiles` 128 x 128 tiles: about one workgroup per CU) with whole K-steps
static bool use_gemm4(int64_t M, int64_t N, int64_t K, int64_t bytes_a, int64_t bytes_b) {
    const int64_t max_tiles = tuning(TUNE_GEMM4_MAX_TILES);
    if (K % BK || K < 2 * BK) return false;
    if (bytes_a >= ((int64_t)1 << 32) || bytes_b >= ((int64_t)1 << 32)) return false;
    return ((M + 127) / 128) * ((N + 127) / 128) <= max_tiles;
}

// Stream-K work split of the 8-phase kernel (template parameter SK; the conv forward, whose 147 / 222 tiles would leave 43 / 13 %
// of the CUs idle).  The grid is 8 x 32 workgroups, blockIdx % 8 = the XCD the dispatcher puts a workgroup on.  Each XCD owns a
// contiguous run of R (16 <= R < 32) output tiles - the same run as in the one-tile-per-workgroup launch - and splits every tile's
// K-tiles ("iterations") at the same point h (an even share would be ceil(R nk / 32); see sigma below):
//   * R HEAD workgroups compute K-tiles [0, h) of one tile each.  They walk K in lockstep like the classic launch, so the tiles
//     that share an operand panel still read it at the same time and find it in the XCD's L2 (cutting the concatenated
//     (tile, K) space into 32 contiguous ranges instead balances just as well, but every workgroup then sits at a different K
//     offset of a different tile: measured 2x the L2-miss traffic);
//   * 32 - R TAIL workgroups share the remaining [h, nk) of all R tiles, in tile order, as equal contiguous ranges: a tile's
//     tail is computed by one tail workgroup or cut between two.  They publish raw fp32 partial tiles (slot 2 tile + part)
//     and set a flag.
// The head workgroup owns the tile: it adds the one or two published parts to its accumulators and runs the epilogue.  Tail
// workgroups have the LOWER block indices of their XCD: they are dispatched first and never wait, the heads only wait for
// them - no deadlock whatever else shares the chip.  The owner clears the flags, so a launch leaves them zero.
struct StreamK {
    float* ws;
    int* flags;
    int n_tiles;
    int sigma;               // cost of a tail K-tile in percent of a head K-tile (tails read their operand panels alone)
    const void* wt;          // weighted plan (tiles of different length): StreamKW in kernel-argument memory, or null
};
constexpr int STREAMK_WGS = 256;
constexpr int STREAMK_SLOTS = 768;                          // 3 per tile (the uniform plan uses 2), at most 8 x 31 tiles
constexpr size_t STREAMK_SLOT = (size_t)256 * 256;          // floats per partial tile

struct StreamKSeg {
    int tile, kb, nk, owner, slot, parts;                    // nk == 0: no such segment; slot: where a tail segment publishes;
};                                                           // parts: how many published parts the owner adds
// segment `seg` of this workgroup (32-bit scalar arithmetic, recomputed where needed instead of kept in registers: the kernel
// has no register to spare across its main loop)
__host__ __device__ __forceinline__ StreamKSeg streamk_plan(int n_tiles, int nk_all, int block, int grid, int seg, int sigma) {
    StreamKSeg d{0, 0, 0, 0, 0, 0};
    const int xq = n_tiles / 8, xr = n_tiles % 8, xcd = block & 7, j = block >> 3;
    const int W = grid >> 3;
    const int tile0 = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int R = xq + (xcd < xr ? 1 : 0), T = W - R;          // tiles of this XCD = head workgroups; tail workgroups
    // head / tail length of every tile: heads take h K-tiles, the T tail workgroups R (nk - h) / T each at sigma % of the
    // heads' speed; equal finishing times give h = sigma R nk / (100 T + sigma R)   (sigma = 100: h = R nk / W)
    const int h = min(nk_all, (sigma * R * nk_all + 100 * T + sigma * R - 1) / (100 * T + sigma * R)), tl = nk_all - h;
    const int pt = T > 0 ? (R * tl + T - 1) / T : 0;             // tail iterations per tail workgroup
    if (j >= T) {                                              // head of tile j - T
        if (seg > 0) return d;
        const int lt = j - T;
        d.tile = tile0 + lt; d.kb = 0; d.nk = h; d.owner = 1;
        d.slot = 2 * d.tile;                                   // (an owner's slot: the first of its parts')
        d.parts = tl > 0 ? ((lt + 1) * tl - 1) / pt - (lt * tl) / pt + 1 : 0;
        return d;
    }
    const int lo = j * pt, hi = min(lo + pt, R * tl);
    const int start = seg == 0 ? lo : (lo / tl + seg) * tl;
    if (tl == 0 || start >= hi) return d;
    const int lt = start / tl, end = min(hi, (lt + 1) * tl);
    d.tile = tile0 + lt; d.kb = h + (start - lt * tl); d.nk = end - start; d.owner = 0;
    d.slot = 2 * d.tile + (start > lt * tl ? 1 : 0);
    return d;
}
// WEIGHTED plan: the tiles of a launch differ in length (the conv forward in position-major row order drops the taps that read the
// zero border for every row of a tile: 20, 30, 40 or 45 of 45).  Same roles as the uniform plan - heads own a tile and compute its
// first K-tiles in lockstep, the tail workgroups of the XCD share what is left, in tile order, as equal contiguous ranges - with
// per-tile numbers:
//   XCD x owns the run of tiles [t0_x, t0_x + R_x): still contiguous (neighbouring row blocks share input lines), but cut so that
//     the K-TILES, not the tiles, divide evenly - an XCD of interior positions gets fewer tiles and more tail workgroups
//     (16 <= R_x <= 31);
//   head of tile t: [0, min(nk_t, h_x)), h_x the smallest h with 100 h T >= sigma Rem(h), Rem(h) = sum max(0, nk_t - h);
//   tail j: K-tiles [j pt, (j + 1) pt) of the concatenated remainders, pt = ceil(Rem / T); a remainder r_t <= 2 pt (checked when
//     the tables are built) is cut between at most three tails: slots 3 tile + (j - P_t / pt).
// (Measured and dropped: equal ranges of the concatenated K-tiles for all 32 workgroups of an XCD, tiles dealt to the XCDs by
//  weight - a perfect split on paper, 2.51 against 2.08 ms for conv1: the workgroups no longer walk K together and the tiles of an
//  XCD no longer neighbour each other, and every operand panel is read alone.)
// Tables (kernel-argument memory, scalar loads): per tile nk_t and P_t = the remainders in front of it in its XCD; per XCD
// t0 / R / h / pt / Rem and for every tail the tile its range starts in.
struct StreamKW {
    uint16_t nk[256], P[256];
    uint16_t h[8], pt[8], rem[8];
    uint8_t t0[8], R[8];
    uint8_t first[8][16];
};
template <typename WP>
__host__ __device__ __forceinline__ StreamKSeg streamk_plan_w(WP w, int n_tiles, int block, int grid, int seg) {
    StreamKSeg d{0, 0, 0, 0, 0, 0};
    (void)n_tiles;
    const int xcd = block & 7, j = block >> 3;
    const int tile0 = w->t0[xcd], R = w->R[xcd], T = (grid >> 3) - R;
    const int h = w->h[xcd], pt = w->pt[xcd], rem = w->rem[xcd];
    if (j >= T) {                                              // head of tile j - T
        if (seg > 0) return d;
        const int t = tile0 + j - T, nk = w->nk[t], P = w->P[t], ht = nk < h ? nk : h, r = nk - ht;
        d.tile = t; d.kb = 0; d.nk = ht; d.owner = 1; d.slot = 3 * t;
        d.parts = r > 0 ? (P + r - 1) / pt - P / pt + 1 : 0;
        return d;
    }
    const int lo = j * pt, hi = lo + pt < rem ? lo + pt : rem;
    if (pt == 0 || lo >= hi) return d;
    int cnt = 0;
    for (int u = w->first[xcd][j]; u < R; ++u) {               // tiles with a remainder that meets [lo, hi), in order
        const int t = tile0 + u, nk = w->nk[t], P = w->P[t], ht = nk < h ? nk : h, r = nk - ht;
        if (r == 0) continue;
        if (P >= hi) break;
        if (P + r <= lo) continue;
        if (cnt++ < seg) continue;
        const int start = P > lo ? P : lo, end = P + r < hi ? P + r : hi;
        d.tile = t; d.kb = ht + (start - P); d.nk = end - start; d.owner = 0;
        d.slot = 3 * t + (j - P / pt);
        return d;
    }
    return d;
}
// host: the tables for n_tiles tiles of nk[t] K-tiles; false = this launch cannot use the weighted plan
static bool streamk_w_build(StreamKW& w, int n_tiles, const uint16_t* nk, int sigma) {
    if (n_tiles < 128 || n_tiles >= 256) return false;
    const int Wg = STREAMK_WGS / 8;
    int nkmax_all = 0;
    for (int t = 0; t < 256; ++t) { w.nk[t] = t < n_tiles ? nk[t] : 0; w.P[t] = 0; }
    for (int t = 0; t < n_tiles; ++t) { nkmax_all = std::max<int>(nkmax_all, nk[t]); if (nk[t] < 1) return false; }
    // head length of a run of tiles [a, a + R): the smallest h with 100 h T >= sigma Rem(h)
    auto head_len = [&](int a, int R) -> int {
        const int T = Wg - R;
        int lo = 1, hi = nkmax_all;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            int64_t rem = 0;
            for (int u = 0; u < R; ++u) rem += std::max(0, (int)nk[a + u] - mid);
            if ((int64_t)100 * mid * T >= (int64_t)sigma * rem) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    // contiguous runs of 16 .. 31 tiles per XCD that minimise the LONGEST head (dynamic programme over the cut points: the launch
    // takes as long as its slowest XCD; equal K-tiles per XCD is not it - an XCD's head length also depends on how many tail
    // workgroups its tile count leaves)
    {
        const int INF = 1 << 30;
        std::vector<int> best((size_t)9 * (n_tiles + 1), INF), from((size_t)9 * (n_tiles + 1), -1);
        best[0] = 0;
        for (int x = 0; x < 8; ++x)
            for (int t = 0; t <= n_tiles; ++t) {
                const int cur = best[(size_t)x * (n_tiles + 1) + t];
                if (cur == INF) continue;
                for (int R = 16; R <= 31 && t + R <= n_tiles; ++R) {
                    const int left = n_tiles - t - R;
                    if (left < 16 * (7 - x) || left > 31 * (7 - x)) continue;
                    const int v = std::max(cur, head_len(t, R));
                    int& slot_ = best[(size_t)(x + 1) * (n_tiles + 1) + t + R];
                    if (v < slot_) { slot_ = v; from[(size_t)(x + 1) * (n_tiles + 1) + t + R] = t; }
                }
            }
        if (best[(size_t)8 * (n_tiles + 1) + n_tiles] == INF) return false;
        int t = n_tiles;
        for (int x = 8; x > 0; --x) {
            const int f = from[(size_t)x * (n_tiles + 1) + t];
            w.t0[x - 1] = (uint8_t)f;
            w.R[x - 1] = (uint8_t)(t - f);
            t = f;
        }
    }
    for (int xcd = 0; xcd < 8; ++xcd) {
        const int tile0 = w.t0[xcd];
        const int R = w.R[xcd], T = Wg - R;
        if (R < 16 || T < 1 || T > 16) return false;
        int nkmax = 0;
        for (int u = 0; u < R; ++u) nkmax = std::max<int>(nkmax, nk[tile0 + u]);
        auto rem_of = [&](int h) { int64_t r = 0; for (int u = 0; u < R; ++u) r += std::max(0, (int)nk[tile0 + u] - h); return r; };
        int h = nkmax;
        for (int c = 0; c <= nkmax; ++c)
            if ((int64_t)100 * c * T >= (int64_t)sigma * rem_of(c)) { h = c; break; }
        const int64_t rem = rem_of(h);
        if (h < 1 || rem > 65535) return false;
        const int pt = rem > 0 ? (int)((rem + T - 1) / T) : 0;
        w.h[xcd] = (uint16_t)h; w.pt[xcd] = (uint16_t)pt; w.rem[xcd] = (uint16_t)rem;
        int64_t P = 0;
        for (int u = 0; u < R; ++u) {
            const int r = std::max(0, (int)nk[tile0 + u] - h);
            w.P[tile0 + u] = (uint16_t)P;
            if (r > 0 && (P + r - 1) / pt - P / pt > 2) return false;      // more than three parts
            P += r;
        }
        for (int j = 0; j < 16; ++j) {
            w.first[xcd][j] = 0;
            if (j >= T || pt == 0) continue;
            const int64_t lo = (int64_t)j * pt;
            for (int u = 0; u < R; ++u) {
                const int r = std::max(0, (int)nk[tile0 + u] - h);
                if (r > 0 && w.P[tile0 + u] + r > lo) { w.first[xcd][j] = (uint8_t)u; break; }
            }
        }
    }
    return true;
}
template <bool WEIGHTED>
__device__ __forceinline__ StreamKSeg streamk_segment(const StreamK& sk, int nk_all, int seg) {
    if (WEIGHTED && sk.wt) {
        typedef const __attribute__((address_space(4))) StreamKW* wptr;
        return streamk_plan_w((wptr)sk.wt, sk.n_tiles, (int)blockIdx.x, (int)gridDim.x, seg);
    }
    return streamk_plan(sk.n_tiles, nk_all, (int)blockIdx.x, (int)gridDim.x, seg, sk.sigma);
}

// (the body is a device function so that the plain launch and the grouped launch - many small problems, one grid - share it:
//  bx / by = this workgroup's tile slot and split-K part inside ITS problem, gx / gy = that problem's tile and split counts)
// SK: ONE stream-K segment (`seg` of this workgroup) per call; returns false when the workgroup has no such segment.  The
// segment loop lives in gemm8_sk_kernel, which re-reads the kernel arguments from the kernarg segment for every segment:
// with the loop in here every argument (three operand / epilogue structs) stayed live across the whole pipelined main loop
// and 95 SGPRs + 48 VGPRs were spilled (196 B of scratch per lane).
// TM: rows of the tile, 256 or 192.  The 192-row form (K-major A only) exists for the 12 576-row problems with 768 / 1 536
// output columns: 150 / 300 tiles of 256 x 256 fill 59 % of the 256 CUs in their (last) round, 198 / 396 tiles of 192 x 256 fill
// 77 % at three quarters of the work per tile.  The A operand is staged row-linearly (64 rows per direct-to-LDS instruction of the
// workgroup), so a 96-row half is one and a half instructions: the middle instruction's rows 64-95 belong to half 0 and are
// issued by waves 0-3, rows 96-127 to half 1 by waves 4-7 - which is why the counted vmcnt waits differ between the two wave
// groups (6 outstanding loads for waves 0-3, 5 for waves 4-7, see below).  A wave owns 48 rows (three 16-row blocks) of each half.
template <typename TC, typename SrcA, typename SrcB, bool SK = false, int TM = 256>
__device__ __forceinline__ bool gemm8_body(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, int grid_m, StreamK sk,
                                           const int bx, const int by, const int gx, const int gy, const int seg = 0) {
    static_assert(!SrcA::DYN, "only the B operand may gather per K-tile");
    static_assert(TM == 256 || (TM == 192 && SrcA::KMAJOR && !SK), "the 192-row tile needs a K-major A operand");
    constexpr int NI = TM / 64;                      // 16-row blocks per wave and A half
    constexpr int HALF = TM / 2, WROWS = TM / 4;     // rows per A half / per wave and half
    using T = bf16_t;
    using TL = Tile64<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 512;
    constexpr bool AK = SrcA::KMAJOR, BKM = SrcB::KMAJOR;
    int tid_ = threadIdx.x;
    if constexpr (SK) asm volatile("" : "+v"(tid_));   // (per segment: nothing derived from the thread id is carried from one segment into the next)
    const int tid = tid_, wave = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave_u >> 2, wc = wave_u & 3;
    // K % 64 != 0 is supported for the form with BOTH operands contraction-strided plain matrices (weight gradients over a
    // row count like 12 576 = 196 x 64 + 32): the loads of the last K-tile's rows past the end are pointed at 16 zero bytes
    constexpr bool RAGGED_OK = !AK && !BKM && !SrcA::DYN && !SrcB::DYN;
    const int64_t nk_all = RAGGED_OK ? (K + BK - 1) / BK : K / BK;
    const int tail = RAGGED_OK ? (int)(K % BK) : 0;
    // grid_m < 0: walk M fastest instead (few row tiles, many column tiles - the weight-gradient shapes - so
    // that the tiles sharing a B column-panel sit next to each other)
    const int64_t n_tiles = SK ? (int64_t)sk.n_tiles : (int64_t)gx, gm_t = grid_m < 0 ? -grid_m : grid_m, gn_t = n_tiles / gm_t;

    {
    int64_t tile, kb, nk;
    if constexpr (SK) {
        int s_ = seg;
        asm volatile("" : "+s"(s_));
        const StreamKSeg d = streamk_segment<SrcA::SKIP>(sk, (int)nk_all, s_);
        if (d.nk == 0) return false;
        tile = d.tile; kb = d.kb; nk = d.nk;
        if constexpr (SrcA::SKIP) {                  // (the tile's tap list; its K-tile count is what the weighted plan was built from)
            sa.set_tile((grid_m < 0 ? tile % gm_t : tile / gn_t) * TM);
        }
    } else {
        const int64_t xq = n_tiles / 8, xr = n_tiles % 8, xcd = bx % 8;
        tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + bx / 8;
        int64_t nk_tile = nk_all;
        if constexpr (SrcA::SKIP) {                  // conv forward in position-major row order: the tile's taps
            nk_tile = sa.set_tile((grid_m < 0 ? tile % gm_t : tile / gn_t) * TM);
        }
        if constexpr (SrcB::NPERM) {                 // conv weight gradient in position-major row order: only the K-tiles its tap stays inside for
            if (sb.tpp) {
                const int64_t bn_ = grid_m < 0 ? tile / gm_t : tile % gn_t;
                nk_tile = sb.set_tap((int)((uint32_t)sb.col_of_block(bn_) / (uint32_t)sb.g.Cin));
            }
        }
        // K range of this workgroup (gridDim.y > 1: split-K, partial sums added with atomics by the epilogue)
        const int64_t per_split = (nk_tile + gy - 1) / gy;
        kb = (int64_t)by * per_split;
        nk = min(nk_tile, kb + per_split) - kb;              // K-tiles of this workgroup, numbered 0 .. nk-1 below
        if (nk <= 0) return false;
    }
    const int64_t bm = grid_m < 0 ? tile % gm_t : tile / gn_t, bn = grid_m < 0 ? tile / gm_t : tile % gn_t;
    const int64_t m0 = bm * TM;
    int64_t n0 = bn * 256;
    if constexpr (SrcB::NPERM) {
        if (sb.nblk || sb.lblk0) { n0 = sb.col_of_block(bn); N = sb.R; }     // (the launch's N only counted its column blocks)
        if (sb.zero_base && by == 0) {                           // 1 KB row segments, two per pass of the workgroup
            const int64_t segs = (int64_t)sb.zero_rows * sb.zero_blks;
            for (int64_t q = (int64_t)bx * 2 + (tid >> 8); q < segs; q += (int64_t)gx * 2) {
                const int64_t blk = q / sb.zero_rows, row = q - blk * sb.zero_rows;
                sb.zero_base[row * sb.R + sb.col_of_lblock(sb.zero_blk0 + blk) + (tid & 255)] = 0.f;
            }
        }
    }
    sa.r0 = m0;
    sb.r0 = n0;

    uint32_t offa[4], offb[4];                       // chunk i = Tile64 i of the operand (rows / columns 64 i ..)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offa[i] = sa.lane_off(tid, i, NTHR);
        offb[i] = sb.lane_off(tid, i, NTHR);
    }
    // gathered B operand (conv weight gradient): one position per thread and K-tile, fetched three K-tiles ahead
    // with a load the compiler does not track (it would drain the direct-to-LDS prefetch at the first use) and
    // retired by the kernel's own counted wait
    // K-tile number (of this tile's list) -> offset along the contraction index: the convolution sources may permute (forward /
    // input gradient: channel-block-major) or skip (weight gradient: zero-border positions of the tile's tap)
    auto kof = [&](int64_t kt_abs) -> int64_t {
        if constexpr (SrcB::NPERM) return sb.k_of_tile(kt_abs);
        else return sa.k_of_tile(kt_abs);
    };
    int32_t dpos[3] = {0, 0, 0}, dnext = 0;
    uint32_t dyn_cur = 0;
    if constexpr (SrcB::DYN) {
#pragma unroll
        for (int u = 0; u < 3; ++u) dpos[u] = *sb.dyn_ptr(tid, kof(kb + u));
    }

    f32x4 acc[2][2][NI][2];                          // [A half][B half][16-row block][16-col block]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int STAGE_BYTES = 8 * TL::BYTES;       // A: Tile64 0..3, B: Tile64 4..7
    constexpr int B_OFF = 4 * TL::BYTES;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    // half-tile h of operand A / B of K-tile kt -> buffer buf (no-op past the end of the K range)
    uint32_t bad_a = 0, bad_b = 0;                  // bit i: chunk i of this thread is a K row past the end in the ragged last K-tile
    if constexpr (RAGGED_OK) {
        if (tail) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bad_a |= (sa.k_row(tid, i, NTHR) >= tail ? 1u : 0u) << i;
                bad_b |= (sb.k_row(tid, i, NTHR) >= tail ? 1u : 0u) << i;
            }
        }
    }
    auto stage_a = [&](int buf, int h, int64_t kt) {
        if (kt >= nk) return;
        char* base = smem + buf * STAGE_BYTES + 64 * wave_u * 16;
        if constexpr (TM == 192) {
            // rows 0-63: instruction 0, rows 64-127: instruction 1 (waves 0-3 hold its rows 64-95, waves 4-7 rows 96-127),
            // rows 128-191: instruction 2; half 0 = rows 0-95, half 1 = rows 96-191
            const int whole = h ? 2 : 0;
            __builtin_amdgcn_global_load_lds((glb_ptr)(sa.k_base(whole, kof(kb + kt)) + offa[whole]), (lds_ptr)(base + NTHR * whole * 16), 16, 0, 0);
            if ((wave_u >= 4) == (h == 1))
                __builtin_amdgcn_global_load_lds((glb_ptr)(sa.k_base(1, kof(kb + kt)) + offa[1]), (lds_ptr)(base + NTHR * 1 * 16), 16, 0, 0);
            return;
        }
        const bool rag = RAGGED_OK && tail && (kb + kt) == nk_all - 1;          // wave-uniform
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * h + ii;
            const char* src = sa.k_base(i, kof(kb + kt)) + offa[i];
            if (rag && ((bad_a >> i) & 1u)) src = reinterpret_cast<const char*>(g_zero16);
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(base + NTHR * i * 16), 16, 0, 0);
        }
    };
    auto stage_b = [&](int buf, int h, int64_t kt, uint32_t dyn) {
        if (kt >= nk) return;
        char* base = smem + buf * STAGE_BYTES + B_OFF + 64 * wave_u * 16;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * h + ii;
            const uint32_t off = SrcB::DYN ? dyn : offb[i];
            const char* src = sb.k_base(i, kof(kb + kt)) + off;
            if (RAGGED_OK && tail && (kb + kt) == nk_all - 1 && ((bad_b >> i) & 1u)) src = reinterpret_cast<const char*>(g_zero16);
            __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(base + NTHR * i * 16), 16, 0, 0);
        }
    };

    // prologue: K-tile 0 completely, K-tile 1 without its A1 (phase 1 issues that)
    {
        uint32_t d0 = 0, d1 = 0;
        if constexpr (SrcB::DYN) { d0 = sb.dyn_off(tid, dpos[0]); d1 = sb.dyn_off(tid, dpos[1]); dyn_cur = sb.dyn_off(tid, dpos[2]); }
        stage_b(0, 0, 0, d0); stage_a(0, 0, 0); stage_b(0, 1, 0, d0); stage_a(0, 1, 0);
        stage_b(1, 0, 1, d1); stage_a(1, 0, 1); stage_b(1, 1, 1, d1);
    }
    // three half-tile stages may stay in flight: 2 + 2 + 2 instructions - with the 192-row tile an A half is two instructions for
    // one wave group and one for the other (stage_a), i.e. 6 for waves 0-3 (A half 0 of K-tile 1 / t + 2 in flight) and 5 for waves 4-7
    if (nk > 1) {
        if (TM == 192 && wave_u >= 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // second wave group runs one barrier behind

    Frag<T> fa[NI][2], fb0[2][2], fb1[2][2];
    // contraction-strided operands: per-lane LDS addresses of the wave's column blocks in buffer 0, half 0
    uint32_t tra[4] = {0, 0, 0, 0}, trb[2] = {0, 0};
    if constexpr (!AK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tra[i] = tr_lane_base(smem + wr * TL::BYTES, 16 * i, lane);
    }
    if constexpr (!BKM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) trb[j] = tr_lane_base(smem + B_OFF + (wc >> 1) * TL::BYTES, (wc & 1) * 32 + 16 * j, lane);
    }
#define SHG_G8_RA(H)                                                                                              \
    if constexpr (AK) {                                                                                           \
        const char* t_ = X + ((H) * HALF + wr * WROWS) * TL::ROWB;     /* (row-linear: 64 rows per Tile64) */     \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                            \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) fa[i][ks] = lds_row_frag<T>(t_, 16 * i + li, 32 * ks, g); \
    } else {                                                                                                      \
        fa[0][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(tra[0] + xoff); fa[0][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(tra[0] + xoff); \
        fa[1][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(tra[1] + xoff); fa[1][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(tra[1] + xoff); \
        fa[2][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(tra[2] + xoff); fa[2][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(tra[2] + xoff); \
        fa[3][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(tra[3] + xoff); fa[3][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(tra[3] + xoff); \
    }
#define SHG_G8_RB(H, FB)                                                                                          \
    if constexpr (BKM) {                                                                                          \
        const char* t_ = X + B_OFF + (2 * (H) + (wc >> 1)) * TL::BYTES;                                           \
        const int c0_ = (wc & 1) * 32;                                                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) FB[j][ks] = lds_row_frag<T>(t_, c0_ + 16 * j + li, 32 * ks, g); \
    } else {                                                                                                      \
        FB[0][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(trb[0] + xoff); FB[0][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(trb[0] + xoff); \
        FB[1][0] = tr_frag<(H) * 2 * TL::BYTES, 0>(trb[1] + xoff); FB[1][1] = tr_frag<(H) * 2 * TL::BYTES, 1>(trb[1] + xoff); \
    }
#define SHG_G8_MMA(AH, BH, FB)                                               \
    __builtin_amdgcn_s_setprio(1);                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                         \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                       \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) mma(acc[AH][BH][i][j], FB[j][ks], fa[i][ks]); \
    __builtin_amdgcn_s_setprio(0);

    int cur = 0;
    for (int64_t kt = 0; kt < nk; ++kt) {
        const char* X = smem + cur * STAGE_BYTES;
        const uint32_t xoff = (uint32_t)cur * STAGE_BYTES;
        // ---- phase 1: quadrant (A0, B0)
        if constexpr (SrcB::DYN) {
            if (kt + 3 < nk)
                asm volatile("global_load_dword %0, %1, off" : "=v"(dnext) : "v"(sb.dyn_ptr(tid, kof(kb + kt + 3))) : "memory");
        }
        SHG_G8_RB(0, fb0)
        __builtin_amdgcn_sched_barrier(0);
        SHG_G8_RA(0)
        stage_a(cur ^ 1, 1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        // the B0 reads (issued first) have returned: B0 may be restaged by the other wave group after the barrier
        if constexpr (AK && NI == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // (the 2 NI reads of A may be outstanding)
        else if constexpr (AK) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);           // register-only MFMAs would otherwise be hoisted above the wait
        SHG_G8_MMA(0, 0, fb0)
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (A0, B1)
        SHG_G8_RB(1, fb1)
        stage_b(cur, 0, kt + 2, dyn_cur);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);           // register-only MFMAs would otherwise be hoisted above the wait
        SHG_G8_MMA(0, 1, fb1)
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (A1, B1)
        SHG_G8_RA(1)
        stage_a(cur, 0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);           // register-only MFMAs would otherwise be hoisted above the wait
        SHG_G8_MMA(1, 1, fb1)
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: quadrant (A1, B0); K-tile t+1 is complete after this wait
        stage_b(cur, 1, kt + 2, dyn_cur);
        if (kt + 2 < nk) {
            if (TM == 192 && wave_u >= 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (SrcB::DYN) {                    // the position fetched in phase 1 has landed (older than the 6)
            asm volatile("" : "+v"(dnext));
            dyn_cur = sb.dyn_off(tid, dnext);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        SHG_G8_MMA(1, 0, fb0)
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
#undef SHG_G8_MMA
#undef SHG_G8_RA
#undef SHG_G8_RB
    if (wr == 0) __builtin_amdgcn_s_barrier();      // re-align the wave groups
    __syncthreads();
    // Everything below addresses memory through these re-derived lane coordinates.  In the stream-K kernel the body sits in a
    // loop over segments: values that only depend on the thread id (staging pointers, slot addresses, row-writer columns) are
    // loop-invariant there, get hoisted in front of the FIRST segment and then live - or rather: are spilled - across every main
    // loop.  Laundering the thread id here pins their computation behind the main loop.
    int tid2 = tid;
    if constexpr (SK) asm volatile("" : "+v"(tid2));
    const int wave2 = tid2 >> 6, lane2 = tid2 & 63, g2 = lane2 >> 4, li2 = lane2 & 15;
    if constexpr (SrcB::NPERM && !SK) {
        if (ep.sumsq) {                              // (whole tiles, plain stores: the accumulators ARE the stored values)
            float ss = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 v = acc[a][b][i][j];
                            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                        }
            const double tot = wave_sum_f64((double)ss);
            if (lane2 == 0) atomicAdd(ep.sumsq, tot);
        }
    }

    if constexpr (SK) {
        int s_ = seg;
        asm volatile("" : "+s"(s_));
        const StreamKSeg d = streamk_segment<SrcA::SKIP>(sk, (int)nk_all, s_);
        if (!d.owner) {
            // tail segment: publish the raw accumulators (register r of thread t at float4 index r * 512 + t)
            f32x4* slot = reinterpret_cast<f32x4*>(sk.ws + (size_t)d.slot * STREAMK_SLOT) + tid2;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) slot[(((a * 2 + b) * 4 + i) * 2 + j) * NTHR] = acc[a][b][i][j];
            __threadfence();
            __syncthreads();
            if (tid2 == 0) __hip_atomic_store(sk.flags + d.slot, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();                             // the accumulators' LDS staging area is the next segment's operand buffer
            return true;
        }
        // the tile's tail was computed by one or two tail workgroups of this XCD (lower block indices): add what they published.
        // ASSUMPTION this wait rests on: the hardware dispatches the workgroups of one launch in increasing block index, and a
        // dispatched workgroup runs to completion without needing any later workgroup (tails never wait).  A head therefore only
        // ever waits for workgroups that were dispatched before it; other kernels sharing the chip can delay the tails but not
        // starve them.  tests/test_host_cpu.py checks on the plan that every publisher has a lower block index than its owner.
        for (int part = 0; part < d.parts; ++part) {
            const int sl = d.slot + part;
            if (tid2 == 0) {
                // (relaxed polls: an ACQUIRE load at agent scope invalidates the XCD's L2 on EVERY poll - under the tails that are
                //  still streaming their operand panels through it; the one acquire fence below orders the reads of the slot)
                // (bounded: ~2 s.  A plan error would otherwise hang the GPU; this way it ends as a wrong result that the tests see)
                for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(sk.flags + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                    __builtin_amdgcn_s_sleep(16);
                __hip_atomic_store(sk.flags + sl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const f32x4* slot = reinterpret_cast<const f32x4*>(sk.ws + (size_t)sl * STREAMK_SLOT) + tid2;
            // (eight vectors at a time with a scheduling fence between the groups: left to itself the scheduler issues all 32
            //  loads first - 128 more live registers beside the 128 accumulators - and everything else the epilogue needs is
            //  spilled around the main loop)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[a][b][i][j] += slot[(((a * 2 + b) * 4 + i) * 2 + j) * NTHR];
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }

    // epilogue: per A half, a wave2 holds 64 rows x (32 + 32) columns, the second 32 columns 128 further right in C.  Two
    // neighbouring waves (wc = 2q, 2q + 1) therefore own the two halves of the same 64-column blocks: they stage both blocks
    // together (one 64 x 64 fp32 piece each) and each writes ONE of them, in whole 128-byte rows of bf16 (256-byte of fp32) -
    // writing the 32-column halves separately made every store a partial L2 line.
    float* stg0 = reinterpret_cast<float*>(smem) + (wave2 & ~1) * (64 * STG_LD);     // the pair's left block (columns 64 q ..)
    float* stg1 = stg0 + 64 * STG_LD;                                               // its right block (columns 128 + 64 q ..)
    const int half = wc & 1;
    float csum_carry[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bias-gradient sums of both A halves (same columns)
    // (the A half is a compile-time constant of the lambda: as a `#pragma unroll` loop the compiler stops unrolling it once the
    //  row writer grows past its size threshold, indexes acc[a] dynamically and moves the whole accumulator array into scratch -
    //  528 bytes per lane2, zero-filled before the main loop)
    auto epilogue_half = [&](auto a_c) {
        constexpr int a = decltype(a_c)::value;
        if (a) __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>((b ? stg1 : stg0) + (16 * i + li2) * STG_LD + 32 * half + 16 * j + 4 * g2) = acc[a][b][i][j];
        __syncthreads();
        RowWriter<TC>::template run<!SK, WROWS>(half ? stg1 : stg0, ep, m0 + HALF * a + WROWS * wr, n0 + 128 * half + 64 * (wc >> 1), M, N,
                                                lane2, 0, csum_carry);
    };
    if constexpr (SK) {                              // (the stream-K body keeps the loop form: its register allocation is at the limit
#pragma unroll                                       //  and this form spills least - 168 against 244 bytes per lane2)
        for (int a = 0; a < 2; ++a) {
            if (a) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        *reinterpret_cast<f32x4*>((b ? stg1 : stg0) + (16 * i + li2) * STG_LD + 32 * half + 16 * j + 4 * g2) = acc[a][b][i][j];
            __syncthreads();
            RowWriter<TC>::template run<false>(half ? stg1 : stg0, ep, m0 + 128 * a + 64 * wr, n0 + 128 * half + 64 * (wc >> 1), M, N, lane2, 0,
                                               csum_carry);
        }
    } else {
        epilogue_half(std::integral_constant<int, 0>{});
        epilogue_half(std::integral_constant<int, 1>{});
    }
    if (ep.csum) RowWriter<TC>::flush_csum(csum_carry, ep, n0 + 128 * half + 64 * (wc >> 1), N, lane2);
    if constexpr (SK) __syncthreads();               // the staging area is the next segment's first operand buffer
    }
    return true;
}

template <typename TC, typename SrcA, typename SrcB, bool SK = false, int TM = 256>
__global__ __launch_bounds__(512) void gemm8_kernel(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K,
                                                    int grid_m, StreamK sk) {
    static_assert(!SK, "stream-K launches go through gemm8_sk_kernel");
    gemm8_body<TC, SrcA, SrcB, false, TM>(sa, sb, ep, M, N, K, grid_m, sk, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}

// Stream-K launch: all arguments in ONE struct, read afresh from the kernarg segment (scalar loads through a pointer the
// compiler cannot see through) at the top of every segment, so that nothing but the segment counter lives across a segment.
template <typename TC, typename SrcA, typename SrcB> struct G8SkCore {
    SrcA sa;
    SrcB sb;
    Epilogue<TC> ep;
    int64_t M, N, K;
    int grid_m;
    StreamK sk;
    int weighted;            // G8SkArgs::wt holds the weighted plan's tables
};
template <typename TC, typename SrcA, typename SrcB> struct G8SkArgs {
    G8SkCore<TC, SrcA, SrcB> core;
    StreamKW wt;             // (never copied into registers: read in place with scalar loads)
};
template <typename TC, typename SrcA, typename SrcB>
__global__ __launch_bounds__(512) void gemm8_sk_kernel(G8SkArgs<TC, SrcA, SrcB> unused_by_name) {
#if defined(__HIP_DEVICE_COMPILE__)                   // (the host pass only needs the symbol: it cannot copy out of address space 4)
    typedef const __attribute__((address_space(4))) G8SkArgs<TC, SrcA, SrcB>* karg_ptr;
#pragma nounroll
    for (int seg = 0; seg < 64; ++seg) {
        karg_ptr p = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        const G8SkCore<TC, SrcA, SrcB> a = p->core;
        StreamK sk = a.sk;
        if (a.weighted) sk.wt = (const void*)&p->wt;            // (stays kernel-argument memory: streamk_segment reads it with scalar loads)
        if (!gemm8_body<TC, SrcA, SrcB, true>(a.sa, a.sb, a.ep, a.M, a.N, a.K, a.grid_m, sk, (int)blockIdx.x, 0, (int)gridDim.x, 1, seg)) break;
    }
#endif
}

// Grouped launch: up to G8_MAX_GROUP independent problems (the weight gradients of several layers, each far too small to
// fill 256 CUs with 256 x 256 tiles) in ONE grid.  Every problem owns a contiguous, 8-aligned range of block indices
// [first, first + tiles * split) - 8-aligned so that the block -> XCD mapping of the tile walk stays what the body assumes.
constexpr int G8_MAX_GROUP = 14;
template <typename TC, typename SrcA, typename SrcB> struct G8Entry {
    SrcA sa;
    SrcB sb;
    Epilogue<TC> ep;
    int64_t M, N, K;
    int grid_m, tiles, split, first;      // tiles: tile slots of this entry IN THIS LAUNCH; first: its first block index
    int tile0, total;                      // ... which are the slots tile0 .. tile0 + tiles - 1 of the problem's `total`
};
template <typename TC, typename SrcA, typename SrcB> struct G8Group {
    int n;
    G8Entry<TC, SrcA, SrcB> e[G8_MAX_GROUP];
};
template <typename TC, typename SrcA, typename SrcB>
__global__ __launch_bounds__(512) void gemm8_group_kernel(G8Group<TC, SrcA, SrcB> grp) {
    int p = 0;
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.e[i].first) p = i;
    const G8Entry<TC, SrcA, SrcB>& e = grp.e[p];
    const int local = (int)blockIdx.x - e.first;
    if (local >= e.tiles * e.split) return;                    // padding of the range up to a multiple of 8
    gemm8_body<TC, SrcA, SrcB, false>(e.sa, e.sb, e.ep, e.M, e.N, e.K, e.grid_m, StreamK{nullptr, nullptr, 0, 100, nullptr}, e.tile0 + local % e.tiles,
                                      local / e.tiles, e.total, e.split);
}

// Partial-sum slots and flags of the stream-K launches live in a CALLER-OWNED workspace (shg_streamk_workspace_bytes /
// shg_streamk_workspace_init; include/shg_vqa.h): 4 KiB of flags followed by STREAMK_SLOTS partial tiles.  One workspace serves
// the launches of ONE stream at a time (launches on a stream are ordered; two streams could overlap and need one each).
constexpr size_t STREAMK_FLAG_BYTES = 4096;
static StreamK streamk_view(void* ws) {
    StreamK sk{nullptr, nullptr, 0, 100, nullptr};
    if (ws) {
        sk.flags = reinterpret_cast<int*>(ws);
        sk.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + STREAMK_FLAG_BYTES);
    }
    return sk;
}


static std::atomic<int64_t> g_streamk_launches{0};
static int streamk_sigma() {
    const int x = (int)tuning(TUNE_STREAMK_SIGMA);   // measured: conv1 2246 / 2195 / 2190 / 2204 us, conv2 628 / 580 / 586 / 606 us at 100 / 108 / 116 / 125
    return x < 100 ? 100 : (x > 200 ? 200 : x);
}

template <typename TC, typename SrcA, typename SrcB, int ALLOW_SK = 0>     // ALLOW_SK: bit of SHG_STREAMK that enables the split
static int launch8(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, hipStream_t st, const char* what,
                   int split = 1, void* streamk_ws = nullptr) {
    const int64_t gm = (M + 255) / 256, gn = (N + 255) / 256, tiles = gm * gn, nk = K / BK;
    if (tiles > 0x7fffffff) return fail_arg("gemm: grid too large");
    const size_t lds = std::max<size_t>(2 * 8 * Tile64<bf16_t>::BYTES, (size_t)8 * 64 * STG_LD * 4);
    if constexpr (ALLOW_SK) {
        // stream-K when the even split is worth more than its segment overhead: rounds of whole tiles against the busiest
        // XCD's share of K-tiles (32 workgroups each), and ranges long enough to hide a second pipeline fill.
        // SHG_STREAMK: bit 0 conv forward (default on: conv2's 147 tiles 777 -> 620 us, conv1 inside the step 2.18 -> 2.04 ms);
        // the split needs the caller's workspace (streamk_ws), else the launch stays one tile per workgroup.  Measured and not
        // instantiated any more:
        // bit 1 conv input gradient (measured slower: 791 -> 859 us), bit 2 conv weight gradient (2x slower: the gathered-B
        // variant of the segment loop does not keep its registers)
        const int streamk = (int)tuning(TUNE_STREAMK);
        // (every XCD needs 16 <= R < 32 tiles: heads and tails both exist and a tile's tail is cut at most once)
        const int64_t r_min = tiles / 8, r_max = (tiles + 7) / 8, sg = streamk_sigma();
        const int64_t per_wg = (sg * r_max * nk + 100 * (32 - r_max) + sg * r_max - 1) / (100 * (32 - r_max) + sg * r_max);   // head length
        if ((streamk & ALLOW_SK) && streamk_ws && split == 1 && !ep.atomic && r_min >= 16 && r_max < 32 && per_wg >= 64 &&
            nk - per_wg >= 8 && tiles * nk < ((int64_t)1 << 23)) {             // (32-bit plan arithmetic: sigma R nk stays below 2^31)
            StreamK sk = streamk_view(streamk_ws);
            {
                auto kern = gemm8_sk_kernel<TC, SrcA, SrcB>;
                static std::atomic<uint64_t> raised_sk{0};   // per instantiation, one bit per device
                raise_lds_limit(raised_sk, reinterpret_cast<const void*>(kern), (int)lds);
                sk.n_tiles = (int)tiles;
                sk.sigma = (int)sg;
                G8SkArgs<TC, SrcA, SrcB> args{{sa, sb, ep, M, N, K, tile_order(gm, gn), sk, 0}, {}};
                if constexpr (SrcA::SKIP) {
                    if (sa.g.rpp || sa.g.rpt) {      // position- / frame-major rows: tiles of different length -> the weighted plan
                        uint16_t nkt[256];
                        const int go = args.core.grid_m;
                        const int64_t gm_t = go < 0 ? -go : go, gn_t = tiles / gm_t;
                        for (int64_t t = 0; t < tiles; ++t) {
                            const int64_t bm = go < 0 ? t % gm_t : t / gn_t;
                            nkt[t] = (uint16_t)conv_tile_nk(sa.g, bm * 256, M);
                        }
                        // (the tables depend on the shape only: built once per shape and thread, ~2 ms of host time)
                        static thread_local std::vector<uint16_t> key;
                        static thread_local StreamKW cached;
                        static thread_local int cached_ok = 0, cached_sigma = 0;
                        if (key.size() != (size_t)tiles || cached_sigma != (int)sg || !std::equal(key.begin(), key.end(), nkt)) {
                            key.assign(nkt, nkt + tiles);
                            cached_sigma = (int)sg;
                            cached_ok = streamk_w_build(cached, (int)tiles, nkt, (int)sg) ? 1 : 0;
                        }
                        if (cached_ok) { args.wt = cached; args.core.weighted = 1; }
                        else { args.core.sa.g.rpp = 0; args.core.sa.g.rpt = 0; }   // (no weighted plan for these lengths: every tap, the uniform plan)
                    }
                }
                g_streamk_launches.fetch_add(1, std::memory_order_relaxed);
                hipLaunchKernelGGL(kern, dim3(STREAMK_WGS), dim3(512), lds, st, args);
                return check_launch(what);
            }
        }
    }
    // 192-row tiles (plain K-major A, no split): chosen when fewer "tile rounds x work per tile" cover the problem on 256 CUs -
    // 12 576 x 768: 150 tiles in one round at 59 % fill -> 198 tiles at three quarters of the work each; x 1 536: two rounds ->
    // two rounds of 0.75; x 2 304 / x 3 072 stay (594 / 792 tiles would need a third / fourth round).  "gemm8_tile_m" forces 256 / 192.
    if constexpr (std::is_same<SrcA, PlainSrc<bf16_t, true>>::value && ALLOW_SK == 0) {
        const int64_t gm192 = (M + 191) / 192, tiles192 = gm192 * gn, want = tuning(TUNE_GEMM8_TILE_M);
        const int64_t cost256 = 4 * ((tiles + 255) / 256), cost192 = 3 * ((tiles192 + 255) / 256);
        if (split == 1 && !ep.atomic && tiles192 <= 0x7fffffff && (want == 192 || (want == 0 && cost192 < cost256))) {   // (default: 256)
            auto kern = gemm8_kernel<TC, SrcA, SrcB, false, 192>;
            static std::atomic<uint64_t> raised192{0};
            raise_lds_limit(raised192, reinterpret_cast<const void*>(kern), (int)lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles192, 1), dim3(512), lds, st, sa, sb, ep, M, N, K,
                               tile_order(gm192, gn), StreamK{nullptr, nullptr, 0, 100, nullptr});
            return check_launch(what);
        }
    }
    auto kern = gemm8_kernel<TC, SrcA, SrcB, false>;
    static std::atomic<uint64_t> raised{0};          // per instantiation, one bit per device
    raise_lds_limit(raised, reinterpret_cast<const void*>(kern), (int)lds);
    if (split > 1) ep.atomic = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, split), dim3(512), lds, st, sa, sb, ep, M, N, K,
                       tile_order(gm, gn), StreamK{nullptr, nullptr, 0, 100, nullptr});
    return check_launch(what);
}

// the 8-phase kernel is used for bf16 problems with whole K-steps, >= 2 of them, at least `min_tiles` (about half the CUs)
// 256 x 256 tiles and operands addressable with 32-bit byte offsets
static bool use_gemm8(int64_t M, int64_t N, int64_t K, int64_t bytes_a, int64_t bytes_b) {
    const int mode = (int)tuning(TUNE_GEMM8);
    const int64_t min_tiles = tuning(TUNE_GEMM8_MIN_TILES);   // 96 tiles (4096 x 1536): the 128 x 128 kernel wins, 21 vs 31 us
    if (!mode || K % BK || K < 2 * BK) return false;
    if (bytes_a >= ((int64_t)1 << 32) || bytes_b >= ((int64_t)1 << 32)) return false;
    return ((M + 255) / 256) * ((N + 255) / 256) >= min_tiles;
}

template <typename T, typename TC, typename SrcA, typename SrcB, int TM, int TN, int WM, int WN>
static int launch_cfg(SrcA sa, SrcB sb, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, hipStream_t st, const char* what,
                      bool allow_split) {
    constexpr int BMc = TM * 64, BNc = TN * 64, NTHR = 64 * WM * WN;
    const int64_t gm = (M + BMc - 1) / BMc, gn = (N + BNc - 1) / BNc;
    if (gm * gn > 0x7fffffff) return fail_arg("gemm: grid too large");
    // two operand stages / one 64 x 64 fp32 epilogue staging piece per wave
    const size_t lds = std::max<size_t>(2 * (TM + TN) * Tile64<T>::BYTES, (size_t)WM * WN * 64 * STG_LD * 4);
    // split-K (weight gradients: few output tiles, very long contraction): aim at ~1.5 workgroups per CU while
    // keeping >= 4 K-steps per split (measured best of 128/256/384/512/768 on the step's shapes: more
    // splits are throttled by the ~1.3 TB/s chip-wide fp32 atomic rate); partial sums go into the running C
    // with fp32 atomics in 256-byte row segments.
    int split = 1;
    const int64_t nk = (K + BK - 1) / BK;
    if (allow_split && ep.accumulate && gm * gn < 384) {
        const int64_t target = std::max<int64_t>(1, tuning(TUNE_SPLITK_TARGET));
        const int64_t min_steps = std::max<int64_t>(1, tuning(TUNE_SPLITK_MIN_STEPS));
        split = (int)std::min<int64_t>((target + gm * gn - 1) / (gm * gn), std::max<int64_t>(1, nk / min_steps));
        if (split > 1) ep.atomic = 1;
    }
    auto kern = gemm_kernel<T, TC, SrcA, SrcB, TM, TN, WM, WN>;
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};      // per instantiation, one bit per device
        raise_lds_limit(raised, reinterpret_cast<const void*>(kern), (int)lds);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(gm * gn), split), dim3(NTHR), lds, st, sa, sb, ep, M, N, K, tile_order(gm, gn));
    return check_launch(what);
}

// number of 256 x 256 tiles from which the large configuration is used (one workgroup per CU, 256 CUs)
constexpr int64_t LARGE_MIN_TILES = 128;

static bool use_large(int dtype_is_bf16, int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const int64_t min_k = tuning(TUNE_LARGE_MIN_K);
    return dtype_is_bf16 && tiles >= LARGE_MIN_TILES && K >= min_k;
}

template <typename T, typename TC, bool AK, bool BK_>
static int gemm_plain(const T* A, const T* B, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                      hipStream_t st, const char* what) {
    PlainSrc<T, AK> sa{A, lda, 0, M, K};
    PlainSrc<T, BK_> sb{B, ldb, 0, N, K};
    const bool split_ok = !AK && !BK_ && std::is_same<TC, float>::value;
    if constexpr (std::is_same<T, bf16_t>::value && AK) {          // forward (NT) and input-gradient (NN) forms
        if (use_gemm8(M, N, K, M * lda * 2, (BK_ ? N : (int64_t)64) * ldb * 2))
            return launch8<TC, PlainSrc<T, true>, PlainSrc<T, BK_>>(sa, sb, ep, M, N, K, st, what);
    }
    if constexpr (std::is_same<T, bf16_t>::value && AK) {
        if (use_gemm4(M, N, K, M * lda * 2, (BK_ ? N : (int64_t)64) * ldb * 2))
            return launch4<TC, PlainSrc<T, true>, PlainSrc<T, BK_>>(sa, sb, ep, M, N, K, st, what);
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (use_large(1, M, N, K)) return launch_cfg<T, TC, PlainSrc<T, AK>, PlainSrc<T, BK_>, 4, 4, 2, 4>(sa, sb, ep, M, N, K, st, what, split_ok);
    }
    return launch_cfg<T, TC, PlainSrc<T, AK>, PlainSrc<T, BK_>, 2, 2, 2, 2>(sa, sb, ep, M, N, K, st, what, split_ok);
}

template <typename T, typename TC>
static int gemm_dispatch(const void* a, const void* b, Epilogue<TC> ep, int64_t M, int64_t N, int64_t K, int64_t lda,
                         int64_t ldb, int a_kmajor, int b_kmajor, hipStream_t st) {
    const T* A = (const T*)a;
    const T* B = (const T*)b;
    if (a_kmajor && b_kmajor) return gemm_plain<T, TC, true, true>(A, B, ep, M, N, K, lda, ldb, st, "gemm_nt");
    if (a_kmajor && !b_kmajor) return gemm_plain<T, TC, true, false>(A, B, ep, M, N, K, lda, ldb, st, "gemm_nn");
    if (!a_kmajor && b_kmajor) return gemm_plain<T, TC, false, true>(A, B, ep, M, N, K, lda, ldb, st, "gemm_tt");
    return gemm_plain<T, TC, false, false>(A, B, ep, M, N, K, lda, ldb, st, "gemm_tn");
}

// ---------------------------------------------------------------------------------------------
// small helper kernels of the conv path
// ---------------------------------------------------------------------------------------------
// pos_in[m]  = ((b*Tin + to)*Hp + h)*Wp + w         (tap (0,0,0) of output position m in the padded input)
// pos_out[m] = ((b*To  + to)*Hp + h+1)*Wp + w+1      (where output m lives in a spatially padded output)
__global__ void conv_pos_kernel(int32_t* pos_in, int32_t* pos_out, int B, int Tin, int To, int H, int W) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = (int64_t)B * To * H * W;
    if (m >= M) return;
    const int w = (int)(m % W), h = (int)((m / W) % H), to = (int)((m / ((int64_t)W * H)) % To), b = (int)(m / ((int64_t)W * H * To));
    const int Hp = H + 2, Wp = W + 2;
    pos_in[m] = (int32_t)((((int64_t)b * Tin + to) * Hp + h) * Wp + w);
    pos_out[m] = (int32_t)((((int64_t)b * To + to) * Hp + h + 1) * Wp + w + 1);
}

// Row order 1 (position-major): row r = ((h W + w) B + b) To + to.  Same tables for that order, plus std2row[m] = r for the
// standard row m = ((b To + to) H + h) W + w (a producer that computes rows in standard order - the next convolution's input
// gradient - writes them where this order expects them).
// Row order 2 (frame-major): row r = ((to B + b) H + h) W + w - every output frame's rows together.
__global__ void conv_pos_grouped_kernel(int32_t* pos_in, int32_t* pos_out, int32_t* std2row, int32_t* row2std, int B, int Tin, int To, int H,
                                        int W, int order) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = (int64_t)B * To * H * W;
    if (m >= M) return;
    const int w = (int)(m % W), h = (int)((m / W) % H), to = (int)((m / ((int64_t)W * H)) % To), b = (int)(m / ((int64_t)W * H * To));
    const int Hp = H + 2, Wp = W + 2;
    const int64_t r = order == 2 ? (((int64_t)to * B + b) * H + h) * W + w : (((int64_t)h * W + w) * B + b) * To + to;
    pos_in[r] = (int32_t)((((int64_t)b * Tin + to) * Hp + h) * Wp + w);
    pos_out[r] = (int32_t)((((int64_t)b * To + to) * Hp + h + 1) * Wp + w + 1);
    std2row[m] = (int32_t)r;
    row2std[r] = (int32_t)m;
}

// [B,C,T,H,W] fp32 -> [B,T,H+2,W+2,C] (T) with a zero border.  LDS-tiled transpose: a block moves a
// 64(c) x 64(hw-chunk) tile so that both the reads (along hw) and the writes (along c) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void ncdhw_to_padded_cl_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C,
                                                                 int Tn, int H, int W) {
    __shared__ float tile[64][65];
    const int HW = H * W;
    const int bt = blockIdx.z;              // b * T + t
    const int b = bt / Tn, t = bt % Tn;
    const int c0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, s = s0 + tx;
        tile[r][tx] = (c < C && s < HW) ? x[(((int64_t)b * C + c) * Tn + t) * HW + s] : 0.f;
    }
    __syncthreads();
    const int Hp = H + 2, Wp = W + 2;
    for (int r = ty; r < 64; r += 4) {
        const int s = s0 + r, c = c0 + tx;
        if (s < HW && c < C) {
            const int h = s / W, w = s % W;
            y[((((int64_t)bt) * Hp + h + 1) * Wp + w + 1) * C + c] = from_f32<T>(tile[tx][r]);
        }
    }
}

}  // namespace shg

using namespace shg;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int gemm_entry(const void* a, const void* b, void* c, const float* bias, int dtype_ab, int dtype_c, int64_t M,
                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
                      int accumulate, int act, void* pre, void* stream, float p_drop = 0.f,
                      const uint64_t* seed_state = nullptr, uint64_t stream_id = 0, int save_grad = 0) {
    if (!a || !b || !c) return fail_arg("gemm: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) return fail_arg("gemm: sizes must be positive");
    if ((dtype_ab != SHG_F32 && dtype_ab != SHG_BF16) || (dtype_c != SHG_F32 && dtype_c != SHG_BF16)) return fail_arg("gemm: bad dtype");
    if (dtype_ab == SHG_F32 && dtype_c == SHG_BF16) return fail_arg("gemm: fp32 operands need an fp32 C");
    if (act < 0 || act > 2) return fail_arg("gemm: bad activation");
    if (accumulate && (act != SHG_ACT_NONE || pre)) return fail_arg("gemm: accumulate excludes an activation / pre-activation output");
    const int epc = dtype_ab == SHG_BF16 ? 8 : 4;
    if (!al16(a) || !al16(b)) return fail_arg("gemm: A and B must be 16-byte aligned");
    if (lda % epc || ldb % epc) return fail_arg("gemm: lda/ldb must keep rows 16-byte aligned");
    // operands are read in whole 16-byte chunks: a row's contiguous extent is rounded up to the chunk,
    // so the row (lda/ldb) must reach that far and whatever sits in the padding must be finite
    // (it only ever meets zero-filled data of the other operand).
    const int64_t ea = ((a_kmajor ? K : M) + epc - 1) / epc * epc, eb = ((b_kmajor ? K : N) + epc - 1) / epc * epc;
    if (lda < ea || ldb < eb || ldc < N) return fail_arg("gemm: leading dimension too small for 16-byte row reads");
    {   // the kernels address an operand with 32-bit byte offsets from a per-K-step base
        const int64_t es = dtype_ab == SHG_BF16 ? 2 : 4;
        if ((a_kmajor ? M : (int64_t)64) * lda * es >= ((int64_t)1 << 32) || (b_kmajor ? N : (int64_t)64) * ldb * es >= ((int64_t)1 << 32))
            return fail_arg("gemm: operand spans 4 GiB or more");
    }
    hipStream_t st = (hipStream_t)stream;
    const int vlen = dtype_c == SHG_F32 ? 4 : 8;      // elements per 16-byte output vector
    const int vec_ok = (ldc % vlen == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0) &&
                       (!pre || ((N % vlen == 0) && al16(pre)));
    if (p_drop > 0.f && !vec_ok) return fail_arg("gemm: the dropout epilogue needs 16-byte aligned rows of C");
    const uint32_t dthr = dropout_threshold(p_drop);
    const float dscale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (dtype_c == SHG_F32) {
        Epilogue<float> ep{(float*)c, ldc, bias, nullptr, act, accumulate, vec_ok, (float*)pre, 0, nullptr, nullptr, dthr, dscale, seed_state, stream_id};
        return dtype_ab == SHG_F32 ? gemm_dispatch<float, float>(a, b, ep, M, N, K, lda, ldb, a_kmajor, b_kmajor, st)
                                   : gemm_dispatch<bf16_t, float>(a, b, ep, M, N, K, lda, ldb, a_kmajor, b_kmajor, st);
    }
    if (save_grad && !vec_ok) return fail_arg("gemm: SHG_ACT_SAVE_GRAD needs 16-byte aligned rows of C and pre");
    Epilogue<bf16_t> ep{(bf16_t*)c, ldc, bias, nullptr, act, accumulate, vec_ok, (bf16_t*)pre, 0, nullptr, nullptr, dthr, dscale, seed_state, stream_id,
                        tuning(TUNE_EPILOGUE_SIDE) ? 0 : 1, save_grad};
    return gemm_dispatch<bf16_t, bf16_t>(a, b, ep, M, N, K, lda, ldb, a_kmajor, b_kmajor, st);
}

// Weight gradients of several nn.Linear layers in one grid (include/shg_vqa.h: shg_wgrad_group).
extern "C" int shg_wgrad_group(const shg_wgrad_problem_t* probs, int n, int dtype, void* stream) {
    SHG_REPEAT(128, shg_wgrad_group(probs, n, dtype, stream));          // (accumulating: the gradients double - timing runs only)
    if (!probs || n < 0) return fail_arg("wgrad_group: bad argument");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("wgrad_group: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    using SA = PlainSrc<bf16_t, false>;
    using Grp = G8Group<float, SA, SA>;
    // "wgrad_group": bit 0 grouped launches, bit 1 also for row counts that are not a multiple of 64 (ragged last K-tile), bit 2
    // launches of at most 256 workgroups
    const int mode = (int)tuning(TUNE_WGRAD_GROUP);
    int i = 0;
    while (i < n) {
        // take a run of problems the 8-phase kernel can do: bf16, whole 64-row K-tiles, 16-byte aligned rows
        int j = i;
        int64_t tiles = 0;
        auto ok8 = [&](const shg_wgrad_problem_t& q) {
            return mode && dtype == SHG_BF16 && (q.rows % BK == 0 || (mode & 2)) && q.rows >= 2 * BK && q.n_out % 8 == 0 && q.n_in % 8 == 0 && q.ldy % 8 == 0 &&
                   q.ldx % 8 == 0 && al16(q.dy) && al16(q.x) && al16(q.gw) && q.n_in % 4 == 0 &&
                   (int64_t)BK * q.ldy * 2 < ((int64_t)1 << 32) && (int64_t)BK * q.ldx * 2 < ((int64_t)1 << 32);
        };
        while (j < n && ok8(probs[j])) {
            tiles += ((probs[j].n_out + 255) / 256) * ((probs[j].n_in + 255) / 256);
            ++j;
        }
        if (j - i >= 2 && tiles >= 48) {
            for (int k = i; k < j; ++k) {
                const shg_wgrad_problem_t& q = probs[k];
                if (!q.dy || !q.x || !q.gw || q.rows <= 0 || q.n_out <= 0 || q.n_in <= 0) return fail_arg("wgrad_group: bad problem");
            }
            auto kern = gemm8_group_kernel<float, SA, SA>;
            const size_t lds = std::max<size_t>(2 * 8 * Tile64<bf16_t>::BYTES, (size_t)8 * 64 * STG_LD * 4);
            static std::atomic<uint64_t> raised{0};
            raise_lds_limit(raised, reinterpret_cast<const void*>(kern), (int)lds);
            // Launches of at most 256 workgroups - ONE round of the 256 CUs: a workgroup owns its CU for a whole contraction
            // (197 K-tiles = 300 us at 12 576 rows), so a launch of 336 workgroups ran two rounds, the second one a third full
            // (round 2 packed whatever the executor's queue held: 272-368 workgroups per launch, 718 / 756 us for the relation
            // layers' two groups).  A problem whose tiles do not fit the current launch continues in the next one (tile0).
            // Small runs (< 160 tiles in all) stay one launch with the contraction split over gridDim-like slots (atomics).
            const bool small = tiles < 160 && j - i <= G8_MAX_GROUP;
            // ("wgrad_group" bit 2 off: launches as large as the queue, as in round 2; "wgrad_group_cap": fewer than one round, so
            //  that the dependent chains on the other streams always find free CUs; "wgrad_group_split": contraction of every
            //  problem in that many parts = shorter-lived workgroups, partial sums added with atomics)
            const int cap = (mode & 4) ? (int)std::min<int64_t>(256, std::max<int64_t>(64, tuning(TUNE_WGRAD_GROUP_CAP) / 8 * 8)) : (1 << 20);
            const int force_split = (int)std::min<int64_t>(8, std::max<int64_t>(1, tuning(TUNE_WGRAD_GROUP_SPLIT)));
            Grp g{};
            int used = 0;
            auto launch = [&]() -> int {
                if (!g.n) return 0;
                hipLaunchKernelGGL(kern, dim3((unsigned)used), dim3(512), lds, st, g);
                g = Grp{};
                used = 0;
                return check_launch("wgrad_group");
            };
            for (int k = i; k < j; ++k) {
                const shg_wgrad_problem_t& q = probs[k];
                const int64_t gm = (q.n_out + 255) / 256, gn = (q.n_in + 255) / 256, nk = (q.rows + BK - 1) / BK;
                const int total = (int)(gm * gn);
                int split = 1;
                if (small) split = (int)std::max<int64_t>(1, std::min<int64_t>((224 + tiles - 1) / tiles, nk / 12));
                else if (force_split > 1 && nk / force_split >= 12) split = force_split;
                // a weight applied more than once (the cross layers' shared modules, modeling_capsbert.py:1247-1249) has several
                // problems adding into ONE gradient inside this run: those add with atomics
                bool shared = false;
                for (int k2 = i; k2 < j; ++k2) shared = shared || (k2 != k && probs[k2].gw == q.gw);
                int t0 = 0;
                while (t0 < total) {
                    if (g.n == G8_MAX_GROUP || (!small && used >= cap)) {
                        if (int e = launch()) return e;
                    }
                    int c = total - t0;
                    if (!small) {
                        const int room = (cap - used) / split;          // (a multiple of 8)
                        if (c > room) c = room / 8 * 8;                 // an unfinished problem leaves the launch on an 8-boundary
                        if (c == 0) {
                            if (int e = launch()) return e;
                            continue;
                        }
                    }
                    G8Entry<float, SA, SA>& e = g.e[g.n++];
                    e.sa = SA{(const bf16_t*)q.dy, q.ldy, 0, q.n_out, q.rows};
                    e.sb = SA{(const bf16_t*)q.x, q.ldx, 0, q.n_in, q.rows};
                    e.ep = Epilogue<float>{q.gw, q.n_in, nullptr, nullptr, SHG_ACT_NONE, 1, 1, nullptr, (split > 1 || shared) ? 1 : 0};
                    e.M = q.n_out; e.N = q.n_in; e.K = q.rows;
                    e.grid_m = tile_order(gm, gn);
                    e.tiles = c;
                    e.tile0 = t0;
                    e.total = total;
                    e.split = split;
                    e.first = used;
                    used += (c * split + 7) / 8 * 8;
                    t0 += c;
                }
            }
            if (int e = launch()) return e;
            i = j;
            continue;
        }
        // not groupable (fp32 parity mode, ragged row count, a lone problem): the plain weight-gradient GEMM
        const int end = j > i ? j : i + 1;
        for (int k = i; k < end; ++k) {
            const shg_wgrad_problem_t& q = probs[k];
            if (int e = shg_gemm(q.dy, q.x, q.gw, nullptr, dtype, SHG_F32, q.n_out, q.n_in, q.rows, q.ldy, q.ldx, q.n_in, 0, 0, 1, stream)) return e;
        }
        i = end;
    }
    return 0;
}

// host evaluation of the device-side work split (tests/test_abi.py checks that heads and tails cover every K-tile of every
// tile exactly once): out = {tile, first K-tile, K-tiles, owner, slot, parts}; returns 0 when the segment does not exist
extern "C" int shg_streamk_plan(int n_tiles, int nk, int block, int seg, int* out) {
    if (!out || n_tiles < 128 || n_tiles >= 256 || nk < 2 || block < 0 || block >= shg::STREAMK_WGS || seg < 0) return shg::fail_arg("streamk_plan: bad argument");
    const shg::StreamKSeg d = shg::streamk_plan(n_tiles, nk, block, shg::STREAMK_WGS, seg, shg::streamk_sigma());
    out[0] = d.tile; out[1] = d.kb; out[2] = d.nk; out[3] = d.owner; out[4] = d.slot; out[5] = d.parts;
    return d.nk > 0 ? 1 : 0;
}

// the weighted plan on the host: nk_tile[n_tiles] K-tiles per tile -> the same descriptor; -1: no weighted plan for these lengths
extern "C" int shg_streamk_plan_weighted(int n_tiles, const uint16_t* nk_tile, int block, int seg, int* out) {
    if (!out || !nk_tile || n_tiles < 128 || n_tiles >= 256 || block < 0 || block >= shg::STREAMK_WGS || seg < 0) return shg::fail_arg("streamk_plan_weighted: bad argument");
    static thread_local shg::StreamKW w;
    static thread_local std::vector<uint16_t> cached;
    if (cached.size() != (size_t)n_tiles || !std::equal(cached.begin(), cached.end(), nk_tile)) {
        cached.assign(nk_tile, nk_tile + n_tiles);
        if (!shg::streamk_w_build(w, n_tiles, nk_tile, shg::streamk_sigma())) { cached.clear(); return -1; }
    }
    const shg::StreamKSeg d = shg::streamk_plan_w((const shg::StreamKW*)&w, n_tiles, block, shg::STREAMK_WGS, seg);
    out[0] = d.tile; out[1] = d.kb; out[2] = d.nk; out[3] = d.owner; out[4] = d.slot; out[5] = d.parts;
    return d.nk > 0 ? 1 : 0;
}

extern "C" int64_t shg_gemm_streamk_launches(void) { return shg::g_streamk_launches.load(std::memory_order_relaxed); }

extern "C" int shg_gemm(const void* a, const void* b, void* c, const float* bias, int dtype_ab, int dtype_c, int64_t M,
                        int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
                        int accumulate, void* stream) {
    if (!accumulate && dtype_c == SHG_BF16)
        SHG_REPEAT(((M + 255) / 256) * ((N + 255) / 256) >= 120 ? 16 : 32,
                   shg_gemm(a, b, c, bias, dtype_ab, dtype_c, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, accumulate, stream));
    if (accumulate && dtype_c == SHG_BF16)      // (`C +=` onto a residual gradient: doubles that contribution - timing runs only)
        SHG_REPEAT(2048, shg_gemm(a, b, c, bias, dtype_ab, dtype_c, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, accumulate, stream));
    return gemm_entry(a, b, c, bias, dtype_ab, dtype_c, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, accumulate, SHG_ACT_NONE,
                      nullptr, stream);
}

extern "C" int shg_gemm_kseg(const void* a, const void* b, void* c, int dtype, int64_t M, int64_t N, int64_t seg_k, int n_seg,
                             int64_t lda, int64_t ldb, int64_t ldc, int64_t a_seg_stride, int64_t b_seg_stride, int accumulate,
                             void* stream) {
    if (!a || !b || !c) return fail_arg("gemm_kseg: null pointer");
    if (M <= 0 || N <= 0 || seg_k <= 0 || n_seg < 1 || n_seg > 64) return fail_arg("gemm_kseg: bad sizes");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("gemm_kseg: bad dtype");
    const int64_t es = dtype == SHG_BF16 ? 2 : 4, epc = 16 / es;
    if (a_seg_stride % epc || b_seg_stride % epc) return fail_arg("gemm_kseg: segment strides must keep 16-byte alignment");
    const int64_t K = seg_k * n_seg, gm = (M + 255) / 256, gn = (N + 255) / 256;
    if (dtype == SHG_BF16 && seg_k % BK == 0 && K / BK < 4000 && N % 8 == 0 && ldc % 8 == 0 && al16(a) && al16(b) && al16(c) && lda % 8 == 0 &&
        ldb % 8 == 0 && lda >= seg_k && ldb >= N && ldc >= N && tuning(TUNE_GEMM8) && gm * gn >= tuning(TUNE_GEMM8_MIN_TILES) &&
        M * lda * es < ((int64_t)1 << 32) && (int64_t)64 * ldb * es < ((int64_t)1 << 32)) {
        SHG_REPEAT(2048, shg_gemm_kseg(a, b, c, dtype, M, N, seg_k, n_seg, lda, ldb, ldc, a_seg_stride, b_seg_stride, 1, stream));
        const int st_tiles = (int)(seg_k / BK);
        const uint32_t magic = (1u << 20) / (uint32_t)st_tiles + 1u;
        SegSrc<bf16_t, true> sa{{(const bf16_t*)a, lda, 0, M, K}, st_tiles, magic, a_seg_stride};
        SegSrc<bf16_t, false> sb{{(const bf16_t*)b, ldb, 0, N, K}, st_tiles, magic, b_seg_stride};
        Epilogue<bf16_t> ep{(bf16_t*)c, ldc, nullptr, nullptr, SHG_ACT_NONE, accumulate ? 1 : 0, 1, nullptr, 0, nullptr, nullptr, 0, 1.f, nullptr, 0,
                            tuning(TUNE_EPILOGUE_SIDE) ? 0 : 1};
        return launch8<bf16_t, decltype(sa), decltype(sb)>(sa, sb, ep, M, N, K, (hipStream_t)stream, "gemm_kseg");
    }
    for (int s = 0; s < n_seg; ++s)                  // elsewhere: the segments one after the other
        if (int e = shg_gemm((const char*)a + s * a_seg_stride * es, (const char*)b + s * b_seg_stride * es, c, nullptr, dtype, dtype, M, N, seg_k, lda,
                             ldb, ldc, 1, 0, (accumulate || s > 0) ? 1 : 0, stream))
            return e;
    return 0;
}

static int drop_args_ok(float p_drop, const uint64_t* seed_state, int64_t N) {
    if (p_drop < 0.f || p_drop >= 1.f) return fail_arg("gemm: p_drop must be in [0, 1)");
    if (p_drop > 0.f && !seed_state) return fail_arg("gemm: dropout needs seed_state");
    if (p_drop > 0.f && (N % 8)) return fail_arg("gemm: the dropout epilogue needs N % 8 == 0");
    return 0;
}

extern "C" int shg_gemm_dact(const void* dy, const void* w, void* dx, const void* pre, float* dbias, int dtype, int64_t M,
                             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int act, float p_drop,
                             const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    SHG_REPEAT(1024, shg_gemm_dact(dy, w, dx, pre, dbias, dtype, M, N, K, lda, ldb, ldc, act, p_drop, seed_state, stream_id, stream));
    if (!dy || !w || !dx || !pre) return fail_arg("gemm_dact: null pointer");
    if (M <= 0 || N <= 0 || K <= 0) return fail_arg("gemm_dact: sizes must be positive");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("gemm_dact: bad dtype");
    if (act < 0 || act > 3 || (act == SHG_ACT_SAVED_GRAD && dtype != SHG_BF16)) return fail_arg("gemm_dact: bad activation");
    if (int e = drop_args_ok(p_drop, seed_state, N)) return e;
    const int epc = dtype == SHG_BF16 ? 8 : 4;
    if (N % 8 || K % epc) return fail_arg("gemm_dact: N must be a multiple of 8 and K of the 16-byte chunk");
    if (!al16(dy) || !al16(w) || !al16(dx) || !al16(pre)) return fail_arg("gemm_dact: pointers must be 16-byte aligned");
    if (lda % epc || ldb % epc || ldc % epc || lda < K || ldb < N || ldc < N) return fail_arg("gemm_dact: bad leading dimension");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = dropout_threshold(p_drop);
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (dtype == SHG_F32) {
        Epilogue<float> ep{(float*)dx, ldc, nullptr, nullptr, act, 0, 1, nullptr, 0, (const float*)pre, dbias, thr, scale, seed_state, stream_id};
        return gemm_dispatch<float, float>(dy, w, ep, M, N, K, lda, ldb, 1, 0, st);
    }
    Epilogue<bf16_t> ep{(bf16_t*)dx, ldc, nullptr, nullptr, act, 0, 1, nullptr, 0, (const bf16_t*)pre, dbias, thr, scale, seed_state, stream_id,
                        tuning(TUNE_EPILOGUE_SIDE) ? 0 : 1};
    return gemm_dispatch<bf16_t, bf16_t>(dy, w, ep, M, N, K, lda, ldb, 1, 0, st);
}

extern "C" int shg_gemm_act(const void* a, const void* b, void* c, const float* bias, int dtype_ab, int dtype_c, int64_t M,
                            int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
                            int act, void* pre, float p_drop, const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    if (dtype_c == SHG_BF16)
        SHG_REPEAT(((M + 255) / 256) * ((N + 255) / 256) >= 120 ? 16 : 32,
                   shg_gemm_act(a, b, c, bias, dtype_ab, dtype_c, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, act, pre, p_drop, seed_state, stream_id, stream));
    if (int e = drop_args_ok(p_drop, seed_state, N)) return e;
    const int save_grad = (act & SHG_ACT_SAVE_GRAD) ? 1 : 0;
    act &= ~SHG_ACT_SAVE_GRAD;
    if (save_grad && (dtype_c != SHG_BF16 || !pre || p_drop > 0.f || N % 8))
        return fail_arg("gemm_act: SHG_ACT_SAVE_GRAD needs a bf16 output, a `pre` buffer, N % 8 == 0 and no dropout");
    return gemm_entry(a, b, c, bias, dtype_ab, dtype_c, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, 0, act, pre, stream, p_drop,
                      seed_state, stream_id, save_grad);
}

extern "C" int64_t shg_conv3d_k533_workspace_bytes(int B, int T, int H, int W) {
    if (B < 1 || T < 5 || H < 1 || W < 1) return -1;
    const int64_t M = (int64_t)B * (T - 4) * H * W;
    return 2 * ((M * 4 + 255) / 256) * 256;
}

static int conv_check(int dtype, int B, int T, int H, int W, int Cin, int Cout, const void* ws) {
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("conv3d: bad dtype");
    if (B < 1 || T < 5 || H < 1 || W < 1) return fail_arg("conv3d: need T >= 5 and positive sizes");
    if (Cin % 64 || Cout % 8) return fail_arg("conv3d: Cin must be a multiple of 64 and Cout of 8");
    if ((int64_t)B * T * (H + 2) * (W + 2) > 0x7fffffff) return fail_arg("conv3d: position index overflows int32");
    if (!ws || !al16(ws)) return fail_arg("conv3d: workspace missing or unaligned");
    return 0;
}

extern "C" int shg_conv3d_k533_prepare(void* workspace, int B, int T, int H, int W, void* stream) {
    if (!workspace || B < 1 || T < 5 || H < 1 || W < 1) return fail_arg("conv3d_prepare: bad argument");
    const int64_t M = (int64_t)B * (T - 4) * H * W;
    int32_t* pos_in = (int32_t*)workspace;
    int32_t* pos_out = (int32_t*)((char*)workspace + shg_conv3d_k533_workspace_bytes(B, T, H, W) / 2);
    hipLaunchKernelGGL(conv_pos_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos_in, pos_out, B, T, T - 4, H, W);
    return check_launch("conv3d_prepare");
}

extern "C" int64_t shg_conv3d_k533_workspace_bytes_ex(int B, int T, int H, int W, int row_order) {
    const int64_t two = shg_conv3d_k533_workspace_bytes(B, T, H, W);
    if (two < 0 || row_order < 0 || row_order > 2) return -1;
    return row_order ? two * 2 : two;
}

extern "C" int shg_conv3d_k533_prepare_ex(void* workspace, int B, int T, int H, int W, int row_order, void* stream) {
    if (row_order == 0) return shg_conv3d_k533_prepare(workspace, B, T, H, W, stream);
    if (!workspace || B < 1 || T < 5 || H < 1 || W < 1 || row_order < 1 || row_order > 2) return fail_arg("conv3d_prepare_ex: bad argument");
    const int64_t M = (int64_t)B * (T - 4) * H * W, seg = shg_conv3d_k533_workspace_bytes(B, T, H, W) / 2;
    int32_t* pos_in = (int32_t*)workspace;
    int32_t* pos_out = (int32_t*)((char*)workspace + seg);
    int32_t* std2row = (int32_t*)((char*)workspace + 2 * seg);
    int32_t* row2std = (int32_t*)((char*)workspace + 3 * seg);
    hipLaunchKernelGGL(conv_pos_grouped_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos_in, pos_out, std2row,
                       row2std, B, T, T - 4, H, W, row_order);
    return check_launch("conv3d_prepare_ex");
}

extern "C" int64_t shg_streamk_workspace_bytes(void) {
    return (int64_t)(STREAMK_FLAG_BYTES + STREAMK_SLOTS * STREAMK_SLOT * sizeof(float));
}

extern "C" int shg_streamk_workspace_init(void* ws, void* stream) {
    if (!ws || !al16(ws)) return fail_arg("streamk_workspace_init: workspace missing or unaligned");
    hipError_t e = hipMemsetAsync(ws, 0, STREAMK_FLAG_BYTES, (hipStream_t)stream);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int shg_conv3d_k533_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int B, int T,
                                   int H, int W, int Cin, int Cout, int act, int pad_out, void* y_pre,
                                   const void* workspace, void* streamk_ws, void* stream) {
    return shg_conv3d_k533_fwd_rows(x, w, bias, y, dtype, B, T, H, W, Cin, Cout, act, pad_out, y_pre, nullptr, nullptr, 0, workspace, streamk_ws, stream);
}

extern "C" int shg_conv3d_k533_fwd_rows(const void* x, const void* w, const float* bias, void* y, int dtype, int B, int T,
                                        int H, int W, int Cin, int Cout, int act, int pad_out, void* y_pre, const int32_t* pre_rows,
                                        const int32_t* y_rows, int row_order, const void* workspace, void* streamk_ws, void* stream) {
    SHG_REPEAT(64, shg_conv3d_k533_fwd_rows(x, w, bias, y, dtype, B, T, H, W, Cin, Cout, act, pad_out, y_pre, pre_rows, y_rows, row_order, workspace,
                                            streamk_ws, stream));
    if (row_order < 0 || row_order > 1) return fail_arg("conv3d_fwd_rows: row_order must be 0 or 1");
    if (pad_out && y_rows) return fail_arg("conv3d_fwd_rows: y_rows is for the dense output (pad_out = 0)");
    if (!x || !w || !y) return fail_arg("conv3d_fwd: null pointer");
    if (streamk_ws && !al16(streamk_ws)) return fail_arg("conv3d_fwd: streamk workspace must be 16-byte aligned");
    if (y_pre && !al16(y_pre)) return fail_arg("conv3d_fwd: y_pre must be 16-byte aligned");
    if (int e = conv_check(dtype, B, T, H, W, Cin, Cout, workspace)) return e;
    if (!al16(x) || !al16(w) || !al16(y)) return fail_arg("conv3d_fwd: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)B * (T - 4) * H * W, N = Cout, K = (int64_t)45 * Cin;
    const int32_t* pos_in = (const int32_t*)workspace;
    const int32_t* pos_out = (const int32_t*)((const char*)workspace + shg_conv3d_k533_workspace_bytes(B, T, H, W) / 2);
    ConvGeom g = conv_geom(Cin, H + 2, W + 2);
    // position-major rows (the caller's tables are of row order 1): a tile drops the taps that read only the zero border for all of
    // its rows ("conv_k_order" bit 5; needs the channel-block-major K order and the stream-K launch's weighted plan to pay)
    if (row_order == 1 && g.kperm && dtype == SHG_BF16 && (tuning(TUNE_CONV_K_ORDER) & 32) && (int64_t)45 * (Cin / 64) < 4000) {
        g.rpp = B * (T - 4);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) {
        ConvRowSrc<float, 256> sa{(const float*)x, pos_in, 0, M, g};
        PlainSrc<float, true> sb{(const float*)w, K, 0, N, K};
        Epilogue<float> ep{(float*)y, Cout, bias, pad_out ? pos_out : y_rows, act, 0, 1, (float*)y_pre, 0};
        ep.prow = pre_rows;
        return launch_cfg<float, float, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, M, N, K, st, "conv3d_k533_fwd", false);
    }
    PlainSrc<bf16_t, true> sb{(const bf16_t*)w, K, 0, N, K};
    Epilogue<bf16_t> ep{(bf16_t*)y, Cout, bias, pad_out ? pos_out : y_rows, act, 0, 1, (bf16_t*)y_pre, 0};
    ep.prow = pre_rows;
    if (use_gemm8(M, N, K, (int64_t)B * T * (H + 2) * (W + 2) * Cin * 2, N * K * 2)) {
        if (g.rpp) {                                 // the instantiation with per-tile tap lists (and the weighted stream-K plan)
            ConvRowSrc<bf16_t, 512, true> sa{(const bf16_t*)x, pos_in, 0, M, g};
            return launch8<bf16_t, decltype(sa), decltype(sb), 1>(sa, sb, ep, M, N, K, st, "conv3d_k533_fwd", 1, streamk_ws);
        }
        ConvRowSrc<bf16_t, 512> sa{(const bf16_t*)x, pos_in, 0, M, g};
        return launch8<bf16_t, decltype(sa), decltype(sb), 1>(sa, sb, ep, M, N, K, st, "conv3d_k533_fwd", 1, streamk_ws);
    }
    g.rpp = 0;                                       // (the other kernels run every tap)
    if (use_large(1, M, N, K)) {
        ConvRowSrc<bf16_t, 512> sa{(const bf16_t*)x, pos_in, 0, M, g};
        return launch_cfg<bf16_t, bf16_t, decltype(sa), decltype(sb), 4, 4, 2, 4>(sa, sb, ep, M, N, K, st, "conv3d_k533_fwd", false);
    }
    ConvRowSrc<bf16_t, 256> sa{(const bf16_t*)x, pos_in, 0, M, g};
    return launch_cfg<bf16_t, bf16_t, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, M, N, K, st, "conv3d_k533_fwd", false);
}

// sum of squares of `blks` 256-column blocks (logical blocks lb0 .. of a [rows, R] fp32 matrix, the weight gradient's column
// order) / of a contiguous fp32 range, added to *out
__global__ __launch_bounds__(256) void sumsq_blocks_kernel(const float* __restrict__ w, int64_t R, int rows, int lb0, int blks, int nblk,
                                                           int lpt, double* __restrict__ out) {
    ConvColSrc<bf16_t, 512> m{};
    m.nblk = nblk;
    m.lpt = lpt;
    double acc = 0.0;
    const int64_t segs = (int64_t)rows * blks;
    for (int64_t q = blockIdx.x; q < segs; q += gridDim.x) {
        const int64_t blk = q / rows, row = q - blk * rows;
        const float v = w[row * R + m.col_of_lblock(lb0 + blk) + threadIdx.x];
        acc += (double)(v * v);
    }
    __shared__ double sh[4];
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sumsq_range_kernel(const float* __restrict__ w, int64_t n, double* __restrict__ out) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += (double)(w[i] * w[i]);
    __shared__ double sh[4];
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// sumsq != null (only with whole 256 x 256 tiles on the 8-phase kernel, conv_wgrad_impl decides): OVERWRITE mode - dw rows
// [c0, c0 + cn) = the gradient, *sumsq += the sum of their squares out of the accumulators
static int conv_wgrad_core(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin, int Cout, int c0,
                           int cn, int accumulate, double* sumsq, int row_order, const void* workspace, void* stream) {
    if (!x || !dy || !dw) return fail_arg("conv3d_wgrad: null pointer");
    if (int e = conv_check(dtype, B, T, H, W, Cin, Cout, workspace)) return e;
    if (c0 < 0 || cn < 1 || c0 + cn > Cout || c0 % 8 || cn % 8) return fail_arg("conv3d_wgrad: bad output-channel slice (multiples of 8 inside [0, Cout))");
    if (!al16(x) || !al16(dy) || !al16(dw)) return fail_arg("conv3d_wgrad: pointers must be 16-byte aligned");
    // dW[co][(tap, ci)] = sum_m dY[m][co] * Xgather[m][(tap, ci)] : GEMM with M' = Cout, N' = 45*Cin, K' = M; an output-channel
    // slice [c0, c0 + cn) is the rows c0.. of that GEMM: columns c0.. of dY (row stride Cout), rows c0.. of dW (contiguous)
    const int64_t Mo = (int64_t)B * (T - 4) * H * W, Ncols = (int64_t)45 * Cin;
    const int32_t* pos_in = (const int32_t*)workspace;
    ConvGeom g = conv_geom(Cin, H + 2, W + 2);
    hipStream_t st = (hipStream_t)stream;
    Epilogue<float> ep{dw + (int64_t)c0 * Ncols, Ncols, nullptr, nullptr, SHG_ACT_NONE, accumulate, 1, nullptr, 0};
    if (dtype == SHG_F32) {
        PlainSrc<float, false> sa{(const float*)dy + c0, Cout, 0, cn, Mo};
        ConvColSrc<float, 256> sb{(const float*)x, pos_in, 0, Ncols, Mo, g};
        return launch_cfg<float, float, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, cn, Ncols, Mo, st, "conv3d_k533_wgrad", false);
    }
    PlainSrc<bf16_t, false> sa{(const bf16_t*)dy + c0, Cout, 0, cn, Mo};
    if (use_gemm8(cn, Ncols, Mo, (int64_t)64 * Cout * 2, (int64_t)B * T * (H + 2) * (W + 2) * Cin * 2)) {
        ConvColSrc<bf16_t, 512> sb{(const bf16_t*)x, pos_in, 0, Ncols, Mo, g};
        if (Cin % 256 == 0 && (tuning(TUNE_CONV_K_ORDER) & 4)) sb.nblk = Cin / 256;
        // position-major rows, whole K-tiles per position: skip the positions where the tile's tap reads the zero border
        if (row_order == 1 && Cin % 256 == 0 && H >= 2 && W >= 2 && ((int64_t)B * (T - 4)) % BK == 0 && Mo / BK < 4000 && (tuning(TUNE_CONV_K_ORDER) & 8)) {
            sb.tpp = (int)((int64_t)B * (T - 4) / BK);
            sb.tpp_magic = (1 << 20) / sb.tpp + 1;
            sb.Hin = H;
            sb.Win = W;
            sb.lpt = (sb.nblk && (tuning(TUNE_CONV_K_ORDER) & 16)) ? 1 : 0;
        }
        // Tile-count quantisation: conv1's 3 x 360 = 1 080 tiles are 4.22 rounds of 256 CUs - the fifth round runs 56
        // workgroups for the full K = 18 816 while 200 CUs idle (0.36 ms of a 2.3 ms launch, the last kernel of backward).
        // The column blocks of the whole rounds go out as one launch; the remaining blocks as a second launch with the
        // contraction split over gridDim.y (fp32 atomic adds into the running sum - which is why this needs `accumulate`),
        // so that the remainder takes a fraction of a round.  SHG_CONV_WGRAD_REMAINDER=0 switches it off.
        const int rem_on = (int)tuning(TUNE_CONV_WGRAD_REMAINDER);
        const int64_t tiles_m = (cn + 255) / 256, gn = Ncols / 256, total = tiles_m * gn, rounds = total / 256, rem = total % 256;
        const bool fused = sumsq != nullptr;                                  // whole tiles: the accumulators are the stored values
        if (fused) ep.sumsq = sumsq;
        if (rem_on && (accumulate || fused) && Ncols % 256 == 0 && rounds >= 1 && rem > 0 && Mo / BK >= 16) {
            const int64_t gn_a = rounds * 256 / tiles_m, gn_b = gn - gn_a, tiles_b = tiles_m * gn_b;
            // split of the remainder launch: the one with the shortest modelled time (rounds of 256 workgroups, each
            // 19 us + 1.52 us per K-tile - the kernel's measured K scan), if that beats one plain round by 30 %
            // (conv2's 149 remainder tiles at a split of three: modelled 236 against 317 us, measured 594 against 576 us for the launch)
            const double nk = (double)(Mo / BK);
            auto model = [&](int sp) { return (double)((tiles_b * sp + 255) / 256) * (19.0 + nk / sp * 1.52); };
            int split = 1;
            for (int sp = 2; sp <= 8 && nk / sp >= 8.0; ++sp)
                if (model(sp) < model(split)) split = sp;
            if (gn_a >= 1 && gn_b >= 1 && split >= 2 && model(split) < 0.7 * model(1)) {
                if (fused) {                 // the first launch zeroes the remainder's blocks, the split launch adds into them
                    sb.zero_base = ep.c; sb.zero_blk0 = (int)gn_a; sb.zero_blks = (int)gn_b; sb.zero_rows = cn;
                }
                if (int e = launch8<float, decltype(sa), decltype(sb)>(sa, sb, ep, cn, gn_a * 256, Mo, st, "conv3d_k533_wgrad")) return e;
                sb.zero_base = nullptr;
                float* const c_rows = ep.c;
                if (sb.nblk || fused) sb.lblk0 = (int)gn_a;
                else { sb.cbase = gn_a * 256; ep.c += gn_a * 256; }
                ep.sumsq = nullptr;
                ep.accumulate = 1;
                if (int e = launch8<float, decltype(sa), decltype(sb)>(sa, sb, ep, cn, gn_b * 256, Mo, st, "conv3d_k533_wgrad", split)) return e;
                if (fused) {
                    hipLaunchKernelGGL(sumsq_blocks_kernel, dim3((unsigned)std::min<int64_t>(1024, cn * gn_b)), dim3(256), 0, st, c_rows, Ncols, cn,
                                       (int)gn_a, (int)gn_b, sb.nblk, sb.lpt, sumsq);
                    return check_launch("conv3d_k533_wgrad_sumsq");
                }
                return 0;
            }
        }
        return launch8<float, decltype(sa), decltype(sb)>(sa, sb, ep, cn, Ncols, Mo, st, "conv3d_k533_wgrad");
    }
    if (use_large(1, cn, Ncols, Mo)) {
        ConvColSrc<bf16_t, 512> sb{(const bf16_t*)x, pos_in, 0, Ncols, Mo, g};
        return launch_cfg<bf16_t, float, decltype(sa), decltype(sb), 4, 4, 2, 4>(sa, sb, ep, cn, Ncols, Mo, st, "conv3d_k533_wgrad", false);
    }
    ConvColSrc<bf16_t, 256> sb{(const bf16_t*)x, pos_in, 0, Ncols, Mo, g};
    return launch_cfg<bf16_t, float, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, cn, Ncols, Mo, st, "conv3d_k533_wgrad", false);
}

// The overwrite form.  Fused where the 8-phase kernel runs whole tiles; elsewhere (fp32 parity mode, small problems, slices that
// are not multiples of 256 rows) the rows are zeroed, the usual accumulating launch follows - the very arithmetic of the plain
// scheme, bit for bit - and one pass over the finished rows adds their squares.
static int conv_wgrad_impl(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin, int Cout, int c0,
                           int cn, int accumulate, double* sumsq, int row_order, const void* workspace, void* stream) {
    if (row_order < 0 || row_order > 1) return fail_arg("conv3d_wgrad: row_order must be 0 (standard) or 1 (position-major)");
    if (!sumsq) return conv_wgrad_core(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, accumulate, nullptr, row_order, workspace, stream);
    if (!x || !dy || !dw) return fail_arg("conv3d_wgrad: null pointer");
    if (int e = conv_check(dtype, B, T, H, W, Cin, Cout, workspace)) return e;
    if (c0 < 0 || cn < 1 || c0 + cn > Cout || c0 % 8 || cn % 8) return fail_arg("conv3d_wgrad: bad output-channel slice (multiples of 8 inside [0, Cout))");
    const int64_t Mo = (int64_t)B * (T - 4) * H * W, Ncols = (int64_t)45 * Cin;
    const bool fuse = dtype == SHG_BF16 && cn % 256 == 0 && Ncols % 256 == 0 &&
                      use_gemm8(cn, Ncols, Mo, (int64_t)64 * Cout * 2, (int64_t)B * T * (H + 2) * (W + 2) * Cin * 2);
    if (fuse) return conv_wgrad_core(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, 0, sumsq, row_order, workspace, stream);
    float* rows = dw + (int64_t)c0 * Ncols;
    hipError_t e = hipMemsetAsync(rows, 0, (size_t)cn * Ncols * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return (int)e; }
    if (int rc = conv_wgrad_core(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, 1, nullptr, row_order, workspace, stream)) return rc;
    hipLaunchKernelGGL(sumsq_range_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, rows, (int64_t)cn * Ncols, sumsq);
    return check_launch("conv3d_k533_wgrad_sumsq");
}

extern "C" int shg_conv3d_k533_wgrad_slice(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W,
                                           int Cin, int Cout, int c0, int cn, int accumulate, const void* workspace, void* stream) {
    SHG_REPEAT(256, shg_conv3d_k533_wgrad_slice(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, accumulate, workspace, stream));
    return conv_wgrad_impl(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, accumulate, nullptr, 0, workspace, stream);
}

extern "C" int shg_conv3d_k533_wgrad_ex(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W, int Cin,
                                        int Cout, int c0, int cn, int accumulate, double* sumsq, int row_order, const void* workspace,
                                        void* stream) {
    if (sumsq && (reinterpret_cast<uintptr_t>(sumsq) & 7)) return fail_arg("conv3d_wgrad_ex: sumsq must be 8-byte aligned");
    if (sumsq && accumulate) return fail_arg("conv3d_wgrad_ex: the sum of squares comes with the overwrite form (accumulate = 0)");
    SHG_REPEAT(256, conv_wgrad_impl(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, accumulate, nullptr, row_order, workspace, stream));
    return conv_wgrad_impl(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, accumulate, sumsq, row_order, workspace, stream);
}

extern "C" int shg_conv3d_k533_wgrad_sumsq(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W,
                                           int Cin, int Cout, int c0, int cn, double* sumsq, const void* workspace, void* stream) {
    if (!sumsq || (reinterpret_cast<uintptr_t>(sumsq) & 7)) return fail_arg("conv3d_wgrad_sumsq: sumsq must be an 8-byte aligned device pointer");
    SHG_REPEAT(256, conv_wgrad_impl(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, 0, nullptr, 0, workspace, stream));
    return conv_wgrad_impl(x, dy, dw, dtype, B, T, H, W, Cin, Cout, c0, cn, 0, sumsq, 0, workspace, stream);
}


extern "C" int shg_conv3d_k533_wgrad(const void* x, const void* dy, float* dw, int dtype, int B, int T, int H, int W,
                                     int Cin, int Cout, int accumulate, const void* workspace, void* stream) {
    return shg_conv3d_k533_wgrad_slice(x, dy, dw, dtype, B, T, H, W, Cin, Cout, 0, Cout, accumulate, workspace, stream);
}

extern "C" int shg_conv3d_k533_dgrad(const void* dy_padded, const void* w, void* dx, int dtype, int B, int Tp, int H, int W,
                                     int Cin, int Cout, const void* workspace, void* stream) {
    return shg_conv3d_k533_dgrad_rows(dy_padded, w, dx, dtype, B, Tp, H, W, Cin, Cout, nullptr, 0, workspace, nullptr, stream);
}

extern "C" int shg_conv3d_k533_dgrad_rows(const void* dy_padded, const void* w, void* dx, int dtype, int B, int Tp, int H, int W,
                                          int Cin, int Cout, const int32_t* dx_rows, int row_order, const void* workspace, void* streamk_ws,
                                          void* stream) {
    SHG_REPEAT(512, shg_conv3d_k533_dgrad_rows(dy_padded, w, dx, dtype, B, Tp, H, W, Cin, Cout, dx_rows, row_order, workspace, streamk_ws, stream));
    if (row_order != 0 && row_order != 2) return fail_arg("conv3d_dgrad_rows: row_order must be 0 (standard) or 2 (frame-major)");
    if (streamk_ws && !al16(streamk_ws)) return fail_arg("conv3d_dgrad_rows: streamk workspace must be 16-byte aligned");
    // dx[b,t,h,w,ci] = sum_{tap',co} dYp[b, t+kt', h+kh', w+kw', co] * W[co][44-tap'][ci]; dYp = dy padded by
    // 4 in T and 1 in H/W, so this is the forward gather over dYp with the weight read "contraction strided".
    if (!dy_padded || !w || !dx) return fail_arg("conv3d_dgrad: null pointer");
    if (int e = conv_check(dtype, B, Tp, H, W, Cout, Cin, workspace)) return e;    // the gathered tensor has Cout channels
    if (Cin % 8) return fail_arg("conv3d_dgrad: Cin must be a multiple of 8");
    if (!al16(dy_padded) || !al16(w) || !al16(dx)) return fail_arg("conv3d_dgrad: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)B * (Tp - 4) * H * W, N = Cin, K = (int64_t)45 * Cout;
    const int32_t* pos_in = (const int32_t*)workspace;
    ConvGeom g = conv_geom(Cout, H + 2, W + 2);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) {
        ConvRowSrc<float, 256> sa{(const float*)dy_padded, pos_in, 0, M, g};
        ConvWeightColSrc<float> sb{(const float*)w, 0, N, K, Cin, Cout, (uint32_t)(0x100000000ull / (uint32_t)Cout) + 1u};
        Epilogue<float> ep{(float*)dx, Cin, nullptr, dx_rows, SHG_ACT_NONE, 0, 1, nullptr, 0};
        return launch_cfg<float, float, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, M, N, K, st, "conv3d_k533_dgrad", false);
    }
    ConvWeightColSrc<bf16_t> sb{(const bf16_t*)w, 0, N, K, Cin, Cout, (uint32_t)(0x100000000ull / (uint32_t)Cout) + 1u};
    Epilogue<bf16_t> ep{(bf16_t*)dx, Cin, nullptr, dx_rows, SHG_ACT_NONE, 0, 1, nullptr, 0};
    if (Cout % 64 == 0 && use_gemm8(M, N, K, (int64_t)B * Tp * (H + 2) * (W + 2) * Cout * 2, (int64_t)64 * 45 * Cin * 2)) {
        // frame-major rows (the caller's tables are of row order 2): 40 of the 60 (output frame, kt) pairs of a dy padded by four frames
        // read data; a tile keeps the temporal taps that do for any of its frames, the stream-K launch balances the lengths with the
        // weighted plan ("conv_k_order" bit 6)
        if (row_order == 2 && g.kperm && streamk_ws && (tuning(TUNE_CONV_K_ORDER) & 64) && (int64_t)45 * (Cout / 64) < 4000) {
            g.rpt = B * H * W;
            g.tv_lo = 4;
            g.tv_hi = Tp - 4;
            ConvRowSrc<bf16_t, 512, true> sa{(const bf16_t*)dy_padded, pos_in, 0, M, g};
            return launch8<bf16_t, decltype(sa), decltype(sb), 1>(sa, sb, ep, M, N, K, st, "conv3d_k533_dgrad", 1, streamk_ws);
        }
        ConvRowSrc<bf16_t, 512> sa{(const bf16_t*)dy_padded, pos_in, 0, M, g};
        return launch8<bf16_t, decltype(sa), decltype(sb)>(sa, sb, ep, M, N, K, st, "conv3d_k533_dgrad");
    }
    if (use_large(1, M, N, K)) {
        ConvRowSrc<bf16_t, 512> sa{(const bf16_t*)dy_padded, pos_in, 0, M, g};
        return launch_cfg<bf16_t, bf16_t, decltype(sa), decltype(sb), 4, 4, 2, 4>(sa, sb, ep, M, N, K, st, "conv3d_k533_dgrad", false);
    }
    ConvRowSrc<bf16_t, 256> sa{(const bf16_t*)dy_padded, pos_in, 0, M, g};
    return launch_cfg<bf16_t, bf16_t, decltype(sa), decltype(sb), 2, 2, 2, 2>(sa, sb, ep, M, N, K, st, "conv3d_k533_dgrad", false);
}

extern "C" int shg_ncdhw_to_padded_cl(const float* x, void* y, int dtype, int B, int C, int T, int H, int W, void* stream) {
    if (!x || !y) return fail_arg("ncdhw_to_padded_cl: null pointer");
    if (B < 1 || C < 1 || T < 1 || H < 1 || W < 1 || (int64_t)B * T > 65535) return fail_arg("ncdhw_to_padded_cl: bad sizes");
    dim3 grid((H * W + 63) / 64, (C + 63) / 64, B * T), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(ncdhw_to_padded_cl_kernel<float>, grid, block, 0, st, x, (float*)y, B, C, T, H, W);
    else if (dtype == SHG_BF16) hipLaunchKernelGGL(ncdhw_to_padded_cl_kernel<bf16_t>, grid, block, 0, st, x, (bf16_t*)y, B, C, T, H, W);
    else return fail_arg("ncdhw_to_padded_cl: bad dtype");
    return check_launch("ncdhw_to_padded_cl");
}

// Fused epilogue kernels: bias + activation + dropout + residual + LayerNorm (fwd/bwd), bias +
// activation + dropout (fwd/bwd) and the deterministic second stage of their column reductions.
// All of them are HBM-bound streaming kernels: one wave per row, 16-byte vector accesses, fp32
// statistics via wave shuffles, no LDS in the forward path.
#include <math.h>

#include "common.h"

namespace shg {

constexpr int LN_MAX_CHUNKS = 8;        // 16-byte chunks per lane held in registers
constexpr int ROWS_PER_PARTIAL = 16;  // rows folded into one partial row of the column sums
constexpr int MAX_PARTIALS = 4096;

__host__ inline int colsum_partials(int64_t rows) {
    int64_t n = (rows + ROWS_PER_PARTIAL - 1) / ROWS_PER_PARTIAL;
    if (n < 1) n = 1;
    if (n > MAX_PARTIALS) n = MAX_PARTIALS;
    return (int)n;
}

// V consecutive fp32 parameters (bias / gamma / beta slice of this lane's chunk) as 16-byte loads.  Per-element `p ? p[c] : 0`
// inside the element loop compiled to one dependent 4-byte load + wait per element (8-16 serial L2 round trips per row: the
// r-layer LayerNorm ran at 30 us where an add of the same traffic takes 10).  p == nullptr: zeros (wave-uniform branch).
template <int V> __device__ __forceinline__ void load_param_vec(const float* __restrict__ p, int c0, float (&out)[V]) {
    if (p) {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(p + c0 + 4 * q);
            out[4 * q] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) out[j] = 0.f;
    }
}

// FAST: bf16 storage (gelu_fast, common.h); the fp32 parity mode keeps libm's erff
template <int ACT, bool FAST> __device__ __forceinline__ float act_fwd(float u) {
    if (ACT == SHG_ACT_GELU) return FAST ? gelu_fast(u) : gelu_erf(u);
    if (ACT == SHG_ACT_RELU) return fmaxf(u, 0.f);
    return u;
}
template <int ACT, bool FAST> __device__ __forceinline__ float act_grad(float u) {
    if (ACT == SHG_ACT_GELU) return FAST ? gelu_fast_grad(u) : gelu_erf_grad(u);
    if (ACT == SHG_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    return 1.f;
}

// ------------------------------------------------------------------------------------------------
// forward: one wave per row
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT, int NCH, bool DROP, int VB = 16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                     const T* __restrict__ residual, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     T* __restrict__ z_out, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, const T* __restrict__ pos,
                                                     T* __restrict__ y_pos, int64_t rows, int cols, float eps,
                                                     uint32_t drop_thr, float drop_scale,
                                                     const uint64_t* __restrict__ seed_state, uint64_t stream_id) {
    using VT = VecB<T, VB>;
    constexpr int V = VT::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / V;
    const uint64_t seed = DROP ? dropout_seed(seed_state, stream_id) : 0;
    float zv[NCH][V], gv[NCH][V], bev[NCH][V];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            const int c0 = ch * V;
            const int64_t off = row * cols + c0;
            VT xv = loadv<VB>(x + off);
            VT rv;
            if (residual) rv = loadv<VB>(residual + off);
            float bv[V];
            load_param_vec<V>(bias, c0, bv);
            load_param_vec<V>(gamma, c0, gv[i]);         // (used after the two reductions: in flight meanwhile)
            load_param_vec<V>(beta, c0, bev[i]);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float u = xv.get(j) + bv[j];
                u = act_fwd<ACT, (sizeof(T) == 2)>(u);
                if (DROP) u = dropout_keep_run(seed, (uint64_t)off, j, drop_thr) ? u * drop_scale : 0.f;
                if (residual) u += rv.get(j);
                zv[i][j] = u;
                sum += u;
            }
            if (z_out) {
                VT zo;
#pragma unroll
                for (int j = 0; j < V; ++j) zo.set(j, zv[i][j]);
                storev(z_out + off, zo);
            }
        }
    }
    // statistics are taken on the values as they are stored (rounded to T) so that the backward
    // pass, which re-reads z_out, sees exactly the same normalised activations
    if (z_out && sizeof(T) == 2) {
        sum = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + i * 64 < nchunk)
#pragma unroll
                for (int j = 0; j < V; ++j) { zv[i][j] = to_f32(from_f32<T>(zv[i][j])); sum += zv[i][j]; }
    }
    const float mean = wave_sum_dpp(sum) / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + i * 64 < nchunk)
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = zv[i][j] - mean; sq += d * d; }
    const float var = wave_sum_dpp(sq) / (float)cols;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk) {
            const int c0 = ch * V;
            VT yo;
#pragma unroll
            for (int j = 0; j < V; ++j) yo.set(j, (zv[i][j] - mean) * rstd * gv[i][j] + bev[i][j]);
            storev(y + row * cols + c0, yo);
            if (y_pos) {                 // second output: y (as stored) + pos, rounded once (the decoder's `tgt + query_pos`)
                const VT pv = loadv<VB>(pos + row * cols + c0);
                VT po;
#pragma unroll
                for (int j = 0; j < V; ++j) po.set(j, yo.get(j) + pv.get(j));
                storev(y_pos + row * cols + c0, po);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: block = 4 waves, loops over its chunk of rows; per-lane column partials in registers,
// reduced across the 4 waves through LDS, one partial row per block.
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT, int NCH, bool DROP, int VB = 16>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                     const T* __restrict__ x, const float* __restrict__ bias,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     T* __restrict__ dres, float* __restrict__ dgamma_p,
                                                     float* __restrict__ dbeta_p, float* __restrict__ dbias_p,
                                                     int64_t rows, int cols, uint32_t drop_thr, float drop_scale,
                                                     const uint64_t* __restrict__ seed_state, uint64_t stream_id) {
    using VT = VecB<T, VB>;
    constexpr int V = VT::N;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [3][4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = cols / V;
    const int nblk = gridDim.x;
    const int64_t rows_per_blk = (rows + nblk - 1) / nblk;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r_end = min(rows, r_begin + rows_per_blk);
    const uint64_t seed = DROP ? dropout_seed(seed_state, stream_id) : 0;

    float ag[NCH][V], ab[NCH][V], abias[NCH][V];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; abias[i][j] = 0.f; }

    for (int64_t row = r_begin + wave; row < r_end; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        float gyv[NCH][V], xh[NCH][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const int c0 = ch * V;
                const int64_t off = row * cols + c0;
                VT dv = loadv<VB>(dy + off), zz = loadv<VB>(z + off);
                float gm[V];
                load_param_vec<V>(gamma, c0, gm);        // (L1 / L2 hit; kept out of the registers that live across rows)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float g = dv.get(j);
                    const float h = (zz.get(j) - mu) * rs;
                    const float gy = g * gm[j];
                    xh[i][j] = h; gyv[i][j] = gy;
                    s1 += gy; s2 += gy * h;
                    ag[i][j] += g * h; ab[i][j] += g;
                }
            }
        }
        s1 = wave_sum_dpp(s1) / (float)cols;
        s2 = wave_sum_dpp(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const int c0 = ch * V;
                const int64_t off = row * cols + c0;
                VT dzv, dxv, xv;
                float bsv[V];
                if (ACT != SHG_ACT_NONE) { xv = loadv<VB>(x + off); load_param_vec<V>(bias, c0, bsv); }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float dz = rs * (gyv[i][j] - s1 - xh[i][j] * s2);
                    dzv.set(j, dz);
                    float d = dz;
                    if (DROP) d = dropout_keep_run(seed, (uint64_t)off, j, drop_thr) ? d * drop_scale : 0.f;
                    if (ACT != SHG_ACT_NONE) d *= act_grad<ACT, (sizeof(T) == 2)>(xv.get(j) + bsv[j]);
                    dxv.set(j, d);
                    abias[i][j] += to_f32(from_f32<T>(d));
                }
                if (dres) storev(dres + off, dzv);
                if (dx) storev(dx + off, dxv);
            }
        }
    }
    // cross-wave reduction of the column partials
    float* rg = red;
    float* rb = red + 4 * cols;
    float* rbi = red + 8 * cols;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                rg[wave * cols + ch * V + j] = ag[i][j];
                rb[wave * cols + ch * V + j] = ab[i][j];
                rbi[wave * cols + ch * V + j] = abias[i][j];
            }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        const int64_t o = (int64_t)blockIdx.x * cols + c;
        if (dgamma_p) dgamma_p[o] = rg[c] + rg[cols + c] + rg[2 * cols + c] + rg[3 * cols + c];
        if (dbeta_p) dbeta_p[o] = rb[c] + rb[cols + c] + rb[2 * cols + c] + rb[3 * cols + c];
        if (dbias_p) dbias_p[o] = rbi[c] + rbi[cols + c] + rbi[2 * cols + c] + rbi[3 * cols + c];
    }
}

// ------------------------------------------------------------------------------------------------
// bias + activation + dropout
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT, bool DROP>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                           T* __restrict__ y, int64_t n_vec, int cols,
                                                           uint32_t drop_thr, float drop_scale,
                                                           const uint64_t* __restrict__ seed_state, uint64_t stream_id) {
    constexpr int V = Vec16<T>::N;
    const uint64_t seed = DROP ? dropout_seed(seed_state, stream_id) : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t off = i * V;
        const int c0 = (int)(off % cols);
        Vec16<T> xv = load16(x + off), yv;
        float bv[V];
        load_param_vec<V>(bias, c0, bv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float u = act_fwd<ACT, (sizeof(T) == 2)>(xv.get(j) + bv[j]);
            if (DROP) u = dropout_keep_run(seed, (uint64_t)off, j, drop_thr) ? u * drop_scale : 0.f;
            yv.set(j, u);
        }
        store16(y + off, yv);
    }
}

// Optional views of bias_act_bwd (the conv stack's backward, ops._VisualConvTokens): the incoming gradient read from every
// group of `dy_group_stride` rows skipping the first `dy_row_offset` (the token gradients [B, 1 + 392, C] without the cls
// rows - no contiguous copy), and a second copy of the result scattered by a row table into a zero-bordered buffer (the padded
// layout the input-gradient convolution gathers from - no F.pad pass).
struct BwdView {
    int64_t dy_rows_per_group, dy_group_stride, dy_row_offset;      // rows_per_group == 0: dy is [rows, cols] as it is
    void* dx2;                                                      // second output (same dtype) or null
    const int32_t* row2;                                            // its row of result row r (null: r)
    const int32_t* xrow;                                            // row of x AND of dx that belongs to row r (null: r)
};
template <typename T, int ACT, int NCH, bool DROP>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                           const T* __restrict__ dy, T* __restrict__ dx,
                                                           float* __restrict__ dbias_p, int64_t rows, int cols,
                                                           uint32_t drop_thr, float drop_scale,
                                                           const uint64_t* __restrict__ seed_state, uint64_t stream_id, BwdView vw) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = cols / V;
    const int nblk = gridDim.x;
    const int64_t rows_per_blk = (rows + nblk - 1) / nblk;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r_end = min(rows, r_begin + rows_per_blk);
    const uint64_t seed = DROP ? dropout_seed(seed_state, stream_id) : 0;
    float ab[NCH][V], bsv[NCH][V];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
        for (int j = 0; j < V; ++j) { ab[i][j] = 0.f; bsv[i][j] = 0.f; }
        if (lane + i * 64 < nchunk) load_param_vec<V>(bias, (lane + i * 64) * V, bsv[i]);
    }
    for (int64_t row = r_begin + wave; row < r_end; row += 4) {
        const int64_t dy_row = vw.dy_rows_per_group > 0
                                   ? (row / vw.dy_rows_per_group) * vw.dy_group_stride + vw.dy_row_offset + row % vw.dy_rows_per_group
                                   : row;
        const int64_t row2 = vw.row2 ? (int64_t)vw.row2[row] : row;
        const int64_t xr = vw.xrow ? (int64_t)vw.xrow[row] : row;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const int c0 = ch * V;
                const int64_t off = xr * cols + c0;
                Vec16<T> xv = load16(x + off), gv = load16(dy + dy_row * cols + c0), dv;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float d = gv.get(j);
                    if (DROP) d = dropout_keep_run(seed, (uint64_t)off, j, drop_thr) ? d * drop_scale : 0.f;
                    d *= act_grad<ACT, (sizeof(T) == 2)>(xv.get(j) + bsv[i][j]);
                    dv.set(j, d);
                    ab[i][j] += to_f32(from_f32<T>(d));
                }
                store16(dx + off, dv);
                if (vw.dx2) store16(reinterpret_cast<T*>(vw.dx2) + row2 * cols + c0, dv);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk)
#pragma unroll
            for (int j = 0; j < V; ++j) red[wave * cols + ch * V + j] = ab[i][j];
    }
    __syncthreads();
    if (dbias_p)
        for (int c = threadIdx.x; c < cols; c += 256)
            dbias_p[(int64_t)blockIdx.x * cols + c] = red[c] + red[cols + c] + red[2 * cols + c] + red[3 * cols + c];
}

// column sums of a [rows, cols] activation (bias gradient of a GEMM whose bias sits in its epilogue)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                             int64_t rows, int cols, int64_t ld) {
    const int nblk = gridDim.y;
    const int64_t rows_per_blk = (rows + nblk - 1) / nblk;
    const int64_t r_begin = (int64_t)blockIdx.y * rows_per_blk;
    const int64_t r_end = min(rows, r_begin + rows_per_blk);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float acc = 0.f;
    for (int64_t r = r_begin; r < r_end; ++r) acc += to_f32(x[r * ld + c]);
    partial[(int64_t)blockIdx.y * cols + c] = acc;
}

// column sums straight into the gradient vector: one wave per row, 16-byte loads, four rows in flight per wave; the four
// waves' partial sums meet in LDS and leave as 64-lane atomics (consecutive lanes = consecutive columns)
constexpr int COLSUM_ROWS_PER_BLOCK = 64;
template <typename T, int NCH>
__global__ __launch_bounds__(256) void colsum_atomic_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                            int cols, int64_t ld) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = cols / V;
    const int64_t r_begin = (int64_t)blockIdx.x * COLSUM_ROWS_PER_BLOCK;
    const int64_t r_end = min(rows, r_begin + COLSUM_ROWS_PER_BLOCK);
    float acc[NCH][V];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[i][j] = 0.f;
#pragma unroll 4
    for (int64_t row = r_begin + wave; row < r_end; row += 4) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                const Vec16<T> v = load16(x + row * ld + ch * V);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[i][j] += v.get(j);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + i * 64;
        if (ch < nchunk)
#pragma unroll
            for (int j = 0; j < V; ++j) red[wave * cols + ch * V + j] = acc[i][j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) atomicAdd(out + c, red[c] + red[cols + c] + red[2 * cols + c] + red[3 * cols + c]);
}

// second stage of the column reductions: block = 64 columns x 16 partial-row groups; the partial rows
// are cut into gridDim.y slices whose sums are combined with fp32 atomics (<= 8 adders per address)
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, int n_partials, int cols,
                                                             float* __restrict__ out) {
    __shared__ float sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int per = (n_partials + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(n_partials, p0 + per);
    float s = 0.f;
    if (c < cols)
        for (int p = p0 + ty; p < p1; p += 16) s += partial[(int64_t)p * cols + c];
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols && p0 < p1) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][tx];
        atomicAdd(out + c, t);
    }
}

// up to four reductions of equally shaped partial buffers in one launch (blockIdx.z picks the pair):
// the gamma / beta / bias gradients of one LayerNorm backward
struct FinishSet {
    const float* partial[4];
    float* out[4];
};
__global__ __launch_bounds__(1024) void colsum_finish_multi_kernel(FinishSet fs, int n_partials, int cols) {
    __shared__ float sh[16][64];
    const float* __restrict__ partial = fs.partial[blockIdx.z];
    float* __restrict__ out = fs.out[blockIdx.z];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int per = (n_partials + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(n_partials, p0 + per);
    float s = 0.f;
    if (c < cols)
        for (int p = p0 + ty; p < p1; p += 16) s += partial[(int64_t)p * cols + c];
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols && p0 < p1) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][tx];
        atomicAdd(out + c, t);
    }
}

template <typename T> static bool aligned16(const T* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
static int launch_ln_fwd(const void* x, const float* bias, const void* residual, const float* gamma, const float* beta,
                         void* y, void* z_out, float* mean, float* rstd, const void* pos, void* y_pos, int64_t rows, int cols, int act, float eps,
                         float p_drop, const uint64_t* seed_state, uint64_t stream_id, hipStream_t st) {
    const uint32_t thr = dropout_threshold(p_drop);
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nch = (cols / Vec16<T>::N + 63) / 64;
    // rows whose sixteen-byte chunks do not fill the lanes of the last pass (768 bf16 values: 96 chunks = 1.5 per lane) but whose
    // eight-byte chunks do (192 = 3 per lane): the half-width form keeps every lane busy and a third fewer values per lane in registers
    const bool half = sizeof(T) == 2 && (cols / 8) % 64 != 0 && cols % 256 == 0 && cols / 256 == 3 && tuning(TUNE_LN_HALF_VEC);
    if (!aligned16(gamma) || !aligned16(beta) || (bias && !aligned16(bias))) return fail_arg("bias_act_drop_res_ln_fwd: bias / gamma / beta must be 16-byte aligned");
#define LN_FWD4(ACT, NCH, DROP, VBYTES)                                                                               \
    hipLaunchKernelGGL((ln_fwd_kernel<T, ACT, NCH, DROP, VBYTES>), grid, block, 0, st, (const T*)x, bias, (const T*)residual, gamma, \
                       beta, (T*)y, (T*)z_out, mean, rstd, (const T*)pos, (T*)y_pos, rows, cols, eps, thr, scale, seed_state, stream_id)
#define LN_FWD3(ACT, NCH, DROP) LN_FWD4(ACT, NCH, DROP, 16)
#define LN_FWD2(ACT, NCH) do { if (thr) LN_FWD3(ACT, NCH, true); else LN_FWD3(ACT, NCH, false); } while (0)
#define LN_FWD(ACT) do { if constexpr (sizeof(T) == 2) { if (half) { if (thr) LN_FWD4(ACT, 3, true, 8); else LN_FWD4(ACT, 3, false, 8); break; } } \
                         if (nch <= 2) LN_FWD2(ACT, 2); else if (nch == 3) LN_FWD2(ACT, 3); else if (nch <= 4) LN_FWD2(ACT, 4); else LN_FWD2(ACT, 8); } while (0)
    if (act == SHG_ACT_NONE) LN_FWD(SHG_ACT_NONE);
    else if (act == SHG_ACT_GELU) LN_FWD(SHG_ACT_GELU);
    else LN_FWD(SHG_ACT_RELU);
#undef LN_FWD
#undef LN_FWD2
#undef LN_FWD3
#undef LN_FWD4
    return check_launch("bias_act_drop_res_ln_fwd");
}

template <typename T>
static int launch_ln_bwd(const void* dy, const void* z, const void* x, const float* bias, const float* gamma,
                         const float* mean, const float* rstd, void* dx, void* dres, float* dg, float* db, float* dbi,
                         int n_partials, int64_t rows, int cols, int act, float p_drop, const uint64_t* seed_state,
                         uint64_t stream_id, hipStream_t st) {
    const uint32_t thr = dropout_threshold(p_drop);
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    dim3 grid(n_partials), block(256);
    const size_t lds = (size_t)12 * cols * sizeof(float);
    const int nch = (cols / Vec16<T>::N + 63) / 64;
    const bool half = sizeof(T) == 2 && (cols / 8) % 64 != 0 && cols % 256 == 0 && cols / 256 == 3 && tuning(TUNE_LN_HALF_VEC);   // (launch_ln_fwd)
    if (!aligned16(gamma) || (bias && !aligned16(bias))) return fail_arg("bias_act_drop_res_ln_bwd: bias / gamma must be 16-byte aligned");
#define LN_BWD4(ACT, NCH, DROP, VBYTES)                                                                              \
    hipLaunchKernelGGL((ln_bwd_kernel<T, ACT, NCH, DROP, VBYTES>), grid, block, lds, st, (const T*)dy, (const T*)z, (const T*)x, bias, \
                       gamma, mean, rstd, (T*)dx, (T*)dres, dg, db, dbi, rows, cols, thr, scale, seed_state, stream_id)
#define LN_BWD3(ACT, NCH, DROP) LN_BWD4(ACT, NCH, DROP, 16)
#define LN_BWD2(ACT, NCH) do { if (thr) LN_BWD3(ACT, NCH, true); else LN_BWD3(ACT, NCH, false); } while (0)
#define LN_BWD(ACT) do { if constexpr (sizeof(T) == 2) { if (half) { if (thr) LN_BWD4(ACT, 3, true, 8); else LN_BWD4(ACT, 3, false, 8); break; } } \
                         if (nch <= 2) LN_BWD2(ACT, 2); else if (nch == 3) LN_BWD2(ACT, 3); else if (nch <= 4) LN_BWD2(ACT, 4); else LN_BWD2(ACT, 8); } while (0)
    if (act == SHG_ACT_NONE) LN_BWD(SHG_ACT_NONE);
    else if (act == SHG_ACT_GELU) LN_BWD(SHG_ACT_GELU);
    else LN_BWD(SHG_ACT_RELU);
#undef LN_BWD
#undef LN_BWD2
#undef LN_BWD3
#undef LN_BWD4
    return check_launch("bias_act_drop_res_ln_bwd");
}

static int check_cols(int dtype, int cols, const char* who, int max_chunks = LN_MAX_CHUNKS) {
    const int v = dtype == SHG_BF16 ? 8 : 4;
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("bad dtype");
    if (cols <= 0 || cols % v != 0 || cols > 64 * max_chunks * v) return fail_arg(who);
    return 0;
}

}  // namespace shg

using namespace shg;

extern "C" int shg_colsum_partials(int64_t rows) { return colsum_partials(rows); }

extern "C" int shg_colsum_partial(const void* x, int dtype, int64_t rows, int cols, int64_t ld, float* partial,
                                  int n_partials, void* stream) {
    if (!x || !partial || rows <= 0 || cols <= 0 || ld < cols) return fail_arg("colsum_partial: bad argument");
    if (n_partials < 1 || n_partials > MAX_PARTIALS) return fail_arg("colsum_partial: bad n_partials");
    dim3 grid((cols + 255) / 256, n_partials), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, block, 0, st, (const float*)x, partial, rows, cols, ld);
    else if (dtype == SHG_BF16) hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, partial, rows, cols, ld);
    else return fail_arg("colsum_partial: bad dtype");
    return check_launch("colsum_partial");
}

template <typename T>
static int colsum_accumulate_t(const T* x, int64_t rows, int cols, int64_t ld, float* out, hipStream_t st) {
    constexpr int V = Vec16<T>::N;
    if (cols % V || ld % V || (reinterpret_cast<uintptr_t>(x) & 15)) return fail_arg("colsum_accumulate: cols / ld / alignment must allow 16-byte loads");
    const int nch = (cols / V + 63) / 64;
    const dim3 grid((unsigned)((rows + COLSUM_ROWS_PER_BLOCK - 1) / COLSUM_ROWS_PER_BLOCK)), block(256);
    const size_t lds = (size_t)4 * cols * sizeof(float);
    switch (nch) {
        case 1: hipLaunchKernelGGL((colsum_atomic_kernel<T, 1>), grid, block, lds, st, x, out, rows, cols, ld); break;
        case 2: hipLaunchKernelGGL((colsum_atomic_kernel<T, 2>), grid, block, lds, st, x, out, rows, cols, ld); break;
        case 3: case 4: hipLaunchKernelGGL((colsum_atomic_kernel<T, 4>), grid, block, lds, st, x, out, rows, cols, ld); break;
        case 5: case 6: case 7: case 8: hipLaunchKernelGGL((colsum_atomic_kernel<T, 8>), grid, block, lds, st, x, out, rows, cols, ld); break;
        default: return fail_arg("colsum_accumulate: cols too large");
    }
    return check_launch("colsum_accumulate");
}

extern "C" int shg_colsum_accumulate(const void* x, int dtype, int64_t rows, int cols, int64_t ld, float* out, void* stream) {
    SHG_REPEAT(4096, shg_colsum_accumulate(x, dtype, rows, cols, ld, out, stream));
    if (!x || !out || rows <= 0 || cols <= 0 || cols > 4096 || ld < cols) return fail_arg("colsum_accumulate: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) return colsum_accumulate_t<float>((const float*)x, rows, cols, ld, out, st);
    if (dtype == SHG_BF16) return colsum_accumulate_t<bf16_t>((const bf16_t*)x, rows, cols, ld, out, st);
    return fail_arg("colsum_accumulate: bad dtype");
}

extern "C" int shg_colsum_finish(const float* partial, int n_partials, int cols, float* out, int accumulate, void* stream) {
    if (!partial || !out || n_partials < 1 || cols < 1) return fail_arg("colsum_finish: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), st);
    const int slices = n_partials >= 256 ? 8 : (n_partials >= 64 ? 4 : 1);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 63) / 64, slices), dim3(1024), 0, st, partial, n_partials, cols, out);
    return check_launch("colsum_finish");
}

extern "C" int shg_colsum_finish_multi(const float* const* partials, float* const* outs, int count, int n_partials, int cols,
                                       void* stream) {
    if (!partials || !outs || count < 1 || count > 4 || n_partials < 1 || cols < 1) return fail_arg("colsum_finish_multi: bad argument");
    FinishSet fs{};
    for (int i = 0; i < count; ++i) {
        if (!partials[i] || !outs[i]) return fail_arg("colsum_finish_multi: null pointer");
        fs.partial[i] = partials[i];
        fs.out[i] = outs[i];
    }
    const int slices = n_partials >= 256 ? 8 : (n_partials >= 64 ? 4 : 1);
    hipLaunchKernelGGL(colsum_finish_multi_kernel, dim3((cols + 63) / 64, slices, count), dim3(1024), 0, (hipStream_t)stream, fs,
                       n_partials, cols);
    return check_launch("colsum_finish_multi");
}

extern "C" int shg_bias_act_drop_res_ln_fwd_pos(const void* x, const float* bias, const void* residual, const float* gamma,
                                                const float* beta, void* y, void* z_out, float* mean, float* rstd,
                                                const void* pos, void* y_pos, int dtype, int64_t rows, int cols, int act,
                                                float eps, float p_drop, const uint64_t* seed_state, uint64_t stream_id,
                                                void* stream) {
    SHG_REPEAT(4, shg_bias_act_drop_res_ln_fwd_pos(x, bias, residual, gamma, beta, y, z_out, mean, rstd, pos, y_pos, dtype, rows, cols, act,
                                                   eps, p_drop, seed_state, stream_id, stream));
    if (!x || !gamma || !beta || !y || !mean || !rstd) return fail_arg("ln_fwd: null pointer");
    if ((pos == nullptr) != (y_pos == nullptr)) return fail_arg("ln_fwd: pos and y_pos go together");
    if (int e = check_cols(dtype, cols, "ln_fwd: cols must be a multiple of the 16-byte vector and <= 2048 (f32) / 4096 (bf16)")) return e;
    if (!aligned16(x) || !aligned16(y) || (residual && !aligned16(residual)) || (z_out && !aligned16(z_out)) ||
        (pos && (!aligned16(pos) || !aligned16(y_pos))))
        return fail_arg("ln_fwd: pointers must be 16-byte aligned");
    if (act < 0 || act > 2 || p_drop < 0.f || p_drop >= 1.f) return fail_arg("ln_fwd: bad act/p_drop");
    if (p_drop > 0.f && !seed_state) return fail_arg("ln_fwd: dropout needs seed_state");
    if (rows <= 0) return rows == 0 ? 0 : fail_arg("ln_fwd: negative rows");
    hipStream_t st = (hipStream_t)stream;
    return dtype == SHG_F32
               ? launch_ln_fwd<float>(x, bias, residual, gamma, beta, y, z_out, mean, rstd, pos, y_pos, rows, cols, act, eps, p_drop, seed_state, stream_id, st)
               : launch_ln_fwd<bf16_t>(x, bias, residual, gamma, beta, y, z_out, mean, rstd, pos, y_pos, rows, cols, act, eps, p_drop, seed_state, stream_id, st);
}

extern "C" int shg_bias_act_drop_res_ln_fwd(const void* x, const float* bias, const void* residual, const float* gamma,
                                            const float* beta, void* y, void* z_out, float* mean, float* rstd,
                                            int dtype, int64_t rows, int cols, int act, float eps, float p_drop,
                                            const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    return shg_bias_act_drop_res_ln_fwd_pos(x, bias, residual, gamma, beta, y, z_out, mean, rstd, nullptr, nullptr, dtype, rows,
                                            cols, act, eps, p_drop, seed_state, stream_id, stream);
}

// ------------------------------------------------------------------------------------------------
// elementwise helpers of the decoder executor
// ------------------------------------------------------------------------------------------------
namespace shg {
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n_vec) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const Vec16<T> av = load16(a + i * V), bv = load16(b + i * V);
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.set(j, av.get(j) + bv.get(j));
        store16(out + i * V, o);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void add2_kernel(T* __restrict__ acc1, T* __restrict__ acc2, const T* __restrict__ c, int init2,
                                                   int64_t n_vec) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const Vec16<T> cv = load16(c + i * V);
        if (acc1) {
            const Vec16<T> a = load16(acc1 + i * V);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < V; ++j) o.set(j, a.get(j) + cv.get(j));
            store16(acc1 + i * V, o);
        }
        if (!acc2) continue;
        if (init2) {
            store16(acc2 + i * V, cv);
        } else {
            const Vec16<T> a = load16(acc2 + i * V);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < V; ++j) o.set(j, a.get(j) + cv.get(j));
            store16(acc2 + i * V, o);
        }
    }
}
}  // namespace shg

namespace shg {
// out[b, 0, :] = cls + pos[0];  out[b, 1 + t, :] = tok[b, t, :] + pos[1 + t]   (fp32 sums, rounded once)
template <typename T>
__global__ __launch_bounds__(256) void tokens_assemble_kernel(const T* __restrict__ tok, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, T* __restrict__ out, int B, int n_tok, int C) {
    constexpr int V = Vec16<T>::N;
    const int cv = C / V;
    const int64_t n_vec = (int64_t)B * n_tok * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * V;
        const int64_t row = i / cv;
        const int t = (int)(row % n_tok);
        const int64_t b = row / n_tok;
        Vec16<T> o;
        if (t == 0) {
#pragma unroll
            for (int j = 0; j < V; ++j) o.set(j, cls[c0 + j] + pos[c0 + j]);
        } else {
            const Vec16<T> x = load16(tok + (b * (n_tok - 1) + (t - 1)) * C + c0);
#pragma unroll
            for (int j = 0; j < V; ++j) o.set(j, x.get(j) + pos[(int64_t)t * C + c0 + j]);
        }
        store16(out + row * C + c0, o);
    }
}
}  // namespace shg

extern "C" int shg_tokens_assemble(const void* tok, const float* cls, const float* pos, void* out, int dtype, int B, int n_tok,
                                   int C, void* stream) {
    if (!tok || !cls || !pos || !out) return fail_arg("tokens_assemble: null pointer");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("tokens_assemble: bad dtype");
    const int V = dtype == SHG_BF16 ? 8 : 4;
    if (B < 1 || n_tok < 2 || C < V || C % V) return fail_arg("tokens_assemble: bad sizes");
    if (!aligned16(tok) || !aligned16(out)) return fail_arg("tokens_assemble: pointers must be 16-byte aligned");
    const int64_t n_vec = (int64_t)B * n_tok * (C / V);
    dim3 grid((unsigned)std::min<int64_t>((n_vec + 255) / 256, 4096)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(tokens_assemble_kernel<float>, grid, block, 0, st, (const float*)tok, cls, pos, (float*)out, B, n_tok, C);
    else hipLaunchKernelGGL(tokens_assemble_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)tok, cls, pos, (bf16_t*)out, B, n_tok, C);
    return check_launch("tokens_assemble");
}

extern "C" int shg_add(const void* a, const void* b, void* out, int dtype, int64_t n, void* stream) {
    if (!a || !b || !out) return fail_arg("add: null pointer");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("add: bad dtype");
    const int V = dtype == SHG_BF16 ? 8 : 4;
    if (n <= 0 || n % V) return n == 0 ? 0 : fail_arg("add: n must be a positive multiple of the 16-byte vector");
    if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return fail_arg("add: pointers must be 16-byte aligned");
    const int64_t n_vec = n / V;
    dim3 grid((unsigned)std::min<int64_t>((n_vec + 255) / 256, 4096)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(add_kernel<float>, grid, block, 0, st, (const float*)a, (const float*)b, (float*)out, n_vec);
    else hipLaunchKernelGGL(add_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n_vec);
    return check_launch("add");
}

extern "C" int shg_add2_accumulate(void* acc1, void* acc2, const void* c, int init2, int dtype, int64_t n, void* stream) {
    if ((!acc1 && !acc2) || !c) return fail_arg("add2: null pointer");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("add2: bad dtype");
    const int V = dtype == SHG_BF16 ? 8 : 4;
    if (n <= 0 || n % V) return n == 0 ? 0 : fail_arg("add2: n must be a positive multiple of the 16-byte vector");
    if ((acc1 && !aligned16(acc1)) || (acc2 && !aligned16(acc2)) || !aligned16(c)) return fail_arg("add2: pointers must be 16-byte aligned");
    const int64_t n_vec = n / V;
    dim3 grid((unsigned)std::min<int64_t>((n_vec + 255) / 256, 4096)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(add2_kernel<float>, grid, block, 0, st, (float*)acc1, (float*)acc2, (const float*)c, init2, n_vec);
    else hipLaunchKernelGGL(add2_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)acc1, (bf16_t*)acc2, (const bf16_t*)c, init2, n_vec);
    return check_launch("add2");
}

extern "C" int shg_bias_act_drop_res_ln_bwd(const void* dy, const void* z, const void* x, const float* bias,
                                            const float* gamma, const float* mean, const float* rstd, void* dx,
                                            void* dres, float* dgamma_partial, float* dbeta_partial,
                                            float* dbias_partial, int n_partials, int dtype, int64_t rows, int cols,
                                            int act, float p_drop, const uint64_t* seed_state, uint64_t stream_id,
                                            void* stream) {
    SHG_REPEAT(8, shg_bias_act_drop_res_ln_bwd(dy, z, x, bias, gamma, mean, rstd, dx, dres, dgamma_partial, dbeta_partial, dbias_partial,
                                               n_partials, dtype, rows, cols, act, p_drop, seed_state, stream_id, stream));
    if (!dy || !z || !gamma || !mean || !rstd) return fail_arg("ln_bwd: null pointer");
    if (act != SHG_ACT_NONE && !x) return fail_arg("ln_bwd: x is required when act != NONE");
    if (int e = check_cols(dtype, cols, "ln_bwd: unsupported cols")) return e;
    if (n_partials < 1 || n_partials > MAX_PARTIALS) return fail_arg("ln_bwd: bad n_partials");
    if ((size_t)12 * cols * sizeof(float) > 160 * 1024) return fail_arg("ln_bwd: cols too large for LDS");
    if (rows <= 0) return rows == 0 ? 0 : fail_arg("ln_bwd: negative rows");
    hipStream_t st = (hipStream_t)stream;
    return dtype == SHG_F32
               ? launch_ln_bwd<float>(dy, z, x, bias, gamma, mean, rstd, dx, dres, dgamma_partial, dbeta_partial, dbias_partial, n_partials, rows, cols, act, p_drop, seed_state, stream_id, st)
               : launch_ln_bwd<bf16_t>(dy, z, x, bias, gamma, mean, rstd, dx, dres, dgamma_partial, dbeta_partial, dbias_partial, n_partials, rows, cols, act, p_drop, seed_state, stream_id, st);
}

extern "C" int shg_bias_act_fwd(const void* x, const float* bias, void* y, int dtype, int64_t rows, int cols, int act,
                                float p_drop, const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    if (!x || !y) return fail_arg("bias_act_fwd: null pointer");
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("bias_act_fwd: bad dtype");
    const int V = dtype == SHG_BF16 ? 8 : 4;
    if (cols <= 0 || cols % V) return fail_arg("bias_act_fwd: cols must be a multiple of the 16-byte vector");
    if (act < 0 || act > 2 || p_drop < 0.f || p_drop >= 1.f) return fail_arg("bias_act_fwd: bad act/p_drop");
    if (rows <= 0) return rows == 0 ? 0 : fail_arg("bias_act_fwd: negative rows");
    const int64_t n_vec = rows * cols / V;
    const uint32_t thr = dropout_threshold(p_drop);
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    dim3 grid((unsigned)std::min<int64_t>((n_vec + 255) / 256, 2048)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return fail_arg("bias_act_fwd: bias must be 16-byte aligned");
#define BA_FWD1(T, ACT, DROP) hipLaunchKernelGGL((bias_act_fwd_kernel<T, ACT, DROP>), grid, block, 0, st, (const T*)x, bias, (T*)y, n_vec, cols, thr, scale, seed_state, stream_id)
#define BA_FWD(T, ACT) do { if (thr) BA_FWD1(T, ACT, true); else BA_FWD1(T, ACT, false); } while (0)
    if (dtype == SHG_F32) { if (act == 0) BA_FWD(float, 0); else if (act == 1) BA_FWD(float, 1); else BA_FWD(float, 2); }
    else { if (act == 0) BA_FWD(bf16_t, 0); else if (act == 1) BA_FWD(bf16_t, 1); else BA_FWD(bf16_t, 2); }
#undef BA_FWD
#undef BA_FWD1
    return check_launch("bias_act_fwd");
}

extern "C" int shg_bias_act_bwd_view(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial,
                                     int n_partials, int dtype, int64_t rows, int cols, int act, float p_drop,
                                     const uint64_t* seed_state, uint64_t stream_id, int64_t dy_rows_per_group,
                                     int64_t dy_group_stride, int64_t dy_row_offset, void* dx2, const int32_t* dx2_rows, void* stream) {
    return shg_bias_act_bwd_rows(x, bias, dy, dx, dbias_partial, n_partials, dtype, rows, cols, act, p_drop, seed_state, stream_id,
                                 dy_rows_per_group, dy_group_stride, dy_row_offset, dx2, dx2_rows, nullptr, stream);
}

extern "C" int shg_bias_act_bwd_rows(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial,
                                     int n_partials, int dtype, int64_t rows, int cols, int act, float p_drop,
                                     const uint64_t* seed_state, uint64_t stream_id, int64_t dy_rows_per_group,
                                     int64_t dy_group_stride, int64_t dy_row_offset, void* dx2, const int32_t* dx2_rows,
                                     const int32_t* x_rows, void* stream) {
    if (!x || !dy || !dx) return fail_arg("bias_act_bwd: null pointer");
    if (x_rows && p_drop > 0.f) return fail_arg("bias_act_bwd_rows: a row table with dropout is not supported (the mask is indexed by x's element)");
    if (dy_rows_per_group < 0 || (dy_rows_per_group > 0 && (dy_row_offset < 0 || dy_group_stride < dy_rows_per_group + dy_row_offset)))
        return fail_arg("bias_act_bwd: bad view of dy");
    if (dx2 && (reinterpret_cast<uintptr_t>(dx2) & 15)) return fail_arg("bias_act_bwd: dx2 must be 16-byte aligned");
    const BwdView vw{dy_rows_per_group, dy_group_stride, dy_row_offset, dx2, dx2_rows, x_rows};
    if (int e = check_cols(dtype, cols, "bias_act_bwd: unsupported cols", 16)) return e;
    if (n_partials < 1 || n_partials > MAX_PARTIALS) return fail_arg("bias_act_bwd: bad n_partials");
    if (rows <= 0) return rows == 0 ? 0 : fail_arg("bias_act_bwd: negative rows");
    const uint32_t thr = dropout_threshold(p_drop);
    const float scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    dim3 grid(n_partials), block(256);
    const size_t lds = (size_t)4 * cols * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int nch = (cols / (dtype == SHG_BF16 ? 8 : 4) + 63) / 64;
    if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return fail_arg("bias_act_bwd: bias must be 16-byte aligned");
#define BA_BWD3(T, ACT, NCH, DROP) hipLaunchKernelGGL((bias_act_bwd_kernel<T, ACT, NCH, DROP>), grid, block, lds, st, (const T*)x, bias, (const T*)dy, (T*)dx, dbias_partial, rows, cols, thr, scale, seed_state, stream_id, vw)
#define BA_BWD2(T, ACT, NCH) do { if (thr) BA_BWD3(T, ACT, NCH, true); else BA_BWD3(T, ACT, NCH, false); } while (0)
#define BA_BWD(T, ACT) do { if (nch <= 2) BA_BWD2(T, ACT, 2); else if (nch <= 4) BA_BWD2(T, ACT, 4); else if (nch <= 8) BA_BWD2(T, ACT, 8); else BA_BWD2(T, ACT, 16); } while (0)
    if (dtype == SHG_F32) { if (act == 0) BA_BWD(float, 0); else if (act == 1) BA_BWD(float, 1); else BA_BWD(float, 2); }
    else { if (act == 0) BA_BWD(bf16_t, 0); else if (act == 1) BA_BWD(bf16_t, 1); else BA_BWD(bf16_t, 2); }
#undef BA_BWD
#undef BA_BWD2
#undef BA_BWD3
    return check_launch("bias_act_bwd");
}

extern "C" int shg_bias_act_bwd(const void* x, const float* bias, const void* dy, void* dx, float* dbias_partial,
                                int n_partials, int dtype, int64_t rows, int cols, int act, float p_drop,
                                const uint64_t* seed_state, uint64_t stream_id, void* stream) {
    return shg_bias_act_bwd_view(x, bias, dy, dx, dbias_partial, n_partials, dtype, rows, cols, act, p_drop, seed_state, stream_id, 0, 0,
                                 0, nullptr, nullptr, stream);
}

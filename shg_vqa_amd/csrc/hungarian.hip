// Per-frame Hungarian matcher on the GPU (replaces lxrt/matcher.py:62-80 + scipy LSAP).
//
// One wave per frame-problem.  Phase 1 (all 64 lanes): for every query row of the frame, a wave-cooperative
// softmax (coalesced row read, shuffle reductions) and the gather of the target columns ->
// cost = -softmax[q, tgt_j] (fp32) into LDS as float64.  Phase 2 (one lane): the shortest-augmenting-path
// solver with SciPy's scan order and tie rules in float64 on the LDS copy, so the indices are bit-identical
// to scipy.optimize.linear_sum_assignment.  Problems are at most 8x8: latency-bound, not HBM-bound
// (reads 4096 x 457 logits = 7.5 MB per call), so the grid is one small wave per frame (512 waves at B = 32 -
// eight frames per wave, as in round 1, left 192 of 256 CUs without work and took 183 us per call).
#include <math.h>

#include "common.h"

namespace shg {

constexpr int HMAX = 8;             // max queries per frame / targets per frame
constexpr int FRAMES_PER_WAVE = 8;

struct LsapScratch {                // one problem, lives in LDS
    double cost[HMAX * HMAX];       // solver orientation: [row = target][col = query]
    double u[HMAX], v[HMAX], spc[HMAX];
    int col4row[HMAX], row4col[HMAX], path[HMAX], remaining[HMAX];
    unsigned char in_sr[HMAX], in_sc[HMAX];
};

// Solves the (n_tgt x n_q, n_tgt <= n_q) problem held in s.cost with leading dimension HMAX.
// This is the "transposed" orientation SciPy uses when the matcher's (queries x targets) matrix has
// more rows than columns.  On return s.col4row[t] = query assigned to target t.
__device__ void lsap_wide(LsapScratch& s, int nr, int nc) {
    for (int i = 0; i < nr; ++i) { s.u[i] = 0.0; s.col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { s.v[j] = 0.0; s.row4col[j] = -1; }
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int i = cur, sink = -1, n_rem = nc;
        for (int r = 0; r < nr; ++r) s.in_sr[r] = 0;
        for (int j = 0; j < nc; ++j) { s.in_sc[j] = 0; s.spc[j] = INFINITY; s.path[j] = -1; s.remaining[j] = nc - 1 - j; }
        while (sink == -1) {
            int best = -1;
            double lowest = INFINITY;
            s.in_sr[i] = 1;
            const double ui = s.u[i];
            for (int it = 0; it < n_rem; ++it) {
                const int j = s.remaining[it];
                const double r = min_val + s.cost[i * HMAX + j] - ui - s.v[j];
                double sj = s.spc[j];
                if (r < sj) { s.path[j] = i; s.spc[j] = r; sj = r; }
                if (sj < lowest || (sj == lowest && s.row4col[j] == -1)) { lowest = sj; best = it; }
            }
            min_val = lowest;
            if (best < 0) return;   // cannot happen with finite costs
            const int j = s.remaining[best];
            if (s.row4col[j] == -1) sink = j; else i = s.row4col[j];
            s.in_sc[j] = 1;
            s.remaining[best] = s.remaining[--n_rem];
        }
        s.u[cur] += min_val;
        for (int r = 0; r < nr; ++r)
            if (s.in_sr[r] && r != cur) s.u[r] += min_val - s.spc[s.col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (s.in_sc[j]) s.v[j] -= min_val - s.spc[j];
        int j = sink;
        for (;;) {
            const int r = s.path[j];
            s.row4col[j] = r;
            const int prev = s.col4row[r];
            s.col4row[r] = j;
            j = prev;
            if (r == cur) break;
        }
    }
}

// Writes the matcher's output convention for one frame: queries ascending with their targets.
// rows = queries (R), cols = targets (n); SciPy solves the transpose when n < R and then orders by
// query; when n == R it solves directly (rows = queries).
__device__ void emit_assignment(LsapScratch& s, int R, int n, bool transposed, int64_t* out_q, int64_t* out_t) {
    for (int k = 0; k < R; ++k) { out_q[k] = -1; out_t[k] = -1; }
    if (n == 0) return;
    if (!transposed) {           // solved with rows = queries: col4row[q] = target
        for (int q = 0; q < R; ++q) { out_q[q] = q; out_t[q] = s.col4row[q]; }
        return;
    }
    // col4row[t] = query; stable order by query index
    int k = 0;
    for (int q = 0; q < R; ++q)
        for (int t = 0; t < n; ++t)
            if (s.col4row[t] == q) { out_q[k] = q; out_t[k] = t; ++k; }
}

// One wave per frame.  NV > 0: the row has at most 64 NV classes and is held in NV registers per lane; the loop over the
// frame's queries is fully unrolled, so the loads of all rows are in flight together (the kernel is latency-bound: 8 rows of
// <= 1 KB per wave).  NV == 0: any class count, rows read twice.  Either way a lane adds its classes in increasing order and
// the lane sums meet in the same butterfly, so both paths give the same bits.
template <typename T, int NV>
__global__ __launch_bounds__(64) void hungarian_per_frame_kernel(
    const T* __restrict__ logits, int n_frames, int R, int C, const int64_t* __restrict__ tgt,
    const int32_t* __restrict__ tgt_len, int64_t background, int64_t* __restrict__ out_q,
    int64_t* __restrict__ out_t, int64_t* __restrict__ out_grid) {
    __shared__ LsapScratch s;
    const int lane = threadIdx.x;
    const int f = blockIdx.x;
    if (f >= n_frames) return;
    const int n = min(max(tgt_len[f], 0), R);
    // phase 1: softmax statistics + gathered cost entries, cost = -softmax[q, tgt_j] (fp32) as float64
    if (n > 0) {
        // (a class id outside [0, C) is a caller error - the reference's out_prob[:, tgt_ids] raises IndexError,
        // matcher.py:74; the host wrappers check it - here it is clamped so that the read stays inside the row)
        const int64_t cls = lane < n ? min(max(tgt[(int64_t)f * R + lane], (int64_t)0), (int64_t)C - 1) : 0;
        if constexpr (NV > 0) {
            float val[HMAX][NV], picked[HMAX];
#pragma unroll
            for (int q = 0; q < HMAX; ++q) {
                const T* row = logits + ((int64_t)f * R + min(q, R - 1)) * C;
#pragma unroll
                for (int k = 0; k < NV; ++k) val[q][k] = lane + 64 * k < C ? to_f32(row[lane + 64 * k]) : -INFINITY;
                picked[q] = to_f32(row[cls]);
            }
#pragma unroll
            for (int q = 0; q < HMAX; ++q) {
                if (q < R) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int k = 0; k < NV; ++k) mx = fmaxf(mx, val[q][k]);
                    mx = wave_max(mx);
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        if (lane + 64 * k < C) sum += expf(val[q][k] - mx);
                    sum = wave_sum(sum);
                    if (lane < n) {
                        const double cst = (double)(-(expf(picked[q] - mx) / sum));
                        // n == R: solve with rows = queries; n < R: solve the transpose (rows = targets)
                        if (n == R) s.cost[q * HMAX + lane] = cst;
                        else s.cost[lane * HMAX + q] = cst;
                    }
                }
            }
        } else {
            for (int q = 0; q < R; ++q) {
                const T* row = logits + ((int64_t)f * R + q) * C;
                float mx = -INFINITY;
                for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f32(row[c]));
                mx = wave_max(mx);
                float sum = 0.f;
                for (int c = lane; c < C; c += 64) sum += expf(to_f32(row[c]) - mx);
                sum = wave_sum(sum);
                if (lane < n) {
                    const double cst = (double)(-(expf(to_f32(row[cls]) - mx) / sum));
                    if (n == R) s.cost[q * HMAX + lane] = cst;
                    else s.cost[lane * HMAX + q] = cst;
                }
            }
        }
    }
    __syncthreads();
    // phase 2: the 8 x 8 solve is sequential (SciPy's scan order and tie rules): one lane
    if (lane == 0) {
        int64_t* oq = out_q + (int64_t)f * R;
        int64_t* ot = out_t + (int64_t)f * R;
        if (n > 0) {
            if (n == R) lsap_wide(s, R, R); else lsap_wide(s, n, R);
        }
        emit_assignment(s, R, n, n != R, oq, ot);
        if (out_grid) {
            int64_t* g = out_grid + (int64_t)f * R;
            for (int k = 0; k < R; ++k) g[k] = background;
            for (int k = 0; k < n; ++k) g[oq[k]] = tgt[(int64_t)f * R + ot[k]];
        }
    }
}

// explicit-cost entry point: cost [n, rows, cols_max] fp32
__global__ __launch_bounds__(64) void lsap_batched_kernel(const float* __restrict__ cost, int n_prob, int R,
                                                          int cmax, const int32_t* __restrict__ n_cols,
                                                          int64_t* __restrict__ out_r, int64_t* __restrict__ out_c) {
    __shared__ LsapScratch scratch[FRAMES_PER_WAVE];
    const int lane = threadIdx.x;
    if (lane >= FRAMES_PER_WAVE) return;
    const int p = blockIdx.x * FRAMES_PER_WAVE + lane;
    if (p >= n_prob) return;
    LsapScratch& s = scratch[lane];
    const int n = min(max(n_cols[p], 0), cmax);
    const float* c = cost + (int64_t)p * R * cmax;
    const bool transposed = n < R;
    for (int q = 0; q < R; ++q)
        for (int t = 0; t < n; ++t) {
            const double v = (double)c[q * cmax + t];
            if (transposed) s.cost[t * HMAX + q] = v; else s.cost[q * HMAX + t] = v;
        }
    int64_t oq[HMAX], ot[HMAX];
    if (n > 0) { if (transposed) lsap_wide(s, n, R); else lsap_wide(s, R, R); }
    emit_assignment(s, R, n, transposed, oq, ot);
    const int width = min(R, cmax);
    for (int k = 0; k < width; ++k) { out_r[(int64_t)p * width + k] = oq[k]; out_c[(int64_t)p * width + k] = ot[k]; }
}

// ------------------------------------------------------------------------------------------------
// Larger problems (the per-clip branch of the matcher, matcher.py:82-104: num_queries x labels-of-the-clip,
// up to 128 x 128): one wave per problem.  Same algorithm and float64 arithmetic; the column scan of
// every shortest-path step is spread over the 64 lanes (column it, it+64, ..) and the lane results are
// combined so that the winner is the one SciPy's sequential scan picks: the minimum; on an exact tie the LAST
// unassigned column in scan order, or - when no tied column is unassigned - the FIRST in scan order.
// ------------------------------------------------------------------------------------------------
constexpr int WMAX = 128;

struct WaveLsap {                                   // carved from dynamic LDS
    float* cost;                                    // [nr][ldc] solver orientation
    double *u, *v, *spc;
    int *col4row, *row4col, *path, *remaining, *q2t;
    unsigned char *in_sr, *in_sc;
};

__device__ void wave_lsap_solve(const WaveLsap& s, int nr, int nc, int ldc, int lane) {
    for (int i = lane; i < nr; i += 64) { s.u[i] = 0.0; s.col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { s.v[j] = 0.0; s.row4col[j] = -1; }
    __syncthreads();
    for (int cur = 0; cur < nr; ++cur) {
        for (int r = lane; r < nr; r += 64) s.in_sr[r] = 0;
        for (int j = lane; j < nc; j += 64) { s.in_sc[j] = 0; s.spc[j] = INFINITY; s.path[j] = -1; s.remaining[j] = nc - 1 - j; }
        __syncthreads();
        double min_val = 0.0;
        int i = cur, sink = -1, n_rem = nc;
        while (sink == -1) {
            if (lane == 0) s.in_sr[i] = 1;
            const double ui = s.u[i];
            const float* crow = s.cost + (size_t)i * ldc;
            // lane-local sequential scan over its columns (increasing it), SciPy's update rule
            double lowest = INFINITY;
            int best = -1, best_free = 0;
            for (int it = lane; it < n_rem; it += 64) {
                const int j = s.remaining[it];
                const double r = min_val + (double)crow[j] - ui - s.v[j];
                double sj = s.spc[j];
                if (r < sj) { s.path[j] = i; s.spc[j] = r; sj = r; }
                const int fr = s.row4col[j] == -1;
                if (sj < lowest || (sj == lowest && fr)) { lowest = sj; best = it; best_free = fr; }
            }
            // combine: min value; tie -> unassigned beats assigned; unassigned: larger it; assigned: smaller it
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(lowest, o, 64);
                const int ob = __shfl_xor(best, o, 64), of = __shfl_xor(best_free, o, 64);
                bool take = false;
                if (ob >= 0) {
                    if (best < 0 || ov < lowest) take = true;
                    else if (ov == lowest) {
                        if (of != best_free) take = of > best_free;
                        else take = of ? (ob > best) : (ob < best);
                    }
                }
                if (take) { lowest = ov; best = ob; best_free = of; }
            }
            if (best < 0) return;                    // cannot happen with finite costs
            min_val = lowest;
            const int j = s.remaining[best];
            const int rj = s.row4col[j];
            __syncthreads();                         // every lane has read remaining[] / row4col[] of this step
            if (rj == -1) sink = j; else i = rj;
            if (lane == 0) {
                s.in_sc[j] = 1;
                s.remaining[best] = s.remaining[n_rem - 1];
            }
            --n_rem;
            __syncthreads();
        }
        if (lane == 0) s.u[cur] += min_val;
        for (int r = lane; r < nr; r += 64)
            if (s.in_sr[r] && r != cur) s.u[r] += min_val - s.spc[s.col4row[r]];
        for (int j = lane; j < nc; j += 64)
            if (s.in_sc[j]) s.v[j] -= min_val - s.spc[j];
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int r = s.path[j];
                s.row4col[j] = r;
                const int prev = s.col4row[r];
                s.col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(64) void hungarian_wave_kernel(
    const T* __restrict__ logits, int n_prob, int R, int C, const int64_t* __restrict__ tgt,
    const int32_t* __restrict__ tgt_len, int64_t background, int64_t* __restrict__ out_q,
    int64_t* __restrict__ out_t, int64_t* __restrict__ out_grid) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int lane = threadIdx.x, f = blockIdx.x;
    if (f >= n_prob) return;
    WaveLsap s;
    char* p = wsm;
    s.cost = (float*)p; p += (size_t)R * R * 4;
    s.u = (double*)p; p += (size_t)R * 8;
    s.v = (double*)p; p += (size_t)R * 8;
    s.spc = (double*)p; p += (size_t)R * 8;
    s.col4row = (int*)p; p += (size_t)R * 4;
    s.row4col = (int*)p; p += (size_t)R * 4;
    s.path = (int*)p; p += (size_t)R * 4;
    s.remaining = (int*)p; p += (size_t)R * 4;
    s.q2t = (int*)p; p += (size_t)R * 4;
    s.in_sr = (unsigned char*)p; p += R;
    s.in_sc = (unsigned char*)p;
    const int n = min(max(tgt_len[f], 0), R);
    const bool transposed = n < R;                  // SciPy transposes when rows (queries) > cols (targets)
    int64_t* oq = out_q + (int64_t)f * R;
    int64_t* ot = out_t + (int64_t)f * R;
    for (int k = lane; k < R; k += 64) { oq[k] = -1; ot[k] = -1; if (out_grid) out_grid[(int64_t)f * R + k] = background; }
    if (n == 0) return;
    // cost = -softmax[q, tgt_t] (fp32), solver orientation [target][query] when transposed, else [query][target]
    for (int q = 0; q < R; ++q) {
        const T* row = logits + ((int64_t)f * R + q) * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, to_f32(row[c]));
        mx = wave_max(mx);
        float sum = 0.f;
        for (int c = lane; c < C; c += 64) sum += expf(to_f32(row[c]) - mx);
        sum = wave_sum(sum);
        for (int t = lane; t < n; t += 64) {
            const int64_t cls = min(max(tgt[(int64_t)f * R + t], (int64_t)0), (int64_t)C - 1);      // (see above: kept inside the row)
            const float pr = expf(to_f32(row[cls]) - mx) / sum;
            if (transposed) s.cost[(size_t)t * R + q] = -pr; else s.cost[(size_t)q * R + t] = -pr;
        }
    }
    __syncthreads();
    wave_lsap_solve(s, transposed ? n : R, R, R, lane);     // (n == R: square, rows = queries)
    // output: pairs ordered by query index
    if (!transposed) {
        for (int q = lane; q < R; q += 64) {
            oq[q] = q;
            ot[q] = s.col4row[q];
            if (out_grid) out_grid[(int64_t)f * R + q] = tgt[(int64_t)f * R + s.col4row[q]];
        }
        return;
    }
    for (int q = lane; q < R; q += 64) s.q2t[q] = -1;
    __syncthreads();
    for (int t = lane; t < n; t += 64) s.q2t[s.col4row[t]] = t;
    __syncthreads();
    if (lane == 0) {
        int k = 0;
        for (int q = 0; q < R; ++q) {
            const int t = s.q2t[q];
            if (t >= 0) {
                oq[k] = q;
                ot[k] = t;
                ++k;
                if (out_grid) out_grid[(int64_t)f * R + q] = tgt[(int64_t)f * R + t];
            }
        }
    }
}

}  // namespace shg

extern "C" int shg_hungarian_per_frame(const void* logits, int dtype, int n_frames, int per_frame, int n_classes,
                                       const int64_t* tgt, const int32_t* tgt_len, int64_t background_class,
                                       int64_t* out_query, int64_t* out_target, int64_t* out_grid, void* stream) {
    using namespace shg;
    if (!logits || !tgt || !tgt_len || !out_query || !out_target) return fail_arg("hungarian: null pointer");
    if (per_frame < 1 || per_frame > WMAX) return fail_arg("hungarian: per_frame must be in [1,128]");
    if (n_frames < 0 || n_classes < 1) return fail_arg("hungarian: bad sizes");
    if (n_frames == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (per_frame > HMAX) {                          // one wave per problem (per-clip matching)
        const size_t lds = (size_t)per_frame * per_frame * 4 + 3 * (size_t)per_frame * 8 + 5 * (size_t)per_frame * 4 +
                           2 * (size_t)per_frame + 64;
        if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("hungarian: bad dtype");
        static std::atomic<uint64_t> raised{0};      // one bit per device
        static std::atomic<uint64_t> raised_b{0};
        raise_lds_limit(raised, reinterpret_cast<const void*>(hungarian_wave_kernel<float>), 96 * 1024);
        raise_lds_limit(raised_b, reinterpret_cast<const void*>(hungarian_wave_kernel<bf16_t>), 96 * 1024);
        if (dtype == SHG_F32)
            hipLaunchKernelGGL(hungarian_wave_kernel<float>, dim3(n_frames), dim3(64), lds, st, (const float*)logits, n_frames,
                               per_frame, n_classes, tgt, tgt_len, background_class, out_query, out_target, out_grid);
        else
            hipLaunchKernelGGL(hungarian_wave_kernel<bf16_t>, dim3(n_frames), dim3(64), lds, st, (const bf16_t*)logits, n_frames,
                               per_frame, n_classes, tgt, tgt_len, background_class, out_query, out_target, out_grid);
        return check_launch("hungarian_per_clip");
    }
    dim3 grid(n_frames), block(64);
    if (dtype != SHG_F32 && dtype != SHG_BF16) return fail_arg("hungarian: bad dtype");
#define SHG_HPF(T, NV) hipLaunchKernelGGL((hungarian_per_frame_kernel<T, NV>), grid, block, 0, st, (const T*)logits, n_frames, \
                                          per_frame, n_classes, tgt, tgt_len, background_class, out_query, out_target, out_grid)
#define SHG_HPF_T(T) do { if (n_classes <= 256) SHG_HPF(T, 4); else if (n_classes <= 512) SHG_HPF(T, 8); else SHG_HPF(T, 0); } while (0)
    if (dtype == SHG_F32) SHG_HPF_T(float); else SHG_HPF_T(bf16_t);
#undef SHG_HPF_T
#undef SHG_HPF
    return check_launch("hungarian_per_frame");
}

extern "C" int shg_lsap_batched(const float* cost, int n, int rows, int cols_max, const int32_t* n_cols,
                                int64_t* out_row, int64_t* out_col, void* stream) {
    using namespace shg;
    if (!cost || !n_cols || !out_row || !out_col) return fail_arg("lsap: null pointer");
    if (rows < 1 || rows > HMAX || cols_max < 1 || cols_max > HMAX || cols_max > rows)
        return fail_arg("lsap: need 1 <= cols_max <= rows <= 8");
    if (n <= 0) return n == 0 ? 0 : fail_arg("lsap: negative n");
    dim3 grid((n + FRAMES_PER_WAVE - 1) / FRAMES_PER_WAVE), block(64);
    hipLaunchKernelGGL(lsap_batched_kernel, grid, block, 0, (hipStream_t)stream, cost, n, rows, cols_max, n_cols,
                       out_row, out_col);
    return check_launch("lsap_batched");
}

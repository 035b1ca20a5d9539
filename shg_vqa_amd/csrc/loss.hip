// Loss kernels: class-weighted cross entropy over the query slots (set loss) and BCE-with-logits.
// Small, latency-bound problems (4096 x 457 logits); one wave per row, coalesced reads.
#include <math.h>

#include "common.h"

namespace shg {

template <typename T>
__global__ __launch_bounds__(256) void wce_rows_kernel(const T* __restrict__ logits, int64_t rows, int C,
                                                       const int64_t* __restrict__ target,
                                                       const float* __restrict__ cw, int64_t background,
                                                       float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* x = logits + row * C;
    float mx = -INFINITY;
    int arg = 0;
    for (int c = lane; c < C; c += 64) {
        const float v = to_f32(x[c]);
        if (v > mx) { mx = v; arg = c; }
    }
    // wave arg-max with lowest-index tie-break (torch.max / topk semantics on equal values are
    // unspecified; lowest index is what the CPU kernels return)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (ov > mx || (ov == mx && oa < arg)) { mx = ov; arg = oa; }
    }
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(to_f32(x[c]) - mx);
    sum = wave_sum(sum);
    if (lane == 0) {
        const int64_t t = target[row];
        const float lse = mx + logf(sum);
        const float w = cw ? cw[t] : 1.f;
        stats[row] = lse;
        stats[rows + row] = w * (lse - to_f32(x[t]));
        stats[2 * rows + row] = w;
        // class_error bookkeeping rides in the sign bit-free 4th plane: matched & correct
        stats[3 * rows + row] = (t != background) ? ((arg == (int)t) ? 2.f : 1.f) : 0.f;
    }
}

__global__ __launch_bounds__(256) void wce_sum_kernel(const float* __restrict__ stats, int64_t rows, float* __restrict__ sums) {
    __shared__ double sh[4][256];
    double a = 0, b = 0, c = 0, d = 0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) {
        a += stats[rows + r];
        b += stats[2 * rows + r];
        const float f = stats[3 * rows + r];
        c += (f == 2.f) ? 1.0 : 0.0;
        d += (f >= 1.f) ? 1.0 : 0.0;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c; sh[3][threadIdx.x] = d;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = (float)sh[threadIdx.x][0];
}

template <typename T>
__global__ __launch_bounds__(256) void wce_bwd_kernel(const T* __restrict__ logits, int64_t rows, int C,
                                                      const int64_t* __restrict__ target,
                                                      const float* __restrict__ cw, const float* __restrict__ stats,
                                                      const float* __restrict__ sums, const float* __restrict__ gscale,
                                                      T* __restrict__ dlogits, int64_t ldd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t t = target[row];
    const float w = cw ? cw[t] : 1.f;
    const float coef = (gscale ? gscale[0] : 1.f) * w / sums[1];
    const float lse = stats[row];
    const T* x = logits + row * C;
    T* d = dlogits + row * ldd;
    for (int c = lane; c < (int)ldd; c += 64) {
        const float p = c < C ? expf(to_f32(x[c]) - lse) : 0.f;
        d[c] = from_f32<T>(c < C ? coef * (p - (c == t ? 1.f : 0.f)) : 0.f);   // pad columns are zeroed
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bce_kernel(const T* __restrict__ logits, int64_t n, int64_t rows, int C,
                                                  const float* __restrict__ target, const float* __restrict__ gscale,
                                                  float* __restrict__ loss, T* __restrict__ dlogits, int64_t ldd) {
    __shared__ double sh[256];
    const float g = (gscale ? gscale[0] : 1.f) / (float)rows;
    double acc = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float x = to_f32(logits[i]), y = target[i];
        acc += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
        if (dlogits) dlogits[(i / C) * ldd + (i % C)] = from_f32<T>(g * (1.f / (1.f + expf(-x)) - y));
    }
    if (dlogits && ldd > C)
        for (int64_t i = threadIdx.x; i < rows * (ldd - C); i += 256) dlogits[(i / (ldd - C)) * ldd + C + (i % (ldd - C))] = from_f32<T>(0.f);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = (float)(sh[0] / (double)rows);   // = C * mean over rows*C
}

// total = bce * bce_scale + rel[0] / rel[1] + act[0] / act[1] and the step's reported scalars in ONE launch (and one for the
// gradients): as torch scalar arithmetic this was 14 forward and ~24 backward kernels of one thread each between the end of the
// forward pass and the first kernel of backward - a strictly serial stretch of the step.
__global__ void loss_combine_fwd_kernel(const float* __restrict__ rel, const float* __restrict__ act, const float* __restrict__ bce,
                                        float bce_scale, float* __restrict__ total, float* __restrict__ diag) {
    const float rel_ce = rel[0] / rel[1], act_ce = act[0] / act[1];
    total[0] = bce[0] * bce_scale + rel_ce + act_ce;
    diag[0] = bce[0];
    diag[1] = rel_ce;
    diag[2] = act_ce;
    diag[3] = 100.f - 100.f * rel[2] / fmaxf(rel[3], 1.f);      // class error of the matched slots (agqaHGQA.py:221-229)
    diag[4] = 100.f - 100.f * act[2] / fmaxf(act[3], 1.f);
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ rel, const float* __restrict__ act,
                                        float bce_scale, float* __restrict__ d_rel, float* __restrict__ d_act, float* __restrict__ d_bce) {
    const float gt = g ? g[0] : 1.f;
    d_rel[0] = gt / rel[1]; d_rel[1] = -gt * rel[0] / (rel[1] * rel[1]); d_rel[2] = 0.f; d_rel[3] = 0.f;
    d_act[0] = gt / act[1]; d_act[1] = -gt * act[0] / (act[1] * act[1]); d_act[2] = 0.f; d_act[3] = 0.f;
    d_bce[0] = gt * bce_scale;
}

}  // namespace shg

using namespace shg;

extern "C" int shg_weighted_ce_fwd(const void* logits, int dtype, int64_t rows, int n_classes, const int64_t* target,
                                   const float* class_weight, int64_t background_class, float* row_stats, float* sums,
                                   void* stream) {
    if (!logits || !target || !row_stats || !sums) return fail_arg("weighted_ce_fwd: null pointer");
    if (rows <= 0 || n_classes <= 0) return fail_arg("weighted_ce_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == SHG_F32)
        hipLaunchKernelGGL(wce_rows_kernel<float>, grid, block, 0, st, (const float*)logits, rows, n_classes, target, class_weight, background_class, row_stats);
    else if (dtype == SHG_BF16)
        hipLaunchKernelGGL(wce_rows_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, rows, n_classes, target, class_weight, background_class, row_stats);
    else return fail_arg("weighted_ce_fwd: bad dtype");
    hipLaunchKernelGGL(wce_sum_kernel, dim3(1), dim3(256), 0, st, row_stats, rows, sums);
    return check_launch("weighted_ce_fwd");
}

extern "C" int shg_weighted_ce_bwd(const void* logits, int dtype, int64_t rows, int n_classes, const int64_t* target,
                                   const float* class_weight, const float* row_stats, const float* sums,
                                   const float* gscale, void* dlogits, int64_t ldd, void* stream) {
    if (!logits || !target || !row_stats || !sums || !dlogits) return fail_arg("weighted_ce_bwd: null pointer");
    if (rows <= 0 || n_classes <= 0 || ldd < n_classes) return fail_arg("weighted_ce_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == SHG_F32)
        hipLaunchKernelGGL(wce_bwd_kernel<float>, grid, block, 0, st, (const float*)logits, rows, n_classes, target, class_weight, row_stats, sums, gscale, (float*)dlogits, ldd);
    else if (dtype == SHG_BF16)
        hipLaunchKernelGGL(wce_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, rows, n_classes, target, class_weight, row_stats, sums, gscale, (bf16_t*)dlogits, ldd);
    else return fail_arg("weighted_ce_bwd: bad dtype");
    return check_launch("weighted_ce_bwd");
}

extern "C" int shg_bce_logits_fwd_bwd(const void* logits, int dtype, int64_t rows, int n_classes, const float* target,
                                      const float* gscale, float* loss, void* dlogits, int64_t ldd, void* stream) {
    if (!logits || !target) return fail_arg("bce: null pointer");
    if (rows <= 0 || n_classes <= 0 || (dlogits && ldd < n_classes)) return fail_arg("bce: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = rows * n_classes;
    if (dtype == SHG_F32)
        hipLaunchKernelGGL(bce_kernel<float>, dim3(1), dim3(256), 0, st, (const float*)logits, n, rows, n_classes, target, gscale, loss, (float*)dlogits, ldd);
    else if (dtype == SHG_BF16)
        hipLaunchKernelGGL(bce_kernel<bf16_t>, dim3(1), dim3(256), 0, st, (const bf16_t*)logits, n, rows, n_classes, target, gscale, loss, (bf16_t*)dlogits, ldd);
    else return fail_arg("bce: bad dtype");
    return check_launch("bce_logits");
}

extern "C" int shg_loss_combine_fwd(const float* rel_sums, const float* act_sums, const float* bce, float bce_scale, float* total,
                                    float* diag, void* stream) {
    if (!rel_sums || !act_sums || !bce || !total || !diag) return shg::fail_arg("loss_combine_fwd: null pointer");
    hipLaunchKernelGGL(shg::loss_combine_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rel_sums, act_sums, bce, bce_scale, total, diag);
    return shg::check_launch("loss_combine_fwd");
}

extern "C" int shg_loss_combine_bwd(const float* d_total, const float* rel_sums, const float* act_sums, float bce_scale,
                                    float* d_rel_sums, float* d_act_sums, float* d_bce, void* stream) {
    if (!rel_sums || !act_sums || !d_rel_sums || !d_act_sums || !d_bce) return shg::fail_arg("loss_combine_bwd: null pointer");
    hipLaunchKernelGGL(shg::loss_combine_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_total, rel_sums, act_sums, bce_scale,
                       d_rel_sums, d_act_sums, d_bce);
    return shg::check_launch("loss_combine_bwd");
}

// Optimiser on a flat fp32 parameter arena: global gradient norm, clip + BertAdam update with a
// bf16 shadow copy written in the same pass.  Pure HBM streaming: 4 reads (p, g, m, v) and 3-4
// writes (p, m, v, shadow) of 16 bytes per lane.
#include <math.h>

#include <cstdlib>

#include "common.h"

namespace shg {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    const int64_t n4 = n / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)x[i] * (double)x[i];
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// extra: one more addend that kernels have been accumulating with atomics (the convolutions' fused sums); it is reset for the next step
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partial, int n_partial, float* __restrict__ out,
                                                          double* __restrict__ extra = nullptr) {
    __shared__ double sh[256];
    double acc = 0.0;
    if (extra && threadIdx.x == 0) { acc = extra[0]; extra[0] = 0.0; }
    for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(sh[0]);
}

__device__ __forceinline__ double warmup_linear(double x, double warmup) {
    if (x < warmup) return x / warmup;
    const double y = (x - 1.0) / (warmup - 1.0);
    return y > 0.0 ? y : 0.0;
}

// UNROLL: independent 16-byte vectors per lane and iteration (1, 2 or 4); NT_MAIN: non-temporal loads of p / g / m / v and stores
// of p / m / v; NT_AUX: non-temporal stores of the zeroed gradient and of the bf16 shadow as well ("bertadam_mode" tuning switch:
// bit 0 UNROLL 2, bit 3 UNROLL 4, bit 1 NT_AUX, bit 2 switches NT_MAIN off)
template <int UNROLL, bool NT_MAIN, bool NT_AUX>
__global__ __launch_bounds__(256) void bertadam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       bf16_t* __restrict__ shadow, int64_t n,
                                                       const float* __restrict__ grad_norm, float max_norm, float lr,
                                                       float warmup, int64_t t_total, float b1, float b2, float eps,
                                                       float wd, const int64_t* __restrict__ step_state, int zero_g) {
    float clip = 1.f;
    if (grad_norm && max_norm > 0.f) clip = fminf(max_norm / (grad_norm[0] + 1e-6f), 1.f);
    double lr_d = (double)lr;
    if (t_total != -1) lr_d *= warmup_linear((double)step_state[0] / (double)t_total, (double)warmup);
    const float lr_t = (float)lr_d;
    const float one_b1 = 1.f - b1, one_b2 = 1.f - b2;
    const int64_t n4 = n / 4;
    // pure streaming (30 bytes per parameter, every byte touched once): non-temporal accesses keep the arenas out of
    // the L2 / Infinity Cache, and two independent vectors per iteration double the loads in flight per lane
    auto update = [&](f32x4& pv, const f32x4& gv, f32x4& mv, f32x4& vv, bf16x4& sv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = gv[j] * clip;
            mv[j] = mv[j] * b1 + one_b1 * gg;
            vv[j] = vv[j] * b2 + one_b2 * gg * gg;
            const float upd = mv[j] / (sqrtf(vv[j]) + eps) + wd * pv[j];
            pv[j] -= lr_t * upd;
            sv[j] = (bf16_t)pv[j];
        }
    };
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    bf16x4* s4 = reinterpret_cast<bf16x4*>(shadow);
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto ld = [](const f32x4* q) { return NT_MAIN ? __builtin_nontemporal_load(q) : *q; };
    auto st = [](f32x4 x, f32x4* q) { if (NT_MAIN) __builtin_nontemporal_store(x, q); else *q = x; };
    for (; UNROLL > 1 && i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f32x4 pa[UNROLL], ga[UNROLL], ma[UNROLL], va[UNROLL];
        bf16x4 sa[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t k = i + u * stride;
            pa[u] = ld(p4 + k); ga[u] = ld(g4 + k); ma[u] = ld(m4 + k); va[u] = ld(v4 + k);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) update(pa[u], ga[u], ma[u], va[u], sa[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t k = i + u * stride;
            st(pa[u], p4 + k); st(ma[u], m4 + k); st(va[u], v4 + k);
            if (zero_g) { if (NT_AUX) __builtin_nontemporal_store(zero4, g4 + k); else g4[k] = zero4; }   // next step's weight gradients accumulate from zero
            if (shadow) { if (NT_AUX) __builtin_nontemporal_store(sa[u], s4 + k); else s4[k] = sa[u]; }   // bf16 operands of the next step's GEMMs
        }
    }
    for (; i < n4; i += stride) {
        f32x4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
        bf16x4 sa;
        update(pa, ga, ma, va, sa);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
        if (shadow) s4[i] = sa;
        if (zero_g) g4[i] = zero4;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gg = g[i] * clip;
            const float mm = m[i] * b1 + one_b1 * gg, vv = v[i] * b2 + one_b2 * gg * gg;
            m[i] = mm; v[i] = vv;
            const float pp = p[i] - lr_t * (mm / (sqrtf(vv) + eps) + wd * p[i]);
            p[i] = pp;
            if (shadow) shadow[i] = (bf16_t)pp;
            if (zero_g) g[i] = 0.f;
        }
}

__global__ void add_i64_kernel(int64_t* p, int64_t d) { p[0] += d; }

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[i * 4 + j] = from_f32<T>(v[j]);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = from_f32<T>(src[i]);
}

}  // namespace shg

using namespace shg;

extern "C" int shg_sumsq(const float* x, int64_t n, double* partial, int n_partial, float* out_norm, void* stream) {
    if (!x || !partial || !out_norm) return fail_arg("sumsq: null pointer");
    if (n < 0 || n_partial < 1 || n_partial > 65535) return fail_arg("sumsq: bad sizes");
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail_arg("sumsq: x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(n_partial), dim3(256), 0, st, x, n, partial);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, partial, n_partial, out_norm, (double*)nullptr);
    return check_launch("sumsq");
}

extern "C" int shg_sumsq_partial(const float* x, int64_t n, double* partial, int n_partial, void* stream) {
    if (!x || !partial) return fail_arg("sumsq_partial: null pointer");
    if (n < 0 || n_partial < 1 || n_partial > 65535) return fail_arg("sumsq_partial: bad sizes");
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail_arg("sumsq_partial: x must be 16-byte aligned");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(n_partial), dim3(256), 0, (hipStream_t)stream, x, n, partial);
    return check_launch("sumsq_partial");
}

extern "C" int shg_sumsq_final(const double* partial, int n_partial, double* extra, float* out_norm, void* stream) {
    if (!partial || !out_norm) return fail_arg("sumsq_final: null pointer");
    if (n_partial < 1 || n_partial > (1 << 20)) return fail_arg("sumsq_final: bad sizes");
    if (extra && (reinterpret_cast<uintptr_t>(extra) & 7)) return fail_arg("sumsq_final: extra must be 8-byte aligned");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_partial, out_norm, extra);
    return check_launch("sumsq_final");
}

extern "C" int shg_bertadam_arena(float* param, float* grad, float* m, float* v, void* shadow_bf16, int64_t n,
                                  const float* grad_norm, float max_norm, float lr, float warmup, int64_t t_total,
                                  float b1, float b2, float eps, float weight_decay, int64_t* step_state, int bump_step,
                                  void* stream) {
    if (!param || !grad || !m || !v || !step_state) return fail_arg("bertadam: null pointer");
    if (n < 0) return fail_arg("bertadam: negative n");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return fail_arg("bertadam: arenas must be 16-byte aligned");
    if (shadow_bf16 && (reinterpret_cast<uintptr_t>(shadow_bf16) & 7)) return fail_arg("bertadam: shadow must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (n > 0) {
        const int adam_mode = (int)tuning(TUNE_BERTADAM_MODE);
        const int64_t adam_blocks = std::max<int64_t>(1, tuning(TUNE_BERTADAM_BLOCKS));   // measured: 2 048 .. 8 192 blocks 4.3-4.5 TB/s, 16 384 4.8
        const int64_t blocks = std::min<int64_t>((n / 4 + 255) / 256 + 1, adam_blocks);
#define SHG_ADAM(U, NM, NA)                                                                                                   \
    hipLaunchKernelGGL((bertadam_kernel<U, NM, NA>), dim3((unsigned)blocks), dim3(256), 0, st, param, grad, m, v,           \
                       (bf16_t*)shadow_bf16, n, grad_norm, max_norm, lr, warmup, t_total, b1, b2, eps, weight_decay, step_state, \
                       (bump_step >> 1) & 1)
        const int unroll = (adam_mode & 8) ? 4 : ((adam_mode & 1) ? 2 : 1);
        const bool nt_main = !(adam_mode & 4), nt_aux = (adam_mode & 2) != 0;
        if (unroll == 4) { if (nt_main) { if (nt_aux) SHG_ADAM(4, true, true); else SHG_ADAM(4, true, false); } else SHG_ADAM(4, false, false); }
        else if (unroll == 2) { if (nt_main) { if (nt_aux) SHG_ADAM(2, true, true); else SHG_ADAM(2, true, false); } else SHG_ADAM(2, false, false); }
        else SHG_ADAM(1, false, false);
#undef SHG_ADAM
    }
    if (bump_step & 1) hipLaunchKernelGGL(add_i64_kernel, dim3(1), dim3(1), 0, st, step_state, (int64_t)1);
    return check_launch("bertadam_arena");
}

extern "C" int shg_add_i64(int64_t* p, int64_t delta, void* stream) {
    if (!p) return fail_arg("add_i64: null pointer");
    hipLaunchKernelGGL(add_i64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, delta);
    return check_launch("add_i64");
}

extern "C" int shg_cast_f32(const float* src, void* dst, int dtype, int64_t n, void* stream) {
    if (!src || !dst) return fail_arg("cast: null pointer");
    if (n < 0) return fail_arg("cast: negative n");
    if (n == 0) return 0;
    if (reinterpret_cast<uintptr_t>(src) & 15) return fail_arg("cast: src must be 16-byte aligned");
    const int64_t blocks = std::min<int64_t>((n / 4 + 255) / 256 + 1, 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SHG_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, src, (float*)dst, n);
    else if (dtype == SHG_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, src, (bf16_t*)dst, n);
    else return fail_arg("cast: bad dtype");
    return check_launch("cast_f32");
}

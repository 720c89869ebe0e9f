// Train-mode BatchNorm2d for the ResNet path (torchvision Bottleneck BNs and the decoder BNs,
// /root/reference/dream/models.py:22-32, 37-136; nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1, affine,
// running stats tracked).  NHWC fp32: channel c of pixel p is x[p*C + c].  All kernels are HBM-bound
// streaming kernels (float4 per lane = 4 channels of one pixel, consecutive lanes on consecutive
// channels/pixels); per-channel reductions keep fp64 partial sums per workgroup and are combined in a
// fixed order by a finalize kernel (deterministic, no atomics).
//
//   forward :  mean, biased var over the N = B*H*W pixels;  y = (x-mean)*invstd*gamma + beta (+res) (ReLU)
//              running_mean/var <- (1-m)*running + m*batch (unbiased var), num_batches_tracked += 1
//   backward:  g = dy (* (y>0) when a ReLU follows);  dbeta = sum g;  dgamma = sum g * xhat;
//              dx = gamma*invstd * (g - dbeta/N - xhat*dgamma/N);   g is also returned (the gradient of the
//              Bottleneck identity branch).
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

constexpr int kStatBlocks = 512;      // partial-sum rows

// partials[blk][c][0..1] (double): sums of v0 and v1 over the pixels this workgroup visited
// MODE 0: v0 = x, v1 = x*x            MODE 1: v0 = g, v1 = g * xhat
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float *x, const float *dy, const float *y_act,
                                                        const float *mean, const float *invstd, double *partials,
                                                        size_t npix, int C, int relu) {
    // thread layout: lanes over channel quads first (coalesced), then pixels
    const int C4 = C >> 2;
    const int tpp = C4 < 256 ? C4 : 256;                  // threads per pixel row (C4 <= 256 handled; larger C loops)
    const int rows_per_block = 256 / tpp;
    const int cq = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    DREAM_DYNAMIC_LDS(double, sred);                      // [rows_per_block][tpp][4][2], reused per channel loop
    for (int c0 = cq; c0 < C4; c0 += tpp) {
        double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
        f32x4 mu = {0, 0, 0, 0}, is = {1, 1, 1, 1};
        if (MODE == 1) { mu = ((const f32x4 *)mean)[c0]; is = ((const f32x4 *)invstd)[c0]; }
        if (prow < rows_per_block) {
            // one pixel's terms, added in pixel order (the summation order -- hence the result, bit for bit -- does not depend
            // on how many loads are in flight)
            auto add = [&](const f32x4 v, f32x4 g, const f32x4 ya) {
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a0[k] += (double)v[k]; a1[k] += (double)v[k] * (double)v[k]; }
                } else {
                    if (relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) g[k] = ya[k] > 0.0f ? g[k] : 0.0f;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xh = (v[k] - mu[k]) * is[k];
                        a0[k] += (double)g[k];
                        a1[k] += (double)g[k] * (double)xh;
                    }
                }
            };
            const size_t stride = (size_t)gridDim.x * rows_per_block;
            size_t p = (size_t)blockIdx.x * rows_per_block + prow;
            // four pixels per trip: a workgroup has one load per thread and trip in flight otherwise, and 2 workgroups per CU
            // cover only ~1 TB/s of the memory latency
            for (; p + 3 * stride < npix; p += 4 * stride) {
                f32x4 v[4], g[4], ya[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t i = (p + u * stride) * C4 + c0;
                    v[u] = ((const f32x4 *)x)[i];
                    g[u] = MODE == 1 ? ((const f32x4 *)dy)[i] : v[u];
                    ya[u] = (MODE == 1 && relu) ? ((const f32x4 *)y_act)[i] : v[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) add(v[u], g[u], ya[u]);
            }
            for (; p < npix; p += stride) {
                const size_t i = p * C4 + c0;
                const f32x4 v = ((const f32x4 *)x)[i];
                add(v, MODE == 1 ? ((const f32x4 *)dy)[i] : v, (MODE == 1 && relu) ? ((const f32x4 *)y_act)[i] : v);
            }
        }
        // reduce the rows_per_block rows of this workgroup
        __syncthreads();
        if (prow < rows_per_block) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sred[((prow * tpp + cq) * 4 + k) * 2 + 0] = a0[k];
                sred[((prow * tpp + cq) * 4 + k) * 2 + 1] = a1[k];
            }
        }
        __syncthreads();
        if (prow == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double s0 = 0, s1 = 0;
                for (int r = 0; r < rows_per_block; ++r) {
                    s0 += sred[((r * tpp + cq) * 4 + k) * 2 + 0];
                    s1 += sred[((r * tpp + cq) * 4 + k) * 2 + 1];
                }
                partials[((size_t)blockIdx.x * C + c0 * 4 + k) * 2 + 0] = s0;
                partials[((size_t)blockIdx.x * C + c0 * 4 + k) * 2 + 1] = s1;
            }
        }
    }
}

// finalize kernels: 64 channels per workgroup, kSlices waves each summing every kSlices-th partial row (coalesced over
// channels), then a fixed-order combine through LDS -> deterministic, and the dependent load->add chain per thread is
// <= 512/16 = 32 long (these tiny kernels run 216 times per ResNet-101 step: their latency, not bandwidth, matters)
constexpr int kSlices = 16;
DREAM_DEVICE void sum_partials(const double *partials, int nblk, int C, int c, int slice, double *s, double *ss) {
    double a = 0, b = 0;
    if (c < C) {
        const double2 *src = (const double2 *)partials + c;
        int r = slice;
        for (; r + 3 * kSlices < nblk; r += 4 * kSlices) {        // four independent loads in flight
            const double2 v0 = src[(size_t)r * C], v1 = src[(size_t)(r + kSlices) * C];
            const double2 v2 = src[(size_t)(r + 2 * kSlices) * C], v3 = src[(size_t)(r + 3 * kSlices) * C];
            a += v0.x; b += v0.y;
            a += v1.x; b += v1.y;
            a += v2.x; b += v2.y;
            a += v3.x; b += v3.y;
        }
        for (; r < nblk; r += kSlices) { const double2 v = src[(size_t)r * C]; a += v.x; b += v.y; }
    }
    __shared__ double red[2][kSlices][64];
    const int l = threadIdx.x & 63;
    red[0][slice][l] = a;
    red[1][slice][l] = b;
    __syncthreads();
    double t0 = 0, t1 = 0;
#pragma unroll
    for (int k = 0; k < kSlices; ++k) { t0 += red[0][k][l]; t1 += red[1][k][l]; }
    *s = t0;
    *ss = t1;
}

__global__ void __launch_bounds__(64 * kSlices) bn_stats_finalize_kernel(const double *partials, int nblk, int C, double n, float eps,
                                                                float momentum, float *mean, float *invstd,
                                                                float *running_mean, float *running_var,
                                                                long long *num_batches_tracked) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double s, ss;
    sum_partials(partials, nblk, C, c, slice, &s, &ss);
    if (slice == 0 && c < C) {
        const double m = s / n;
        double var = ss / n - m * m;
        if (var < 0) var = 0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = n > 1 ? var * n / (n - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

__global__ void __launch_bounds__(64 * kSlices) bn_bwd_finalize_kernel(const double *partials, int nblk, int C, float *dgamma, float *dbeta) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double s, ss;
    sum_partials(partials, nblk, C, c, slice, &s, &ss);
    if (slice == 0 && c < C) {
        dbeta[c] = (float)s;
        dgamma[c] = (float)ss;
    }
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const f32x4 *x, const f32x4 *mean, const f32x4 *invstd, const f32x4 *gamma,
                                                       const f32x4 *beta, const f32x4 *residual, f32x4 *y, size_t n4, int C4, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const f32x4 v = x[i], mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (v[k] - mu[k]) * is[k] * ga[k] + be[k];
        if (residual) {
            const f32x4 r = residual[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += r[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.0f);
        }
        y[i] = o;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const f32x4 *x, const f32x4 *dy, const f32x4 *y_act, const f32x4 *mean,
                                                           const f32x4 *invstd, const f32x4 *gamma, const f32x4 *dgamma,
                                                           const f32x4 *dbeta, f32x4 *dx, f32x4 *g_out, size_t n4, int C4,
                                                           float inv_n, int relu) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const f32x4 v = x[i], mu = mean[c], is = invstd[c], ga = gamma[c], dg = dgamma[c], db = dbeta[c];
        f32x4 g = dy[i];
        if (relu) {
            const f32x4 ya = y_act[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = ya[k] > 0.0f ? g[k] : 0.0f;
        }
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (v[k] - mu[k]) * is[k];
            o[k] = ga[k] * is[k] * (g[k] - db[k] * inv_n - xh * dg[k] * inv_n);
        }
        dx[i] = o;
        if (g_out) g_out[i] = g;
    }
}

// partial-sum workgroups: enough to fill the chip on big tensors, few on small ones (>= 16 float4 per thread)
inline size_t stat_blocks(size_t npix, int C4, int rows) {
    size_t by_rows = (npix + rows - 1) / rows;
    size_t by_work = (npix * (size_t)C4 + 256 * 16 - 1) / (256 * 16);
    size_t nb = by_rows < by_work ? by_rows : by_work;
    if (nb > (size_t)kStatBlocks) nb = kStatBlocks;
    return nb ? nb : 1;
}

inline unsigned stream_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    return (unsigned)(g ? g : 1);
}

}  // namespace

extern "C" size_t dream_bn_workspace(int C) { return (size_t)kStatBlocks * C * 2 * sizeof(double) + (size_t)C * sizeof(float); }

extern "C" int dream_bn_train_fwd_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *residual,
                                           float *y, float *save_mean, float *save_invstd, float *running_mean,
                                           float *running_var, long long *num_batches_tracked, void *workspace,
                                           size_t npix, int C, float eps, float momentum, int relu, void *stream) {
    DREAM_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "bn_train_fwd: null pointer");
    DREAM_REQUIRE(C > 0 && C % 4 == 0 && (C <= 1024 || C % 1024 == 0) && npix > 0, "bn_train_fwd: unsupported C=%d", C);
    const int C4 = C / 4, tpp = C4 < 256 ? C4 : 256, rows = 256 / tpp;
    size_t nb = stat_blocks(npix, C4, rows);
    const size_t lds = (size_t)rows * tpp * 8 * sizeof(double);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, x, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (double *)workspace, npix, C, 0);
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * kSlices), 0, (hipStream_t)stream,
                       (const double *)workspace, (int)nb, C, (double)npix, eps, momentum, save_mean, save_invstd,
                       running_mean, running_var, num_batches_tracked);
    DREAM_LAUNCH_OK();
    const size_t n4 = npix * C4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x,
                       (const f32x4 *)save_mean, (const f32x4 *)save_invstd, (const f32x4 *)gamma, (const f32x4 *)beta,
                       (const f32x4 *)residual, (f32x4 *)y, n4, C4, relu);
    DREAM_LAUNCH_OK();
    return 0;
}

// per-channel sum over pixels (bias gradient of a conv whose output feeds a BatchNorm): out[c] = sum_p x[p][c]
extern "C" int dream_channel_sum_nhwc_f32(const float *x, float *out, void *workspace, size_t npix, int C, void *stream) {
    DREAM_REQUIRE(x && out && workspace && C > 0 && C % 4 == 0 && (C <= 1024 || C % 1024 == 0) && npix > 0, "channel_sum: bad arguments");
    const int C4 = C / 4, tpp = C4 < 256 ? C4 : 256, rows = 256 / tpp;
    size_t nb = stat_blocks(npix, C4, rows);
    const size_t lds = (size_t)rows * tpp * 8 * sizeof(double);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, x, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (double *)workspace, npix, C, 0);
    DREAM_LAUNCH_OK();
    // dbeta slot (sum of v0) is what we want; dgamma slot (sum of squares) goes to a scratch tail of the workspace
    float *scratch = (float *)((double *)workspace + (size_t)kStatBlocks * C * 2);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * kSlices), 0, (hipStream_t)stream,
                       (const double *)workspace, (int)nb, C, scratch, out);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_bn_train_bwd_nhwc_f32(const float *x, const float *dy, const float *y_act, const float *gamma,
                                           const float *save_mean, const float *save_invstd, float *dx, float *g_out,
                                           float *dgamma, float *dbeta, void *workspace, size_t npix, int C, int relu,
                                           void *stream) {
    DREAM_REQUIRE(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "bn_train_bwd: null pointer");
    DREAM_REQUIRE(!relu || y_act, "bn_train_bwd: relu needs the activation output");
    DREAM_REQUIRE(C > 0 && C % 4 == 0 && (C <= 1024 || C % 1024 == 0) && npix > 0, "bn_train_bwd: unsupported C=%d", C);
    const int C4 = C / 4, tpp = C4 < 256 ? C4 : 256, rows = 256 / tpp;
    size_t nb = stat_blocks(npix, C4, rows);
    const size_t lds = (size_t)rows * tpp * 8 * sizeof(double);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, x, dy, y_act, save_mean,
                       save_invstd, (double *)workspace, npix, C, relu);
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * kSlices), 0, (hipStream_t)stream,
                       (const double *)workspace, (int)nb, C, dgamma, dbeta);
    DREAM_LAUNCH_OK();
    const size_t n4 = npix * C4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x,
                       (const f32x4 *)dy, (const f32x4 *)y_act, (const f32x4 *)save_mean, (const f32x4 *)save_invstd,
                       (const f32x4 *)gamma, (const f32x4 *)dgamma, (const f32x4 *)dbeta, (f32x4 *)dx, (f32x4 *)g_out, n4, C4,
                       (float)(1.0 / (double)npix), relu);
    DREAM_LAUNCH_OK();
    return 0;
}

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
#include "stat_tree.h"
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


// ------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same BatchNorm with fewer passes over memory and fewer launches (a ResNet-101 step at 16 frames per GPU spent 24 %
// of its time in 656 BatchNorm launches, most of them on 10-MB tensors where a launch costs more than its traffic).
//   * the per-channel statistics are FINISHED INSIDE the launch that sums them: workgroup = (block of pixels) x (slab of 64
//     channels); it writes one row of fp64 partial sums for its slab and arrives at the slab's two-level ticket tree
//     (stat_tree.h): the last arriver of a group of rows adds the group, the last group adds the groups -- fixed order
//     (deterministic), no finalize launch, no fence, nobody spins;
//   * the forward statistics are published as the affine map  y = a z + b  (a = gamma * invstd, b = beta - mean * a), so that a
//     consumer can apply BatchNorm + ReLU while LOADING z (gemm1x1.hip: PRE) and the normalised tensor is never written; where it
//     must exist (Bottleneck outputs: the residual sum) ONE kernel writes it from (z, a, b);
//   * producers that can, sum the statistics in their own epilogue (gemm1x1.hip: EPI 1 / 2) and skip the reduce launch too.
// Every kernel evaluates BatchNorm + ReLU as max(fmaf(a, z, b), 0): the backward pass recomputes exactly the forward's mask.

constexpr int kStatRows = 256;        // at most this many partial rows per launch

struct BnStatParams {
    const float *z, *dy, *y_act;      // MODE 1: dy = incoming gradient; mask 1: y_act > 0, mask 2: fmaf(a, z, b) > 0
    const float *ab;                  // [2][C] (mask 2)
    const float *mean, *invstd;       // MODE 1: of the BatchNorm
    StatTree st;                      // rows = workgroups along x
    size_t npix;
    int C, mask;
    BnFwdOut fwd;                     // MODE 0 outputs
    float *dgamma, *dbeta;            // MODE 1 outputs
};

// MODE 0: v0 = z, v1 = z * z          MODE 1: v0 = g, v1 = g * xhat   (g = dy masked)
template <int MODE>
__global__ void __launch_bounds__(256) bn_stats_slab_kernel(const BnStatParams p) {
    __shared__ double sred[16 * 16 * 8];              // [pixel row][channel quad][4 channels][2]
    const int cq = threadIdx.x & 15, prow = threadIdx.x >> 4;
    const int slab = blockIdx.y, C4 = p.C >> 2;
    const int c4 = slab * 16 + cq;                    // this thread's channel quad
    const bool cok = c4 < C4;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    f32x4 mu = {0, 0, 0, 0}, is = {1, 1, 1, 1}, za = {0, 0, 0, 0}, zb = {0, 0, 0, 0};
    if (MODE == 1 && cok) {
        mu = ((const f32x4 *)p.mean)[c4];
        is = ((const f32x4 *)p.invstd)[c4];
        if (p.mask == 2) { za = ((const f32x4 *)p.ab)[c4]; zb = ((const f32x4 *)(p.ab + p.C))[c4]; }
    }
    const int mask = p.mask;
    auto add = [&](const f32x4 v, f32x4 g, const f32x4 ya) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0[k] += (double)v[k]; a1[k] += (double)v[k] * (double)v[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool on = mask == 0 || (mask == 1 ? ya[k] > 0.0f : __builtin_fmaf(za[k], v[k], zb[k]) > 0.0f);
                const float gk = on ? g[k] : 0.0f;
                const float xh = (v[k] - mu[k]) * is[k];
                a0[k] += (double)gk;
                a1[k] += (double)gk * (double)xh;
            }
        }
    };
    if (cok) {
        const size_t stride = (size_t)gridDim.x * 16;
        size_t px = (size_t)blockIdx.x * 16 + prow;
        for (; px + 3 * stride < p.npix; px += 4 * stride) {          // four pixels per trip in flight, added in pixel order
            f32x4 v[4], g[4], ya[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = (px + u * stride) * C4 + c4;
                v[u] = ((const f32x4 *)p.z)[i];
                g[u] = MODE == 1 ? ((const f32x4 *)p.dy)[i] : v[u];
                ya[u] = (MODE == 1 && mask == 1) ? ((const f32x4 *)p.y_act)[i] : v[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u], g[u], ya[u]);
        }
        for (; px < p.npix; px += stride) {
            const size_t i = px * C4 + c4;
            const f32x4 v = ((const f32x4 *)p.z)[i];
            add(v, MODE == 1 ? ((const f32x4 *)p.dy)[i] : v, (MODE == 1 && mask == 1) ? ((const f32x4 *)p.y_act)[i] : v);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sred[((prow * 16 + cq) * 4 + k) * 2 + 0] = a0[k];
        sred[((prow * 16 + cq) * 4 + k) * 2 + 1] = a1[k];
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;                    // wavefront 0 publishes the workgroup's row and arrives at the tree
    if (prow == 0 && cok) {
        double *dst = p.st.rows + ((size_t)blockIdx.x * p.C + (size_t)c4 * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double s0 = 0, s1 = 0;
            for (int r = 0; r < 16; ++r) {
                s0 += sred[((r * 16 + cq) * 4 + k) * 2 + 0];
                s1 += sred[((r * 16 + cq) * 4 + k) * 2 + 1];
            }
            coherent_store(dst + 2 * k, s0);
            coherent_store(dst + 2 * k + 1, s1);
        }
    }
    double t0, t1;
    if (!stat_tree_arrive(p.st, slab, (int)blockIdx.x, (int)threadIdx.x, &t0, &t1)) return;
    const int c = slab * 64 + (int)threadIdx.x;
    if (MODE == 0) {
        bn_finish_forward(p.fwd, c, p.C, (double)p.npix, t0, t1);
    } else if (c < p.C) {
        p.dbeta[c] = (float)t0;
        p.dgamma[c] = (float)t1;
    }
}

// y = a z + b (+ residual) (ReLU) from the published affine map
__global__ void __launch_bounds__(256) bn_apply_ab_kernel(const f32x4 *z, const float *ab, const f32x4 *residual, f32x4 *y, size_t n4,
                                                          int C4, int relu) {
    const f32x4 *a4 = (const f32x4 *)ab, *b4 = (const f32x4 *)(ab + 4 * (size_t)C4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const f32x4 v = z[i], a = a4[c], b = b4[c];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(a[k], v[k], b[k]);
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

// dz = gamma * invstd * (g - dbeta / N - xhat * dgamma / N);  g = dy masked (mask 0: as is, 1: y_act > 0, 2: fmaf(a, z, b) > 0)
__global__ void __launch_bounds__(256) bn_bwd_apply_mask_kernel(const f32x4 *z, const f32x4 *dy, const f32x4 *y_act, const float *ab,
                                                                const f32x4 *mean, const f32x4 *invstd, const f32x4 *gamma,
                                                                const f32x4 *dgamma, const f32x4 *dbeta, f32x4 *dz, f32x4 *g_out,
                                                                size_t n4, int C4, float inv_n, int mask) {
    const f32x4 *a4 = (const f32x4 *)ab, *b4 = (const f32x4 *)(ab + 4 * (size_t)C4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const f32x4 v = z[i], mu = mean[c], is = invstd[c], ga = gamma[c], dg = dgamma[c], db = dbeta[c];
        f32x4 g = dy[i];
        if (mask == 1) {
            const f32x4 ya = y_act[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = ya[k] > 0.0f ? g[k] : 0.0f;
        } else if (mask == 2) {
            const f32x4 a = a4[c], b = b4[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = __builtin_fmaf(a[k], v[k], b[k]) > 0.0f ? g[k] : 0.0f;
        }
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (v[k] - mu[k]) * is[k];
            o[k] = ga[k] * is[k] * (g[k] - db[k] * inv_n - xh * dg[k] * inv_n);
        }
        dz[i] = o;
        if (g_out) g_out[i] = g;
    }
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

// ---- round 4 entry points (see the kernels above) ----------------------------------------------------------------------------------
extern "C" size_t dream_bn_stats_workspace(int C) { return stat_tree_doubles(kStatRows, C) * sizeof(double); }
extern "C" int dream_bn_stats_counters(int C) { return stat_tree_counters(kStatRows, C); }

static int g_stat_px_per_row = 128;     // A/B hook (dream_bn_stats_set_pixels_per_row): pixels a workgroup sums into its partial row
static inline int host_stat_rows(size_t npix) {
    size_t nb = (npix + g_stat_px_per_row - 1) / g_stat_px_per_row;
    return (int)(nb < 1 ? 1 : (nb > (size_t)kStatRows ? (size_t)kStatRows : nb));
}
extern "C" int dream_bn_stats_set_pixels_per_row(int px) {
    DREAM_REQUIRE(px >= 16 && px <= (1 << 20), "bn_stats: %d pixels per partial row", px);
    g_stat_px_per_row = px;
    return 0;
}

// Batch statistics of z [npix][C] in ONE launch: save_mean, save_invstd, out_ab = (gamma * invstd, beta - mean * gamma * invstd),
// running statistics updated (nn.BatchNorm2d, train mode).  counters: dream_bn_stats_counters(C) zero words, left zero.
extern "C" int dream_bn_stats_nhwc_f32(const float *z, const float *gamma, const float *beta, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, float eps, float momentum, float *out_ab, float *save_mean,
                                       float *save_invstd, void *workspace, unsigned *counters, size_t npix, int C, void *stream) {
    DREAM_REQUIRE(z && gamma && beta && out_ab && save_mean && save_invstd && workspace && counters, "bn_stats: null pointer");
    DREAM_REQUIRE(C > 0 && C % 4 == 0 && npix > 0, "bn_stats: unsupported C=%d", C);
    BnStatParams p = {};
    const int nrows = host_stat_rows(npix);
    p.z = z; p.st = stat_tree_make(workspace, counters, nrows, C); p.npix = npix; p.C = C;
    p.fwd.gamma = gamma; p.fwd.beta = beta; p.fwd.running_mean = running_mean; p.fwd.running_var = running_var;
    p.fwd.nbt = num_batches_tracked; p.fwd.eps = eps; p.fwd.momentum = momentum;
    p.fwd.ab = out_ab; p.fwd.mean = save_mean; p.fwd.invstd = save_invstd;
    hipLaunchKernelGGL(bn_stats_slab_kernel<0>, dim3((unsigned)nrows, (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

// y = ab[0] z + ab[1] (+ residual) (ReLU)
extern "C" int dream_bn_apply_ab_nhwc_f32(const float *z, const float *ab, const float *residual, float *y, size_t npix, int C,
                                          int relu, void *stream) {
    DREAM_REQUIRE(z && ab && y && C > 0 && C % 4 == 0 && npix > 0, "bn_apply_ab: bad arguments");
    const size_t n4 = npix * (size_t)(C / 4);
    hipLaunchKernelGGL(bn_apply_ab_kernel, dim3(stream_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)z, ab,
                       (const f32x4 *)residual, (f32x4 *)y, n4, C / 4, relu);
    DREAM_LAUNCH_OK();
    return 0;
}

// dbeta = sum g, dgamma = sum g * xhat in ONE launch; g = dy masked: mask 0 none, 1 y_act > 0, 2 ab[0] z + ab[1] > 0
extern "C" int dream_bn_bwd_stats_nhwc_f32(const float *z, const float *dy, const float *y_act, const float *ab, const float *save_mean,
                                           const float *save_invstd, float *dgamma, float *dbeta, void *workspace, unsigned *counters,
                                           size_t npix, int C, int mask, void *stream) {
    DREAM_REQUIRE(z && dy && save_mean && save_invstd && dgamma && dbeta && workspace && counters, "bn_bwd_stats: null pointer");
    DREAM_REQUIRE(mask == 0 || (mask == 1 && y_act) || (mask == 2 && ab), "bn_bwd_stats: mask %d without its source", mask);
    DREAM_REQUIRE(C > 0 && C % 4 == 0 && npix > 0, "bn_bwd_stats: unsupported C=%d", C);
    BnStatParams p = {};
    const int nrows = host_stat_rows(npix);
    p.z = z; p.dy = dy; p.y_act = y_act; p.ab = ab; p.mean = save_mean; p.invstd = save_invstd;
    p.st = stat_tree_make(workspace, counters, nrows, C); p.npix = npix; p.C = C; p.mask = mask;
    p.dgamma = dgamma; p.dbeta = dbeta;
    hipLaunchKernelGGL(bn_stats_slab_kernel<1>, dim3((unsigned)nrows, (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

// dz = gamma * invstd * (g - dbeta / N - xhat * dgamma / N), g = dy masked as above; g_out (optional) receives g
extern "C" int dream_bn_bwd_apply_nhwc_f32(const float *z, const float *dy, const float *y_act, const float *ab, const float *gamma,
                                           const float *save_mean, const float *save_invstd, const float *dgamma, const float *dbeta,
                                           float *dz, float *g_out, size_t npix, int C, int mask, void *stream) {
    DREAM_REQUIRE(z && dy && gamma && save_mean && save_invstd && dgamma && dbeta && dz, "bn_bwd_apply: null pointer");
    DREAM_REQUIRE(mask == 0 || (mask == 1 && y_act) || (mask == 2 && ab), "bn_bwd_apply: mask %d without its source", mask);
    DREAM_REQUIRE(C > 0 && C % 4 == 0 && npix > 0, "bn_bwd_apply: unsupported C=%d", C);
    const size_t n4 = npix * (size_t)(C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_mask_kernel, dim3(stream_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)z,
                       (const f32x4 *)dy, (const f32x4 *)y_act, ab ? ab : gamma, (const f32x4 *)save_mean, (const f32x4 *)save_invstd,
                       (const f32x4 *)gamma, (const f32x4 *)dgamma, (const f32x4 *)dbeta, (f32x4 *)dz, (f32x4 *)g_out, n4, C / 4,
                       (float)(1.0 / (double)npix), mask);
    DREAM_LAUNCH_OK();
    return 0;
}

// HBM-bound helpers of the belief-map network: 2x2 max-pool (+backward), nearest-x2 upsample
// backward, ReLU backward, NCHW<->NHWC layout changes, weight (un)packing, fused MSE loss +
// gradient, Adam / SGD.  All are streaming kernels: 16-byte accesses per lane, consecutive lanes on
// consecutive addresses, grid-stride loops capped at 256 CUs x 8 workgroups.
//
// Reference call sites: nn.MaxPool2d(2) dream/models.py:589,765-771; nn.Upsample dream/models.py:691,703;
// nn.ReLU(inplace) throughout models.py; torch.nn.MSELoss dream/network.py:260-261,359;
// torch.optim.Adam / SGD dream/network.py:666-685.
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

constexpr int kMaxBlocks = 256 * 8;

inline unsigned grid_for(size_t work_items, int block = 256) {
    size_t g = ceil_div_sz(work_items, (size_t)block);
    if (g > (size_t)kMaxBlocks) g = kMaxBlocks;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- max-pool 2x2, NHWC, channels in float4 groups ------------------------------------------------
__global__ void __launch_bounds__(256) maxpool2_kernel(const f32x4 *x, f32x4 *y, int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const f32x4 *s = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C4 + c;
        const f32x4 v00 = s[0], v01 = s[C4], v10 = s[(size_t)W * C4], v11 = s[(size_t)W * C4 + C4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
        y[i] = o;
    }
}

// backward: the gradient goes to the first maximal element in (row-major) window order, which is
// what ATen's max_pool2d_with_indices records.  Rows/cols beyond 2*floor(H/2) get zero.
// One thread per 2x2 WINDOW (and channel quad): its four inputs are loaded once, the window's gradient goes to the first maximum, the
// other three outputs are zeros (round 4: one thread per INPUT pixel loaded every window four times over: 3.9 TB/s on 64 x 400 x 400).
// The thread of the last window of a row / column also zeroes the odd extent's leftover column / row.
template <bool RELU>
__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const f32x4 *dy, const f32x4 *x, f32x4 *dx,
                                                           int B, int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const size_t base = (((size_t)b * H + 2 * oy) * W + 2 * ox) * C4 + c, row = (size_t)W * C4;
        const f32x4 v[4] = {x[base], x[base + C4], x[base + row], x[base + row + C4]};
        const f32x4 g = dy[i];
        f32x4 o[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int arg = 0;
            float best = v[0][k];
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (v[j][k] > best) { best = v[j][k]; arg = j; }
            // RELU: x is a ReLU output, and the gradient continues through that ReLU (mask x > 0) in the same pass
            const float gk = (!RELU || best > 0.0f) ? g[k] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j][k] = arg == j ? gk : 0.0f;
        }
        dx[base] = o[0];
        dx[base + C4] = o[1];
        dx[base + row] = o[2];
        dx[base + row + C4] = o[3];
        const bool last_x = (ox == Wo - 1) && (W & 1), last_y = (oy == Ho - 1) && (H & 1);
        if (last_x) { dx[base + 2 * C4] = zero; dx[base + row + 2 * C4] = zero; }
        if (last_y) { dx[base + 2 * row] = zero; dx[base + 2 * row + C4] = zero; }
        if (last_x && last_y) dx[base + 2 * row + 2 * C4] = zero;
    }
}

// Every second pixel of every second row (the positions a stride-2 1x1 conv reads): y[b,i,j,:] = x[b,2i,2j,:], Ho = (H + 1) / 2.  The
// ResNet trunk's three stride-2 downsample convs (dream/models.py:22-32 -> torchvision Bottleneck.downsample) then run as plain GEMMs
// over the gathered rows (round 6) -- and the transpose: x[b,y,x,:] = ys[b,y/2,x/2,:] at even (y, x), zeros elsewhere (their data
// gradient back on the block input's grid).
__global__ void __launch_bounds__(256) subsample2_kernel(const f32x4 *x, f32x4 *y, int B, int H, int W, int C4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        y[i] = x[(((size_t)b * H + 2 * oy) * W + 2 * ox) * C4 + c];
    }
}
__global__ void __launch_bounds__(256) scatter2_kernel(const f32x4 *ys, f32x4 *x, int B, int H, int W, int C4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t total = (size_t)B * H * W * C4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int px = (int)(r % W);
        r /= W;
        const int py = (int)(r % H);
        const int b = (int)(r / H);
        x[i] = ((px | py) & 1) ? zero : ys[(((size_t)b * Ho + (py >> 1)) * Wo + (px >> 1)) * C4 + c];
    }
}

// The patches a 3x3 stride-2 pad-1 conv reads, one row of 9 C columns per output pixel: col[(b,i,j)][t C + c] = x[b, 2i - 1 + ky, 2j - 1 + kx, c],
// t = 3 ky + kx, zeros outside the image; Ho = (H - 1) / 2 + 1.  On small maps (ResNet-101's layer4.0.conv2 at 16 frames: 2704 output pixels)
// the direct kernel cannot fill the chip (36 TFLOP/s); as a GEMM over these rows the conv, its weight gradient and its data gradient run
// on the 1x1 GEMM kernels.  col2im3s2 is the transpose: dx[b,y,x,:] = sum of the (one, two or four) patch entries that read pixel (y, x),
// added in a fixed order (ky, then kx).
__global__ void __launch_bounds__(256) im2col3s2_kernel(const f32x4 *x, f32x4 *col, int B, int H, int W, int C4) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * 9 * C4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int t = (int)(r % 9);
        r /= 9;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int py = 2 * oy - 1 + t / 3, px = 2 * ox - 1 + t % 3;
        col[i] = (py >= 0 && py < H && px >= 0 && px < W) ? x[(((size_t)b * H + py) * W + px) * C4 + c] : zero;
    }
}
__global__ void __launch_bounds__(256) col2im3s2_kernel(const f32x4 *col, f32x4 *dx, int B, int H, int W, int C4) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int px = (int)(r % W);
        r /= W;
        const int py = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = py + 1 - ky;                       // = 2 oy
            if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = px + 1 - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
                acc = acc + col[((((size_t)b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * 9 + 3 * ky + kx) * C4 + c];
            }
        }
        dx[i] = acc;
    }
}

// ConvTranspose2d(k4, s2, p1) as a GEMM + this gather (round 6, small maps: ResNet-101's first decoder layer, 2048 -> 256 on 13 x 13 maps at 16
// frames, is 2704 input pixels -- too few tiles for the Winograd kernel to fill the chip): g[b,i,j][(4 ky + kx) C + c] = sum over ci of
// x[b,i,j,ci] w[ci][c][ky][kx] is a 1x1 GEMM with N = 16 C, and output pixel (Y, X) = (2 i - 1 + ky, 2 j - 1 + kx) collects its (at most
// four) contributions: z[b,Y,X,c] = bias[c] + the sum over ky, kx (ascending) of the entries with (Y + 1 - ky) and (X + 1 - kx) even
// (training: scale == null, shift = the bias); with a scale (evaluation: the folded BatchNorm) z = sum * scale + shift, then the ReLU.
__global__ void __launch_bounds__(256) col2im4s2_kernel(const f32x4 *g, const f32x4 *scale, const f32x4 *bias, f32x4 *z, int B, int H, int W,
                                                        int C4, int relu) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int X = (int)(r % Wo);
        r /= Wo;
        const int Y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 acc = (bias && !scale) ? bias[c] : zero;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ((Y + 1) & 1) + 2 * a, iy = (Y + 1 - ky) >> 1;
            if (Y + 1 - ky < 0 || iy >= H) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kx = ((X + 1) & 1) + 2 * e, ix = (X + 1 - kx) >> 1;
                if (X + 1 - kx < 0 || ix >= W) continue;
                acc = acc + g[((((size_t)b * H + iy) * W + ix) * 16 + 4 * ky + kx) * C4 + c];
            }
        }
        if (scale) acc = acc * scale[c] + (bias ? bias[c] : zero);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        }
        z[i] = acc;
    }
}

// nearest x2 upsample backward: dx[b,y,x,:] = sum of the 2x2 block of dy
__global__ void __launch_bounds__(256) upsample2_bwd_kernel(const f32x4 *dy, f32x4 *dx, int B, int H, int W, int C4) {
    const int Hs = H / 2, Ws = W / 2;
    const size_t total = (size_t)B * Hs * Ws * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int sx = (int)(r % Ws);
        r /= Ws;
        const int sy = (int)(r % Hs);
        const int b = (int)(r / Hs);
        const f32x4 *s = dy + (((size_t)b * H + 2 * sy) * W + 2 * sx) * C4 + c;
        const f32x4 a = s[0], bb = s[C4], cc = s[(size_t)W * C4], d = s[(size_t)W * C4 + C4];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (a[k] + bb[k]) + (cc[k] + d[k]);
        dx[i] = o;
    }
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(const float *dy, const float *y, float *dx, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 g = ((const f32x4 *)dy)[i], v = ((const f32x4 *)y)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.0f ? g[k] : 0.0f;
        ((f32x4 *)dx)[i] = o;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dx[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// ---- layout: [B,C,H,W] <-> [B,H,W,C] through a 64x(C<=64) LDS tile ----------------------------------
// One workgroup moves 64 consecutive pixels of one image x up to 64 channels; reads are coalesced
// along pixels (NCHW side) and writes along channels (NHWC side), or the reverse.
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float *x, float *y, int C, int HW) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < C && p0 + tx < HW) tile[c][tx] = x[((size_t)b * C + c0 + c) * HW + p0 + tx];
    __syncthreads();
    for (int pp = ty; pp < 64; pp += 4)
        if (c0 + tx < C && p0 + pp < HW) y[((size_t)b * HW + p0 + pp) * C + c0 + tx] = tile[tx][pp];
}
// [B,C,H,W] -> [B,H,W,Cpad] with zero fill for c >= C (C, Cpad small: one thread per (pixel, c))
__global__ void __launch_bounds__(256) nchw_to_nhwc_pad_kernel(const float *x, float *y, int B, int C, int HW, int Cpad) {
    const size_t total = (size_t)B * HW * Cpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const size_t r = i / Cpad;
        const size_t pix = r % HW, b = r / HW;
        y[i] = c < C ? x[(b * C + c) * HW + pix] : 0.0f;
    }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float *x, float *y, int C, int HW) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int pp = ty; pp < 64; pp += 4)
        if (c0 + tx < C && p0 + pp < HW) tile[pp][tx] = x[((size_t)b * HW + p0 + pp) * C + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < C && p0 + tx < HW) y[((size_t)b * C + c0 + c) * HW + p0 + tx] = tile[tx][c];
}

// ---- weight packing ------------------------------------------------------------------------------
// w is the OIHW tensor [Cout][Cin][3][3] as torch stores it; t = ky*3 + kx.
// mode 0 (forward operator):        packed[t][r][c] = w[o=r][i=c][t]        r < Cout, c < Cin
// mode 1 (data-gradient operator):  packed[t][r][c] = w[o=c][i=r][8 - t]   r < Cin,  c < Cout
//   i.e. rows are the channels the conv kernel PRODUCES and columns the channels it CONSUMES;
//   flipping the taps turns correlation with dL/dy into the transposed convolution.
// packed is [9][RowsPad][ColsPad], zero padded.
__global__ void __launch_bounds__(256) pack_w_kernel(const float *w, float *packed, int Cout, int Cin,
                                                     int RowsPad, int ColsPad, int mode, int ntaps) {
    const size_t total = (size_t)ntaps * RowsPad * ColsPad;
    const int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % ColsPad);
        size_t q = idx / ColsPad;
        const int r = (int)(q % RowsPad);
        const int t = (int)(q / RowsPad);
        float v = 0.0f;
        if (r < rows && c < cols)
            v = (mode == 0) ? w[((size_t)r * Cin + c) * ntaps + t] : w[((size_t)c * Cin + r) * ntaps + (ntaps - 1 - t)];
        packed[idx] = v;
    }
}
__global__ void __launch_bounds__(256) unpack_w_kernel(const float *packed, float *w, int Cout, int Cin,
                                                       int CoutPad, int CinPad, int ntaps) {
    const size_t total = (size_t)Cout * Cin * ntaps;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx % ntaps);
        size_t r = idx / ntaps;
        const int i = (int)(r % Cin);
        const int o = (int)(r / Cin);
        w[idx] = packed[((size_t)t * CoutPad + o) * CinPad + i];
    }
}

// nn.Upsample(scale_factor=2) (nearest) followed by Conv2d(k3,s1,p1) (dream/models.py:691-710) IS a
// ConvTranspose2d(k4,s2,p1): every output pixel (2m+a, 2n+b) sees only the 2x2 input pixels around (m, n), each through
// the SUM of the 3x3 taps that land on it.  w3 OIHW [Cout,Cin,3,3] -> wT4 [Cin,Cout,4,4] (ConvTranspose layout) with
//   wT4[i][o][ky][kx] = sum_{r in R(ky)} sum_{c in R(kx)} w3[o][i][r][c],   R(0)={2}, R(1)={1,2}, R(2)={0,1}, R(3)={0}
// so the decoder's two upsample convs run 4 MACs per output instead of 9 (zero padding of the upsampled grid and of the
// input coincide at the borders).
__global__ void __launch_bounds__(256) upsample_conv_weight_kernel(const float *w3, float *wT4, int Cout, int Cin) {
    const size_t total = (size_t)Cin * Cout * 16;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int kx = (int)(idx & 3), ky = (int)((idx >> 2) & 3);
        const size_t io = idx >> 4;
        const int o = (int)(io % Cout), i = (int)(io / Cout);
        const int r0 = ky == 0 ? 2 : (ky == 1 ? 1 : 0), r1 = ky == 0 ? 2 : (ky == 1 ? 2 : (ky == 2 ? 1 : 0));
        const int c0 = kx == 0 ? 2 : (kx == 1 ? 1 : 0), c1 = kx == 0 ? 2 : (kx == 1 ? 2 : (kx == 2 ? 1 : 0));
        const float *src = w3 + ((size_t)o * Cin + i) * 9;
        float acc = 0.0f;
        for (int r = r0; r <= r1; ++r)
            for (int c = c0; c <= c1; ++c) acc += src[r * 3 + c];
        wT4[idx] = acc;
    }
}

// ---- MSE loss (mean) forward + gradient -------------------------------------------------------------
__global__ void __launch_bounds__(256) mse_kernel(const float *o, const float *t, float *g, double *block_sums,
                                                  size_t n, float scale) {
    __shared__ double part[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = o[i] - t[i];
        acc += (double)d * (double)d;
        if (g) g[i] = d * scale;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += lane_xor(acc, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// Fixed-order sum of the per-workgroup partial sums (one workgroup; thread i takes partials i, i+256, ...; then a fixed
// tree): the loss value is bit-reproducible run to run, which an atomicAdd of the partials is not.
__global__ void __launch_bounds__(256) loss_finalize_kernel(const double *block_sums, int nblocks, float *loss_sum) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += block_sums[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] = (float)red[0];
}

// SmoothL1Loss (beta = 1, mean): 0.5 d^2 for |d| < 1 else |d| - 0.5;  gradient d or sign(d), times 1/N
__global__ void __launch_bounds__(256) smoothl1_kernel(const float *o, const float *t, float *g, double *block_sums,
                                                       size_t n, float scale) {
    __shared__ double part[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = o[i] - t[i];
        const float ad = fabsf(d);
        acc += ad < 1.0f ? 0.5 * (double)d * (double)d : (double)ad - 0.5;
        if (g) g[i] = (ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f)) * scale;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += lane_xor(acc, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ---- optimizers ------------------------------------------------------------------------------------
// torch.optim.Adam defaults (no weight decay, no amsgrad):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(256) adam_kernel(float *p, const float *g, float *m, float *v, size_t n,
                                                   float step_size, float inv_sqrt_bc2, float b1, float b2, float eps) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);     // torch: lerp(m, g, 1-b1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}
__global__ void __launch_bounds__(256) sgd_kernel(float *p, const float *g, size_t n, float lr) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = p[i] - lr * g[i];
}


// ConvTranspose2d(k4,s2,p1) weight [Cin][Cout][4][4] -> [phase a*2+b][tap ty*2+tx][RowsPad][ColsPad] with
// ky = 3 - 2*ty - a, kx = 3 - 2*tx - b (see dream_conv_transpose4x4s2_nhwc_f32); rows = Cout, cols = Cin.
__global__ void __launch_bounds__(256) pack_wT4_kernel(const float *wT, float *packed, int Cin, int Cout,
                                                       int RowsPad, int ColsPad) {
    const size_t total = (size_t)16 * RowsPad * ColsPad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % ColsPad);
        size_t q = idx / ColsPad;
        const int r = (int)(q % RowsPad);
        const int pt = (int)(q / RowsPad);
        const int ph = pt >> 2, t = pt & 3;
        const int ky = 3 - 2 * (t >> 1) - (ph >> 1), kx = 3 - 2 * (t & 1) - (ph & 1);
        packed[idx] = (r < Cout && c < Cin) ? wT[(((size_t)c * Cout + r) * 4 + ky) * 4 + kx] : 0.0f;
    }
}

// eval-mode BatchNorm (+ optional preceding conv bias) folded to y = conv * scale + shift
__global__ void __launch_bounds__(256) bn_fold_kernel(const float *gamma, const float *beta, const float *mean,
                                                      const float *var, const float *conv_bias, float eps,
                                                      float *scale, float *shift, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        const float s = gamma[c] / sqrtf(var[c] + eps);
        scale[c] = s;
        shift[c] = beta[c] - mean[c] * s + (conv_bias ? conv_bias[c] * s : 0.0f);
    }
}

// im2col for the ResNet stem: NCHW image -> NHWC [B,Ho,Wo,Kpad], k = (c*KH + ky)*KW + kx (the OIHW
// flattening of the weight), zero for out-of-image taps and for k >= C*KH*KW.  One thread per output float;
// writes are fully coalesced, reads hit L1/L2 (the image is 1.9 MB per frame).
__global__ void __launch_bounds__(256) im2col_nchw_kernel(const float *x, float *y, int B, int C, int H, int W,
                                                          int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad) {
    const size_t total = (size_t)B * Ho * Wo * Kpad;
    const int K = C * KH * KW;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % Kpad);
        size_t r = idx / Kpad;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v = 0.0f;
        if (k < K) {
            const int kx = k % KW, ky = (k / KW) % KH, c = k / (KW * KH);
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * C + c) * H + iy) * W + ix];
        }
        y[idx] = v;
    }
}

// The same for the one shape the path uses -- ResNet-101's stem, Conv2d(3, 64, 7, stride 2, pad 3) (dream/models.py:22 via
// torchvision) -- with every divisor a compile-time constant and 32-bit indices (the generic kernel spends ~6 64-bit divisions per
// float: 1.1 TB/s of stores at 16 frames, 2.5 % of a ResNet-101 step and 4.5 % of its evaluation at 128 frames).  Workgroup =
// one output row of one image; a thread produces four consecutive k of one output pixel (one 16-byte store, 4 gathers from the
// L2-resident image).
template <int C, int KH, int KW, int STRIDE, int PAD>
__global__ void __launch_bounds__(256) im2col_nchw_fixed_kernel(const float *x, float *y, int H, int W, int Ho, int Wo, int Kpad) {
    constexpr int K = C * KH * KW;
    const int oy = (int)blockIdx.x, b = (int)blockIdx.y;
    const int K4 = Kpad >> 2;
    const float *img = x + (size_t)b * C * H * W;
    f32x4 *row = (f32x4 *)(y + ((size_t)b * Ho + oy) * (size_t)Wo * Kpad);
    const int n = Wo * K4;
    for (int i = (int)threadIdx.x; i < n; i += 256) {
        const int ox = (int)((unsigned)i / (unsigned)K4);                 // one 32-bit division per four outputs
        const int k0 = (i - ox * K4) * 4;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            if (k < K) {
                const int kx = k % KW, ky = (k / KW) % KH, c = k / (KW * KH);
                const int iy = oy * STRIDE - PAD + ky, ix = ox * STRIDE - PAD + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[e] = img[(c * H + iy) * W + ix];
            }
        }
        row[i] = v;
    }
}

// nn.MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (ResNet stem); padding never wins (-inf)
__global__ void __launch_bounds__(256) maxpool3s2_kernel(const f32x4 *x, f32x4 *y, int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float ninf = -__builtin_huge_valf();
        f32x4 o = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - 1 + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = x[(((size_t)b * H + iy) * W + ix) * C4 + c];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], v[k]);
            }
        }
        y[i] = o;
    }
}


// MaxPool2d(3,2,1) backward.  One thread per input float4: visit the (<= 4) windows that contain this pixel,
// recompute each window's first maximum in scan order (ATen max_pool2d_with_indices semantics: strict > while
// scanning rows then columns, so the first maximal element wins) and take that window's gradient if it is us.
__global__ void __launch_bounds__(256) maxpool3s2_bwd_kernel(const f32x4 *dy, const f32x4 *x, f32x4 *dx, int B, int H, int W,
                                                             int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        // windows oy with 2*oy-1 <= iy <= 2*oy+1
        const int oy_lo = iy / 2, oy_hi = (iy + 1) / 2, ox_lo = ix / 2, ox_hi = (ix + 1) / 2;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= Wo) continue;
                const f32x4 g = dy[(((size_t)b * Ho + oy) * Wo + ox) * C4 + c];
                const float ninf = -__builtin_huge_valf();
                f32x4 best = {ninf, ninf, ninf, ninf};
                int arg[4] = {-1, -1, -1, -1};
                for (int dyy = 0; dyy < 3; ++dyy) {
                    const int yy = 2 * oy - 1 + dyy;
                    if (yy < 0 || yy >= H) continue;
                    for (int dxx = 0; dxx < 3; ++dxx) {
                        const int xx = 2 * ox - 1 + dxx;
                        if (xx < 0 || xx >= W) continue;
                        const f32x4 v = x[(((size_t)b * H + yy) * W + xx) * C4 + c];
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (v[k] > best[k] || arg[k] < 0) { best[k] = v[k]; arg[k] = yy * W + xx; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (arg[k] == iy * W + ix) o[k] += g[k];
            }
        }
        dx[i] = o;
    }
}

// Training form of the stem's pool: the forward pass also stores WHICH element of the 3x3 window won (0..8 = 3 dy + dx, the first
// maximum in scan order: ATen's max_pool2d_with_indices semantics), one byte per output; the backward pass then visits the <= 4
// windows that contain an input pixel and compares ONE byte each, where maxpool3s2_bwd_kernel recomputes every window's arg-max from
// nine loads (20 loads per input float4 on average: 0.46 ms of a ResNet-101 step at 16 frames, 0.8 TB/s).  Workgroup = one row of
// one image (no 64-bit division chains).
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) maxpool3s2_idx_kernel(const f32x4 *x, f32x4 *y, u8x4 *idx, int H, int W, int C4, int Ho, int Wo) {
    const int oy = (int)blockIdx.x, b = (int)blockIdx.y;
    const f32x4 *img = x + (size_t)b * H * W * C4;
    const size_t orow = ((size_t)b * Ho + oy) * (size_t)Wo * C4;
    const int n = Wo * C4;
    for (int i = (int)threadIdx.x; i < n; i += 256) {
        const int ox = (int)((unsigned)i / (unsigned)C4), c = i - ox * C4;
        const float ninf = -__builtin_huge_valf();
        f32x4 best = {ninf, ninf, ninf, ninf};
        int arg[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = img[(iy * W + ix) * C4 + c];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[k] || arg[k] < 0) { best[k] = v[k]; arg[k] = 3 * dy + dx; }
            }
        }
        y[orow + i] = best;
        idx[orow + i] = u8x4{(unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]};
    }
}

__global__ void __launch_bounds__(256) maxpool3s2_idx_bwd_kernel(const f32x4 *dy, const u8x4 *idx, f32x4 *dx, int H, int W, int C4,
                                                                 int Ho, int Wo) {
    const int iy = (int)blockIdx.x, b = (int)blockIdx.y;
    const size_t obase = (size_t)b * Ho * Wo * C4;
    f32x4 *row = dx + ((size_t)b * H + iy) * (size_t)W * C4;
    const int n = W * C4;
    for (int i = (int)threadIdx.x; i < n; i += 256) {
        const int ix = (int)((unsigned)i / (unsigned)C4), c = i - ix * C4;
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        // windows oy with 2 oy - 1 <= iy <= 2 oy + 1, in the order the recomputing kernel adds them (same bits)
        const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1, ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= Wo) continue;
                const size_t w = obase + ((size_t)oy * Wo + ox) * C4 + c;
                const u8x4 a = idx[w];
                const int me = 3 * (iy - (2 * oy - 1)) + (ix - (2 * ox - 1));          // this pixel's position inside that window
                const f32x4 g = dy[w];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((int)a[k] == me) o[k] += g[k];
            }
        }
        row[i] = o;
    }
}

__global__ void __launch_bounds__(256) add_inplace_kernel(float *dst, const float *src, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = ((f32x4 *)dst)[i];
        const f32x4 b = ((const f32x4 *)src)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += b[k];
        ((f32x4 *)dst)[i] = a;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}

// out = a + b (encoder skip tensors joining the decoder, dream/models.py:774-799); optionally publishes max|out|.
__global__ void __launch_bounds__(256) add_kernel(const float *a, const float *b, float *out, size_t n, unsigned *amax) {
    const size_t n4 = n / 4;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 u = ((const f32x4 *)a)[i];
        const f32x4 v = ((const f32x4 *)b)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u[k] += v[k];
            m = fmaxf(m, fabsf(u[k]));
        }
        ((f32x4 *)out)[i] = u;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float u = a[i] + b[i];
        out[i] = u;
        m = fmaxf(m, fabsf(u));
    }
    if (amax) publish_amax(amax, m);
}

// Multi-stage input (dream/models.py:487-493): NHWC [B,H,W,Cpad] <- cat(image NCHW [B,Ci,H,W], maps NCHW [B,K,H/up,W/up]
// nearest-upsampled by `up`), channels >= Ci+K zero.  One thread per (pixel, channel quad).
__global__ void __launch_bounds__(256) stage_input_kernel(const float *img, const float *maps, float *out, int B, int H, int W,
                                                          int Ci, int K, int up, int Cpad, unsigned *amax) {
    const size_t total = (size_t)B * H * W * (Cpad / 4);
    const int Hm = H / up, Wm = W / up;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cq = (int)(i % (Cpad / 4));
        const size_t pix = i / (Cpad / 4);
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cq * 4 + k;
            float t = 0.f;
            if (c < Ci) t = img[(((size_t)b * Ci + c) * H + y) * W + x];
            else if (c < Ci + K) {
                const int ym = y / up, xm = x / up;
                t = maps[(((size_t)b * K + (c - Ci)) * Hm + ym) * Wm + xm];
            }
            v[k] = t;
            m = fmaxf(m, fabsf(t));
        }
        ((f32x4 *)out)[i] = v;
    }
    if (amax) publish_amax(amax, m);
}

// Backward of the map half of stage_input: dmaps[b,k,ym,xm] (+)= sum over the up x up block of g[b,y,x,Ci+k].
__global__ void __launch_bounds__(256) stage_input_bwd_kernel(const float *g, float *dmaps, int B, int H, int W, int Ci, int K,
                                                              int up, int Cpad, int accumulate) {
    const int Hm = H / up, Wm = W / up;
    const size_t total = (size_t)B * K * Hm * Wm;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xm = (int)(i % Wm), ym = (int)((i / Wm) % Hm), k = (int)((i / ((size_t)Wm * Hm)) % K);
        const int b = (int)(i / ((size_t)Wm * Hm * K));
        float s = 0.f;
        for (int y = ym * up; y < (ym + 1) * up; ++y)
            for (int x = xm * up; x < (xm + 1) * up; ++x) s += g[(((size_t)b * H + y) * W + x) * Cpad + Ci + k];
        dmaps[i] = accumulate ? dmaps[i] + s : s;
    }
}

}  // namespace

extern "C" int dream_maxpool2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2: bad arguments (C=%d must be a multiple of 4)", C);
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)x, (f32x4 *)y, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_maxpool2_bwd_nhwc_f32(const float *dy, const float *x, float *dx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(dy && x && dx && B > 0 && H >= 2 && W >= 2 && C % 4 == 0, "maxpool2_bwd: bad arguments");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)dy, (const f32x4 *)x, (f32x4 *)dx, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_maxpool2_relu_bwd_nhwc_f32(const float *dy, const float *x, float *dx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(dy && x && dx && B > 0 && H >= 2 && W >= 2 && C % 4 == 0, "maxpool2_relu_bwd: bad arguments");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)dy, (const f32x4 *)x, (f32x4 *)dx, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_subsample2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "subsample2: bad arguments (C=%d must be a multiple of 4)", C);
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (f32x4 *)y, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_scatter2_nhwc_f32(const float *ys, float *x, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(ys && x && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "scatter2: bad arguments (C=%d must be a multiple of 4)", C);
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(scatter2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)ys, (f32x4 *)x, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_im2col3s2_nhwc_f32(const float *x, float *col, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "im2col3s2: bad arguments (C=%d must be a multiple of 4)", C);
    const size_t total = (size_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * 9 * (C / 4);
    hipLaunchKernelGGL(im2col3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (f32x4 *)col, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_col2im3s2_nhwc_f32(const float *col, float *dx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(col && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "col2im3s2: bad arguments (C=%d must be a multiple of 4)", C);
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(col2im3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)col, (f32x4 *)dx, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_col2im4s2_nhwc_f32(const float *g, const float *scale, const float *shift, float *z, int B, int H, int W, int C, int flags,
                                        void *stream) {
    DREAM_REQUIRE(g && z && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "col2im4s2: bad arguments (C=%d must be a multiple of 4)", C);
    DREAM_REQUIRE((flags & ~DREAM_CONV_RELU) == 0, "col2im4s2: unsupported flags 0x%x", flags);
    const size_t total = (size_t)B * 2 * H * 2 * W * (C / 4);
    hipLaunchKernelGGL(col2im4s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)g, (const f32x4 *)scale,
                       (const f32x4 *)shift, (f32x4 *)z, B, H, W, C / 4, (flags & DREAM_CONV_RELU) ? 1 : 0);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_upsample2_bwd_nhwc_f32(const float *dy, float *dx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(dy && dx && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "upsample2_bwd: bad arguments");
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)dy, (f32x4 *)dx, B, H, W, C / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_relu_bwd_f32(const float *dy, const float *y, float *dx, size_t n, void *stream) {
    DREAM_REQUIRE(dy && y && dx, "relu_bwd: null pointer");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
    const dim3 grid(ceil_div(H * W, 64), ceil_div(C, 64), B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, H * W);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_nchw_to_nhwc_pad_f32(const float *x, float *y, int B, int C, int H, int W, int Cpad, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc_pad: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for((size_t)B * H * W * Cpad)), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, C, H * W, Cpad);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
    const dim3 grid(ceil_div(H * W, 64), ceil_div(C, 64), B);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, H * W);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_pack_conv3x3_weight(const float *w_oihw, float *packed, int Cout, int Cin, int CoutPad,
                                         int CinPad, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "pack_conv3x3_weight: bad arguments");
    // CoutPad / CinPad are the padded ROW / COLUMN counts of the packed tensor (see pack_w_kernel)
    DREAM_REQUIRE(CoutPad >= (mode == 0 ? Cout : Cin) && CinPad >= (mode == 0 ? Cin : Cout), "pack_conv3x3_weight: padding smaller than the tensor");
    hipLaunchKernelGGL(pack_w_kernel, dim3(grid_for((size_t)9 * CoutPad * CinPad)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, packed, Cout, Cin, CoutPad, CinPad, mode, 9);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_pack_conv_weight(const float *w_oihw, float *packed, int Cout, int Cin, int ntaps, int RowsPad,
                                      int ColsPad, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && ntaps >= 1 && (mode == 0 || mode == 1), "pack_conv_weight: bad arguments");
    DREAM_REQUIRE(RowsPad >= (mode == 0 ? Cout : Cin) && ColsPad >= (mode == 0 ? Cin : Cout), "pack_conv_weight: padding smaller than the tensor");
    hipLaunchKernelGGL(pack_w_kernel, dim3(grid_for((size_t)ntaps * RowsPad * ColsPad)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, packed, Cout, Cin, RowsPad, ColsPad, mode, ntaps);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_pack_convT4x4_weight(const float *wT, float *packed, int Cin, int Cout, int RowsPad, int ColsPad,
                                          void *stream) {
    DREAM_REQUIRE(wT && packed && Cin > 0 && Cout > 0 && RowsPad >= Cout && ColsPad >= Cin, "pack_convT4x4_weight: bad arguments");
    hipLaunchKernelGGL(pack_wT4_kernel, dim3(grid_for((size_t)16 * RowsPad * ColsPad)), dim3(256), 0, (hipStream_t)stream,
                       wT, packed, Cin, Cout, RowsPad, ColsPad);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_upsample_conv3x3_weight_as_convT4x4(const float *w_oihw, float *wT4, int Cout, int Cin, void *stream) {
    DREAM_REQUIRE(w_oihw && wT4 && Cout > 0 && Cin > 0, "upsample_conv3x3_weight: bad arguments");
    hipLaunchKernelGGL(upsample_conv_weight_kernel, dim3(grid_for((size_t)16 * Cout * Cin)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, wT4, Cout, Cin);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_bn_fold_f32(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                                 const float *conv_bias, float eps, float *scale, float *shift, int C, void *stream) {
    DREAM_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, conv_bias, eps, scale, shift, C);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_im2col_nchw_f32(const float *x, float *y, int B, int C, int H, int W, int KH, int KW, int stride,
                                     int pad, int Kpad, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && C > 0 && KH > 0 && KW > 0 && stride > 0 && Kpad >= C * KH * KW, "im2col: bad arguments");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    DREAM_REQUIRE(Ho > 0 && Wo > 0, "im2col: empty output");
    if (C == 3 && KH == 7 && KW == 7 && stride == 2 && pad == 3 && Kpad % 4 == 0 && (size_t)Wo * (Kpad / 4) < ((size_t)1 << 30) &&
        (size_t)C * H * W < ((size_t)1 << 30) && B <= 65535) {
        hipLaunchKernelGGL((im2col_nchw_fixed_kernel<3, 7, 7, 2, 3>), dim3((unsigned)Ho, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                           x, y, H, W, Ho, Wo, Kpad);
        DREAM_LAUNCH_OK();
        return 0;
    }
    hipLaunchKernelGGL(im2col_nchw_kernel, dim3(grid_for((size_t)B * Ho * Wo * Kpad)), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, C, H, W, KH, KW, stride, pad, Ho, Wo, Kpad);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_maxpool3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3s2: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)x, (f32x4 *)y, B, H, W, C / 4, Ho, Wo);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_unpack_conv3x3_weight(const float *packed, float *w_oihw, int Cout, int Cin, int CoutPad,
                                           int CinPad, void *stream) {
    DREAM_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && CoutPad >= Cout && CinPad >= Cin, "unpack: bad arguments");
    hipLaunchKernelGGL(unpack_w_kernel, dim3(grid_for((size_t)9 * Cout * Cin)), dim3(256), 0, (hipStream_t)stream,
                       packed, w_oihw, Cout, Cin, CoutPad, CinPad, 9);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_unpack_conv_weight(const float *packed, float *w, int Rows, int Cols, int ntaps, int RowsPad,
                                        int ColsPad, void *stream) {
    DREAM_REQUIRE(w && packed && Rows > 0 && Cols > 0 && ntaps > 0 && RowsPad >= Rows && ColsPad >= Cols, "unpack_conv_weight: bad arguments");
    hipLaunchKernelGGL(unpack_w_kernel, dim3(grid_for((size_t)ntaps * Rows * Cols)), dim3(256), 0, (hipStream_t)stream,
                       packed, w, Rows, Cols, RowsPad, ColsPad, ntaps);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_maxpool3s2_bwd_nhwc_f32(const float *dy, const float *x, float *dx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(dy && x && dx && B > 0 && H > 0 && W > 0 && C % 4 == 0, "maxpool3s2_bwd: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for((size_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)dy, (const f32x4 *)x, (f32x4 *)dx, B, H, W, C / 4, Ho, Wo);
    DREAM_LAUNCH_OK();
    return 0;
}
// Many tensors gathered into one flat buffer by ONE launch (the optimizer's flat gradient buffer: torch hands every parameter its
// own gradient tensor, and 318 hipMemcpyAsync calls cost a ResNet-101 step 1.4 ms at its very end -- 3 % -- for 216 MB).  The
// destination side (where each tensor lives in the flat buffer, cut into chunks of at most 64 K floats) never changes and sits in
// a device-resident table; the source pointers of the step travel as one small array.
namespace {
__global__ void __launch_bounds__(256) multi_copy_kernel(const float *const *srcs, const dream_copy_chunk *chunks) {
    const dream_copy_chunk c = chunks[blockIdx.x];
    const float *src = srcs[c.job] + c.src_off;
    float *dst = c.dst;
    const unsigned n = c.n;
    if ((((size_t)src | (size_t)dst) & 15) == 0) {
        const unsigned n4 = n >> 2;
        for (unsigned i = threadIdx.x; i < n4; i += 256) ((f32x4 *)dst)[i] = ((const f32x4 *)src)[i];
        for (unsigned i = (n4 << 2) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    } else {
        for (unsigned i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}
}  // namespace

extern "C" size_t dream_copy_chunk_bytes(void) { return sizeof(dream_copy_chunk); }

// chunks[c] = {dst, src_off, n, job}: copy n floats from srcs[job] + src_off to dst; srcs: device array of device pointers
extern "C" int dream_multi_copy_f32(const void *srcs, const void *chunks, int nchunks, void *stream) {
    DREAM_REQUIRE(srcs && chunks && nchunks >= 0, "multi_copy: bad arguments");
    if (nchunks == 0) return 0;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const float *const *)srcs,
                       (const dream_copy_chunk *)chunks);
    DREAM_LAUNCH_OK();
    return 0;
}

// MaxPool2d(3,2,1) for training: y and the winner's position inside its window (uint8, 0..8) / the backward pass from those
extern "C" int dream_maxpool3s2_idx_nhwc_f32(const float *x, float *y, unsigned char *idx, int B, int H, int W, int C, void *stream) {
    DREAM_REQUIRE(x && y && idx && B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3s2_idx: bad arguments");
    DREAM_REQUIRE((size_t)H * W * C < ((size_t)1 << 31), "maxpool3s2_idx: image too large for 32-bit offsets");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_idx_kernel, dim3((unsigned)Ho, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x,
                       (f32x4 *)y, (u8x4 *)idx, H, W, C / 4, Ho, Wo);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_maxpool3s2_idx_bwd_nhwc_f32(const float *dy, const unsigned char *idx, float *dx, int B, int H, int W, int C,
                                                 void *stream) {
    DREAM_REQUIRE(dy && idx && dx && B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3s2_idx_bwd: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_idx_bwd_kernel, dim3((unsigned)H, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)dy,
                       (const u8x4 *)idx, (f32x4 *)dx, H, W, C / 4, Ho, Wo);
    DREAM_LAUNCH_OK();
    return 0;
}
// common.h: zero / copy whole words with a kernel (graph-safe replacements of hipMemsetAsync / hipMemcpyAsync)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned *dst, size_t n) {
    const bool wide = (((size_t)dst) & 15) == 0;
    const size_t n4 = wide ? n / 4 : 0;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ((u32x4 *)dst)[i] = z;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 0u;
}
__global__ void __launch_bounds__(256) copy_words_kernel(unsigned *dst, const unsigned *src, size_t n) {
    const bool wide = ((((size_t)dst) | ((size_t)src)) & 15) == 0;
    const size_t n4 = wide ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ((u32x4 *)dst)[i] = ((const u32x4 *)src)[i];
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int dream_zero_words(void *dst, size_t nbytes, hipStream_t stream) {
    DREAM_REQUIRE(dst && nbytes % 4 == 0 && (((size_t)dst) & 3) == 0, "zero_words: whole aligned 32-bit words expected");
    if (nbytes == 0) return 0;
    hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for(nbytes / 16 + 1)), dim3(256), 0, stream, (unsigned *)dst, nbytes / 4);
    DREAM_LAUNCH_OK();
    return 0;
}
int dream_copy_words(void *dst, const void *src, size_t nbytes, hipStream_t stream) {
    DREAM_REQUIRE(dst && src && nbytes % 4 == 0 && ((((size_t)dst) | ((size_t)src)) & 3) == 0, "copy_words: whole aligned 32-bit words expected");
    if (nbytes == 0) return 0;
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid_for(nbytes / 16 + 1)), dim3(256), 0, stream, (unsigned *)dst, (const unsigned *)src, nbytes / 4);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_copy_f32(float *dst, const float *src, size_t n, void *stream) {
    DREAM_REQUIRE(dst && src, "copy: null pointer");
    return dream_copy_words(dst, src, n * sizeof(float), (hipStream_t)stream) ? 2 : 0;
}
extern "C" int dream_add_inplace_f32(float *dst, const float *src, size_t n, void *stream) {
    DREAM_REQUIRE(dst && src, "add_inplace: null pointer");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_add_f32(const float *a, const float *b, float *out, size_t n, unsigned *amax_out, void *stream) {
    DREAM_REQUIRE(a && b && out, "add: null pointer");
    if (amax_out && dream_zero_words(amax_out, sizeof(unsigned), (hipStream_t)stream)) return 2;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, amax_out);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_stage_input_nhwc_f32(const float *img_nchw, const float *maps_nchw, float *out_nhwc, int B, int H, int W,
                                          int Ci, int K, int up, int Cpad, unsigned *amax_out, void *stream) {
    DREAM_REQUIRE(img_nchw && maps_nchw && out_nhwc && B > 0 && H > 0 && W > 0 && Ci > 0 && K > 0, "stage_input: bad arguments");
    DREAM_REQUIRE(up >= 1 && H % up == 0 && W % up == 0 && Cpad % 4 == 0 && Cpad >= Ci + K, "stage_input: bad up / Cpad");
    if (amax_out && dream_zero_words(amax_out, sizeof(unsigned), (hipStream_t)stream)) return 2;
    hipLaunchKernelGGL(stage_input_kernel, dim3(grid_for((size_t)B * H * W * (Cpad / 4))), dim3(256), 0, (hipStream_t)stream,
                       img_nchw, maps_nchw, out_nhwc, B, H, W, Ci, K, up, Cpad, amax_out);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_stage_input_bwd_f32(const float *g_nhwc, float *dmaps_nchw, int B, int H, int W, int Ci, int K, int up,
                                         int Cpad, int accumulate, void *stream) {
    DREAM_REQUIRE(g_nhwc && dmaps_nchw && B > 0 && H > 0 && W > 0 && Ci > 0 && K > 0, "stage_input_bwd: bad arguments");
    DREAM_REQUIRE(up >= 1 && H % up == 0 && W % up == 0 && Cpad >= Ci + K, "stage_input_bwd: bad up / Cpad");
    hipLaunchKernelGGL(stage_input_bwd_kernel, dim3(grid_for((size_t)B * K * (H / up) * (W / up))), dim3(256), 0,
                       (hipStream_t)stream, g_nhwc, dmaps_nchw, B, H, W, Ci, K, up, Cpad, accumulate);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" size_t dream_loss_workspace(size_t n) { return (size_t)grid_for(n) * sizeof(double); }
extern "C" int dream_mse_fwd_bwd_f32(const float *out, const float *target, float *grad, float *loss_sum, void *workspace,
                                     size_t n, double n_total, void *stream) {
    DREAM_REQUIRE(out && target && loss_sum && workspace && n > 0 && n_total > 0, "mse: bad arguments");
    const unsigned nb = grid_for(n);
    hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, out, target, grad, (double *)workspace,
                       n, (float)(2.0 / n_total));
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)nb, loss_sum);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_smoothl1_fwd_bwd_f32(const float *out, const float *target, float *grad, float *loss_sum, void *workspace,
                                          size_t n, double n_total, void *stream) {
    DREAM_REQUIRE(out && target && loss_sum && workspace && n > 0 && n_total > 0, "smoothl1: bad arguments");
    const unsigned nb = grid_for(n);
    hipLaunchKernelGGL(smoothl1_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, out, target, grad, (double *)workspace,
                       n, (float)(1.0 / n_total));
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, (int)nb, loss_sum);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_adam_step_f32(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1,
                                   float beta2, float eps, int step, void *stream) {
    DREAM_REQUIRE(p && g && m && v && step >= 1, "adam: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps);
    DREAM_LAUNCH_OK();
    return 0;
}
extern "C" int dream_sgd_step_f32(float *p, const float *g, size_t n, float lr, void *stream) {
    DREAM_REQUIRE(p && g, "sgd: null pointer");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, n, lr);
    DREAM_LAUNCH_OK();
    return 0;
}

// First encoder convolution: NCHW fp32 image (Cin <= 4) -> NHWC fp32 [B,H,W,Cout], 3x3 s1 p1 + bias
// (+ReLU).  Replaces the fresh Conv2d(3,64,3,1,1) at /root/reference/dream/models.py:592-597.
//
// K = 9*Cin = 27 is far too shallow for the matrix cores and the layer is store-bound (it writes
// 256 B per pixel for 1.7 kFLOP), so it runs on the vector ALU with lane == output channel:
// every lane keeps its 9*Cin weights in VGPRs, the input patch sits in LDS and is read with
// wave-uniform (broadcast) ds_read_b128/b64, and each pixel's 64 channels leave as one fully
// coalesced 256-B store.  Four horizontally adjacent pixels are produced together so each
// broadcast read feeds 12 FMAs.
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {
constexpr int FT = 16;            // 16 x 16 output pixels per workgroup
constexpr int FPW = FT + 4;       // patch row: 1 halo + 16 + 1 halo, padded to 20 floats (16-B rows)
constexpr int FPH = FT + 2;
constexpr int FMAXC = 4;

struct FirstParams {
    const float *x;
    const float *w;
    const float *bias;
    float *y;
    unsigned *amax_out;
    int B, H, W, Cin, Cout;
    int tiles_x, tiles_y;
    int relu;
};

__global__ void __launch_bounds__(256) conv3x3_first_kernel(const FirstParams p) {
    DREAM_DYNAMIC_LDS(float, smem);     // [Cin][FPH][FPW]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    int t = blockIdx.x;
    const int tix = t % p.tiles_x;
    t /= p.tiles_x;
    const int tiy = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int y0 = tiy * FT, x0 = tix * FT;
    const int cout = blockIdx.y * 64 + lane;

    // stage the patch: coalesced along x inside each NCHW plane, zero outside the image
    const int npatch = p.Cin * FPH * FPW;
    for (int idx = tid; idx < npatch; idx += 256) {
        const int c = idx / (FPH * FPW);
        const int rem = idx - c * (FPH * FPW);
        const int py = rem / FPW, px = rem - py * FPW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        float v = 0.0f;
        if (px < FT + 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            v = p.x[(((size_t)b * p.Cin + c) * p.H + gy) * p.W + gx];
        smem[idx] = v;
    }
    // this lane's filter: w[cout][c][ky][kx] (OIHW as stored by torch)
    float wr[FMAXC * 9];
#pragma unroll
    for (int i = 0; i < FMAXC * 9; ++i) wr[i] = (i < p.Cin * 9) ? p.w[(size_t)cout * p.Cin * 9 + i] : 0.0f;
    const float bv = p.bias ? p.bias[cout] : 0.0f;
    __syncthreads();

    // wave w owns rows 4w..4w+3; 4 groups of 4 pixels per row
    float amax = 0.0f;
    for (int g = 0; g < 16; ++g) {
        const int row = wave * 4 + (g >> 2), xg = (g & 3) * 4;
        float acc0 = bv, acc1 = bv, acc2 = bv, acc3 = bv;
#pragma unroll
        for (int c = 0; c < FMAXC; ++c) {
            if (c < p.Cin) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float *src = smem + (c * FPH + row + ky) * FPW + xg;   // wave-uniform address
                    const f32x4 v0 = *(const f32x4 *)src;
                    const float v4 = src[4], v5 = src[5];
                    const float w0 = wr[c * 9 + ky * 3 + 0], w1 = wr[c * 9 + ky * 3 + 1], w2 = wr[c * 9 + ky * 3 + 2];
                    acc0 = fmaf(v0[0], w0, acc0); acc0 = fmaf(v0[1], w1, acc0); acc0 = fmaf(v0[2], w2, acc0);
                    acc1 = fmaf(v0[1], w0, acc1); acc1 = fmaf(v0[2], w1, acc1); acc1 = fmaf(v0[3], w2, acc1);
                    acc2 = fmaf(v0[2], w0, acc2); acc2 = fmaf(v0[3], w1, acc2); acc2 = fmaf(v4, w2, acc2);
                    acc3 = fmaf(v0[3], w0, acc3); acc3 = fmaf(v4, w1, acc3); acc3 = fmaf(v5, w2, acc3);
                }
            }
        }
        if (p.relu) {
            acc0 = fmaxf(acc0, 0.0f); acc1 = fmaxf(acc1, 0.0f); acc2 = fmaxf(acc2, 0.0f); acc3 = fmaxf(acc3, 0.0f);
        }
        const int oy = y0 + row, ox = x0 + xg;
        if (oy < p.H) {
            float *dst = p.y + (((size_t)b * p.H + oy) * p.W + ox) * p.Cout + cout;
            if (ox + 0 < p.W) { dst[0] = acc0; amax = fmaxf(amax, fabsf(acc0)); }
            if (ox + 1 < p.W) { dst[(size_t)p.Cout] = acc1; amax = fmaxf(amax, fabsf(acc1)); }
            if (ox + 2 < p.W) { dst[(size_t)2 * p.Cout] = acc2; amax = fmaxf(amax, fabsf(acc2)); }
            if (ox + 3 < p.W) { dst[(size_t)3 * p.Cout] = acc3; amax = fmaxf(amax, fabsf(acc3)); }
        }
    }
    if (p.amax_out != nullptr) publish_amax(p.amax_out, amax);
}
}  // namespace

static int first_impl(const float *x_nchw, const float *w_oihw, const float *bias, float *y_nhwc, int B, int H, int W,
                      int Cin, int Cout, int relu, void *stream, unsigned *amax_out) {
    DREAM_REQUIRE(x_nchw && w_oihw && y_nhwc, "null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0, "bad shape");
    DREAM_REQUIRE(Cin >= 1 && Cin <= FMAXC, "first conv supports Cin <= %d (got %d)", FMAXC, Cin);
    DREAM_REQUIRE(Cout % 64 == 0, "first conv needs Cout %% 64 == 0 (got %d)", Cout);
    FirstParams p;
    p.x = x_nchw; p.w = w_oihw; p.bias = bias; p.y = y_nhwc; p.amax_out = amax_out;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.tiles_x = ceil_div(W, FT); p.tiles_y = ceil_div(H, FT);
    const size_t lds = (size_t)Cin * FPH * FPW * sizeof(float);
    const dim3 grid((unsigned)((size_t)B * p.tiles_x * p.tiles_y), (unsigned)(Cout / 64));
    hipLaunchKernelGGL(conv3x3_first_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_conv3x3_first_nchw_f32(const float *x_nchw, const float *w_oihw, const float *bias,
                                            float *y_nhwc, int B, int H, int W, int Cin, int Cout,
                                            int relu, void *stream) {
    return first_impl(x_nchw, w_oihw, bias, y_nhwc, B, H, W, Cin, Cout, relu, stream, nullptr);
}
extern "C" int dream_conv3x3_first_nchw_amax_f32(const float *x_nchw, const float *w_oihw, const float *bias,
                                                 float *y_nhwc, unsigned *amax_out, int B, int H, int W, int Cin,
                                                 int Cout, int relu, void *stream) {
    return first_impl(x_nchw, w_oihw, bias, y_nhwc, B, H, W, Cin, Cout, relu, stream, amax_out);
}

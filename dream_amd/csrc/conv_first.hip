// First encoder convolution: NCHW fp32 image (Cin <= 4) -> NHWC fp32 [B,H,W,Cout], 3x3 s1 p1 + bias
// (+ReLU).  Replaces the fresh Conv2d(3,64,3,1,1) at /root/reference/dream/models.py:592-597.
//
// K = 9*Cin = 27 is far too shallow for the matrix cores and the layer is store-bound (it writes
// 256 B per pixel for 1.7 kFLOP), so it runs on the vector ALU with lane == output channel:
// every lane keeps its 9*Cin weights in VGPRs, the input patch sits in LDS and is read with
// wave-uniform (broadcast) ds_read_b128, and each pixel's 64 channels leave as one fully
// coalesced 256-B store.  A lane produces 2 rows x 4 columns of pixels together with PACKED fp32 FMAs
// (v_pk_fma_f32: the two rows in the two halves of a register pair, the weight broadcast to both), so the
// 27 x 8 multiply-adds of a pixel group cost 108 VALU instructions instead of 216 -- the scalar version was
// bound by them (41 TFLOP/s of FMAs at 3.1 TB/s of stores), not by the stores.  The patch is staged twice,
// rows interleaved in pairs (2k, 2k+1) and (2k+1, 2k+2), so that the pair of input rows a filter row needs is
// one aligned register pair per pixel whatever the filter row's parity.
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {
constexpr int FT = 16;            // 16 x 16 output pixels per workgroup
constexpr int FPW = FT + 4;       // patch row: 1 halo + 16 + 1 halo, padded to 20 pixels
constexpr int FPH = FT + 2;
constexpr int FMAXC = 4;
constexpr int FPAIRS = FPH - 1;   // row pairs (r, r + 1), r = 0 .. FPH - 2
constexpr int FPLANE = FPAIRS * FPW * 2;      // floats per channel: [row pair][pixel][2 rows]


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
    DREAM_DYNAMIC_LDS(float, smem);     // [Cin][FPAIRS][FPW][2]: element (c, r, px, h) = patch row r + h, pixel px
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    int t = blockIdx.x;
    const int tix = t % p.tiles_x;
    t /= p.tiles_x;
    const int tiy = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int y0 = tiy * FT, x0 = tix * FT;
    const int cout = blockIdx.y * 64 + lane;

    // stage the patch: coalesced along x inside each NCHW plane, zero outside the image; every patch row but the first and
    // the last is written twice (upper half of pair r - 1, lower half of pair r)
    const int npatch = p.Cin * FPH * FPW;
    for (int idx = tid; idx < npatch; idx += 256) {
        const int c = idx / (FPH * FPW);
        const int rem = idx - c * (FPH * FPW);
        const int py = rem / FPW, px = rem - py * FPW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        float v = 0.0f;
        if (px < FT + 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            v = p.x[(((size_t)b * p.Cin + c) * p.H + gy) * p.W + gx];
        if (py < FPAIRS) smem[c * FPLANE + (py * FPW + px) * 2] = v;
        if (py > 0) smem[c * FPLANE + ((py - 1) * FPW + px) * 2 + 1] = v;
    }
    // this lane's filter: w[cout][c][ky][kx] (OIHW as stored by torch)
    float wr[FMAXC * 9];
#pragma unroll
    for (int i = 0; i < FMAXC * 9; ++i) wr[i] = (i < p.Cin * 9) ? p.w[(size_t)cout * p.Cin * 9 + i] : 0.0f;
    const float bv = p.bias ? p.bias[cout] : 0.0f;
    __syncthreads();

    // wave w owns rows 4w..4w+3 as two row pairs; 4 groups of 4 pixels per row pair
    float amax = 0.0f;
    for (int g = 0; g < 8; ++g) {
        const int row = wave * 4 + 2 * (g >> 2), xg = (g & 3) * 4;           // output rows row, row + 1
        f32x2 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x2{bv, bv};
#pragma unroll
        for (int c = 0; c < FMAXC; ++c) {
            if (c < p.Cin) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // input rows (row + ky, row + 1 + ky) of pixels xg .. xg + 5: six aligned pairs, wave-uniform address
                    const float *src = smem + c * FPLANE + ((row + ky) * FPW + xg) * 2;
                    const f32x4 q0 = *(const f32x4 *)src, q1 = *(const f32x4 *)(src + 4), q2 = *(const f32x4 *)(src + 8);
                    const f32x2 v[6] = {{q0[0], q0[1]}, {q0[2], q0[3]}, {q1[0], q1[1]}, {q1[2], q1[3]}, {q2[0], q2[1]}, {q2[2], q2[3]}};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wv = wr[c * 9 + ky * 3 + kx];
                        const f32x2 w2 = {wv, wv};
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_elementwise_fma(v[i + kx], w2, acc[i]);
                    }
                }
            }
        }
        if (p.relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x2{fmaxf(acc[i][0], 0.0f), fmaxf(acc[i][1], 0.0f)};
        }
        const int ox = x0 + xg;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oy = y0 + row + h;
            if (oy < p.H) {
                float *dst = p.y + (((size_t)b * p.H + oy) * p.W + ox) * p.Cout + cout;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ox + i < p.W) { dst[(size_t)i * p.Cout] = acc[i][h]; amax = fmaxf(amax, fabsf(acc[i][h])); }
            }
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
    const size_t lds = (size_t)Cin * FPLANE * sizeof(float);
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

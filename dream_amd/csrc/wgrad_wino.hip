// Weight gradient of the 3x3 stride-1 pad-1 convolution in the Winograd F(2x2,3x3) domain, on the CDNA4 fp32 matrix cores.
//
// Replaces ATen's conv backward-weight for torch.nn.Conv2d(k=3,s=1,p=1) reached from loss.backward()
// (/root/reference/dream/network.py:335) through the VGG encoder / decoder / head (/root/reference/dream/models.py:598-615,
// 695-710,736-747) and the stride-1 3x3 convs of the ResNet-101 bottlenecks (:22-32).  The direct form (wgrad.hip) spends 36
// multiplications per 2x2 output pixels, input and output channel; here, with  Y = A^T [ sum (G g G^T) .* (B^T d B) ] A :
//
//     dU_p[co][ci] = sum over tiles of  (A dY A^T)_p[tile][co] * (B^T d B)_p[tile][ci]        p = 16 positions
//     dg[co][ci]   = G^T dU G                                                                   (4x4 -> 3x3, once, at the end)
//
// i.e. 16 multiplications, all in fp32 (the transforms only add and subtract; G^T . G has factors 1/2, exact in binary).
//
// GEMM view per position: M = output channels, N = input channels, K = tiles (image, tile row, tile column flattened: up to
// 5.1 M for 128 x 400 x 400), split over tile ranges ("split-K"); partial results go to a workspace and are summed in a FIXED
// order by a second kernel: deterministic, no fp32 atomics.  All global loads go through buffer descriptors with hardware
// bounds masking (zero padding, odd extents, range tails).
//
// Two kernels:
//   * version 2 (wgrad_wino_lds_kernel, below): output channels a multiple of 64 -- every layer of the DREAM networks this
//     path is used for.  The transformed operands are shared by the workgroup through LDS; 0.60-0.69 of the MFMA peak on its
//     own multiplications, 1.8-2.1x the direct kernel (profiles/r02_microbench_wgrad_wino_b128.txt).
//   * version 1 (wgrad_wino_kernel, next): registers only, any multiple of 16 output channels.  v_mfma_f32_16x16x4_f32:
//     lane l supplies A[co = l & 15][k = l >> 4] and B[k = l >> 4][ci = l & 15], so a lane owns ONE tile and ONE channel of each
//     operand per k-step: it loads the tile's 2x2 dy values and its 4x4 input patch (one channel each) as scalars and
//     transforms them in registers into the 16 + 16 operands of the 16 positions' MFMAs.  Wave tile = 32 output x 16 input
//     channels x 16 positions, workgroup = 4 waves; the linear map G^T . G is applied per workgroup (partials hold 3x3 taps).
//     0.50-0.55 of the MFMA peak: its 24 scalar loads per 32 MFMAs are what bounds it (knock-out timing: 0.79-0.93 without
//     the loads, unchanged without the transforms) -- the reason version 2 exists.
#include <type_traits>
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

// Timing diagnostics of the LDS version only (tools/wgw_diag.py builds separate libraries with -DDREAM_WGW_DIAG=k; never the
// product): bit 0 no global loads, bit 1 no transform pieces, bit 2 no per-stage barrier.  Bits 0-2 make the results wrong by construction.
#ifndef DREAM_WGW_DIAG
#define DREAM_WGW_DIAG 0
#endif

namespace {

struct WgWinoParams {
    const float *x;          // [B,H,W,Cin]
    const float *dy;         // [B,H,W,Cdy]  (Cdy >= Cout: gradient tensors may carry padded channels)
    float *partial;          // [nsplit][9][Cout][Cin]
    int B, H, W, Cin, Cout, Cdy;
    int TY, TX, ntiles;
    unsigned long long magic_tpi, magic_tx;
    int tiles_per_split;     // multiple of 4
    int ncog;                // groups of 32 output channels
};

constexpr int NCO = 2;       // 16-channel blocks of output channels per wave
int g_wgw_version = 0;       // 0 = by shape (the LDS version where it applies), 1 = force the register-only version (tests, A/B)

__global__ void __launch_bounds__(256, 2) wgrad_wino_kernel(const WgWinoParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int c16 = lane & 15, kt = lane >> 4;
    const int cog = (int)blockIdx.x % p.ncog, cig = (int)blockIdx.x / p.ncog;
    const int co0 = cog * (16 * NCO), ci0 = cig * 64 + wave * 16;
    const int split = blockIdx.y;
    const int k_begin = split * p.tiles_per_split;
    const int k_end = (k_begin + p.tiles_per_split < p.ntiles) ? k_begin + p.tiles_per_split : p.ntiles;
    const int tiles_per_img = p.TY * p.TX;

    // buffers relative to the first image of this split's tile range (32-bit offsets)
    const int b0 = div_magic40(k_begin, p.magic_tpi);
    const size_t ximg = (size_t)p.H * p.W * p.Cin, yimg = (size_t)p.H * p.W * p.Cdy;
    const BufferRsrc xbuf = make_buffer(p.x + (size_t)b0 * ximg, (size_t)(p.B - b0) * ximg * sizeof(float));
    const BufferRsrc ybuf = make_buffer(p.dy + (size_t)b0 * yimg, (size_t)(p.B - b0) * yimg * sizeof(float));
    const unsigned x_px = (unsigned)(p.Cin * 4), x_row = (unsigned)(p.W * p.Cin * 4);
    const unsigned y_px = (unsigned)(p.Cdy * 4), y_row = (unsigned)(p.W * p.Cdy * 4);

    f32x4 acc[16][NCO];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp)
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb) acc[pp][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    float xr[16], yr[NCO][4];                    // raw patch / raw dy of the k-step in flight

    // issue the loads of the k-step whose first tile is ks (this lane: tile ks + kt)
    auto load_step = [&](int ks) {
        const int tau = ks + kt;
        const bool tv = tau < k_end;
        const int b = div_magic40(tau, p.magic_tpi), rem = tau - b * tiles_per_img;
        const int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        const unsigned xb = (unsigned)(((((b - b0) * p.H + y0) * p.W + x0) * p.Cin + ci0 + c16) * 4);
        const unsigned yb = (unsigned)(((((b - b0) * p.H + 2 * ty) * p.W + 2 * tx) * p.Cdy + co0 + c16) * 4);
        const bool interior = tv && y0 >= 0 && y0 + 3 < p.H && x0 >= 0 && x0 + 3 < p.W;
        if (wave_ballot(interior) == ~0ull) {    // every tile of this k-step is inside its image: no masks
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) xr[4 * r + c] = buffer_load_f32(xbuf, xb + r * x_row + c * x_px, 0);
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) {
                yr[cb][0] = buffer_load_f32(ybuf, yb + cb * 64, 0);
                yr[cb][1] = buffer_load_f32(ybuf, yb + cb * 64 + y_px, 0);
                yr[cb][2] = buffer_load_f32(ybuf, yb + cb * 64 + y_row, 0);
                yr[cb][3] = buffer_load_f32(ybuf, yb + cb * 64 + y_row + y_px, 0);
            }
        } else {
            bool rok[4], cok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rok[r] = tv && (y0 + r) >= 0 && (y0 + r) < p.H;
                cok[r] = (x0 + r) >= 0 && (x0 + r) < p.W;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    xr[4 * r + c] = buffer_load_f32(xbuf, (rok[r] && cok[c]) ? xb + r * x_row + c * x_px : BUFFER_OOB, 0);
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) {     // output pixels (2ty + i, 2tx + j) = patch (1 + i, 1 + j)
                yr[cb][0] = buffer_load_f32(ybuf, (rok[1] && cok[1]) ? yb + cb * 64 : BUFFER_OOB, 0);
                yr[cb][1] = buffer_load_f32(ybuf, (rok[1] && cok[2]) ? yb + cb * 64 + y_px : BUFFER_OOB, 0);
                yr[cb][2] = buffer_load_f32(ybuf, (rok[2] && cok[1]) ? yb + cb * 64 + y_row : BUFFER_OOB, 0);
                yr[cb][3] = buffer_load_f32(ybuf, (rok[2] && cok[2]) ? yb + cb * 64 + y_row + y_px : BUFFER_OOB, 0);
            }
        }
    };

    load_step(k_begin);
    for (int ks = k_begin; ks < k_end; ks += 4) {
        // ---- operands of this k-step from the raw values: V = B^T d B (32 additions), dM' = |A dY A^T| pattern (12 each) ------
        float v[16];
        {
            float w[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {          // along the rows: B^T d
                w[0 * 4 + c] = xr[0 * 4 + c] - xr[2 * 4 + c];
                w[1 * 4 + c] = xr[1 * 4 + c] + xr[2 * 4 + c];
                w[2 * 4 + c] = xr[2 * 4 + c] - xr[1 * 4 + c];
                w[3 * 4 + c] = xr[1 * 4 + c] - xr[3 * 4 + c];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {          // along the columns: (B^T d) B
                v[4 * i + 0] = w[4 * i + 0] - w[4 * i + 2];
                v[4 * i + 1] = w[4 * i + 1] + w[4 * i + 2];
                v[4 * i + 2] = w[4 * i + 2] - w[4 * i + 1];
                v[4 * i + 3] = w[4 * i + 1] - w[4 * i + 3];
            }
        }
        float m[NCO][16];
#pragma unroll
        for (int cb = 0; cb < NCO; ++cb) {
            // A dY A^T with A = [[1,0],[1,1],[1,-1],[0,-1]]; the minus signs of row 3 / column 3 are applied to dU at the end
            const float y00 = yr[cb][0], y01 = yr[cb][1], y10 = yr[cb][2], y11 = yr[cb][3];
            const float s0 = y00 + y10, s1 = y01 + y11, d0 = y00 - y10, d1 = y01 - y11;
            m[cb][0] = y00; m[cb][1] = y00 + y01; m[cb][2] = y00 - y01; m[cb][3] = y01;
            m[cb][4] = s0;  m[cb][5] = s0 + s1;   m[cb][6] = s0 - s1;   m[cb][7] = s1;
            m[cb][8] = d0;  m[cb][9] = d0 + d1;   m[cb][10] = d0 - d1;  m[cb][11] = d1;
            m[cb][12] = y10; m[cb][13] = y10 + y11; m[cb][14] = y10 - y11; m[cb][15] = y11;
        }
        // ---- next k-step's loads fly during the MFMAs ---------------------------------------------------------------------
        if (ks + 4 < k_end) load_step(ks + 4);
#pragma unroll
        for (int pp = 0; pp < 16; ++pp)
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb) acc[pp][cb] = mfma_f32_16x16x4(m[cb][pp], v[pp], acc[pp][cb]);
    }

    // ---- dg = G^T dU G per (co, ci), lane-local (C/D layout: row = 4 (l >> 4) + reg, col = l & 15) ----------------------------
    // signs of the positions whose A dY A^T entry carries a minus: (i, 3) for i < 3 and (3, j) for j < 3
    float *out = p.partial + (size_t)split * 9 * p.Cout * p.Cin;
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u[16];
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                const bool neg = ((pp & 3) == 3) != ((pp >> 2) == 3);
                u[pp] = neg ? -acc[pp][cb][r] : acc[pp][cb][r];
            }
            float t[3][4];                          // G^T dU, G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hs = 0.5f * (u[4 + j] + u[8 + j]), hd = 0.5f * (u[4 + j] - u[8 + j]);
                t[0][j] = u[j] + hs;
                t[1][j] = hd;
                t[2][j] = hs + u[12 + j];
            }
            const int co = co0 + cb * 16 + kt * 4 + r, ci = ci0 + c16;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float hs = 0.5f * (t[a][1] + t[a][2]), hd = 0.5f * (t[a][1] - t[a][2]);
                const float g0 = t[a][0] + hs, g1 = hd, g2 = hs + t[a][3];
                if (co < p.Cout) {
                    out[((size_t)(a * 3 + 0) * p.Cout + co) * p.Cin + ci] = g0;
                    out[((size_t)(a * 3 + 1) * p.Cout + co) * p.Cin + ci] = g1;
                    out[((size_t)(a * 3 + 2) * p.Cout + co) * p.Cin + ci] = g2;
                }
            }
        }
}

// sum over k = 0 .. nsplit - 1 of src[k * stride], added in index order; eight loads in flight (issued one by one, each waiting for
// the previous sum, a thread sees a full memory latency per term: 16 x 16 terms made this kernel 3x the time of the MFMA kernel
// it follows on the 25 x 25 maps of ResNet-101).  Same order of additions, same bits.
DREAM_DEVICE float sum_splits(const float *src, size_t stride, int nsplit) {
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < nsplit; ++k) s += src[(size_t)k * stride];
    return s;
}

// dw_oihw[co][ci][tap] = sum over splits (fixed order) of partial[s][tap][co][ci]
__global__ void __launch_bounds__(256) wgrad_wino_reduce_kernel(const float *partial, float *dw, int nsplit, int Cout, int Cin) {
    const size_t n = (size_t)Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) dw[i * 9 + tap] = sum_splits(partial + (size_t)tap * n + i, (size_t)9 * n, nsplit);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Version 2: the transformed operands are SHARED by the workgroup through LDS.
//
//   workgroup = 8 waves = 64 output x 64 input channels x all 16 positions; wave w owns positions 2w and 2w+1 for the whole
//   64 x 64 block: 2 x 16 accumulators of v_mfma_f32_16x16x4_f32 = 128 registers, two waves per SIMD.
//   K runs over tiles in stages of 8 (tile = wave index for the producers): per stage the 512 threads load the 8 tiles' input
//   patches (64 channels) and dy tiles (64 channels) with float4 loads -- a thread owns (tile, channel quad, patch row): 256
//   contiguous bytes per patch row and wave -- transform them in registers (row transform lane-local, column transform
//   through a DPP exchange inside the quad / pair, as in conv_wino.hip) and store V = B^T d B and dM = A dY A^T to LDS as
//   [position][tile][channel].  The MFMA operands come back as ONE b128 read per operand, position and k-step: lane l reads
//   channels 4 (l & 15) .. +3 of tile (l >> 4) and uses component m for the MFMA whose 16 rows (columns) are the channels
//   4 i + m -- two LDS reads feed sixteen MFMAs.  Global loads per MFMA: 6 float4 per 64 (v1: 24 scalar loads per 32).
//   Software pipeline: stage s multiplies LDS buffer s & 1 while the raw data of stage s+1 (loaded during stage s-1) is
//   transformed into the other buffer and the loads of stage s+2 are issued; every piece of that work is a few instructions
//   placed after one MFMA (pinned with sched_barrier).  (A third register set -- loads two stages ahead -- changed nothing.)
//   Knock-out timing (tools/wgw_diag.py, profiles/r02_wgw_diag.txt, 512 -> 512 @ 50 x 50 x 128): MFMAs + LDS operand reads alone
//   0.96 of the MFMA peak; with the transform pieces 0.85; with the six global loads per wave and stage 0.72 -- the producers'
//   VALU / LDS / memory instructions take issue slots from the MFMAs of both waves of the SIMD (a buffer_load_dwordx4 costs
//   the SIMD 60-180 cycles of issue, MI355X_MICROARCH.md).  Tried without gain: a third register set (loads two stages
//   ahead), the producer work of the two halves of the workgroup at different MFMA numbers (per-slot branches or a
//   duplicated loop: register spills).
//   Signs: the minus signs of A dY A^T and the negated fourth row of V (see conv_wino.hip) cancel except on the positions of
//   the fourth COLUMN; they are applied by the reduction kernel, which also sums the split-K partials in a fixed order and
//   applies dg = G^T dU G.
constexpr int LT = 8;                         // tiles per stage
constexpr int LPS = LT * 64;                  // floats per position plane: [tile][64 channels]
// plane offsets, skewed so that the b128 transform stores of eight consecutive lanes cover the 32 banks ds_write_b128 sees
// exactly once: V -- a quad's four lanes write rows 0..3 (8 floats apart), two quads side by side; dM -- the four lanes
DREAM_DEVICE constexpr int l_plane_v(int p) { return p * LPS + 8 * (p >> 2); }
// (dy row, column half) write rows {0,1} / {3,2} x columns {0,1} / {2,3}: 8 floats per column half, 16 per row half -- plus
// 32 (a full turn of the banks) per row, so that the skews grow with p and no plane runs into the next one
DREAM_DEVICE constexpr int l_plane_y(int p) { return p * LPS + 8 * ((p & 3) >> 1) + 16 * (p >> 3) + 32 * (p >> 2); }
constexpr int LOPS = 16 * LPS + 128;          // floats per operand and buffer
constexpr size_t L_LDS_BYTES = (size_t)4 * LOPS * sizeof(float);     // 2 buffers x (dM, V)

struct WgWinoLdsParams {
    const float *x;          // [B,H,W,Cin]
    const float *dy;         // [B,H,W,Cdy]
    float *partial;          // [nsplit][16][Cout][Cin]
    float *bias_partial;     // BIAS: [nsplit][Cout] column sums of dy over the split's tiles (behind `partial` in the workspace)
    int B, H, W, Cin, Cout, Cdy;
    int TY, TX, ntiles;
    unsigned long long magic_tpi, magic_tx;
    int tiles_per_split;     // multiple of 16 (two stages)
    int ncog;                // groups of 64 output channels
    int nsplit;              // CONVT: splits per phase (partial = [phase][nsplit][9][Cout][Cin], bias_partial = [phase][nsplit][Cout])
};

// UPS: x is stored at half resolution and the conv ran on its nearest x2 upsample (compile-time: the run-time form of the few
// extra selects cost 19 spilled registers)
// BIAS: the bias gradient (column sums of dy) rides in the dy loader.  Every dy element of the split's tiles passes through the
// registers of exactly one (wave, dy row, channel quad) of each workgroup, so the workgroups of input-channel block 0 add it up
// on the way (two packed additions per stage and lane next to 64 MFMAs) -- the stand-alone pass over every gradient tensor
// (1.8 % of a vgg_q training step, at the HBM roofline) is not launched.  Fixed order: lane, then (wave, row) through LDS, then the
// splits in the reduction kernel.
// CONVT (round 6): the weight gradient of nn.ConvTranspose2d(k4, s2, p1) (the ResNet decoder, dream/models.py:37-136), one output
// phase (a, b) = blockIdx.z per workgroup.  Phase (a, b) of the transposed conv is a 2 x 2-tap conv of x, i.e. a pad-1 3x3 conv whose
// kernel g3 is zero outside rows {a, a + 1} and columns {b, b + 1} (conv_wino.hip, convT4x4_phase_kernels); its output is the phase
// view Y_ab(i, j) = dY(2 i + a, 2 j + b).  dg3 = G^T dU G is needed on those 2 x 2 taps only, and rows {a, a + 1} of G^T reach positions
// {a, a + 1, a + 2} of the 4 x 4 domain only (G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]): NINE of the sixteen positions -- the
// F(2x2,2x2) minimal-filtering count, 9 multiplications per 2 x 2 outputs of a phase where the direct form (wgrad.hip) takes 16.
// Wave w owns active position q = w (row a + q / 3, column b + q % 3) for the whole 64 x 64 block; the ninth position is shared:
// wave w takes its blocks (m = w >> 1, n = 2 (w & 1) .. + 1).  36 MFMAs per wave and stage instead of 64; the producers load the same
// 6 float4 per stage and transform three of V's four columns.
// PHB (CONVT): the phase's column b is a compile-time constant of the body -- the kernel branches once on blockIdx.z & 1 --, so that the
// patch column the phase never reads (3 for b = 0, 0 for b = 1: V's live columns b .. b + 2 are combinations of patch columns b .. b + 2
// only) is not loaded: FIVE float4 loads per stage instead of six next to 36 MFMAs, and the eight wave-uniform selects per stage that
// picked the operands of the outer live column are gone.  Likewise the patch ROW the phase never reads (3 for a = 0, 0 for a = 1; a is
// wave-uniform at run time): its lanes' loads are masked out of range (no memory traffic; their V row is never read).  Same values,
// same order of additions as the six-load form: bit-identical (profiles/r06_ab_convT_wgrad_five_loads.txt).
template <bool UPS, bool BIAS, bool CONVT, int PHB>
DREAM_DEVICE void wgrad_wino_lds_body(const WgWinoLdsParams &p, float *smem) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int ph_a = CONVT ? (int)(blockIdx.z >> 1) : 0;                                                   // output phase
    constexpr int ph_b = CONVT ? PHB : 0;
    const int cog = (int)blockIdx.x % p.ncog, cig = (int)blockIdx.x / p.ncog;
    const int co0 = cog * 64, ci0 = cig * 64;
    const int split = blockIdx.y;
    const int k_begin = split * p.tiles_per_split;
    const int k_end = (k_begin + p.tiles_per_split < p.ntiles) ? k_begin + p.tiles_per_split : p.ntiles;
    const int nstages = ((k_end - k_begin + 2 * LT - 1) / (2 * LT)) * 2;      // even: the stage loop is unrolled by two
    const int tiles_per_img = p.TY * p.TX;

    // buffers relative to the first image of this split's tile range (32-bit offsets)
    const int b0 = div_magic40(k_begin, p.magic_tpi);
    const int Hs = UPS ? p.H / 2 : p.H, Ws = UPS ? p.W / 2 : p.W;               // stored extent of x
    const int Hd = CONVT ? 2 * p.H : p.H, Wd = CONVT ? 2 * p.W : p.W;         // stored extent of dy (CONVT: the full-resolution gradient)
    const size_t ximg = (size_t)Hs * Ws * p.Cin, yimg = (size_t)Hd * Wd * p.Cdy;
    const BufferRsrc xbuf = make_buffer(p.x + (size_t)b0 * ximg, (size_t)(p.B - b0) * ximg * sizeof(float));
    const BufferRsrc ybuf = make_buffer(p.dy + (size_t)b0 * yimg, (size_t)(p.B - b0) * yimg * sizeof(float));

    // ---- producer roles: the wave's tile of a stage is (stage tile + wave); its decomposition is scalar work ----------------
    // V: patch row vr = lane & 3, channel quad vq = lane >> 2.  dM: the lane owns ONE pixel of the tile's 2 x 2 dy pixels -- row yr2 = lane & 1,
    // column yh = (lane >> 1) & 1 -- for channel quad yq = lane >> 2: one float4 load per stage (was: both pixels of the row in each of the
    // two lanes of a column pair, i.e. every dy element loaded twice); the other column / row come from the quad neighbours by DPP.  The
    // lane stores transformed columns {0, 1} (yh = 0) or {3, 2} (yh = 1) of rows {0, 1} (yr2 = 0) or {3, 2} (yr2 = 1).
    const int vr = lane & 3, vq = lane >> 2;
    const int yr2 = lane & 1, yh = (lane >> 1) & 1, yq = lane >> 2;
    const float vsb = (vr == 1) ? 1.0f : -1.0f;             // column transform of V: u_r + vsb * u_partner (conv_wino.hip)
    const float ysg = (yr2 == 0) ? 1.0f : -1.0f;            // mixed dM row: partner + ysg * own
    const float csg = yh ? -1.0f : 1.0f;                    // mixed dM column: partner + csg * own  (y0 + y1 / y0 - y1)
    const int v_store = l_plane_v(4 * vr) + wave * 64 + 4 * vq;              // V[4 vr + j]: + j * LPS
    // dM rows of this lane: row A = its own values (row 0 for the upper dy row, row 3 for the lower one), row B = the mixed one (row 1 =
    // upper + lower; row 2 is stored NEGATED, lower - upper: own + ysg * partner is one v_fmac_f32 with a DPP source, in place -- the
    // reduction kernels flip the sign of the row-2 positions, exactly).  Columns: A = the pixel's own (0 / 3), B = the mixed one (1 / 2).
    const int y_store_aa = l_plane_y((yr2 == 0 ? 0 : 12) + (yh ? 3 : 0)) + wave * 64 + 4 * yq;
    const int y_store_ab = l_plane_y((yr2 == 0 ? 0 : 12) + (yh ? 2 : 1)) + wave * 64 + 4 * yq;
    const int y_store_ba = l_plane_y((yr2 == 0 ? 4 : 8) + (yh ? 3 : 0)) + wave * 64 + 4 * yq;
    const int y_store_bb = l_plane_y((yr2 == 0 ? 4 : 8) + (yh ? 2 : 1)) + wave * 64 + 4 * yq;

    f32x4 xr[2][4];                                          // [register set][column]
    f32x4 yr[2];                                             // [register set]: the lane's dy pixel
    f32x4 ycb[2];                                            // the mixed column of the stage being transformed (piece 4 -> piece 5)
    const bool bias_wg = BIAS && cig == 0;                   // workgroup-uniform
    f32x4 bsum = {0.0f, 0.0f, 0.0f, 0.0f};                   // dy of (wave's tiles, pixel (yr2, yh), quad yq)

    // one load of stage st into register set `set`: n = 0..3 the patch row's columns, 4..5 the dy row's columns.
    // VALU instructions share the SIMD's issue slots with the MFMAs (measured: every VALU instruction in this loop shows up
    // in the run time), so the address is split: everything that depends on the tile is wave-uniform and computed on the
    // scalar unit (offset of the tile's first output pixel + column), the lane adds its constant (patch row, channel quad)
    // with ONE v_add.  Pixels outside the image are loaded as zeros by the hardware's range check (offsets with bit 31 set are
    // beyond any buffer), and the two reasons for it keep their own out-of-range words so that the add is all a load costs: a dead
    // COLUMN (or a tile past the split's end) is wave-uniform -- the scalar part becomes OOB_WAVE by a scalar select --, a dead ROW
    // is per lane and the same for all loads of the stage -- the lane part becomes OOB_LANE by one select per operand and stage
    // (was: add + select + or per load).  With valid scalar parts in [0, 2^30) (the plans keep a split's span below that) and lane
    // constants below 2^24 in magnitude, every sum that involves an out-of-range word lands in [2^31, 2^32):
    //   valid + OOB_LANE in [0xC0000000, 2^32),  OOB_WAVE + lane constant = 0xE0000000 +- 2^24,  OOB_WAVE + OOB_LANE = 0xA0000000 (mod 2^32).
    constexpr unsigned OOB_LANE = 0xC0000000u, OOB_WAVE = 0xE0000000u;
    // patch row vr - 1 relative to the tile's first output row; with the fused upsample the tile's 4 x 4 patch of the upsampled
    // image is rows / columns {-1, 0, 0, +1} of the stored one around the tile's source pixel ((2 t - 1 + r) >> 1 = t + ((r - 1) >> 1))
    const int lane_dx = UPS ? (int)(((((vr - 1) >> 1) * Ws) * p.Cin + 4 * vq) * 4) : (int)((((vr - 1) * p.W) * p.Cin + 4 * vq) * 4);
    const int lane_dy = CONVT ? (int)(((yr2 * 2 * Wd + 2 * yh) * p.Cdy + 4 * yq) * 4) : (int)(((yr2 * p.W + yh) * p.Cdy + 4 * yq) * 4);
    // the patch row in the row test: the lanes of the phase's dead patch row (CONVT) fail it for every tile
    const int vr_chk = (CONVT && vr == (ph_a ? 0 : 3)) ? (1 << 29) : vr;
    const int x_px = p.Cin * 4;
    auto issue_load = [&](int set, int st, int n) {
        const int tau = k_begin + st * LT + wave;                               // wave-uniform from here ...
        const bool tv = tau < k_end;
        const int b = div_magic40(tau, p.magic_tpi), rem = tau - b * tiles_per_img;
        const int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
        const int pix = ((b - b0) * p.H + 2 * ty) * p.W + 2 * tx;               // the tile's first output pixel
        if (n < 4) {
            if (CONVT && n == 3) return;                                         // three live patch columns: ph_b + n
            const int c = n + ph_b;
            const int s_off = UPS ? ((((b - b0) * Hs + ty) * Ws + tx) * p.Cin + ci0) * 4 + ((c - 1) >> 1) * x_px
                                    : (pix * p.Cin + ci0) * 4 + (c - 1) * x_px;  // ... to here
            const bool col_ok = tv & ((unsigned)(2 * tx - 1 + c) < (unsigned)p.W);                  // wave-uniform
            const bool row_ok = (unsigned)(2 * ty - 1 + vr_chk) < (unsigned)p.H;                    // per lane, the same for the stage's loads
            xr[set][n] = buffer_load_x4(xbuf, (col_ok ? (unsigned)s_off : OOB_WAVE) + (row_ok ? (unsigned)lane_dx : OOB_LANE), 0);
        } else {
            if (n == 5) return;                                                  // one dy load per stage
            const int s_off = CONVT ? ((((b - b0) * Hd + 4 * ty + ph_a) * Wd + 4 * tx + ph_b) * p.Cdy + co0) * 4
                                    : (pix * p.Cdy + co0) * 4;
            const bool pix_ok = ((2 * ty + yr2) < p.H) & ((2 * tx + yh) < p.W);                      // per lane
            yr[set] = buffer_load_x4(ybuf, (tv ? (unsigned)s_off : OOB_WAVE) + (pix_ok ? (unsigned)lane_dy : OOB_LANE), 0);
        }
    };
    // piece k = 0..3 of the V transform (transformed column j = k), 4..5 of the dM transform (column 2 yh + k - 4)
    // (CONVT: k = 0 stands for the phase's live outer column -- 0 for b = 0, 3 for b = 1)
    auto transform_piece = [&](int set, float *buf, int k) {
        if (k < 4) {
            const f32x4 *d = xr[set];
            f32x4 u;
            int kcol = k;
            if (CONVT) {                           // d[0..2] = patch columns ph_b .. ph_b + 2
                if (ph_b == 0) u = k == 0 ? d[0] - d[2] : k == 1 ? d[1] + d[2] : d[2] - d[1];
                else u = k == 0 ? d[0] - d[2] : k == 1 ? d[0] + d[1] : d[1] - d[0];
                kcol = k == 0 ? (ph_b ? 3 : 0) : k;
            } else {
                u = k == 0 ? d[0] - d[2] : k == 1 ? d[1] + d[2] : k == 2 ? d[2] - d[1] : d[1] - d[3];
            }
            // u + vsb * (u of the partner row), fused: four v_fmac_f32 with a DPP source (was: four v_mov_b32_dpp + two v_pk_fma_f32)
            const f32x4 v = fma_quad_perm_2211(u, vsb);
            *(f32x4 *)(buf + LOPS + v_store + kcol * LPS) = v;
        } else {
            // A dY A^T of the tile's 2 x 2 pixels, a lane = one pixel.  Piece 4: along the row -- (y0, y1) -> y0, y0 + y1, y0 - y1, y1 (the
            // minus sign of the fourth column is applied at the end): the lane's own value IS column 0 / 3, the mixed column is
            // partner + csg * own (y1 + y0 / y0 - y1); both go to row A.  Piece 5: along the column, in place: own + ysg * partner for the two
            // columns = row 1 (upper lanes) / minus row 2 (lower lanes).
            if (k == 4) {
                const f32x4 own = yr[set];
                if (BIAS && bias_wg) bsum = bsum + own;                           // out-of-range pixels and tiles were loaded as zeros
                f32x4 cb;
#pragma unroll
                for (int e = 0; e < 4; ++e) cb[e] = __builtin_fmaf(csg, own[e], quad_perm_2301(own[e]));
                ycb[set] = cb;
                *(f32x4 *)(buf + y_store_aa) = own;
                *(f32x4 *)(buf + y_store_ab) = cb;
            } else {
                const f32x4 ma = fma_quad_perm_1032(yr[set], ysg), mb = fma_quad_perm_1032(ycb[set], ysg);
                *(f32x4 *)(buf + y_store_ba) = ma;
                *(f32x4 *)(buf + y_store_bb) = mb;
            }
        }
    };

    // ---- consumer: MFMA operands ---------------------------------------------------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
  if constexpr (!CONVT) {
    const int a_lane = lg * 64 + 4 * li;                     // + plane + k-step * 256
    f32x4 acc[2][4][4];                                      // [position of this wave][m: co = 4 i + m][n: ci = 4 j + n]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[pl][m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // planes of positions 2 wave + pl (pl = 0, 1): same row (wave >> 1), columns 2 (wave & 1) + pl
    const int py_base = (2 * wave) * LPS + 8 * (wave & 1) + 16 * (wave >> 2) + 32 * (wave >> 1);
    const int pv_base = (2 * wave) * LPS + 8 * (wave >> 1);
    f32x4 oy[2], ov[2];
    auto read_ops = [&](int set, const float *buf, int grp) {   // grp = k-step * 2 + position
        const int ks = grp >> 1, pl = grp & 1;
        oy[set] = *(const f32x4 *)(buf + py_base + pl * LPS + ks * 256 + a_lane);
        ov[set] = *(const f32x4 *)(buf + LOPS + pv_base + pl * LPS + ks * 256 + a_lane);
    };

    // one stage: 4 groups (2 k-steps x 2 positions) of 16 MFMAs on LDS buffer P; after MFMA number n of the stage one small
    // piece of the producer work: transform pieces of stage st + 1 (register set 1 - P -> LDS buffer 1 - P), then the loads
    // of stage st + 2 (-> register set P, free since the previous stage), operand reads of the next group.
    auto stage = [&](auto parity, int st) {
        constexpr int P = decltype(parity)::value;
        const float *cur = smem + P * 2 * LOPS;
        float *nxt = smem + (1 - P) * 2 * LOPS;
        read_ops(0, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int pl = grp & 1;
#pragma unroll
            for (int mn = 0; mn < 16; ++mn) {
                const int m = mn >> 2, n4 = mn & 3, n = grp * 16 + mn;
                acc[pl][m][n4] = mfma_f32_16x16x4(oy[grp & 1][m], ov[grp & 1][n4], acc[pl][m][n4]);
                if (mn == 4 && grp < 3) read_ops((grp + 1) & 1, cur, grp + 1);
                if (!(DREAM_WGW_DIAG & 1) && n >= 1 && n < 13 && (n & 1) == 1) issue_load(P, st + 2, (n - 1) >> 1);        // 6 loads, every other MFMA
                if (!(DREAM_WGW_DIAG & 2) && n >= 16 && n < 64 && (n & 7) == 0) transform_piece(1 - P, nxt, (n - 16) >> 3);  // 6 pieces, every 8th MFMA
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(DREAM_WGW_DIAG & 4)) __syncthreads();
    };

    // ---- prologue: stage 0 into buffer 0, loads of stage 1 in flight ----------------------------------------------------------
#pragma unroll
    for (int n = 0; n < 6; ++n) issue_load(0, 0, n);
#pragma unroll
    for (int k = 0; k < 6; ++k) transform_piece(0, smem, k);
#pragma unroll
    for (int n = 0; n < 6; ++n) issue_load(1, 1, n);
    __syncthreads();
    for (int st = 0; st < nstages; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        stage(std::integral_constant<int, 1>{}, st + 1);
    }

    // ---- partial dU: lane holds rows i = 4 (l >> 4) + r, column j = l & 15 of every 16 x 16 block (m, n) -----------------------
    float *out = p.partial + (size_t)split * 16 * p.Cout * p.Cin;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + 4 * (4 * lg + r) + m;
                const f32x4 v = {acc[pl][m][0][r], acc[pl][m][1][r], acc[pl][m][2][r], acc[pl][m][3][r]};
                *(f32x4 *)(out + ((size_t)(2 * wave + pl) * p.Cout + co) * p.Cin + ci0 + 4 * li) = v;
            }
  } else {
    // ---- CONVT: nine active positions -------------------------------------------------------------------------------------------
    const int a_lane = lg * 64 + 4 * li;
    const int q_own = wave;                                   // active position index 0..7: row ph_a + q / 3, column ph_b + q % 3
    const int p_own = 4 * (ph_a + q_own / 3) + ph_b + q_own % 3, p_sh = 4 * (ph_a + 2) + ph_b + 2;      // the shared one: q = 8
    auto plane_y = [](int pp) { return pp * LPS + 8 * ((pp & 3) >> 1) + 16 * (pp >> 3) + 32 * (pp >> 2); };      // = l_plane_y
    auto plane_v = [](int pp) { return pp * LPS + 8 * (pp >> 2); };                                               // = l_plane_v
    const int py_own = plane_y(p_own), pv_own = plane_v(p_own), py_sh = plane_y(p_sh), pv_sh = plane_v(p_sh);
    const int sh_m = wave >> 1, sh_n = wave & 1;              // shared position: blocks (m = sh_m, n = 2 sh_n, 2 sh_n + 1)
    f32x4 acc[4][4], acc_sh[2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    acc_sh[0] = acc_sh[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // the shared position's operands: the wave needs component sh_m of dM's quad and components 2 sh_n, 2 sh_n + 1 of V's -- read exactly
    // those (a b32 and a b64 read at wave-constant offsets) instead of two b128 reads + ten wave-uniform selects per stage
    f32x4 oy[2], ov[2];
    float sy[2];
    f32x2 sv[2];
    auto read_ops = [&](int set, const float *buf, int ks) {
        oy[set] = *(const f32x4 *)(buf + py_own + ks * 256 + a_lane);
        ov[set] = *(const f32x4 *)(buf + LOPS + pv_own + ks * 256 + a_lane);
        sy[set] = buf[py_sh + ks * 256 + a_lane + sh_m];
        sv[set] = *(const f32x2 *)(buf + LOPS + pv_sh + ks * 256 + a_lane + 2 * sh_n);
    };
    // one stage: 2 k-steps x (16 MFMAs of the wave's own position + 2 of the shared one); producer pieces behind single MFMAs as in
    // the 3x3 form: the six loads of stage st + 2 behind MFMAs 1, 3, .. 11, the five transform pieces of stage st + 1 behind MFMAs 14, 18,
    // .. 30, the operand reads of the second k-step behind MFMA 4
    auto stage = [&](auto parity, int st) {
        constexpr int P = decltype(parity)::value;
        const float *cur = smem + P * 2 * LOPS;
        float *nxt = smem + (1 - P) * 2 * LOPS;
        read_ops(0, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mn = 0; mn < 18; ++mn) {
                const int n = ks * 18 + mn;
                if (mn < 16) {
                    const int m = mn >> 2, n4 = mn & 3;
                    acc[m][n4] = mfma_f32_16x16x4(oy[ks][m], ov[ks][n4], acc[m][n4]);
                } else {
                    acc_sh[mn - 16] = mfma_f32_16x16x4(sy[ks], sv[ks][mn - 16], acc_sh[mn - 16]);
                }
                if (mn == 4 && ks == 0) read_ops(1, cur, 1);
                if (!(DREAM_WGW_DIAG & 1) && n >= 1 && n < 13 && (n & 1) == 1) issue_load(P, st + 2, (n - 1) >> 1);
                if (!(DREAM_WGW_DIAG & 2) && n >= 14 && n < 34 && ((n - 14) & 3) == 0) {
                    const int piece = (n - 14) >> 2;              // 0: V's outer live column, 1, 2: V columns 1, 2; 3, 4: dM
                    transform_piece(1 - P, nxt, piece < 3 ? piece : piece + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(DREAM_WGW_DIAG & 4)) __syncthreads();
    };
#pragma unroll
    for (int n = 0; n < 6; ++n) issue_load(0, 0, n);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (k != 3) transform_piece(0, smem, k);
#pragma unroll
    for (int n = 0; n < 6; ++n) issue_load(1, 1, n);
    __syncthreads();
    for (int st = 0; st < nstages; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        stage(std::integral_constant<int, 1>{}, st + 1);
    }
    // ---- partial dU of this phase and split: [9][Cout][Cin] ---------------------------------------------------------------------
    float *out = p.partial + ((size_t)blockIdx.z * p.nsplit + split) * 9 * p.Cout * p.Cin;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 4 * (4 * lg + r) + m;
            const f32x4 v = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
            *(f32x4 *)(out + ((size_t)q_own * p.Cout + co) * p.Cin + ci0 + 4 * li) = v;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + 4 * (4 * lg + r) + sh_m;
        float *dst = out + ((size_t)8 * p.Cout + co) * p.Cin + ci0 + 4 * li + 2 * sh_n;
        dst[0] = acc_sh[0][r];
        dst[1] = acc_sh[1][r];
    }
  }
    if (BIAS && bias_wg) {                                   // the stage loop ended with a barrier: LDS is free
        *(f32x4 *)(smem + (4 * wave + 2 * yr2 + yh) * 64 + 4 * yq) = bsum;
        __syncthreads();
        if (threadIdx.x < 64) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < 32; ++r) s += smem[r * 64 + threadIdx.x];
            p.bias_partial[((size_t)(CONVT ? blockIdx.z * p.nsplit : 0) + split) * p.Cout + co0 + threadIdx.x] = s;
        }
    }
}

template <bool UPS, bool BIAS, bool CONVT = false>
__global__ void __launch_bounds__(512, 1) wgrad_wino_lds_kernel(const WgWinoLdsParams p) {
    DREAM_DYNAMIC_LDS(float, smem);
    if (CONVT && (blockIdx.z & 1)) wgrad_wino_lds_body<UPS, BIAS, CONVT, 1>(p, smem);      // workgroup-uniform
    else wgrad_wino_lds_body<UPS, BIAS, CONVT, 0>(p, smem);
}

// dw_oihw[co][ci][3][3] = G^T (sum over splits, fixed order, of dU) G with the deferred signs (fourth column of the positions).
// One thread per weight summed 16 positions x nsplit partials, one workgroup per CU: 79 us for the 67 MB of a 256 x 256-channel
// layer of ResNet-101 at 16 frames (0.85 TB/s, 3 % of that training step).  Now a weight's four COLUMNS of positions go to the four
// waves of a workgroup (lane = weight: 256 contiguous bytes per load), sixteen loads in flight per thread, G^T applied per column,
// the columns exchanged through LDS, G applied by the first three waves (row a each).  Same additions in the same order: same bits.
// db (optional): the bias gradient = sum over splits of the main kernel's column sums.
__global__ void __launch_bounds__(256) wgrad_wino_lds_reduce_kernel(const float *partial, float *dw, const float *bias_partial, float *db,
                                                                    int nsplit, int Cout, int Cin) {
    __shared__ float t_s[4][3][64];
    const size_t n = (size_t)Cout * Cin;                     // a multiple of 64 * 64
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    if (db != nullptr) {
        const int co = (int)blockIdx.x * 256 + (int)threadIdx.x;
        if (co < Cout) db[co] = sum_splits(bias_partial + co, (size_t)Cout, nsplit);
    }
    // column j of dU: positions j, 4 + j, 8 + j, 12 + j
    const float *src = partial + (size_t)j * n + i;
    const size_t rs = (size_t)4 * n, ss = (size_t)16 * n;    // row and split strides
    float u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        float v[4][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[kk][r] = src[(size_t)(k + kk) * ss + (size_t)r * rs];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] += v[kk][r];
    }
    for (; k < nsplit; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] += src[(size_t)k * ss + (size_t)r * rs];
    if (j == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = -u[r];
    }
    u[2] = -u[2];                                   // the main kernel stores row 2 of A dY A^T negated (its dM transform, piece 5)
    {                                               // G^T dU, G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
        const float hs = 0.5f * (u[1] + u[2]), hd = 0.5f * (u[1] - u[2]);
        t_s[j][0][lane] = u[0] + hs;
        t_s[j][1][lane] = hd;
        t_s[j][2][lane] = hs + u[3];
    }
    __syncthreads();
    if (j < 3) {                                    // row a = j of (G^T dU) G
        const float t0 = t_s[0][j][lane], t1 = t_s[1][j][lane], t2 = t_s[2][j][lane], t3 = t_s[3][j][lane];
        const float hs = 0.5f * (t1 + t2), hd = 0.5f * (t1 - t2);
        dw[i * 9 + j * 3 + 0] = t0 + hs;
        dw[i * 9 + j * 3 + 1] = hd;
        dw[i * 9 + j * 3 + 2] = hs + t3;
    }
}

// ---- CONVT: dwT[ci][co][ky][kx] of nn.ConvTranspose2d(k4, s2, p1) from the four phases' nine-position partials -------------------------
// partial [phase][nsplit][9][Cout][Cin] (position index q = 3 * row + column over rows a .. a + 2, columns b .. b + 2 of the 4 x 4
// domain).  Per weight: sum the splits in index order, negate the positions of the fourth column (the deferred signs, as in the 3x3
// form), apply the two live rows of G^T (G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]: rows {0, 1} over positions {0, 1, 2} for a = 0,
// rows {1, 2} over {1, 2, 3} for a = 1) and the two live columns of G, and store tap (r, c) of the phase's zero-padded 3x3 kernel where
// it came from: wT[ci][co][a + 3 - 2 r][b + 3 - 2 c] (conv_wino.hip, convT4x4_phase_kernels).  Waves 0..2 of a workgroup take one
// column of positions each (lane = weight: 256 contiguous bytes per load), waves 0..1 then one tap row each.
// db (optional): the bias gradient = sum over phases and splits (fixed order) of the main kernel's column sums of dy.
__global__ void __launch_bounds__(256) wgrad_wino_convT_reduce_kernel(const float *partial, float *dwT, const float *bias_partial, float *db,
                                                                      int nsplit, int Cout, int Cin) {
    __shared__ float t_s[3][2][64];
    const size_t n = (size_t)Cout * Cin;                     // a multiple of 64 * 64
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int phase = blockIdx.y, a = phase >> 1, b = phase & 1;
    const size_t i = (size_t)blockIdx.x * 64 + lane;        // = co * Cin + ci
    if (db != nullptr && phase == 0) {
        const int co = (int)blockIdx.x * 256 + (int)threadIdx.x;
        if (co < Cout) db[co] = sum_splits(bias_partial + co, (size_t)Cout, 4 * nsplit);
    }
    if (j < 3) {                                             // column j of the active positions: q = j, 3 + j, 6 + j
        const float *src = partial + ((size_t)phase * nsplit * 9 + j) * n + i;
        float u[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < nsplit; ++k) {
            float v[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) v[r] = src[((size_t)k * 9 + 3 * r) * n];
#pragma unroll
            for (int r = 0; r < 3; ++r) u[r] += v[r];
        }
        if (b + j == 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r) u[r] = -u[r];
        }
        u[2 - a] = -u[2 - a];                               // domain row 2 (active row 2 - a): stored negated by the main kernel
        // the two live rows of G^T over the three active row positions
        float t0, t1;
        if (a == 0) { t0 = u[0] + 0.5f * (u[1] + u[2]); t1 = 0.5f * (u[1] - u[2]); }       // taps r = 0, 1 from positions 0, 1, 2
        else        { t0 = 0.5f * (u[0] - u[1]); t1 = 0.5f * (u[0] + u[1]) + u[2]; }       // taps r = 1, 2 from positions 1, 2, 3
        t_s[j][0][lane] = t0;
        t_s[j][1][lane] = t1;
    }
    __syncthreads();
    if (j < 2) {                                             // tap row r = a + j; its two live columns
        const float c0 = t_s[0][j][lane], c1 = t_s[1][j][lane], c2 = t_s[2][j][lane];
        float g0, g1;
        if (b == 0) { g0 = c0 + 0.5f * (c1 + c2); g1 = 0.5f * (c1 - c2); }
        else        { g0 = 0.5f * (c0 - c1); g1 = 0.5f * (c0 + c1) + c2; }
        const int co = (int)(i / (size_t)Cin), ci = (int)(i - (size_t)co * Cin);
        const int r = a + j, ky = a + 3 - 2 * r;
        float *dst = dwT + (((size_t)ci * Cout + co) * 4 + ky) * 4;
        dst[b + 3 - 2 * b] = g0;                             // tap column c = b
        dst[b + 3 - 2 * (b + 1)] = g1;                       // tap column c = b + 1
    }
}

// nn.Upsample(2, nearest) + Conv2d(3x3, pad 1) IS a ConvTranspose2d(k4, s2, p1) whose phase taps are sums of the 3x3 taps (output row
// 2 i + a reads source rows {i - 1, i, i} (a = 0) or {i, i, i + 1} (a = 1): dream_upsample_conv3x3_weight_as_convT4x4), so the 3x3
// kernel's gradient is the matching sum of the transposed conv's: dw3[r][c] = sum over ky in K(r), kx in K(c) of dwT[ky][kx] with
// K(0) = {2, 3}, K(1) = {1, 2}, K(2) = {0, 1} (ky = a + 3 - 2 r' of phase a's tap row r').  dwT [Cin][Cout][4][4] -> dw OIHW [Cout][Cin][3][3].
__global__ void __launch_bounds__(256) convT4x4_grad_to_conv3x3_kernel(const float *dwT, float *dw, int Cin, int Cout) {
    const size_t n = (size_t)Cin * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i / (size_t)Cin), ci = (int)(i - (size_t)co * Cin);
        const float *t = dwT + ((size_t)ci * Cout + co) * 16;
        float v[4][4];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k >> 2][k & 3] = t[k];
        float *o = dw + i * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ky = 2 - r, kx = 2 - c;               // K(r) = {2 - r, 3 - r}
                o[3 * r + c] = (v[ky][kx] + v[ky][kx + 1]) + (v[ky + 1][kx] + v[ky + 1][kx + 1]);
            }
    }
}

struct PlanLds { int nsplit, tiles_per_split, ntiles, TY, TX; };

// one workgroup per CU (131 KB of LDS): the grid is (channel blocks) x (splits) ~ 256 workgroups of equal work
PlanLds make_plan_lds(int B, int H, int W, int Cin, int Cout) {
    PlanLds pl;
    pl.TY = (H + 1) / 2; pl.TX = (W + 1) / 2;
    pl.ntiles = B * pl.TY * pl.TX;
    const int blocks = (Cout / 64) * (Cin / 64);
    int want = (int)((wgrad_target_workgroups(256) + blocks / 2) / blocks);
    const int max_by_work = (pl.ntiles + 16 * LT - 1) / (16 * LT);          // at least 16 stages per split
    if (want > max_by_work) want = max_by_work;
    if (want < 1) want = 1;
    const size_t img_bytes = (size_t)H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4;
    const long tiles_per_img = (long)pl.TY * pl.TX;
    while (((size_t)((pl.ntiles + want - 1) / want / tiles_per_img) + 2) * img_bytes >= ((size_t)1 << 30) && want < pl.ntiles) want *= 2;
    pl.tiles_per_split = ((pl.ntiles + want - 1) / want + 2 * LT - 1) / (2 * LT) * (2 * LT);
    pl.nsplit = (pl.ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    return pl;
}

bool use_lds_version(int Cin, int Cout, int Cdy) { return g_wgw_version != 1 && Cin % 64 == 0 && Cout % 64 == 0 && Cdy % 4 == 0; }

struct Plan { int nsplit, tiles_per_split, ntiles, TY, TX; };

Plan make_plan(int B, int H, int W, int Cin, int Cout) {
    Plan pl;
    pl.TY = (H + 1) / 2; pl.TX = (W + 1) / 2;
    pl.ntiles = B * pl.TY * pl.TX;
    const int blocks = ((Cout + 16 * NCO - 1) / (16 * NCO)) * (Cin / 64);
    int want = (int)((wgrad_target_workgroups(1024) + blocks - 1) / blocks);     // ~4 workgroups per CU in total at full width
    const int max_by_work = (pl.ntiles + 255) / 256;               // at least 64 k-steps per split
    if (want > max_by_work) want = max_by_work;
    if (want < 1) want = 1;
    // 32-bit byte offsets relative to the first image of a split: keep a split's images under 1 GB
    const size_t img_bytes = (size_t)H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4;
    const long tiles_per_img = (long)pl.TY * pl.TX;
    while (((size_t)((pl.ntiles + want - 1) / want / tiles_per_img) + 2) * img_bytes >= ((size_t)1 << 30) && want < pl.ntiles) want *= 2;
    pl.tiles_per_split = ((pl.ntiles + want - 1) / want + 3) / 4 * 4;
    pl.nsplit = (pl.ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    return pl;
}

}  // namespace

extern "C" size_t dream_conv3x3_wgrad_winograd_workspace(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0) return 0;
    const Plan pl = make_plan(B, H, W, Cin, Cout);
    size_t bytes = (size_t)pl.nsplit * 9 * Cout * Cin * sizeof(float);
    if (Cout % 64 == 0) {                               // the LDS version keeps all 16 positions per split
        const PlanLds pl2 = make_plan_lds(B, H, W, Cin, Cout);
        const size_t b2 = (size_t)pl2.nsplit * (16 * (size_t)Cout * Cin + Cout) * sizeof(float);      // + the bias partials
        if (b2 > bytes) bytes = b2;
    }
    return bytes;
}

// Test / A-B hook: 0 = choose by shape (default), 1 = always the register-only version.
extern "C" int dream_conv3x3_wgrad_winograd_set_version(int version) {
    DREAM_REQUIRE(version == 0 || version == 1, "winograd wgrad: version %d", version);
    g_wgw_version = version;
    return 0;
}

// x [B,H,W,Cin] (or [B,H/2,W/2,Cin] with DREAM_CONV_UPSAMPLE2X: the conv that follows nn.Upsample(2)), dy [B,H,W,Cdy]
// (Cdy >= Cout) NHWC -> dw_oihw [Cout,Cin,3,3] (overwritten).  Cin % 64 == 0, Cout % 16 == 0.
// workspace: dream_conv3x3_wgrad_winograd_workspace() bytes.
// dbias [Cout] (optional): the bias gradient = column sums of dy, accumulated in the dy loader of the LDS kernel -- only where
// dream_conv3x3_wgrad_winograd_fuses_bias() says so; elsewhere it is dream_channel_sum_nhwc_f32(dy) and a non-null dbias is an error.
extern "C" int dream_conv3x3_wgrad_winograd_fuses_bias(int Cin, int Cout, int Cdy) {
    return Cin > 0 && Cout > 0 && Cdy >= Cout && use_lds_version(Cin, Cout, Cdy) ? 1 : 0;
}

extern "C" int dream_conv3x3_wgrad_winograd_bias_nhwc_f32(const float *x, const float *dy, float *dw_oihw, float *dbias, void *workspace,
                                                          int B, int H, int W, int Cin, int Cout, int Cdy, int flags, void *stream) {
    DREAM_REQUIRE(x && dy && dw_oihw && workspace, "winograd wgrad: null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cdy >= Cout, "winograd wgrad: bad shape");
    DREAM_REQUIRE(Cin % 64 == 0 && Cout % 16 == 0 && Cdy % 16 == 0, "winograd wgrad: Cin %% 64, Cout %% 16, Cdy %% 16 (got %d, %d, %d)", Cin, Cout, Cdy);
    DREAM_REQUIRE((flags & ~DREAM_CONV_UPSAMPLE2X) == 0, "winograd wgrad: unsupported flags 0x%x", flags);
    const bool ups = (flags & DREAM_CONV_UPSAMPLE2X) != 0;      // x is [B,H/2,W/2,Cin], the conv ran on its nearest x2 upsample
    DREAM_REQUIRE(!ups || (H % 2 == 0 && W % 2 == 0 && Cin % 64 == 0 && Cout % 64 == 0 && g_wgw_version != 1),
                  "winograd wgrad: the fused upsample needs even extents and the LDS kernel (channels a multiple of 64)");
    if (use_lds_version(Cin, Cout, Cdy)) {
        const PlanLds pl = make_plan_lds(B, H, W, Cin, Cout);
        DREAM_REQUIRE((long)B * pl.TY * pl.TX < ((long)1 << 24), "winograd wgrad: too many tiles");
        // the kernel's out-of-range words (OOB_LANE / OOB_WAVE) need the byte offsets inside a split's span of images below 2^30
        DREAM_REQUIRE(((size_t)(pl.tiles_per_split / ((long)pl.TY * pl.TX)) + 2) * (size_t)H * W * (size_t)(Cin > Cdy ? Cin : Cdy) * 4 < ((size_t)1 << 30),
                      "winograd wgrad: a split's span of images is too large for 32-bit offsets");
        WgWinoLdsParams p;
        p.x = x; p.dy = dy; p.partial = (float *)workspace;
        p.bias_partial = p.partial + (size_t)pl.nsplit * 16 * Cout * Cin;
        p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Cdy = Cdy;
        p.TY = pl.TY; p.TX = pl.TX; p.ntiles = pl.ntiles;
        p.magic_tpi = (((unsigned long long)1 << 40) + (unsigned long long)(pl.TY * pl.TX) - 1) / (unsigned long long)(pl.TY * pl.TX);
        p.magic_tx = (((unsigned long long)1 << 40) + (unsigned long long)pl.TX - 1) / (unsigned long long)pl.TX;
        p.tiles_per_split = pl.tiles_per_split;
        p.ncog = Cout / 64;
        p.nsplit = pl.nsplit;
        void (*const kernels[4])(WgWinoLdsParams) = {wgrad_wino_lds_kernel<false, false>, wgrad_wino_lds_kernel<true, false>,
                                                     wgrad_wino_lds_kernel<false, true>, wgrad_wino_lds_kernel<true, true>};
        for (int v = 0; v < 4; ++v)
            if (dream_allow_full_lds((const void *)kernels[v])) return 2;
        const dim3 grid((unsigned)(p.ncog * (Cin / 64)), (unsigned)pl.nsplit);
        hipLaunchKernelGGL(kernels[(ups ? 1 : 0) + (dbias ? 2 : 0)], grid, dim3(512), L_LDS_BYTES, (hipStream_t)stream, p);
        DREAM_LAUNCH_OK();
        // one workgroup per 64 weights; the first ceil(Cout / 256) of them also sum the bias partials
        hipLaunchKernelGGL(wgrad_wino_lds_reduce_kernel, dim3((unsigned)((size_t)Cout * Cin / 64)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)workspace, dw_oihw, (const float *)p.bias_partial, dbias, pl.nsplit, Cout, Cin);
        DREAM_LAUNCH_OK();
        return 0;
    }
    DREAM_REQUIRE(dbias == nullptr, "winograd wgrad: the bias gradient is fused in the LDS kernel only (see dream_conv3x3_wgrad_winograd_fuses_bias)");
    const Plan pl = make_plan(B, H, W, Cin, Cout);
    DREAM_REQUIRE((long)B * pl.TY * pl.TX < ((long)1 << 24), "winograd wgrad: too many tiles");
    WgWinoParams p;
    p.x = x; p.dy = dy; p.partial = (float *)workspace;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Cdy = Cdy;
    p.TY = pl.TY; p.TX = pl.TX; p.ntiles = pl.ntiles;
    p.magic_tpi = (((unsigned long long)1 << 40) + (unsigned long long)(pl.TY * pl.TX) - 1) / (unsigned long long)(pl.TY * pl.TX);
    p.magic_tx = (((unsigned long long)1 << 40) + (unsigned long long)pl.TX - 1) / (unsigned long long)pl.TX;
    p.tiles_per_split = pl.tiles_per_split;
    p.ncog = (Cout + 16 * NCO - 1) / (16 * NCO);
    const dim3 grid((unsigned)(p.ncog * (Cin / 64)), (unsigned)pl.nsplit);
    hipLaunchKernelGGL(wgrad_wino_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    size_t rgrid = ((size_t)Cout * Cin + 255) / 256;
    if (rgrid > 2048) rgrid = 2048;
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, (hipStream_t)stream, (const float *)workspace,
                       dw_oihw, pl.nsplit, Cout, Cin);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_conv3x3_wgrad_winograd_nhwc_f32(const float *x, const float *dy, float *dw_oihw, void *workspace, int B, int H,
                                                     int W, int Cin, int Cout, int Cdy, int flags, void *stream) {
    return dream_conv3x3_wgrad_winograd_bias_nhwc_f32(x, dy, dw_oihw, nullptr, workspace, B, H, W, Cin, Cout, Cdy, flags, stream);
}

// ---- nn.ConvTranspose2d(k4, s2, p1) weight gradient by minimal filtering (round 6; dream/models.py:37-136 via network.py:335) -----------
// x [B,H,W,Cin], dy [B,2H,2W,Cout] NHWC -> dwT [Cin][Cout][4][4] (the module's own layout, overwritten), dbias [Cout] or null
// (= column sums of dy, summed in the dy loader).  Cin % 64 == 0, Cout % 64 == 0.  ONE launch of the nine-position kernel over
// (channel blocks) x (splits) x (four phases) + one reduction launch; 9 multiplications per 2 x 2 outputs of a phase where
// dream_convT4x4_wgrad_nhwc_f32 (the direct kernel) takes 16.
namespace {
PlanLds make_plan_convT(int B, int H, int W, int Cin, int Cout) {
    PlanLds pl;
    pl.TY = (H + 1) / 2; pl.TX = (W + 1) / 2;
    pl.ntiles = B * pl.TY * pl.TX;
    const int blocks = (Cout / 64) * (Cin / 64) * 4;                          // the four phases run side by side
    int want = (int)((wgrad_target_workgroups(256) + blocks / 2) / blocks);
    const int max_by_work = (pl.ntiles + 16 * LT - 1) / (16 * LT);          // at least 16 stages per split
    if (want > max_by_work) want = max_by_work;
    if (want < 1) want = 1;
    // 32-bit byte offsets relative to the first image of a split: dy is stored at 2H x 2W
    const size_t img_bytes = (size_t)4 * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4;
    const long tiles_per_img = (long)pl.TY * pl.TX;
    while (((size_t)((pl.ntiles + want - 1) / want / tiles_per_img) + 2) * img_bytes >= ((size_t)1 << 30) && want < pl.ntiles) want *= 2;
    pl.tiles_per_split = ((pl.ntiles + want - 1) / want + 2 * LT - 1) / (2 * LT) * (2 * LT);
    pl.nsplit = (pl.ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    return pl;
}
}  // namespace

extern "C" int dream_convT4x4_wgrad_winograd_applies(int Cin, int Cout) { return Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0 ? 1 : 0; }

extern "C" size_t dream_convT4x4_wgrad_winograd_workspace(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || !dream_convT4x4_wgrad_winograd_applies(Cin, Cout)) return 0;
    const PlanLds pl = make_plan_convT(B, H, W, Cin, Cout);
    return (size_t)4 * pl.nsplit * (9 * (size_t)Cout * Cin + Cout) * sizeof(float);
}

extern "C" int dream_convT4x4_wgrad_winograd_nhwc_f32(const float *x, const float *dy, float *dwT, float *dbias, void *workspace, int B,
                                                      int H, int W, int Cin, int Cout, void *stream) {
    DREAM_REQUIRE(x && dy && dwT && workspace, "winograd convT wgrad: null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0, "winograd convT wgrad: bad shape");
    DREAM_REQUIRE(dream_convT4x4_wgrad_winograd_applies(Cin, Cout), "winograd convT wgrad: Cin %% 64, Cout %% 64 (got %d, %d)", Cin, Cout);
    const PlanLds pl = make_plan_convT(B, H, W, Cin, Cout);
    DREAM_REQUIRE((long)B * pl.TY * pl.TX < ((long)1 << 24), "winograd convT wgrad: too many tiles");
    DREAM_REQUIRE((size_t)4 * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 29), "winograd convT wgrad: image too large for 32-bit offsets");
    DREAM_REQUIRE(((size_t)(pl.tiles_per_split / ((long)pl.TY * pl.TX)) + 2) * (size_t)4 * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 4 < ((size_t)1 << 30),
                  "winograd convT wgrad: a split's span of images is too large for 32-bit offsets");
    WgWinoLdsParams p;
    p.x = x; p.dy = dy; p.partial = (float *)workspace;
    p.bias_partial = p.partial + (size_t)4 * pl.nsplit * 9 * Cout * Cin;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Cdy = Cout;
    p.TY = pl.TY; p.TX = pl.TX; p.ntiles = pl.ntiles;
    p.magic_tpi = (((unsigned long long)1 << 40) + (unsigned long long)(pl.TY * pl.TX) - 1) / (unsigned long long)(pl.TY * pl.TX);
    p.magic_tx = (((unsigned long long)1 << 40) + (unsigned long long)pl.TX - 1) / (unsigned long long)pl.TX;
    p.tiles_per_split = pl.tiles_per_split;
    p.ncog = Cout / 64;
    p.nsplit = pl.nsplit;
    void (*const kernels[2])(WgWinoLdsParams) = {wgrad_wino_lds_kernel<false, false, true>, wgrad_wino_lds_kernel<false, true, true>};
    for (int v = 0; v < 2; ++v)
        if (dream_allow_full_lds((const void *)kernels[v])) return 2;
    const dim3 grid((unsigned)(p.ncog * (Cin / 64)), (unsigned)pl.nsplit, 4);
    hipLaunchKernelGGL(kernels[dbias ? 1 : 0], grid, dim3(512), L_LDS_BYTES, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(wgrad_wino_convT_reduce_kernel, dim3((unsigned)((size_t)Cout * Cin / 64), 4), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, dwT, (const float *)p.bias_partial, dbias, pl.nsplit, Cout, Cin);
    DREAM_LAUNCH_OK();
    return 0;
}

// dwT [Cin][Cout][4][4] (gradient of the transposed conv that an upsample + 3x3 conv is) -> dw OIHW [Cout][Cin][3][3]
extern "C" int dream_upsample_conv3x3_wgrad_from_convT4x4(const float *dwT, float *dw_oihw, int Cin, int Cout, void *stream) {
    DREAM_REQUIRE(dwT && dw_oihw && Cin > 0 && Cout > 0, "upsample-conv wgrad combine: bad arguments");
    size_t grid = ((size_t)Cin * Cout + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(convT4x4_grad_to_conv3x3_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, dwT, dw_oihw, Cin, Cout);
    DREAM_LAUNCH_OK();
    return 0;
}

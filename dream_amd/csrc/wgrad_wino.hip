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
// 5.1 M for 128 x 400 x 400).  v_mfma_f32_16x16x4_f32: lane l supplies A[co = l & 15][k = l >> 4] and B[k = l >> 4][ci = l & 15],
// so a lane owns ONE tile and ONE channel of each operand per k-step: it loads the tile's 2x2 dy values (one channel) and
// its 4x4 input patch (one channel) as scalars -- NHWC makes the 16 lanes of a channel group read one contiguous 64-byte
// segment per pixel -- and transforms them IN REGISTERS into the 16 + 16 operands of the 16 positions' MFMAs.  No LDS at all.
//   * wave tile = 32 output channels x 16 input channels x 16 positions (2 x 16 accumulators, 128 VGPRs); workgroup = 4 waves
//     = 32 output x 64 input channels; the four waves load the same dy (L1 hits) and their own input channels;
//   * split-K over tile ranges, partial 3x3 gradients (the linear map G^T . G applied per workgroup) to a workspace,
//     summed in a fixed order by a second kernel: deterministic, no fp32 atomics;
//   * buffer loads with hardware bounds masking (zero padding, odd extents, range tails); an interior fast path skips the
//     per-load validity selects when every tile of the k-step lies inside the image;
//   * loads of k-step s+1 are issued before the 32 MFMAs of step s; the negations of A dY A^T are folded into the final
//     transform (signs of dU positions), so the operand transforms are 12 + 32 additions per lane and k-step.
// Measured (profiles/r02_microbench_wgrad_wino_b128.txt): 1.15-1.67x the direct kernel on the layers it is used for, 0.50-0.55 of
// the MFMA peak on its own multiplications.  What bounds it (knock-out builds, profiles/r02_wgw_diag.txt): with the loads
// removed the same loop runs at 0.79-0.93, with the transforms removed nothing changes -- the 24 scalar loads per 32 MFMAs
// (64-byte segments, 256 B per wave instruction) saturate the texture-address path; variants with less address arithmetic,
// loads two steps ahead or loads spread between the MFMAs were all slower.  The fix is structural (share the loaded patch
// across the workgroup through LDS, float4 loads: ~0.1 load instructions per MFMA instead of 0.75) and is the next version.
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

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

// dw_oihw[co][ci][tap] = sum over splits (fixed order) of partial[s][tap][co][ci]
__global__ void __launch_bounds__(256) wgrad_wino_reduce_kernel(const float *partial, float *dw, int nsplit, int Cout, int Cin) {
    const size_t n = (size_t)Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float s = 0.0f;
            for (int k = 0; k < nsplit; ++k) s += partial[((size_t)k * 9 + tap) * n + i];
            dw[i * 9 + tap] = s;
        }
    }
}

struct Plan { int nsplit, tiles_per_split, ntiles, TY, TX; };

Plan make_plan(int B, int H, int W, int Cin, int Cout) {
    Plan pl;
    pl.TY = (H + 1) / 2; pl.TX = (W + 1) / 2;
    pl.ntiles = B * pl.TY * pl.TX;
    const int blocks = ((Cout + 16 * NCO - 1) / (16 * NCO)) * (Cin / 64);
    int want = (1024 + blocks - 1) / blocks;                       // ~4 workgroups per CU in total
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
    return (size_t)pl.nsplit * 9 * Cout * Cin * sizeof(float);
}

// x [B,H,W,Cin], dy [B,H,W,Cdy] (Cdy >= Cout) NHWC -> dw_oihw [Cout,Cin,3,3] (overwritten).  Cin % 64 == 0, Cout % 16 == 0.
// workspace: dream_conv3x3_wgrad_winograd_workspace() bytes.  (The bias gradient is dream_channel_sum_nhwc_f32(dy).)
extern "C" int dream_conv3x3_wgrad_winograd_nhwc_f32(const float *x, const float *dy, float *dw_oihw, void *workspace, int B, int H,
                                                     int W, int Cin, int Cout, int Cdy, void *stream) {
    DREAM_REQUIRE(x && dy && dw_oihw && workspace, "winograd wgrad: null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cdy >= Cout, "winograd wgrad: bad shape");
    DREAM_REQUIRE(Cin % 64 == 0 && Cout % 16 == 0 && Cdy % 16 == 0, "winograd wgrad: Cin %% 64, Cout %% 16, Cdy %% 16 (got %d, %d, %d)", Cin, Cout, Cdy);
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

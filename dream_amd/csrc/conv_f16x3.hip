// Split-precision ("fp16x3") implicit-GEMM convolution: fp32 in, fp32 out, fp32-class accuracy, on the fp16 matrix
// cores (2.5 PFLOP/s dense) instead of the fp32 ones (157 TFLOP/s).
//
// Every operand is split into two halfs,  v*2^e = hi + lo  (hi = fp16(v*2^e), lo = fp16(v*2^e - hi)),  which
// together carry 22 mantissa bits; a product a*b is evaluated as  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b  (three
// v_mfma_f32_32x32x16_f16, fp32 accumulation; the dropped lo*lo term is 2^-22 relative).  The power-of-two scales
// 2^ea (activations) / 2^ew (weights) place the largest magnitude of each tensor in [2^13, 2^14) so that hi and lo
// stay normal fp16 numbers for everything within 2^17 of the tensor maximum (smaller elements keep an ABSOLUTE
// accuracy of 2^-39 of the maximum); the scales are exact and are divided out in the epilogue.
//   * activations stay fp32 in HBM; they are split while the patch is staged into LDS, using the per-tensor
//     max|x| that the PRODUCING kernel published with an atomicMax (amax side channel, no extra pass);
//   * weights are pre-split by dream_pack_conv_weight_f16x3 into two fp16 planes + the exponent ew.
// Structure, tiling, tap table, fused upsample / zero-stuffing and the epilogue are those of conv_mfma.hip (same
// reference call sites: dream/models.py:594-615, 695-747); 3 MFMAs replace 8 fp32 MFMAs per 16 k's: 5.3x the MFMA
// rate of the fp32 kernel, so the measured error must justify it: tests/parity (fp64 reference) show <= 2e-6
// relative per layer, i.e. the same class as the fp32 kernel's own summation-order noise.
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

struct Conv16Params {
    const float *x;
    const _Float16 *w_hi;    // [ntaps][CoutPad][Cin]
    const _Float16 *w_lo;
    const int *w_exp;        // device scalar: weights were multiplied by 2^w_exp before the split
    const unsigned *amax_in; // device scalar: bit pattern of max|x| of the input tensor
    const float *scale;
    const float *shift;
    const float *residual;
    float *y;
    unsigned *amax_out;
    int B, H, W, Hin, Win, Hs, Ws, Ho, Wo;
    int Cin, Cout, CoutPad;
    int TH, TW, PH, PW, tiles_x, tiles_y, rcpTW;
    int in_scale, in_step, lane_stride, pad_y, pad_x;
    unsigned long long tap_w;   // 16 x 4-bit: weight slice of each tap of this launch
    int ntaps;
    unsigned long long tap_dy, tap_dx;
    int out_scale, out_oy, out_ox;
    int flags;
};

namespace {

constexpr int KC = 32;              // k's per stage (two 32x32x16 MFMA k-steps)
constexpr int S16 = KC + 8;         // LDS row stride in halfs (80 B: odd number of 16-B slots)

DREAM_DEVICE float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

template <int MR, int NR, int WM, int WN, int NPM, bool PRIO = false>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_f16x3_kernel(const Conv16Params p) {
    constexpr int NT = 64 * WM * WN;                // 4 or 8 wavefronts per workgroup
    constexpr int BN = 32 * NR * WN;
    constexpr int Q = KC / 4;                       // float4 pieces per patch row
    constexpr int NA_IT = (NPM * Q + NT - 1) / NT;
    constexpr int NB_PIECES = BN * (KC / 8);        // 16-B pieces per weight plane per stage
    constexpr int NB_IT = (NB_PIECES + NT - 1) / NT;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 wavefronts per workgroup");

    DREAM_DYNAMIC_LDS(_Float16, smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int PW = p.PW, TW = p.TW, NP = p.PH * PW;
    _Float16 *sAh = smem;
    _Float16 *sAl = sAh + NP * S16;
    _Float16 *sB = sAl + NP * S16;                  // [buf][plane][BN][S16]

    // XCD-aware placement (see conv_mfma.hip): each XCD works on a contiguous range of tiles so halos meet in its L2
    int t = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (t >= p.B * p.tiles_x * p.tiles_y) return;
    const int tix = t % p.tiles_x;
    t /= p.tiles_x;
    const int tiy = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int y0 = tiy * p.TH, x0 = tix * TW;
    const int n0 = blockIdx.y * BN;
    const bool zst = (p.flags & DREAM_CONV_ZEROSTUFF2X) != 0;
    const bool ups = (p.flags & DREAM_CONV_UPSAMPLE2X) != 0 || zst;
    const bool pool = (p.flags & DREAM_CONV_POOL2) != 0;
    const float *xb = p.x + (size_t)b * p.Hs * p.Ws * p.Cin;

    // input scale: max|x| * 2^ea in [2^13, 2^14)
    const unsigned abits = *p.amax_in;
    const int aexp = (int)((abits >> 23) & 255) - 127;
    int ea = (abits == 0u) ? 0 : 13 - aexp;
    ea = ea < -100 ? -100 : (ea > 100 ? 100 : ea);
    const float sa = pow2f(ea);
    const float inv = pow2f(-(ea + *p.w_exp) < -126 ? -126 : (-(ea + *p.w_exp) > 127 ? 127 : -(ea + *p.w_exp)));

    int a_goff[NA_IT], a_soff[NA_IT];
#pragma unroll
    for (int it = 0; it < NA_IT; ++it) {
        const int idx = tid + it * NT;
        const int pp = idx / Q, q = idx % Q;
        a_soff[it] = (pp < NP) ? pp * S16 + q * 4 : -1;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 * p.in_scale - p.pad_y + py * p.in_step, gx = x0 * p.in_scale - p.pad_x + px * p.in_step;
        const bool inb = (pp < NP) && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win && !(zst && ((gy | gx) & 1));
        const int sy = ups ? (gy >> 1) : gy, sx = ups ? (gx >> 1) : gx;
        a_goff[it] = inb ? (sy * p.Ws + sx) * p.Cin + q * 4 : -1;
    }
    int b_goff[NB_IT], b_soff[NB_IT];
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
        const int idx = tid + it * NT;
        const int n = idx / (KC / 8), q = idx % (KC / 8);
        b_soff[it] = (idx < NB_PIECES) ? n * S16 + q * 8 : -1;
        b_goff[it] = (n0 + n) * p.Cin + q * 8;
    }
    const size_t w_tap_stride = (size_t)p.CoutPad * p.Cin;

    int a_frag[MR], b_frag[NR];
#pragma unroll
    for (int ms = 0; ms < MR; ++ms) {
        int m = (wm * MR + ms) * 32 + li;
        if (m >= p.TH * TW) m = 0;
        int ty, tx;
        tile_xy(m, TW, p.rcpTW, pool, &ty, &tx);
        a_frag[ms] = (ty * PW + tx) * p.lane_stride * S16 + lh * 8;
    }
#pragma unroll
    for (int ns = 0; ns < NR; ++ns) b_frag[ns] = ((wn * NR + ns) * 32 + li) * S16 + lh * 8;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int ms = 0; ms < MR; ++ms)
#pragma unroll
        for (int ns = 0; ns < NR; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.0f;

    f32x4 a_reg[NA_IT];
    f16x8 bh_reg[NB_IT], bl_reg[NB_IT];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    auto load_a = [&](int c0) {
#pragma unroll
        for (int it = 0; it < NA_IT; ++it)
            a_reg[it] = (a_goff[it] >= 0) ? *(const f32x4 *)(xb + a_goff[it] + c0) : zero4;
    };
    auto store_a = [&]() {     // split v*2^ea into hi + lo while writing the patch
#pragma unroll
        for (int it = 0; it < NA_IT; ++it) {
            if (a_soff[it] >= 0) {
                f16x4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = a_reg[it][k] * sa;
                    hi[k] = (_Float16)v;
                    lo[k] = (_Float16)(v - (float)hi[k]);
                }
                *(f16x4 *)(sAh + a_soff[it]) = hi;
                *(f16x4 *)(sAl + a_soff[it]) = lo;
            }
        }
    };
    auto load_b = [&](int tap, int c0) {
        const size_t base = (size_t)((p.tap_w >> (4 * tap)) & 15) * w_tap_stride + c0;
#pragma unroll
        for (int it = 0; it < NB_IT; ++it)
            if (b_soff[it] >= 0) {
                bh_reg[it] = *(const f16x8 *)(p.w_hi + base + b_goff[it]);
                bl_reg[it] = *(const f16x8 *)(p.w_lo + base + b_goff[it]);
            }
    };
    auto store_b = [&](int buf) {
        _Float16 *dh = sB + (buf * 2 + 0) * BN * S16, *dl = sB + (buf * 2 + 1) * BN * S16;
#pragma unroll
        for (int it = 0; it < NB_IT; ++it)
            if (b_soff[it] >= 0) {
                *(f16x8 *)(dh + b_soff[it]) = bh_reg[it];
                *(f16x8 *)(dl + b_soff[it]) = bl_reg[it];
            }
    };

    const int nchunks = p.Cin / KC;
    load_a(0);
    load_b(0, 0);
    store_a();
    store_b(0);
    __syncthreads();

    int buf = 0, tap = 0, chunk = 0;
    const int ntaps = p.ntaps, nstages = nchunks * ntaps;
    for (int st = 0; st < nstages; ++st) {
        const bool last_tap = (tap == ntaps - 1);
        const bool more_chunks = (chunk + 1 < nchunks);
        const bool have_next = (st + 1 < nstages);
        if (have_next) load_b(last_tap ? 0 : tap + 1, last_tap ? (chunk + 1) * KC : chunk * KC);
        if (last_tap && more_chunks) load_a((chunk + 1) * KC);

        const int tdy = (int)((p.tap_dy >> (4 * tap)) & 15), tdx = (int)((p.tap_dx >> (4 * tap)) & 15);
        const int toff = (tdy * PW + tdx) * S16;
        const _Float16 *bh = sB + (buf * 2 + 0) * BN * S16, *bl = sB + (buf * 2 + 1) * BN * S16;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 16) {
            f16x8 ah[MR], al[MR], wh[NR], wl[NR];
#pragma unroll
            for (int ms = 0; ms < MR; ++ms) {
                ah[ms] = *(const f16x8 *)(sAh + toff + a_frag[ms] + kk);
                al[ms] = *(const f16x8 *)(sAl + toff + a_frag[ms] + kk);
            }
#pragma unroll
            for (int ns = 0; ns < NR; ++ns) {
                wh[ns] = *(const f16x8 *)(bh + b_frag[ns] + kk);
                wl[ns] = *(const f16x8 *)(bl + b_frag[ns] + kk);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(1);     // co-resident waves of the other workgroup are in their load phase
#pragma unroll
            for (int ms = 0; ms < MR; ++ms)
#pragma unroll
                for (int ns = 0; ns < NR; ++ns) {
                    acc[ms][ns] = mfma_f32_32x32x16_f16(al[ms], wh[ns], acc[ms][ns]);
                    acc[ms][ns] = mfma_f32_32x32x16_f16(ah[ms], wl[ns], acc[ms][ns]);
                    acc[ms][ns] = mfma_f32_32x32x16_f16(ah[ms], wh[ns], acc[ms][ns]);
                }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }

        if (have_next) store_b(buf ^ 1);
        if (last_tap && more_chunks) {
            __syncthreads();
            store_a();
        }
        __syncthreads();
        buf ^= 1;
        if (last_tap) { tap = 0; ++chunk; } else ++tap;
    }

    // ---- epilogue -----------------------------------------------------------------------------------------
    const bool relu = (p.flags & DREAM_CONV_RELU) != 0;
    const bool nchw = (p.flags & DREAM_CONV_OUT_NCHW) != 0;
    float scale_v[NR], shift_v[NR];
    int ncol[NR];
#pragma unroll
    for (int ns = 0; ns < NR; ++ns) {
        ncol[ns] = n0 + (wn * NR + ns) * 32 + li;
        const bool cok = ncol[ns] < p.Cout;
        scale_v[ns] = inv * ((p.scale != nullptr && cok) ? p.scale[ncol[ns]] : 1.0f);
        shift_v[ns] = (p.shift != nullptr && cok) ? p.shift[ncol[ns]] : 0.0f;
    }
    const int npix = p.TH * TW;
    float amax = 0.0f;
#pragma unroll
    for (int ms = 0; ms < MR; ++ms) {
        if (!pool) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * MR + ms) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int ty = (m * p.rcpTW) >> 16, tx = m - ty * TW;
                const int oy = (y0 + ty) * p.out_scale + p.out_oy, ox = (x0 + tx) * p.out_scale + p.out_ox;
                const bool ok = (m < npix) && (y0 + ty < p.H) && (x0 + tx < p.W) && oy < p.Ho && ox < p.Wo;
#pragma unroll
                for (int ns = 0; ns < NR; ++ns) {
                    if (ok && ncol[ns] < p.Cout) {
                        const size_t o = nchw
                            ? (((size_t)b * p.Cout + ncol[ns]) * p.Ho + oy) * p.Wo + ox
                            : (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + ncol[ns];
                        float v = acc[ms][ns][r] * scale_v[ns] + shift_v[ns];
                        if (p.residual != nullptr) v = v + p.residual[o];
                        if (relu) v = fmaxf(v, 0.0f);
                        p.y[o] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                }
            }
        } else {
            // fused MaxPool2d(2): registers 4g..4g+3 of a lane are one 2x2 window (window-major tile order)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int m0 = (wm * MR + ms) * 32 + 8 * g4 + 4 * lh;
                const int q = m0 >> 2, hw = TW >> 1;
                const int wy = (q * p.rcpTW) >> 16, wx = q - wy * hw;
                const bool ok = (m0 < npix) && (y0 + 2 * wy + 1 < p.H) && (x0 + 2 * wx + 1 < p.W);
                const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
#pragma unroll
                for (int ns = 0; ns < NR; ++ns) {
                    if (ok && ncol[ns] < p.Cout) {
                        float best = -__builtin_huge_valf();
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = acc[ms][ns][4 * g4 + j] * scale_v[ns] + shift_v[ns];
                            if (relu) v = fmaxf(v, 0.0f);
                            best = fmaxf(best, v);
                        }
                        p.y[(((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + ncol[ns]] = best;
                        amax = fmaxf(amax, fabsf(best));
                    }
                }
            }
        }
    }
    if (p.amax_out != nullptr) publish_amax(p.amax_out, amax);
}

// ---- weight packing: amax -> exponent -> two fp16 planes ----------------------------------------------------
__global__ void __launch_bounds__(256) absmax_kernel(const float *x, size_t n, unsigned *out) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, lane_xor(m, k));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ void __launch_bounds__(256) pack_w16_kernel(const float *w, _Float16 *hi, _Float16 *lo, const unsigned *amax,
                                                       int *exp_out, int Cout, int Cin, int RowsPad, int ColsPad, int mode,
                                                       int ntaps) {
    const unsigned abits = *amax;
    const int e = (abits == 0u) ? 0 : 13 - ((int)((abits >> 23) & 255) - 127);
    const float s = pow2f(e < -100 ? -100 : (e > 100 ? 100 : e));
    if (blockIdx.x == 0 && threadIdx.x == 0) *exp_out = e < -100 ? -100 : (e > 100 ? 100 : e);
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
        v *= s;
        const _Float16 h = (_Float16)v;
        hi[idx] = h;
        lo[idx] = (_Float16)(v - (float)h);
    }
}

struct Variant16 {
    const char *name;
    int BM, BN, NP_MAX, threads;
    void (*kernel)(const Conv16Params);
};
const Variant16 kVariants16[] = {
    {"f16x3 m2n2w2x2", 128, 128, 192, 256, conv_f16x3_kernel<2, 2, 2, 2, 192>},
    {"f16x3 m2n2w4x1", 256, 64, 352, 256, conv_f16x3_kernel<2, 2, 4, 1, 352>},
    {"f16x3 m2n1w4x1", 256, 32, 352, 256, conv_f16x3_kernel<2, 1, 4, 1, 352>},
    {"f16x3 m1n2w2x2", 64, 128, 128, 256, conv_f16x3_kernel<1, 2, 2, 2, 128>},
    {"f16x3 m2n2w4x2", 256, 128, 352, 512, conv_f16x3_kernel<2, 2, 4, 2, 352>},   // 8 waves, 1 workgroup per CU
    {"f16x3 m2n2w8x1", 512, 64, 640, 512, conv_f16x3_kernel<2, 2, 8, 1, 640>},    // 512 px x 64 cout
    {"f16x3 m2n2w4x1 prio", 256, 64, 352, 256, conv_f16x3_kernel<2, 2, 4, 1, 352, true>},   // A/B arm: s_setprio around the MFMAs
    {"f16x3 m2n2w4x2 prio", 256, 128, 352, 512, conv_f16x3_kernel<2, 2, 4, 2, 352, true>},
};
constexpr int kNum16 = 8;
int g_forced16 = -1;

void choose_tile16(int H, int W, int BM, int np_max, int lane_stride, int kext, bool even, int *th_out, int *tw_out) {
    long best_tiles = -1;
    int best_np = 0, bth = even ? 2 : 1, btw = even ? 2 : 1;
    const int He = even ? (H + 1) / 2 * 2 : H, We = even ? (W + 1) / 2 * 2 : W;
    // divisor (tw, or tw/2 with the fused pool) < 128 keeps the (m * rcpTW) >> 16 division exact for m < 512
    for (int tw = even ? 2 : 1; tw <= BM && tw <= (even ? 254 : 127); tw += even ? 2 : 1) {
        int th = BM / tw;
        if (even) th &= ~1;
        if (th < 1) break;
        if (th > He) th = He;
        const int twc = tw > We ? We : tw;
        const int np = ((th - 1) * lane_stride + kext) * ((twc - 1) * lane_stride + kext);
        if (np > np_max) continue;
        const long tiles = (long)ceil_div(H, th) * ceil_div(W, twc);
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && np < best_np)) {
            best_tiles = tiles; best_np = np; bth = th; btw = twc;
        }
    }
    *th_out = bth;
    *tw_out = btw;
}

}  // namespace

extern "C" int dream_conv_f16x3_set_variant(int v) {
    DREAM_REQUIRE(v >= -1 && v < kNum16, "variant out of range");
    g_forced16 = v;
    return 0;
}

extern "C" int dream_absmax_f32(const float *x, size_t n, unsigned *amax_out, void *stream) {
    DREAM_REQUIRE(x && amax_out && n > 0, "absmax: bad arguments");
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, amax_out);
    DREAM_LAUNCH_OK();
    return 0;
}

// OIHW fp32 -> two fp16 planes [ntaps][RowsPad][ColsPad] (hi, lo) of w * 2^(*exp_out); scratch: one zeroed uint32
extern "C" int dream_pack_conv_weight_f16x3(const float *w_oihw, void *hi, void *lo, int *exp_out, unsigned *scratch,
                                            int Cout, int Cin, int ntaps, int RowsPad, int ColsPad, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && hi && lo && exp_out && scratch && Cout > 0 && Cin > 0 && ntaps > 0 && (mode == 0 || mode == 1),
                  "pack_conv_weight_f16x3: bad arguments");
    DREAM_REQUIRE(RowsPad >= (mode == 0 ? Cout : Cin) && ColsPad >= (mode == 0 ? Cin : Cout), "pack_conv_weight_f16x3: padding too small");
    if (dream_zero_words(scratch, sizeof(unsigned), (hipStream_t)stream)) return 2;
    const size_t n = (size_t)Cout * Cin * ntaps;
    size_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, n, scratch);
    DREAM_LAUNCH_OK();
    const size_t total = (size_t)ntaps * RowsPad * ColsPad;
    g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(pack_w16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w_oihw, (_Float16 *)hi,
                       (_Float16 *)lo, (const unsigned *)scratch, exp_out, Cout, Cin, RowsPad, ColsPad, mode, ntaps);
    DREAM_LAUNCH_OK();
    return 0;
}

namespace {
struct Geom16 {
    int H, W, Hin, Win, Hs, Ws, Ho, Wo;      // position grid, logical / stored input extent, output extent
    int pad, kext, ntaps;
    int tap_dy[16], tap_dx[16];
    int out_scale, out_oy, out_ox;
    int tap_w[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
};

int launch16(const float *x, const unsigned *amax_in, const void *w_hi, const void *w_lo, const int *w_exp,
             const float *scale, const float *shift, const float *residual, float *y, unsigned *amax_out, int B, int Cin,
             int Cout, int CoutPad, const Geom16 &g, int flags, void *stream) {
    DREAM_REQUIRE(x && amax_in && w_hi && w_lo && w_exp && y, "conv_f16x3: null pointer");
    DREAM_REQUIRE(Cin % KC == 0, "conv_f16x3: Cin=%d must be a multiple of %d", Cin, KC);
    Conv16Params p;
    p.x = x; p.w_hi = (const _Float16 *)w_hi; p.w_lo = (const _Float16 *)w_lo; p.w_exp = w_exp; p.amax_in = amax_in;
    p.scale = scale; p.shift = shift; p.residual = residual; p.y = y; p.amax_out = amax_out;
    p.B = B; p.Hin = g.Hin; p.Win = g.Win; p.Hs = g.Hs; p.Ws = g.Ws; p.Ho = g.Ho; p.Wo = g.Wo; p.H = g.H; p.W = g.W;
    p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    const long pixels = (long)B * g.H * g.W;
    // measured (profiles/r01_microbench_f16x3.txt): the 256-px x 64-cout tile beats 128 x 128 on every layer (the
    // weight tile is the dominant LDS fill at this MFMA rate and is amortised over twice the pixels)
    int v = Cout > 32 ? 1 : 2;
    // interleaved A/B at B=128 (profiles/r01_ab_f16x3_b128.txt): from 128 channels on, the 8-wave 256 x 128 tile with
    // s_setprio around the MFMA cluster wins by 1-5 %; s_setprio alone on the 4-wave tile loses 2-5 %
    if (Cout >= 128 && Cin >= 128) v = 7;
    if (Cout > 64 && ((pixels + 255) / 256) * ceil_div(Cout, 64) < 512) v = 3;       // tiny grids: 64-px tiles
    if (g_forced16 >= 0) v = g_forced16;
    const Variant16 &var = kVariants16[v];
    DREAM_REQUIRE(CoutPad % var.BN == 0 && CoutPad >= Cout, "CoutPad=%d must be a multiple of %d", CoutPad, var.BN);
    const bool pool = (flags & DREAM_CONV_POOL2) != 0;
    DREAM_REQUIRE(!pool || (!(flags & DREAM_CONV_OUT_NCHW) && residual == nullptr && g.H >= 2 && g.W >= 2 && g.out_scale == 1),
                  "fused max-pool: NHWC output, no residual");
    choose_tile16(g.H, g.W, var.BM, var.NP_MAX, 1, g.kext, pool, &p.TH, &p.TW);
    p.PH = p.TH - 1 + g.kext; p.PW = p.TW - 1 + g.kext;
    p.tiles_x = ceil_div(g.W, p.TW); p.tiles_y = ceil_div(g.H, p.TH);
    p.rcpTW = pool ? (65536 + p.TW / 2 - 1) / (p.TW / 2) : (65536 + p.TW - 1) / p.TW;
    if (pool) { p.Ho = g.H / 2; p.Wo = g.W / 2; }
    p.in_scale = 1; p.in_step = 1; p.lane_stride = 1; p.pad_y = g.pad; p.pad_x = g.pad;
    p.ntaps = g.ntaps;
    p.tap_dy = 0; p.tap_dx = 0; p.tap_w = 0;
    for (int t = 0; t < g.ntaps; ++t) {
        p.tap_dy |= (unsigned long long)g.tap_dy[t] << (4 * t);
        p.tap_dx |= (unsigned long long)g.tap_dx[t] << (4 * t);
        p.tap_w |= (unsigned long long)g.tap_w[t] << (4 * t);
    }
    p.out_scale = g.out_scale; p.out_oy = g.out_oy; p.out_ox = g.out_ox;
    p.flags = flags;
    const size_t lds = ((size_t)2 * p.PH * p.PW + (size_t)4 * var.BN) * S16 * sizeof(_Float16);
    DREAM_REQUIRE(lds <= 160 * 1024, "LDS request %zu too large", lds);
    if (dream_allow_full_lds((const void *)var.kernel)) return 2;
    const dim3 grid((unsigned)(ceil_div((int)((size_t)B * p.tiles_x * p.tiles_y), 8) * 8), (unsigned)ceil_div(Cout, var.BN));
    hipLaunchKernelGGL(var.kernel, grid, dim3(var.threads), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

// ConvTranspose2d(k4,s2,p1) weight [Cin][Cout][4][4] -> two fp16 planes [phase][tap][RowsPad][ColsPad] of w * 2^exp
__global__ void __launch_bounds__(256) pack_wT4_16_kernel(const float *wT, _Float16 *hi, _Float16 *lo, const unsigned *amax,
                                                          int *exp_out, int Cin, int Cout, int RowsPad, int ColsPad) {
    const unsigned abits = *amax;
    int e = (abits == 0u) ? 0 : 13 - ((int)((abits >> 23) & 255) - 127);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sc = pow2f(e);
    if (blockIdx.x == 0 && threadIdx.x == 0) *exp_out = e;
    const size_t total = (size_t)16 * RowsPad * ColsPad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % ColsPad);
        size_t q = idx / ColsPad;
        const int r = (int)(q % RowsPad);
        const int pt = (int)(q / RowsPad);
        const int ph = pt >> 2, t = pt & 3;
        const int ky = 3 - 2 * (t >> 1) - (ph >> 1), kx = 3 - 2 * (t & 1) - (ph & 1);
        const float v = ((r < Cout && c < Cin) ? wT[(((size_t)c * Cout + r) * 4 + ky) * 4 + kx] : 0.0f) * sc;
        const _Float16 h = (_Float16)v;
        hi[idx] = h;
        lo[idx] = (_Float16)(v - (float)h);
    }
}
}  // namespace

extern "C" int dream_pack_convT4x4_weight_f16x3(const float *wT, void *hi, void *lo, int *exp_out, unsigned *scratch, int Cin,
                                                int Cout, int RowsPad, int ColsPad, void *stream) {
    DREAM_REQUIRE(wT && hi && lo && exp_out && scratch && Cin > 0 && Cout > 0 && RowsPad >= Cout && ColsPad >= Cin,
                  "pack_convT4x4_weight_f16x3: bad arguments");
    if (dream_zero_words(scratch, sizeof(unsigned), (hipStream_t)stream)) return 2;
    const size_t n = (size_t)Cin * Cout * 16;
    size_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, wT, n, scratch);
    DREAM_LAUNCH_OK();
    const size_t total = (size_t)16 * RowsPad * ColsPad;
    g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(pack_wT4_16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, wT, (_Float16 *)hi, (_Float16 *)lo,
                       (const unsigned *)scratch, exp_out, Cin, Cout, RowsPad, ColsPad);
    DREAM_LAUNCH_OK();
    return 0;
}

// ConvTranspose2d(k=4,s=2,p=1) on the split-precision path: same sub-pixel decomposition as
// dream_conv_transpose4x4s2_nhwc_f32 (four 2x2-tap launches); planes from dream_pack_convT4x4_weight_f16x3.
extern "C" int dream_conv_transpose4x4s2_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi,
                                                        const void *w_lo, const int *w_exp, const float *scale,
                                                        const float *shift, float *y, unsigned *amax_out, int B, int H, int W,
                                                        int Cin, int Cout, int CoutPad, int flags, void *stream) {
    DREAM_REQUIRE((flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X | DREAM_CONV_OUT_NCHW | DREAM_CONV_POOL2)) == 0,
                  "convT4x4_f16x3: unsupported flags");
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        Geom16 g;
        g.H = H; g.W = W; g.Hin = H; g.Win = W; g.Hs = H; g.Ws = W; g.Ho = 2 * H; g.Wo = 2 * W;
        g.pad = 1; g.kext = 3; g.ntaps = 4;
        for (int t = 0; t < 4; ++t) { g.tap_dy[t] = (t >> 1) + a; g.tap_dx[t] = (t & 1) + b; }
        g.out_scale = 2; g.out_oy = a; g.out_ox = b;
        const size_t off = (size_t)ph * 4 * CoutPad * Cin;
        if (int rc = launch16(x, amax_in, (const _Float16 *)w_hi + off, (const _Float16 *)w_lo + off, w_exp, scale, shift, nullptr,
                              y, amax_out, B, Cin, Cout, CoutPad, g, flags, stream))
            return rc;
    }
    return 0;
}

// ConvTranspose2d(k=3,s=2,p=1,output_padding 1) on the split-precision path: the sub-pixel decomposition of
// dream_conv_transpose3x3s2_nhwc_f32 (1/2/2/4-tap launches picking their slices of the mode-1 packed planes).
extern "C" int dream_conv_transpose3x3s2_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi,
                                                        const void *w_lo, const int *w_exp, const float *bias, float *y,
                                                        unsigned *amax_out, int B, int H, int W, int Cin, int Cout,
                                                        int CoutPad, int flags, void *stream) {
    DREAM_REQUIRE((flags & ~DREAM_CONV_RELU) == 0, "convT3x3_f16x3: only the ReLU flag is supported");
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        Geom16 g;
        g.H = H; g.W = W; g.Hin = H; g.Win = W; g.Hs = H; g.Ws = W; g.Ho = 2 * H; g.Wo = 2 * W;
        g.pad = 0; g.kext = 2; g.ntaps = 0;
        for (int iy = 0; iy <= a; ++iy)
            for (int ix = 0; ix <= b; ++ix) {
                const int ky = a ? 2 - 2 * iy : 1, kx = b ? 2 - 2 * ix : 1, t = g.ntaps++;
                g.tap_dy[t] = iy; g.tap_dx[t] = ix;
                g.tap_w[t] = 8 - (3 * ky + kx);
            }
        g.out_scale = 2; g.out_oy = a; g.out_ox = b;
        if (int rc = launch16(x, amax_in, w_hi, w_lo, w_exp, nullptr, bias, nullptr, y, amax_out, B, Cin, Cout, CoutPad, g, flags, stream))
            return rc;
    }
    return 0;
}

// k x k (1 | 3) stride-1 conv, same contract as dream_conv2d_nhwc_f32, on the split-precision path.
// amax_in: device scalar with the bit pattern of max|x| (from the producer's amax_out or dream_absmax_f32);
// amax_out: optional, atomicMax of max|y| (caller zeroes it).  Cin % 32 == 0.
extern "C" int dream_conv2d_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi, const void *w_lo,
                                           const int *w_exp, const float *scale, const float *shift, const float *residual,
                                           float *y, unsigned *amax_out, int B, int H, int W, int Cin, int Cout, int CoutPad,
                                           int ksize, int stride, int flags, void *stream) {
    DREAM_REQUIRE(ksize == 1 || ksize == 3, "conv2d_f16x3: kernel size %d not supported", ksize);
    DREAM_REQUIRE(stride == 1, "conv2d_f16x3: stride %d not supported (strided convs stay on the fp32 kernel)", stride);
    const bool ups = (flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X)) != 0;
    DREAM_REQUIRE(!(flags & DREAM_CONV_UPSAMPLE2X) || (H % 2 == 0 && W % 2 == 0), "fused x2 upsample needs even H, W");
    Geom16 g;
    g.H = H; g.W = W; g.Hin = H; g.Win = W; g.Hs = ups ? (H + 1) / 2 : H; g.Ws = ups ? (W + 1) / 2 : W; g.Ho = H; g.Wo = W;
    g.pad = ksize / 2; g.kext = ksize; g.ntaps = ksize * ksize;
    for (int t = 0; t < g.ntaps; ++t) { g.tap_dy[t] = t / ksize; g.tap_dx[t] = t % ksize; }
    g.out_scale = 1; g.out_oy = 0; g.out_ox = 0;
    return launch16(x, amax_in, w_hi, w_lo, w_exp, scale, shift, residual, y, amax_out, B, Cin, Cout, CoutPad, g, flags, stream);
}


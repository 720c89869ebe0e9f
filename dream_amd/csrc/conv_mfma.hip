// Convolutions on NHWC fp32 tensors as an implicit GEMM on the CDNA4 fp32 matrix cores: k x k taps given by a tap table
// (1x1, 3x3 stride 1 / 2, 4x4 stride 2, the 2x2-tap sub-pixel phases of the transposed convs), with bias / folded
// BatchNorm scale-shift, residual add or ReLU mask, ReLU, 2x2 max-pool, NCHW store and an output stride fused into the
// epilogue, and a nearest-x2 upsample or zero-stuffing of the input fused into the patch loader.
//
// Replaces torch.nn.Conv2d(k=3,s=1,p=1) at /root/reference/dream/models.py:594-615 (VGG19 encoder), :695-710 (upsample
// decoder) and :736-747 (belief-map head), nn.ConvTranspose2d at :621-686 and :37-136, and the torchvision ResNet-101
// convs behind :22-32.  The same kernel run on mode-1 packed weights is the data-gradient (conv backward-input)
// operator.  Transposed convs run as sub-pixel phases (no products with zeros); upsample + conv3x3 runs as the
// equivalent 4x4 transposed conv; the DREAM_CONV_UPSAMPLE2X / DREAM_CONV_ZEROSTUFF2X patch-loader forms compute the same
// results with 2.25x / 4x the MACs and are kept as the reference forms (and for a data gradient with a residual).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[k][n]
//   m = output pixel of a TH x TW patch of one image (BM = 32*MR*WM rows per workgroup)
//   n = output channel                                (BN = 32*NR*WN columns per workgroup)
//   k = (tap, cin): 9 taps x Cin, walked as  for cin-chunk(KC) { for tap(9) { KC } }
// One workgroup = 4 wavefronts (WM x WN); each wavefront owns MR x NR accumulators of
// v_mfma_f32_32x32x2_f32 (exact fp32: an fmaf chain, so results are independent of tiling).
//
// LDS: the input patch for the current cin-chunk -- (TH+2) x (TW+2) pixels x KC floats, loaded ONCE
// and re-read by all 9 taps (per-lane ds_read addresses make the tap shift free) -- plus a
// double-buffered [BN][KC] weight tile per tap.  Pixel / cout rows are padded to KC+4 floats so a
// ds_read_b128 lane group lands on distinct 16-B slots.  A lane's ds_read_b128 delivers 4
// consecutive k's; lanes 0-31 take k..k+3 and lanes 32-63 k+4..k+7 of the same row, which feeds 4
// MFMAs (the k order inside the sum is a permutation shared by A and B).
//
// Global traffic: activations are read as 64/128-B pieces of NHWC pixels (whole lines per 4/8
// lanes), weights as 64/128-B rows of the tap-major packed tensor; both are prefetched into
// registers one stage ahead (issued before the MFMA block, written to LDS after it).
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

struct ConvParams {
    const float *x;
    const float *w;          // [ntaps][CoutPad][Cin]
    const float *scale;      // per-channel multiplier (eval-mode BatchNorm fold) or null
    const float *shift;      // per-channel addend (bias / BN shift) or null
    const float *residual;   // tensor of the output's shape added before the ReLU, or null
    float *y;
    unsigned *amax_out;      // optional: atomicMax of the bit pattern of max|y| (y >= 0 bit patterns are ordered)
    int B, H, W;             // grid of output positions per image that the tiles cover
    int Hin, Win;            // logical input extent (after fused upsample / zero-stuffing)
    int Hs, Ws;              // extent of the tensor actually read
    int Ho, Wo;              // extent of the tensor written
    int Cin, Cout, CoutPad;
    int TH, TW, PH, PW;      // pixel tile and staged patch
    int tiles_x, tiles_y;
    int rcpTW;               // ceil(65536 / TW): m / TW == (m * rcpTW) >> 16 for m < 256
    int in_scale;            // input row of patch row 0 = y0 * in_scale - pad_y
    int in_step;             // input pixels between consecutive patch pixels (2 for a strided 1x1)
    int lane_stride;         // patch pixels between consecutive output positions (2 for a strided 3x3)
    int pad_y, pad_x;
    int ntaps;
    unsigned long long tap_w;            // 16 x 4-bit: which [CoutPad][Cin] weight slice each tap of this launch uses
    unsigned long long tap_dy, tap_dx;   // 16 x 4-bit patch offsets of the taps (decoded with scalar ALU ops:
                                         // no kernarg load on the per-stage critical path)
    int out_scale, out_oy, out_ox;   // output pixel = position * out_scale + (out_oy, out_ox)
    int flags;
};

template <int MR, int NR, int WM, int WN, int KC, int NPM>
struct ConvCfg {
    static constexpr int BM = 32 * MR * WM;
    static constexpr int BN = 32 * NR * WN;
    static constexpr int S = KC + 4;                 // padded row stride in floats
    static constexpr int Q = KC / 4;                 // float4 pieces per row
    static constexpr int NP_MAX = NPM;               // patch pixels the tile chooser may use
    static constexpr int NA_IT = (NP_MAX * Q + 255) / 256;
    static constexpr int NB_IT = (BN * Q + 255) / 256;
    static constexpr int NB_FULL = (BN * Q) % 256 == 0;
};

template <int MR, int NR, int WM, int WN, int KC, int NPM>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(const ConvParams p) {
    using C = ConvCfg<MR, NR, WM, WN, KC, NPM>;
    constexpr int S = C::S, Q = C::Q, BN = C::BN;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");

    DREAM_DYNAMIC_LDS(float, smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_index();
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int PW = p.PW, TW = p.TW;
    const int NP = p.PH * PW;
    float *sA = smem;
    float *sB0 = smem + NP * S;
    float *sB1 = sB0 + BN * S;

    // ---- which tile -------------------------------------------------------------------------
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed, speed only) and every XCD has its own L2, so each
    // XCD gets a CONTIGUOUS range of tiles: neighbouring tiles (which share their halo rows / columns) meet in one L2
    // instead of fetching the halo once per XCD.  gridDim.x is padded to a multiple of 8.
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

    // ---- staging plan (fixed for the whole kernel) ---------------------------------------------
    int a_goff[C::NA_IT];     // element offset of this thread's float4 inside image b, or -1 (zero pad)
    int a_soff[C::NA_IT];     // LDS float offset, or -1 when this slot does not exist
#pragma unroll
    for (int it = 0; it < C::NA_IT; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx / Q, q = idx % Q;
        a_soff[it] = (pp < NP) ? pp * S + q * 4 : -1;
        const int py = pp / PW, px = pp - py * PW;
        const int gy = y0 * p.in_scale - p.pad_y + py * p.in_step, gx = x0 * p.in_scale - p.pad_x + px * p.in_step;
        const bool inb = (pp < NP) && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win &&
                         !(zst && ((gy | gx) & 1));      // zero-stuffed input: odd rows/cols are zeros
        const int sy = ups ? (gy >> 1) : gy, sx = ups ? (gx >> 1) : gx;
        a_goff[it] = inb ? (sy * p.Ws + sx) * p.Cin + q * 4 : -1;
    }
    int b_goff[C::NB_IT], b_soff[C::NB_IT];
#pragma unroll
    for (int it = 0; it < C::NB_IT; ++it) {
        const int idx = tid + it * 256;
        const int n = idx / Q, q = idx % Q;
        const bool ok = C::NB_FULL || (n < BN);
        b_soff[it] = ok ? n * S + q * 4 : -1;
        b_goff[it] = (n0 + n) * p.Cin + q * 4;
    }
    const size_t w_tap_stride = (size_t)p.CoutPad * p.Cin;

    // ---- fragment addresses -----------------------------------------------------------------------
    int a_frag[MR], b_frag[NR];
#pragma unroll
    for (int ms = 0; ms < MR; ++ms) {
        int m = (wm * MR + ms) * 32 + li;
        if (m >= p.TH * TW) m = 0;                       // idle rows compute garbage that is never stored
        int ty, tx;
        tile_xy(m, TW, p.rcpTW, pool, &ty, &tx);
        a_frag[ms] = (ty * PW + tx) * p.lane_stride * S + lh * 4;
    }
#pragma unroll
    for (int ns = 0; ns < NR; ++ns) b_frag[ns] = ((wn * NR + ns) * 32 + li) * S + lh * 4;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int ms = 0; ms < MR; ++ms)
#pragma unroll
        for (int ns = 0; ns < NR; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.0f;

    f32x4 a_reg[C::NA_IT], b_reg[C::NB_IT];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    auto load_a = [&](int c0) {
#pragma unroll
        for (int it = 0; it < C::NA_IT; ++it)
            a_reg[it] = (a_goff[it] >= 0) ? *(const f32x4 *)(xb + a_goff[it] + c0) : zero4;
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < C::NA_IT; ++it)
            if (a_soff[it] >= 0) *(f32x4 *)(sA + a_soff[it]) = a_reg[it];
    };
    auto load_b = [&](int tap, int c0) {
        const float *wt = p.w + (size_t)((p.tap_w >> (4 * tap)) & 15) * w_tap_stride + c0;
#pragma unroll
        for (int it = 0; it < C::NB_IT; ++it)
            if (C::NB_FULL || b_soff[it] >= 0) b_reg[it] = *(const f32x4 *)(wt + b_goff[it]);
    };
    auto store_b = [&](float *sB) {
#pragma unroll
        for (int it = 0; it < C::NB_IT; ++it)
            if (C::NB_FULL || b_soff[it] >= 0) *(f32x4 *)(sB + b_soff[it]) = b_reg[it];
    };

    const int nchunks = p.Cin / KC;

    // ---- prologue: stage (chunk 0, tap 0) --------------------------------------------------------
    load_a(0);
    load_b(0, 0);
    store_a();
    store_b(sB0);
    __syncthreads();

    int buf = 0, tap = 0, chunk = 0;
    const int ntaps = p.ntaps, nstages = nchunks * ntaps;
    for (int st = 0; st < nstages; ++st) {
        const bool last_tap = (tap == ntaps - 1);
        const bool more_chunks = (chunk + 1 < nchunks);
        const bool have_next = (st + 1 < nstages);
        // prefetch the next stage's operands into registers (in flight during the MFMAs)
        if (have_next) load_b(last_tap ? 0 : tap + 1, last_tap ? (chunk + 1) * KC : chunk * KC);
        if (last_tap && more_chunks) load_a((chunk + 1) * KC);

        // ---- MFMA block: KC k's of this tap --------------------------------------------------------
        const int tdy = (int)((p.tap_dy >> (4 * tap)) & 15), tdx = (int)((p.tap_dx >> (4 * tap)) & 15);
        const float *sAt = sA + (tdy * PW + tdx) * S;
        const float *sBt = buf ? sB1 : sB0;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 8) {
            f32x4 af[MR], bf[NR];
#pragma unroll
            for (int ms = 0; ms < MR; ++ms) af[ms] = *(const f32x4 *)(sAt + a_frag[ms] + kk);
#pragma unroll
            for (int ns = 0; ns < NR; ++ns) bf[ns] = *(const f32x4 *)(sBt + b_frag[ns] + kk);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ms = 0; ms < MR; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NR; ++ns)
                        acc[ms][ns] = mfma_f32_32x32x2(af[ms][r], bf[ns][r], acc[ms][ns]);
        }

        // ---- publish the next stage -----------------------------------------------------------------
        if (have_next) store_b(buf ? sB0 : sB1);          // other buffer: last read one stage ago
        if (last_tap && more_chunks) {
            __syncthreads();                               // every wave is done with this chunk's patch
            store_a();
        }
        __syncthreads();
        buf ^= 1;
        if (last_tap) { tap = 0; ++chunk; } else ++tap;
    }

    // ---- epilogue: (BN scale), bias/shift, residual, ReLU, store ---------------------------------------
    const bool relu = (p.flags & DREAM_CONV_RELU) != 0;
    const bool nchw = (p.flags & DREAM_CONV_OUT_NCHW) != 0;
    const bool mask = (p.flags & DREAM_CONV_RELUMASK) != 0;
    const bool late = (p.flags & DREAM_CONV_RES_AFTER_RELU) != 0;      // the residual is a skip connection: added after the ReLU
    float scale_v[NR], shift_v[NR];
    int ncol[NR];
#pragma unroll
    for (int ns = 0; ns < NR; ++ns) {
        ncol[ns] = n0 + (wn * NR + ns) * 32 + li;
        const bool cok = ncol[ns] < p.Cout;
        scale_v[ns] = (p.scale != nullptr && cok) ? p.scale[ncol[ns]] : 1.0f;
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
                // oy < Ho / ox < Wo only bites for the sub-pixel phases of an odd-sized transposed conv output
                const bool ok = (m < npix) && (y0 + ty < p.H) && (x0 + tx < p.W) && oy < p.Ho && ox < p.Wo;
#pragma unroll
                for (int ns = 0; ns < NR; ++ns) {
                    if (ok && ncol[ns] < p.Cout) {
                        const size_t o = nchw
                            ? (((size_t)b * p.Cout + ncol[ns]) * p.Ho + oy) * p.Wo + ox
                            : (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + ncol[ns];
                        float v = acc[ms][ns][r];
                        if (p.scale != nullptr) v = v * scale_v[ns];
                        v = v + shift_v[ns];
                        float rv = 0.0f;
                        if (p.residual != nullptr) {
                            rv = p.residual[o];
                            v = mask ? (rv > 0.0f ? v : 0.0f) : (late ? v : v + rv);
                        }
                        if (relu) v = fmaxf(v, 0.0f);
                        if (late) v = v + rv;
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
                            float v = acc[ms][ns][4 * g4 + j];
                            if (p.scale != nullptr) v = v * scale_v[ns];
                            v = v + shift_v[ns];
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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct Variant {
    const char *name;
    int BM, BN, KC, NP_MAX;
    void (*kernel)(const ConvParams);
};

#define DREAM_VARIANT(MR, NR, WM, WN, KC, NPM)                                                   \
    {                                                                                            \
        "m" #MR "n" #NR "w" #WM "x" #WN "k" #KC, 32 * MR * WM, 32 * NR * WN, KC, NPM,           \
            conv_mfma_kernel<MR, NR, WM, WN, KC, NPM>                                            \
    }

const Variant kVariants[] = {
    DREAM_VARIANT(2, 2, 2, 2, 32, 192),   // 0: 128 px x 128 cout
    DREAM_VARIANT(2, 2, 4, 1, 32, 352),   // 1: 256 px x  64 cout
    DREAM_VARIANT(2, 1, 4, 1, 32, 352),   // 2: 256 px x  32 cout
    DREAM_VARIANT(2, 2, 2, 2, 16, 192),   // 3
    DREAM_VARIANT(2, 2, 4, 1, 16, 352),   // 4
    DREAM_VARIANT(2, 1, 4, 1, 16, 352),   // 5
    DREAM_VARIANT(1, 2, 4, 1, 32, 192),   // 6: 128 px x  64 cout
    DREAM_VARIANT(1, 1, 4, 1, 32, 192),   // 7: 128 px x  32 cout
    DREAM_VARIANT(1, 2, 2, 2, 32, 128),   // 8:  64 px x 128 cout (small feature maps, small batch)
    DREAM_VARIANT(1, 1, 2, 2, 32, 128),   // 9:  64 px x  64 cout: one 32x32 tile per wave, shortest K loop per wave
    DREAM_VARIANT(1, 1, 2, 2, 16, 128),   // 10: same, KC 16
    DREAM_VARIANT(2, 2, 2, 2, 16, 608),   // 11: 128 px x 128 cout, big patch (stride-2 3x3 convs)
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr int kNumSelectable = 11;         // variants a caller may force (the big-patch one is picked automatically)
constexpr int kBigPatchVariant = 11;
int g_forced_variant = -1;

// pixel tile: maximise useful rows per workgroup, then minimise the staged patch
void choose_tile(int H, int W, int BM, int np_max, int lane_stride, int kext, bool even, int *th_out, int *tw_out) {
    long best_tiles = -1;
    int best_np = 0, bth = even ? 2 : 1, btw = even ? 2 : 1;
    const int He = even ? (H + 1) / 2 * 2 : H, We = even ? (W + 1) / 2 * 2 : W;
    for (int tw = even ? 2 : 1; tw <= BM && tw <= 127 * (even ? 2 : 1) && tw <= 254; tw += even ? 2 : 1) {
        int th = BM / tw;
        if (even) th &= ~1;
        if (th < 1) break;
        if (th > He) th = He;
        const int twc = tw > We ? We : tw;
        const int np = ((th - 1) * lane_stride + kext) * ((twc - 1) * lane_stride + kext);
        if (np > np_max) continue;
        const long tiles = (long)ceil_div(H, th) * ceil_div(W, twc);
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && np < best_np)) {
            best_tiles = tiles;
            best_np = np;
            bth = th;
            btw = twc;
        }
    }
    *th_out = bth;
    *tw_out = btw;
}

int pick_variant(long pixels, int Cin, int Cout, bool big_patch) {
    if (big_patch) return kBigPatchVariant;
    // Measured on MI355X (profiles/r01_ab_variants_b128.txt, r01_microbench_resnet.txt): the KC=16 variants (half the
    // LDS, 3-4 workgroups per CU) win by 1-8 % whenever the grid fills the chip; when it does not (ResNet trunk at
    // 25x25 / 13x13, small batches) the shorter per-wave K loop and the larger workgroup count of the small tiles win by
    // up to 24 %.  Rule: the largest tile whose grid reaches ~2.3 workgroups per CU, else the tile with the most
    // workgroups.  The KC=32 small tiles need Cin % 32 == 0.
    const bool k32 = (Cin % 32 == 0);
    static const int wide[] = {3, 8, 6, 9, 7}, mid[] = {4, 6, 9, 7}, narrow32[] = {2, 7}, narrow16[] = {5};
    const int *cand = Cout > 64 ? wide : (Cout > 32 ? mid : (k32 ? narrow32 : narrow16));
    const int ncand = Cout > 64 ? 5 : (Cout > 32 ? 4 : (k32 ? 2 : 1));
    int best = cand[0];
    long best_wgs = -1;
    for (int i = 0; i < ncand; ++i) {
        const Variant &v = kVariants[cand[i]];
        if (v.KC == 32 && !k32) continue;
        const long wgs = ((pixels + v.BM - 1) / v.BM) * (long)ceil_div(Cout, v.BN);
        if (wgs >= 600) return cand[i];
        if (wgs > best_wgs) { best_wgs = wgs; best = cand[i]; }
    }
    return best;
}

struct ConvGeom {
    int H, W;                 // grid of output positions
    int Hin, Win, Hs, Ws;     // logical input extent, stored input extent
    int Ho, Wo;               // output tensor extent
    int in_scale, in_step, lane_stride, pad;
    int ntaps;
    int tap_dy[16], tap_dx[16];
    int tap_w[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};   // weight slice of each tap
    int kext;                 // patch rows needed beyond (TH-1)*lane_stride
    int out_scale, out_oy, out_ox;
};

int launch_conv(const float *x, const float *w, const float *scale, const float *shift, const float *residual,
                float *y, int B, int Cin, int Cout, int CoutPad, const ConvGeom &g, int flags, void *stream,
                unsigned *amax_out = nullptr) {
    DREAM_REQUIRE(x && w && y, "null pointer");
    DREAM_REQUIRE(B > 0 && g.H > 0 && g.W > 0 && Cin > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, g.H, g.W, Cin, Cout);
    DREAM_REQUIRE(Cin % 16 == 0, "Cin=%d must be a multiple of 16 (pad the channels)", Cin);
    DREAM_REQUIRE((size_t)g.Hs * g.Ws * (size_t)Cin < ((size_t)1 << 31) && (size_t)g.Ho * g.Wo * (size_t)Cout < ((size_t)1 << 31),
                  "image too large for 32-bit offsets");
    const bool big_patch = g.lane_stride > 1;
    int v = (g_forced_variant >= 0 && !big_patch) ? g_forced_variant : pick_variant((long)B * g.H * g.W, Cin, Cout, big_patch);
    if (Cin % kVariants[v].KC != 0) v = (kVariants[v].BN >= 128) ? 3 : (kVariants[v].BN == 64 ? 4 : 5);
    const Variant &var = kVariants[v];
    DREAM_REQUIRE(CoutPad % var.BN == 0 && CoutPad >= Cout, "CoutPad=%d must be a multiple of %d (variant %s)", CoutPad, var.BN, var.name);

    ConvParams p;
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y; p.amax_out = amax_out;
    p.B = B; p.H = g.H; p.W = g.W; p.Hin = g.Hin; p.Win = g.Win; p.Hs = g.Hs; p.Ws = g.Ws; p.Ho = g.Ho; p.Wo = g.Wo;
    p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    const bool pool = (flags & DREAM_CONV_POOL2) != 0;
    choose_tile(g.H, g.W, var.BM, var.NP_MAX, g.lane_stride, g.kext, pool, &p.TH, &p.TW);
    p.PH = (p.TH - 1) * g.lane_stride + g.kext;
    p.PW = (p.TW - 1) * g.lane_stride + g.kext;
    DREAM_REQUIRE(p.PH * p.PW <= var.NP_MAX, "internal: patch %dx%d exceeds variant %s", p.PH, p.PW, var.name);
    p.tiles_x = ceil_div(g.W, p.TW);
    p.tiles_y = ceil_div(g.H, p.TH);
    p.rcpTW = pool ? (65536 + p.TW / 2 - 1) / (p.TW / 2) : (65536 + p.TW - 1) / p.TW;
    if (pool) { p.Ho = g.H / 2; p.Wo = g.W / 2; }
    p.in_scale = g.in_scale; p.in_step = g.in_step; p.lane_stride = g.lane_stride; p.pad_y = g.pad; p.pad_x = g.pad;
    p.ntaps = g.ntaps;
    const int S = var.KC + 4;
    p.tap_dy = 0; p.tap_dx = 0; p.tap_w = 0;
    for (int t = 0; t < g.ntaps; ++t) {
        p.tap_dy |= (unsigned long long)g.tap_dy[t] << (4 * t);
        p.tap_dx |= (unsigned long long)g.tap_dx[t] << (4 * t);
        p.tap_w |= (unsigned long long)g.tap_w[t] << (4 * t);
    }
    p.out_scale = g.out_scale; p.out_oy = g.out_oy; p.out_ox = g.out_ox;
    p.flags = flags;

    const size_t lds = ((size_t)p.PH * p.PW + 2 * (size_t)var.BN) * S * sizeof(float);
    DREAM_REQUIRE(lds <= 160 * 1024, "LDS request %zu too large", lds);
    if (dream_allow_full_lds((const void *)var.kernel)) return 2;
    const dim3 grid((unsigned)(ceil_div((int)((size_t)B * p.tiles_x * p.tiles_y), 8) * 8), (unsigned)ceil_div(Cout, var.BN));
    hipLaunchKernelGGL(var.kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" int dream_conv3x3_set_variant(int variant) {
    DREAM_REQUIRE(variant >= -1 && variant < kNumSelectable, "variant %d out of range", variant);
    g_forced_variant = variant;
    return 0;
}
extern "C" int dream_conv3x3_num_variants(void) { return kNumSelectable; }
extern "C" const char *dream_conv3x3_variant_name(int variant) {
    return (variant >= 0 && variant < kNumVariants) ? kVariants[variant].name : "heuristic";
}
extern "C" size_t dream_conv3x3_cout_pad(int Cout) {
    // a multiple of every BN in the variant table, so any variant can run any layer
    return (size_t)ceil_div(Cout, 128) * 128;
}

// k x k convolution (k in {1,3}), stride in {1,2}, pad = k/2; H, W are the INPUT extent (after the fused
// x2 upsample / zero-stuffing when those flags are set).
static int conv2d_impl(const float *x, const float *w_packed, const float *scale, const float *shift,
                       const float *residual, float *y, int B, int H, int W, int Cin, int Cout,
                       int CoutPad, int ksize, int stride, int flags, void *stream, unsigned *amax_out) {
    DREAM_REQUIRE(ksize == 1 || ksize == 3, "conv2d: kernel size %d not supported (1 or 3)", ksize);
    DREAM_REQUIRE(stride == 1 || stride == 2, "conv2d: stride %d not supported (1 or 2)", stride);
    const bool ups = (flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X)) != 0;
    // zero-stuffing also accepts an odd extent 2*Hs-1 (data gradient of a stride-2 conv with an odd input)
    DREAM_REQUIRE(!(flags & DREAM_CONV_UPSAMPLE2X) || (H % 2 == 0 && W % 2 == 0), "fused x2 upsample needs even H, W (got %dx%d)", H, W);
    DREAM_REQUIRE(!ups || stride == 1, "fused upsample with a strided conv is not supported");
    DREAM_REQUIRE(!(flags & DREAM_CONV_RES_AFTER_RELU) || (residual != nullptr && !(flags & (DREAM_CONV_POOL2 | DREAM_CONV_RELUMASK))),
                  "conv2d: residual-after-ReLU needs a residual and excludes the fused pool / the ReLU mask");
    DREAM_REQUIRE(!(flags & DREAM_CONV_POOL2) || (stride == 1 && !(flags & DREAM_CONV_OUT_NCHW) && residual == nullptr && H >= 2 && W >= 2),
                  "fused max-pool: stride 1, NHWC output, no residual");
    ConvGeom g;
    const int pad = ksize / 2;
    g.Hin = H; g.Win = W;
    g.Hs = ups ? (H + 1) / 2 : H; g.Ws = ups ? (W + 1) / 2 : W;
    g.Ho = (H + 2 * pad - ksize) / stride + 1;
    g.Wo = (W + 2 * pad - ksize) / stride + 1;
    g.H = g.Ho; g.W = g.Wo;
    g.in_scale = stride;
    g.in_step = (ksize == 1) ? stride : 1;          // a strided 1x1 only stages the pixels it uses
    g.lane_stride = (ksize == 1) ? 1 : stride;
    g.pad = pad;
    g.ntaps = ksize * ksize;
    for (int t = 0; t < g.ntaps; ++t) { g.tap_dy[t] = t / ksize; g.tap_dx[t] = t % ksize; }
    g.kext = ksize;
    g.out_scale = 1; g.out_oy = 0; g.out_ox = 0;
    return launch_conv(x, w_packed, scale, shift, residual, y, B, Cin, Cout, CoutPad, g, flags, stream, amax_out);
}
extern "C" int dream_conv2d_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                                     const float *residual, float *y, int B, int H, int W, int Cin, int Cout,
                                     int CoutPad, int ksize, int stride, int flags, void *stream) {
    return conv2d_impl(x, w_packed, scale, shift, residual, y, B, H, W, Cin, Cout, CoutPad, ksize, stride, flags, stream, nullptr);
}
// same, and additionally atomicMax(max|y|) into *amax_out (caller zeroes it): the dynamic range the split-precision
// kernel needs for the NEXT layer's input scaling
extern "C" int dream_conv2d_amax_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                                          const float *residual, float *y, unsigned *amax_out, int B, int H, int W, int Cin,
                                          int Cout, int CoutPad, int ksize, int stride, int flags, void *stream) {
    return conv2d_impl(x, w_packed, scale, shift, residual, y, B, H, W, Cin, Cout, CoutPad, ksize, stride, flags, stream, amax_out);
}

extern "C" int dream_conv3x3_nhwc_f32(const float *x, const float *w_packed, const float *bias, float *y, int B,
                                      int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream) {
    return dream_conv2d_nhwc_f32(x, w_packed, nullptr, bias, nullptr, y, B, H, W, Cin, Cout, CoutPad, 3, 1, flags, stream);
}

// ConvTranspose2d(k=4, s=2, p=1) by sub-pixel decomposition: output pixel (2m+a, 2n+b) is a 2x2 convolution of
// the input around (m, n) with the taps ky = 3 - 2*ty - a, kx = 3 - 2*tx - b (ty, tx in {0,1}); four launches,
// one per phase (a, b), each writing its quarter of the [B,2H,2W,Cout] output.  No multiplications by zero.
// w_packed: [4 phases][4 taps][CoutPad][Cin] from dream_pack_convT4x4_weight.
extern "C" int dream_conv_transpose4x4s2_nhwc_f32(const float *x, const float *w_packed, const float *scale,
                                                  const float *shift, float *y, int B, int H, int W, int Cin,
                                                  int Cout, int CoutPad, int flags, void *stream) {
    DREAM_REQUIRE((flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X | DREAM_CONV_OUT_NCHW)) == 0, "convT4x4: unsupported flags");
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        ConvGeom g;
        g.H = H; g.W = W; g.Hin = H; g.Win = W; g.Hs = H; g.Ws = W; g.Ho = 2 * H; g.Wo = 2 * W;
        g.in_scale = 1; g.in_step = 1; g.lane_stride = 1; g.pad = 1;
        g.ntaps = 4;
        for (int t = 0; t < 4; ++t) { g.tap_dy[t] = (t >> 1) + a; g.tap_dx[t] = (t & 1) + b; }
        g.kext = 3;
        g.out_scale = 2; g.out_oy = a; g.out_ox = b;
        const float *wp = w_packed + (size_t)ph * 4 * CoutPad * Cin;
        if (int rc = launch_conv(x, wp, scale, shift, nullptr, y, B, Cin, Cout, CoutPad, g, flags, stream)) return rc;
    }
    return 0;
}

// ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (dream/models.py:621-686) by sub-pixel decomposition: output row
// 2m + a comes from ky = 1 (input row m) for a = 0 and from ky = 2 (row m) and ky = 0 (row m + 1) for a = 1, so the four
// output phases are stride-1 convolutions with 1 / 2 / 2 / 4 taps over a (TH+1) x (TW+1) patch -- 9 MACs per four outputs
// where the zero-stuffed form (DREAM_CONV_ZEROSTUFF2X) spends 36.  w_packed is the SAME mode-1 packing the zero-stuffed
// form uses ([9][CoutPad][Cin], slice t = transposed weights of tap 8 - (3 ky + kx)): the launches pick their slices.
namespace {
int conv_transpose3x3s2_impl(const float *x, const float *w_packed, const float *bias, float *y, int B, int H, int W,
                             int Ho, int Wo, int Cin, int Cout, int CoutPad, int flags, void *stream, const float *residual = nullptr) {
    DREAM_REQUIRE((flags & ~(DREAM_CONV_RELU | DREAM_CONV_RES_AFTER_RELU)) == 0, "convT3x3: only the ReLU / residual-after-ReLU flags are supported");
    DREAM_REQUIRE(!(flags & DREAM_CONV_RES_AFTER_RELU) || residual != nullptr, "convT3x3: residual-after-ReLU without a residual");
    DREAM_REQUIRE((Ho == 2 * H || Ho == 2 * H - 1) && (Wo == 2 * W || Wo == 2 * W - 1), "convT3x3: output %dx%d for input %dx%d", Ho, Wo, H, W);
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        ConvGeom g;
        g.H = H; g.W = W; g.Hin = H; g.Win = W; g.Hs = H; g.Ws = W; g.Ho = Ho; g.Wo = Wo;
        g.in_scale = 1; g.in_step = 1; g.lane_stride = 1; g.pad = 0;
        g.ntaps = 0;
        for (int iy = 0; iy <= a; ++iy)                      // patch offset iy: ky = 1 (a = 0); ky = 2, 0 (a = 1)
            for (int ix = 0; ix <= b; ++ix) {
                const int ky = a ? 2 - 2 * iy : 1, kx = b ? 2 - 2 * ix : 1, t = g.ntaps++;
                g.tap_dy[t] = iy; g.tap_dx[t] = ix;
                g.tap_w[t] = 8 - (3 * ky + kx);
            }
        g.kext = 2;
        g.out_scale = 2; g.out_oy = a; g.out_ox = b;
        if (int rc = launch_conv(x, w_packed, nullptr, bias, residual, y, B, Cin, Cout, CoutPad, g, flags, stream)) return rc;
    }
    return 0;
}
}  // namespace

// the same with a tensor of the OUTPUT's shape [B,2H,2W,Cout] added in the epilogue (before the ReLU, or after it with
// DREAM_CONV_RES_AFTER_RELU: the skip connection behind deconv_0_1, dream/models.py:796-799)
extern "C" int dream_conv_transpose3x3s2_res_nhwc_f32(const float *x, const float *w_packed, const float *bias, const float *residual,
                                                      float *y, int B, int H, int W, int Cin, int Cout, int CoutPad, int flags,
                                                      void *stream) {
    DREAM_REQUIRE(residual != nullptr, "convT3x3 (residual): null residual");
    return conv_transpose3x3s2_impl(x, w_packed, bias, y, B, H, W, 2 * H, 2 * W, Cin, Cout, CoutPad, flags, stream, residual);
}

extern "C" int dream_conv_transpose3x3s2_nhwc_f32(const float *x, const float *w_packed, const float *bias, float *y, int B,
                                                  int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream) {
    return conv_transpose3x3s2_impl(x, w_packed, bias, y, B, H, W, 2 * H, 2 * W, Cin, Cout, CoutPad, flags, stream);
}

// Data gradient of a k x k (1 | 3) STRIDE-2 pad-k/2 convolution (the three strided 3x3 convs and three strided 1x1
// downsample convs of ResNet-101, torchvision Bottleneck "v1.5" behind dream/models.py:22-32): the transposed conv of dy
// [B,Hy,Wy,C] with the mode-1 packed forward weights, output dx [B,Hx,Wx,Cx] with Hx in {2Hy-1, 2Hy}.  k = 3: the
// sub-pixel phases above (no products with stuffed zeros); k = 1: dx is zero except at even positions, which are one
// 1x1 convolution of dy written with an output stride of 2.
extern "C" int dream_conv2d_s2_bwd_data_nhwc_f32(const float *dy, const float *w_packed_mode1, float *dx, int B, int Hy, int Wy,
                                                 int C, int Hx, int Wx, int Cx, int RowsPad, int ksize, void *stream) {
    DREAM_REQUIRE(dy && w_packed_mode1 && dx && (ksize == 1 || ksize == 3), "conv2d_s2_bwd_data: bad arguments");
    if (ksize == 3) return conv_transpose3x3s2_impl(dy, w_packed_mode1, nullptr, dx, B, Hy, Wy, Hx, Wx, C, Cx, RowsPad, 0, stream);
    DREAM_REQUIRE((Hx == 2 * Hy || Hx == 2 * Hy - 1) && (Wx == 2 * Wy || Wx == 2 * Wy - 1), "conv2d_s2_bwd_data: output %dx%d for input %dx%d", Hx, Wx, Hy, Wy);
    if (dream_zero_words(dx, (size_t)B * Hx * Wx * Cx * sizeof(float), (hipStream_t)stream)) return 2;     // (a kernel: common.h)
    ConvGeom g;
    g.H = Hy; g.W = Wy; g.Hin = Hy; g.Win = Wy; g.Hs = Hy; g.Ws = Wy; g.Ho = Hx; g.Wo = Wx;
    g.in_scale = 1; g.in_step = 1; g.lane_stride = 1; g.pad = 0;
    g.ntaps = 1; g.tap_dy[0] = 0; g.tap_dx[0] = 0;
    g.kext = 1;
    g.out_scale = 2; g.out_oy = 0; g.out_ox = 0;
    return launch_conv(dy, w_packed_mode1, nullptr, nullptr, nullptr, dx, B, C, Cx, RowsPad, g, 0, stream);
}

// Data gradient of ConvTranspose2d(k4,s2,p1): dx[m] = sum_k dy[2m - 1 + k] * wT[.][.][k] -- a 4x4 stride-2 pad-1
// convolution of dy [B,2H,2W,Cout_T] producing [B,H,W,Cin_T].  w_packed: [16][RowsPad >= Cin_T][Cout_T] from
// dream_pack_conv_weight(wT viewed as OIHW with O = Cin_T, I = Cout_T, ntaps 16, mode 0).
extern "C" int dream_conv4x4s2_nhwc_f32(const float *x, const float *w_packed, const float *residual, float *y, int B,
                                        int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream) {
    DREAM_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv4x4s2: even input extent expected (got %dx%d)", H, W);
    ConvGeom g;
    g.Hin = H; g.Win = W; g.Hs = H; g.Ws = W;
    g.Ho = H / 2; g.Wo = W / 2; g.H = g.Ho; g.W = g.Wo;
    g.in_scale = 2; g.in_step = 1; g.lane_stride = 2; g.pad = 1;
    g.ntaps = 16;
    for (int t = 0; t < 16; ++t) { g.tap_dy[t] = t / 4; g.tap_dx[t] = t % 4; }
    g.kext = 4;
    g.out_scale = 1; g.out_oy = 0; g.out_ox = 0;
    return launch_conv(x, w_packed, nullptr, nullptr, residual, y, B, Cin, Cout, CoutPad, g, flags, stream);
}

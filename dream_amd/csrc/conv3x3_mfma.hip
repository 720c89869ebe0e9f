// conv 3x3 / stride 1 / pad 1 (+bias, +ReLU, optional fused nearest-x2 upsample of the input,
// optional NCHW store) on NHWC fp32 tensors as an implicit GEMM on the CDNA4 fp32 matrix cores.
//
// Replaces torch.nn.Conv2d(k=3,s=1,p=1) at /root/reference/dream/models.py:594-615 (VGG19 encoder),
// :695-710 (upsample decoder, with the nn.Upsample at :691,:703 fused into the patch load) and
// :736-747 (belief-map head).  The same kernel run on mode-1 packed weights is the data-gradient
// (conv backward-input) operator, and with DREAM_CONV_ZEROSTUFF2X it is ConvTranspose2d(k=3,s=2,p=1,
// output_padding=1) (dream/models.py:621-686): a transposed conv == a stride-1 conv of the
// zero-stuffed input with the flipped kernel; the stuffing is done in the patch loader.
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

struct Conv3x3Params {
    const float *x;
    const float *w;
    const float *bias;
    float *y;
    int B, H, W;      // output == logical input extent
    int Hs, Ws;       // extent of the tensor actually read (H/2, W/2 when the x2 upsample is fused)
    int Cin, Cout, CoutPad;
    int TH, TW, PW;   // pixel tile and patch width (TW + 2)
    int tiles_x, tiles_y;
    int rcpTW;        // ceil(65536 / TW): m / TW == (m * rcpTW) >> 16 for m < 256
    int flags;
};

template <int MR, int NR, int WM, int WN, int KC>
struct ConvCfg {
    static constexpr int BM = 32 * MR * WM;
    static constexpr int BN = 32 * NR * WN;
    static constexpr int S = KC + 4;                 // padded row stride in floats
    static constexpr int Q = KC / 4;                 // float4 pieces per row
    static constexpr int NP_MAX = (BM == 64) ? 128 : (BM == 128) ? 192 : 352;   // patch pixels the tile chooser may use
    static constexpr int NA_IT = (NP_MAX * Q + 255) / 256;
    static constexpr int NB_IT = (BN * Q + 255) / 256;
    static constexpr int NB_FULL = (BN * Q) % 256 == 0;
};

template <int MR, int NR, int WM, int WN, int KC>
__global__ void __launch_bounds__(256, 2) conv3x3_mfma_kernel(const Conv3x3Params p) {
    using C = ConvCfg<MR, NR, WM, WN, KC>;
    constexpr int S = C::S, Q = C::Q, BN = C::BN;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");

    DREAM_DYNAMIC_LDS(float, smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_index();
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int PW = p.PW, TW = p.TW;
    const int NP = (p.TH + 2) * PW;
    float *sA = smem;
    float *sB0 = smem + NP * S;
    float *sB1 = sB0 + BN * S;

    // ---- which tile -------------------------------------------------------------------------
    int t = blockIdx.x;
    const int tix = t % p.tiles_x;
    t /= p.tiles_x;
    const int tiy = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int y0 = tiy * p.TH, x0 = tix * TW;
    const int n0 = blockIdx.y * BN;
    const bool zst = (p.flags & DREAM_CONV_ZEROSTUFF2X) != 0;
    const bool ups = (p.flags & DREAM_CONV_UPSAMPLE2X) != 0 || zst;
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
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inb = (pp < NP) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W &&
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
        const int ty = (m * p.rcpTW) >> 16, tx = m - ty * TW;
        a_frag[ms] = (ty * PW + tx) * S + lh * 4;
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
        const float *wt = p.w + (size_t)tap * w_tap_stride + c0;
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

    int buf = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_chunks = (chunk + 1 < nchunks);
        for (int tap = 0; tap < 9; ++tap) {
            const bool last_tap = (tap == 8);
            const bool have_next = !(last_tap && !more_chunks);
            // prefetch the next stage's operands into registers (in flight during the MFMAs)
            if (have_next) load_b(last_tap ? 0 : tap + 1, last_tap ? (chunk + 1) * KC : chunk * KC);
            if (last_tap && more_chunks) load_a((chunk + 1) * KC);

            // ---- MFMA block: KC k's of this tap ----------------------------------------------------
            const int ky = tap / 3, kx = tap - ky * 3;
            const float *sAt = sA + (ky * PW + kx) * S;
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

            // ---- publish the next stage -------------------------------------------------------------
            if (have_next) store_b(buf ? sB0 : sB1);      // other buffer: last read one stage ago
            if (last_tap && more_chunks) {
                __syncthreads();                           // every wave is done with this chunk's patch
                store_a();
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue: bias, ReLU, store -------------------------------------------------------------
    const bool relu = (p.flags & DREAM_CONV_RELU) != 0;
    const bool nchw = (p.flags & DREAM_CONV_OUT_NCHW) != 0;
    float bias_v[NR];
    int ncol[NR];
#pragma unroll
    for (int ns = 0; ns < NR; ++ns) {
        ncol[ns] = n0 + (wn * NR + ns) * 32 + li;
        bias_v[ns] = (p.bias != nullptr && ncol[ns] < p.Cout) ? p.bias[ncol[ns]] : 0.0f;
    }
    const int npix = p.TH * TW;
#pragma unroll
    for (int ms = 0; ms < MR; ++ms) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (wm * MR + ms) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ty = (m * p.rcpTW) >> 16, tx = m - ty * TW;
            const int oy = y0 + ty, ox = x0 + tx;
            const bool ok = (m < npix) && (oy < p.H) && (ox < p.W);
#pragma unroll
            for (int ns = 0; ns < NR; ++ns) {
                float v = acc[ms][ns][r] + bias_v[ns];
                if (relu) v = fmaxf(v, 0.0f);
                if (ok && ncol[ns] < p.Cout) {
                    const size_t o = nchw
                        ? (((size_t)b * p.Cout + ncol[ns]) * p.H + oy) * p.W + ox
                        : (((size_t)b * p.H + oy) * p.W + ox) * p.Cout + ncol[ns];
                    p.y[o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct Variant {
    const char *name;
    int BM, BN, KC, NP_MAX;
    void (*kernel)(const Conv3x3Params);
};

#define DREAM_VARIANT(MR, NR, WM, WN, KC)                                                        \
    {                                                                                            \
        "m" #MR "n" #NR "w" #WM "x" #WN "k" #KC, ConvCfg<MR, NR, WM, WN, KC>::BM,              \
            ConvCfg<MR, NR, WM, WN, KC>::BN, KC, ConvCfg<MR, NR, WM, WN, KC>::NP_MAX,            \
            conv3x3_mfma_kernel<MR, NR, WM, WN, KC>                                              \
    }

const Variant kVariants[] = {
    DREAM_VARIANT(2, 2, 2, 2, 32),   // 0: 128 px x 128 cout
    DREAM_VARIANT(2, 2, 4, 1, 32),   // 1: 256 px x  64 cout
    DREAM_VARIANT(2, 1, 4, 1, 32),   // 2: 256 px x  32 cout
    DREAM_VARIANT(2, 2, 2, 2, 16),   // 3
    DREAM_VARIANT(2, 2, 4, 1, 16),   // 4
    DREAM_VARIANT(2, 1, 4, 1, 16),   // 5
    DREAM_VARIANT(1, 2, 4, 1, 32),   // 6: 128 px x  64 cout
    DREAM_VARIANT(1, 1, 4, 1, 32),   // 7: 128 px x  32 cout
    DREAM_VARIANT(1, 2, 2, 2, 32),   // 8:  64 px x 128 cout (small feature maps, small batch)
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
int g_forced_variant = -1;
bool g_attr_set[kNumVariants] = {};

// pixel tile: maximise useful rows per workgroup, then minimise the halo
void choose_tile(int H, int W, int BM, int np_max, int *th_out, int *tw_out) {
    long best_tiles = -1;
    int best_np = 0, bth = 1, btw = 1;
    for (int tw = 1; tw <= BM && tw <= 255; ++tw) {
        int th = BM / tw;
        if (th < 1) break;
        if (th > H) th = H;
        const int twc = tw > W ? W : tw;
        const int np = (th + 2) * (twc + 2);
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

int pick_variant(int B, int H, int W, int Cin, int Cout) {
    const bool k32 = (Cin % 32 == 0);
    const long pixels = (long)B * H * W;
    if (Cout > 64) {
        // enough 128x128 tiles to fill 256 CUs x 2 workgroups?  otherwise the 64-px variant
        const long tiles128 = ((pixels + 127) / 128) * (long)ceil_div(Cout, 128);
        if (k32 && tiles128 < 512) return 8;
        return k32 ? 0 : 3;
    }
    if (Cout > 32) {
        const long tiles256 = (pixels + 255) / 256;
        if (k32 && tiles256 < 512) return 6;
        return k32 ? 1 : 4;
    }
    {
        const long tiles256 = (pixels + 255) / 256;
        if (k32 && tiles256 < 512) return 7;
        return k32 ? 2 : 5;
    }
}

}  // namespace

extern "C" int dream_conv3x3_set_variant(int variant) {
    DREAM_REQUIRE(variant >= -1 && variant < kNumVariants, "variant %d out of range", variant);
    g_forced_variant = variant;
    return 0;
}
extern "C" int dream_conv3x3_num_variants(void) { return kNumVariants; }
extern "C" const char *dream_conv3x3_variant_name(int variant) {
    return (variant >= 0 && variant < kNumVariants) ? kVariants[variant].name : "heuristic";
}
extern "C" size_t dream_conv3x3_cout_pad(int Cout) {
    // a multiple of every BN in the variant table, so any variant can run any layer
    return (size_t)ceil_div(Cout, 128) * 128;
}

extern "C" int dream_conv3x3_nhwc_f32(const float *x, const float *w_packed, const float *bias,
                                      float *y, int B, int H, int W, int Cin, int Cout,
                                      int CoutPad, int flags, void *stream) {
    DREAM_REQUIRE(x && w_packed && y, "null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    DREAM_REQUIRE(Cin % 16 == 0, "Cin=%d must be a multiple of 16 (pad the channels)", Cin);
    const bool ups = (flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X)) != 0;
    DREAM_REQUIRE(!ups || (H % 2 == 0 && W % 2 == 0), "fused x2 upsample / zero-stuffing needs even H, W (got %dx%d)", H, W);
    DREAM_REQUIRE((size_t)H * W * (size_t)(Cin > Cout ? Cin : Cout) < ((size_t)1 << 31), "image too large for 32-bit offsets");

    int v = g_forced_variant >= 0 ? g_forced_variant : pick_variant(B, H, W, Cin, Cout);
    if (Cin % kVariants[v].KC != 0) v = (kVariants[v].BN >= 128) ? 3 : (kVariants[v].BN == 64 ? 4 : 5);
    const Variant &var = kVariants[v];
    DREAM_REQUIRE(CoutPad % var.BN == 0 && CoutPad >= Cout, "CoutPad=%d must be a multiple of %d (variant %s)", CoutPad, var.BN, var.name);

    Conv3x3Params p;
    p.x = x; p.w = w_packed; p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W;
    p.Hs = ups ? H / 2 : H; p.Ws = ups ? W / 2 : W;
    p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    choose_tile(H, W, var.BM, var.NP_MAX, &p.TH, &p.TW);
    p.PW = p.TW + 2;
    p.tiles_x = ceil_div(W, p.TW);
    p.tiles_y = ceil_div(H, p.TH);
    p.rcpTW = (65536 + p.TW - 1) / p.TW;
    p.flags = flags;

    const int S = var.KC + 4;
    const size_t lds = ((size_t)(p.TH + 2) * p.PW + 2 * (size_t)var.BN) * S * sizeof(float);
    DREAM_REQUIRE(lds <= 160 * 1024, "LDS request %zu too large", lds);
    if (!g_attr_set[v]) {
        DREAM_HIP_OK(hipFuncSetAttribute((const void *)var.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_attr_set[v] = true;
    }
    const dim3 grid((unsigned)((size_t)B * p.tiles_x * p.tiles_y), (unsigned)ceil_div(Cout, var.BN));
    hipLaunchKernelGGL(var.kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

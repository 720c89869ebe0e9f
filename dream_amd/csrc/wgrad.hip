// Convolution weight + bias gradients (replace ATen's conv backward-weight reached from loss.backward(),
// /root/reference/dream/network.py:335, for the Conv2d / ConvTranspose2d layers of dream/models.py:22-136,594-747).
//
//   dW[t][r][c] = sum over (b, position m) of  T[b,m][r] * P[b, m + d(t)][c]
//
// T ("tile tensor") is sampled at the positions themselves and gives the ROWS of dW, P ("patch tensor") is sampled
// through the taps and gives the COLUMNS:  Conv2d: T = dy, P = x;  ConvTranspose2d: T = x, P = dy.
// GEMM view per tap: dW_t (R x C) = T^T (R x M) * P_t (M x C): the reduction runs over POSITIONS (up to 20.5 M at
// 128x400x400), so the work is split over position tiles ("split-K"); partials go to a workspace and are summed in a
// FIXED order by a second kernel, so the result is deterministic (no fp32 atomics).
//
// One kernel template, three register blockings of the wave's 32x32 fp32 MFMA tiles (<= 9 accumulators = 144 VGPRs):
//   <1,1,9,128>  3x3 convs: all NINE tap accumulators of a 64x64 (rows x cols) workgroup tile; T tile and P patch are
//                staged in LDS once per 128-position tile and feed 9 MFMAs per k-step (k = 2 positions);
//   <2,1,4,64>   stride-2 transposed convs, one launch per output PHASE (a,b): the dy pixels of one phase form a
//                stride-1 grid, so each phase is a 2x2-tap (k=4) or 1/2/2/4-tap (k=3) stride-1 problem with a compact
//                patch -- no zero-stuffing and no 4x oversized stride-2 patch; 128x64 tile, 2 row blocks x 4 taps;
//   <2,2,1,64>   1x1 convs: 128x128 tile (2x2 blocks per wave) -- a plain split-K GEMM that re-reads its operands half
//                as often as 64x64 tiles would.
// LDS operand reads are ds_read_b32 of 32 consecutive floats per half-wave (conflict-free for any row stride).
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

constexpr int WG_MAXSLOT = 9;
constexpr int WG_NPMAX = 192;     // patch pixels for the 128-position variants
constexpr int WG_NPMAX_STRIDED = 352;   // stride-2 3x3 taps need ~4x the tile; 1 workgroup per CU then

struct WgradParams {
    const float *tile_t;     // tensor whose channels become the ROWS of dW, sampled at the tile positions
    const float *patch_t;    // tensor whose channels become the COLUMNS of dW, sampled through the taps
    float *part;             // [splitk][ntaps_total][RowsPad][Cp]
    float *bias_part;        // [splitk][RowsPad] (column sums of tile_t) or null
    int B, Ht, Wt;           // tile tensor extent == grid of positions
    int Hin, Win, Hs, Ws;    // logical / stored extent of the patch tensor
    int Ct, Cp, RowsPad;     // channels of tile / patch tensor
    int TH, TW, PH, PW, tiles_x, tiles_y, rcpTW;
    int in_scale, in_step, lane_stride, pad_y, pad_x;
    int ntaps_total, ntaps;                    // ntaps = taps of THIS launch (<= NT)
    unsigned long long tap_dy, tap_dx, tap_out;   // per launch slot: 4-bit patch offsets and output tap index
    int tiles_total, splitk, flags;
    int nrb, ncb;            // row / column blocks of the workgroup grid
};

template <int RB, int CB, int NT, int PIX>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradParams p) {
    constexpr int RW = 64 * RB, CW = 64 * CB;       // workgroup tile: RW rows x CW cols (waves 2 x 2)
    constexpr int QY = RW / 4, QX = CW / 4;         // float4 per staged row
    constexpr int RPY = 256 / QY, RPX = 256 / QX;   // rows staged per pass
    static_assert(RB * CB * NT <= WG_MAXSLOT, "too many accumulators");
    DREAM_DYNAMIC_LDS(float, smem);
    float *sY = smem;                      // [PIX][RW]
    float *sX = smem + PIX * RW;           // [NP][CW]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int wo = wave >> 1, wi = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed, speed only).  All (row block, col block) tiles of one
    // split-K slice re-read the same T tiles and P patches, so they are given consecutive slots on ONE XCD and meet
    // in its L2 instead of each fetching from the fabric: b = (slice_group * nblocks + block) * 8 + xcd.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nblk = p.nrb * p.ncb;
    const int blk = slot % nblk, ks = (slot / nblk) * 8 + xcd;
    if (ks >= p.splitk) return;
    const int rbk = blk % p.nrb, cbk = blk / p.nrb;
    const int co0 = rbk * RW, ci0 = cbk * CW;
    const int PW = p.PW, TW = p.TW, npix = p.TH * p.TW, NP = p.PH * PW;
    const bool zst = (p.flags & DREAM_CONV_ZEROSTUFF2X) != 0;
    const bool ups = (p.flags & DREAM_CONV_UPSAMPLE2X) != 0 || zst;

    f32x16 acc[RB * CB * NT];
#pragma unroll
    for (int t = 0; t < RB * CB * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f32x4 bsum = {0.0f, 0.0f, 0.0f, 0.0f};          // this thread's share of the column sums of tile_t

    int toff[NT];                                   // LDS offsets of this launch's taps (wave-uniform)
#pragma unroll
    for (int t = 0; t < NT; ++t)
        toff[t] = ((int)((p.tap_dy >> (4 * t)) & 15) * PW + (int)((p.tap_dx >> (4 * t)) & 15)) * CW;

    const int qy = tid % QY, prow_y = tid / QY;     // staging of the tile tensor
    const int qx = tid % QX, prow_x = tid / QX;     // staging of the patch
    const bool y_chan_ok = (co0 + qy * 4) < p.Ct;   // channel counts are multiples of 4 (host wrapper)
    const bool x_chan_ok = (ci0 + qx * 4) < p.Cp;
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int tile = ks; tile < p.tiles_total; tile += p.splitk) {
        int t = tile;
        const int tix = t % p.tiles_x;
        t /= p.tiles_x;
        const int tiy = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int y0 = tiy * p.TH, x0 = tix * TW;
        const float *xb = p.patch_t + (size_t)b * p.Hs * p.Ws * p.Cp;
        const float *dyb = p.tile_t + (size_t)b * p.Ht * p.Wt * p.Ct;

        __syncthreads();                            // previous tile fully consumed
        // Staging issues ALL of a batch's global loads before the first LDS write (loads from clamped, always-legal
        // addresses + a select, no exec-masked branches): a load -> wait -> ds_write chain per row made the 20 rows a
        // thread stages 20 dependent memory round trips (~35k cycles per tile, as long as the tile's MFMA work).
        // ---- stage the tile tensor (rows m >= npix or outside the image are zero) ------------------
        {
            f32x4 tr[PIX / RPY];
            unsigned okm = 0;
#pragma unroll
            for (int it = 0; it < PIX / RPY; ++it) {
                const int m = prow_y + it * RPY;
                const int ty = (m * p.rcpTW) >> 16, tx = m - ty * TW;
                const int oy = y0 + ty, ox = x0 + tx;
                const bool ok = m < npix && oy < p.Ht && ox < p.Wt && y_chan_ok;
                okm |= (ok ? 1u : 0u) << it;
                tr[it] = *(const f32x4 *)(ok ? dyb + ((size_t)oy * p.Wt + ox) * p.Ct + co0 + qy * 4 : p.tile_t);
            }
#pragma unroll
            for (int it = 0; it < PIX / RPY; ++it) {
                const f32x4 v = ((okm >> it) & 1u) ? tr[it] : zero4;
                *(f32x4 *)(sY + (prow_y + it * RPY) * RW + qy * 4) = v;
                bsum += v;
            }
        }
        // ---- stage the patch, XB rows per thread in flight --------------------------------------------------
        constexpr int XB = 6;
        for (int pp0 = prow_x; pp0 < NP; pp0 += RPX * XB) {
            f32x4 xr[XB];
            unsigned okm = 0;
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int pp = pp0 + j * RPX;
                const int py = pp / PW, px = pp - py * PW;
                const int gy = y0 * p.in_scale - p.pad_y + py * p.in_step, gx = x0 * p.in_scale - p.pad_x + px * p.in_step;
                const bool ok = pp < NP && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win && x_chan_ok && !(zst && ((gy | gx) & 1));
                const int sy = ups ? (gy >> 1) : gy, sx = ups ? (gx >> 1) : gx;
                okm |= (ok ? 1u : 0u) << j;
                xr[j] = *(const f32x4 *)(ok ? xb + ((size_t)sy * p.Ws + sx) * p.Cp + ci0 + qx * 4 : p.patch_t);
            }
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int pp = pp0 + j * RPX;
                if (pp < NP) *(f32x4 *)(sX + pp * CW + qx * 4) = ((okm >> j) & 1u) ? xr[j] : zero4;
            }
        }
        __syncthreads();

        // ---- k-steps (2 positions each) x (row block, col block, tap) ----------------------------------
        // Operands of step s+1 are read from LDS while the MFMAs of step s run (two register sets, loop unrolled by
        // two): a ds_read -> s_waitcnt -> MFMA chain per tap leaves the matrix pipe idle for most of the LDS latency.
        const float *aY = sY + wo * (32 * RB) + li;
        const float *bX = sX + wi * (32 * CB) + li;
        const int nsteps = (npix + 1) >> 1;
        float a0[RB], a1[RB], b0[CB * NT], b1[CB * NT];
        auto load_step = [&](int st, float (&a)[RB], float (&b)[CB * NT]) {
            const int m = 2 * st + lh;
            const int mc = m < npix ? m : 0;         // tile row m is zero there; keep the patch address legal
            const int ty = (mc * p.rcpTW) >> 16, tx = mc - ty * TW;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = aY[m * RW + rb * 32];
            const float *bp = bX + (ty * PW + tx) * p.lane_stride * CW;
#pragma unroll
            for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) b[tp * CB + cb] = bp[toff[tp] + cb * 32];
        };
        auto mma_step = [&](const float (&a)[RB], const float (&b)[CB * NT]) {
#pragma unroll
            for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        acc[(rb * CB + cb) * NT + tp] = mfma_f32_32x32x2(a[rb], b[tp * CB + cb], acc[(rb * CB + cb) * NT + tp]);
        };
        // sched_barrier: hipcc's scheduler (at ~250 VGPRs it optimises for register pressure) otherwise sinks each
        // block of ds_reads down to its own MFMAs, which puts the LDS latency back in front of every k-step
        // Requested instruction order inside one (loads of step s+1, MFMAs of step s) region: an MFMA first, the address
        // arithmetic and one ds_read in the shadow of each MFMA.  Without it the two waves that share a SIMD (one per
        // resident workgroup, barrier-aligned) do their address / LDS phase at the same time and the matrix pipe idles.
        auto interleave = [&]() {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // MFMA
            __builtin_amdgcn_sched_group_barrier(0x006, 16, 0);           // VALU | SALU: addresses of the next step
#pragma unroll
            for (int i = 0; i < RB + CB * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // one ds_read
                if (i + 1 < RB * CB * NT) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, RB * CB * NT, 0); // remaining MFMAs
            __builtin_amdgcn_sched_barrier(0);
        };
        load_step(0, a0, b0);
        int st = 0;
        for (; st + 1 < nsteps; st += 2) {
            __builtin_amdgcn_sched_barrier(0);
            load_step(st + 1, a1, b1);
            mma_step(a0, b0);
            interleave();
            load_step(st + 2 < nsteps ? st + 2 : nsteps - 1, a0, b0);
            mma_step(a1, b1);
            interleave();
        }
        if (st < nsteps) mma_step(a0, b0);
    }

    // ---- write partials ----------------------------------------------------------------------------
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
        {
            const int tout = (int)((p.tap_out >> (4 * tp)) & 15);
            float *part = p.part + ((size_t)ks * p.ntaps_total + tout) * p.RowsPad * p.Cp;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = co0 + wo * (32 * RB) + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int i = ci0 + wi * (32 * CB) + cb * 32 + li;
                        if (o < p.RowsPad && i < p.Cp) part[(size_t)o * p.Cp + i] = acc[(rb * CB + cb) * NT + tp][r];
                    }
        }
    }
    if (cbk == 0 && p.bias_part != nullptr) {
        // threads sharing qy (same 4 channels) differ in prow_y: reduce the RPY rows through LDS
        __syncthreads();
        *(f32x4 *)(smem + (prow_y * QY + qy) * 4) = bsum;
        __syncthreads();
        if (tid < RW) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < RPY; ++r) s += smem[(r * QY + (tid >> 2)) * 4 + (tid & 3)];
            if (co0 + tid < p.RowsPad) p.bias_part[(size_t)ks * p.RowsPad + co0 + tid] = s;
        }
    }
}

// fixed-order reduction over the split-K partials
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *part, float *out, size_t n, int splitk) {
    const size_t n4 = n / 4;                        // n is a multiple of 4 (channel counts are)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 *src = (const f32x4 *)part + i;
        int k = 0;
        for (; k + 8 <= splitk; k += 8) {           // eight independent loads in flight, summed in index order
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * n4];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k + 4 <= splitk; k += 4) {
            const f32x4 v0 = src[(size_t)k * n4], v1 = src[(size_t)(k + 1) * n4];
            const f32x4 v2 = src[(size_t)(k + 2) * n4], v3 = src[(size_t)(k + 3) * n4];
            s += v0;
            s += v1;
            s += v2;
            s += v3;
        }
        for (; k < splitk; ++k) s += src[(size_t)k * n4];
        ((f32x4 *)out)[i] = s;
    }
}

struct WgradVariant {
    void (*kernel)(WgradParams);
    int RW, CW, NT, PIX;
};
// NT is the EXACT tap count of a launch (no runtime tap test inside the k-loop)
WgradVariant g_wvariants[] = {
    {wgrad_kernel<1, 1, 9, 128>, 64, 64, 9, 128},     // 0: 3x3 convs
    {wgrad_kernel<2, 1, 4, 64>, 128, 64, 4, 64},      // 1: transposed-conv phase with 4 taps
    {wgrad_kernel<2, 2, 1, 64>, 128, 128, 1, 64},     // 2: 1x1 convs / 1-tap phase, >= 128 x 128
    {wgrad_kernel<1, 1, 4, 128>, 64, 64, 4, 128},     // 3: 4-tap phase, < 128 rows
    {wgrad_kernel<2, 1, 2, 64>, 128, 64, 2, 64},      // 4: 2-tap phase (k = 3 transposed conv)
    {wgrad_kernel<1, 1, 2, 128>, 64, 64, 2, 128},     // 5: 2-tap phase, < 128 rows
    {wgrad_kernel<1, 1, 1, 128>, 64, 64, 1, 128},     // 6: 1 tap, small
};
int g_wgrad_forced = -1;          // A/B switch (dream_wgrad_set_variant)


// One launch group = the taps that share a patch origin (all taps of a conv; one output phase of a transposed conv).
struct TapGroup {
    int pad_y, pad_x, ntaps;
    int dy[WG_MAXSLOT], dx[WG_MAXSLOT], out[WG_MAXSLOT];
};
struct WgradGeom {
    int Ht, Wt, Hin, Win, Hs, Ws;
    int in_scale, in_step, lane_stride, kext;
    int ntaps;                    // total taps of dW
    int ngroups;
    TapGroup group[4];
};

// Blocking for a launch of `ntaps` taps (exact): wide tiles when both dW dimensions fill them.
int pick_wvariant(int ntaps, int Cp, int RowsPad, int forced) {
    const bool rows128 = RowsPad >= 128 && forced != 0;
    switch (ntaps) {
    case 9: return 0;
    case 4: return rows128 ? 1 : 3;
    case 2: return rows128 ? 4 : 5;
    case 1: return (rows128 && Cp >= 128) ? 2 : 6;
    default: return -1;
    }
}

int main_group(const WgradGeom &g) {
    int best = 0;
    for (int i = 1; i < g.ngroups; ++i)
        if (g.group[i].ntaps > g.group[best].ntaps) best = i;
    return best;
}

void choose_tile_w(const WgradGeom &g, int pix, int *th_out, int *tw_out) {
    const int np_max = g.lane_stride > 1 ? WG_NPMAX_STRIDED : (pix >= 128 ? WG_NPMAX : 2 * pix);
    long best = -1;
    int bnp = 0, bth = 1, btw = 1;
    for (int tw = 1; tw <= pix; ++tw) {
        for (int th = pix / tw; th >= 1; th = (g.lane_stride > 1 ? th - 1 : 0)) {
            int thc = th > g.Ht ? g.Ht : th;
            const int twc = tw > g.Wt ? g.Wt : tw;
            const int np = ((thc - 1) * g.lane_stride + g.kext) * ((twc - 1) * g.lane_stride + g.kext);
            if (np > np_max) continue;
            const long tiles = (long)ceil_div(g.Ht, thc) * ceil_div(g.Wt, twc);
            if (best < 0 || tiles < best || (tiles == best && np < bnp)) { best = tiles; bnp = np; bth = thc; btw = twc; }
            break;
        }
    }
    *th_out = bth;
    *tw_out = btw;
}

int pick_splitk_g(int B, const WgradGeom &g, int Cp, int RowsPad, int forced) {
    const int v = pick_wvariant(g.group[main_group(g)].ntaps, Cp, RowsPad, forced);
    if (v < 0) return 1;
    const WgradVariant &var = g_wvariants[v];
    int th, tw;
    choose_tile_w(g, var.PIX, &th, &tw);
    const long tiles = (long)B * ceil_div(g.Ht, th) * ceil_div(g.Wt, tw);
    const long ctiles = (long)ceil_div(RowsPad, var.RW) * ceil_div(Cp, var.CW);
    // 512 workgroups = 256 CUs x 2 resident: every extra split adds a RowsPad x Cp partial to write and re-read, which
    // for the 1x1 layers of the ResNet trunk outweighs the operands themselves (profiles/r01_ab_wgrad.txt)
    long sk = wgrad_target_workgroups(512) / ctiles;
    if (sk < 1) sk = 1;
    if (sk > tiles) sk = tiles;
    if (sk > 1024) sk = 1024;
    return (int)sk;
}

size_t wgrad_workspace_bytes(int B, const WgradGeom &g, int Cp, int RowsPad) {
    // the A/B switch may change the split: size for the larger of the two (the switch is passed down as an argument --
    // replicas of the single-process data-parallel path call in here concurrently, nothing global may be touched)
    int sk = 0;
    for (int f = -1; f <= 0; ++f) {
        const int s = pick_splitk_g(B, g, Cp, RowsPad, f);
        sk = s > sk ? s : sk;
    }
    return ((size_t)sk * g.ntaps * RowsPad * Cp + (size_t)(sk + 1) * RowsPad) * sizeof(float);
}

int launch_wgrad(const float *tile_t, const float *patch_t, float *dw_packed, float *dbias, int nbias, void *workspace,
                 int B, int Ct, int RowsPad, int Cp, const WgradGeom &g, int flags, void *stream) {
    DREAM_REQUIRE(tile_t && patch_t && dw_packed && workspace, "wgrad: null pointer");
    DREAM_REQUIRE(B > 0 && Ct % 4 == 0 && Cp % 4 == 0 && RowsPad >= Ct && RowsPad % 4 == 0, "wgrad: bad channels (%d, %d, pad %d)", Ct, Cp, RowsPad);
    WgradParams p;
    p.tile_t = tile_t; p.patch_t = patch_t;
    p.B = B; p.Ht = g.Ht; p.Wt = g.Wt; p.Hin = g.Hin; p.Win = g.Win; p.Hs = g.Hs; p.Ws = g.Ws;
    p.Ct = Ct; p.Cp = Cp; p.RowsPad = RowsPad; p.flags = flags;
    p.in_scale = g.in_scale; p.in_step = g.in_step; p.lane_stride = g.lane_stride;
    p.ntaps_total = g.ntaps;
    const int forced = g_wgrad_forced;                  // read once: the A/B hook may flip it while a launch is being set up
    p.splitk = pick_splitk_g(B, g, Cp, RowsPad, forced);   // one split for all groups: they share the partial buffer
    p.part = (float *)workspace;
    float *bias_part = p.part + (size_t)p.splitk * g.ntaps * RowsPad * Cp;
    for (int gi = 0; gi < g.ngroups; ++gi) {
        const TapGroup &tg = g.group[gi];
        const int v = pick_wvariant(tg.ntaps, Cp, RowsPad, forced);
        DREAM_REQUIRE(v >= 0, "wgrad: no blocking for a %d-tap launch", tg.ntaps);
        WgradVariant &var = g_wvariants[v];
        choose_tile_w(g, var.PIX, &p.TH, &p.TW);
        p.PH = (p.TH - 1) * g.lane_stride + g.kext;
        p.PW = (p.TW - 1) * g.lane_stride + g.kext;
        p.tiles_x = ceil_div(g.Wt, p.TW); p.tiles_y = ceil_div(g.Ht, p.TH);
        p.rcpTW = (65536 + p.TW - 1) / p.TW;
        p.tiles_total = B * p.tiles_x * p.tiles_y;
        const size_t lds = ((size_t)var.PIX * var.RW + (size_t)p.PH * p.PW * var.CW) * sizeof(float);
        DREAM_REQUIRE(lds <= 160 * 1024, "wgrad: LDS request %zu too large", lds);
        if (dream_allow_full_lds((const void *)var.kernel)) return 2;
        p.nrb = ceil_div(RowsPad, var.RW); p.ncb = ceil_div(Cp, var.CW);
        const dim3 grid((unsigned)(p.nrb * p.ncb * ceil_div(p.splitk, 8) * 8));
        p.pad_y = tg.pad_y; p.pad_x = tg.pad_x; p.ntaps = tg.ntaps;
        p.tap_dy = 0; p.tap_dx = 0; p.tap_out = 0;
        for (int t = 0; t < tg.ntaps; ++t) {
            p.tap_dy |= (unsigned long long)tg.dy[t] << (4 * t);
            p.tap_dx |= (unsigned long long)tg.dx[t] << (4 * t);
            p.tap_out |= (unsigned long long)tg.out[t] << (4 * t);
        }
        p.bias_part = (dbias != nullptr && gi == 0) ? bias_part : nullptr;
        hipLaunchKernelGGL(var.kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
        DREAM_LAUNCH_OK();
    }
    const size_t n = (size_t)g.ntaps * RowsPad * Cp;
    size_t gr = (n / 4 + 255) / 256;
    if (gr > 2048) gr = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gr), dim3(256), 0, (hipStream_t)stream,
                       (const float *)p.part, dw_packed, n, p.splitk);
    DREAM_LAUNCH_OK();
    if (dbias) {
        // bias partials are [splitk][RowsPad]; only the first nbias entries are wanted
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                           (const float *)bias_part, bias_part + (size_t)p.splitk * RowsPad, (size_t)RowsPad, p.splitk);
        DREAM_LAUNCH_OK();
        if (dream_copy_words(dbias, bias_part + (size_t)p.splitk * RowsPad, (size_t)nbias * sizeof(float), (hipStream_t)stream)) return 2;
    }
    return 0;
}

WgradGeom conv_geom(int H, int W, int ksize, int stride, int flags) {
    const bool ups = (flags & (DREAM_CONV_UPSAMPLE2X | DREAM_CONV_ZEROSTUFF2X)) != 0;
    WgradGeom g;
    const int pad = ksize / 2;
    g.Hin = H; g.Win = W;
    g.Hs = ups ? (H + 1) / 2 : H; g.Ws = ups ? (W + 1) / 2 : W;
    g.Ht = (H + 2 * pad - ksize) / stride + 1;
    g.Wt = (W + 2 * pad - ksize) / stride + 1;
    g.in_scale = stride;
    g.in_step = (ksize == 1) ? stride : 1;
    g.lane_stride = (ksize == 1) ? 1 : stride;
    g.kext = ksize;
    g.ntaps = ksize * ksize;
    g.ngroups = 1;
    TapGroup &tg = g.group[0];
    tg.pad_y = pad; tg.pad_x = pad; tg.ntaps = g.ntaps;
    for (int t = 0; t < g.ntaps; ++t) { tg.dy[t] = t / ksize; tg.dx[t] = t % ksize; tg.out[t] = t; }
    return g;
}

WgradGeom convT_geom(int H, int W, int k) {
    // ConvTranspose2d(k, stride 2, pad 1, output 2H x 2W; k = 4, or k = 3 with output_padding 1):
    //   dW_T[i][o][ky][kx] = sum_m x[m][i] * dy[2m - 1 + (ky,kx)][o]          tile = x (H x W), patch = dy
    // Output row 2j + a is reached through ky of parity (a+1)&1 only, from input row j + (a + 1 - ky)/2.  So per
    // phase (a, b) the dy pixels form the stride-1 grid dyP[j][i] = dy[2j+a][2i+b] and
    //   dW_T[.][.][ky][kx] = sum_m x[m] * dyP[m - (a + 1 - ky)/2, ...]:
    // a = 0: ky = 1 -> offset 0, ky = 3 -> offset +1 (patch origin m);  a = 1: ky = 0 -> -1, ky = 2 -> 0 (origin m-1).
    // In the kernel's patch addressing g = m0*in_scale - pad + p*in_step with in_scale = in_step = 2, pad = a (0 | 1).
    WgradGeom g;
    g.Ht = H; g.Wt = W; g.Hin = 2 * H; g.Win = 2 * W; g.Hs = 2 * H; g.Ws = 2 * W;
    g.in_scale = 2; g.in_step = 2; g.lane_stride = 1; g.kext = 2;
    g.ntaps = k * k;
    g.ngroups = 4;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            TapGroup &tg = g.group[a * 2 + b];
            tg.pad_y = a; tg.pad_x = b; tg.ntaps = 0;
            for (int ky = (a + 1) & 1; ky < k; ky += 2)
                for (int kx = (b + 1) & 1; kx < k; kx += 2) {
                    const int t = tg.ntaps++;
                    tg.dy[t] = (a == 0) ? (ky - 1) / 2 : ky / 2;      // patch row of this tap relative to the origin
                    tg.dx[t] = (b == 0) ? (kx - 1) / 2 : kx / 2;
                    tg.out[t] = ky * k + kx;
                }
        }
    return g;
}

// ---- first layer (NCHW image, Cin <= 4): lane == cout, 9*Cin accumulators per lane ---------------------
constexpr int FT = 16, FPW = FT + 4, FPH = FT + 2, FMAXC = 4;

struct FirstWgradParams {
    const float *x;
    const float *dy;
    float *part;          // [nblocks*4 waves][Cout][9*Cin + 1]  (last column = bias)
    int B, H, W, Cin, Cout, tiles_x, tiles_y, tiles_total;
};

__global__ void __launch_bounds__(256) first_wgrad_kernel(const FirstWgradParams p) {
    DREAM_DYNAMIC_LDS(float, smem);      // [Cin][FPH][FPW]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_index();
    const int cout = blockIdx.y * 64 + lane;
    float acc[FMAXC * 9];
#pragma unroll
    for (int i = 0; i < FMAXC * 9; ++i) acc[i] = 0.0f;
    float bacc = 0.0f;
    const int npatch = p.Cin * FPH * FPW;
    for (int tile = blockIdx.x; tile < p.tiles_total; tile += gridDim.x) {
        int t = tile;
        const int tix = t % p.tiles_x;
        t /= p.tiles_x;
        const int tiy = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int y0 = tiy * FT, x0 = tix * FT;
        __syncthreads();
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
        __syncthreads();
        for (int g = 0; g < 16; ++g) {
            const int row = wave * 4 + (g >> 2), xg = (g & 3) * 4;
            const int oy = y0 + row, ox = x0 + xg;
            float d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[j] = (oy < p.H && ox + j < p.W) ? p.dy[(((size_t)b * p.H + oy) * p.W + ox + j) * p.Cout + cout] : 0.0f;
            bacc += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
            for (int c = 0; c < FMAXC; ++c) {
                if (c < p.Cin) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float *src = smem + (c * FPH + row + ky) * FPW + xg;
                        const f32x4 v0 = *(const f32x4 *)src;
                        const float v[6] = {v0[0], v0[1], v0[2], v0[3], src[4], src[5]};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[c * 9 + ky * 3 + kx] = fmaf(d[j], v[j + kx], acc[c * 9 + ky * 3 + kx]);
                    }
                }
            }
        }
    }
    const int ncol = 9 * p.Cin + 1;
    float *dst = p.part + (((size_t)blockIdx.x * 4 + wave) * p.Cout + cout) * ncol;
#pragma unroll
    for (int i = 0; i < FMAXC * 9; ++i)
        if (i < 9 * p.Cin) dst[i] = acc[i];
    dst[9 * p.Cin] = bacc;
}

// 4096 partial rows of 64 x 28 sums: one thread per output summed them one dependent load at a time (1.3 ms per vgg_q training
// step for a 7-KB result).  Now 64 outputs x 16 row slices per workgroup, eight loads in flight per thread, fp64 accumulation,
// the slices combined through LDS in slice order (a fixed order: deterministic).
__global__ void __launch_bounds__(1024) first_wgrad_reduce_kernel(const float *part, float *dw, float *dbias,
                                                                  int nparts, int Cout, int ncol) {
    __shared__ double red[16][64];
    const int total = Cout * ncol;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < total) {
        const float *src = part + i;
        int k = slice;
        for (; k + 7 * 16 < nparts; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + 16 * j) * total];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; k < nparts; k += 16) s += (double)src[(size_t)k * total];
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && i < total) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][lane];
        const int o = i / ncol, c = i - o * ncol;
        if (c == ncol - 1) dbias[o] = (float)t;
        else dw[(size_t)o * (ncol - 1) + c] = (float)t;
    }
}

}  // namespace

extern "C" size_t dream_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int CoutPad, int ksize, int stride) {
    return wgrad_workspace_bytes(B, conv_geom(H, W, ksize, stride, 0), Cin, CoutPad);
}
// k x k conv (k = 1 | 3, stride 1 | 2, pad k/2) weight + bias gradient.  x [B,H,W,Cin] (half-res with the fused
// upsample / zero-stuff flags), dy [B,Ho,Wo,Cout] -> dw_packed [k*k][CoutPad][Cin] (mode-0 layout), dbias [Cout] or null
extern "C" int dream_conv2d_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, float *dbias, void *workspace,
                                           int B, int H, int W, int Cin, int Cout, int CoutPad, int ksize, int stride,
                                           int flags, void *stream) {
    DREAM_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_wgrad: unsupported k=%d s=%d", ksize, stride);
    return launch_wgrad(dy, x, dw_packed, dbias, Cout, workspace, B, Cout, CoutPad, Cin, conv_geom(H, W, ksize, stride, flags),
                        flags, stream);
}
extern "C" size_t dream_conv3x3_wgrad_workspace(int B, int H, int W, int Cin, int CoutPad) {
    return dream_conv2d_wgrad_workspace(B, H, W, Cin, CoutPad, 3, 1);
}
extern "C" int dream_conv3x3_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, float *dbias,
                                            void *workspace, int B, int H, int W, int Cin, int Cout,
                                            int CoutPad, int flags, void *stream) {
    return dream_conv2d_wgrad_nhwc_f32(x, dy, dw_packed, dbias, workspace, B, H, W, Cin, Cout, CoutPad, 3, 1, flags, stream);
}
extern "C" size_t dream_convT4x4_wgrad_workspace(int B, int H, int W, int CinPad, int Cout) {
    return wgrad_workspace_bytes(B, convT_geom(H, W, 4), Cout, CinPad);
}
extern "C" size_t dream_convT_wgrad_workspace(int B, int H, int W, int CinPad, int Cout, int ksize) {
    return wgrad_workspace_bytes(B, convT_geom(H, W, ksize), Cout, CinPad);
}
// ConvTranspose2d(k = 3 (output_padding 1) | 4, stride 2, pad 1) weight gradient -> dw_packed [k*k][CinPad][Cout]
extern "C" int dream_convT_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, void *workspace, int B,
                                          int H, int W, int Cin, int CinPad, int Cout, int ksize, void *stream) {
    DREAM_REQUIRE(ksize == 3 || ksize == 4, "convT_wgrad: kernel size %d not supported", ksize);
    return launch_wgrad(x, dy, dw_packed, nullptr, 0, workspace, B, Cin, CinPad, Cout, convT_geom(H, W, ksize), 0, stream);
}
// ConvTranspose2d(k4,s2,p1) weight gradient: x [B,H,W,Cin], dy [B,2H,2W,Cout] -> dw_packed [16][CinPad][Cout]
// (tap t = ky*4+kx; unpack with dream_unpack_conv_weight(rows = Cin, cols = Cout, ntaps = 16) gives [Cin,Cout,4,4])
extern "C" int dream_convT4x4_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, void *workspace, int B,
                                             int H, int W, int Cin, int CinPad, int Cout, void *stream) {
    return launch_wgrad(x, dy, dw_packed, nullptr, 0, workspace, B, Cin, CinPad, Cout, convT_geom(H, W, 4), 0, stream);
}

// A/B switch: -1 = heuristic, 0 = always the 64-row blockings
extern "C" int dream_wgrad_set_variant(int variant) {
    DREAM_REQUIRE(variant >= -1 && variant <= 0, "wgrad variant %d out of range", variant);
    g_wgrad_forced = variant;
    return 0;
}

static int first_wgrad_blocks(int B, int H, int W) {
    const long tiles = (long)B * ceil_div(W, FT) * ceil_div(H, FT);
    return (int)(tiles < 1024 ? tiles : 1024);
}
extern "C" size_t dream_conv3x3_first_wgrad_workspace(int B, int H, int W, int Cin, int Cout) {
    return (size_t)first_wgrad_blocks(B, H, W) * 4 * Cout * (9 * Cin + 1) * sizeof(float);
}

extern "C" int dream_conv3x3_first_wgrad_f32(const float *x_nchw, const float *dy_nhwc, float *dw_oihw, float *dbias,
                                             void *workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                                             int Cout, void *stream) {
    DREAM_REQUIRE(x_nchw && dy_nhwc && dw_oihw && dbias && workspace, "first_wgrad: null pointer");
    DREAM_REQUIRE(Cin >= 1 && Cin <= FMAXC && Cout % 64 == 0, "first_wgrad: Cin <= %d and Cout %% 64 == 0 required", FMAXC);
    FirstWgradParams p;
    p.x = x_nchw; p.dy = dy_nhwc; p.part = (float *)workspace;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_x = ceil_div(W, FT); p.tiles_y = ceil_div(H, FT);
    p.tiles_total = B * p.tiles_x * p.tiles_y;
    const int nblocks = first_wgrad_blocks(B, H, W);
    const int ncol = 9 * Cin + 1;
    const size_t need = (size_t)nblocks * 4 * Cout * ncol * sizeof(float);
    DREAM_REQUIRE(workspace_bytes >= need, "first_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    const size_t lds = (size_t)Cin * FPH * FPW * sizeof(float);
    hipLaunchKernelGGL(first_wgrad_kernel, dim3(nblocks, Cout / 64), dim3(256), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3(ceil_div(Cout * ncol, 64)), dim3(1024), 0, (hipStream_t)stream,
                       (const float *)p.part, dw_oihw, dbias, nblocks * 4, Cout, ncol);
    DREAM_LAUNCH_OK();
    return 0;
}

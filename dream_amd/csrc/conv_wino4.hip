// 3x3 stride-1 pad-1 convolution on NHWC fp32 tensors by the Winograd minimal-filtering algorithm F(4x4, 3x3) on the CDNA4 fp32
// matrix cores: 36 multiplications per 4x4 output tile and input channel -- 2.25 per output where F(2x2,3x3) (conv_wino.hip)
// needs 4 and the direct algorithm 9.  Every product and sum is IEEE fp32; the weight transform runs once, in fp64, at pack time.
//
// Replaces torch.nn.Conv2d(k=3,s=1,p=1) (+ReLU, + the following MaxPool2d(2)) at /root/reference/dream/models.py:598-615 (VGG19
// encoder), :695-710 (decoder convs not preceded by an upsample) for every layer behind the 3-channel first one -- 88 % of a
// DREAM-vgg-Q forward pass -- and, on mode-1 packed weights, their data gradients; nn.ConvTranspose2d(k4,s2,p1) of the ResNet decoder
// (models.py:37-136) through the 25-position phase patterns (PAT below).  The description that follows is the WIDE workgroup shape
// (more than 64 output channels); W4Cfg has the narrow one.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, Y: 4x4 outputs
//
// Interpolation points (0, 1, -1, 1/2, -2, inf), NOT the usual (0, +-1, +-2, inf): measured against an fp64 direct convolution on
// the vgg_q layer shapes (profiles/r03_f4_accuracy.txt) the error relative to the output maximum is 1.0e-6 .. 2.7e-6 (the usual
// points: 2.5e-6 .. 1.2e-5; F(2x2,3x3): 2e-7 .. 5e-7; the direct fp32 kernel: 5e-7 .. 1.4e-6) -- mixing a small and a large
// point keeps the entries of B^T, G and A^T within [1/8, 8] and the transformed operands well scaled.  With these points
//
//   B^T = [ 1 -1.5 -2    1.5  1    0 ]     A^T = [ 1  1  1  1     1  0 ]     G = [  1      0     0    ]
//         [ 0 -1    0.5  2.5  1    0 ]           [ 0  1 -1  1/2  -2  0 ]         [  1/3    1/3   1/3  ]
//         [ 0  1   -2.5  0.5  1    0 ]           [ 0  1  1  1/4   4  0 ]         [ -1/3    1/3  -1/3  ]
//         [ 0 -2   -1    2    1    0 ]           [ 0  1 -1  1/8  -8  1 ]         [ -16/15 -8/15 -4/15 ]
//         [ 0  0.5 -1   -0.5  1    0 ]                                           [  1/15  -2/15  4/15 ]
//         [ 0  1   -1.5 -2    1.5  1 ]                                           [  0      0     1    ]
//
// GEMM view: for each of the 36 positions p of the transformed 6x6 domain,  M_p[tile][cout] = sum_cin V_p[tile][cin] U_p[cin][cout].
//   * tiles are numbered over (image, tile row, tile column); a workgroup takes 16 consecutive ones (256 output pixels) and
//     128 output channels; wavefront = 16 tiles x 16 channels x all 36 positions: 36 accumulators of v_mfma_f32_16x16x4_f32
//     (144 VGPRs), so the inverse transform A^T M A is lane-local;
//   * V = B^T d B is computed by the workgroup for one 16-channel chunk at a time, in two passes through LDS, WHILE the MFMAs of
//     the previous chunk run: pass 1 -- a thread owns (tile, channel quad, patch ROW): six b128 loads, the transform along the
//     row in registers, six b128 stores to a staging tile; pass 2 -- a thread owns (tile, channel quad, transformed COLUMN):
//     six b128 reads down the column, the transform along it, six b128 stores into the V buffer the next chunk's MFMAs read
//     (B^T has 4-5 non-zeros per row: the cross-lane exchange F(2x2)'s kernel does with one DPP move would take four here).
//     384 items per pass on 512 threads: 48 per wavefront and pass, the same work on every wavefront between two barriers;
//   * U (the transformed weights, packed [Cin/16][36][CoutPad][16]) never touches LDS: every wavefront streams its own 16-channel
//     operand rows straight from L2 into a ring of eight registers, 1 KB coalesced per position, six positions ahead;
//   * all global traffic through buffer descriptors (32-bit offsets, zero padding and store masking by the bounds check);
//   * positions are multiplied in PAIRS with alternating accumulators (a dependent v_mfma_f32_16x16x4_f32 has 40 cycles of
//     latency against 32 of issue), one memory instruction or transform piece behind each pair of MFMAs (pinned order);
//   * persistent workgroups walking over the tile blocks of their XCD's range, the next block's first chunk transformed during
//     the current block's last chunk -- as conv_wino.hip;
//   * epilogue: the lane's four tiles go through A^T M A in register pairs (packed fp32), parameters re-read through the constant
//     address space (s_load), offsets = tile base + scalar when all 16 tiles of the wavefront are interior.
#include <type_traits>
#include <stdlib.h>
#include <dream_cdna4.h>
#include "common.h"
#include "pack_device.h"
#include "../../include/dream_hip.h"

// Timing diagnostics only (tools/wino4_diag.py builds separate libraries with -DDREAM_W4_DIAG=k; never the product library; results
// are then wrong by construction): bit 0 no patch loads, bit 1 no weight stream, bit 2 no barriers, bit 3 no pass 1 / pass 2, bit 4 patch loads out of range, bit 5 every weight load reads position 0 of
// chunk 0 (L1 hits: the instruction stream without its L2 traffic), bit 6 every chunk's patch loads read chunk 0's channels, bit 7 no
// epilogue, bit 8 the epilogue's stores out of range.
#ifndef DREAM_W4_DIAG
#define DREAM_W4_DIAG 0
#endif
// (the non-temporal hint on the output stores, buffer_store_f32_nt, measured 0-3 % SLOWER in round-robin A/B: tools/wino4_diag.py)
#ifndef DREAM_W4_STORE
#define DREAM_W4_STORE buffer_store_f32
#endif
#ifndef DREAM_W4_MIDBARRIER
#define DREAM_W4_MIDBARRIER 0
#endif
#ifndef DREAM_W4_STAGGER_N
#define DREAM_W4_STAGGER_N 0
#define DREAM_W4_STAGGER_PCT 0
#endif
// patch loads of wavefronts 4-7 (wide shape, plain 3x3 conv) this many slots later than those of wavefronts 0-3 (0: same slots)
// cache policy of the wide shape's weight stream / of the patch loads (aux bits: 1 sc0, 2 nt, 16 sc1): A/B builds (tools/wino4_diag.py 60xx)
#ifndef DREAM_W4_WAUX
#define DREAM_W4_WAUX 0
#endif
#ifndef DREAM_W4_XAUX
#define DREAM_W4_XAUX 0
#endif
#ifndef DREAM_W4_STAG_LX
#define DREAM_W4_STAG_LX 0
#endif
#ifndef DREAM_W4_RUNNING_WOFF
#define DREAM_W4_RUNNING_WOFF 1
#endif
// static priority for the second-dispatched half of the workgroup's wavefronts (MI355X_MICROARCH.md "two waves per SIMD", item 4)
#ifndef DREAM_W4_SETPRIO
#define DREAM_W4_SETPRIO 0
#endif

namespace {

struct Wino4Params {
    const float *x;          // [B,H,W,Cin]
    const float *u;          // [Cin/K][36][CoutPad][K] (+ AHEAD zero positions), K = 16 (wide) or 8 (narrow)
    const float *scale;      // per-channel multiplier (eval-mode BatchNorm fold) or null
    const float *shift;      // per-channel addend (bias / BN shift) or null
    const float *residual;   // ReLU mask source (DREAM_CONV_RELUMASK) or addend of the output's shape, or null
    float *y;                // [B,H,W,Cout]  (or [B,H/2,W/2,Cout] with DREAM_CONV_POOL2)
    float *y_full;           // MODE 4 (training forward of a conv that feeds a 2x2 max-pool): the un-pooled [B,H,W,Cout] as well, else null
    int B, H, W, Cin, Cout, CoutPad;
    int TY, TX;              // 4x4 tiles per image
    int ntiles;              // B * TY * TX  (< 2^24)
    int nblk, blk_per_xcd;   // blocks of 16 tiles; each XCD's workgroups walk over a contiguous range of them
    unsigned long long magic_tpi, magic_tx;   // ceil(2^40 / (TY * TX)), ceil(2^40 / TX)
    int flags;
    int out_scale, out_oy, out_ox;   // output pixel of conv position (y, x): (out_scale y + out_oy, out_scale x + out_ox) -- (1, 0, 0) for a
                                     // conv, (2, a, b) for phase (a, b) of a stride-2 transposed conv (y: [B, out_scale H, out_scale W, Cout])
    int in_scale, in_oy, in_ox;      // stored input pixel of conv position (y, x), likewise: (2, a, b) for the phase views of the gradient in the
                                     // transposed conv's data gradient (x: [B, in_scale H, in_scale W, Cin]); read by the PAT kernels only
    int stagger_n, stagger_unit;     // start-up stagger: workgroup w sleeps (hash(w) % stagger_n) * stagger_unit x s_sleep(127) before its first
                                     // block, so that the workgroups' epilogues (a burst of 128 KB of stores per CU) stop coinciding (0: off)
    int ymap;                        // 0: output-channel block = blockIdx.y, every XCD walks all of them over its own eighth of the tile blocks;
                                     // ny (2 | 4): 1-D grid, XCD k owns channel block k % ny and shares the tile blocks with the other 8 / ny - 1
                                     // XCDs of that block (blk_per_xcd = their share): an XCD's L2 then streams ONE block's transformed weights
};

constexpr int W4T = 16;       // tiles per workgroup
constexpr int W4P = 36;       // positions
// weight ring of the wide shape: 8 (six positions ahead) fits since the epilogue rewrite (255 VGPRs, no spills) and is 1.5-3 % faster
// than 6 on every layer (round-robin A/B, profiles/r03_wino4_diag.txt); the narrow shape's positions take half the time: 12 or 18
#ifndef DREAM_W4_RING
#define DREAM_W4_RING 8
#endif
#ifndef DREAM_W4_NARROW_RING
#define DREAM_W4_NARROW_RING 18
#endif
// the phase-pattern kernels (25 positions): weight ring (a divisor of 50) and the distance of pass 1 from the end of the chunk
// (ring 10 = eight positions ahead: 0.62 of the MFMA peak on 256 -> 256 @ 200x200 against 0.56 with a ring of five, tools/microbench_convT4.py;
// the kernel has the registers: 25 accumulators instead of 36)
#ifndef DREAM_W4_PAT_RING
#define DREAM_W4_PAT_RING 10
#endif
#ifndef DREAM_W4_PAT_S1OFF
#define DREAM_W4_PAT_S1OFF 8
#endif


// Two workgroup shapes.  WIDE (layers with more than 64 output channels): 8 wavefronts x 16 output channels, 16 input channels
// per chunk; one workgroup per CU.  NARROW (up to 64 output channels: VGG's conv1_2, its data gradient, the decoder's 64-channel
// layers): 4 wavefronts x 16 channels, 8 input channels per chunk -- the same 16 tiles, the same 48 transform items per wavefront
// and pass, half the MFMAs per position (two k-steps), every LDS region half the size, so that TWO workgroups share a CU and one's
// epilogue and barriers hide behind the other's MFMAs.
template <bool NARROW>
struct W4Cfg {
    static constexpr int NW = NARROW ? 4 : 8;            // wavefronts: 16 output channels each
    static constexpr int K = NARROW ? 8 : 16;            // input channels per chunk
    static constexpr int Q = K / 4;                      // channel quads per chunk (transform items: tile x quad x row)
    static constexpr int KS = K / 4;                     // MFMA k-steps per position
    static constexpr int PAD = 16 * NW;                  // output channels the packed weights are padded to
    // V plane p = 6 i + j: [16 tiles][K channels]: one wavefront-wide b128 (b64) read.  WIDE: planes 272 floats apart (68 float4
    // slots = 4 mod 8), so that the pass-2 stores of the lanes (q = 0..3, j, j + 1) of a store group fall on all 32 banks of the
    // LDS store path.  NARROW: the eight lanes of a store group are (q = 0..1, four consecutive tiles) of ONE plane: any stride.
    static constexpr int VPS = NARROW ? W4T * K : W4T * K + 16;     // plane stride in floats
    static constexpr int VB = W4P * VPS;                 // floats per V buffer
    // staging tile of pass 1: float4 slot [(tile, row)][QS q + j], SROW slots per (tile, row): the 8 lanes of a store group write
    // eight different residues mod 8 (WIDE: q = 0..3 of two consecutive rows, slots 6 q + j and 25 + 6 q + j; NARROW: q = 0..1 of
    // four consecutive rows, slots 7 q + j + 14 r)
    static constexpr int SROW = NARROW ? 14 : 25;
    static constexpr int QS = NARROW ? 7 : 6;
    static constexpr int SB = W4T * 6 * SROW * 4;        // floats
    // operand registers of the weight stream: position k of the chunk of parity PH lives in bq[(36 PH + k) % RING] (chunks run in
    // pairs, so RING must divide 72).  NARROW positions take half the time: twice the positions in flight for the same latency.
    static constexpr int RING = NARROW ? DREAM_W4_NARROW_RING : DREAM_W4_RING;
    static constexpr int AHEAD = RING - 2;               // positions the weight stream runs ahead of the MFMAs (two are in use)
    static_assert(72 % RING == 0, "the weight ring must divide two chunks' positions");
    using vec = std::conditional_t<NARROW, f32x2, f32x4>;   // one lane's MFMA operands of a position: KS floats
};
// float4 slot of channel quad q in row t of a V plane
template <bool NARROW>
DREAM_DEVICE int v4_slot(int q, int t) { return NARROW ? q ^ ((t >> 3) & 1) : q ^ ((t >> 2) & 2); }

DREAM_DEVICE f32x4 fma4(float c, f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_fmaf(c, a[k], b[k]);
    return r;
}

// row j of B^T applied to d[0..5] (see the header); 4 packed operations per float4 pair, nothing kept between rows.  Differences are
// packed by hand (pk_sub4: v_pk_add_f32 with negated second operand): hipcc packs fma and add (v_pk_fma_f32, v_pk_add_f32: two
// channels per instruction) but not sub.
DREAM_DEVICE f32x4 sub4(f32x4 y, f32x4 x) { return pk_sub4(y, x); }
DREAM_DEVICE f32x4 bt_row(int j, const f32x4 *d) {
    switch (j) {
        case 0: return fma4(-2.0f, d[2], fma4(1.5f, sub4(d[3], d[1]), d[0] + d[4]));
        case 1: return fma4(2.5f, d[3], fma4(0.5f, d[2], sub4(d[4], d[1])));
        case 2: return fma4(0.5f, d[3], fma4(-2.5f, d[2], d[4] + d[1]));
        case 3: return fma4(2.0f, sub4(d[3], d[1]), sub4(d[4], d[2]));
        case 4: return fma4(-0.5f, sub4(d[3], d[1]), sub4(d[4], d[2]));
        default: return fma4(-2.0f, d[3], fma4(1.5f, sub4(d[4], d[2]), d[1] + d[5]));
    }
}

// PAT: which of the 36 positions carry non-zero transformed weights.  0: all (3x3 conv).  1 + 2 a + b: phase (a, b) of
// nn.ConvTranspose2d(k4, s2, p1) written as a 3x3 conv whose kernel has only 2 x 2 non-zero taps (rows {0,1} for a = 0, {1,2} for
// a = 1; columns likewise -- conv_wino.hip): G g G^T then vanishes on row 5 (a = 0: the last row of G is (0, 0, 1)) or row 0 (a = 1:
// the first is (1, 0, 0)) of the 6 x 6 domain, and likewise on a column.  25 positions are left -- the F(4x4, 2x2) minimal-filtering
// count, 25 multiplications per 4 x 4 outputs of a phase where F(2x2, 2x2) takes 36 -- and the other eleven are never loaded or
// multiplied.
DREAM_DEVICE constexpr bool pat4_row_active(int pat, int i) { return pat == 0 || (((pat - 1) >> 1) == 0 ? i <= 4 : i >= 1); }
DREAM_DEVICE constexpr bool pat4_col_active(int pat, int j) { return pat == 0 || (((pat - 1) & 1) == 0 ? j <= 4 : j >= 1); }
DREAM_DEVICE constexpr bool pat4_active(int pat, int pp) { return pat4_row_active(pat, pp / 6) && pat4_col_active(pat, pp % 6); }
DREAM_DEVICE constexpr int pat4_count(int pat) { return pat == 0 ? 36 : 25; }
struct Pat4Table { int pos[36]; };                  // pos[k]: the k-th active position (a table: indexed by unrolled-loop constants)
constexpr Pat4Table pat4_table(int pat) {
    Pat4Table t = {};
    int n = 0;
    for (int pp = 0; pp < 36; ++pp)
        if (pat4_active(pat, pp)) t.pos[n++] = pp;
    return t;
}
template <int PAT>
struct Pat4 { static constexpr Pat4Table table = pat4_table(PAT); };
#define pat4_pos(PAT, k) (Pat4<PAT>::table.pos[k])

// MODE: 0 plain, 1 fused 2x2 max-pool, 2 residual add, 3 ReLU mask (conv_wino.hip), 4 fused 2x2 max-pool AND the un-pooled tensor (training)
template <int MODE, bool NARROW, int PAT = 0>
__global__ void __launch_bounds__(64 * W4Cfg<NARROW>::NW, 2) conv_wino4_kernel(const Wino4Params p) {
    using C = W4Cfg<NARROW>;
    using vec = typename C::vec;
    constexpr int W4K = C::K, W4NW = C::NW, V4PS = C::VPS, V4B = C::VB, S4ROW = C::SROW;
    constexpr int NPOS = pat4_count(PAT);              // positions this kernel multiplies
    // the weight ring must divide the positions of two chunks (2 x 25: five or ten registers)
    constexpr int W4_RING = PAT ? DREAM_W4_PAT_RING : C::RING, W4_AHEAD = W4_RING - 2;
    static_assert((2 * NPOS) % W4_RING == 0, "the weight ring must divide two chunks' positions");
    DREAM_DYNAMIC_LDS(float, sV);                      // 2 x V buffer, then the staging tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_index();
    // 384 (narrow: 192) items per pass on 512 (256) threads: every wavefront takes 48 of each pass (lanes 48..63 repeat the items of lanes 32..47:
    // same loads, same values to the same LDS addresses from a different store group -- no branch, no predication), so that all
    // eight wavefronts carry the same work between two barriers and none of them idles at one.
    const int item_id = wave * 48 + (lane < 48 ? lane : lane - 16);

    const int xcd = (int)(blockIdx.x & 7), J = (int)(gridDim.x >> 3);
    const int part = p.ymap ? xcd / p.ymap : xcd;      // which share of the tile blocks this XCD walks
    const int blk_hi = (part + 1) * p.blk_per_xcd;
    const int blk_end = blk_hi < p.nblk ? blk_hi : p.nblk;
    int tb = part * p.blk_per_xcd + (int)(blockIdx.x >> 3);
    if (tb >= blk_end) return;
    const int n0 = (p.ymap ? xcd % p.ymap : (int)blockIdx.y) * (16 * W4NW);
    if (p.stagger_n > 1) {
        // The persistent workgroups start together and take the same time per block, so all 256 CUs reach their epilogues together:
        // 32 MB of stores at once drain at the HBM write rate (~9 us), and on gfx9 the next block's weight loads count behind them in
        // the in-order vmcnt.  Spread over a fraction of a block time, a CU's 128 KB leave in ~2 us.
        const unsigned w = blockIdx.x + blockIdx.y * gridDim.x;
        const int naps = (int)((w * 2654435761u >> 16) % (unsigned)p.stagger_n) * p.stagger_unit;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int tiles_per_img = p.TY * p.TX;
    const int Si = PAT ? p.in_scale : 1;                                      // spacing of the conv positions in the stored input
    const size_t img_floats = (size_t)(Si * p.H) * (Si * p.W) * p.Cin;

    // ---- pass 1 item: (tile t1, channel quad q1, patch row r1)
    const int q1 = item_id % C::Q, r1 = (item_id / C::Q) % 6, t1 = (item_id / C::Q) / 6;
    const int s1_off = ((t1 * 6 + r1) * S4ROW + C::QS * q1) * 4;             // staging float offset of column j = 0
    // ---- pass 2 item: (tile t2, channel quad q2, transformed column j2); narrow: the items run (q, four tiles, j, tile / 4), so
    //      that a store group of eight lanes fills eight consecutive float4 slots of one V plane
    const int q2 = item_id % C::Q;
    const int j2 = NARROW ? (item_id >> 3) % 6 : (item_id >> 2) % 6;
    const int t2 = NARROW ? 4 * (item_id / 48) + ((item_id >> 1) & 3) : (item_id >> 2) / 6;
    const int s2_off = ((t2 * 6) * S4ROW + C::QS * q2 + j2) * 4;              // staging float offset of row r = 0 (rows: + S4ROW * 4)
    const int v2_off = j2 * V4PS + t2 * W4K + 4 * v4_slot<NARROW>(q2, t2);    // V float offset of plane (i = 0, j2); plane 6 i + j2: + 6 i * V4PS

    // offsets of the item's six loads: byte offset of column 0 relative to the first image of the block + a validity bit per
    // column (BUFFER_OOB where the patch leaves the image: the hardware returns zeros).  Recomputed per block from the thread's
    // packed (t1, r1, q1) -- `opaque` keeps the compiler from hoisting the unpacked values into registers it does not have.
    unsigned goff0;
    int item1 = t1 | (r1 << 8) | (q1 << 16);           // bits 24..29: validity of the six columns (plan_item)
    auto plan_item = [&](int tile0, int b0) {
        asm volatile("" : "+v"(item1));
        const int t = item1 & 255, r = (item1 >> 8) & 255, q = (item1 >> 16) & 255;
        const int tau = tile0 + t;
        const bool tv = tau < p.ntiles;
        const int b = div_magic40(tau, p.magic_tpi), rem = tau - b * tiles_per_img;
        const int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
        const int gy = 4 * ty - 1 + r, x0 = 4 * tx - 1;
        const bool rok = tv & ((unsigned)gy < (unsigned)p.H);
        goff0 = PAT ? (unsigned)(((((b - b0) * Si * p.H + Si * gy + p.in_oy) * (Si * p.W) + Si * x0 + p.in_ox) * p.Cin + 4 * q) * 4)
                    : (unsigned)(((((b - b0) * p.H + gy) * p.W + x0) * p.Cin + 4 * q) * 4);
        item1 &= 0xffffff;
#pragma unroll
        for (int c = 0; c < 6; ++c) item1 |= (rok & ((unsigned)(x0 + c) < (unsigned)p.W)) ? (1 << (24 + c)) : 0;
    };
    const unsigned px_in = (unsigned)(Si * p.Cin * 4);
    auto item_offset = [&](int c) {
        if (DREAM_W4_DIAG & 16) {                      // diagnostics: the loads are issued but out of range (zeros, no memory traffic)
            unsigned o = BUFFER_OOB;
            asm volatile("" : "+v"(o));
            return o;
        }
        return (item1 >> (24 + c)) & 1 ? goff0 + (unsigned)c * px_in : BUFFER_OOB;
    };
    auto block_tile0 = [&](int blk) { return blk < blk_end ? blk * W4T : p.ntiles; };
    auto block_xbuf = [&](int b0) {
        return make_buffer(p.x + (size_t)b0 * img_floats, ((size_t)(p.B - b0) * img_floats) * sizeof(float));
    };

    // ---- MFMA operand addresses: A lane l -> tile (l & 15), channels KS (l >> 4) .. + KS - 1 of the chunk (one float4 / float2
    //      feeds the 4 / 2 MFMAs of a position: MFMA r multiplies channel KS (l >> 4) + r, on both operands)
    const int lt = lane & 15, lg = lane >> 4;
    const int a_off = NARROW ? lt * W4K + 4 * v4_slot<NARROW>(lg >> 1, lt) + 2 * (lg & 1) : lt * W4K + 4 * v4_slot<NARROW>(lg, lt);
    const unsigned b_lane = (unsigned)(((wave * 16 + lt) * W4K + C::KS * lg) * 4);
    const unsigned u_pos_stride = (unsigned)(p.CoutPad * W4K * 4);
    const BufferRsrc ubuf = make_buffer(p.u + (size_t)n0 * W4K, ((size_t)((p.Cin / W4K) * W4P + C::AHEAD) * p.CoutPad - (size_t)n0) * W4K * sizeof(float));

    f32x4 acc[W4P];
    const int nchunks = p.Cin / W4K;
    if (DREAM_W4_SETPRIO && wave >= W4NW / 2) __builtin_amdgcn_s_setprio(1);

    // weight stream: the k-th position of the chunk sequence (k counted from the start of the block, 36 per chunk) lives in
    // bq[k % 8]; the chunk loop is unrolled by two so that the ring index is a compile-time constant (72 % 8 == 0)
    vec bq[W4_RING];
    auto load_u = [&](unsigned soff) {
        if constexpr (NARROW) return buffer_load_x2(ubuf, b_lane, soff);
        else return buffer_load_x4_aux<DREAM_W4_WAUX>(ubuf, b_lane, soff);
    };
#pragma unroll
    for (int k = 0; k < W4_AHEAD; ++k) bq[k] = load_u((unsigned)pat4_pos(PAT, k) * u_pos_stride);
    // The full kernel's weight stream is SEQUENTIAL in memory (position k of chunk c sits at (36 c + k) positions): one running
    // wave-uniform byte offset, bumped after every load and reset where the stream wraps around to the next block's chunk 0.  (As
    // `(36 c + k) * stride` hipcc kept the 36 products k * stride in scalar registers it does not have: 18 v_readlane_b32 + their
    // hazard s_nops per chunk in the MFMA loop.)  The phase patterns skip positions and keep the table form.
    unsigned woff = (unsigned)W4_AHEAD * u_pos_stride;
    DREAM_OPAQUE_SGPR(woff);

    // LDS addresses: one per-lane base register per (role, V buffer), everything else in the instructions' 16-bit immediate
    // offsets (every plane / row offset below is < 64 KB from its base).  `opaque` keeps the compiler from folding the buffer
    // offsets into per-access address registers (it would need one per plane beyond 64 KB).  The chunks of a block alternate
    // between the two V buffers and their number is even, so chunk parity PH reads buffer PH and fills buffer 1 - PH.
    // (offsets in float4 units, so that the accesses stay provably 16-byte aligned: b128 LDS instructions)
    f32x4 *const sV4 = (f32x4 *)sV;
    vec *const sVa = (vec *)sV;                        // the A operands: float4 (float2) units
    int vrd[2], vwr[2], s1p = (2 * V4B + s1_off) / 4, s2p = (2 * V4B + s2_off) / 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        vrd[h] = (h * V4B + a_off) / C::KS;
        vwr[h] = (h * V4B + v2_off) / 4;
        asm volatile("" : "+v"(vrd[h]), "+v"(vwr[h]));
    }
    asm volatile("" : "+v"(s1p), "+v"(s2p));
    f32x4 d[6];                                        // pass 1: the patch row; pass 2: the staged column
    auto pass1_piece = [&](int j) { sV4[s1p + j] = bt_row(j, d); };
    auto pass2_read = [&](int r) { d[r] = sV4[s2p + r * S4ROW]; };
    auto pass2_piece = [&](int i, int h) { if (pat4_row_active(PAT, i)) sV4[vwr[h] + i * (6 * V4PS / 4)] = bt_row(i, d); };   // rows no MFMA reads are not made

    vec a[2][2];                                       // [set][position of the pair]
    // narrow: two ds_read_b64 (2 LDS cycles each, 64 banks: conflict-free on this layout); merged into one ds_read2st64_b64 by the
    // compiler they would run in 16-lane groups on 32 banks -- 8 cycles, and 2-way conflicts on top.  Volatile accesses are not merged.
    auto read_a = [&](int set, int k, int h) {         // the A operands of active positions k and k + 1
        const int p0 = pat4_pos(PAT, k), p1 = pat4_pos(PAT, k + 1 < NPOS ? k + 1 : k);
        if constexpr (NARROW) {
            a[set][0] = lds_read_unmerged(sVa + vrd[h] + p0 * V4PS / C::KS);
            a[set][1] = lds_read_unmerged(sVa + vrd[h] + p1 * V4PS / C::KS);
        } else {
            a[set][0] = sVa[vrd[h] + p0 * V4PS / C::KS];
            if (k + 1 < NPOS) a[set][1] = sVa[vrd[h] + p1 * V4PS / C::KS];
        }
    };

    // ---- first block of this workgroup: plan, chunk 0 through both passes into buffer 0 ------------------------------------------
    int tile0 = block_tile0(tb);
    int b0 = div_magic40(tile0, p.magic_tpi);
    BufferRsrc xbuf = block_xbuf(b0);
    plan_item(tile0, b0);
#pragma unroll
    for (int c = 0; c < 6; ++c) d[c] = buffer_load_x4(xbuf, item_offset(c), 0u);
#pragma unroll
    for (int j = 0; j < 6; ++j) pass1_piece(j);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; ++r) pass2_read(r);
#pragma unroll
    for (int i = 0; i < 6; ++i) pass2_piece(i, 0);
    __syncthreads();

    // One chunk: 18 slots of two positions x 4 k-steps on V buffer PH; meanwhile the NEXT chunk goes through the two passes:
    // its six patch loads behind the first MFMA pairs, pass 1 in slots S1 .. S1 + 2 (two pieces per slot), a barrier, the six
    // staging reads in slot S2, pass 2 in slots S2 + 1 .. S2 + 3, the end-of-chunk barrier.  PH: parity of the chunk inside
    // the block (chunks run in pairs: the ring index of the weight stream is a compile-time constant, 72 % 8 == 0).  `last`
    // (wave-uniform; only the odd chunk can be the last one): the next chunk is chunk 0 of the NEXT block -- its plan is
    // computed here, the weight stream wraps around.  One instantiation per parity, one call site each: the loop below.
    auto chunk = [&](auto ph_tag, auto stag_tag, bool last, int c, int tile0n, int b0n) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int STAG = decltype(stag_tag)::value;
#ifndef DREAM_W4_S1
#define DREAM_W4_S1 10
#define DREAM_W4_S2 13
#define DREAM_W4_LX 3
#endif
        // NSLOT slots of two active positions (18; 13 with a phase pattern, the last one a single position).  The full kernel's
        // schedule is S1 10 / S2 13 / loads in slots 0..2; a pattern keeps the distances from the END of the chunk (S1 5, S2 8) and
        // issues its six patch loads in slot 0, so that they have four slots to arrive.
        constexpr int NSLOT = (NPOS + 1) / 2;
        constexpr int S1 = PAT ? NSLOT - DREAM_W4_PAT_S1OFF : DREAM_W4_S1, S2 = PAT ? S1 + 3 : DREAM_W4_S2; // pass 1 in slots S1 .. S1 + 2; staging reads in slot S2, pass 2 in S2 + 1 .. S2 + 3
        constexpr int LX = PAT ? 1 : DREAM_W4_LX;         // patch loads in slots LX0 .. LX0 + LX - 1 (6 / LX per slot)
        // STAG (wavefronts 4-7 of the wide shape: the SECOND wavefront of every SIMD): the patch loads come DREAM_W4_STAG_LX slots later.  A
        // patch load misses to HBM, loads return in order, so the weight operands issued behind it stall the wavefront three slots
        // later -- both wavefronts of a SIMD at the same slots, matrix pipe idle.  Shifted, one wavefront's MFMAs cover the other's stall.
        // (The whole block loop is instantiated twice, chosen once per wavefront: no branch and no control-flow join inside it.)
        constexpr int LX0 = STAG ? DREAM_W4_STAG_LX : 0;
        static_assert(LX0 + LX <= S1, "the patch loads must be issued before pass 1");
        const unsigned coff = last ? 0u : (unsigned)((c + 1) * W4K * 4);
        const int cnext = last ? 0 : (c + 1) * W4P;                          // first position of the next chunk in the weight stream
        if (PH == 1 && last) xbuf = block_xbuf(b0n);                        // this block's loads are all issued: from here on the next block's
        read_a(0, 0, PH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int k0 = 2 * s;                                            // the slot's active positions k0 (and k0 + 1)
            const bool two = k0 + 1 < NPOS;
            const int pp = pat4_pos(PAT, k0), pp1 = pat4_pos(PAT, two ? k0 + 1 : k0);
            auto load_b = [&](int half) {                                    // weight operands of active position k0 + half + AHEAD
                if ((DREAM_W4_DIAG & 2) || (half && !two)) return;
                const int kn = k0 + half + W4_AHEAD;
                if constexpr (PAT == 0 && DREAM_W4_RUNNING_WOFF && !(DREAM_W4_DIAG & 32)) {
                    if (kn == NPOS && last) woff = 0u;                        // wave-uniform: s_cselect
                    bq[(PH * NPOS + kn) % W4_RING] = load_u(woff);
                    woff += u_pos_stride;
                    DREAM_OPAQUE_SGPR(woff);                            // opaque: no re-derivation from the chunk counter
                } else {
                    const int spos = kn >= NPOS ? cnext + pat4_pos(PAT, kn - NPOS) : c * W4P + pat4_pos(PAT, kn);
                    bq[(PH * NPOS + kn) % W4_RING] = load_u((DREAM_W4_DIAG & 32) ? 0u : (unsigned)spos * u_pos_stride);
                }
            };
            auto load_x = [&](int col) { if (!(DREAM_W4_DIAG & 1)) d[col] = buffer_load_x4_aux<DREAM_W4_XAUX>(xbuf, item_offset(col), (DREAM_W4_DIAG & 64) ? 0u : coff); };
            auto pair = [&](int r) {
                const int i0 = (PH * NPOS + k0) % W4_RING, i1 = (PH * NPOS + k0 + 1) % W4_RING;
                acc[pp] = mfma_f32_16x16x4(a[s & 1][0][r], bq[i0][r], acc[pp]);
                if (two) acc[pp1] = mfma_f32_16x16x4(a[s & 1][1][r], bq[i1][r], acc[pp1]);
                __builtin_amdgcn_sched_barrier(0);
            };
            // the four interleave points of a slot; the narrow shape has two MFMA pairs per slot: two points behind each
            auto after = [&](int k) {
                if (k == 0) {
                    load_b(0);
                    if (s + 1 < NSLOT) read_a((s + 1) & 1, k0 + 2, PH);
                    if (PH == 1 && s == 0 && last) plan_item(tile0n, b0n);
                    if (s >= LX0 && s < LX0 + LX) { for (int c2 = 0; c2 < 3 / LX; ++c2) load_x((6 / LX) * (s - LX0) + c2); }
                    if (!(DREAM_W4_DIAG & 8) && s >= S1 && s < S1 + 3) pass1_piece(2 * (s - S1));
                    if (!(DREAM_W4_DIAG & 8) && s == S2) { pass2_read(0); pass2_read(1); }
                    if (!(DREAM_W4_DIAG & 8) && s > S2 && s <= S2 + 3) pass2_piece(2 * (s - S2 - 1), 1 - PH);
                } else if (k == 1) {
                    load_b(1);
                    if (s >= LX0 && s < LX0 + LX) { for (int c2 = 3 / LX; c2 < 6 / LX; ++c2) load_x((6 / LX) * (s - LX0) + c2); }
                    if (!(DREAM_W4_DIAG & 8) && s == S2) { pass2_read(2); pass2_read(3); }
                } else if (k == 2) {
                    if (!(DREAM_W4_DIAG & 8) && s >= S1 && s < S1 + 3) pass1_piece(2 * (s - S1) + 1);
                    if (!(DREAM_W4_DIAG & 8) && s == S2) { pass2_read(4); pass2_read(5); }
                    if (!(DREAM_W4_DIAG & 8) && s > S2 && s <= S2 + 3) pass2_piece(2 * (s - S2 - 1) + 1, 1 - PH);
                }
            };
            if constexpr (NARROW) {
                pair(0);
                after(0);
                after(1);
                pair(1);
                after(2);
            } else {
                pair(0);
                after(0);
                pair(1);
                after(1);
                pair(2);
                after(2);
                pair(3);
            }
            if (s == S1 + 2 && !(DREAM_W4_DIAG & 4)) {                       // pass 1 of the next chunk is staged
                // No workgroup barrier is needed here: the (tile, quad, row) items of pass 1 and the (tile, quad, column) items
                // of pass 2 of one wavefront cover the SAME tiles (wide: tiles 2w, 2w + 1; narrow: 4w .. 4w + 3), so a
                // wavefront's pass 2 reads only what that wavefront's pass 1 wrote, and a wavefront's LDS instructions execute
                // in order.  (DREAM_W4_MIDBARRIER=1 restores the barrier of round 3 for A/B runs.)
                if (DREAM_W4_MIDBARRIER) __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(DREAM_W4_DIAG & 4)) __syncthreads();
    };
    const std::integral_constant<int, 0> ph0{};
    const std::integral_constant<int, 1> ph1{};

    // ---- inverse transform Y = A^T M A (lane-local), scale / shift / residual / ReLU / 2x2 max-pool, store ------------------------
    constexpr bool pool = MODE == 1 || MODE == 4, both = MODE == 4, has_res = MODE == 2 || MODE == 3, mask = MODE == 3;
    auto epilogue = [&](auto site_tag, int tile0e, int b0e) {
        // Everything the epilogue needs is read again from the kernel-argument segment (scalar loads, once per block): kept in
        // SGPRs across the MFMA phases these values push the kernel past its scalar register file (spills through VGPR lanes).
        if (DREAM_W4_DIAG & 128) {                     // diagnostics: no epilogue (the accumulators stay live)
#pragma unroll
            for (int pp = 0; pp < W4P; ++pp)
                if (pat4_active(PAT, pp)) asm volatile("" :: "v"(acc[pp]));
            return;
        }
        const auto &e = *DREAM_KERNARG_SITE(p, decltype(site_tag)::value);
        const bool relu = (e.flags & DREAM_CONV_RELU) != 0;
        const bool late = MODE == 2 && (e.flags & DREAM_CONV_RES_AFTER_RELU) != 0;    // the residual is a skip connection: added after the ReLU
        // Ho x Wo: grid of stored conv positions; So: their spacing in the stored tensor (2 for a transposed conv's phase)
        const int Ho = pool ? e.H / 2 : e.H, Wo = pool ? e.W / 2 : e.W, So = e.out_scale;
        const size_t out_img = (size_t)(So * Ho) * (So * Wo) * e.Cout;
        const unsigned px_b = (unsigned)(So * e.Cout * 4), row_b = (unsigned)(So * So * Wo * e.Cout * 4);
        const int ln = lane_id();                      // from the hardware (mbcnt), not from a register kept across the MFMA phases
        const int lt = ln & 15, lg = ln >> 4;
        const int col = n0 + wave * 16 + lt;
        const bool cok = col < e.Cout;
        const float sc = (e.scale != nullptr && cok) ? e.scale[col] : 1.0f;
        const float sh = (e.shift != nullptr && cok) ? e.shift[col] : 0.0f;
        const BufferRsrc ybuf = make_buffer(e.y + (size_t)b0e * out_img, (DREAM_W4_DIAG & 256) ? 0 : (size_t)(e.B - b0e) * out_img * sizeof(float));
        const BufferRsrc rbuf = make_buffer(has_res ? e.residual + (size_t)b0e * out_img : e.y,
                                            has_res ? (size_t)(e.B - b0e) * out_img * sizeof(float) : 0);
        // MODE 4: the un-pooled tensor [B, H, W, Cout] as well (the training forward pass keeps it for the backward pass: ReLU mask, pool
        // routing), stored from the same registers the pooled maxima are taken from -- no separate max-pool pass over it
        const size_t full_img = (size_t)e.H * e.W * e.Cout;
        const unsigned fpx_b = (unsigned)(e.Cout * 4), frow_b = (unsigned)(e.W * e.Cout * 4);
        const BufferRsrc fbuf = make_buffer(both ? e.y_full + (size_t)b0e * full_img : e.y, both ? (size_t)(e.B - b0e) * full_img * sizeof(float) : 0);
        // C/D layout: reg r of lane l is tile 4 (l >> 4) + r of the block: the lane's four tiles are consecutive.  They go through the
        // inverse transform in PAIRS (regs 2 rp, 2 rp + 1: an aligned register pair of every accumulator), on packed fp32 operations.
        const int tau0 = tile0e + lg * 4;
        int b = div_magic40(tau0, p.magic_tpi);
        const int rem = tau0 - b * tiles_per_img;
        int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
        constexpr int NS = pool ? 2 : 4;               // stored positions of a tile: a 4x4 block of outputs, or its 2x2 block of pooled outputs
        // stored-position bases and bounds of the lane's four tiles (two pairs), up front: the second pair's residual / mask loads start
        // while the first pair is still being stored
        unsigned basef[2][2];                          // MODE 4: the same for the un-pooled tensor (4x4 stored positions per tile)
        int limf[2][2];
        unsigned base4[2][2];
        int lim4[2][2];                                // bits 0..3: stored row i inside the image, bits 4..7: stored column jj (0 for a tile that stores nothing)
        constexpr int FULL = pool ? 0x33 : 0xff;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool tok = cok & ((tau0 + 2 * rp + h) < p.ntiles);
                const int oy = NS * ty, ox = NS * tx;
                base4[rp][h] = (unsigned)(((((b - b0e) * So * Ho + So * oy + e.out_oy) * (So * Wo) + So * ox + e.out_ox) * e.Cout + col) * 4);
                int m = 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) m |= ((oy + i) < Ho ? 1 << i : 0) | ((ox + i) < Wo ? 16 << i : 0);
                lim4[rp][h] = tok ? m : 0;
                if (both) {
                    basef[rp][h] = (unsigned)(((((b - b0e) * e.H + 4 * ty) * e.W + 4 * tx) * e.Cout + col) * 4);
                    int mf = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) mf |= ((4 * ty + i) < e.H ? 1 << i : 0) | ((4 * tx + i) < e.W ? 16 << i : 0);
                    limf[rp][h] = tok ? mf : 0;
                }
                const bool wrap_x = (tx + 1 == p.TX);
                const bool wrap_y = wrap_x & (ty + 1 == p.TY);
                tx = wrap_x ? 0 : tx + 1;
                ty = wrap_y ? 0 : (wrap_x ? ty + 1 : ty);
                b += wrap_y ? 1 : 0;
            }
        // byte offset of stored position (i, jj) of tile h of pair rp, BUFFER_OOB where nothing may be written -- computed where it is
        // used (the accumulators leave no registers for a table)
        auto out_off4 = [&](int rp, int h, int i, int jj) {
            const bool inb = ((lim4[rp][h] >> i) & (lim4[rp][h] >> (4 + jj)) & 1) != 0;
            return inb ? base4[rp][h] + (unsigned)i * row_b + (unsigned)jj * px_b : BUFFER_OOB;
        };
        // `interior` (wave-uniform) -- every tile of the wavefront's pair lies wholly inside the image: the lane's offset is the tile's base
        // and the position inside the tile rides in the scalar offset; otherwise per-element masks
        bool interior4[2];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) interior4[rp] = wave_all((lim4[rp][0] == FULL) & (lim4[rp][1] == FULL));
        // residual / mask values of column 0 of a pair (8 loads): in flight during the row transforms (pair 0) / the last column of pair 0
        auto load_col0 = [&](int rp, float (&r)[2][4]) {
            if (interior4[rp]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[h][i] = buffer_load_f32(rbuf, base4[rp][h], (unsigned)i * row_b);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[h][i] = buffer_load_f32(rbuf, out_off4(rp, h, i, 0), 0u);
            }
        };
        float r0[2][4], r0n[2][4];
        if (has_res) load_col0(0, r0);
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const unsigned (&base)[2] = base4[rp];
            auto out_off = [&](int h, int i, int jj) { return out_off4(rp, h, i, jj); };
            const bool all_interior = interior4[rp];
            auto fm = [](float c, f32x2 x, f32x2 y) { return __builtin_elementwise_fma(f32x2{c, c}, x, y); };
            // A^T M A in two lane-local steps that shrink the live set: rows first (6 x 6 -> 6 x 4, in place of the
            // accumulator values just read), then one output column at a time, stored as soon as it exists
            f32x2 sA[6][4];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x2 m[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)          // positions without weights (PAT) were never multiplied: zeros of the sums
                    m[j] = pat4_active(PAT, 6 * i + j) ? f32x2{acc[6 * i + j][2 * rp], acc[6 * i + j][2 * rp + 1]} : f32x2{0.0f, 0.0f};
                const f32x2 t1 = pk_sub2(m[1], m[2]), t2 = m[1] + m[2];
                sA[i][0] = (m[0] + t2) + (m[3] + m[4]);
                sA[i][1] = fm(-2.0f, m[4], fm(0.5f, m[3], t1));
                sA[i][2] = fm(4.0f, m[4], fm(0.25f, m[3], t2));
                sA[i][3] = fm(-8.0f, m[4], fm(0.125f, m[3], t1)) + m[5];
            }
            // The residual / ReLU-mask values of a column are loaded as a GROUP, one column ahead of their use: the loads of column
            // jj + 1 are issued before the stores of column jj.  (Loaded where they are used -- round 3-4 -- every one of the 64 was
            // followed by s_waitcnt vmcnt(0) and its store: the compiler cannot move a load above the previous store, the two buffers
            // may alias (they DO in the transposed conv's data gradient, which accumulates its phases in place) -- 64 serialised
            // memory round trips per wavefront and block in the data-gradient launches.  Safe under aliasing: a thread reads exactly the
            // elements it writes, and the loads of column jj + 1 touch other elements than the stores of column jj.)
            auto finish = [&](float v, float rv) {                    // residual or mask, ReLU (scale / shift applied before)
                if (has_res) v = mask ? (rv > 0.0f ? v : 0.0f) : (late ? v : v + rv);
                const float vr = fmaxf(v, 0.0f);
                v = relu ? vr : v;
                return (MODE == 2 && late) ? v + rv : v;
            };
            // `interior` (wave-uniform) -- every tile of the wavefront lies wholly inside the image: the lane's offset is the tile's base
            // and the position inside the tile rides in the scalar offset; otherwise per-element masks
            auto columns = [&](auto interior_tag, auto &rcur) {
                constexpr bool interior = decltype(interior_tag)::value;
                auto voff = [&](int h, int i, int j2) { return interior ? base[h] : out_off(h, i, j2); };
                auto soff = [&](int i, int j2) { return interior ? (unsigned)i * row_b + (unsigned)j2 * px_b : 0u; };
                float keep[2][2];                                     // pool: the even column's values wait for the odd one
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float rnext[2][4];
                    if (has_res && jj == 3 && rp == 0) load_col0(1, r0n);
                    if (has_res && jj + 1 < 4) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int i = 0; i < 4; ++i) rnext[h][i] = buffer_load_f32(rbuf, voff(h, i, jj + 1), soff(i, jj + 1));
                    }
                    const f32x2 t1 = pk_sub2(sA[1][jj], sA[2][jj]), t2 = sA[1][jj] + sA[2][jj];
                    f32x2 o4[4];
                    o4[0] = (sA[0][jj] + t2) + (sA[3][jj] + sA[4][jj]);
                    o4[1] = fm(-2.0f, sA[4][jj], fm(0.5f, sA[3][jj], t1));
                    o4[2] = fm(4.0f, sA[4][jj], fm(0.25f, sA[3][jj], t2));
                    o4[3] = fm(-8.0f, sA[4][jj], fm(0.125f, sA[3][jj], t1)) + sA[5][jj];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = o4[i] * f32x2{sc, sc} + f32x2{sh, sh};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (pool) {
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = finish(o4[i][h], 0.0f);
                            if (both) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const bool inb = ((limf[rp][h] >> i) & (limf[rp][h] >> (4 + jj)) & 1) != 0;
                                    DREAM_W4_STORE(fbuf, v[i], inb ? basef[rp][h] + (unsigned)i * frow_b + (unsigned)jj * fpx_b : BUFFER_OOB, 0u);
                                }
                            }
                            if ((jj & 1) == 0) {
                                keep[h][0] = fmaxf(v[0], v[1]);
                                keep[h][1] = fmaxf(v[2], v[3]);
                            } else {
                                DREAM_W4_STORE(ybuf, fmaxf(keep[h][0], fmaxf(v[0], v[1])), voff(h, 0, jj >> 1), soff(0, jj >> 1));
                                DREAM_W4_STORE(ybuf, fmaxf(keep[h][1], fmaxf(v[2], v[3])), voff(h, 1, jj >> 1), soff(1, jj >> 1));
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                DREAM_W4_STORE(ybuf, finish(o4[i][h], has_res ? rcur[h][i] : 0.0f), voff(h, i, jj), soff(i, jj));
                        }
                    }
                    if (has_res && jj + 1 < 4) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int i = 0; i < 4; ++i) rcur[h][i] = rnext[h][i];
                    }
                }
            };
            if (all_interior) columns(std::true_type{}, r0);
            else columns(std::false_type{}, r0);
            if (has_res && rp == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) r0[h][i] = r0n[h][i];
            }
        }
    };

    // ---- the blocks of this workgroup (chunks in pairs: nchunks is even, host side) ----------------------------------------------
    auto run_blocks = [&](auto stag_tag) __attribute__((always_inline)) {
        while (true) {
            const int tile0n = block_tile0(tb + J);
            const int b0n = div_magic40(tile0n, p.magic_tpi);
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int pp = 0; pp < W4P; ++pp)
                if (pat4_active(PAT, pp)) acc[pp] = zero;
            for (int c = 0; c < nchunks; c += 2) {
                chunk(ph0, stag_tag, false, c, tile0n, b0n);
                chunk(ph1, stag_tag, c + 2 == nchunks, c + 1, tile0n, b0n);
            }
            epilogue(stag_tag, tile0, b0);
            tb += J;
            if (tb >= blk_end) break;
            tile0 = tile0n;
            b0 = b0n;
        }
    };
    if constexpr (!NARROW && PAT == 0 && DREAM_W4_STAG_LX > 0) {
        if (wave >= W4NW / 2) run_blocks(std::integral_constant<int, 1>{});
        else run_blocks(std::integral_constant<int, 0>{});
    } else {
        run_blocks(std::integral_constant<int, 0>{});
    }
}

// OIHW (mode 0) or, for the data-gradient operator, IOHW with flipped taps (mode 1) -> U = G g G^T in fp64, rounded once to fp32,
// laid out [cols/K][36 positions][RowsPad][K] with (K, RowsPad) = (16, rows up to a multiple of 128), or (8, 64) for rows <= 64
// (the narrow workgroup shape): pack_device.h (dream_pack::winograd4)
static_assert(W4Cfg<false>::K == 16 && W4Cfg<false>::PAD == 128 && W4Cfg<true>::K == 8 && W4Cfg<true>::PAD == 64 && W4P == 36,
              "pack_device.h assumes 16-channel chunks padded to 128 rows / 8-channel chunks padded to 64 rows, 36 positions");
__global__ void __launch_bounds__(256) wino4_pack_kernel(const float *w, float *u, int Cout, int Cin, int mode) {
    dream_pack::winograd4(w, u, Cout, Cin, mode, (int)blockIdx.x, (int)gridDim.x);
}

int g_max_workgroups4 = 0;     // test hook: cap on resident workgroups (0 = the chip's 256 CUs, two narrow workgroups on each)
int g_stagger_n4 = -1, g_stagger_pct4 = 0;   // start-up stagger (Wino4Params::stagger_n): phases and spread in per cent of a block's time; -1 = by
                               // the environment (DREAM_W4_STAGGER="<phases>,<per cent>", default off)
int g_ymap4 = -1;              // channel blocks pinned to XCDs (Wino4Params::ymap): -1 = by the environment (DREAM_W4_YMAP=0 switches it off).
                               // Measured round 4 (tools/ab_wino4_pinning.py, b=128): 0.5-1.3 % faster on every layer with 2 or 4 channel
                               // blocks, headline +0.6 %: an XCD's L2 streams one block's transformed weights instead of all four

bool narrow_rows(int rows) { return rows <= W4Cfg<true>::PAD; }
struct PackShape { int k, rows_pad, ahead; };
PackShape pack_shape(int rows) {
    if (narrow_rows(rows)) return {W4Cfg<true>::K, W4Cfg<true>::PAD, W4Cfg<true>::AHEAD};
    return {W4Cfg<false>::K, (rows + W4Cfg<false>::PAD - 1) / W4Cfg<false>::PAD * W4Cfg<false>::PAD, W4Cfg<false>::AHEAD};
}

template <int MODE, bool NARROW, int PAT = 0>
int launch_wino4(const Wino4Params &p, void *stream) {
    using C = W4Cfg<NARROW>;
    void (*kernel)(const Wino4Params) = conv_wino4_kernel<MODE, NARROW, PAT>;
    const size_t lds = ((size_t)2 * C::VB + C::SB) * sizeof(float);
    if (dream_allow_full_lds((const void *)kernel)) return 2;
    const int ny = (p.Cout + 16 * C::NW - 1) / (16 * C::NW);
    const int resident = g_max_workgroups4 > 0 ? g_max_workgroups4 : (NARROW ? 512 : 256);
    Wino4Params q = p;
    dim3 grid;
    if (g_ymap4 && !NARROW && (ny == 2 || ny == 4) && resident >= 64 && p.nblk >= resident) {
        // channel blocks pinned to XCDs (see Wino4Params::ymap): every XCD runs resident / 8 workgroups on its share of the tile blocks
        q.ymap = ny;
        q.blk_per_xcd = (p.nblk + 8 / ny - 1) / (8 / ny);
        grid = dim3((unsigned)(resident / 8 * 8), 1u);
    } else {
        int gx = resident / ny / 8 * 8;
        if (gx < 8) gx = 8;
        if (gx > (p.nblk + 7) / 8 * 8) gx = (p.nblk + 7) / 8 * 8;
        grid = dim3((unsigned)gx, (unsigned)ny);
    }
    hipLaunchKernelGGL(kernel, grid, dim3(64 * C::NW), lds, (hipStream_t)stream, q);
    DREAM_LAUNCH_OK();
    return 0;
}

template <bool NARROW>
int launch_wino4_mode(const Wino4Params &p, int mode, void *stream) {
    switch (mode) {
        case 0: return launch_wino4<0, NARROW>(p, stream);
        case 1: return launch_wino4<1, NARROW>(p, stream);
        case 2: return launch_wino4<2, NARROW>(p, stream);
        case 4: return launch_wino4<4, NARROW>(p, stream);
        default: return launch_wino4<3, NARROW>(p, stream);
    }
}

}  // namespace

extern "C" size_t dream_conv3x3_winograd4_weight_floats(int rows, int cols) {
    const PackShape ps = pack_shape(rows);
    return ((size_t)(cols / ps.k) * W4P + ps.ahead) * ps.rows_pad * ps.k;
}

// A/B hook: 1 = pin the output-channel blocks to XCDs (Wino4Params::ymap), 0 = every XCD walks all of them (blockIdx.y),
// -1 = by the environment (DREAM_W4_YMAP=0 switches it off; default on).  Same results either way (bit for bit).
extern "C" int dream_conv3x3_winograd4_set_channel_block_pinning(int on) {
    DREAM_REQUIRE(on >= -1 && on <= 1, "winograd F(4x4): pinning %d", on);
    g_ymap4 = on;
    return 0;
}

// A/B hook: start-up stagger of the persistent workgroups: `phases` (0 / 1 = off) spread over `percent` of one block's time; phases < 0 = by
// DREAM_W4_STAGGER again.  Same results either way (bit for bit).
extern "C" int dream_conv3x3_winograd4_set_stagger(int phases, int percent) {
    DREAM_REQUIRE(phases <= 1024 && percent >= 0 && percent <= 400, "winograd F(4x4): stagger %d phases over %d %%", phases, percent);
    g_stagger_n4 = phases < 0 ? -1 : (phases > 1 ? phases : 0);
    g_stagger_pct4 = percent;
    return 0;
}

extern "C" int dream_conv3x3_winograd4_set_max_workgroups(int n) {
    DREAM_REQUIRE(n >= 0, "winograd F(4x4): max workgroups %d", n);
    g_max_workgroups4 = n;
    return 0;
}

// w: OIHW [Cout,Cin,3,3]; mode 0: forward operator (rows = Cout, cols = Cin); mode 1: data-gradient operator (rows = Cin,
// cols = Cout, taps flipped).  u: dream_conv3x3_winograd4_weight_floats(rows, cols) floats.
extern "C" int dream_pack_conv3x3_winograd4_weight(const float *w_oihw, float *u, int Cout, int Cin, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && u && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "winograd F(4x4) pack: bad arguments");
    const int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;
    const PackShape ps = pack_shape(rows);
    DREAM_REQUIRE(cols % (2 * ps.k) == 0, "winograd F(4x4) pack: %d input channels, must be a multiple of %d", cols, 2 * ps.k);
    const size_t total = (size_t)(cols / ps.k) * ps.rows_pad * ps.k;
    size_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_oihw, u, Cout, Cin, mode);
    DREAM_LAUNCH_OK();
    if (dream_zero_words(u + (size_t)(cols / ps.k) * W4P * ps.rows_pad * ps.k, (size_t)ps.ahead * ps.rows_pad * ps.k * sizeof(float), (hipStream_t)stream))
        return 2;
    return 0;
}

namespace {

// geometry shared by the conv and the transposed-conv entry points
int wino4_setup(Wino4Params &p, const float *x, const float *u_packed, const float *scale, const float *shift, const float *residual,
                float *y, int B, int H, int W, int Cin, int Cout, int flags, int out_scale, int in_scale = 1) {
    DREAM_REQUIRE(x && u_packed && y, "winograd F(4x4) conv: null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "winograd F(4x4) conv: bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    const PackShape ps = pack_shape(Cout);
    DREAM_REQUIRE(Cin % (2 * ps.k) == 0, "winograd F(4x4) conv: Cin=%d must be a multiple of %d", Cin, 2 * ps.k);
    const size_t span_imgs = (size_t)W4T / ((size_t)((H + 3) / 4) * ((W + 3) / 4)) + 2;
    DREAM_REQUIRE(span_imgs * in_scale * in_scale * H * W * (size_t)Cin * sizeof(float) < ((size_t)1 << 31) &&
                  span_imgs * out_scale * out_scale * H * W * (size_t)Cout * sizeof(float) < ((size_t)1 << 31),
                  "winograd F(4x4) conv: image too large for 32-bit offsets");
    p.x = x; p.u = u_packed; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y; p.y_full = nullptr;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.CoutPad = Cout <= ps.rows_pad ? ps.rows_pad : (Cout + ps.rows_pad - 1) / ps.rows_pad * ps.rows_pad;
    DREAM_REQUIRE(((size_t)(Cin / ps.k) * W4P + ps.ahead) * (size_t)p.CoutPad * ps.k * sizeof(float) < ((size_t)1 << 31), "winograd F(4x4) conv: weights too large");
    p.TY = (H + 3) / 4; p.TX = (W + 3) / 4;
    const long ntiles = (long)B * p.TY * p.TX;
    DREAM_REQUIRE(ntiles < ((long)1 << 24), "winograd F(4x4) conv: %ld tiles, the tile decomposition handles < 2^24", ntiles);
    p.ntiles = (int)ntiles;
    p.nblk = (p.ntiles + W4T - 1) / W4T;
    p.blk_per_xcd = (p.nblk + 7) / 8;
    p.magic_tpi = (((unsigned long long)1 << 40) + (unsigned long long)(p.TY * p.TX) - 1) / (unsigned long long)(p.TY * p.TX);
    p.magic_tx = (((unsigned long long)1 << 40) + (unsigned long long)p.TX - 1) / (unsigned long long)p.TX;
    p.flags = flags;
    p.out_scale = out_scale; p.out_oy = 0; p.out_ox = 0;
    p.in_scale = in_scale; p.in_oy = 0; p.in_ox = 0;
    p.ymap = 0;
    if (g_stagger_n4 < 0) {
        const char *e = getenv("DREAM_W4_STAGGER");
        int n = DREAM_W4_STAGGER_N, pct = DREAM_W4_STAGGER_PCT;
        if (e != nullptr && sscanf(e, "%d,%d", &n, &pct) != 2) { n = DREAM_W4_STAGGER_N; pct = DREAM_W4_STAGGER_PCT; }
        g_stagger_n4 = n > 1 ? n : 0;
        g_stagger_pct4 = pct;
    }
    // a chunk takes ~13.5 k cycles (wide; the narrow shape's half-size chunks ~7 k), one s_sleep(127) 8128: naps per phase step
    {
        const double chunk_cycles = narrow_rows(Cout) ? 7000.0 : 13500.0;
        const double block_naps = (double)(Cin / ps.k) * chunk_cycles / 8128.0;
        p.stagger_n = g_stagger_n4;
        p.stagger_unit = g_stagger_n4 > 1 ? (int)(block_naps * g_stagger_pct4 / 100.0 / g_stagger_n4 + 0.5) : 0;
        if (p.stagger_n > 1 && p.stagger_unit < 1) p.stagger_unit = 1;
    }
    if (g_ymap4 < 0) {
        const char *e = getenv("DREAM_W4_YMAP");
        g_ymap4 = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return 0;
}

}  // namespace

// y = conv3x3(x, pad 1) * scale + shift (+ residual | ReLU mask) (ReLU) (2x2 max-pool), all NHWC fp32, by F(4x4,3x3).
// Cin a multiple of 32 (of 16 for Cout <= 64); flags: DREAM_CONV_RELU, DREAM_CONV_POOL2 (output [B, H/2, W/2, Cout], floor),
// DREAM_CONV_RELUMASK.
extern "C" int dream_conv3x3_winograd4_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                                const float *residual, float *y, int B, int H, int W, int Cin, int Cout, int flags,
                                                void *stream) {
    DREAM_REQUIRE((flags & ~(DREAM_CONV_RELU | DREAM_CONV_POOL2 | DREAM_CONV_RELUMASK | DREAM_CONV_RES_AFTER_RELU)) == 0, "winograd F(4x4) conv: unsupported flags 0x%x", flags);
    DREAM_REQUIRE(!(flags & DREAM_CONV_RES_AFTER_RELU) || (residual != nullptr && !(flags & (DREAM_CONV_POOL2 | DREAM_CONV_RELUMASK))),
                  "winograd F(4x4) conv: residual-after-ReLU needs a residual and excludes the fused pool / the ReLU mask");
    DREAM_REQUIRE(!(flags & DREAM_CONV_POOL2) || (residual == nullptr && H >= 2 && W >= 2), "winograd F(4x4) conv: fused max-pool takes no residual");
    DREAM_REQUIRE(!(flags & DREAM_CONV_RELUMASK) || residual != nullptr, "winograd F(4x4) conv: ReLU mask without a mask tensor");
    Wino4Params p;
    if (int rc = wino4_setup(p, x, u_packed, scale, shift, residual, y, B, H, W, Cin, Cout, flags, 1)) return rc;
    const int mode = (flags & DREAM_CONV_POOL2) ? 1 : (flags & DREAM_CONV_RELUMASK) ? 3 : (residual != nullptr ? 2 : 0);
    return narrow_rows(Cout) ? launch_wino4_mode<true>(p, mode, stream) : launch_wino4_mode<false>(p, mode, stream);
}

// Training forward of a conv that feeds nn.MaxPool2d(2) (dream/models.py:589,765-771): y_full = relu(conv3x3(x) * scale + shift)
// [B,H,W,Cout] AND y_pool = maxpool2x2(y_full) [B,H/2,W/2,Cout] from one launch -- the pooled maxima are taken from the registers the
// un-pooled values are stored from, so the stand-alone max-pool pass (a read of y_full) disappears.  Same values as the two launches, bit for bit.
extern "C" int dream_conv3x3_winograd4_pool_both_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                                          float *y_full, float *y_pool, int B, int H, int W, int Cin, int Cout, int flags,
                                                          void *stream) {
    DREAM_REQUIRE((flags & ~DREAM_CONV_RELU) == 0, "winograd F(4x4) conv + pool: unsupported flags 0x%x", flags);
    DREAM_REQUIRE(y_full != nullptr && y_pool != nullptr && H >= 2 && W >= 2, "winograd F(4x4) conv + pool: null output or map too small");
    Wino4Params p;
    if (int rc = wino4_setup(p, x, u_packed, scale, shift, nullptr, y_pool, B, H, W, Cin, Cout, flags, 1)) return rc;
    p.y_full = y_full;
    return narrow_rows(Cout) ? launch_wino4_mode<true>(p, 4, stream) : launch_wino4_mode<false>(p, 4, stream);
}

// nn.ConvTranspose2d(k4, s2, p1) (+ folded BatchNorm / bias, ReLU) by minimal filtering on the F(4x4) kernel: every output phase
// (a, b) is a 2 x 2-tap stride-1 conv of x; written as a 3x3 conv with a zero-padded kernel (dream_convT4x4_phase_weights) its
// transformed weights vanish on eleven of the 36 positions, so the kernel runs it with 25 multiplications per 4 x 4 outputs of
// the phase -- F(4x4, 2x2) -- where the F(2x2) kernel takes 36 (dream_conv_transpose4x4s2_winograd_nhwc_f32) and the direct
// sub-pixel form 64.  x [B,H,W,Cin] -> y [B,2H,2W,Cout]; four launches (one per phase); more than 64 output channels, Cin a
// multiple of 32.  u4: dream_pack_convT4x4_winograd4_weight(): 4 x dream_conv3x3_winograd4_weight_floats(Cout, Cin) floats.
// flags: DREAM_CONV_RELU.
extern "C" size_t dream_convT4x4_winograd4_weight_floats(int Cout, int Cin) { return 4 * dream_conv3x3_winograd4_weight_floats(Cout, Cin); }

// wT [Cin][Cout][4][4] -> u4; scratch: 4 * Cout * Cin * 9 floats (the four zero-padded 3x3 kernels).  mode 0: forward operator (u4:
// dream_convT4x4_winograd4_weight_floats(Cout, Cin)); mode 1: data-gradient operator (dream_conv4x4s2_winograd4_nhwc_f32; u4:
// dream_convT4x4_winograd4_weight_floats(Cin, Cout))
extern "C" int dream_pack_convT4x4_winograd4_weight(const float *wT, float *u4, float *scratch, int Cin, int Cout, int mode, void *stream) {
    DREAM_REQUIRE(wT && u4 && scratch && Cin > 0 && Cout > 0 && (mode == 0 || mode == 1), "winograd F(4x4) convT pack: bad arguments");
    if (int rc = dream_convT4x4_phase_weights(wT, scratch, Cin, Cout, mode, stream)) return rc;
    const int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;       // conv output / input channels
    const size_t per_u = dream_conv3x3_winograd4_weight_floats(rows, cols);
    for (int ph = 0; ph < 4; ++ph)
        if (int rc = dream_pack_conv3x3_winograd4_weight(scratch + (size_t)ph * Cout * Cin * 9, u4 + ph * per_u, rows, cols, 0, stream)) return rc;
    return 0;
}

// Data gradient of the transposed conv = a 4x4 stride-2 pad-1 conv of dY [B,2H,2W,Cout] -> dX [B,H,W,Cin], as the sum over the four
// output phases of a 2 x 2-tap conv on the phase's stride-2 view of dY: the same 25-position scheme (patterns mirrored), the phases
// accumulate into dX through the residual input of the epilogue.  Cin > 64, Cout a multiple of 32.  u4: mode 1 of the pack above.
extern "C" int dream_conv4x4s2_winograd4_nhwc_f32(const float *dy, const float *u4, float *dx, int B, int H, int W, int Cout, int Cin,
                                                  void *stream) {
    DREAM_REQUIRE(Cin > 64, "winograd F(4x4) conv4x4s2: needs more than 64 output channels (the wide workgroup shape), got %d", Cin);
    Wino4Params p;
    if (int rc = wino4_setup(p, dy, u4, nullptr, nullptr, nullptr, dx, B, H, W, Cout, Cin, 0, 1, 2)) return rc;
    const size_t per_u = dream_conv3x3_winograd4_weight_floats(Cin, Cout);
    for (int ph = 0; ph < 4; ++ph) {
        p.u = u4 + ph * per_u;
        p.in_oy = ph >> 1; p.in_ox = ph & 1;
        p.residual = ph == 0 ? nullptr : dx;         // phases 1..3 add to what is there (same thread reads and writes an element)
        int rc;
        // the data gradient's kernels have their taps in the opposite corner: pattern of phase (1 - a, 1 - b)
        switch (ph) {
            case 0: rc = launch_wino4<0, false, 4>(p, stream); break;
            case 1: rc = launch_wino4<2, false, 3>(p, stream); break;
            case 2: rc = launch_wino4<2, false, 2>(p, stream); break;
            default: rc = launch_wino4<2, false, 1>(p, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" int dream_conv_transpose4x4s2_winograd4_nhwc_f32(const float *x, const float *u4, const float *scale, const float *shift,
                                                            float *y, int B, int H, int W, int Cin, int Cout, int flags, void *stream) {
    DREAM_REQUIRE((flags & ~DREAM_CONV_RELU) == 0, "winograd F(4x4) convT: unsupported flags 0x%x", flags);
    DREAM_REQUIRE(Cout > 64, "winograd F(4x4) convT: needs more than 64 output channels (the wide workgroup shape), got %d", Cout);
    Wino4Params p;
    if (int rc = wino4_setup(p, x, u4, scale, shift, nullptr, y, B, H, W, Cin, Cout, flags, 2)) return rc;
    const size_t per_u = dream_conv3x3_winograd4_weight_floats(Cout, Cin);
    for (int ph = 0; ph < 4; ++ph) {
        p.u = u4 + ph * per_u;
        p.out_oy = ph >> 1; p.out_ox = ph & 1;
        int rc;
        switch (ph) {
            case 0: rc = launch_wino4<0, false, 1>(p, stream); break;
            case 1: rc = launch_wino4<0, false, 2>(p, stream); break;
            case 2: rc = launch_wino4<0, false, 3>(p, stream); break;
            default: rc = launch_wino4<0, false, 4>(p, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

// 3x3 stride-1 pad-1 convolution on NHWC fp32 tensors by the Winograd minimal-filtering algorithm F(2x2, 3x3)
// (Lavin & Gray 2016) on the CDNA4 fp32 matrix cores: 16 multiplications per 2x2 output tile and input channel instead of
// the direct algorithm's 36, i.e. 2.25x fewer MFMA cycles for the same fp32 result up to round-off (every product and sum is
// IEEE fp32; the transforms only add / subtract, the weight transform is done once, in fp64, at pack time).
//
// The same kernel runs nn.ConvTranspose2d(k4,s2,p1) (the ResNet decoder, /root/reference/dream/models.py:37-136, and -- with
// the equivalent weights -- the convs that follow nn.Upsample(2) in the VGG decoder, :691-710) and its data gradient by minimal
// filtering: an output phase is a 3x3 conv with only 2 x 2 non-zero taps, whose transformed weights vanish on seven of the 16
// positions (template parameter PAT below): 9 multiplications per 2 x 2 outputs of a phase instead of 16.
//
// Replaces torch.nn.Conv2d(k=3,s=1,p=1) (+ReLU, + the following MaxPool2d(2)) at /root/reference/dream/models.py:598-615
// (VGG19 encoder), :695-710 (upsample decoder, the convs not preceded by an upsample), :736-747 (head), the 3x3 stride-1
// convs of the ResNet-101 bottlenecks behind :22-32 in evaluation mode (folded BatchNorm), and -- on mode-1 packed
// weights -- their data gradients.  The direct kernel (conv_mfma.hip) stays the reference form and runs everything
// Winograd does not cover (strides, 1x1, transposed convs, NCHW store of the last head conv).
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// GEMM view: for each of the 16 positions p of the transformed 4x4 domain,  M_p[tile][cout] = sum_cin V_p[tile][cin] U_p[cin][cout].
//   * tiles are numbered over (image, tile row, tile column) and a workgroup takes 32 consecutive ones, so every map size
//     (400 .. 13 pixels) fills its workgroups (no 2-D tile padding);
//   * workgroup = NW wavefronts (4 or 8) = 32 tiles x 16 NW output channels; wavefront = 32 tiles x 16 channels x all 16
//     positions: 2 x 16 accumulators of v_mfma_f32_16x16x4_f32 (128 VGPRs), so the inverse transform is lane-local.
//     NW = 8 (128 channels, one workgroup per CU) halves the input-transform work per MFMA; NW = 4 (two workgroups per CU)
//     serves the 64-channel layers;
//   * V (the transformed input) is computed by the workgroup for one 16-channel chunk at a time into a double-buffered LDS
//     tile (2 x 32 KB; float4 slots XOR-swizzled and plane groups skewed by 32 bytes so that the b128 transform stores and the
//     b128 MFMA-operand reads are both conflict-free) WHILE the MFMAs of the previous chunk run: a thread owns (tile,
//     channel quad, patch row), loads the row's four pixels once, transforms along the row, gets the other rows' values
//     for the column transform from its quad neighbours (DPP quad_perm) -- no element of the input is loaded twice;
//   * U (the transformed weights, packed [Cin/16][16][CoutPad][16]) never touches LDS: every wavefront streams its own
//     16-channel operand rows straight from L2 into a ring of eight registers, 1 KB coalesced per position, six positions
//     ahead (loads return in order: an operand must never queue behind a patch load that misses to HBM);
//   * all global traffic goes through buffer descriptors: 32-bit offsets, zero padding and store masking by the hardware's
//     bounds check (offset BUFFER_OOB), no 64-bit address arithmetic; tile -> (image, row, column) by magic multiplication;
//   * the instruction order of the main loop is pinned (sched_barrier per pair of MFMAs): one patch load per pair, MFMA
//     operands read one position ahead into the register set of the other parity, alternating accumulators;
//   * workgroups are PERSISTENT: the grid is what the chip holds at once, each workgroup walks over the tile blocks of its
//     XCD's range and loads + transforms the first chunk of its next block during the last chunk of the current one, so the
//     load latency of a prologue is paid once per workgroup, not once per block (measured: +5 .. +30 % per layer, most on
//     the layers with few input channels); the epilogue variant (plain / fused pool / residual / ReLU mask) is a template
//     parameter, so the code between two blocks' MFMA phases is straight-line (exact s_waitcnt counts, its 16 mask or
//     residual loads per tile group in flight together).
#include <type_traits>
#include <dream_cdna4.h>
#include "common.h"
#include "pack_device.h"
#include "../../include/dream_hip.h"

// Timing diagnostics only (tools/wino_diag.py builds separate libraries with -DDREAM_WINO_DIAG=k; never the product
// library): bit 0 skips the input transform (loads + V stores), bit 1 the weight stream, bit 2 the per-chunk barrier,
// bit 3 makes all 32 tiles read the first tile's patch, bit 4 makes every chunk read the first chunk's channels, bit 6 swaps
// the two instruction orders of the main loop (results stay correct).
// Results are then wrong by construction; the point is what each part costs.
#ifndef DREAM_WINO_DIAG
#define DREAM_WINO_DIAG 0
#endif

namespace {

struct WinoParams {
    const float *x;          // [B,H,W,Cin]
    const float *u;          // [Cin/16][16][CoutPad][16] (+ B_AHEAD zero positions)
    const float *scale;      // per-channel multiplier (eval-mode BatchNorm fold) or null
    const float *shift;      // per-channel addend (bias / BN shift) or null
    const float *residual;   // ReLU mask source (DREAM_CONV_RELUMASK) or addend of the output's shape, or null
    float *y;                // [B,H,W,Cout]  (or [B,H/2,W/2,Cout] with DREAM_CONV_POOL2)
    int B, H, W, Cin, Cout, CoutPad;
    int TY, TX;              // 2x2 tiles per image
    int ntiles;              // B * TY * TX  (< 2^24)
    int nblk, blk_per_xcd;   // blocks of 32 tiles; each XCD's workgroups walk over a contiguous range of them
    int out_scale, out_oy, out_ox;   // output pixel of conv position (y, x): (out_scale y + out_oy, out_scale x + out_ox) -- (1,0,0)
                                     // for a conv, (2, a, b) for phase (a, b) of a stride-2 transposed conv
    int in_scale, in_oy, in_ox;      // stored input pixel of conv position (y, x), likewise: (2, a, b) for the phase views of the
                                     // gradient in the transposed conv's data gradient; the stored tensor is in_scale H x in_scale W
    unsigned long long magic_tpi, magic_tx;   // ceil(2^40 / (TY * TX)), ceil(2^40 / TX): tile -> (image, row, column) without divides
    int flags;
};

constexpr int WT = 32;       // tiles per workgroup
constexpr int WKC = 16;      // input channels per chunk
constexpr int WPAD = 128;    // output channels the packed weights are padded to (a multiple of every workgroup width)
// float offset of V plane p = 4i + j inside a buffer: 512 floats per plane plus a 32-byte skew per row group i, so that the
// b128 transform stores of a quad's four lanes (planes 4r + j, r = 0..3, same tile) fall on banks 0 / 8 / 16 / 24 of the 32
// the LDS store path distinguishes (MI355X_MICROARCH.md: ds_write_b128 is serviced 8 lanes at a time, bank = (a/4) mod 32)
DREAM_DEVICE constexpr int v_plane(int p) { return p * (WT * WKC) + 8 * (p >> 2); }
constexpr int VB = 16 * WT * WKC + 32;   // floats per V buffer
constexpr int B_RING = 8;    // operand registers of the weight stream
constexpr int B_AHEAD = 6;   // positions the weight stream runs ahead of the MFMAs: far enough that an operand is never queued
                             // behind the patch loads of an item (loads return in order)

// physical float4 slot of logical slot q (k = 4q .. 4q+3) in row t of a V plane
DREAM_DEVICE int v_slot(int q, int t) { return q ^ ((t >> 2) & 2); }

// MODE: what the epilogue does besides scale / shift / ReLU -- 0 nothing, 1 fused 2x2 max-pool, 2 residual add, 3 ReLU mask
// (data gradient through a ReLU: zero where the forward activation was).  Compile-time: the epilogue of a persistent
// workgroup sits between two blocks' MFMA phases and must be straight-line code (exact s_waitcnt counts, loads batched).
// PAT: which of the 16 positions carry non-zero transformed weights.  0: all (3x3 conv).  1 + 2 a + b: phase (a, b) of
// nn.ConvTranspose2d(k4, s2, p1) written as a 3x3 conv whose kernel has only 2 x 2 non-zero taps (rows {0,1} for a = 0, {1,2} for
// a = 1; columns likewise): G g G^T then vanishes on row 3 (a = 0) / row 0 (a = 1) of the 4 x 4 domain, and likewise on a
// column -- nine positions are left, their MFMAs are the F(2x2,2x2) minimal-filtering count (9 instead of 16 per 2 x 2
// outputs of a phase), and the other seven are never loaded or multiplied.
DREAM_DEVICE constexpr bool pat_row_active(int pat, int i) { return pat == 0 || (((pat - 1) >> 1) == 0 ? i <= 2 : i >= 1); }
DREAM_DEVICE constexpr bool pat_col_active(int pat, int j) { return pat == 0 || (((pat - 1) & 1) == 0 ? j <= 2 : j >= 1); }
DREAM_DEVICE constexpr bool pat_active(int pat, int pp) { return pat_row_active(pat, pp >> 2) && pat_col_active(pat, pp & 3); }
DREAM_DEVICE constexpr int pat_count(int pat) { return pat == 0 ? 16 : 9; }
DREAM_DEVICE constexpr int pat_pos(int pat, int k) {          // k-th active position
    int n = 0;
    for (int pp = 0; pp < 16; ++pp)
        if (pat_active(pat, pp)) {
            if (n == k) return pp;
            ++n;
        }
    return 0;
}

template <int NW, int MODE, int PAT>
__global__ void __launch_bounds__(64 * NW, 2) conv_wino_kernel(const WinoParams p) {
    constexpr int NT = 64 * NW;                        // threads
    constexpr int NPOS = pat_count(PAT);               // positions this kernel multiplies
    constexpr int RING = NPOS == 16 ? B_RING : NPOS;   // operand ring of the weight stream: a divisor of NPOS (slots carry over chunks)
    constexpr int ITEMS = 512 / NT;                    // (tile, quad, row) items per thread and chunk: 2 (NW 4) or 1 (NW 8)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    DREAM_DYNAMIC_LDS(float, sV);                      // 2 x V buffer (16 skewed planes of [32 tiles][16 channels]), then the offset table
    u32x4 *sG = (u32x4 *)(sV + 2 * VB);                // [ITEMS][NT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_index();

    // PERSISTENT workgroups: the grid is what the chip holds at once (host side); a workgroup walks over tile blocks
    // tb, tb + J, tb + 2J .. of its XCD's contiguous range (XCD-aware placement, see conv_mfma.hip: workgroup g runs on XCD
    // g % 8; the J workgroups of an XCD work on J neighbouring blocks at any time and meet in its L2).  What this buys: the
    // first chunk of the NEXT block is loaded and transformed during the last chunk of the current one, so only the very
    // first block of a workgroup pays the load latency of a prologue (~2.5 us of a 14-56 us block).
    const int xcd = (int)(blockIdx.x & 7), J = (int)(gridDim.x >> 3);
    const int blk_hi = (xcd + 1) * p.blk_per_xcd;
    const int blk_end = blk_hi < p.nblk ? blk_hi : p.nblk;
    int tb = xcd * p.blk_per_xcd + (int)(blockIdx.x >> 3);
    if (tb >= blk_end) return;
    const int n0 = blockIdx.y * (16 * NW);
    const int tiles_per_img = p.TY * p.TX;
    const size_t img_floats = (size_t)(p.in_scale * p.H) * (p.in_scale * p.W) * p.Cin;     // stored input image

    // ---- input-transform plan: thread -> ITEMS items (tile t, channel quad q, patch row r); r = the lane's index in its quad
    // Loads go through buffer descriptors (dream_cdna4.h): a 32-bit byte offset per (item, column) relative to the first image
    // the BLOCK touches, BUFFER_OOB where the patch leaves the image (the hardware returns zeros: no compare / select per
    // load), the chunk's channel offset in the scalar operand.  The offsets are parked in LDS ([ITEMS][NT] uint4, each entry
    // private to its thread, conflict-free b128 access) and re-read per chunk: VGPRs that accumulators + weight ring + patch
    // cannot spare.
    int soff[ITEMS];                                   // LDS float offset of V[p = 4r][t][slot q]; + j * 512 for p = 4r + j
    const int qr = lane & 3;                           // patch row of this lane's items (e & 3 with NT a multiple of 4)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int e = tid + it * NT;
        const int q = (e >> 2) & 3, t = (e >> 4) & 31;
        soff[it] = v_plane(4 * qr) + t * WKC + 4 * v_slot(q, t);
    }
    auto plan_item = [&](int it, int tile0, int b0) -> u32x4 {
        const int e = tid + it * NT;
        const int q = (e >> 2) & 3, t = (e >> 4) & 31;
        const int tau = (DREAM_WINO_DIAG & 8) ? tile0 : tile0 + t;
        const bool tv = tau < p.ntiles;
        const int b = div_magic40(tau, p.magic_tpi), rem = tau - b * tiles_per_img;
        const int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
        const int gy = 2 * ty - 1 + qr, x0 = 2 * tx - 1;
        const bool rok = tv & ((unsigned)gy < (unsigned)p.H);          // bitwise: no short-circuit branches inside a chunk
        const int Si = p.in_scale;                                    // conv position -> stored pixel (WinoParams)
        const unsigned off0 = (unsigned)(((((b - b0) * (Si * p.H) + Si * gy + p.in_oy) * (Si * p.W) + Si * x0 + p.in_ox) * p.Cin + 4 * q) * 4);   // column 0 (wraps when outside)
        const unsigned px = (unsigned)(Si * p.Cin * 4);
        u32x4 g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = rok & ((unsigned)(x0 + c) < (unsigned)p.W);
            g[c] = ok ? off0 + (unsigned)c * px : BUFFER_OOB;
        }
        sG[it * NT + tid] = g;
        return g;
    };
    // first tile / first image / input descriptor of a block; a block past the end of the range reads nothing (all OOB)
    auto block_tile0 = [&](int blk) { return blk < blk_end ? blk * WT : p.ntiles; };
    auto block_xbuf = [&](int b0) {
        return make_buffer(p.x + (size_t)b0 * img_floats, ((size_t)(p.B - b0) * img_floats) * sizeof(float));
    };
    // column transform across the quad: row r of B^T (u_0..u_3) = u_r + sb * u_partner(r), partner = {2, 2, 1, 1}, for
    // r = 0, 1, 2; the lane of r = 3 computes u_3 - u_1 = MINUS row 3 -- the packed weights carry the matching sign in their
    // positions 12..15 (wino_pack_kernel), so the product is unchanged and every lane needs one fma per value.
    const float sb = (qr == 1) ? 1.0f : -1.0f;

    // ---- MFMA operand addresses ------------------------------------------------------------------------------------------
    // A: lane l -> tile row (l & 15) (+16 for the second block), k = 4 (l >> 4) .. +3 (one float4, feeds 4 MFMAs)
    const int lt = lane & 15, lg = lane >> 4;
    int a_off[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int t = blk * 16 + lt;
        a_off[blk] = t * WKC + 4 * v_slot(lg, t);
    }
    // B: lane l -> output channel n0 + 16 wave + (l & 15), k = 4 (l >> 4) .. +3; position s at scalar offset s * stride
    const unsigned b_lane = (unsigned)(((wave * 16 + lt) * WKC + 4 * lg) * 4);
    const unsigned u_pos_stride = (unsigned)(p.CoutPad * WKC * 4);          // bytes between consecutive positions
    const BufferRsrc ubuf = make_buffer(p.u + (size_t)n0 * WKC, ((size_t)((p.Cin / WKC) * 16 + B_AHEAD) * p.CoutPad - (size_t)n0) * WKC * sizeof(float));

    f32x4 acc[16][2];
    const int nchunks = p.Cin / WKC;

    // weight stream: ring of B_RING operand registers, position s lives in bq[s % B_RING] (16 positions per chunk: the
    // register depends on the position inside the chunk only).  Unconditional and branch-free, so the compiler can count
    // outstanding loads exactly (s_waitcnt vmcnt(N) instead of vmcnt(0) at merge points); in the last chunk of a block the
    // stream wraps around to the first positions of the next block (same weights).
    f32x4 bq[RING];
#pragma unroll
    for (int k = 0; k < B_AHEAD; ++k) bq[k] = buffer_load_x4(ubuf, b_lane, (unsigned)pat_pos(PAT, k) * u_pos_stride);

    // patch row of an item: its four loads / row transform, quad exchange, column transform, four V stores
    f32x4 d[ITEMS][4];
    u32x4 goff[ITEMS];
    // piece j (0..3) of an item's transform: column j of the row transform u = d B (B^T d B = B^T (d B)), the quad exchange
    // for the column transform, one b128 store -- ~8 VALU instructions, placed after ONE pair of MFMAs each, so that the
    // wave's next MFMA is never more than the other wave's MFMA time away
    auto item_piece = [&](int it, int j, float *vbuf) {
        const f32x4 u = j == 0 ? d[it][0] - d[it][2] : j == 1 ? d[it][1] + d[it][2] : j == 2 ? d[it][2] - d[it][1] : d[it][1] - d[it][3];
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(sb, quad_perm_2211(u[k]), u[k]);     // exact: sb = +-1
        *(f32x4 *)(vbuf + soff[it] + j * (WT * WKC)) = v;
    };
    auto item_store = [&](int it, float *vbuf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) item_piece(it, j, vbuf);
    };

    // MFMA operands of position pp from V buffer vbuf: read one position ahead of their use into the register set of the
    // other parity (no copies).  The schedule is pinned in groups of two MFMAs (sched_barrier): at ~200 VGPRs hipcc
    // schedules for register pressure -- it sinks every load to its use (a full LDS / memory latency with no MFMA of this
    // wave in flight) and issues the four MFMAs of one accumulator back to back (40-cycle dependent latency vs 32 issue).
    f32x4 a[2][2];
    auto read_a = [&](int set, int pp, const float *vbuf) {
        a[set][0] = *(const f32x4 *)(vbuf + v_plane(pp) + a_off[0]);
        a[set][1] = *(const f32x4 *)(vbuf + v_plane(pp) + a_off[1]);
    };

    // ---- first block of this workgroup: plan, chunk 0 into buffer 0 (the only exposed load latency of the workgroup) --------
    int tile0 = block_tile0(tb);
    int b0 = div_magic40(tile0, p.magic_tpi);
    BufferRsrc xbuf = block_xbuf(b0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        goff[it] = plan_item(it, tile0, b0);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[it][c] = buffer_load_x4(xbuf, goff[it][c], 0u);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) item_store(it, sV);
    __syncthreads();
    int par = 0;                                       // V buffer holding the chunk about to be multiplied

    // One chunk: 16 positions x (2 tile blocks x 4 k-steps) MFMAs on V buffer `par`, while the NEXT chunk is loaded,
    // transformed and stored into the other buffer.  Patch loads: ONE per pair of MFMAs (a load touches 16 separate 64-byte
    // segments and the texture-address unit takes tens of cycles to accept it; issued back to back they stall the wave's
    // in-order instruction stream, MFMAs included); item it loads during position it, is transformed and stored ten
    // positions later.  FIRST: the block's first chunk starts the accumulators from zero (no clearing pass).  LAST: the
    // next chunk is chunk 0 of the NEXT block -- its plan is computed here (a few dozen VALU instructions per item) and
    // kept in the offset table, the weight stream wraps around.
    auto chunk = [&](auto first_tag, auto last_tag, int c, const BufferRsrc &xnext, int tile0n, int b0n) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const float *cur = sV + par * VB;
        float *nxt = sV + (par ^ 1) * VB;
        const unsigned coff = LAST ? 0u : (unsigned)((c + 1) * WKC * 4);
        read_a(0, pat_pos(PAT, 0), cur);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NPOS; ++k) {                                    // k-th active position pp
            constexpr int SP = NPOS == 16 ? 10 : 5;                         // item it is transformed during slot SP + it
            const int pp = pat_pos(PAT, k);
            const bool tf = !(DREAM_WINO_DIAG & 1);
            const int li = (k < ITEMS) ? k : -1;                            // item loaded during this slot
            const int si = (k >= SP && k < SP + ITEMS) ? k - SP : -1;
            auto load_b = [&]() {
                if (DREAM_WINO_DIAG & 2) return;
                const int kn = k + B_AHEAD;                                 // slot the load is for: this chunk's, or the next one's
                const int s = kn >= NPOS ? (LAST ? 0 : (c + 1) * 16) + pat_pos(PAT, kn - NPOS) : c * 16 + pat_pos(PAT, kn);
                bq[kn % RING] = buffer_load_x4(ubuf, b_lane, (unsigned)s * u_pos_stride);
            };
            auto load_one = [&](int col) {
                d[li][col] = buffer_load_x4(LAST ? xnext : xbuf, goff[li][col], (DREAM_WINO_DIAG & 16) ? 0u : coff);
            };
            auto pair = [&](int r) {
                if (FIRST && r == 0) {
                    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                    acc[pp][0] = mfma_f32_16x16x4(a[k & 1][0][r], bq[k % RING][r], zero);
                    acc[pp][1] = mfma_f32_16x16x4(a[k & 1][1][r], bq[k % RING][r], zero);
                } else {
                    acc[pp][0] = mfma_f32_16x16x4(a[k & 1][0][r], bq[k % RING][r], acc[pp][0]);
                    acc[pp][1] = mfma_f32_16x16x4(a[k & 1][1][r], bq[k % RING][r], acc[pp][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // Two instruction orders, chosen by measurement (profiles/r02_wino_diag.txt; DIAG bit 6 swaps them): the 4-wave
            // kernel (two items per thread, two workgroups per CU) is ~3 % faster with its loads ahead of the position's first
            // MFMAs and the transform in one piece, the 8-wave kernel ~2 % faster the other way round.
            if ((NW == 4) != ((DREAM_WINO_DIAG & 64) != 0)) {
                load_b();
                if (tf && li >= 0) {
                    if (LAST) goff[li] = plan_item(li, tile0n, b0n);
                    else goff[li] = sG[li * NT + tid];
                    load_one(0);
                }
                pair(0);
                if (k + 1 < NPOS) read_a((k + 1) & 1, pat_pos(PAT, k + 1), cur);
                if (tf && li >= 0) load_one(1);
                pair(1);
                if (tf && li >= 0) load_one(2);
                if (tf && si >= 0) item_store(si, nxt);
                pair(2);
                if (tf && li >= 0) load_one(3);
                pair(3);
            } else {
                // every position STARTS with MFMAs (operands were read during the previous position); everything else --
                // the weight load, the operand reads of the next position, one patch load or one transform piece -- follows
                // a pair, never more than one memory instruction and ~8 VALU instructions between two pairs
                pair(0);
                load_b();
                if (k + 1 < NPOS) read_a((k + 1) & 1, pat_pos(PAT, k + 1), cur);
                if (tf && li >= 0) {
                    if (LAST) goff[li] = plan_item(li, tile0n, b0n);
                    else goff[li] = sG[li * NT + tid];                      // written by this thread: no barrier
                    load_one(0);
                }
                if (tf && si >= 0) item_piece(si, 0, nxt);
                pair(1);
                if (tf && li >= 0) load_one(1);
                if (tf && si >= 0) item_piece(si, 1, nxt);
                pair(2);
                if (tf && li >= 0) load_one(2);
                if (tf && si >= 0) item_piece(si, 2, nxt);
                pair(3);
                if (tf && li >= 0) load_one(3);
                if (tf && si >= 0) item_piece(si, 3, nxt);
                if (k == NPOS - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(DREAM_WINO_DIAG & 4)) __syncthreads();
        par ^= 1;
    };
    const std::true_type yes{};
    const std::false_type no{};

    // ---- inverse transform Y = A^T M A (lane-local), scale / shift / residual / ReLU / 2x2 max-pool, store ----------------
    // Output addresses are 32-bit byte offsets from the first image of the block, BUFFER_OOB for everything that must not be
    // written (tile beyond the batch, channel beyond Cout, odd-extent overhang): masked stores without branches.
    const bool relu = (p.flags & DREAM_CONV_RELU) != 0;
    constexpr bool pool = MODE == 1, has_res = MODE >= 2, mask = MODE == 3;
    const bool late = MODE == 2 && (p.flags & DREAM_CONV_RES_AFTER_RELU) != 0;      // the residual is a skip connection: added after the ReLU
    const int col = n0 + wave * 16 + lt;
    const bool cok = col < p.Cout;
    const float sc = (p.scale != nullptr && cok) ? p.scale[col] : 1.0f;
    const float sh = (p.shift != nullptr && cok) ? p.shift[col] : 0.0f;
    const int Ho = pool ? p.H / 2 : p.H, Wo = pool ? p.W / 2 : p.W;           // grid of conv positions that are stored
    const int S = pool ? 1 : p.out_scale;                                     // conv position -> output pixel (see WinoParams)
    const int Wy = S * Wo;
    const size_t out_img = (size_t)(S * Ho) * Wy * p.Cout;
    const unsigned px_b = (unsigned)(S * p.Cout * 4), row_b = (unsigned)(S * Wy * p.Cout * 4);
    const unsigned phase_b = pool ? 0u : (unsigned)((p.out_oy * Wy + p.out_ox) * p.Cout * 4);
    auto epilogue = [&](int tile0e, int b0e) {
        const BufferRsrc ybuf = make_buffer(p.y + (size_t)b0e * out_img, (size_t)(p.B - b0e) * out_img * sizeof(float));
        const BufferRsrc rbuf = make_buffer(has_res ? p.residual + (size_t)b0e * out_img : p.y,
                                            has_res ? (size_t)(p.B - b0e) * out_img * sizeof(float) : 0);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            // the lane's four tiles of this block are consecutive: decompose the first, step the others
            const int tau0 = tile0e + blk * 16 + lg * 4;                // C/D layout: row = 4 (l >> 4) + reg, col = l & 15
            int b = div_magic40(tau0, p.magic_tpi);
            const int rem = tau0 - b * tiles_per_img;
            int ty = div_magic40(rem, p.magic_tx), tx = rem - ty * p.TX;
            // per tile: the byte offsets of its 2x2 outputs (or of its pooled output), BUFFER_OOB where nothing may be written
            auto tile_offsets = [&](int r, unsigned *o) {
                // pooled output (floor(H/2) x floor(W/2)): the window of a tile with ty < Ho, tx < Wo lies entirely inside the image
                const bool tok = cok & ((tau0 + r) < p.ntiles) & (!pool | ((ty < Ho) & (tx < Wo)));
                const int oy = pool ? ty : 2 * ty, ox = pool ? tx : 2 * tx;
                const unsigned base = (unsigned)(((((b - b0e) * (S * Ho) + S * oy) * Wy + S * ox) * p.Cout + col) * 4) + phase_b;
                if (pool) {
                    o[0] = tok ? base : BUFFER_OOB;
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const bool inb = tok & ((oy + i) < Ho) & ((ox + jj) < Wo);
                            o[2 * i + jj] = inb ? base + i * row_b + jj * px_b : BUFFER_OOB;
                        }
                }
                const bool wrap_x = (tx + 1 == p.TX);                   // step to the next tile
                const bool wrap_y = wrap_x & (ty + 1 == p.TY);
                tx = wrap_x ? 0 : tx + 1;
                ty = wrap_y ? 0 : (wrap_x ? ty + 1 : ty);
                b += wrap_y ? 1 : 0;
            };
            unsigned off[4][4];                                         // [tile][2 i + jj]
            float rv[4][4];
            if (has_res) {                                              // all 16 mask / residual loads of the block in flight
#pragma unroll                                                          // before the first inverse transform
                for (int r = 0; r < 4; ++r) {
                    tile_offsets(r, off[r]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[r][e] = buffer_load_f32(rbuf, off[r][e], 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!has_res) tile_offsets(r, off[r]);
                // positions without weights (PAT) were never multiplied: they are zeros of the sums
                auto M = [&](int pp) { return pat_active(PAT, pp) ? acc[pp][blk][r] : 0.0f; };
                float s[2][4];                                          // A^T M : rows [1,1,1,0], [0,1,-1,-1]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[0][j] = M(j) + M(4 + j) + M(8 + j);
                    s[1][j] = M(4 + j) - M(8 + j) - M(12 + j);
                }
                float out[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    out[i][0] = s[i][0] + s[i][1] + s[i][2];
                    out[i][1] = s[i][1] - s[i][2] - s[i][3];
                }
                float best = -__builtin_huge_valf();
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float v = out[i][jj] * sc + sh;                 // sc = 1 / sh = 0 when absent (exact)
                        if (has_res) v = mask ? (rv[r][2 * i + jj] > 0.0f ? v : 0.0f) : (late ? v : v + rv[r][2 * i + jj]);
                        const float vr = fmaxf(v, 0.0f);
                        v = relu ? vr : v;
                        if (MODE == 2) v = late ? v + rv[r][2 * i + jj] : v;
                        if (pool) best = fmaxf(best, v);
                        else buffer_store_f32(ybuf, v, off[r][2 * i + jj], 0);
                    }
                if (pool) buffer_store_f32(ybuf, best, off[r][0], 0);
            }
        }
    };

    // ---- the blocks of this workgroup -----------------------------------------------------------------------------------
    // Written out twice (first block, then the loop): at a control-flow join hipcc takes the more pessimistic of the incoming
    // s_waitcnt states, and the state after the prologue (a handful of loads in flight) differs from the one after an
    // epilogue (its stores still in flight); joined, every block would start by waiting for the previous block's stores.
    auto block = [&]() {
        const int tile0n = block_tile0(tb + J);
        const int b0n = div_magic40(tile0n, p.magic_tpi);
        const BufferRsrc xnext = block_xbuf(b0n);
        chunk(yes, no, 0, xnext, tile0n, b0n);                     // nchunks >= 2 (host side)
        for (int c = 1; c < nchunks - 1; ++c) chunk(no, no, c, xnext, tile0n, b0n);
        chunk(no, yes, nchunks - 1, xnext, tile0n, b0n);
        epilogue(tile0, b0);
        tb += J;
        tile0 = tile0n;
        b0 = b0n;
        xbuf = xnext;
    };
    block();
    while (tb < blk_end) block();
}

// OIHW (mode 0) or, for the data-gradient operator, IOHW with flipped taps (mode 1: rows = Cin_fwd, cols = Cout_fwd)
// -> U = G g G^T in fp64, rounded once to fp32, laid out [cols/16][16 positions][RowsPad][16]: pack_device.h (dream_pack::winograd2)
static_assert(WKC == 16 && WPAD == 128, "pack_device.h assumes 16-channel chunks and rows padded to 128");
__global__ void __launch_bounds__(256) wino_pack_kernel(const float *w, float *u, int Cout, int Cin, int mode) {
    dream_pack::winograd2(w, u, Cout, Cin, mode, (int)blockIdx.x, (int)gridDim.x);
}

constexpr int kCUs = 256;     // MI355X
int g_max_workgroups = 0;     // test hook: cap on resident workgroups (0 = the chip's capacity)

template <int NW, int MODE, int PAT = 0>
int launch_wino(const WinoParams &p, void *stream) {
    void (*kernel)(const WinoParams) = conv_wino_kernel<NW, MODE, PAT>;
    const size_t lds = (size_t)2 * VB * sizeof(float) + (size_t)512 * 16;          // V buffers + the offset table
    if (dream_allow_full_lds((const void *)kernel)) return 2;
    // persistent grid: as many workgroups as the 256 CUs hold at once (2 of the 4-wave, 1 of the 8-wave kind per CU), a
    // multiple of 8 (XCDs), split over the output-channel blocks; fewer when there are fewer tile blocks than that
    const int ny = (p.Cout + 16 * NW - 1) / (16 * NW);
    int resident = g_max_workgroups > 0 ? g_max_workgroups : kCUs * (NW == 4 ? 2 : 1);
    int gx = resident / ny / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > (p.nblk + 7) / 8 * 8) gx = (p.nblk + 7) / 8 * 8;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    hipLaunchKernelGGL(kernel, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

int g_variant = 0;

}  // namespace

extern "C" size_t dream_conv3x3_winograd_weight_floats(int rows, int cols) {
    const size_t rows_pad = (size_t)((rows + WPAD - 1) / WPAD) * WPAD;
    return ((size_t)(cols / WKC) * 16 + B_AHEAD) * rows_pad * WKC;      // + the zero positions the weight stream over-reads
}

// w: OIHW [Cout,Cin,3,3]; mode 0: forward operator (rows = Cout, cols = Cin); mode 1: data-gradient operator
// (rows = Cin, cols = Cout, taps flipped).  u: dream_conv3x3_winograd_weight_floats(rows, cols) floats.
extern "C" int dream_pack_conv3x3_winograd_weight(const float *w_oihw, float *u, int Cout, int Cin, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && u && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "winograd pack: bad arguments");
    const int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;
    DREAM_REQUIRE(cols % WKC == 0, "winograd pack: %d input channels, must be a multiple of %d", cols, WKC);
    const int rows_pad = (rows + WPAD - 1) / WPAD * WPAD;
    const size_t total = (size_t)(cols / WKC) * rows_pad * WKC;
    size_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_oihw, u, Cout, Cin, mode);
    DREAM_LAUNCH_OK();
    DREAM_HIP_OK(hipMemsetAsync(u + (size_t)(cols / WKC) * 16 * rows_pad * WKC, 0, (size_t)B_AHEAD * rows_pad * WKC * sizeof(float),
                                (hipStream_t)stream));
    return 0;
}

// Workgroup width: 0 (default) = by layer (128 output channels per workgroup when the layer has more than 64, else 64);
// 4 / 8 force the 64- / 128-channel kernel.  Same results bit for bit.
// Test hook: cap the number of co-resident workgroups the persistent grid is sized for (0 = the chip's 256 CUs), so that
// small problems walk over several tile blocks per workgroup.  Same results bit for bit.
extern "C" int dream_conv3x3_winograd_set_max_workgroups(int n) {
    DREAM_REQUIRE(n >= 0, "winograd: max workgroups %d", n);
    g_max_workgroups = n;
    return 0;
}

extern "C" int dream_conv3x3_winograd_set_variant(int variant) {
    DREAM_REQUIRE(variant == 0 || variant == 4 || variant == 8, "winograd variant %d: 0 (by layer), 4 or 8 wavefronts", variant);
    g_variant = variant;
    return 0;
}

namespace {

// geometry shared by the conv and the transposed-conv entry points
int wino_setup(WinoParams &p, const float *x, const float *u_packed, const float *scale, const float *shift, const float *residual,
               float *y, int B, int H, int W, int Cin, int Cout, int flags, int out_scale, int in_scale = 1) {
    DREAM_REQUIRE(x && u_packed && y, "winograd conv: null pointer");
    DREAM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "winograd conv: bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    DREAM_REQUIRE(Cin % WKC == 0 && Cin >= 2 * WKC, "winograd conv: Cin=%d must be a multiple of %d and at least %d", Cin, WKC, 2 * WKC);
    // 32-bit byte offsets relative to the first image a workgroup touches: its 32 tiles span at most this many images
    const size_t span_imgs = (size_t)WT / ((size_t)((H + 1) / 2) * ((W + 1) / 2)) + 2;
    const size_t out_px = (size_t)out_scale * out_scale * H * W;
    DREAM_REQUIRE(span_imgs * in_scale * in_scale * H * W * (size_t)Cin * sizeof(float) < ((size_t)1 << 31) && span_imgs * out_px * (size_t)Cout * sizeof(float) < ((size_t)1 << 31),
                  "winograd conv: image too large for 32-bit offsets");
    p.x = x; p.u = u_packed; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.CoutPad = (Cout + WPAD - 1) / WPAD * WPAD;
    DREAM_REQUIRE(((size_t)(Cin / WKC) * 16 + B_AHEAD) * (size_t)p.CoutPad * WKC * sizeof(float) < ((size_t)1 << 31), "winograd conv: weights too large");
    p.TY = (H + 1) / 2; p.TX = (W + 1) / 2;
    const long ntiles = (long)B * p.TY * p.TX;
    DREAM_REQUIRE(ntiles < ((long)1 << 24), "winograd conv: %ld tiles, the tile decomposition handles < 2^24", ntiles);
    p.ntiles = (int)ntiles;
    p.nblk = (p.ntiles + WT - 1) / WT;
    p.blk_per_xcd = (p.nblk + 7) / 8;
    p.magic_tpi = (((unsigned long long)1 << 40) + (unsigned long long)(p.TY * p.TX) - 1) / (unsigned long long)(p.TY * p.TX);
    p.magic_tx = (((unsigned long long)1 << 40) + (unsigned long long)p.TX - 1) / (unsigned long long)p.TX;
    p.flags = flags;
    p.out_scale = out_scale; p.out_oy = 0; p.out_ox = 0;
    p.in_scale = in_scale; p.in_oy = 0; p.in_ox = 0;
    return 0;
}

// [Cin][Cout][4][4] ConvTranspose2d(k4, s2, p1) weight -> four OIHW [Cout][Cin][3][3] conv kernels, one per output phase
// (a, b): output (2 i + a, 2 j + b) = sum over the 2 x 2 inputs (i + dy, j + dx), dy in {-1, 0} (a = 0) or {0, +1} (a = 1), of
// x * wT[a + 1 - 2 dy][b + 1 - 2 dx]; as a pad-1 3x3 correlation the tap (dy, dx) sits at [dy + 1][dx + 1], the rest is zero.
// bwd = 1: the kernels of the DATA GRADIENT instead -- dx(i, j) = sum over phases of a 2 x 2-tap correlation of the phase view
// P_ab(i, j) = dY(2 i + a, 2 j + b): dx(i) += P(i - dy) wT[a + 1 - 2 dy], i.e. tap r = 1 - dy of a pad-1 3x3 correlation carries
// wT[a - 1 + 2 r] (rows {1,2} for a = 0, {0,1} for a = 1); output channels = Cin of the transposed conv, input channels = Cout:
// w3[phase][ci][co][3][3].
__global__ void __launch_bounds__(256) convT4x4_phase_kernels(const float *wT, float *w3, int Cin, int Cout, int bwd) {
    const size_t per_phase = (size_t)Cout * Cin * 9, total = 4 * per_phase;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int tap = (int)(i % 9);
        size_t rest = i / 9;
        const int inner = (int)(rest % (bwd ? Cout : Cin));
        rest /= (bwd ? Cout : Cin);
        const int outer = (int)(rest % (bwd ? Cin : Cout)), phase = (int)(rest / (bwd ? Cin : Cout));
        const int ci = bwd ? outer : inner, co = bwd ? inner : outer;
        const int a = phase >> 1, b = phase & 1, r = tap / 3, c = tap % 3;
        int ky, kx;
        bool used;
        if (!bwd) {
            const int dy = r - 1, dx = c - 1;
            ky = a + 1 - 2 * dy; kx = b + 1 - 2 * dx;
            used = (a == 0 ? dy <= 0 : dy >= 0) && (b == 0 ? dx <= 0 : dx >= 0);
        } else {
            ky = a - 1 + 2 * r; kx = b - 1 + 2 * c;
            used = (a == 0 ? r >= 1 : r <= 1) && (b == 0 ? c >= 1 : c <= 1);
        }
        w3[i] = used ? wT[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx] : 0.0f;
    }
}

}  // namespace

// y = conv3x3(x, pad 1) * scale + shift (+ residual | ReLU mask) (ReLU) (2x2 max-pool), all NHWC fp32.
// Supported flags: DREAM_CONV_RELU, DREAM_CONV_POOL2 (output [B, H/2, W/2, Cout], floor), DREAM_CONV_RELUMASK (residual = mask source).
extern "C" int dream_conv3x3_winograd_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                               const float *residual, float *y, int B, int H, int W, int Cin, int Cout,
                                               int flags, void *stream) {
    DREAM_REQUIRE((flags & ~(DREAM_CONV_RELU | DREAM_CONV_POOL2 | DREAM_CONV_RELUMASK | DREAM_CONV_RES_AFTER_RELU)) == 0, "winograd conv: unsupported flags 0x%x", flags);
    DREAM_REQUIRE(!(flags & DREAM_CONV_RES_AFTER_RELU) || (residual != nullptr && !(flags & (DREAM_CONV_POOL2 | DREAM_CONV_RELUMASK))),
                  "winograd conv: residual-after-ReLU needs a residual and excludes the fused pool / the ReLU mask");
    DREAM_REQUIRE(!(flags & DREAM_CONV_POOL2) || (residual == nullptr && H >= 2 && W >= 2), "winograd conv: fused max-pool takes no residual");
    DREAM_REQUIRE(!(flags & DREAM_CONV_RELUMASK) || residual != nullptr, "winograd conv: ReLU mask without a mask tensor");
    WinoParams p;
    if (int rc = wino_setup(p, x, u_packed, scale, shift, residual, y, B, H, W, Cin, Cout, flags, 1)) return rc;
    const int nw = g_variant ? g_variant : (Cout > 64 ? 8 : 4);
    const int mode = (flags & DREAM_CONV_POOL2) ? 1 : (flags & DREAM_CONV_RELUMASK) ? 3 : (residual != nullptr ? 2 : 0);
    switch (mode + (nw == 8 ? 4 : 0)) {
        case 0: return launch_wino<4, 0>(p, stream);
        case 1: return launch_wino<4, 1>(p, stream);
        case 2: return launch_wino<4, 2>(p, stream);
        case 3: return launch_wino<4, 3>(p, stream);
        case 4: return launch_wino<8, 0>(p, stream);
        case 5: return launch_wino<8, 1>(p, stream);
        case 6: return launch_wino<8, 2>(p, stream);
        default: return launch_wino<8, 3>(p, stream);
    }
}

// nn.ConvTranspose2d(k4, s2, p1) (+ folded BatchNorm / bias, ReLU) of the ResNet decoder (dream/models.py:37-136) by minimal
// filtering: every output phase (a, b) is a 2 x 2-tap stride-1 conv of x; written as a 3x3 conv with a zero-padded kernel its
// Winograd-transformed weights vanish on seven of the 16 positions, so the Winograd kernel runs it with 9 multiplications per
// 2 x 2 outputs of the phase instead of the 16 of the direct sub-pixel form (dream_conv_transpose4x4s2_nhwc_f32) -- 1.78x fewer,
// same fp32 arithmetic.  x [B,H,W,Cin] -> y [B,2H,2W,Cout]; four launches (one per phase) sharing nothing but x.
//   u4: dream_pack_convT4x4_winograd_weight(): 4 x dream_conv3x3_winograd_weight_floats(Cout, Cin) floats.  flags: DREAM_CONV_RELU.
extern "C" size_t dream_convT4x4_winograd_weight_floats(int Cout, int Cin) { return 4 * dream_conv3x3_winograd_weight_floats(Cout, Cin); }

// wT [Cin][Cout][4][4] -> w3: the four zero-padded 3x3 kernels of the output phases, 4 x OIHW [Cout][Cin][3][3] (bwd = 0), or of the
// data gradient, 4 x [Cin][Cout][3][3] (bwd = 1): convT4x4_phase_kernels above
extern "C" int dream_convT4x4_phase_weights(const float *wT, float *w3, int Cin, int Cout, int bwd, void *stream) {
    DREAM_REQUIRE(wT && w3 && Cin > 0 && Cout > 0 && (bwd == 0 || bwd == 1), "convT phase weights: bad arguments");
    const size_t total = (size_t)4 * Cout * Cin * 9;
    size_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(convT4x4_phase_kernels, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, wT, w3, Cin, Cout, bwd);
    DREAM_LAUNCH_OK();
    return 0;
}

// wT [Cin][Cout][4][4] -> u4; scratch: 4 * Cout * Cin * 9 floats (the four zero-padded 3x3 kernels)
// mode 0: forward operator (u4: 4 x dream_conv3x3_winograd_weight_floats(Cout, Cin)); mode 1: data-gradient operator
// (dream_conv4x4s2_winograd_nhwc_f32; u4: 4 x dream_conv3x3_winograd_weight_floats(Cin, Cout))
extern "C" int dream_pack_convT4x4_winograd_weight(const float *wT, float *u4, float *scratch, int Cin, int Cout, int mode, void *stream) {
    DREAM_REQUIRE(wT && u4 && scratch && Cin > 0 && Cout > 0 && (mode == 0 || mode == 1), "winograd convT pack: bad arguments");
    if (int rc = dream_convT4x4_phase_weights(wT, scratch, Cin, Cout, mode, stream)) return rc;
    const int rows = mode == 0 ? Cout : Cin, cols = mode == 0 ? Cin : Cout;       // conv output / input channels
    const size_t per_u = dream_conv3x3_winograd_weight_floats(rows, cols);
    for (int ph = 0; ph < 4; ++ph)
        if (int rc = dream_pack_conv3x3_winograd_weight(scratch + (size_t)ph * Cout * Cin * 9, u4 + ph * per_u, rows, cols, 0, stream)) return rc;
    return 0;
}

// Data gradient of the transposed conv = a 4x4 stride-2 pad-1 conv of dY [B,2H,2W,Cout] -> dX [B,H,W,Cin], as the sum over the
// four output phases of a 2 x 2-tap conv on the phase's stride-2 view of dY: the same nine-position scheme (patterns mirrored),
// the phases accumulate into dX through the residual input of the epilogue.  u4: dream_pack_convT4x4_winograd_weight(mode 1).
extern "C" int dream_conv4x4s2_winograd_nhwc_f32(const float *dy, const float *u4, float *dx, int B, int H, int W, int Cout, int Cin,
                                                 void *stream) {
    DREAM_REQUIRE(Cin > 64, "winograd conv4x4s2: needs more than 64 output channels (the 8-wave kernel), got %d", Cin);
    WinoParams p;
    if (int rc = wino_setup(p, dy, u4, nullptr, nullptr, nullptr, dx, B, H, W, Cout, Cin, 0, 1, 2)) return rc;
    const size_t per_u = dream_conv3x3_winograd_weight_floats(Cin, Cout);
    for (int ph = 0; ph < 4; ++ph) {
        p.u = u4 + ph * per_u;
        p.in_oy = ph >> 1; p.in_ox = ph & 1;
        p.residual = ph == 0 ? nullptr : dx;         // phases 1..3 add to what is there (same thread reads and writes an element)
        int rc;
        // the data gradient's kernels have their taps in the opposite corner: pattern of phase (1 - a, 1 - b)
        switch (ph) {
            case 0: rc = launch_wino<8, 0, 4>(p, stream); break;
            case 1: rc = launch_wino<8, 2, 3>(p, stream); break;
            case 2: rc = launch_wino<8, 2, 2>(p, stream); break;
            default: rc = launch_wino<8, 2, 1>(p, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" int dream_conv_transpose4x4s2_winograd_nhwc_f32(const float *x, const float *u4, const float *scale, const float *shift,
                                                           float *y, int B, int H, int W, int Cin, int Cout, int flags, void *stream) {
    DREAM_REQUIRE((flags & ~DREAM_CONV_RELU) == 0, "winograd convT: unsupported flags 0x%x", flags);
    DREAM_REQUIRE(Cout > 64, "winograd convT: needs more than 64 output channels (the 8-wave kernel), got %d", Cout);
    WinoParams p;
    if (int rc = wino_setup(p, x, u4, scale, shift, nullptr, y, B, H, W, Cin, Cout, flags, 2)) return rc;
    const size_t per_u = dream_conv3x3_winograd_weight_floats(Cout, Cin);
    for (int ph = 0; ph < 4; ++ph) {
        p.u = u4 + ph * per_u;
        p.out_oy = ph >> 1; p.out_ox = ph & 1;
        int rc;
        switch (ph) {
            case 0: rc = launch_wino<8, 0, 1>(p, stream); break;
            case 1: rc = launch_wino<8, 0, 2>(p, stream); break;
            case 2: rc = launch_wino<8, 0, 3>(p, stream); break;
            default: rc = launch_wino<8, 0, 4>(p, stream); break;
        }
        if (rc) return rc;
    }
    return 0;
}

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
#include "stat_tree.h"
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

// STAT != 0: a train-mode BatchNorm folded into this launch (csrc/stat_tree.h; the 1x1 convs' form is in gemm1x1.hip).
//   1: batch statistics of y (sum, sum of squares per output channel over the `count` stored outputs of ALL launches that share
//      the tree -- one conv, or the four phase launches of a transposed conv) -> scale / shift of the FOLLOWING BatchNorm;
//   2: with MODE 3 and `residual` = the masked BatchNorm's INPUT z: y = g * [fmaf(a, z, b) > 0] (the ReLU mask recomputed
//      exactly as the forward pass evaluated it), per-channel sums of y (dbeta) and y * xhat (dgamma).
// A persistent workgroup is one producer row (row0 + blockIdx.x) of each 64-channel column block it covers.
// (A second kernel argument of the STAT kernels only: inside WinoParams the extra fields cost the plain kernels scalar registers.)
struct WinoStat {
    StatTree st;
    int row0;
    double count;
    BnFwdOut fwd;                           // STAT 1
    const float *zab, *mean, *invstd;       // STAT 2: [2][Cout], [Cout], [Cout] of the masked BatchNorm
    float *dgamma, *dbeta;                  // STAT 2 out
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

// The end of a STAT workgroup: its per-lane sums (channel n0 + 16 wave + (lane & 15), the lane's share of the tiles) -> one fp64
// row per 64-channel column block, then the ticket tree; the wavefront that finishes a column block writes the BatchNorm's results.
template <int NW, int STAT>
DREAM_DEVICE void wino_stat_finish(const WinoParams &p, const WinoStat &q, int n0, int wave, int lane, double s0, double s1) {
    // the four lanes l, l + 16, l + 32, l + 48 hold the same channel on different tiles: (0 + 1) + (2 + 3) in every lane
    s0 += lane_xor(s0, 16);
    s1 += lane_xor(s1, 16);
    s0 += lane_xor(s0, 32);
    s1 += lane_xor(s1, 32);
    const int col = n0 + wave * 16 + (lane & 15);
    const int row = q.row0 + (int)blockIdx.x;
    if ((lane >> 4) == 0 && col < p.Cout) {
        double *dst = q.st.rows + ((size_t)row * p.Cout + col) * 2;
        coherent_store(dst, s0);
        coherent_store(dst + 1, s1);
    }
    publish_wait();                                    // this wave's part of the row is out before the barrier releases the arriver
    __syncthreads();
    if ((wave & 3) != 0) return;                       // one wavefront per 64-channel column block carries on
    const int cb = (n0 >> 6) + (wave >> 2);
    if (cb * 64 >= p.Cout) return;
    double t0, t1;
    if (!stat_tree_arrive(q.st, cb, row, lane, &t0, &t1)) return;
    const int c = cb * 64 + lane;
    if (STAT == 1) {
        bn_finish_forward(q.fwd, c, p.Cout, q.count, t0, t1);
    } else if (c < p.Cout) {
        q.dbeta[c] = (float)t0;
        q.dgamma[c] = (float)t1;
    }
}

#define WINO_STAT 0
#include "conv_wino_body.inc"
#undef WINO_STAT
#define WINO_STAT 1
#include "conv_wino_body.inc"
#undef WINO_STAT

// OIHW (mode 0) or, for the data-gradient operator, IOHW with flipped taps (mode 1: rows = Cin_fwd, cols = Cout_fwd)
// -> U = G g G^T in fp64, rounded once to fp32, laid out [cols/16][16 positions][RowsPad][16]: pack_device.h (dream_pack::winograd2)
static_assert(WKC == 16 && WPAD == 128, "pack_device.h assumes 16-channel chunks and rows padded to 128");
__global__ void __launch_bounds__(256) wino_pack_kernel(const float *w, float *u, int Cout, int Cin, int mode) {
    dream_pack::winograd2(w, u, Cout, Cin, mode, (int)blockIdx.x, (int)gridDim.x);
}

constexpr int kCUs = 256;     // MI355X
int g_max_workgroups = 0;     // test hook: cap on resident workgroups (0 = the chip's capacity)

// persistent grid: as many workgroups as the 256 CUs hold at once (2 of the 4-wave, 1 of the 8-wave kind per CU), a
// multiple of 8 (XCDs), split over the output-channel blocks; fewer when there are fewer tile blocks than that
// read per call (four launches per layer): tests and A/B runs switch it through the environment
bool small_grid_nw4() { const char *e = getenv("DREAM_WINO_SMALL_GRID"); return !(e != nullptr && e[0] == '0'); }

int wino_grid_x(int nw, int Cout, int nblk) {
    const int ny = (Cout + 16 * nw - 1) / (16 * nw);
    int resident = g_max_workgroups > 0 ? g_max_workgroups : kCUs * (nw == 4 ? 2 : 1);
    int gx = resident / ny / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > (nblk + 7) / 8 * 8) gx = (nblk + 7) / 8 * 8;
    return gx;
}

template <int NW, int MODE, int PAT = 0>
int launch_wino(const WinoParams &p, void *stream) {
    void (*kernel)(const WinoParams) = conv_wino_kernel<NW, MODE, PAT>;
    const size_t lds = (size_t)2 * VB * sizeof(float) + (size_t)512 * 16;          // V buffers + the offset table
    if (dream_allow_full_lds((const void *)kernel)) return 2;
    const int ny = (p.Cout + 16 * NW - 1) / (16 * NW);
    const int gx = wino_grid_x(NW, p.Cout, p.nblk);
    const dim3 grid((unsigned)gx, (unsigned)ny);
    hipLaunchKernelGGL(kernel, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}
template <int NW, int MODE, int STAT>
int launch_wino_stat(const WinoParams &p, const WinoStat &q, void *stream) {
    void (*kernel)(const WinoParams, const WinoStat) = conv_wino_stat_kernel<NW, MODE, STAT>;
    const size_t lds = (size_t)2 * VB * sizeof(float) + (size_t)512 * 16;
    if (dream_allow_full_lds((const void *)kernel)) return 2;
    const int ny = (p.Cout + 16 * NW - 1) / (16 * NW);
    const dim3 grid((unsigned)wino_grid_x(NW, p.Cout, p.nblk), (unsigned)ny);
    hipLaunchKernelGGL(kernel, grid, dim3(64 * NW), lds, (hipStream_t)stream, p, q);
    DREAM_LAUNCH_OK();
    return 0;
}

int g_variant = 0;
int wino_nw(int Cout) { return g_variant ? g_variant : (Cout > 64 ? 8 : 4); }
int wino_stat_rows(int B, int H, int W, int Cout) {       // = the persistent grid's x extent: one producer row per workgroup (STAT)
    const long ntiles = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
    return wino_grid_x(wino_nw(Cout), Cout, (int)((ntiles + WT - 1) / WT));
}

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
    if (dream_zero_words(u + (size_t)(cols / WKC) * 16 * rows_pad * WKC, (size_t)B_AHEAD * rows_pad * WKC * sizeof(float), (hipStream_t)stream))
        return 2;
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
    const int nw = wino_nw(Cout);
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

// ---- train-mode BatchNorm folded into the 3x3 conv's launches (the ResNet-101 Bottlenecks' conv2; the 1x1 form: gemm1x1.hip) ----
// bytes of the partial-sum workspace / zero words of counters of the two entry points below (the counters are left zero).
// NOTE: they depend on dream_conv3x3_winograd_set_variant / _set_max_workgroups (test hooks): query after setting those.
extern "C" size_t dream_conv3x3_winograd_bn_workspace(int B, int H, int W, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64 != 0) return 0;
    return stat_tree_doubles(wino_stat_rows(B, H, W, Cout), Cout) * sizeof(double);
}
extern "C" int dream_conv3x3_winograd_bn_counters(int B, int H, int W, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 64 != 0) return 0;
    return stat_tree_counters(wino_stat_rows(B, H, W, Cout), Cout);
}

// y = conv3x3(x, pad 1) + shift, and the batch statistics of y for the BatchNorm that follows, finished inside the launch (the
// last workgroup to arrive per 64-channel block sums the workgroups' fp64 rows in a fixed order): save_mean / save_invstd,
// out_ab[0] = gamma * invstd, out_ab[1] = beta - mean * out_ab[0], running statistics as nn.BatchNorm2d updates them.
// Replaces conv2 -> bn2 (statistics pass) of torchvision's Bottleneck behind /root/reference/dream/models.py:22-32 in training.
extern "C" int dream_conv3x3_winograd_bnstats_nhwc_f32(const float *x, const float *u_packed, const float *shift, float *y, int B, int H,
                                                       int W, int Cin, int Cout, const float *gamma, const float *beta,
                                                       float *running_mean, float *running_var, long long *num_batches_tracked,
                                                       float eps, float momentum, float *out_ab, float *save_mean, float *save_invstd,
                                                       void *workspace, unsigned *counters, void *stream) {
    DREAM_REQUIRE(gamma && beta && out_ab && save_mean && save_invstd && workspace && counters, "winograd conv + BatchNorm statistics: null pointer");
    DREAM_REQUIRE(Cout % 64 == 0, "winograd conv + BatchNorm statistics: %d output channels, must be a multiple of 64", Cout);
    WinoParams p = {};
    if (int rc = wino_setup(p, x, u_packed, nullptr, shift, nullptr, y, B, H, W, Cin, Cout, 0, 1)) return rc;
    const int nw = wino_nw(Cout);
    WinoStat q = {};
    q.st = stat_tree_make(workspace, counters, wino_grid_x(nw, Cout, p.nblk), Cout);
    q.row0 = 0;
    q.count = (double)B * H * W;
    q.fwd.gamma = gamma; q.fwd.beta = beta; q.fwd.running_mean = running_mean; q.fwd.running_var = running_var;
    q.fwd.nbt = num_batches_tracked; q.fwd.eps = eps; q.fwd.momentum = momentum;
    q.fwd.ab = out_ab; q.fwd.mean = save_mean; q.fwd.invstd = save_invstd;
    return nw == 8 ? launch_wino_stat<8, 0, 1>(p, q, stream) : launch_wino_stat<4, 0, 1>(p, q, stream);
}

// Data gradient of a 3x3 conv whose INPUT was relu(BatchNorm(z)), with that BatchNorm's backward reductions in the epilogue:
// g = conv3x3(dy; mode-1 weights) * [ab[0] z + ab[1] > 0] -> g_out, and -- finished inside the launch -- dbeta = sum g,
// dgamma = sum g * (z - mean) * invstd.  dy [B,H,W,Cin], z / g_out [B,H,W,Cout] (Cout = the forward conv's input channels).
extern "C" int dream_conv3x3_winograd_bwd_bnmask_nhwc_f32(const float *dy, const float *u_packed_t, const float *z, float *g_out, int B,
                                                          int H, int W, int Cin, int Cout, const float *ab, const float *mean,
                                                          const float *invstd, float *dgamma, float *dbeta, void *workspace,
                                                          unsigned *counters, void *stream) {
    DREAM_REQUIRE(z && ab && mean && invstd && dgamma && dbeta && workspace && counters, "winograd data gradient + BatchNorm mask: null pointer");
    DREAM_REQUIRE(Cout % 64 == 0, "winograd data gradient + BatchNorm mask: %d channels, must be a multiple of 64", Cout);
    WinoParams p = {};
    if (int rc = wino_setup(p, dy, u_packed_t, nullptr, nullptr, z, g_out, B, H, W, Cin, Cout, DREAM_CONV_RELUMASK, 1)) return rc;
    const int nw = wino_nw(Cout);
    WinoStat q = {};
    q.st = stat_tree_make(workspace, counters, wino_grid_x(nw, Cout, p.nblk), Cout);
    q.row0 = 0;
    q.count = (double)B * H * W;
    q.zab = ab; q.mean = mean; q.invstd = invstd; q.dgamma = dgamma; q.dbeta = dbeta;
    return nw == 8 ? launch_wino_stat<8, 3, 2>(p, q, stream) : launch_wino_stat<4, 3, 2>(p, q, stream);
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
        // a grid of a few dozen eight-wavefront workgroups leaves most CUs empty (ResNet's first decoder layer at 16 frames: 2048 -> 256 on
        // 13x13 maps = 25 tile blocks x 2 channel blocks on 256 CUs, 1.2 ms): four-wavefront workgroups of 64 channels spread the same
        // wavefronts over twice the CUs (DREAM_WINO_SMALL_GRID=0: always eight).  Measured (profiles/r05_ab_convT_small_grid.txt): 50 -> 100
        // workgroups +0.8 % on the training step, +3.2 % on resnet_h inference at 16 frames; 98 -> 196 (resnet_f at 32 frames) LOSES 8-40 % (two
        // four-wavefront workgroups may share a CU while others stay empty): the rule stops at 72.
        if (small_grid_nw4() && p.nblk * ((Cout + 127) / 128) < 72) {
            switch (ph) {
                case 0: rc = launch_wino<4, 0, 1>(p, stream); break;
                case 1: rc = launch_wino<4, 0, 2>(p, stream); break;
                case 2: rc = launch_wino<4, 0, 3>(p, stream); break;
                default: rc = launch_wino<4, 0, 4>(p, stream); break;
            }
        } else {
            switch (ph) {
                case 0: rc = launch_wino<8, 0, 1>(p, stream); break;
                case 1: rc = launch_wino<8, 0, 2>(p, stream); break;
                case 2: rc = launch_wino<8, 0, 3>(p, stream); break;
                default: rc = launch_wino<8, 0, 4>(p, stream); break;
            }
        }
        if (rc) return rc;
    }
    return 0;
}

// 1x1 stride-1 convolution = plain GEMM  Y[pos][cout] = sum_cin X[pos][cin] W[cout][cin]  on NHWC fp32 tensors, on the
// CDNA4 fp32 matrix cores, WITHOUT LDS and without barriers.
//
// Replaces torch.nn.Conv2d(k=1) of the ResNet-101 bottlenecks behind /root/reference/dream/models.py:22-32 (conv1 / conv3 /
// the stride-1 downsample of layer1) in evaluation (folded BatchNorm scale / shift, residual add, ReLU in the epilogue) and
// training, and -- on mode-1 packed weights -- their data gradients (+ the sum with the other branch's gradient).
// 37 % of a ResNet training step at 16 frames per GPU are such GEMMs of ~10 000 x 1024 x 256: 33 us at the roofline, of which
// the LDS-staged direct kernel (conv_mfma.hip: patch staging, barriers per chunk, 128 x 128 workgroup tiles) needs 65-80.
//
//   * both operands are contiguous along K (input channels of a pixel; a packed weight row), so a lane feeds
//     v_mfma_f32_16x16x4_f32 straight from one 16-byte load: lane l = (row i = l & 15, k group g = l >> 4) loads
//     k = 16 t + 4 g .. +3 of its row and uses component e in k-step e -- the same assignment on both operands, so the MFMA
//     contracts matching k's (the order of a sum does not matter to the algebra, and is fixed);
//   * wave tile = 64 positions x 64 output channels: 4 + 4 loads feed 64 MFMAs per 16 k's; 64 accumulator registers, ~130
//     VGPRs: three waves per SIMD hide the load latency, register double-buffering of the operands does the rest;
//   * waves are independent in the K loop (no LDS, no barrier): a workgroup is four consecutive wave tiles, or -- when the
//     problem has too few tiles to give every SIMD two or three waves (25 x 25 and 13 x 13 maps at 16 frames) -- two / one
//     tile(s) whose K range is split over the waves, summed through LDS in a fixed order before the epilogue;
//   * weights are packed [K/16][channel rows][16] with the rows permuted so that MFMA column block n, lane column j holds
//     output channel 4 j + n: the four accumulator blocks of a lane are then four CONSECUTIVE channels of one position and
//     the epilogue loads scale / shift / residual and stores the result as float4 (256 contiguous bytes per 16 lanes);
//   * buffer descriptors: fixed per-lane offsets, the chunk offset in the scalar operand (no address arithmetic in the loop),
//     rows beyond the end read zeros and are not stored.
#include <stdlib.h>
#include <dream_cdna4.h>
#include "common.h"
#include "pack_device.h"
#include "stat_tree.h"
#include "../../include/dream_hip.h"

namespace {

struct GemmParams {
    const float *x;          // [M][x_stride] (first K columns used)
    const float *w;          // packed [K/16][NPad][16]
    const float *scale;      // [N] or null
    const float *shift;      // [N] or null
    const float *residual;   // [M][N] or null
    float *y;                // [M][N]
    int M, K, N, NPad, x_stride;
    int nrb, ncb;            // 64-row / 64-column blocks
    int flags;               // DREAM_CONV_RELU
    int prio_round;          // > 0: workgroups per dispatch round (one per CU); round r of a launch runs at wave priority 3 - r (below)
    // ---- train-mode BatchNorm folded into this GEMM (dream/models.py:22-32: conv -> BatchNorm2d -> ReLU chains of the Bottlenecks)
    // PRE: the A operand is relu(a[k] * x + b[k]) -- the PREVIOUS BatchNorm + ReLU applied in the loader, its output never stored
    const float *pre_ab;     // [2][K]: a then b
    // EPI 1: batch statistics of y (sum, sum of squares per output channel) -> scale / shift of the FOLLOWING BatchNorm
    // EPI 2: y = g * [fmaf(a, z, b) > 0] (ReLU mask recomputed from the masked BatchNorm's input z), sums of y and y * xhat
    //        -> dbeta, dgamma of that BatchNorm
    StatTree st;             // one row of partial sums per 64-row block (stat_tree.h)
    BnFwdOut st_fwd;         // EPI 1: statistics -> scale / shift of the following BatchNorm (st_fwd.mean / invstd: outputs)
    const float *st_z;       // EPI 2: [M][N]
    const float *st_zab;     // EPI 2: [2][N], or null when the mask comes from st_yact
    const float *st_yact;    // EPI 2: stored activation [M][N] whose sign is the ReLU mask (Bottleneck outputs: relu(BN(z) + identity)), or null
    const float *st_mean, *st_invstd;  // EPI 2: of the masked BatchNorm
    float *st_dgamma, *st_dbeta;       // EPI 2 out
};

// y = relu(a * x + b), the one expression every kernel uses for a train-mode BatchNorm + ReLU (single rounding: the ReLU mask
// recomputed in the backward pass is exactly the forward's)
DREAM_DEVICE f32x4 bn_relu4(f32x4 x, f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf(__builtin_fmaf(a[e], x[e], b[e]), 0.0f);
    return r;
}

// Round 6.  What the compiler made of the first version of this kernel (loads written one chunk ahead, `load(1); multiply(0); load(0);
// multiply(1)`): it sank ALL sixteen loads of a loop iteration behind the first 48 MFMAs and waited for them with vmcnt(14) .. vmcnt(1)
// right there -- every wavefront stalled for a full memory round trip per two chunks, and on operands that come from HBM (what the
// training step sees: the warm micro-benchmark had them in the Infinity Cache) the launches ran at 0.24-0.51 of the MFMA peak.  Now the
// order is PINNED (a scheduling barrier per group of four MFMAs): the ten / eight loads of chunk t + 1 are issued one per group during
// the first half of chunk t's MFMAs, into the register set the previous chunk freed.  The epilogue's operands (residual, BatchNorm
// input, stored activation: up to three 16-byte loads per output row) are issued DREAM_G1_EPI_DEPTH rows ahead through buffer
// descriptors (zeros for a null operand or a row past the end: no branches), the first rows before the LAST chunk's MFMAs.
#ifndef DREAM_G1_EPI_DEPTH
#define DREAM_G1_EPI_DEPTH 4
#endif
#ifndef DREAM_G1_EPI_DEPTH_MASK
#define DREAM_G1_EPI_DEPTH_MASK 3
#endif
#ifndef DREAM_G1_EPI_DEPTH_MASK_KS
#define DREAM_G1_EPI_DEPTH_MASK_KS 2
#endif
// RES: the epilogue adds a residual tensor (EPI 0: a separate instantiation, so that launches without one issue no loads for it; EPI 2
// always reads per-row operands: a null residual / stored activation there is an empty descriptor that returns zeros).
// MB: 16-row blocks of a wavefront tile (4: 64 positions x 64 channels, the product; 2: 32 x 64, an experiment kept behind
// dream_conv1x1_set_rows / DREAM_CONV1X1_ROWS=32).  The trunk GEMMs at 16 frames are ~10 000 positions: 2 512 64-row wave tiles on 1 024
// SIMDs = 2.45 per SIMD, i.e. some SIMDs carry three and set the pace (98 K MFMA cycles against 80 K for an even spread); 32-row tiles
// (5 024 half-size ones, four to five per SIMD) come within 2 % of even -- on paper.  Measured: slower (gemm1x1_rows below).
template <int KS, bool PRE = false, int EPI = 0, bool RES = false, int MB = 4>
// (with a K split the 16 KB x MB of partial tiles bound the workgroups per CU: two for MB = 4 -- two wavefronts per SIMD, 256 registers --,
// four for MB = 2 at <= 128 registers)
__global__ void __launch_bounds__(256, MB == 4 ? (KS == 1 ? 3 : 2) : 4) gemm1x1_kernel(const GemmParams p) {
    __shared__ float s_part[KS > 1 ? 4 * MB * 4 * 64 * 4 : 1];      // [wave][m][n][lane] float4: the waves' partial tiles
    __shared__ double s_stat[(KS > 1 && EPI != 0) ? 4 * 16 * 8 : 1];      // [wave][lane & 15][4 channels][2]: the waves' sums
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    // XCD-aware placement: workgroup b runs on XCD b % 8; give each XCD a contiguous range of wave tiles (column blocks of
    // one row block are neighbours: they re-read the same input rows and meet in that XCD's L2)
    const int nwg = (int)gridDim.x;
    const int wg = (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int tile = wg * (4 / KS) + wave / KS;            // KS waves share a tile, each with 1 / KS of the K range
    const int kpart = wave % KS;
    const bool live = tile < p.nrb * p.ncb;
    if (KS == 1 && !live) return;
    // De-phasing (round 6).  The two or three wavefronts a SIMD holds come from workgroups that were dispatched within microseconds of
    // each other: they run their K loops side by side, sharing the matrix pipe, and then reach their epilogues -- 64 rows x up to three
    // 16-byte operand loads + a store per lane, no MFMA -- TOGETHER, with the pipe idle.  Wave priorities by dispatch round (the first
    // 256 workgroups of a launch land one per CU, the next 256 beside them, ...) let the first round's wave win the pipe, so that it
    // finishes its K loop early and its epilogue runs under the second round's MFMAs, and so on.
    if (p.prio_round > 0) {
        const int round = (int)blockIdx.x / p.prio_round;
        if (round == 0) __builtin_amdgcn_s_setprio(3);
        else if (round == 1) __builtin_amdgcn_s_setprio(2);
        else if (round == 2) __builtin_amdgcn_s_setprio(1);
    }
    const int cb = tile % p.ncb, rb = tile / p.ncb;
    const int li = lane & 15, lg = lane >> 4;

    const BufferRsrc xbuf = make_buffer(p.x, (size_t)p.M * p.x_stride * sizeof(float));
    const BufferRsrc wbuf = make_buffer(p.w, (size_t)(p.K / 16) * p.NPad * 16 * sizeof(float));
    unsigned a_off[MB], b_off[4];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int row = rb * (16 * MB) + 16 * m + li;
        a_off[m] = row < p.M ? (unsigned)((row * p.x_stride + 4 * lg) * 4) : BUFFER_OOB;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) b_off[n] = (unsigned)(((cb * 64 + 16 * n + li) * 16 + 4 * lg) * 4);
    const unsigned b_chunk = (unsigned)(p.NPad * 16 * 4);        // bytes between k chunks of the packed weights

    f32x4 acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    f32x4 xa[2][MB], xb[2][4];
    f32x4 pa[2], pb[2];                              // PRE: scale / shift of this lane's four k's of the chunk
    const BufferRsrc abbuf = make_buffer(PRE ? p.pre_ab : p.x, PRE ? (size_t)2 * p.K * sizeof(float) : 0);
    constexpr int NLOADS = (PRE ? 6 : 4) + MB;
    // the i-th load of chunk t into register set `set`: weights first (L2 residents: they return quickly and the first MFMAs of the
    // chunk need all four of them), then the PRE scale / shift, then the four row blocks of the activation
    auto issue = [&](int set, int t, int i) {
        if (i < 4) xb[set][i] = buffer_load_x4(wbuf, b_off[i], (unsigned)t * b_chunk);
        else if (PRE && i == 4) pa[set] = buffer_load_x4(abbuf, (unsigned)(4 * lg * 4), (unsigned)t * 64u);
        else if (PRE && i == 5) pb[set] = buffer_load_x4(abbuf, (unsigned)((p.K + 4 * lg) * 4), (unsigned)t * 64u);
        else xa[set][i - (NLOADS - MB)] = buffer_load_x4(xbuf, a_off[i - (NLOADS - MB)], (unsigned)t * 64u);
    };

    // ---- epilogue operands: a ring of DREAM_G1_EPI_DEPTH rows -------------------------------------------------------------------
    // A wavefront owns NR = 4 MB / KS of the 4 MB (block row, register) units of its lanes, in order: unit u = kpart NR + i is block row
    // m = u >> 2, register r = u & 3 (MB = 2 with a four-way K split: half a block row per wavefront).
    // (EPI 2 loads three operands per row: its ring is shallower, the register budget being 168 at three wavefronts per SIMD)
    // (the 32-row tiles run four wavefronts per SIMD: 128 registers, a ring one row shallower)
    constexpr int DWANT = EPI == 2 ? (KS == 1 ? DREAM_G1_EPI_DEPTH_MASK : DREAM_G1_EPI_DEPTH_MASK_KS) - (MB == 2 ? 1 : 0) : DREAM_G1_EPI_DEPTH;
    constexpr int NR = 4 * MB / KS, D = DWANT < NR ? DWANT : NR;
    static_assert(NR >= 1, "K split too deep for this tile height");
    constexpr bool HAS_RES = RES || EPI == 2;                 // (the statistics form, EPI 1, has no residual input)
    constexpr bool EL = HAS_RES;                              // the rows have operands to load
    const int c0 = cb * 64 + 4 * li;
    const bool cok = c0 < p.N;                                // N % 4 == 0
    // descriptors on the tile's first row (tensors may exceed a descriptor's 2 GB: the base moves, the offsets stay small); a null
    // operand gets an empty descriptor, a row past M or a channel past N reads zeros / stores nothing
    const long trow = (long)rb * (16 * MB);
    const size_t tbytes = live ? (size_t)(p.M - trow) * p.N * sizeof(float) : 0;
    const BufferRsrc rbuf = make_buffer(HAS_RES && p.residual ? p.residual + trow * p.N : p.x, HAS_RES && p.residual ? tbytes : 0);
    const BufferRsrc zbuf = make_buffer(EPI == 2 ? p.st_z + trow * p.N : p.x, EPI == 2 ? tbytes : 0);
    const BufferRsrc yabuf = make_buffer(EPI == 2 && p.st_yact ? p.st_yact + trow * p.N : p.x, EPI == 2 && p.st_yact ? tbytes : 0);
    const BufferRsrc ybuf = make_buffer(p.y + trow * p.N, tbytes);
    // byte offset of row i of this wavefront inside the tile: lane part (OOB for a channel past N: it stays out of range under the
    // additions below, every descriptor being smaller than 2 GB) + wave-uniform part
    const unsigned lane_off = cok ? (unsigned)((4 * lg * p.N + c0) * 4) : BUFFER_OOB;
    const unsigned row_bytes = (unsigned)p.N * 4u;
    auto row_off = [&](int i) {
        const int u = kpart * NR + i;
        return lane_off + (unsigned)(16 * (u >> 2) + (u & 3)) * row_bytes;
    };
    f32x4 e_res[D], e_z[D], e_ya[D];
    auto epi_load = [&](int i) {                              // i compile-time after unrolling
        const unsigned o = row_off(i);
        if (HAS_RES) e_res[i % D] = buffer_load_x4(rbuf, o, 0u);
        if (EPI == 2) {
            e_z[i % D] = buffer_load_x4(zbuf, o, 0u);
            e_ya[i % D] = buffer_load_x4(yabuf, o, 0u);
        }
    };

    // multiply the chunk in register set S; meanwhile issue the loads of chunk tn into the other set (LOAD) or the first rows of the
    // epilogue's operands (EPILOAD: the last chunk)
    auto step = [&](auto s_tag, int tn, auto load_tag, auto epi_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value, EPILOAD = decltype(epi_tag)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (PRE && e == 0) xa[S][m] = bn_relu4(xa[S][m], pa[S], pb[S]);      // eight VALU operations beside the group's MFMAs
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = mfma_f32_16x16x4(xa[S][m][e], xb[S][n][e], acc[m][n]);
                const int g = MB * e + m;                     // group of four MFMAs
                if (LOAD && g < NLOADS) issue(1 - S, tn, g);
                if (EPILOAD && EL && g < D) epi_load(g);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    const int nchunks = p.K / 16 / KS;               // even (host side); this wave's chunks start at t0
    const int t0 = kpart * nchunks;
    if (live) {
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) issue(0, t0, i);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t + 2 < nchunks; t += 2) {
            step(I0{}, t0 + t + 1, Yes{}, No{});
            step(I1{}, t0 + t + 2, Yes{}, No{});
        }
        step(I0{}, t0 + nchunks - 1, Yes{}, No{});
        step(I1{}, 0, No{}, Yes{});
    }
    // split K: partial tiles through LDS; wave kpart then owns the units [kpart NR, (kpart + 1) NR) = block rows below (fixed summation order)
    if (KS > 1) {
        f32x4 *sp = (f32x4 *)s_part;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) sp[((wave * MB + m) * 4 + n) * 64 + lane] = acc[m][n];
        __syncthreads();
        const int w0 = wave - kpart;                 // first wave of this tile
#pragma unroll
        for (int m = 0; m < MB; ++m)
            if (m >= ((kpart * NR) >> 2) && m <= (((kpart + 1) * NR - 1) >> 2)) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f32x4 v = sp[(((w0 + 0) * MB + m) * 4 + n) * 64 + lane];
#pragma unroll
                    for (int k = 1; k < KS; ++k) v = v + sp[(((w0 + k) * MB + m) * 4 + n) * 64 + lane];
                    acc[m][n] = v;
                }
            }
        if (EPI == 0 && !live) return;               // (with statistics the dead waves stay for the second barrier)
    }

    // ---- epilogue: lane holds rows 4 (l >> 4) + r, column j = l & 15 of every block (m, n) = channels 4 j .. 4 j + 3 ----------
    f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p.scale != nullptr && cok) sc = *(const f32x4 *)(p.scale + c0);
    if (p.shift != nullptr && cok) sh = *(const f32x4 *)(p.shift + c0);
    // ReLU without a branch per row: max(v, 0) or max(v, -inf)
    const float relu_lo = (p.flags & DREAM_CONV_RELU) != 0 ? 0.0f : -__builtin_inff();
    double st0[4] = {0.0, 0.0, 0.0, 0.0}, st1[4] = {0.0, 0.0, 0.0, 0.0};      // EPI: this lane's sums over its rows
    f32x4 za = {0.0f, 0.0f, 0.0f, 0.0f}, zb = za, zmu = za, zis = za;
    const bool mask_y = EPI == 2 && p.st_yact != nullptr;
    if (EPI == 2 && cok) {
        if (!mask_y) {
            za = *(const f32x4 *)(p.st_zab + c0);
            zb = *(const f32x4 *)(p.st_zab + p.N + c0);
        }
        zmu = *(const f32x4 *)(p.st_mean + c0);
        zis = *(const f32x4 *)(p.st_invstd + c0);
    }
    // Row i + D's operands are loaded BEFORE row i is stored (safe if the output aliases an operand: a thread reads exactly the elements
    // it writes, D rows ahead).  With a K split the block rows depend on the wavefront (kpart): one instantiation per value, chosen by a
    // wave-uniform branch, so that the accumulator indices stay compile-time constants.
    auto rows = [&](auto kp_tag) {
        constexpr int KP = decltype(kp_tag)::value;            // = kpart, as a constant: the accumulator indices below are compile-time
        const int mrows = p.M - (int)trow - 4 * lg;            // rows of this lane's quad column that exist
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = (KP * NR + i) >> 2, r = (KP * NR + i) & 3;
            f32x4 res = {0.0f, 0.0f, 0.0f, 0.0f}, z = res, ya = res;
            if (EL && HAS_RES) res = e_res[i % D];
            if (EL && EPI == 2) { z = e_z[i % D]; ya = e_ya[i % D]; }
            if (EL && i + D < NR) epi_load(i + D);
            f32x4 v = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
            // the entry points of the statistics forms pass no scale and no ReLU flag (EPI 1: an optional shift; EPI 2: nothing): their
            // epilogues do not spend a multiplication, an addition and a maximum per element on 1, 0 and -inf
            if (EPI == 0) v = v * sc + sh;
            if (EPI == 1) v = v + sh;
            if (EL && HAS_RES) v = v + res;                    // (EPI 2 without a residual adds the empty descriptor's zeros: only -0 becomes +0)
            if (EPI == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_lo);
            }
            const bool rok = 16 * m + r < mrows;              // (a row past M is not stored and counts as zeros: x + 0 is exact)
            if (EPI == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vm = rok ? v[e] : 0.0f;
                    st0[e] += (double)vm;
                    st1[e] += (double)vm * (double)vm;
                }
            }
            if (EPI == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool on = mask_y ? ya[e] > 0.0f : __builtin_fmaf(za[e], z[e], zb[e]) > 0.0f;
                    v[e] = on ? v[e] : 0.0f;
                    const float xh = (z[e] - zmu[e]) * zis[e];
                    // a row past M needs no mask here: its operand rows (dy, residual, z) lie beyond their buffers and were loaded as
                    // zeros, so v = 0 and xh is finite
                    st0[e] += (double)v[e];
                    st1[e] += (double)v[e] * (double)xh;
                }
            }
            buffer_store_x4(ybuf, v, row_off(i), 0u);
            // one row at a time: left alone, the scheduler interleaves all rows' fp64 statistics and the kernel's register count (the
            // maximum over the program) doubles -- 206 VGPRs + 96 AGPRs for <4, false, 2>, one wavefront per SIMD
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (live) {
        if constexpr (KS == 1) {
            rows(std::integral_constant<int, 0>{});
        } else if constexpr (KS == 2) {
            if (kpart == 0) rows(std::integral_constant<int, 0>{});
            else rows(std::integral_constant<int, 1>{});
        } else {
            switch (kpart) {
                case 0: rows(std::integral_constant<int, 0>{}); break;
                case 1: rows(std::integral_constant<int, 1>{}); break;
                case 2: rows(std::integral_constant<int, 2>{}); break;
                default: rows(std::integral_constant<int, 3>{}); break;
            }
        }
    }
    if (EPI != 0) {
        // the four lanes l, l + 16, l + 32, l + 48 hold the same four channels on different rows: (0 + 1) + (2 + 3) in every lane
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st0[e] += lane_xor(st0[e], 16);
            st1[e] += lane_xor(st1[e], 16);
            st0[e] += lane_xor(st0[e], 32);
            st1[e] += lane_xor(st1[e], 32);
        }
        if (KS > 1) {
            // the KS waves of a tile hold disjoint row blocks: summed through LDS in wave order, wave kpart == 0 carries on
            if (lg == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s_stat[((wave * 16 + li) * 4 + e) * 2] = st0[e]; s_stat[((wave * 16 + li) * 4 + e) * 2 + 1] = st1[e]; }
            }
            __syncthreads();
            if (!live || kpart != 0) return;
            if (lg == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        a += s_stat[(((wave + k) * 16 + li) * 4 + e) * 2];
                        b += s_stat[(((wave + k) * 16 + li) * 4 + e) * 2 + 1];
                    }
                    st0[e] = a;
                    st1[e] = b;
                }
            }
        }
        if (lg == 0 && cok) {
            double *dst = p.st.rows + ((size_t)rb * p.N + c0) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { coherent_store(dst + 2 * e, st0[e]); coherent_store(dst + 2 * e + 1, st1[e]); }
        }
        double s0, s1;
        if (!stat_tree_arrive(p.st, cb, rb, lane, &s0, &s1)) return;
        const int c = cb * 64 + lane;
        if (EPI == 1) {
            bn_finish_forward(p.st_fwd, c, p.N, (double)p.M, s0, s1);
        } else if (c < p.N) {
            p.st_dbeta[c] = (float)s0;
            p.st_dgamma[c] = (float)s1;
        }
    }
}

// packed operand layout: pack_device.h (dream_pack::conv1x1)
__global__ void __launch_bounds__(256) gemm1x1_pack_kernel(const float *w, float *packed, int Cout, int Cin, int mode) {
    dream_pack::conv1x1(w, packed, Cout, Cin, mode, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 1x1 stride-1 conv:  dW[co][ci] = sum_pos dY[pos][co] X[pos][ci]  -- a GEMM whose contraction runs over
// POSITIONS, i.e. over the slow axis of both NHWC operands.  v_mfma_f32_16x16x4_f32 wants A[row][k] / B[k][col] with lane
// l = (row / col i = l & 15, k = l >> 4): a lane loads the 16 bytes dY[pos 4 s + (l >> 4)][4 i .. 4 i + 3] and uses component m as
// the A operand of the MFMA whose 16 rows are the channels 4 i + m (the same for X and the columns): ONE load per operand and
// k-step feeds a 4 x 4 block of MFMAs = 64 x 64 channel pairs, each load instruction reads 4 x 256 contiguous bytes, and
// nothing is staged or transformed.  The four waves of a workgroup take a quarter each of the workgroup's position range
// for one 64 x 64 tile, sum through LDS (fixed order) and write ONE partial tile per workgroup; a second kernel sums the
// workgroups' partials in a fixed order (deterministic, no atomics).
// Replaces ATen's conv backward-weight for the 1x1 convs of ResNet-101 (dream/models.py:22-32 via network.py:335).
struct Wgrad1x1Params {
    const float *x;          // [M][Cin]
    const float *dy;         // [M][Cdy]
    float *partial;          // [nsplit][Cout][Cin]
    int M, Cin, Cout, Cdy;
    int ncob, ncib;          // 64-channel blocks
    int chunks_per_wave;     // chunks of 16 positions each wave sums (even)
    const float *pre_ab;     // PRE: x is the INPUT of a BatchNorm + ReLU that was never stored: the operand is relu(a[ci] x + b[ci]); [2][Cin]
};

template <bool PRE>
__global__ void __launch_bounds__(256, 2) wgrad1x1_kernel(const Wgrad1x1Params p) {
    __shared__ float s_part[4 * 64 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = wave_index();
    const int tile = (int)blockIdx.x % (p.ncob * p.ncib), split = (int)blockIdx.x / (p.ncob * p.ncib);
    const int cob = tile % p.ncob, cib = tile / p.ncob;
    const int li = lane & 15, lg = lane >> 4;
    const BufferRsrc xbuf = make_buffer(p.x, (size_t)p.M * p.Cin * sizeof(float));
    const BufferRsrc ybuf = make_buffer(p.dy, (size_t)p.M * p.Cdy * sizeof(float));
    const long pos0 = ((long)split * 4 + wave) * p.chunks_per_wave * 16;          // first position of this wave
    // lane offset: position pos0 + (l >> 4), channels 4 i ..; beyond the tensor the hardware returns zeros
    const bool any = pos0 < p.M;
    const unsigned y_off = any ? (unsigned)(((pos0 + lg) * p.Cdy + cob * 64 + 4 * li) * 4) : BUFFER_OOB;
    const unsigned x_off = any ? (unsigned)(((pos0 + lg) * p.Cin + cib * 64 + 4 * li) * 4) : BUFFER_OOB;
    const unsigned y_step = (unsigned)(4 * p.Cdy * 4), x_step = (unsigned)(4 * p.Cin * 4);      // bytes per k-step (4 positions)

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 ya[2][4], xa[2][4];                        // [register set][k-step of the chunk]
    f32x4 pa = {1.0f, 1.0f, 1.0f, 1.0f}, pb = {0.0f, 0.0f, 0.0f, 0.0f};      // a lane's four input channels never change
    if (PRE) {
        pa = *(const f32x4 *)(p.pre_ab + cib * 64 + 4 * li);
        pb = *(const f32x4 *)(p.pre_ab + p.Cin + cib * 64 + 4 * li);
    }
    // the i-th of the eight loads of chunk c (16 positions) into register set `set`: k-step i >> 1, dy before x
    auto issue = [&](int set, int c, int i) {
        // the whole offset in the VECTOR operand: the hardware's bounds check (zeros beyond the last position) does not
        // see the scalar offset
        const int s = i >> 1;
        if ((i & 1) == 0) ya[set][s] = buffer_load_x4(ybuf, y_off + (unsigned)(4 * c + s) * y_step, 0);
        else xa[set][s] = buffer_load_x4(xbuf, x_off + (unsigned)(4 * c + s) * x_step, 0);
    };
    // multiply the chunk in set S and issue the loads of chunk cn into the other set, one per group of four MFMAs during the first
    // half of the chunk (pinned: left alone, the compiler sinks every load to its use -- round 6, see gemm1x1_kernel)
    auto step = [&](auto s_tag, int cn, auto load_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // (positions beyond the tensor read zeros on BOTH operands: relu(b) there meets dy = 0)
            if (PRE) xa[S][s] = bn_relu4(xa[S][s], pa, pb);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = mfma_f32_16x16x4(ya[S][s][m], xa[S][s][n], acc[m][n]);
                const int g = 4 * s + m;
                if (LOAD && g < 8) issue(1 - S, cn, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    const int nchunks = p.chunks_per_wave;           // even
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(0, 0, i);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c + 2 < nchunks; c += 2) {
        step(I0{}, c + 1, Yes{});
        step(I1{}, c + 2, Yes{});
    }
    step(I0{}, nchunks - 1, Yes{});
    step(I1{}, 0, No{});
    // the workgroup's four partial tiles through LDS; wave w then owns row block m = w
    f32x4 *sp = (f32x4 *)s_part;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sp[((wave * 4 + m) * 4 + n) * 64 + lane] = acc[m][n];
    __syncthreads();
    float *out = p.partial + (size_t)split * p.Cout * p.Cin;
    const int ci = cib * 64 + 4 * li;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (m == wave) {
            f32x4 v[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                v[n] = sp[((0 * 4 + m) * 4 + n) * 64 + lane];
#pragma unroll
                for (int k = 1; k < 4; ++k) v[n] = v[n] + sp[((k * 4 + m) * 4 + n) * 64 + lane];
            }
            // lane holds rows i = 4 (l >> 4) + r of block m (channel co = 4 i + m), column j = l & 15 (channels ci = 4 j + n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cob * 64 + 4 * (4 * lg + r) + m;
                if (co < p.Cout) *(f32x4 *)(out + (size_t)co * p.Cin + ci) = f32x4{v[0][r], v[1][r], v[2][r], v[3][r]};
            }
        }
}

__global__ void __launch_bounds__(256) wgrad1x1_reduce_kernel(const float *partial, float *dw, int nsplit, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
        f32x4 s = *(const f32x4 *)(partial + i);
        int k = 1;
        for (; k + 8 <= nsplit; k += 8) {           // eight loads in flight, summed in index order (same bits as one by one)
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4 *)(partial + (size_t)(k + j) * n + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = s + v[j];
        }
        for (; k + 4 <= nsplit; k += 4) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *(const f32x4 *)(partial + (size_t)(k + j) * n + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = s + v[j];
        }
        for (; k < nsplit; ++k) s = s + *(const f32x4 *)(partial + (size_t)k * n + i);
        *(f32x4 *)(dw + i) = s;
    }
}

struct Wgrad1x1Plan { int nsplit, chunks_per_wave; };

Wgrad1x1Plan wgrad1x1_plan(long M, int Cin, int Cout) {
    const int tiles = ((Cout + 63) / 64) * (Cin / 64);
    const long chunks = (M + 15) / 16;
    const long target = wgrad_target_workgroups(768);             // ~3 workgroups per CU in total at full width
    long nsplit = (target + tiles - 1) / tiles;
    const long max_by_work = (chunks + 31) / 32;                   // at least 8 chunks per wave
    if (nsplit > max_by_work) nsplit = max_by_work;
    if (nsplit < 1) nsplit = 1;
    long cpw = (chunks + nsplit * 4 - 1) / (nsplit * 4);
    cpw = (cpw + 1) / 2 * 2;
    Wgrad1x1Plan pl;
    pl.chunks_per_wave = (int)cpw;
    pl.nsplit = (int)((chunks + cpw * 4 - 1) / (cpw * 4));
    return pl;
}

int g_conv1x1_ksplit = 0;     // test hook: 0 = by shape, 1 / 2 / 4 = force
int g_conv1x1_rows = 0;       // test / A-B hook: 0 = by shape, 64 / 32 = force the wavefront tile height

template <bool PRE, int EPI, bool RES, int MB>
int gemm1x1_launch_ks(const GemmParams &p, int ks, unsigned grid, void *stream) {
    if (ks == 1) hipLaunchKernelGGL((gemm1x1_kernel<1, PRE, EPI, RES, MB>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (ks == 2) hipLaunchKernelGGL((gemm1x1_kernel<2, PRE, EPI, RES, MB>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm1x1_kernel<4, PRE, EPI, RES, MB>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    return 0;
}

template <int MB>
int gemm1x1_launch_form(const GemmParams &p, int ks, unsigned grid, bool pre, int epi, void *stream) {
    if (epi == 0 && pre) return gemm1x1_launch_ks<true, 0, false, MB>(p, ks, grid, stream);          // (the head conv behind the last decoder BatchNorm)
    if (epi == 0) return p.residual != nullptr ? gemm1x1_launch_ks<false, 0, true, MB>(p, ks, grid, stream) : gemm1x1_launch_ks<false, 0, false, MB>(p, ks, grid, stream);
    if (epi == 1) return pre ? gemm1x1_launch_ks<true, 1, false, MB>(p, ks, grid, stream) : gemm1x1_launch_ks<false, 1, false, MB>(p, ks, grid, stream);
    return gemm1x1_launch_ks<false, 2, false, MB>(p, ks, grid, stream);
}

// wavefront tile height (rows) for a problem of M positions x NPad channels
int gemm1x1_rows(long M, int NPad) {
    if (g_conv1x1_rows == 64 || g_conv1x1_rows == 32) return g_conv1x1_rows;
    static const int env_rows = [] { const char *e = getenv("DREAM_CONV1X1_ROWS"); return e ? atoi(e) : 0; }();      // A/B runs: 64 | 32
    if (env_rows == 64 || env_rows == 32) return env_rows;
    // 64 rows.  The 32-row tiles were built for the trunk GEMMs at 16 frames (~10 000 positions: 2.45 64-row wave tiles per SIMD, i.e. the
    // SIMDs that carry three set the pace; 4.9 half-size ones spread evenly) and MEASURED (round 6, profiles/r06_ab_gemm1x1_rows.txt,
    // alternating on one box): 8-12 % SLOWER per launch on the 25 x 25 and 50 x 50 layers (the fixed cost per wavefront -- first-load
    // latency, LDS exchange, epilogue -- doubles), faster only on the six 13 x 13 launches of a step; resnet_h training at 16 frames
    // 377-381 against 383 frames/s.  They stay behind the hook / DREAM_CONV1X1_ROWS=32.
    (void)M; (void)NPad;
    return 64;
}

int gemm1x1_launch(GemmParams p, long M, int K, int N, int x_stride, bool pre, int epi, void *stream) {
    p.M = (int)M; p.K = K; p.N = N; p.NPad = (N + 63) / 64 * 64; p.x_stride = x_stride;
    const int rows = gemm1x1_rows(M, p.NPad);
    p.nrb = (int)((M + rows - 1) / rows); p.ncb = p.NPad / 64;
    const long tiles = (long)p.nrb * p.ncb;
    // 64-row tiles: the chip holds 3072 waves of this kernel (256 CUs x 4 SIMDs x 3): split K until the tiles give ~2 waves per SIMD.
    // 32-row tiles (half the work each): until they give ~4
    int ks = 1;
    const long per_level = rows == 64 ? 768 : 1280;
    if (g_conv1x1_ksplit > 0) ks = g_conv1x1_ksplit;
    else if (p.flags & DREAM_CONV_NO_KSPLIT) ks = 1;
    else if (tiles < per_level && K % 128 == 0) ks = 4;
    else if (tiles < 2 * per_level && K % 64 == 0) ks = 2;
    DREAM_REQUIRE(K % (32 * ks) == 0, "conv1x1: K=%d cannot be split %d ways", K, ks);
    const int per_wg = 4 / ks;
    const unsigned grid = (unsigned)(((tiles + per_wg - 1) / per_wg + 7) / 8 * 8);
    static const int env_prio = [] { const char *e = getenv("DREAM_G1_PRIO"); return e ? atoi(e) : 0; }();       // A/B: workgroups per priority round
    p.prio_round = env_prio;
    return rows == 64 ? gemm1x1_launch_form<4>(p, ks, grid, pre, epi, stream) : gemm1x1_launch_form<2>(p, ks, grid, pre, epi, stream);
}

}  // namespace

// Test / A-B hook: force the wavefront tile height (0 = by problem size, 64, 32).  Same sums in the same order: same bits.
extern "C" int dream_conv1x1_set_rows(int rows) {
    DREAM_REQUIRE(rows == 0 || rows == 64 || rows == 32, "conv1x1: tile height %d", rows);
    g_conv1x1_rows = rows;
    return 0;
}

// Test hook: force the K split (0 = by problem size).  The result depends on it only through the summation order.
extern "C" int dream_conv1x1_set_ksplit(int ks) {
    DREAM_REQUIRE(ks == 0 || ks == 1 || ks == 2 || ks == 4, "conv1x1: K split %d", ks);
    g_conv1x1_ksplit = ks;
    return 0;
}

extern "C" size_t dream_conv1x1_weight_floats(int rows, int K) {
    return (size_t)(K / 16) * (size_t)((rows + 63) / 64 * 64) * 16;
}

// w_oihw [Cout][Cin][1][1]; mode 0: forward operator (rows = Cout, K = Cin); mode 1: data-gradient operator (rows = Cin, K = Cout)
extern "C" int dream_pack_conv1x1_weight(const float *w_oihw, float *packed, int Cout, int Cin, int mode, void *stream) {
    DREAM_REQUIRE(w_oihw && packed && Cout > 0 && Cin > 0 && (mode == 0 || mode == 1), "conv1x1 pack: bad arguments");
    const int rows = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
    DREAM_REQUIRE(K % 32 == 0, "conv1x1 pack: contraction length %d must be a multiple of 32", K);
    const int rows_pad = (rows + 63) / 64 * 64;
    const size_t total = (size_t)(K / 16) * rows_pad * 16;
    size_t grid = (total + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gemm1x1_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, Cout, Cin, mode);
    DREAM_LAUNCH_OK();
    return 0;
}

// y[M][N] = x[M][:K] . w^T * scale + shift (+ residual) (ReLU): M = B*H*W positions of an NHWC tensor with x_stride channels
// per pixel (>= K: padded tensors), N = output channels (a multiple of 4), K a multiple of 32.  flags: DREAM_CONV_RELU.
extern "C" int dream_conv1x1_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                                      const float *residual, float *y, long M, int K, int N, int x_stride, int flags, void *stream) {
    DREAM_REQUIRE(x && w_packed && y, "conv1x1: null pointer");
    DREAM_REQUIRE(M > 0 && K > 0 && N > 0 && x_stride >= K, "conv1x1: bad shape M=%ld K=%d N=%d stride=%d", M, K, N, x_stride);
    DREAM_REQUIRE(K % 32 == 0 && N % 4 == 0 && x_stride % 4 == 0, "conv1x1: K %% 32, N %% 4, stride %% 4 (got %d, %d, %d)", K, N, x_stride);
    DREAM_REQUIRE((flags & ~(DREAM_CONV_RELU | DREAM_CONV_NO_KSPLIT)) == 0, "conv1x1: unsupported flags 0x%x", flags);
    DREAM_REQUIRE((size_t)M * (size_t)x_stride * 4 < ((size_t)1 << 31) && (size_t)M * (size_t)N * 4 < ((size_t)1 << 33),
                  "conv1x1: tensor too large for 32-bit offsets");
    GemmParams p = {};
    p.x = x; p.w = w_packed; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.flags = flags;
    return gemm1x1_launch(p, M, K, N, x_stride, false, 0, stream);
}

// y = relu(pre_ab[0][k] * x + pre_ab[1][k]) . w^T + shift: the conv behind a train-mode BatchNorm + ReLU whose output is never stored
// (round 6: the head conv of the ResNet decoder, dream/models.py:37-136 -- 256 channels -> the keypoint maps; N = their count rounded up to 4)
extern "C" int dream_conv1x1_pre_nhwc_f32(const float *x, const float *w_packed, const float *pre_ab, const float *shift, float *y, long M,
                                          int K, int N, int x_stride, void *stream) {
    DREAM_REQUIRE(x && w_packed && pre_ab && y, "conv1x1_pre: null pointer");
    DREAM_REQUIRE(M > 0 && K > 0 && N > 0 && x_stride >= K, "conv1x1_pre: bad shape M=%ld K=%d N=%d stride=%d", M, K, N, x_stride);
    DREAM_REQUIRE(K % 32 == 0 && N % 4 == 0 && x_stride % 4 == 0, "conv1x1_pre: K %% 32, N %% 4, stride %% 4 (got %d, %d, %d)", K, N, x_stride);
    DREAM_REQUIRE((size_t)M * (size_t)x_stride * 4 < ((size_t)1 << 31) && (size_t)M * (size_t)N * 4 < ((size_t)1 << 33),
                  "conv1x1_pre: tensor too large for 32-bit offsets");
    GemmParams p = {};
    p.x = x; p.w = w_packed; p.shift = shift; p.y = y; p.pre_ab = pre_ab;
    return gemm1x1_launch(p, M, K, N, x_stride, true, 0, stream);
}

// ---- the same GEMM with a train-mode BatchNorm folded in on either side (see GemmParams) ---------------------------------------
// (sized for the 32-row wavefront tiles: one row of partial sums per row block, twice as many as with 64-row tiles)
extern "C" size_t dream_conv1x1_bn_workspace(long M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return stat_tree_doubles((int)((M + 31) / 32), N) * sizeof(double);
}
extern "C" int dream_conv1x1_bn_counters(long M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int a = stat_tree_counters((int)((M + 31) / 32), N), b = stat_tree_counters((int)((M + 63) / 64), N);
    return a > b ? a : b;
}

// Forward of  [BatchNorm(batch stats) -> ReLU ->] conv1x1 -> (statistics of the result for the BatchNorm that follows):
//   y = relu(pre_ab[0][k] * x + pre_ab[1][k]) . w^T + shift          (pre_ab null: y = x . w^T + shift)
//   save_mean / save_invstd = batch statistics of y per channel (biased variance, eps inside the root), out_ab[0] = gamma * invstd,
//   out_ab[1] = beta - mean * out_ab[0]; running statistics updated as nn.BatchNorm2d does (momentum, unbiased variance).
// The statistics are finished inside the launch by the last wave to arrive per 64-channel block (fixed summation order:
// deterministic).  counters: dream_conv1x1_bn_counters(M, N) zero words, left zero; workspace: dream_conv1x1_bn_workspace(M, N) bytes.
extern "C" int dream_conv1x1_bnstats_nhwc_f32(const float *x, const float *w_packed, const float *shift, const float *pre_ab, float *y,
                                              long M, int K, int N, int x_stride, const float *gamma, const float *beta,
                                              float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                                              float momentum, float *out_ab, float *save_mean, float *save_invstd, void *workspace,
                                              unsigned *counters, void *stream) {
    DREAM_REQUIRE(x && w_packed && y && gamma && beta && out_ab && save_mean && save_invstd && workspace && counters,
                  "conv1x1_bnstats: null pointer");
    DREAM_REQUIRE(M > 0 && K > 0 && N > 0 && x_stride >= K, "conv1x1_bnstats: bad shape M=%ld K=%d N=%d stride=%d", M, K, N, x_stride);
    DREAM_REQUIRE(K % 32 == 0 && N % 4 == 0 && x_stride % 4 == 0, "conv1x1_bnstats: K %% 32, N %% 4, stride %% 4 (got %d, %d, %d)", K, N, x_stride);
    DREAM_REQUIRE((size_t)M * (size_t)x_stride * 4 < ((size_t)1 << 31) && (size_t)M * (size_t)N * 4 < ((size_t)1 << 33),
                  "conv1x1_bnstats: tensor too large for 32-bit offsets");
    GemmParams p = {};
    p.x = x; p.w = w_packed; p.shift = shift; p.y = y; p.pre_ab = pre_ab;
    const int trows = gemm1x1_rows(M, (N + 63) / 64 * 64);
    p.st = stat_tree_make(workspace, counters, (int)((M + trows - 1) / trows), N);
    p.st_fwd.gamma = gamma; p.st_fwd.beta = beta; p.st_fwd.running_mean = running_mean; p.st_fwd.running_var = running_var;
    p.st_fwd.nbt = num_batches_tracked; p.st_fwd.eps = eps; p.st_fwd.momentum = momentum;
    p.st_fwd.ab = out_ab; p.st_fwd.mean = save_mean; p.st_fwd.invstd = save_invstd;
    return gemm1x1_launch(p, M, K, N, x_stride, pre_ab != nullptr, 1, stream);
}

// Data gradient of a 1x1 conv whose INPUT was the output of a BatchNorm + ReLU, with that BatchNorm's backward reductions in the
// epilogue:  g = (dy . w (+ residual)) * mask, written to g_out, and -- finished inside the launch -- dbeta = sum g,
// dgamma = sum g * (z - mean) * invstd.  mask = [ab[0] z + ab[1] > 0] (the input relu(BN(z)) was never stored: gemm1x1 PRE), or
// [y_act > 0] when y_act is given (a Bottleneck output relu(BN(z) + identity), which IS stored; `residual` then carries the
// gradient of the other branch that meets it: the downsample's or the identity's).  w_packed_t: mode-1 packing; K = channels of
// dy, N = channels of z.
extern "C" int dream_conv1x1_bwd_bnmask_nhwc_f32(const float *dy, const float *w_packed_t, const float *residual, float *g_out, long M,
                                                 int K, int N, int dy_stride, const float *z, const float *ab, const float *y_act,
                                                 const float *mean, const float *invstd, float *dgamma, float *dbeta, void *workspace,
                                                 unsigned *counters, void *stream) {
    DREAM_REQUIRE(dy && w_packed_t && g_out && z && (ab || y_act) && mean && invstd && dgamma && dbeta && workspace && counters,
                  "conv1x1_bwd_bnmask: null pointer");
    DREAM_REQUIRE(M > 0 && K > 0 && N > 0 && dy_stride >= K, "conv1x1_bwd_bnmask: bad shape M=%ld K=%d N=%d stride=%d", M, K, N, dy_stride);
    DREAM_REQUIRE(K % 32 == 0 && N % 4 == 0 && dy_stride % 4 == 0, "conv1x1_bwd_bnmask: K %% 32, N %% 4, stride %% 4 (got %d, %d, %d)", K, N, dy_stride);
    DREAM_REQUIRE((size_t)M * (size_t)dy_stride * 4 < ((size_t)1 << 31) && (size_t)M * (size_t)N * 4 < ((size_t)1 << 33),
                  "conv1x1_bwd_bnmask: tensor too large for 32-bit offsets");
    GemmParams p = {};
    p.x = dy; p.w = w_packed_t; p.y = g_out; p.residual = residual;
    const int trows = gemm1x1_rows(M, (N + 63) / 64 * 64);
    p.st = stat_tree_make(workspace, counters, (int)((M + trows - 1) / trows), N);
    p.st_z = z; p.st_zab = ab; p.st_yact = y_act; p.st_mean = mean; p.st_invstd = invstd;
    p.st_dgamma = dgamma; p.st_dbeta = dbeta;
    return gemm1x1_launch(p, M, K, N, dy_stride, false, 2, stream);
}

extern "C" size_t dream_conv1x1_wgrad_workspace(long M, int Cin, int Cout) {
    if (M <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0) return 0;
    const Wgrad1x1Plan pl = wgrad1x1_plan(M, Cin, Cout);
    return (size_t)pl.nsplit * Cout * Cin * sizeof(float);
}

// x [M][Cin], dy [M][Cdy] (Cdy >= Cout: gradient tensors may carry padded channels) -> dw [Cout][Cin] (= OIHW with 1x1 taps),
// overwritten.  Cin % 64 == 0, Cout % 4 == 0, Cdy % 4 == 0; workspace: dream_conv1x1_wgrad_workspace() bytes.
static int conv1x1_wgrad(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin, int Cout, int Cdy,
                         const float *pre_ab, void *stream);

extern "C" int dream_conv1x1_wgrad_nhwc_f32(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin, int Cout,
                                            int Cdy, void *stream) {
    return conv1x1_wgrad(x, dy, dw, workspace, M, Cin, Cout, Cdy, nullptr, stream);
}

// the same for a conv whose input was relu(pre_ab[0][ci] * x + pre_ab[1][ci]) (train-mode BatchNorm + ReLU folded into the conv's
// loader, never stored): x is the BatchNorm's input
extern "C" int dream_conv1x1_wgrad_pre_nhwc_f32(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin,
                                                int Cout, int Cdy, const float *pre_ab, void *stream) {
    DREAM_REQUIRE(pre_ab != nullptr, "conv1x1 wgrad (pre): null scale / shift");
    return conv1x1_wgrad(x, dy, dw, workspace, M, Cin, Cout, Cdy, pre_ab, stream);
}

static int conv1x1_wgrad(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin, int Cout, int Cdy,
                         const float *pre_ab, void *stream) {
    DREAM_REQUIRE(x && dy && dw && workspace, "conv1x1 wgrad: null pointer");
    DREAM_REQUIRE(M > 0 && Cin > 0 && Cout > 0 && Cdy >= Cout, "conv1x1 wgrad: bad shape");
    DREAM_REQUIRE(Cin % 64 == 0 && Cout % 4 == 0 && Cdy % 4 == 0, "conv1x1 wgrad: Cin %% 64, Cout %% 4, Cdy %% 4 (got %d, %d, %d)", Cin, Cout, Cdy);
    DREAM_REQUIRE((size_t)(M + 64) * (size_t)(Cin > Cdy ? Cin : Cdy) * 4 < ((size_t)1 << 31), "conv1x1 wgrad: tensor too large for 32-bit offsets");
    const Wgrad1x1Plan pl = wgrad1x1_plan(M, Cin, Cout);
    Wgrad1x1Params p;
    p.x = x; p.dy = dy; p.partial = (float *)workspace;
    p.M = (int)M; p.Cin = Cin; p.Cout = Cout; p.Cdy = Cdy;
    p.ncob = (Cout + 63) / 64; p.ncib = Cin / 64;
    p.chunks_per_wave = pl.chunks_per_wave;
    p.pre_ab = pre_ab;
    const dim3 wgrid((unsigned)(p.ncob * p.ncib * pl.nsplit));
    if (pre_ab != nullptr) hipLaunchKernelGGL(wgrad1x1_kernel<true>, wgrid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad1x1_kernel<false>, wgrid, dim3(256), 0, (hipStream_t)stream, p);
    DREAM_LAUNCH_OK();
    const size_t n = (size_t)Cout * Cin;
    size_t rgrid = (n / 4 + 255) / 256;
    if (rgrid > 1024) rgrid = 1024;
    hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, (hipStream_t)stream, (const float *)workspace, dw,
                       pl.nsplit, n);
    DREAM_LAUNCH_OK();
    return 0;
}

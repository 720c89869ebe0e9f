/* dream_hip.h -- C ABI of libdream_hip.so: the MI355X (gfx950) kernels behind DREAM's
 * keypoint belief-map hot path.
 *
 * The reference (NVlabs/DREAM 1.3.0) is pure Python and has no FFI of its own; on this path it
 * reaches native code only through torch ATen operators.  Every entry
 * point below therefore cites the reference call site whose ATen operator (or NumPy/SciPy
 * routine) it replaces.  The reference-side binding a maintainer would add is the ctypes stub in
 * INTEGRATION.md (and, in this repo, dream_amd/_hip.py).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; dream_hip_last_error() then
 *     returns a thread-local message.  No exceptions cross the boundary.
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the host side);
 *     the library never allocates, frees or retains caller memory.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work is enqueued
 *     asynchronously on it; nothing synchronises.
 *   - activations inside the network are NHWC fp32 ("pixel-major": one pixel's channels are
 *     contiguous, so a wavefront's 64 lanes read/write whole 128-256 B lines); the boundary
 *     tensors of the reference API (network input, belief maps, targets) stay NCHW fp32.
 *   - all arithmetic is IEEE fp32 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 == fmaf chains; the Winograd entry
 *     points add and subtract in fp32 around them); the peak path accumulates in fp64 exactly like scipy/NumPy do.
 */
#ifndef DREAM_HIP_H
#define DREAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DREAM_HIP_ABI_VERSION 2

/* conv flags */
#define DREAM_CONV_RELU        1   /* fuse ReLU into the epilogue (reference: nn.ReLU(inplace) after the conv) */
#define DREAM_CONV_UPSAMPLE2X  2   /* input is [B,H/2,W/2,Cin]; nearest x2 upsample fused into the patch load
                                      (reference: nn.Upsample(scale_factor=2), dream/models.py:691,703) */
#define DREAM_CONV_OUT_NCHW    4   /* store the result as NCHW (the belief-map head) */
#define DREAM_CONV_POOL2      16   /* write max-pool-2x2(ReLU(conv)) instead of the conv output: [B,H/2,W/2,Cout] (floor), i.e. the
                                      following nn.MaxPool2d(2) (dream/models.py:589,765-771) fused into the epilogue */
#define DREAM_CONV_RELUMASK   32   /* `residual` is not added but used as a ReLU mask: y = residual > 0 ? conv : 0.  Backward
                                    * data-gradient convs use it to apply the previous layer's ReLU gradient in their
                                    * epilogue (residual = that layer's output), loss.backward() of dream/network.py:335 */
#define DREAM_CONV_RES_AFTER_RELU 64 /* `residual` is added AFTER the ReLU: y = relu(conv + shift) + residual -- the hourglass skip
                                      connections (dream/models.py:774-799: x = relu(conv(..)) ; x = x + x_0_k_d) folded into the
                                      producing conv's epilogue (inference).  Not with DREAM_CONV_RELUMASK / DREAM_CONV_POOL2. */
#define DREAM_CONV_NO_KSPLIT 128  /* dream_conv1x1_nhwc_f32 only: never split the contraction over wavefronts.  The split (1 / 2 / 4) is chosen by the
                                      number of output tiles, i.e. by the ROW count: a caller whose results must not depend on the batch size in the
                                      last bit (the transposed convs run as GEMM + gather, whose N = 16 Cout always gives enough tiles) fixes it */
#define DREAM_CONV_ZEROSTUFF2X 8   /* input is [B,H/2,W/2,Cin] placed at the even positions of a zero [B,H,W,Cin]
                                      grid: with mode-1 packed weights this is ConvTranspose2d(k=3,s=2,p=1,
                                      output_padding=1) (dream/models.py:621-686) */

int         dream_hip_abi_version(void);
const char *dream_hip_last_error(void);
/* number of visible devices / name of device `dev` (into buf); used by smoke tests */
int         dream_hip_device_count(int *count);
int         dream_hip_device_name(int dev, char *buf, size_t buflen);
/* A HIP stream of its own on device `dev` (hipStreamNonBlocking; priority: 0 normal, -1 high, 1 lowest), never destroyed.  The host
 * side keeps a handful per device for the whole process -- the weight-gradient stream, the gradient-exchange stream, the two streams
 * hipGraph captures are taken on -- instead of drawing from torch's pool of 32 round-robin streams per device, where the 33rd
 * torch.cuda.Stream() of a long-running process IS the first one again (a capture begun on it would swallow another thread's replays). */
int         dream_hip_stream_create(int dev, int priority, void **stream);

/* ---- weight packing ------------------------------------------------------------------------
 * OIHW [Cout,Cin,3,3] (torch.nn.Conv2d.weight, dream/models.py:594-615,695-747) ->
 * tap-major [9][RowsPad][ColsPad] fp32, zero padded (rows = channels the conv kernel produces,
 * columns = channels it consumes; the two Pad arguments are RowsPad, ColsPad in that order).
 * mode 0: forward operator, packed[t][o][i] = w[o][i][t] (rows = Cout, cols = Cin).
 * mode 1: data-gradient operator, packed[t][i][o] = w[o][i][8-t] (rows = Cin, cols = Cout): the
 * same conv kernel applied to dL/dy then yields dL/dx (replaces ATen conv backward-input). */
int dream_pack_conv3x3_weight(const float *w_oihw, float *packed, int Cout, int Cin,
                              int CoutPad, int CinPad, int mode, void *stream);
/* inverse of mode 0 for gradients: [9][CoutPad][CinPad] -> OIHW (only the unpadded part) */
int dream_unpack_conv3x3_weight(const float *packed, float *w_oihw, int Cout, int Cin,
                                int CoutPad, int CinPad, void *stream);
/* generic forms: OIHW [Cout,Cin,kh,kw] with ntaps = kh*kw -> [ntaps][RowsPad][ColsPad] (modes as above), and the
 * ConvTranspose2d(k4,s2,p1) weight [Cin,Cout,4,4] (dream/models.py:37-136) -> [4 phases][4 taps][RowsPad][ColsPad]
 * for the sub-pixel decomposition used by dream_conv_transpose4x4s2_nhwc_f32. */
int dream_pack_conv_weight(const float *w_oihw, float *packed, int Cout, int Cin, int ntaps, int RowsPad,
                           int ColsPad, int mode, void *stream);
int dream_pack_convT4x4_weight(const float *wT, float *packed, int Cin, int Cout, int RowsPad, int ColsPad,
                               void *stream);
/* nn.Upsample(scale_factor=2) + nn.Conv2d(k3,s1,p1) of the upsample decoder (dream/models.py:691-733) as ONE transposed
 * conv: w_oihw [Cout,Cin,3,3] -> wT4 [Cin,Cout,4,4] = the ConvTranspose2d(k4,s2,p1) weight that gives the same result
 * (each 4x4 tap is the sum of the 1, 2 or 4 taps of the 3x3 kernel that land on one input pixel), to be packed with
 * dream_pack_convT4x4_weight[_f16x3] and run by dream_conv_transpose4x4s2[_f16x3]_nhwc_f32: 4 MACs per output instead
 * of the 9 of the DREAM_CONV_UPSAMPLE2X form. */
int dream_upsample_conv3x3_weight_as_convT4x4(const float *w_oihw, float *wT4, int Cout, int Cin, void *stream);
size_t dream_conv3x3_cout_pad(int Cout);  /* padded row count the MFMA kernel wants (multiple of 128) */

/* ---- forward operators -----------------------------------------------------------------------
 * conv 3x3 stride 1 pad 1 + bias (+ReLU), NHWC, implicit GEMM on fp32 MFMA.
 * Replaces torch.nn.Conv2d(k=3,s=1,p=1) at dream/models.py:594-615 (encoder), :695-710 (decoder),
 * :736-747 (head).  x: [B,H,W,Cin] (or [B,H/2,W/2,Cin] with UPSAMPLE2X), Cin % 16 == 0;
 * w: packed mode-0 weights [9][CoutPad][Cin]; bias: [Cout] or NULL; y: [B,H,W,Cout] NHWC, or
 * [B,Cout,H,W] with OUT_NCHW. */
int dream_conv3x3_nhwc_f32(const float *x, const float *w_packed, const float *bias, float *y,
                           int B, int H, int W, int Cin, int Cout, int CoutPad, int flags,
                           void *stream);
/* The same torch.nn.Conv2d(k=3,s=1,p=1) (+ReLU, + the MaxPool2d(2) that follows it; dream/models.py:589-615,695-710,
 * 736-747, and the stride-1 3x3 convs of the ResNet-101 bottlenecks behind :22-32) by the Winograd F(2x2,3x3) algorithm
 * on the fp32 matrix cores: 16 instead of 36 multiplications per 2x2 outputs and input channel, fp32 throughout (the
 * transforms only add and subtract; result equal to the direct form up to fp32 round-off, ~1e-6 of the output
 * magnitude).  x [B,H,W,Cin] NHWC, Cin % 16 == 0; u_packed: dream_pack_conv3x3_winograd_weight of the OIHW weight
 * (dream_conv3x3_winograd_weight_floats(rows, cols) floats; mode 0: forward, rows = Cout, cols = Cin; mode 1: the
 * data-gradient operator, rows = Cin, cols = Cout); y = conv * scale[c] + shift[c] (+ residual, or masked by
 * residual > 0 with DREAM_CONV_RELUMASK) (ReLU) (2x2 max-pool); flags: DREAM_CONV_RELU | DREAM_CONV_POOL2 |
 * DREAM_CONV_RELUMASK; scale / shift / residual may be NULL.  Cin: a multiple of 16, at least 32. */
/* 1x1 stride-1 conv as a plain GEMM without LDS (gemm1x1.hip): the ResNet-101 bottleneck convs behind dream/models.py:22-32
 * and, on mode-1 packed weights, their data gradients.  y[M][N] = x[M][:K] . w^T * scale + shift (+ residual) (ReLU); M = B*H*W
 * positions of an NHWC tensor with x_stride (>= K) channels per pixel; K % 32 == 0, N % 4 == 0; flags: DREAM_CONV_RELU. */
size_t dream_conv1x1_weight_floats(int rows, int K);
int dream_conv1x1_set_ksplit(int ks);   /* test hook: 0 = K split by problem size (default), 1 / 2 / 4 = forced */
/* Test / A-B hook: height of a wavefront tile of the 1x1 GEMM: 0 = by problem size (32 rows where 64-row tiles would leave the 1 024 SIMDs
 * with fewer than four each), 64, 32.  Same sums in the same order: the result does not depend on it. */
int dream_conv1x1_set_rows(int rows);
int dream_pack_conv1x1_weight(const float *w_oihw, float *packed, int Cout, int Cin, int mode, void *stream);
int dream_conv1x1_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift, const float *residual,
                           float *y, long M, int K, int N, int x_stride, int flags, void *stream);
/* y = relu(pre_ab[0][k] x + pre_ab[1][k]) . w^T + shift: the conv behind a train-mode BatchNorm + ReLU applied in its loader (pre_ab [2][K]:
 * the BatchNorm's scale / shift from the batch statistics), so that the normalised activation is never stored (round 6: the decoder's head
 * conv in training, dream/models.py:37-136; the trunk uses dream_conv1x1_bnstats_nhwc_f32, which also sums the next BatchNorm's statistics) */
int dream_conv1x1_pre_nhwc_f32(const float *x, const float *w_packed, const float *pre_ab, const float *shift, float *y, long M, int K, int N,
                               int x_stride, void *stream);
/* Weight gradient of the same conv (gemm1x1.hip): x [M][Cin], dy [M][Cdy >= Cout] -> dw [Cout][Cin], overwritten; Cin % 64 == 0,
 * Cout % 4 == 0, Cdy % 4 == 0; deterministic split over positions (fixed-order sums). */
size_t dream_conv1x1_wgrad_workspace(long M, int Cin, int Cout);
int dream_conv1x1_wgrad_nhwc_f32(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin, int Cout, int Cdy,
                                 void *stream);
size_t dream_conv3x3_winograd_weight_floats(int rows, int cols);
int dream_conv3x3_winograd_set_variant(int variant);   /* workgroup width: 0 = by layer (default), 4 / 8 wavefronts = 64 / 128 channels */
int dream_conv3x3_winograd_set_max_workgroups(int n);  /* test hook: size the persistent grid for n co-resident workgroups (0 = the chip) */
int dream_pack_conv3x3_winograd_weight(const float *w_oihw, float *u_packed, int Cout, int Cin, int mode, void *stream);
int dream_conv3x3_winograd_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                    const float *residual, float *y, int B, int H, int W, int Cin, int Cout, int flags,
                                    void *stream);

/* The same convolution by Winograd F(4x4,3x3) (csrc/conv_wino4.hip): 36 multiplications per 4x4 outputs and input channel
 * (2.25 per output; F(2x2,3x3): 4; direct: 9), interpolation points (0, 1, -1, 1/2, -2, inf), IEEE fp32 throughout, weights
 * transformed in fp64 at pack time.  For the stride-1 3x3 layers of dream/models.py:598-615,695-710 behind the first one.
 * Two workgroup shapes, chosen by the number of output channels (rows of the packed operator): more than 64 -- 128 channels per
 * workgroup, Cin a multiple of 32, weights packed [Cin/16][36][rows up to a multiple of 128][16]; up to 64 -- 64 channels per
 * workgroup, Cin a multiple of 16, weights packed [Cin/8][36][64][8].  Packed weights: dream_conv3x3_winograd4_weight_floats(rows,
 * cols) floats (either shape); mode / flags as above. */
size_t dream_conv3x3_winograd4_weight_floats(int rows, int cols);
int dream_pack_conv3x3_winograd4_weight(const float *w_oihw, float *u, int Cout, int Cin, int mode, void *stream);
int dream_conv3x3_winograd4_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                     const float *residual, float *y, int B, int H, int W, int Cin, int Cout, int flags,
                                     void *stream);
/* Training forward of a conv that feeds nn.MaxPool2d(2) (dream/models.py:589,765-771): y_full = relu(conv * scale + shift) [B,H,W,Cout] AND
 * y_pool = maxpool2x2(y_full) [B,H/2,W/2,Cout] from ONE launch (the same values as conv + dream_maxpool2, bit for bit).  flags: DREAM_CONV_RELU. */
int dream_conv3x3_winograd4_pool_both_nhwc_f32(const float *x, const float *u_packed, const float *scale, const float *shift,
                                               float *y_full, float *y_pool, int B, int H, int W, int Cin, int Cout, int flags, void *stream);
int dream_conv3x3_winograd4_set_max_workgroups(int n);
int dream_conv3x3_winograd4_set_stagger(int phases, int percent);   /* A/B hook: start-up stagger of the persistent workgroups (phases <= 1: off; < 0: by DREAM_W4_STAGGER) */
int dream_conv3x3_winograd4_set_channel_block_pinning(int on);   /* A/B hook: output-channel blocks pinned to XCDs (1) or walked by every XCD (0); -1: by DREAM_W4_YMAP (default: pinned) */     /* test hook, as dream_conv3x3_winograd_set_max_workgroups */

/* ---- all packed weight copies of a network in one launch (training: every conv weight changes every step) ----------------------
 * jobs: DEVICE array of njobs entries; src = the weight tensor as the reference stores it (OIHW), dst = the packed copy
 * (dream_conv1x1_weight_floats / dream_conv3x3_winograd_weight_floats / dream_conv3x3_winograd4_weight_floats floats; the zero tail
 * the Winograd kernels over-read is written by the one-tensor entry points and not touched here), mode as theirs. */
enum { DREAM_PACK_CONV1X1 = 0, DREAM_PACK_WINOGRAD2 = 1, DREAM_PACK_WINOGRAD4 = 2,
       /* one output phase of nn.ConvTranspose2d(k4,s2,p1) (src = wT [Cin][Cout][4][4]): cout / cin = rows / columns of the phase's
        * conv, mode = phase | (bwd << 2), dst = that phase's slice of the u4 tensor of dream_pack_convT4x4_winograd[4]_weight */
       DREAM_PACK_CONVT_WINOGRAD2 = 3, DREAM_PACK_CONVT_WINOGRAD4 = 4 };
typedef struct dream_pack_job {
    const float *src;
    float *dst;
    int kind, cout, cin, mode;
} dream_pack_job;
size_t dream_pack_job_bytes(void);
int dream_pack_weights_batched(const dream_pack_job *jobs_device, int njobs, int workgroups_per_job, void *stream);
/* the same with the workgroups dealt out by SIZE (round 6): spans: DEVICE array, one entry per workgroup -- workgroup i runs part `part` of
 * `nparts` of job `job`.  With a fixed number of workgroups per job the decoder's 2048 -> 256 transposed conv (19 M packed floats) ran on
 * as many workgroups as a 64 x 64 1x1 conv and set the launch's length (0.50 ms for ResNet-101 + decoder; by size: see DESIGN.md 4.9). */
typedef struct dream_pack_span {
    int job, part, nparts, reserved;
} dream_pack_span;
size_t dream_pack_span_bytes(void);
int dream_pack_weights_spans(const dream_pack_job *jobs_device, const dream_pack_span *spans_device, int nspans, void *stream);
/* nn.ConvTranspose2d(k4,s2,p1) (+ folded BatchNorm / bias, ReLU) of the ResNet decoder (dream/models.py:37-136) by minimal
 * filtering on the Winograd kernel: each output phase is a 2x2-tap conv = a 3x3 conv whose transformed weights vanish on 7 of
 * the 16 positions: 9 multiplications per 2x2 outputs of a phase instead of 16, same fp32 arithmetic.  x [B,H,W,Cin] ->
 * y [B,2H,2W,Cout]; Cin % 16 == 0, Cin >= 32, Cout > 64; wT [Cin][Cout][4][4]; u4: dream_convT4x4_winograd_weight_floats()
 * floats; scratch: 4*Cout*Cin*9 floats; flags: DREAM_CONV_RELU. */
size_t dream_convT4x4_winograd_weight_floats(int Cout, int Cin);
/* mode 0: forward operator; mode 1: data-gradient operator (u4: dream_convT4x4_winograd_weight_floats(Cin, Cout) floats) */
int dream_pack_convT4x4_winograd_weight(const float *wT, float *u4, float *scratch, int Cin, int Cout, int mode, void *stream);
/* data gradient of that transposed conv (= a 4x4 stride-2 pad-1 conv): dy [B,2H,2W,Cout] -> dx [B,H,W,Cin], Cin > 64, the four
 * phases of dy summed; same nine-position scheme */
int dream_conv4x4s2_winograd_nhwc_f32(const float *dy, const float *u4_mode1, float *dx, int B, int H, int W, int Cout, int Cin,
                                      void *stream);
int dream_conv_transpose4x4s2_winograd_nhwc_f32(const float *x, const float *u4, const float *scale, const float *shift, float *y,
                                                int B, int H, int W, int Cin, int Cout, int flags, void *stream);
/* The four zero-padded 3x3 kernels of the output phases of that transposed conv (bwd = 0: 4 x OIHW [Cout][Cin][3][3]) or of its data
 * gradient (bwd = 1: 4 x [Cin][Cout][3][3]) from wT [Cin][Cout][4][4]: the input of the Winograd packings above and below. */
int dream_convT4x4_phase_weights(const float *wT, float *w3, int Cin, int Cout, int bwd, void *stream);
/* The same transposed conv on the F(4x4,3x3) kernel: the zero-padded phase kernels vanish on 11 of the 36 positions -- 25
 * multiplications per 4x4 outputs of a phase, F(4x4,2x2), where the F(2x2) form takes 36 and the direct sub-pixel form 64.
 * Cout > 64, Cin a multiple of 32; u4: dream_convT4x4_winograd4_weight_floats(Cout, Cin) floats; scratch: 4*Cout*Cin*9 floats. */
size_t dream_convT4x4_winograd4_weight_floats(int Cout, int Cin);
/* mode 0: forward operator; mode 1: data-gradient operator (u4: dream_convT4x4_winograd4_weight_floats(Cin, Cout) floats) */
int dream_pack_convT4x4_winograd4_weight(const float *wT, float *u4, float *scratch, int Cin, int Cout, int mode, void *stream);
/* data gradient of that transposed conv on the same 25-position scheme: dy [B,2H,2W,Cout] -> dx [B,H,W,Cin], Cin > 64, Cout a
 * multiple of 32, the four phases of dy summed */
int dream_conv4x4s2_winograd4_nhwc_f32(const float *dy, const float *u4_mode1, float *dx, int B, int H, int W, int Cout, int Cin,
                                       void *stream);
int dream_conv_transpose4x4s2_winograd4_nhwc_f32(const float *x, const float *u4, const float *scale, const float *shift, float *y,
                                                 int B, int H, int W, int Cin, int Cout, int flags, void *stream);
/* nn.ConvTranspose2d(k3,s2,p1,output_padding 1) (+ReLU) of the deconv decoder (dream/models.py:621-686) by sub-pixel
 * decomposition: four stride-1 launches with 1/2/2/4 taps, no multiplications by zero (the DREAM_CONV_ZEROSTUFF2X form
 * of dream_conv3x3_nhwc_f32 computes the same result with 4x the MACs).  x [B,H,W,Cin] -> y [B,2H,2W,Cout];
 * w_packed: dream_pack_conv3x3_weight(mode 1) of the [Cin,Cout,3,3] ConvTranspose weight; flags: DREAM_CONV_RELU. */
int dream_conv_transpose3x3s2_nhwc_f32(const float *x, const float *w_packed, const float *bias, float *y, int B,
                                       int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream);
/* the same with a tensor of the output's shape added in the epilogue: before the ReLU, or -- DREAM_CONV_RES_AFTER_RELU -- after it
 * (the skip connection behind deconv_0_1, dream/models.py:796-799); flags: DREAM_CONV_RELU | DREAM_CONV_RES_AFTER_RELU */
int dream_conv_transpose3x3s2_res_nhwc_f32(const float *x, const float *w_packed, const float *bias, const float *residual, float *y,
                                           int B, int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream);
/* Data gradient of a k x k (1 | 3) stride-2 pad-k/2 convolution -- the strided 3x3 and 1x1 (downsample) convs of the
 * torchvision ResNet-101 trunk behind dream/models.py:22-32, reached from loss.backward() (dream/network.py:335):
 * dy [B,Hy,Wy,C] -> dx [B,Hx,Wx,Cx], Hx in {2Hy-1, 2Hy}; w_packed_mode1 = dream_pack_conv_weight(mode 1) of the forward
 * weight.  No products with stuffed zeros (k = 3: sub-pixel phases; k = 1: a 1x1 conv written at the even positions). */
int dream_conv2d_s2_bwd_data_nhwc_f32(const float *dy, const float *w_packed_mode1, float *dx, int B, int Hy, int Wy,
                                      int C, int Hx, int Wx, int Cx, int RowsPad, int ksize, void *stream);
/* general form used by the ResNet path (torchvision Bottleneck convs, dream/models.py:22-32,138-148): k x k
 * (k = 1 or 3), stride 1 or 2, pad k/2, NHWC.  H, W = INPUT extent.  Epilogue: y = conv * scale[c] + shift[c]
 * (+ residual) (ReLU): scale/shift carry an eval-mode BatchNorm (dream_bn_fold_f32) or a bias; residual is the
 * Bottleneck identity branch.  Any of scale / shift / residual may be NULL. */
int dream_conv2d_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                          const float *residual, float *y, int B, int H, int W, int Cin, int Cout, int CoutPad,
                          int ksize, int stride, int flags, void *stream);
/* nn.ConvTranspose2d(k=4,s=2,p=1) (+BN+ReLU) of the ResNet decoder (dream/models.py:37-136): [B,H,W,Cin] ->
 * [B,2H,2W,Cout], sub-pixel decomposition (four 2x2 convolutions, no zero multiplications). */
int dream_conv_transpose4x4s2_nhwc_f32(const float *x, const float *w_packed, const float *scale,
                                       const float *shift, float *y, int B, int H, int W, int Cin, int Cout,
                                       int CoutPad, int flags, void *stream);
/* 4x4 stride-2 pad-1 conv: the data gradient of the ConvTranspose2d(4,2,1) above (x = dL/dy of the transposed
 * conv [B,2H,2W,Cout_T], output [B,H,W,Cin_T]; weights packed as 16 taps, see the kernel comment). */
int dream_conv4x4s2_nhwc_f32(const float *x, const float *w_packed, const float *residual, float *y, int B,
                             int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream);
/* eval-mode nn.BatchNorm2d folded into the conv epilogue: scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale (+ conv_bias*scale when the conv has a bias; conv_bias may be NULL). */
int dream_bn_fold_f32(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                      const float *conv_bias, float eps, float *scale, float *shift, int C, void *stream);
/* ResNet stem helpers: im2col of the NCHW image for the 7x7 stride-2 conv (-> [B,Ho,Wo,Kpad] NHWC, k ordered as
 * the OIHW weight flattening, zero padded) so that conv1 runs as a 1x1 MFMA conv; MaxPool2d(3,2,1) on NHWC. */
int dream_im2col_nchw_f32(const float *x, float *y, int B, int C, int H, int W, int KH, int KW, int stride,
                          int pad, int Kpad, void *stream);
int dream_maxpool3s2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream);
/* ---- split-precision ("fp16x3") path: fp32 in/out, fp32-class accuracy on the fp16 matrix cores ------------------
 * Each operand v*2^e = hi + lo (two fp16), product = hi*hi + hi*lo + lo*hi (3 x v_mfma_f32_32x32x16_f16, fp32
 * accumulate).  Activations are split on the fly using the per-tensor max|x| published by the producing kernel
 * (amax side channel: a device uint32 holding the float's bit pattern, zeroed by the caller, atomicMax'ed by the
 * *_amax entry points); weights are pre-split by dream_pack_conv_weight_f16x3.  Same reference call sites as
 * dream_conv2d_nhwc_f32. */
int dream_conv2d_amax_nhwc_f32(const float *x, const float *w_packed, const float *scale, const float *shift,
                               const float *residual, float *y, unsigned *amax_out, int B, int H, int W, int Cin,
                               int Cout, int CoutPad, int ksize, int stride, int flags, void *stream);
int dream_conv3x3_first_nchw_amax_f32(const float *x_nchw, const float *w_oihw, const float *bias,
                                      float *y_nhwc, unsigned *amax_out, int B, int H, int W, int Cin,
                                      int Cout, int relu, void *stream);
int dream_absmax_f32(const float *x, size_t n, unsigned *amax_out, void *stream);
int dream_pack_conv_weight_f16x3(const float *w_oihw, void *hi, void *lo, int *exp_out, unsigned *scratch,
                                 int Cout, int Cin, int ntaps, int RowsPad, int ColsPad, int mode, void *stream);
int dream_conv2d_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi, const void *w_lo,
                                const int *w_exp, const float *scale, const float *shift, const float *residual,
                                float *y, unsigned *amax_out, int B, int H, int W, int Cin, int Cout, int CoutPad,
                                int ksize, int stride, int flags, void *stream);
int dream_pack_convT4x4_weight_f16x3(const float *wT, void *hi, void *lo, int *exp_out, unsigned *scratch, int Cin,
                                     int Cout, int RowsPad, int ColsPad, void *stream);
int dream_conv_transpose4x4s2_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi,
                                             const void *w_lo, const int *w_exp, const float *scale,
                                             const float *shift, float *y, unsigned *amax_out, int B, int H, int W,
                                             int Cin, int Cout, int CoutPad, int flags, void *stream);
/* ConvTranspose2d(k3,s2,p1,output_padding 1) (+ReLU) on the split-precision path (see dream_conv_transpose3x3s2_nhwc_f32);
 * planes from dream_pack_conv_weight_f16x3(mode 1). */
int dream_conv_transpose3x3s2_f16x3_nhwc_f32(const float *x, const unsigned *amax_in, const void *w_hi, const void *w_lo,
                                             const int *w_exp, const float *bias, float *y, unsigned *amax_out, int B,
                                             int H, int W, int Cin, int Cout, int CoutPad, int flags, void *stream);
int dream_conv_f16x3_set_variant(int variant);
/* variant selection for benchmarking: -1 = heuristic; otherwise index into the variant table */
int dream_conv3x3_set_variant(int variant);
int dream_conv3x3_num_variants(void);
const char *dream_conv3x3_variant_name(int variant);

/* first encoder conv: NCHW fp32 image [B,Cin,H,W] (Cin <= 4, what DreamNetwork.inference receives,
 * dream/network.py:503,519) -> NHWC [B,H,W,64*n]; w in OIHW as stored; fused bias (+ReLU).
 * Replaces the fresh Conv2d(3,64,3,1,1) at dream/models.py:592-597. */
int dream_conv3x3_first_nchw_f32(const float *x_nchw, const float *w_oihw, const float *bias,
                                 float *y_nhwc, int B, int H, int W, int Cin, int Cout,
                                 int relu, void *stream);

/* nn.MaxPool2d(2) (dream/models.py:589,765-771): [B,H,W,C] -> [B,H/2,W/2,C] (floor). C % 4 == 0 */
int dream_maxpool2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream);

/* layout helpers (NCHW <-> NHWC, fp32).  nchw_to_nhwc_pad writes Cpad >= C channels per pixel,
 * zero-filling the padding (used to hand the K-channel loss gradient to the MFMA kernels). */
int dream_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);
int dream_nchw_to_nhwc_pad_f32(const float *x, float *y, int B, int C, int H, int W, int Cpad, void *stream);
int dream_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);

/* ---- peak extraction ---------------------------------------------------------------------------
 * Everything after the CNN in DreamNetwork.inference (dream/network.py:529-581), i.e.
 * dream.image_proc.peaks_from_belief_maps (dream/image_proc.py:914-1018: scipy gaussian_filter
 * sigma 3 -> 4-neighbour local max > 0.01 -> 5x5 float64 centroid of the raw map) followed by the
 * one-peak / best-by->=0.25 selection rule.  maps: [N,H,W] fp32 (N = B*K, i.e. NCHW belief maps);
 * scratch: 2*N*H*W floats; keypoints: [N,2] fp32 (x,y) or -999.999; peak_counts: [N] int32 or NULL.
 * Bit-exact with the reference (fp64 accumulation in scipy's/NumPy's order, no FMA contraction). */
int dream_keypoints_from_belief_maps_f32(const float *maps, float *scratch, float *keypoints,
                                         int32_t *peak_counts, int N, int H, int W,
                                         double offset_due_to_upsampling, void *stream);
/* same with the two DreamNetwork attributes callers may change (dream/network.py:189-191, read at :553-560):
 * use_belief_peak_scores (0: several peaks -> no detection) and belief_peak_next_best_score (the fp32 score difference is
 * compared against this double); the entry point above uses the reference's defaults (1, 0.25). */
/* Test / A-B hook (round 6): 1 = the second Gaussian pass fused with the peak scan (default: the smoothed map is never written, a map is
 * scanned by many workgroups), 0 = three kernels (rounds 1-5), -1 = by DREAM_PEAKS_FUSED.  Bit-identical results. */
int dream_peaks_set_fused(int on);
int dream_keypoints_from_belief_maps_rule_f32(const float *maps, float *scratch, float *keypoints,
                                              int32_t *peak_counts, int N, int H, int W,
                                              double offset_due_to_upsampling, int use_belief_peak_scores,
                                              double belief_peak_next_best_score, void *stream);
/* the full peak list of peaks_from_belief_maps, row-major order per map, at most `cap` per map:
 * xy: [N,cap,2] fp64, score: [N,cap] fp32, counts: [N] (true count, may exceed cap). */
int dream_peaks_from_belief_maps_f32(const float *maps, float *scratch, double *xy, float *score,
                                     int32_t *counts, int N, int H, int W, int cap,
                                     double offset_due_to_upsampling, void *stream);
/* scipy.ndimage.gaussian_filter(m, sigma=3) alone (mode reflect, truncate 4): [N,H,W] -> [N,H,W];
 * tmp: N*H*W floats */
int dream_gaussian_sigma3_f32(const float *maps, float *tmp, float *out, int N, int H, int W,
                              void *stream);

/* SoftArgmaxPavlo.forward (dream/spatial_softmax.py:24-95): maps [B*K,H,W], beta [K] (device),
 * scratch N*H*W floats, out [B*K,2] (x,y). */
int dream_softargmax_f32(const float *maps, const float *beta, float *scratch, float *out,
                         int N, int K, int H, int W, float size_mult, void *stream);

/* ---- the steps right before the path, on the device ------------------------------------------
 * ToTensor + Normalize of uint8 RGB frames [B,H,W,3] -> fp32 [B,3,H,W] (dream/datasets.py:87-94; mean3/stdev3 are HOST
 * pointers to 3 floats), and create_belief_map (dream/image_proc.py:866-910) for N = B*K keypoints: kps [N,2] (x,y) FLOAT64
 * (truncated toward zero on the device exactly as the reference's int() does on its float64 coordinates),
 * blob = the (2w+1)^2 Gaussian window computed on the host exactly as the reference does, out [N,H,W]. */
int dream_normalize_u8_hwc_to_chw_f32(const unsigned char *img, float *out, int B, int H, int W,
                                      const float *mean3, const float *stdev3, void *stream);
int dream_create_belief_maps_f64kps_f32(const double *kps, const float *blob, float *out, int N, int H, int W, int w,
                                        void *stream);
/* ABI 1 took fp32 keypoints under this name; fp32 cannot hold the reference's float64 coordinates (57.9999999 is pixel 57,
 * its fp32 rounding pixel 58), so the entry point now FAILS with a message instead of reading fp32 data as float64. */
int dream_create_belief_maps_f32(const float *kps, const float *blob, float *out, int N, int H, int W, int w,
                                 void *stream);
/* Keypoint frames after peak extraction (dream/image_proc.py:135-147 convert_keypoints_to_netin_from_netout,
 * :215-260 convert_keypoints_to_raw_from_netin; call sites dream/network.py:480-488, dream/analysis.py:219-232):
 * kps_netout [N,2] fp32 (x, y) -> kps_netin, kps_raw [N,2] float64, the reference's arithmetic in its order, sentinels
 * included.  mode 0 = preprocessing "none" (raw = netin); mode 1 = raw = netin / in * span + origin with span/origin =
 * the raw resolution and 0 ("resize", "shrink") or the cropped resolution and corner ("shrink-and-crop"). */
int dream_convert_keypoints_f64(const float *kps_netout, double *kps_netin, double *kps_raw, int N,
                                double out_w, double out_h, double in_w, double in_h, double span_w, double span_h,
                                double origin_x, double origin_y, int mode, void *stream);

/* ---- training operators --------------------------------------------------------------------------
 * MSELoss(mean) forward + gradient (dream/network.py:260-261,359; loss.backward() at :335):
 * loss_sum[0] = sum((o-t)^2) (caller divides by n_total); grad = 2*(o-t)/n_total (grad may be null).  The sum is
 * accumulated in fp64 per workgroup into `workspace` (dream_loss_workspace(n) bytes) and reduced in a fixed order,
 * so the loss is bit-reproducible. */
size_t dream_loss_workspace(size_t n);
int dream_mse_fwd_bwd_f32(const float *out, const float *target, float *grad, float *loss_sum, void *workspace,
                          size_t n, double n_total, void *stream);
/* SmoothL1Loss(beta 1, mean) = the "huber" loss type (dream/network.py:262-263,290-291): same contract as the MSE */
int dream_smoothl1_fwd_bwd_f32(const float *out, const float *target, float *grad, float *loss_sum, void *workspace,
                               size_t n, double n_total, void *stream);
/* elementwise ReLU backward on NHWC tensors: dx = dy * (y > 0) (inplace allowed) */
int dream_relu_bwd_f32(const float *dy, const float *y, float *dx, size_t n, void *stream);
/* MaxPool2d(2) backward: dy [B,H/2,W/2,C], x [B,H,W,C] (forward input), dx [B,H,W,C];
 * gradient goes to the first maximal element in window scan order (ATen semantics). */
int dream_maxpool2_bwd_nhwc_f32(const float *dy, const float *x, float *dx,
                                int B, int H, int W, int C, void *stream);
/* same, continued through the ReLU that produced x (VGG blocks end conv -> ReLU -> MaxPool2d, dream/models.py:589-615):
 * dx = maxpool_bwd(dy) * (x > 0) in one pass. */
int dream_maxpool2_relu_bwd_nhwc_f32(const float *dy, const float *x, float *dx, int B, int H, int W, int C, void *stream);
/* nearest x2 upsample backward: dy [B,H,W,C] -> dx [B,H/2,W/2,C] = sum of the 2x2 block */
/* Round 6: the pixels a stride-2 1x1 convolution reads, gathered -- y[b,i,j,:] = x[b,2i,2j,:], y is [B,(H+1)/2,(W+1)/2,C] -- and the
 * transpose (x[b,y,x,:] = ys[b,y/2,x/2,:] at even (y, x), zeros elsewhere; x is [B,H,W,C], overwritten).  With them the three stride-2
 * downsample convs of the ResNet-101 trunk (/root/reference/dream/models.py:22-32 -> torchvision Bottleneck.downsample, reached from
 * network.py:310-335) run on the 1x1 GEMM entry points above in the forward, weight-gradient and data-gradient direction.  C % 4 == 0. */
int dream_subsample2_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, void *stream);
int dream_scatter2_nhwc_f32(const float *ys, float *x, int B, int H, int W, int C, void *stream);
/* Round 6: the patches a 3x3 stride-2 pad-1 convolution reads as rows of 9 C columns (col [B,Ho,Wo,9 C], column t C + c = tap t = 3 ky + kx,
 * channel c; Ho = (H - 1) / 2 + 1; zeros outside the image) and the transpose (dx [B,H,W,C] = the sum of the patch entries that read each
 * pixel, fixed order).  ResNet-101's layer4.0.conv2 (/root/reference/dream/models.py:22-32 -> torchvision Bottleneck.conv2, stride 2) has too
 * few output pixels at 16 frames per GPU for the direct kernel to fill the chip; over these rows it runs on the 1x1 GEMM entry points. */
int dream_im2col3s2_nhwc_f32(const float *x, float *col, int B, int H, int W, int C, void *stream);
int dream_col2im3s2_nhwc_f32(const float *col, float *dx, int B, int H, int W, int C, void *stream);
/* Round 6: nn.ConvTranspose2d(k4, s2, p1) on small maps as a 1x1 GEMM + a gather (/root/reference/dream/models.py:37-136, the ResNet decoder's
 * first layer): g [B,H,W,16 C] holds, per input pixel, its sixteen tap contributions (column (4 ky + kx) C + c = dream_conv1x1_nhwc_f32 with
 * the weight matrix [16 C][Cin]); z [B,2H,2W,C] = shift (the bias, or null) + the (at most four) contributions that land on each output
 * pixel when scale is null (training), (their sum) * scale + shift with a scale (evaluation: the folded BatchNorm); flags: DREAM_CONV_RELU. */
int dream_col2im4s2_nhwc_f32(const float *g, const float *scale, const float *shift, float *z, int B, int H, int W, int C, int flags, void *stream);
int dream_upsample2_bwd_nhwc_f32(const float *dy, float *dx, int B, int H, int W, int C, void *stream);
/* conv3x3 weight+bias gradient: x [B,H,W,Cin] (or half-res with UPSAMPLE2X), dy [B,H,W,Cout] NHWC
 * -> dw_packed [9][CoutPad][Cin] (mode-0 layout, overwritten), dbias [Cout] (overwritten).
 * workspace: dream_conv3x3_wgrad_workspace() bytes. Deterministic split-K reduction. */
size_t dream_conv3x3_wgrad_workspace(int B, int H, int W, int Cin, int CoutPad);
int dream_conv3x3_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, float *dbias,
                                 void *workspace, int B, int H, int W, int Cin, int Cout,
                                 int CoutPad, int flags, void *stream);
/* The same weight gradient in the Winograd F(2x2,3x3) domain (16 instead of 36 multiplications per 2x2 outputs, fp32
 * throughout): dU_p = sum_tiles (A dY A^T)_p x (B^T d B)_p on the fp32 matrix cores, dW = G^T dU G, deterministic split-K.
 * x [B,H,W,Cin], dy [B,H,W,Cdy] (Cdy >= Cout, both multiples of 16), Cin % 64 == 0 -> dw_oihw [Cout,Cin,3,3] (OIHW,
 * overwritten); the bias gradient is dream_channel_sum_nhwc_f32(dy).  Replaces ATen conv backward-weight reached from
 * loss.backward() (dream/network.py:335) for the stride-1 3x3 convs of dream/models.py:598-615,695-710,736-747,22-32. */
size_t dream_conv3x3_wgrad_winograd_workspace(int B, int H, int W, int Cin, int Cout);
int dream_conv3x3_wgrad_winograd_nhwc_f32(const float *x, const float *dy, float *dw_oihw, void *workspace, int B, int H,
                                          int W, int Cin, int Cout, int Cdy, int flags, void *stream);
/* flags: 0 or DREAM_CONV_UPSAMPLE2X (x is [B,H/2,W/2,Cin]: weight gradient of the conv that follows nn.Upsample(2),
 * dream/models.py:691-710; channels a multiple of 64) */
/* ... with the bias gradient dbias [Cout] = column sums of dy accumulated in the kernel's dy loader (no separate pass over dy).
 * Only where dream_conv3x3_wgrad_winograd_fuses_bias(Cin, Cout, Cdy) returns 1 (Cin, Cout multiples of 64); dbias == NULL: as
 * dream_conv3x3_wgrad_winograd_nhwc_f32.  Same workspace. */
int dream_conv3x3_wgrad_winograd_fuses_bias(int Cin, int Cout, int Cdy);
int dream_conv3x3_wgrad_winograd_bias_nhwc_f32(const float *x, const float *dy, float *dw_oihw, float *dbias, void *workspace,
                                               int B, int H, int W, int Cin, int Cout, int Cdy, int flags, void *stream);
int dream_conv3x3_wgrad_winograd_set_version(int version);   /* test / A-B hook: 0 = by shape (default), 1 = register-only kernel */
/* general forms for the ResNet path (1x1 / 3x3, stride 1 / 2) and the 4x4 transposed conv */
size_t dream_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int CoutPad, int ksize, int stride);
int dream_conv2d_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, float *dbias, void *workspace,
                                int B, int H, int W, int Cin, int Cout, int CoutPad, int ksize, int stride,
                                int flags, void *stream);
size_t dream_convT4x4_wgrad_workspace(int B, int H, int W, int CinPad, int Cout);
int dream_convT4x4_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, void *workspace, int B,
                                  int H, int W, int Cin, int CinPad, int Cout, void *stream);
/* The same gradient by minimal filtering F(2x2,2x2) (round 6; replaces ATen's conv-transpose backward-weight behind loss.backward(),
 * /root/reference/dream/network.py:335, for the decoder of /root/reference/dream/models.py:37-136): every output phase of the
 * transposed conv is a 2 x 2-tap conv whose Winograd-domain weight gradient lives on nine of the sixteen positions of the 4 x 4
 * domain: 9 multiplications per 2 x 2 outputs of a phase instead of the direct form's 16.  x [B,H,W,Cin], dy [B,2H,2W,Cout] ->
 * dwT [Cin][Cout][4][4] (the module's layout, no unpack step), dbias [Cout] or NULL (column sums of dy from the same launch).
 * Cin % 64 == 0 and Cout % 64 == 0 (dream_convT4x4_wgrad_winograd_applies); workspace: dream_convT4x4_wgrad_winograd_workspace() bytes. */
int dream_convT4x4_wgrad_winograd_applies(int Cin, int Cout);
size_t dream_convT4x4_wgrad_winograd_workspace(int B, int H, int W, int Cin, int Cout);
int dream_convT4x4_wgrad_winograd_nhwc_f32(const float *x, const float *dy, float *dwT, float *dbias, void *workspace, int B,
                                           int H, int W, int Cin, int Cout, void *stream);
/* nn.Upsample(2) + Conv2d(3x3) is such a transposed conv (dream_upsample_conv3x3_weight_as_convT4x4): its 3x3 weight gradient from the
 * transposed conv's, dw[co][ci][r][c] = sum of dwT[ci][co][ky][kx] over ky in {2 - r, 3 - r}, kx in {2 - c, 3 - c}. */
int dream_upsample_conv3x3_wgrad_from_convT4x4(const float *dwT, float *dw_oihw, int Cin, int Cout, void *stream);
size_t dream_convT_wgrad_workspace(int B, int H, int W, int CinPad, int Cout, int ksize);
int dream_convT_wgrad_nhwc_f32(const float *x, const float *dy, float *dw_packed, void *workspace, int B,
                               int H, int W, int Cin, int CinPad, int Cout, int ksize, void *stream);
/* Width of the weight-gradient launches PLANNED BY THE CALLING HOST THREAD from now on, in per cent of the chip (5 .. 100, default 100):
 * every weight-gradient entry point splits its contraction until ~256 .. 1024 workgroups exist; at `percent` < 100 it stops at that
 * fraction -- fewer, longer workgroups and proportionally less split-K partial traffic.  For launches that run on a second stream BESIDE
 * the data-gradient chain (training at small per-GPU batches, dream_amd/models.py _SideStream): a narrow launch leaves the other
 * compute units to the chain instead of taking the whole chip in bursts.  The matching *_workspace() functions follow the same
 * thread's setting: query and launch under one setting.  Thread-local, so concurrent replica threads cannot disturb each other. */
int dream_wgrad_set_width(int percent);
/* A/B switch between the weight-gradient kernel's register blockings: -1 = heuristic, 0 = always 64-row tiles. */
int dream_wgrad_set_variant(int variant);
/* [ntaps][RowsPad][ColsPad] -> [Rows][Cols][ntaps] (OIHW / ConvTranspose [Cin,Cout,kh,kw]) */
int dream_unpack_conv_weight(const float *packed, float *w, int Rows, int Cols, int ntaps, int RowsPad, int ColsPad,
                             void *stream);
/* train-mode nn.BatchNorm2d (torchvision ResNet + decoder BNs, dream/models.py:22-32,37-136), NHWC, npix = B*H*W.
 * forward: batch statistics -> save_mean/save_invstd, running stats update (momentum, unbiased var),
 * y = bn(x) (+residual) (ReLU).  backward: g = dy*(y_act>0 if relu); dgamma, dbeta; dx; optional g_out = g
 * (gradient of the Bottleneck identity branch).  workspace: dream_bn_workspace(C) bytes. */
size_t dream_bn_workspace(int C);
int dream_bn_train_fwd_nhwc_f32(const float *x, const float *gamma, const float *beta, const float *residual,
                                float *y, float *save_mean, float *save_invstd, float *running_mean,
                                float *running_var, long long *num_batches_tracked, void *workspace,
                                size_t npix, int C, float eps, float momentum, int relu, void *stream);
int dream_bn_train_bwd_nhwc_f32(const float *x, const float *dy, const float *y_act, const float *gamma,
                                const float *save_mean, const float *save_invstd, float *dx, float *g_out,
                                float *dgamma, float *dbeta, void *workspace, size_t npix, int C, int relu,
                                void *stream);
int dream_channel_sum_nhwc_f32(const float *x, float *out, void *workspace, size_t npix, int C, void *stream);
/* ---- round 4: the same train-mode BatchNorm without its separate passes (torchvision Bottleneck conv -> BN -> ReLU chains,
 * dream/models.py:22-32 via network.py:328-364).  Statistics are published as the affine map y = ab[0][c] * z + ab[1][c]
 * (ab[0] = gamma * invstd, ab[1] = beta - mean * ab[0]; every kernel evaluates BN + ReLU as max(fmaf(a, z, b), 0)) and are
 * FINISHED INSIDE the launch that sums them (a two-level ticket tree per 64-channel slab: the last producer of a group of partial
 * rows adds the group, the last group adds the groups -- fixed order: deterministic; no finalize launch, no fence, nobody spins).  `counters`: dream_bn_stats_counters(C) zero 32-bit words, zero again when
 * the launch ends; a buffer must not be shared by launches that may run concurrently.  workspace: dream_bn_stats_workspace(C)
 * bytes (dream_conv1x1_bn_workspace(M, N) for the GEMM forms). */
size_t dream_bn_stats_workspace(int C);
int dream_bn_stats_counters(int C);
int dream_bn_stats_set_pixels_per_row(int px);           /* A/B hook: pixels one workgroup sums into a partial row (default 128) */
/* batch statistics of z [npix][C] in one launch (+ running statistics update as nn.BatchNorm2d in train mode) */
int dream_bn_stats_nhwc_f32(const float *z, const float *gamma, const float *beta, float *running_mean, float *running_var,
                            long long *num_batches_tracked, float eps, float momentum, float *out_ab, float *save_mean,
                            float *save_invstd, void *workspace, unsigned *counters, size_t npix, int C, void *stream);
/* y = ab[0] z + ab[1] (+ residual) (ReLU) */
int dream_bn_apply_ab_nhwc_f32(const float *z, const float *ab, const float *residual, float *y, size_t npix, int C, int relu,
                               void *stream);
/* backward: g = dy masked (mask 0: as is; 1: y_act > 0; 2: ab[0] z + ab[1] > 0, the ReLU mask recomputed from the BN input);
 * dbeta = sum g, dgamma = sum g * xhat in one launch; then dz = gamma * invstd * (g - dbeta / N - xhat * dgamma / N),
 * g_out (optional) = g */
int dream_bn_bwd_stats_nhwc_f32(const float *z, const float *dy, const float *y_act, const float *ab, const float *save_mean,
                                const float *save_invstd, float *dgamma, float *dbeta, void *workspace, unsigned *counters,
                                size_t npix, int C, int mask, void *stream);
int dream_bn_bwd_apply_nhwc_f32(const float *z, const float *dy, const float *y_act, const float *ab, const float *gamma,
                                const float *save_mean, const float *save_invstd, const float *dgamma, const float *dbeta,
                                float *dz, float *g_out, size_t npix, int C, int mask, void *stream);
/* 1x1 conv (dream_conv1x1_nhwc_f32's GEMM) with the BatchNorm on either side folded in:
 *   y = relu(pre_ab[0][k] x + pre_ab[1][k]) . w^T + shift   (pre_ab null: y = x . w^T + shift)   -- the previous BN + ReLU applied
 *   in the loader, its output never stored; and the batch statistics of y (for the BN that follows) summed in the epilogue and
 *   finished in the launch: out_ab / save_mean / save_invstd / running statistics as dream_bn_stats_nhwc_f32. */
size_t dream_conv1x1_bn_workspace(long M, int N);
int dream_conv1x1_bn_counters(long M, int N);            /* zero 32-bit words the two GEMM forms below need */
int dream_conv1x1_bnstats_nhwc_f32(const float *x, const float *w_packed, const float *shift, const float *pre_ab, float *y,
                                   long M, int K, int N, int x_stride, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, long long *num_batches_tracked, float eps,
                                   float momentum, float *out_ab, float *save_mean, float *save_invstd, void *workspace,
                                   unsigned *counters, void *stream);
/* data gradient of a 1x1 conv whose input was the output of a BN + ReLU, with that BN's backward reductions in the epilogue:
 * g_out = (dy . w (+ residual)) * mask and, finished in the launch, dbeta = sum g, dgamma = sum g * (z - mean) * invstd.
 * mask = [ab[0] z + ab[1] > 0] (input never stored), or [y_act > 0] when y_act is given (a Bottleneck output relu(BN(z) + identity);
 * residual = the gradient of the other branch meeting there).  w_packed_t: mode-1 packing, K = channels of dy, N = channels of z. */
int dream_conv1x1_bwd_bnmask_nhwc_f32(const float *dy, const float *w_packed_t, const float *residual, float *g_out, long M, int K,
                                      int N, int dy_stride, const float *z, const float *ab, const float *y_act, const float *mean,
                                      const float *invstd, float *dgamma, float *dbeta, void *workspace, unsigned *counters,
                                      void *stream);
/* The same two foldings for the stride-1 3x3 convs on the Winograd F(2x2,3x3) kernel (conv2 of torchvision's Bottleneck behind
 * dream/models.py:22-32): the batch statistics of y = conv3x3(x) + shift summed in the kernel's epilogue, and the data gradient
 * g_out = conv3x3(dy; mode-1 weights) * [ab[0] z + ab[1] > 0] with dbeta / dgamma of the masked BatchNorm -- one persistent
 * workgroup = one fp64 partial row, finished in the launch by the ticket tree.  Cout % 64 == 0.  Workspace / counters depend on the
 * kernel's grid: query AFTER dream_conv3x3_winograd_set_variant / _set_max_workgroups. */
size_t dream_conv3x3_winograd_bn_workspace(int B, int H, int W, int Cout);
int dream_conv3x3_winograd_bn_counters(int B, int H, int W, int Cout);
int dream_conv3x3_winograd_bnstats_nhwc_f32(const float *x, const float *u_packed, const float *shift, float *y, int B, int H, int W,
                                            int Cin, int Cout, const float *gamma, const float *beta, float *running_mean,
                                            float *running_var, long long *num_batches_tracked, float eps, float momentum,
                                            float *out_ab, float *save_mean, float *save_invstd, void *workspace, unsigned *counters,
                                            void *stream);
int dream_conv3x3_winograd_bwd_bnmask_nhwc_f32(const float *dy, const float *u_packed_t, const float *z, float *g_out, int B, int H,
                                               int W, int Cin, int Cout, const float *ab, const float *mean, const float *invstd,
                                               float *dgamma, float *dbeta, void *workspace, unsigned *counters, void *stream);
/* weight gradient of a 1x1 conv whose input was relu(pre_ab[0][ci] x + pre_ab[1][ci]) (x = the BN input) */
int dream_conv1x1_wgrad_pre_nhwc_f32(const float *x, const float *dy, float *dw, void *workspace, long M, int Cin, int Cout,
                                     int Cdy, const float *pre_ab, void *stream);
/* the same pool for training: also stores which element of its 3x3 window won (uint8 0..8 = 3 dy + dx, the first maximum in scan
 * order: ATen's max_pool2d_with_indices), so that the backward pass compares one byte per window instead of recomputing nine-way
 * arg-maxima; idx: [B,Ho,Wo,C] bytes */
int dream_maxpool3s2_idx_nhwc_f32(const float *x, float *y, unsigned char *idx, int B, int H, int W, int C, void *stream);
int dream_maxpool3s2_idx_bwd_nhwc_f32(const float *dy, const unsigned char *idx, float *dx, int B, int H, int W, int C, void *stream);
/* MaxPool2d(3,2,1) backward (ATen first-max semantics; overlapping windows accumulate) */
int dream_maxpool3s2_bwd_nhwc_f32(const float *dy, const float *x, float *dx, int B, int H, int W, int C, void *stream);
/* many tensors gathered into one flat buffer by one launch (the optimizer's flat gradient buffer, torch.optim's per-parameter
 * gradients: network.py:335 -> :337).  chunks: device array, chunk c = copy n floats from srcs[job] + src_off to dst (<= 65536 floats
 * per chunk keeps the workgroups even); srcs: device array of the step's source pointers. */
typedef struct dream_copy_chunk {
    float *dst;
    unsigned src_off;
    unsigned n;
    int job;
    int reserved;
} dream_copy_chunk;
size_t dream_copy_chunk_bytes(void);
int dream_multi_copy_f32(const void *srcs, const void *chunks, int nchunks, void *stream);
/* dst += src (gradient accumulation where two branches meet) */
int dream_add_inplace_f32(float *dst, const float *src, size_t n, void *stream);
/* dst[i] = src[i] with a kernel launch -- `tensor.clone()` for code that may be captured into a hipGraph (ATen copies contiguous tensors
 * with hipMemcpyAsync, which becomes a memcpy NODE there; round 6 found a replayed memset node out of order on this runtime and the
 * captured steps hold kernel nodes only since): the gradient copies at the skip connections / between the stages of dream/models.py:774-827 */
int dream_copy_f32(float *dst, const float *src, size_t n, void *stream);

/* ---- gradient exchange of the single-process data-parallel path (SURVEY.md 8b "allreduce_grads", 8e; what
 * torch.nn.DataParallel's ReduceAddCoalesced + the next forward's parameter broadcast do for dream/network.py:244-256,335) ----
 * bufs[i]: flat fp32 gradient buffer of replica i on GPU devices[i] (identical layout, `count` floats), streams[i]: the HIP
 * stream replica i's backward was enqueued on.  Afterwards every buffer holds the element-wise sum, stream-ordered.  Distinct
 * devices: ONE RCCL all-reduce over xGMI (ncclCommInitAll group, cached per device list).  A list that repeats one device
 * (rehearsal of N replicas on one GPU) is summed locally.  dream_allreduce_uses_rccl: which of the two a list takes. */
int dream_allreduce_sum_f32(int ndev, const int *devices, void *const *bufs, size_t count, void *const *streams);
int dream_allreduce_uses_rccl(int ndev, const int *devices);
/* out = a + b: the encoder skip tensors joining the decoder (`x_0_5 + x_0_4_d`, `y_0_5 + x_0_3_d`, ...,
 * dream/models.py:774-807).  amax_out (optional) receives the bit pattern of max|out| for the split-precision kernel. */
int dream_add_f32(const float *a, const float *b, float *out, size_t n, unsigned *amax_out, void *stream);
/* Multi-stage hourglass input (dream/models.py:487-493,:500-553): torch.cat([x, F.interpolate(y_prev, scale_factor=up)], 1)
 * written straight into the NHWC layout the MFMA first conv of stage s>1 reads: img NCHW [B,Ci,H,W], maps NCHW
 * [B,K,H/up,W/up] (up = 4, or 1 when the decoder is full-resolution) -> out [B,H,W,Cpad], channels >= Ci+K zero.
 * _bwd: g NHWC [B,H,W,Cpad] -> dmaps NCHW [B,K,H/up,W/up] = sums of channels Ci..Ci+K-1 over each up x up block
 * (accumulate != 0: added to dmaps). */
int dream_stage_input_nhwc_f32(const float *img_nchw, const float *maps_nchw, float *out_nhwc, int B, int H, int W,
                               int Ci, int K, int up, int Cpad, unsigned *amax_out, void *stream);
int dream_stage_input_bwd_f32(const float *g_nhwc, float *dmaps_nchw, int B, int H, int W, int Ci, int K, int up,
                              int Cpad, int accumulate, void *stream);
/* first-layer weight gradient: x NCHW [B,Cin,H,W], dy NHWC [B,H,W,Cout] -> dw OIHW, dbias */
size_t dream_conv3x3_first_wgrad_workspace(int B, int H, int W, int Cin, int Cout);
int dream_conv3x3_first_wgrad_f32(const float *x_nchw, const float *dy_nhwc, float *dw_oihw,
                                  float *dbias, void *workspace, size_t workspace_bytes,
                                  int B, int H, int W, int Cin, int Cout, void *stream);
/* torch.optim.Adam / SGD step with PyTorch defaults (dream/network.py:666-685) on one flat fp32
 * buffer: p, g, m, v of n elements; step is the 1-based step count. */
int dream_adam_step_f32(float *p, const float *g, float *m, float *v, size_t n, float lr,
                        float beta1, float beta2, float eps, int step, void *stream);
int dream_sgd_step_f32(float *p, const float *g, size_t n, float lr, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DREAM_HIP_H */

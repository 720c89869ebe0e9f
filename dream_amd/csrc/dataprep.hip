// The two steps immediately BEFORE the hot path, moved onto the device so that a training
// step no longer uploads 245.8 MB of normalised fp32 frames + 35.8 MB of fp32 targets per batch of 128:
//   * ToTensor + Normalize(mean, stdev) of uint8 RGB frames  (/root/reference/dream/datasets.py:87-94,
//     /root/reference/dream/network.py:449-459): out = ((u8 / 255) - mean) / stdev in IEEE fp32, HWC -> CHW;
//   * create_belief_map  (/root/reference/dream/image_proc.py:866-910): a (4*sigma+1)^2 Gaussian blob stamped at
//     the int()-truncated keypoint when the window (plus one) fits, otherwise an all-zero map.  The blob values
//     are computed on the host in float64 with NumPy exactly as the reference does and cast to fp32 there, so
//     the device result is bit-identical to `torch.tensor(create_belief_map(...)).float()`.
// Both are pure streaming kernels (HBM-bound: 3 B in / 12 B out per pixel, resp. 4 B out per map pixel).
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

__global__ void __launch_bounds__(256) normalize_u8_kernel(const unsigned char *img, float *out, int B, int H, int W,
                                                           float m0, float m1, float m2, float s0, float s1, float s2) {
    const size_t npix = (size_t)B * H * W;
    const size_t hw = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const size_t b = i / hw, p = i - b * hw;
        const unsigned char *src = img + i * 3;
        float *dst = out + b * 3 * hw + p;
        dst[0] = (((float)src[0] / 255.0f) - m0) / s0;
        dst[hw] = (((float)src[1] / 255.0f) - m1) / s1;
        dst[2 * hw] = (((float)src[2] / 255.0f) - m2) / s2;
    }
}

__global__ void __launch_bounds__(256) belief_maps_kernel(const double *kps, const float *blob, float *out, int N, int H, int W,
                                                          int w) {
    const int side = 2 * w + 1;
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H);
        const size_t n = r / H;
        // Python's int() on the reference's float64 coordinate: truncation toward zero, done on the float64 value
        // (57.9999999 must stay pixel 57; an fp32 upload would round it to 58).  Clamped so the cast is defined.
        const int u = (int)fmin(fmax(kps[n * 2 + 0], -1.0e9), 1.0e9), v = (int)fmin(fmax(kps[n * 2 + 1], -1.0e9), 1.0e9);
        float val = 0.0f;
        if (u - w >= 0 && u + w + 1 < W && v - w >= 0 && v + w + 1 < H) {
            const int dx = x - u, dy = y - v;
            if (dx >= -w && dx <= w && dy >= -w && dy <= w) val = blob[(dy + w) * side + (dx + w)];
        }
        out[i] = val;
    }
}

inline unsigned sgrid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    return (unsigned)(g ? g : 1);
}
// Keypoint frame conversion right after peak extraction (dream/image_proc.py:135-147 net-output -> net-input,
// :215-260 net-input -> raw image): two affine maps in float64 with the reference's operation order (divide, then
// multiply, then add), applied to every row including the -999.999 sentinels, as the reference does.
//   netin = k / out_res * in_res            raw = netin                      (mode 0: "none")
//                                           raw = netin / in_res * span + origin   (mode 1: resize / shrink /
//                                                 shrink-and-crop; span / origin = raw or cropped resolution / corner)
__global__ void __launch_bounds__(256) convert_keypoints_kernel(const float *kps, double *netin, double *raw, int N,
                                                                double ow, double oh, double iw, double ih, double sw,
                                                                double sh, double x0, double y0, int mode) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double nx = dmul(ddiv((double)kps[2 * i], ow), iw), ny = dmul(ddiv((double)kps[2 * i + 1], oh), ih);
    netin[2 * i] = nx;
    netin[2 * i + 1] = ny;
    if (mode == 0) {
        raw[2 * i] = nx;
        raw[2 * i + 1] = ny;
    } else {
        raw[2 * i] = dadd(dmul(ddiv(nx, iw), sw), x0);
        raw[2 * i + 1] = dadd(dmul(ddiv(ny, ih), sh), y0);
    }
}

}  // namespace

extern "C" int dream_normalize_u8_hwc_to_chw_f32(const unsigned char *img, float *out, int B, int H, int W,
                                                 const float *mean3, const float *stdev3, void *stream) {
    DREAM_REQUIRE(img && out && mean3 && stdev3 && B > 0 && H > 0 && W > 0, "normalize_u8: bad arguments");
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(sgrid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W,
                       mean3[0], mean3[1], mean3[2], stdev3[0], stdev3[1], stdev3[2]);
    DREAM_LAUNCH_OK();
    return 0;
}

// kps: [N,2] (x, y) FLOAT64 device (the reference truncates the float64 coordinate); blob: [(2w+1)^2] fp32 device (host-computed, see above); out: [N,H,W]
extern "C" int dream_create_belief_maps_f32(const float *, const float *, float *, int, int, int, int, void *) {
    DREAM_REQUIRE(false, "dream_create_belief_maps_f32 (fp32 keypoints, ABI 1) was withdrawn: pass float64 keypoints to "
                         "dream_create_belief_maps_f64kps_f32");
    return 1;
}

extern "C" int dream_create_belief_maps_f64kps_f32(const double *kps, const float *blob, float *out, int N, int H, int W, int w,
                                                   void *stream) {
    DREAM_REQUIRE(kps && blob && out && N > 0 && H > 0 && W > 0 && w >= 0, "create_belief_maps: bad arguments");
    hipLaunchKernelGGL(belief_maps_kernel, dim3(sgrid((size_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, kps, blob, out, N, H,
                       W, w);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_convert_keypoints_f64(const float *kps_netout, double *kps_netin, double *kps_raw, int N,
                                           double out_w, double out_h, double in_w, double in_h, double span_w,
                                           double span_h, double origin_x, double origin_y, int mode, void *stream) {
    DREAM_REQUIRE(kps_netout && kps_netin && kps_raw && N > 0 && out_w > 0 && out_h > 0 && in_w > 0 && in_h > 0 &&
                  (mode == 0 || mode == 1), "convert_keypoints: bad arguments");
    hipLaunchKernelGGL(convert_keypoints_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, kps_netout, kps_netin,
                       kps_raw, N, out_w, out_h, in_w, in_h, span_w, span_h, origin_x, origin_y, mode);
    DREAM_LAUNCH_OK();
    return 0;
}

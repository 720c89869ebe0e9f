// SoftArgmaxPavlo.forward (/root/reference/dream/spatial_softmax.py:24-95):
//   7x7 average pool (stride 1, zero pad 3, always /49) -> subtract the per-map max ->
//   exp(beta_k * .) -> normalise by (sum + 1e-8) -> expected column (x) and row (y) index.
// Two streaming kernels: (1) the pooled map, (2) one workgroup per map doing the max and the three
// sums with wave-shuffle reductions (64-lane xor butterflies, then 4 partials through LDS).
// fp32 throughout, like the reference; summation order differs from ATen's, so parity is
// tolerance-based (1e-4 on the coordinates, tests/test_softargmax_gpu.py).
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

__global__ void __launch_bounds__(256) avgpool7_kernel(const float *maps, float *out, int N, int H, int W) {
    const size_t total = (size_t)N * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const size_t r = idx / W;
        const int y = (int)(r % H);
        const float *base = maps + (r / H) * (size_t)H * W;
        float acc = 0.0f;
        for (int dy = -3; dy <= 3; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -3; dx <= 3; ++dx) {
                const int xx = x + dx;
                if (xx >= 0 && xx < W) acc += base[(size_t)yy * W + xx];
            }
        }
        out[idx] = acc / 49.0f;      // count_include_pad=True: the divisor is always 49
    }
}

DREAM_DEVICE float block_reduce(float v, float *s4, bool is_max) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o = lane_xor(v, m);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3])) : (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

__global__ void __launch_bounds__(256) softargmax_kernel(const float *pooled, const float *beta, float *out,
                                                         int K, int H, int W, float size_mult) {
    __shared__ float s4[4];
    const int n = blockIdx.x;
    const float *p = pooled + (size_t)n * H * W;
    const int total = H * W;
    const float bk = beta[n % K];
    float mx = -__builtin_huge_valf();
    for (int i = threadIdx.x; i < total; i += 256) mx = fmaxf(mx, p[i]);
    mx = block_reduce(mx, s4, true);
    float se = 0.0f, sx = 0.0f, sy = 0.0f;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int y = i / W, x = i - y * W;
        const float e = expf(bk * (p[i] - mx));
        se += e;
        sx += e * ((float)x * size_mult);
        sy += e * ((float)y * size_mult);
    }
    se = block_reduce(se, s4, false);
    sx = block_reduce(sx, s4, false);
    sy = block_reduce(sy, s4, false);
    if (threadIdx.x == 0) {
        const float den = se + 1e-8f;
        out[(size_t)n * 2 + 0] = sx / den;
        out[(size_t)n * 2 + 1] = sy / den;
    }
}
}  // namespace

extern "C" int dream_softargmax_f32(const float *maps, const float *beta, float *scratch, float *out, int N, int K,
                                    int H, int W, float size_mult, void *stream) {
    DREAM_REQUIRE(maps && beta && scratch && out && N > 0 && K > 0 && H > 0 && W > 0, "softargmax: bad arguments");
    const size_t total = (size_t)N * H * W;
    size_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(avgpool7_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, maps, scratch, N, H, W);
    DREAM_LAUNCH_OK();
    hipLaunchKernelGGL(softargmax_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const float *)scratch, beta, out,
                       K, H, W, size_mult);
    DREAM_LAUNCH_OK();
    return 0;
}

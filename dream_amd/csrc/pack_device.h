// Device bodies of the weight-packing kernels that a training step runs once per conv layer (the optimizer changed every weight):
// shared by the one-tensor kernels of gemm1x1.hip / conv_wino.hip / conv_wino4.hip and by the batched kernel of pack_batched.hip,
// which packs every such tensor of a network in ONE launch (blockIdx.y = job).  `blk` / `nblk`: this workgroup's index and the
// number of workgroups that share the tensor (grid-stride over its elements); 256 threads per workgroup.
#pragma once
#include <dream_cdna4.h>

namespace dream_pack {

// 1x1 conv as a GEMM (gemm1x1.hip): w [Cout][Cin] (mode 0: rows = Cout, k = Cin) or, for the data gradient, the same tensor read as
// [k = Cout][rows = Cin] (mode 1) -> packed [K/16][RowsPad][16] with physical row 64 c + 16 n + j = logical row 64 c + 4 j + n
DREAM_DEVICE void conv1x1(const float *w, float *packed, int Cout, int Cin, int mode, int blk, int nblk) {
    const int rows = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
    const int RowsPad = (rows + 63) / 64 * 64;
    const size_t total = (size_t)(K / 16) * RowsPad * 16;
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
        const int e = (int)(i & 15);
        const size_t rest = i >> 4;
        const int prow = (int)(rest % RowsPad), t = (int)(rest / RowsPad);
        const int b64 = prow >> 6, n = (prow >> 4) & 3, j = prow & 15;
        const int row = b64 * 64 + 4 * j + n, k = 16 * t + e;
        float v = 0.0f;
        if (row < rows) v = mode == 0 ? w[(size_t)row * Cin + k] : w[(size_t)k * Cin + row];
        packed[i] = v;
    }
}

// the 3x3 filter of (row n, input channel k) of the packed operator: OIHW (mode 0) or, for the data gradient, IOHW with flipped taps
DREAM_DEVICE void filter3x3(const float *w, int Cin, int rows, int mode, int n, int k, double g[3][3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = 0.0f;
            if (n < rows) {
                v = (mode == 0) ? w[(((size_t)n * Cin + k) * 3 + a) * 3 + b]
                                : w[(((size_t)k * Cin + n) * 3 + (2 - a)) * 3 + (2 - b)];
            }
            g[a][b] = (double)v;
        }
}

// The same for phase (a, b) of nn.ConvTranspose2d(k4, s2, p1), written as a 3x3 conv with a zero-padded kernel (conv_wino.hip,
// convT4x4_phase_kernels: the four OIHW kernels w3[phase] it materialises -- here read straight from wT [CinT][CoutT][4][4]).
// pm = phase | (bwd << 2).  Forward (bwd 0): row n = output channel co, k = input channel ci, tap (r, c) = offset (r - 1, c - 1):
// ky = a + 1 - 2 dy, kx = b + 1 - 2 dx, used where the offset points up / left for a (b) = 0, down / right for 1.  Data gradient
// (bwd 1): row n = ci (the gradient conv's output channel), k = co: ky = a - 1 + 2 r, kx = b - 1 + 2 c.
DREAM_DEVICE void filterT3x3(const float *wT, int rows, int cols, int pm, int n, int k, double g[3][3]) {
    const int a = (pm >> 1) & 1, b = pm & 1, bwd = (pm >> 2) & 1;
    const int CoutT = bwd ? cols : rows;
    const int ci = bwd ? n : k, co = bwd ? k : n;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
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
            float v = 0.0f;
            if (used && n < rows) v = wT[(((size_t)ci * CoutT + co) * 4 + ky) * 4 + kx];
            g[r][c] = (double)v;
        }
}

// Winograd F(2x2,3x3) (conv_wino.hip): U = G g G^T in fp64, rounded once, [cols/16][16 positions][RowsPad][16]; RowsPad = rows
// rounded up to 128; position row 3 negated (the kernel computes -V[3][.])
// CONVT: w is a transposed conv's weight, (Cout, Cin) = the phase conv's (rows, cols), mode = phase | (bwd << 2) (filterT3x3)
template <bool CONVT = false>
DREAM_DEVICE void winograd2(const float *w, float *u, int Cout, int Cin, int mode, int blk, int nblk) {
    const int rows = CONVT ? Cout : (mode == 0 ? Cout : Cin), cols = CONVT ? Cin : (mode == 0 ? Cin : Cout);
    const int RowsPad = (rows + 127) / 128 * 128;
    const size_t total = (size_t)(cols / 16) * RowsPad * 16;
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
        const int kk = (int)(i % 16);
        const size_t rest = i / 16;
        const int n = (int)(rest % RowsPad);
        const int ch = (int)(rest / RowsPad);
        double g[3][3];
        if (CONVT) filterT3x3(w, rows, cols, mode, n, ch * 16 + kk, g);
        else filter3x3(w, Cin, rows, mode, n, ch * 16 + kk, g);
        double t[4][3];                                             // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {                               // (G g) G^T
            const double rr[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b)
                u[(((size_t)ch * 16 + (a * 4 + b)) * RowsPad + n) * 16 + kk] = (float)(a == 3 ? -rr[b] : rr[b]);
        }
    }
}

// Winograd F(4x4,3x3) (conv_wino4.hip), interpolation points (0, 1, -1, 1/2, -2, inf): [cols/K][36 positions][RowsPad][K] with
// (K, RowsPad) = (16, rows up to a multiple of 128), or (8, 64) when rows <= 64 (the kernel's narrow workgroup shape)
template <bool CONVT = false>
DREAM_DEVICE void winograd4(const float *w, float *u, int Cout, int Cin, int mode, int blk, int nblk) {
    const double G[6][3] = {{1.0, 0.0, 0.0},
                            {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0},
                            {-1.0 / 3.0, 1.0 / 3.0, -1.0 / 3.0},
                            {-16.0 / 15.0, -8.0 / 15.0, -4.0 / 15.0},
                            {1.0 / 15.0, -2.0 / 15.0, 4.0 / 15.0},
                            {0.0, 0.0, 1.0}};
    const int rows = CONVT ? Cout : (mode == 0 ? Cout : Cin), cols = CONVT ? Cin : (mode == 0 ? Cin : Cout);
    const int K = rows <= 64 ? 8 : 16;
    const int RowsPad = rows <= 64 ? 64 : (rows + 127) / 128 * 128;
    const size_t total = (size_t)(cols / K) * RowsPad * K;
    for (size_t i = (size_t)blk * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
        const int kk = (int)(i % K);
        const size_t rest = i / K;
        const int n = (int)(rest % RowsPad);
        const int ch = (int)(rest / RowsPad);
        double g[3][3];
        if (CONVT) filterT3x3(w, rows, cols, mode, n, ch * K + kk, g);
        else filter3x3(w, Cin, rows, mode, n, ch * K + kk, g);
        double t[6][3];                                             // G g
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) t[a][b] = G[a][0] * g[0][b] + G[a][1] * g[1][b] + G[a][2] * g[2][b];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {                           // (G g) G^T
                const double v = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
                u[(((size_t)ch * 36 + (a * 6 + b)) * RowsPad + n) * K + kk] = (float)v;
            }
    }
}

}  // namespace dream_pack

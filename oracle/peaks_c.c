/* oracle/peaks_c.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's peak path, used as the single-core CPU baseline and as an
 * independent cross-check of oracle/peaks.py (bit-for-bit, tests/test_oracle_peaks.py):
 *   peaks_from_belief_maps      /root/reference/dream/image_proc.py:914-1018
 *   keypoint selection rule     /root/reference/dream/network.py:546-577
 * The Gaussian is scipy.ndimage.gaussian_filter(sigma=3) (third-party, restated: two 1-D passes,
 * axis 0 then 1, 'reflect' boundary, double accumulation in the symmetric-kernel order, float32 store
 * after each pass); the centroid is np.average in float64 with NumPy's pairwise summation.
 * The 25 filter taps are passed in by the caller (computed with NumPy exactly as scipy does).
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off oracle/peaks_c.c -o oracle/libpeaks_c.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define R 12

static int reflect(int i, int n) {
    int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

static void pass(const float *in, float *out, int H, int W, int axis, const double *w) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int l = axis == 0 ? y : x, len = axis == 0 ? H : W;
            double acc = (double)in[y * W + x] * w[R];
            for (int i = -R; i < 0; ++i) {
                int a = reflect(l + i, len), b = reflect(l - i, len);
                double va = axis == 0 ? (double)in[a * W + x] : (double)in[y * W + a];
                double vb = axis == 0 ? (double)in[b * W + x] : (double)in[y * W + b];
                acc += (va + vb) * w[R + i];
            }
            out[y * W + x] = (float)acc;
        }
}

void dream_oracle_gaussian_sigma3(const float *in, float *tmp, float *out, int H, int W, const double *taps25) {
    pass(in, tmp, H, W, 0, taps25);
    pass(tmp, out, H, W, 1, taps25);
}

static double pairwise25(const double *a) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = (a[j] + a[8 + j]) + a[16 + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    return res + a[24];
}

static void centroid(const float *ori, int H, int W, int x, int y, double off, double *cx, double *cy) {
    double wv[25], pj[25], pi[25];
    for (int j = -2; j <= 2; ++j)
        for (int i = -2; i <= 2; ++i) {
            int t = (j + 2) * 5 + (i + 2);
            int inb = y + i >= 0 && y + i < H && x + j >= 0 && x + j < W;
            double w = inb ? (double)ori[(y + i) * W + (x + j)] : 0.0;
            wv[t] = w;
            pj[t] = (inb ? (double)(x + j) : 0.0) * w;
            pi[t] = (inb ? (double)(y + i) : 0.0) * w;
        }
    double scl = pairwise25(wv);
    if (scl == 0.0) { *cx = (double)x + off; *cy = (double)y + off; }
    else { *cx = pairwise25(pj) / scl + off; *cy = pairwise25(pi) / scl + off; }
}

/* maps [N][H][W] -> keypoints [N][2] (float32, -999.999 for "no detection"), counts [N]. Returns 0. */
int dream_oracle_keypoints(const float *maps, int N, int H, int W, double offset, const double *taps25,
                           float *keypoints, int *counts) {
    float *tmp = (float *)malloc(sizeof(float) * H * W), *sm = (float *)malloc(sizeof(float) * H * W);
    if (!tmp || !sm) return 1;
    for (int n = 0; n < N; ++n) {
        const float *ori = maps + (size_t)n * H * W;
        dream_oracle_gaussian_sigma3(ori, tmp, sm, H, W, taps25);
        int cnt = 0, best = -1;
        float s1 = -INFINITY, s2 = -INFINITY;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float v = sm[y * W + x];
                float up = y > 0 ? sm[(y - 1) * W + x] : 0.0f, dn = y + 1 < H ? sm[(y + 1) * W + x] : 0.0f;
                float lf = x > 0 ? sm[y * W + x - 1] : 0.0f, rt = x + 1 < W ? sm[y * W + x + 1] : 0.0f;
                if (v >= up && v >= dn && v >= lf && v >= rt && v > 0.01f) {
                    float sc = ori[y * W + x];
                    ++cnt;
                    if (sc > s1) { s2 = s1; s1 = sc; best = y * W + x; }
                    else if (sc > s2) s2 = sc;
                }
            }
        float kx = -999.999f, ky = -999.999f;
        if (cnt == 1 || (cnt > 1 && (float)(s1 - s2) >= 0.25f)) {
            double cx, cy;
            centroid(ori, H, W, best % W, best / W, offset, &cx, &cy);
            kx = (float)cx; ky = (float)cy;
        }
        keypoints[2 * n] = kx; keypoints[2 * n + 1] = ky;
        if (counts) counts[n] = cnt;
    }
    free(tmp); free(sm);
    return 0;
}

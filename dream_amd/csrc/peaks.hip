// Belief-map peak extraction: everything DreamNetwork.inference does after the CNN
// (/root/reference/dream/network.py:529-581), i.e. dream.image_proc.peaks_from_belief_maps
// (/root/reference/dream/image_proc.py:914-1018) and the per-keypoint selection rule.
//
// Bit-exactness contract (tests/test_peaks_gpu.py): the smoothed map equals
// scipy.ndimage.gaussian_filter(m, sigma=3) bit-for-bit and the centroids equal NumPy's float64
// np.average bit-for-bit.  That pins the arithmetic order:
//   * Gaussian: two 1-D passes (axis 0 then axis 1), each pixel accumulated in fp64 as
//       acc = x[l]*w[c];  for i = -12..-1: acc += (x[l+i] + x[l-i]) * w[c+i]
//     (scipy's symmetric-kernel branch of NI_Correlate1D), 'reflect' (half-sample symmetric)
//     boundary, result stored to fp32 after EACH pass.  The 13 distinct taps are the doubles scipy
//     computes for sigma=3/radius=12, hard-coded as hex literals.
//   * centroid: 25 products in fp64 laid out [col offset][row offset], summed with NumPy's
//     8-lane pairwise scheme, divided, offset added, then rounded once to fp32.
//   * no FMA contraction anywhere on these paths (dmul/dadd/ddiv = __dmul_rn/__dadd_rn/__ddiv_rn).
// The Gaussian passes stage their inputs in LDS (25 taps per output from there); fp64 rate is irrelevant at 25 taps / pixel.
#include <stdlib.h>
#include <dream_cdna4.h>
#include "common.h"
#include "../../include/dream_hip.h"

namespace {

constexpr int R = 12;
// scipy _gaussian_kernel1d(sigma=3, order=0, radius=12)[0..12]  (index 12 = centre tap)
__constant__ double kTaps[13] = {
    0x1.763a210dfb306p-15, 0x1.4fbe39149e277p-13, 0x1.0d8a5ad43c165p-11, 0x1.8345966f69518p-10,
    0x1.f1e9915139406p-9,  0x1.1e6bccad344bap-7,  0x1.26defcaeb0202p-6,  0x1.0fa58939b528fp-5,
    0x1.bfde9c12bec92p-5,  0x1.4a614d1afd337p-4,  0x1.b42a57d56c0bep-4,  0x1.01a25f86eb137p-3,
    0x1.105a329f98197p-3};

DREAM_DEVICE int reflect_index(int i, int n) {
    // half-sample symmetric extension  (d c b a | a b c d | d c b a), any distance
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// ... on maps of at least 2R x 2R pixels (SMALL = false) ONE reflection covers every index a stored output depends on (|distance outside|
// <= R <= n); the positions staged beyond that feed outputs outside the map only and are clamped to a valid address.  Smaller maps take the
// general form (two integer divisions per index).
template <bool SMALL>
DREAM_DEVICE int reflect_at(int i, int n) {
    if (SMALL) return reflect_index(i, n);
    const int q = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
    return q < 0 ? 0 : (q >= n ? n - 1 : q);
}

// Element `idx` of a map whose base pointer is wavefront-uniform: a 32-bit byte offset beside the scalar base (global_load ... saddr) instead of
// a 64-bit multiply-add per access -- the launchers require H x W < 2^30.  (PMC, resnet_f at 32 frames: the Gaussian kernels are VALU-bound,
// 62 % / 96 % busy, and two fifths of their vector instructions were index arithmetic; profiles/r06_pmc_peaks.txt.)
DREAM_DEVICE float ld_map(const float *base, int idx) { return *(const float *)((const char *)base + (unsigned)(idx << 2)); }
DREAM_DEVICE void st_map(float *base, int idx, float v) { *(float *)((char *)base + (unsigned)(idx << 2)) = v; }

// One 1-D pass.  Workgroup = 256 threads on a 64-column strip of one map (3-D grid: no 64-bit index division per pixel); lanes
// run along x.  The inputs of the strip are staged ONCE in LDS, reflected where the strip leaves the map, and every output takes
// its 25 taps from there: the row pass (AXIS 1) stages 4 rows x (64 + 24) columns for 4 x 64 outputs, the column pass (AXIS 0)
// (16 + 24) rows x 64 columns for 16 x 64 outputs (four per thread) -- 1.4 / 2.5 coalesced loads per output where the unstaged
// version issued 25 (it ran at 0.26 TB/s on the 400 x 400 maps of DREAM-resnet-F: bound by its load instructions, not by memory).
// Same values, same order of additions as before: bit-identical.
constexpr int GT = 16;            // output rows per workgroup of the column pass
template <int AXIS, bool SMALL>
__global__ void __launch_bounds__(256) gauss_pass_kernel(const float *in, float *out, int N, int H, int W) {
    __shared__ float tile[AXIS == 0 ? (GT + 2 * R) * 64 : 4 * (64 + 2 * R)];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * (AXIS == 0 ? GT : 4);
    const float *base = in + (size_t)blockIdx.z * H * W;
    float *obase = out + (size_t)blockIdx.z * H * W;
    auto tap_sum = [&](auto at) {                     // acc = x[l] w[c]; for i = -12..-1: acc += (x[l+i] + x[l-i]) w[c+i]
        double acc = dmul((double)at(0), kTaps[R]);
#pragma unroll
        for (int i = -R; i < 0; ++i) acc = dadd(acc, dmul(dadd((double)at(i), (double)at(-i)), kTaps[R + i]));
        return (float)acc;
    };
    if (AXIS == 0) {
        const int x = x0 + tx;
        const bool xin = x < W;
        {   // staged row r = map row y0 - R + r, reflected.  A wavefront stages rows wv, wv + 4, ..: the row arithmetic is scalar, and ALL
            // ten loads are in flight before the first LDS store (the rolled loop waited for each load before it issued the next: ten
            // memory latencies in sequence per workgroup -- 71 % of the wave-cycles parked, profiles/r06_pmc_peaks.txt)
            const int wv = wave_index();
            static_assert((GT + 2 * R) % 4 == 0, "rows per wavefront");
            float v[(GT + 2 * R) / 4];
#pragma unroll
            for (int k = 0; k < (GT + 2 * R) / 4; ++k) {
                const int q = reflect_at<SMALL>(y0 - R + wv + 4 * k, H);
                v[k] = xin ? ld_map(base, q * W + x) : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < (GT + 2 * R) / 4; ++k) tile[(wv + 4 * k) * 64 + tx] = v[k];
        }
        __syncthreads();
        // round 6: a thread owns FOUR CONSECUTIVE rows of its column: their 4 x 25 taps are 28 staged values, read and converted to
        // fp64 once (was: rows ty, ty + 4, .. -- 100 LDS reads and conversions for the same four outputs).  Same sums, same order.
        static_assert(GT == 16, "four rows per thread x four waves");
        double d[4 + 2 * R];
#pragma unroll
        for (int j = 0; j < 4 + 2 * R; ++j) d[j] = (double)tile[(4 * ty + j) * 64 + tx];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = y0 + 4 * ty + k;
            double acc = dmul(d[k + R], kTaps[R]);
#pragma unroll
            for (int i = -R; i < 0; ++i) acc = dadd(acc, dmul(dadd(d[k + R + i], d[k + R - i]), kTaps[R + i]));
            if (xin && y < H) st_map(obase, y * W + x, (float)acc);
        }
    } else {
        const int y = y0 + ty;
        {   // staged column c = map column x0 - R + c, reflected: lanes 0-63 the first 64, lanes 0-23 the rest; both loads before the stores
            const bool yin = y < H, more = tx < 2 * R;
            const float v0 = yin ? ld_map(base, y * W + reflect_at<SMALL>(x0 - R + tx, W)) : 0.0f;
            const float v1 = (yin && more) ? ld_map(base, y * W + reflect_at<SMALL>(x0 - R + 64 + tx, W)) : 0.0f;
            tile[ty * (64 + 2 * R) + tx] = v0;
            if (more) tile[ty * (64 + 2 * R) + 64 + tx] = v1;
        }
        __syncthreads();
        const int x = x0 + tx;
        if (x < W && y < H) st_map(obase, y * W + x, tap_sum([&](int i) { return tile[ty * (64 + 2 * R) + tx + R + i]; }));
    }
}

DREAM_DEVICE double pairwise_sum25(const double *a) {
    // NumPy pairwise summation for n = 25 (< 128): 8 strided partial sums, balanced combine, tail
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = dadd(dadd(a[j], a[8 + j]), a[16 + j]);
    const double res = dadd(dadd(dadd(r[0], r[1]), dadd(r[2], r[3])), dadd(dadd(r[4], r[5]), dadd(r[6], r[7])));
    return dadd(res, a[24]);
}

// image_proc.py:961-998
DREAM_DEVICE void centroid_5x5(const float *ori, int H, int W, int x, int y, double offset, double *cx, double *cy) {
    double wv[25], pj[25], pi[25];
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
#pragma unroll
        for (int i = -2; i <= 2; ++i) {
            const int t = (j + 2) * 5 + (i + 2);
            const bool inb = (y + i >= 0) && (y + i < H) && (x + j >= 0) && (x + j < W);
            const double w = inb ? (double)ori[(size_t)(y + i) * W + (x + j)] : 0.0;
            wv[t] = w;
            pj[t] = dmul(inb ? (double)(x + j) : 0.0, w);
            pi[t] = dmul(inb ? (double)(y + i) : 0.0, w);
        }
    }
    const double scl = pairwise_sum25(wv);
    if (scl == 0.0) {            // np.average raises ZeroDivisionError -> integer peak (image_proc.py:995-998)
        *cx = dadd((double)x, offset);
        *cy = dadd((double)y, offset);
    } else {
        *cx = dadd(ddiv(pairwise_sum25(pj), scl), offset);
        *cy = dadd(ddiv(pairwise_sum25(pi), scl), offset);
    }
}

struct Top2 {
    float s1, s2;
    int i1;
};
DREAM_DEVICE Top2 top2_merge(Top2 a, Top2 b) {
    Top2 o;
    const bool a_first = (a.s1 > b.s1) || (a.s1 == b.s1 && (unsigned)a.i1 <= (unsigned)b.i1);
    o.s1 = a_first ? a.s1 : b.s1;
    o.i1 = a_first ? a.i1 : b.i1;
    o.s2 = fmaxf(a_first ? b.s1 : a.s1, fmaxf(a.s2, b.s2));
    return o;
}

// One workgroup per map.  LIST: also emit every peak (row-major order) up to `cap`.
template <bool LIST>
__global__ void __launch_bounds__(256) peaks_kernel(const float *maps, const float *smooth, float *keypoints,
                                                    int32_t *counts, double *xy, float *score, int H, int W,
                                                    int cap, double offset, int use_scores, double next_best) {
    __shared__ int s_wave_cnt[4];
    __shared__ float s_s1[4], s_s2[4];
    __shared__ int s_i1[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ori = maps + (size_t)n * H * W;
    const float *sm = smooth + (size_t)n * H * W;
    const int total = H * W;
    const float thresh = 0.01f;                       // image_proc.py:925 (compared in fp32)
    const float NEG_INF = -__builtin_huge_valf();
    Top2 best = {NEG_INF, NEG_INF, -1};
    int running = 0;                                   // peaks found in earlier chunks (uniform)
    if (!LIST) {
        // Only the count and the two best scores are wanted: no ordered compaction, hence no ballot and no barrier per chunk -- a
        // plain strided scan (the top-2 merge is independent of the order it is applied in: ties go to the smaller index), the
        // per-thread counts summed at the end.  (The chunked form below took 0.69 ms on 544 maps of 400 x 400.)
        int mine_cnt = 0;
        for (int idx = tid; idx < total; idx += 256) {
            const int y = idx / W, x = idx - y * W;
            const float v = sm[idx];
            const float up = y > 0 ? sm[idx - W] : 0.0f, down = y + 1 < H ? sm[idx + W] : 0.0f;
            const float left = x > 0 ? sm[idx - 1] : 0.0f, right = x + 1 < W ? sm[idx + 1] : 0.0f;
            if ((v >= up) && (v >= down) && (v >= left) && (v >= right) && (v > thresh)) {
                const Top2 mine = {ori[idx], NEG_INF, idx};
                best = top2_merge(best, mine);
                ++mine_cnt;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mine_cnt += lane_xor(mine_cnt, m);
        if (lane == 0) s_wave_cnt[wave] = mine_cnt;
        __syncthreads();
        running = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
    }
    for (int base = 0; LIST && base < total; base += 256) {
        const int idx = base + tid;
        bool is_peak = false;
        int x = 0, y = 0;
        if (idx < total) {
            y = idx / W;
            x = idx - y * W;
            const float v = sm[idx];
            const float up = y > 0 ? sm[idx - W] : 0.0f, down = y + 1 < H ? sm[idx + W] : 0.0f;
            const float left = x > 0 ? sm[idx - 1] : 0.0f, right = x + 1 < W ? sm[idx + 1] : 0.0f;
            is_peak = (v >= up) && (v >= down) && (v >= left) && (v >= right) && (v > thresh);
        }
        const unsigned long long mask = wave_ballot(is_peak);
        const int rank_in_wave = popcount64(mask & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave_cnt[wave] = popcount64(mask);
        __syncthreads();
        int wave_off = 0, chunk_total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wave_off += s_wave_cnt[w];
            chunk_total += s_wave_cnt[w];
        }
        if (is_peak) {
            const float sc = ori[idx];
            const Top2 mine = {sc, NEG_INF, idx};
            best = top2_merge(best, mine);
            if (LIST) {
                const int rank = running + wave_off + rank_in_wave;
                if (rank < cap) {
                    double cx, cy;
                    centroid_5x5(ori, H, W, x, y, offset, &cx, &cy);
                    xy[((size_t)n * cap + rank) * 2 + 0] = cx;
                    xy[((size_t)n * cap + rank) * 2 + 1] = cy;
                    score[(size_t)n * cap + rank] = sc;
                }
            }
        }
        running += chunk_total;
        __syncthreads();                                // s_wave_cnt is rewritten next chunk
    }
    // block-wide top-2 reduction
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Top2 other;
        other.s1 = lane_xor(best.s1, m);
        other.s2 = lane_xor(best.s2, m);
        other.i1 = lane_xor(best.i1, m);
        best = top2_merge(best, other);
    }
    if (lane == 0) { s_s1[wave] = best.s1; s_s2[wave] = best.s2; s_i1[wave] = best.i1; }
    __syncthreads();
    if (tid == 0) {
        Top2 b = {s_s1[0], s_s2[0], s_i1[0]};
#pragma unroll
        for (int w = 1; w < 4; ++w) { const Top2 o = {s_s1[w], s_s2[w], s_i1[w]}; b = top2_merge(b, o); }
        if (counts) counts[n] = running;
        if (keypoints) {
            // network.py:546-577 : one peak -> it; several -> (if use_belief_peak_scores) the best one iff
            // best - second >= belief_peak_next_best_score: the fp32 difference of two fp32 scores against the Python
            // float attribute (0.25 by default, network.py:189-191)
            float kx = -999.999f, ky = -999.999f;
            const bool accept = (running == 1) || (running > 1 && use_scores != 0 && (double)(b.s1 - b.s2) >= next_best);
            if (accept) {
                const int y = b.i1 / W, x = b.i1 - y * W;
                double cx, cy;
                centroid_5x5(ori, H, W, x, y, offset, &cx, &cy);
                kx = (float)cx;
                ky = (float)cy;
            }
            keypoints[(size_t)n * 2 + 0] = kx;
            keypoints[(size_t)n * 2 + 1] = ky;
        }
    }
}

// ---- round 6: the second Gaussian pass and the peak scan in one kernel (the keypoint rule only needs counts and the two best peaks) -------
// A workgroup owns FT output rows x 64 columns of one map: it stages FT + 2 rows of the column pass' result (one halo row above and
// below; 64 + 2 + 24 columns, reflected), runs the row pass for those (FT + 2) x 66 pixels into LDS -- the same tap_sum, the same fp32
// rounding: the values ARE the smoothed map's -- and tests its FT x 64 pixels against their four neighbours there.  The smoothed map is
// never written (a third of the three-kernel form's traffic) and a 416 x 416 map is scanned by 210 workgroups instead of one.  A
// workgroup leaves (best score, second best, index of the best, count); finish_kernel merges a map's records (the merge is
// associative and commutative: ties go to the smaller index) and applies network.py:546-577.
constexpr int FT = 13;            // (FT + 2) rows x 17 groups of four columns = 255 work items of the row pass: one per thread
int g_peaks_fused = -1;          // dream_peaks_set_fused: -1 = by DREAM_PEAKS_FUSED (default on), 0 / 1 = forced
struct PeakPart {
    float s1, s2;
    int i1, count;
};

template <bool SMALL>
__global__ void __launch_bounds__(256) gauss_row_peaks_kernel(const float *maps, const float *colpass, PeakPart *parts, int H, int W) {
    constexpr int SW = 68 + 2 * R;                                   // staged columns: map columns x0 - 1 - R .. x0 + 66 + R (the last two: padding
                                                                     // of the 17th group of four; 92 floats: rows stay 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float stage[(FT + 2) * SW];
    __shared__ float sm[(FT + 2) * 68];
    __shared__ PeakPart s_part[4];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, lane = tx;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * FT, n = blockIdx.z;
    const float *ori = maps + (size_t)n * H * W;
    const float *base = colpass + (size_t)n * H * W;
    {   // a wavefront stages whole rows (row and row test wavefront-uniform, no division): lanes 0-63 the first 64 columns, lanes 0-27 the rest
        // ... and all of a wavefront's (up to eight) loads are in flight before its first LDS store
        const int q0 = reflect_at<SMALL>(x0 - 1 - R + tx, W), q1 = reflect_at<SMALL>(x0 - 1 - R + 64 + tx, W);
        const int wv = wave_index();
        const bool more = tx < SW - 64;
        constexpr int NR = (FT + 2 + 3) / 4;
        float v0[NR], v1[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = wv + 4 * k, y = y0 - 1 + r;
            const bool yin = r < FT + 2 && y >= 0 && y < H;            // (rows outside the map are never looked at)
            v0[k] = yin ? ld_map(base, y * W + q0) : 0.0f;
            v1[k] = (yin && more) ? ld_map(base, y * W + q1) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = wv + 4 * k;
            if (r < FT + 2) {
                stage[r * SW + tx] = v0[k];
                if (more) stage[r * SW + 64 + tx] = v1[k];
            }
        }
    }
    __syncthreads();
    // the row pass for (FT + 2) rows x 68 columns (66 needed), a thread = four consecutive columns of one row: 28 staged values (seven
    // 16-byte LDS reads), converted to fp64 once, feed its 4 x 25 taps.  Same sums, same order as gauss_pass_kernel<1>.
    if (tid < (FT + 2) * 17) {
        const int r = tid / 17, c0 = 4 * (tid - r * 17);             // smoothed pixels (row y0 - 1 + r, columns x0 - 1 + c0 .. + 3)
        double d[4 + 2 * R];
        const f32x4 *src = (const f32x4 *)&stage[r * SW + c0];
#pragma unroll
        for (int j = 0; j < (4 + 2 * R) / 4; ++j) {
            const f32x4 v = src[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[4 * j + e] = (double)v[e];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double acc = dmul(d[k + R], kTaps[R]);
#pragma unroll
            for (int i = -R; i < 0; ++i) acc = dadd(acc, dmul(dadd(d[k + R + i], d[k + R - i]), kTaps[R + i]));
            sm[r * 68 + c0 + k] = (float)acc;
        }
    }
    __syncthreads();
    const float thresh = 0.01f;                       // image_proc.py:925 (compared in fp32)
    const float NEG_INF = -__builtin_huge_valf();
    Top2 best = {NEG_INF, NEG_INF, -1};
    int cnt = 0;
    const int x = x0 + tx;
    for (int r = 1 + ty; r <= FT; r += 4) {
        const int y = y0 - 1 + r;
        if (x < W && y < H) {
            const int c = tx + 1;
            const float v = sm[r * 68 + c];
            const float up = y > 0 ? sm[(r - 1) * 68 + c] : 0.0f, down = y + 1 < H ? sm[(r + 1) * 68 + c] : 0.0f;
            const float left = x > 0 ? sm[r * 68 + c - 1] : 0.0f, right = x + 1 < W ? sm[r * 68 + c + 1] : 0.0f;
            if ((v >= up) && (v >= down) && (v >= left) && (v >= right) && (v > thresh)) {
                const int idx = y * W + x;
                const Top2 mine = {ld_map(ori, idx), NEG_INF, idx};
                best = top2_merge(best, mine);
                ++cnt;
            }
        }
    }
    if (__ballot(cnt != 0) != 0ull) {                 // most wavefronts hold no peak at all: their record is the empty one as it stands
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            Top2 other;
            other.s1 = lane_xor(best.s1, m);
            other.s2 = lane_xor(best.s2, m);
            other.i1 = lane_xor(best.i1, m);
            best = top2_merge(best, other);
            cnt += lane_xor(cnt, m);
        }
    }
    if (lane == 0) s_part[ty] = PeakPart{best.s1, best.s2, best.i1, cnt};
    __syncthreads();
    if (tid == 0) {
        Top2 b = {s_part[0].s1, s_part[0].s2, s_part[0].i1};
        int total = s_part[0].count;
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const Top2 o = {s_part[w].s1, s_part[w].s2, s_part[w].i1};
            b = top2_merge(b, o);
            total += s_part[w].count;
        }
        const int nper = (int)(gridDim.x * gridDim.y);
        parts[(size_t)n * nper + blockIdx.y * gridDim.x + blockIdx.x] = PeakPart{b.s1, b.s2, b.i1, total};
    }
}

// one wavefront per map: merge its workgroups' records, then network.py:546-577 (as the tail of peaks_kernel)
__global__ void __launch_bounds__(256) peaks_finish_kernel(const float *maps, const PeakPart *parts, int nper, float *keypoints, int32_t *counts,
                                                           int N, int H, int W, double offset, int use_scores, double next_best) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float NEG_INF = -__builtin_huge_valf();
    Top2 best = {NEG_INF, NEG_INF, -1};
    int running = 0;
    for (int i = lane; i < nper; i += 64) {
        const PeakPart p = parts[(size_t)n * nper + i];
        const Top2 o = {p.s1, p.s2, p.i1};
        best = top2_merge(best, o);
        running += p.count;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Top2 other;
        other.s1 = lane_xor(best.s1, m);
        other.s2 = lane_xor(best.s2, m);
        other.i1 = lane_xor(best.i1, m);
        best = top2_merge(best, other);
        running += lane_xor(running, m);
    }
    if (lane != 0) return;
    const float *ori = maps + (size_t)n * H * W;
    if (counts) counts[n] = running;
    float kx = -999.999f, ky = -999.999f;
    const bool accept = (running == 1) || (running > 1 && use_scores != 0 && (double)(best.s1 - best.s2) >= next_best);
    if (accept) {
        const int y = best.i1 / W, x = best.i1 - y * W;
        double cx, cy;
        centroid_5x5(ori, H, W, x, y, offset, &cx, &cy);
        kx = (float)cx;
        ky = (float)cy;
    }
    keypoints[(size_t)n * 2 + 0] = kx;
    keypoints[(size_t)n * 2 + 1] = ky;
}

bool small_map(int H, int W) { return H < 2 * R || W < 2 * R; }      // -> the kernels' SMALL instances (reflect_at)

int smooth_maps(const float *maps, float *tmp, float *out, int N, int H, int W, hipStream_t s) {
    DREAM_REQUIRE((H + 3) / 4 <= 65535, "gaussian: more than 262140 rows");
    DREAM_REQUIRE((size_t)H * W < ((size_t)1 << 30), "gaussian: maps of 2^30 pixels and more are not supported (32-bit byte offsets inside a map)");
    for (int n0 = 0; n0 < N; n0 += 65535) {                  // grid.z limit
        const int nn = N - n0 < 65535 ? N - n0 : 65535;
        const size_t off = (size_t)n0 * H * W;
        const dim3 grid0((unsigned)((W + 63) / 64), (unsigned)((H + GT - 1) / GT), (unsigned)nn);
        const dim3 grid1((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)nn);
        if (small_map(H, W)) {
            hipLaunchKernelGGL((gauss_pass_kernel<0, true>), grid0, dim3(256), 0, s, maps + off, tmp + off, nn, H, W);
            DREAM_LAUNCH_OK();
            hipLaunchKernelGGL((gauss_pass_kernel<1, true>), grid1, dim3(256), 0, s, (const float *)(tmp + off), out + off, nn, H, W);
        } else {
            hipLaunchKernelGGL((gauss_pass_kernel<0, false>), grid0, dim3(256), 0, s, maps + off, tmp + off, nn, H, W);
            DREAM_LAUNCH_OK();
            hipLaunchKernelGGL((gauss_pass_kernel<1, false>), grid1, dim3(256), 0, s, (const float *)(tmp + off), out + off, nn, H, W);
        }
        DREAM_LAUNCH_OK();
    }
    return 0;
}

}  // namespace

extern "C" int dream_gaussian_sigma3_f32(const float *maps, float *tmp, float *out, int N, int H, int W, void *stream) {
    DREAM_REQUIRE(maps && tmp && out && N > 0 && H > 0 && W > 0, "gaussian: bad arguments");
    return smooth_maps(maps, tmp, out, N, H, W, (hipStream_t)stream);
}

extern "C" int dream_keypoints_from_belief_maps_rule_f32(const float *maps, float *scratch, float *keypoints,
                                                         int32_t *peak_counts, int N, int H, int W,
                                                         double offset_due_to_upsampling, int use_belief_peak_scores,
                                                         double belief_peak_next_best_score, void *stream) {
    DREAM_REQUIRE(maps && scratch && keypoints && N > 0 && H > 0 && W > 0, "keypoints_from_belief_maps: bad arguments");
    DREAM_REQUIRE((size_t)H * W < ((size_t)1 << 30), "keypoints_from_belief_maps: maps of 2^30 pixels and more are not supported");
    const size_t total = (size_t)N * H * W;
    // round 6: column pass, then the row pass fused with the scan (gauss_row_peaks_kernel), then one wavefront per map -- when a map's
    // workgroup records fit where its smoothed copy would have gone (they do from 3 x 3 maps on) and the grid fits
    const int gx = (W + 63) / 64, gy = (H + FT - 1) / FT, nper = gx * gy;
    static const bool env_fused = [] { const char *e = getenv("DREAM_PEAKS_FUSED"); return !(e && e[0] == '0'); }();
    const bool fused = g_peaks_fused < 0 ? env_fused : g_peaks_fused != 0;
    if (fused && (size_t)nper * 4 + 4 <= (size_t)H * W && N <= 65535 && gy <= 65535 && (H + GT - 1) / GT <= 65535) {
        const dim3 grid0((unsigned)gx, (unsigned)((H + GT - 1) / GT), (unsigned)N), grid1((unsigned)gx, (unsigned)gy, (unsigned)N);
        PeakPart *parts = (PeakPart *)(scratch + ((total + 3) & ~(size_t)3));      // (16-byte records)
        if (small_map(H, W)) {
            hipLaunchKernelGGL((gauss_pass_kernel<0, true>), grid0, dim3(256), 0, (hipStream_t)stream, maps, scratch, N, H, W);
            DREAM_LAUNCH_OK();
            hipLaunchKernelGGL(gauss_row_peaks_kernel<true>, grid1, dim3(256), 0, (hipStream_t)stream, maps, (const float *)scratch, parts, H, W);
        } else {
            hipLaunchKernelGGL((gauss_pass_kernel<0, false>), grid0, dim3(256), 0, (hipStream_t)stream, maps, scratch, N, H, W);
            DREAM_LAUNCH_OK();
            hipLaunchKernelGGL(gauss_row_peaks_kernel<false>, grid1, dim3(256), 0, (hipStream_t)stream, maps, (const float *)scratch, parts, H, W);
        }
        DREAM_LAUNCH_OK();
        hipLaunchKernelGGL(peaks_finish_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, maps, (const PeakPart *)parts, nper,
                           keypoints, peak_counts, N, H, W, offset_due_to_upsampling, use_belief_peak_scores, belief_peak_next_best_score);
        DREAM_LAUNCH_OK();
        return 0;
    }
    if (int rc = smooth_maps(maps, scratch, scratch + total, N, H, W, (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(peaks_kernel<false>, dim3(N), dim3(256), 0, (hipStream_t)stream, maps,
                       (const float *)(scratch + total), keypoints, peak_counts, (double *)nullptr, (float *)nullptr,
                       H, W, 0, offset_due_to_upsampling, use_belief_peak_scores, belief_peak_next_best_score);
    DREAM_LAUNCH_OK();
    return 0;
}

// Test / A-B hook: the keypoint rule through the fused row pass + scan (1), the three-kernel form of rounds 1-5 (0), or by
// DREAM_PEAKS_FUSED (-1, default: fused).  Same bits either way.
extern "C" int dream_peaks_set_fused(int on) {
    DREAM_REQUIRE(on >= -1 && on <= 1, "peaks: fused = %d", on);
    g_peaks_fused = on;
    return 0;
}

// the reference's defaults: use_belief_peak_scores = True, belief_peak_next_best_score = 0.25 (network.py:189-191)
extern "C" int dream_keypoints_from_belief_maps_f32(const float *maps, float *scratch, float *keypoints,
                                                    int32_t *peak_counts, int N, int H, int W,
                                                    double offset_due_to_upsampling, void *stream) {
    return dream_keypoints_from_belief_maps_rule_f32(maps, scratch, keypoints, peak_counts, N, H, W, offset_due_to_upsampling,
                                                     1, 0.25, stream);
}

extern "C" int dream_peaks_from_belief_maps_f32(const float *maps, float *scratch, double *xy, float *score,
                                                int32_t *counts, int N, int H, int W, int cap,
                                                double offset_due_to_upsampling, void *stream) {
    DREAM_REQUIRE(maps && scratch && xy && score && counts && N > 0 && H > 0 && W > 0 && cap > 0,
                  "peaks_from_belief_maps: bad arguments");
    const size_t total = (size_t)N * H * W;
    if (int rc = smooth_maps(maps, scratch, scratch + total, N, H, W, (hipStream_t)stream)) return rc;
    hipLaunchKernelGGL(peaks_kernel<true>, dim3(N), dim3(256), 0, (hipStream_t)stream, maps,
                       (const float *)(scratch + total), (float *)nullptr, counts, xy, score, H, W, cap,
                       offset_due_to_upsampling, 1, 0.25);
    DREAM_LAUNCH_OK();
    return 0;
}

// Per-channel sums finished inside the launch that produces them, by a two-level "last arriver finishes" tree (dream_cdna4.h).
//
// A launch has `nrows` producers per 64-channel column block (the wave tiles of a GEMM, the workgroups of a reduction), each with
// one row of fp64 partial sums (v0, v1) per channel.  A single last arriver would have to read nrows KB alone (one wavefront pulls
// ~15 GB/s of fresh lines: 157 rows = 50 us -- measured, round 4), so the rows are summed in groups of G ~ sqrt(nrows): the last
// producer of a group adds its group's rows in index order into one second-level row, the last group to finish adds the
// second-level rows in index order.  Fixed grouping, fixed order: the result does not depend on who arrives when (deterministic),
// and the tail after the last producer is two short reads instead of one long one.
//
// workspace (doubles): rows [nrows][N][2], then rows2 [ngroups][N][2];  counters (zero on entry, zero on exit): per column block
// ngroups first-level words + one second-level word.
#pragma once
#include <dream_cdna4.h>

struct StatTree {
    double *rows;            // [nrows][N][2]
    unsigned *counters;      // [ncb][ngroups + 1]
    int nrows, G, ngroups, N;
};

static inline int stat_tree_group(int nrows) {
    if (nrows <= 16) return nrows > 0 ? nrows : 1;      // one level, one round trip
    int g = 16;                                           // a group of 16 rows is one round trip; up to 256 rows: two of them
    while (g * g < nrows) ++g;
    return g;
}
static inline int stat_tree_groups(int nrows) { const int g = stat_tree_group(nrows); return (nrows + g - 1) / g; }
static inline size_t stat_tree_doubles(int nrows, int N) { return ((size_t)nrows + (size_t)stat_tree_groups(nrows)) * (size_t)N * 2; }
static inline int stat_tree_counters(int nrows, int N) { return ((N + 63) / 64) * (stat_tree_groups(nrows) + 1); }
static inline StatTree stat_tree_make(void *workspace, unsigned *counters, int nrows, int N) {
    StatTree t;
    t.rows = (double *)workspace; t.counters = counters; t.nrows = nrows; t.G = stat_tree_group(nrows);
    t.ngroups = stat_tree_groups(nrows); t.N = N;
    return t;
}

// sum of rows [r0, r0 + n) of `base` for channel c (this lane), in index order; sixteen 16-byte loads in flight (a group of up to 16
// rows is ONE round trip to the memory side)
DREAM_DEVICE void stat_tree_sum(const double *base, int r0, int n, int N, int c, double *s0, double *s1) {
    double a = 0.0, b = 0.0;
    const BufferRsrc buf = make_buffer(base + (size_t)r0 * N * 2, (size_t)n * N * 16);
    const unsigned voff = c < N ? (unsigned)c * 16u : BUFFER_OOB;        // a channel beyond N reads zeros
    const unsigned row = (unsigned)N * 16u;
    int r = 0;
    for (; r + 16 <= n; r += 16) {
        double2_ v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = buffer_load_d2_coherent(buf, voff + (unsigned)(r + j) * row, 0u);
#pragma unroll
        for (int j = 0; j < 16; ++j) { a += v[j].x; b += v[j].y; }
    }
    if (r < n) {
        double2_ v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = buffer_load_d2_coherent(buf, r + j < n ? voff + (unsigned)(r + j) * row : BUFFER_OOB, 0u);
#pragma unroll
        for (int j = 0; j < 16; ++j) { a += v[j].x; b += v[j].y; }      // (rows beyond n read zeros: x + 0 is exact)
    }
    *s0 = a;
    *s1 = b;
}

// Called by one whole wavefront per producer (wave-uniform arguments) AFTER it stored its row `prow` of column block `cb` with
// coherent_store.  Returns true in exactly one wavefront per column block -- the one that finishes the tree -- with the totals of
// channel 64 cb + lane in (*t0, *t1).
DREAM_DEVICE bool stat_tree_arrive(const StatTree &t, int cb, int prow, int lane, double *t0, double *t1) {
    publish_wait();
    const int g = prow / t.G;
    const int first = g * t.G;
    const int gsize = t.nrows - first < t.G ? t.nrows - first : t.G;
    unsigned *c1 = t.counters + (size_t)cb * (t.ngroups + 1) + g;
    if (grid_ticket(c1) != (unsigned)(gsize - 1)) return false;
    grid_counter_reset(c1);
    const int c = cb * 64 + lane;
    double a, b;
    stat_tree_sum(t.rows, first, gsize, t.N, c, &a, &b);
    if (t.ngroups == 1) { *t0 = a; *t1 = b; return true; }
    double *rows2 = t.rows + (size_t)t.nrows * t.N * 2;
    if (c < t.N) {
        coherent_store(rows2 + ((size_t)g * t.N + c) * 2, a);
        coherent_store(rows2 + ((size_t)g * t.N + c) * 2 + 1, b);
    }
    publish_wait();
    unsigned *c2 = t.counters + (size_t)cb * (t.ngroups + 1) + t.ngroups;
    if (grid_ticket(c2) != (unsigned)(t.ngroups - 1)) return false;
    grid_counter_reset(c2);
    stat_tree_sum(rows2, 0, t.ngroups, t.N, c, t0, t1);
    return true;
}

// what the finishing wavefront does with the totals of a FORWARD BatchNorm (nn.BatchNorm2d, train mode) ...
struct BnFwdOut {
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long *nbt;
    float eps, momentum;
    float *ab, *mean, *invstd;          // ab: [2][N]
};
DREAM_DEVICE void bn_finish_forward(const BnFwdOut &o, int c, int N, double count, double s, double ss) {
    if (c >= N) return;
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)o.eps);
    const float a = (float)((double)o.gamma[c] * invstd);
    o.ab[c] = a;
    o.ab[N + c] = (float)((double)o.beta[c] - mean * (double)a);
    o.mean[c] = (float)mean;
    o.invstd[c] = (float)invstd;
    if (o.running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var, mo = (double)o.momentum;
        o.running_mean[c] = (float)((1.0 - mo) * (double)o.running_mean[c] + mo * mean);
        o.running_var[c] = (float)((1.0 - mo) * (double)o.running_var[c] + mo * unbiased);
    }
    if (c == 0 && o.nbt != nullptr) *o.nbt += 1;
}

// TEST INFRASTRUCTURE: a minimal SIMT emulator that lets the UNCHANGED kernel sources under
// dream_amd/csrc/*.hip be compiled as plain host C++ (clang++ -x c++) and executed on CPU cores, so
// (each .hip is its own translation unit, see build_emu.py) so that index arithmetic, LDS addressing, barrier placement and MFMA lane layouts can be checked in
// the GPU-less dev container.  It is never part of the product and is not used on the GPU box.
//
// Model: one workgroup = up to 1024 fibers (user-level contexts) run round-robin by one OS thread;
// __syncthreads() and the wave-level primitives (shuffle, ballot, MFMA) are rendezvous points.
// Workgroups of a launch are distributed over OS threads.  "__shared__" is thread_local static.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __constant__ static const
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct double2 { double x, y; };

typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipMemcpyDeviceToDevice = 3 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; };

namespace emu {
struct Fiber {
    void *sp;
    char *stack;
    uint3_emu tidx;
    bool done;
};
struct Block {                     // per OS thread
    Fiber *fibers;
    int nthreads;
    int cur;
    void *sched_sp;
    dim3 bidx, bdim, gdim;
    char *dyn_lds;
    // rendezvous state
    unsigned long block_gen;
    int block_arrived;
    unsigned long wave_gen[16];
    int wave_arrived[16];
    // exchange slots (double buffered by op parity)
    double xchg_d[2][16][64];
    float xchg_a[2][16][64], xchg_b[2][16][64];
    unsigned long long ballot_bits[2][16];
    unsigned wave_op[16][64];      // per-lane op counters
    const std::function<void()> *body;
};
extern thread_local Block *tb;
void yield();
void block_barrier();
void wave_barrier();
void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem);
inline int lane() { return (int)(tb->fibers[tb->cur].tidx.x & 63); }
inline int wave() { return (int)(tb->fibers[tb->cur].tidx.x >> 6); }
}  // namespace emu

#define threadIdx (emu::tb->fibers[emu::tb->cur].tidx)
#define blockIdx (emu::tb->bidx)
#define blockDim (emu::tb->bdim)
#define gridDim (emu::tb->gdim)

inline void __syncthreads() { emu::block_barrier(); }

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    emu::launch(body, grid, block, shmem);
}

inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulator"; }
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *c) { *c = 0; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
typedef void *hipEvent_t;
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    strcpy(p->name, "SIMT emulator"); strcpy(p->gcnArchName, "host"); p->multiProcessorCount = 0; return hipSuccess;
}
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }

// ---- wave-level primitives --------------------------------------------------------------------------
template <class T>
inline T emu_exchange(T v, int src_lane) {
    emu::Block *b = emu::tb;
    const int w = emu::wave(), l = emu::lane();
    const int par = (b->wave_op[w][l]++) & 1;
    double tmp = 0;
    static_assert(sizeof(T) <= sizeof(double), "");
    memcpy(&tmp, &v, sizeof(T));
    b->xchg_d[par][w][l] = tmp;
    emu::wave_barrier();
    T r;
    memcpy(&r, &b->xchg_d[par][w][src_lane & 63], sizeof(T));
    return r;
}
inline float __shfl_xor(float v, int m, int = 64) { return emu_exchange(v, emu::lane() ^ m); }
inline double __shfl_xor(double v, int m, int = 64) { return emu_exchange(v, emu::lane() ^ m); }
inline int __shfl_xor(int v, int m, int = 64) { return emu_exchange(v, emu::lane() ^ m); }
inline int __shfl_up(int v, int d, int = 64) { int l = emu::lane(); return emu_exchange(v, l - d >= 0 ? l - d : l); }
inline unsigned long long __ballot(int pred) {
    emu::Block *b = emu::tb;
    const int w = emu::wave(), l = emu::lane();
    const int par = (b->wave_op[w][l]++) & 1;
    b->xchg_d[par][w][l] = pred ? 1.0 : 0.0;
    emu::wave_barrier();
    unsigned long long m = 0;
    const int n = (int)(b->bdim.x - w * 64 < 64 ? b->bdim.x - w * 64 : 64);
    for (int i = 0; i < n; ++i) if (b->xchg_d[par][w][i] != 0.0) m |= 1ull << i;
    return m;
}
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned atomicMax(unsigned *p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __ddiv_rn(double a, double b) { return a / b; }
inline float atomicAdd(float *p, float v) { float o; 
    // blocks may run on several OS threads
    o = __atomic_load_n((int *)p, __ATOMIC_RELAXED) * 0.0f;
    for (;;) { int old = __atomic_load_n((int *)p, __ATOMIC_RELAXED); float f; memcpy(&f, &old, 4); float nf = f + v; int ni; memcpy(&ni, &nf, 4);
        if (__atomic_compare_exchange_n((int *)p, &old, ni, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { o = f; break; } }
    return o; }

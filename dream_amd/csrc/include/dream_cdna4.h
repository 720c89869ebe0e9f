// CDNA4 (gfx950) device intrinsics used by the DREAM kernels: thin named wrappers over the
// compiler builtins so that the lane layouts are documented in one place.
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD, 157 TFLOP/s chip peak):
//   A operand: one float per lane, lane l holds A[row = l & 31][k = l >> 5]
//   B operand: one float per lane, lane l holds B[k = l >> 5][col = l & 31]
//   C/D      : 16 floats per lane, reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l & 31]
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DREAM_DEVICE __device__ __forceinline__
// all LDS of a kernel lives in ONE dynamic array whose base is 16-byte aligned (ds_read_b128)
#define DREAM_DYNAMIC_LDS(type, var) extern __shared__ __attribute__((aligned(16))) type var[]

// The kernel's own argument block, re-read from the kernarg segment (wave-uniform scalar loads) at the point of the call: for
// values that are needed once per tile block and would otherwise sit in SGPRs across the main loop (the empty asm makes the
// pointer opaque, so the loads cannot be merged with the kernel's initial argument loads and hoisted).  `arg` must be the
// kernel's single by-value struct.
// TAG: distinct call sites in the two arms of a branch get distinct asm strings (identical statements are merged into the join block, and
// the merged pointer then lives in a VGPR: "illegal VGPR to SGPR copy").
template <int TAG = 0, class T>
DREAM_DEVICE const __attribute__((address_space(4))) T *kernarg_again(const T &) {
    // constant address space: the reads become s_load (a generic pointer would make them flat loads into VGPRs -- VMEM latency, and
    // every buffer descriptor built from them non-uniform: a waterfall loop around each access)
    const __attribute__((address_space(4))) T *kp = (const __attribute__((address_space(4))) T *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; kernarg pointer, site %1" : "+s"(kp) : "n"(TAG));
    return kp;
}
#define DREAM_KERNARG(arg) kernarg_again(arg)
#define DREAM_KERNARG_SITE(arg, tag) kernarg_again<tag>(arg)
// a wave-uniform value the optimiser may not see through (kept in a scalar register)
#define DREAM_OPAQUE_SGPR(x) asm volatile("" : "+s"(x))

DREAM_DEVICE f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/SIMD issue, 40 cycles dependent latency; same 157 TFLOP/s peak):
//   A operand: one float per lane, lane l holds A[row = l & 15][k = l >> 4]
//   B operand: one float per lane, lane l holds B[k = l >> 4][col = l & 15]
//   C/D      : 4 floats per lane, reg r of lane l is D[row = 4*(l>>4) + r][col = l & 15]
DREAM_DEVICE f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense, 32 cycles/SIMD issue):
//   A operand: 8 halfs per lane, lane l holds A[row = l & 31][k = 8*(l>>5) .. 8*(l>>5)+7]
//   B operand: 8 halfs per lane, lane l holds B[k = 8*(l>>5) .. +7][col = l & 31];  C/D as above
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
DREAM_DEVICE f32x16 mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// DREAM_PACKED_F32: 1 = the transforms may use packed fp32 VALU operations (v_pk_fma_f32 / v_pk_add_f32: two floats per instruction),
// 0 = plain v_fma_f32 / v_add_f32 / v_sub_f32 only (the translation unit is then also compiled with the packed-fp32-ops target feature
// off, __graft_entry__.py).  gfx950's SIMD-32 issues a wave64 v_fma_f32 in 2 cycles, so a packed operation buys no throughput, and
// beside MFMAs it is priced at +22 .. 26 cycles over the scalar pair (MI355X_MICROARCH.md, per-instruction constants).
#ifndef DREAM_PACKED_F32
#define DREAM_PACKED_F32 1
#endif

// y - x on four floats as two v_pk_add_f32 with the second operand negated (hipcc emits four v_sub_f32 for a vector subtraction)
DREAM_DEVICE f32x4 pk_sub4(f32x4 y, f32x4 x) {
#if !DREAM_PACKED_F32
    return y - x;
#endif
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ lo, hi;
    const f32x2_ ylo = {y[0], y[1]}, yhi = {y[2], y[3]}, xlo = {x[0], x[1]}, xhi = {x[2], x[3]};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(ylo), "v"(xlo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(yhi), "v"(xhi));
    const f32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
}

// One LDS read that the compiler may not merge with a neighbour into a ds_read2 (volatile, LDS address space kept explicit):
// ds_read2_b64 runs in 16-lane groups on 32 banks (8 cycles), two ds_read_b64 in 32-lane groups on 64 banks (2 cycles each).
template <class T>
DREAM_DEVICE T lds_read_unmerged(const T *p) {
    return *(const volatile __attribute__((address_space(3))) T *)p;
}

// the same on two floats
DREAM_DEVICE f32x2 pk_sub2(f32x2 y, f32x2 x) {
#if !DREAM_PACKED_F32
    return y - x;
#endif
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(y), "v"(x));
    return r;
}

// Raw buffer loads (buffer_load_dwordx4 ... offen): 16 bytes per lane at base + voffset + soffset, voffset a 32-bit VGPR,
// soffset a wave-uniform SGPR.  The hardware bounds-checks voffset against the descriptor's size and returns ZEROS for an
// out-of-range lane, so zero padding costs no compare / select: an invalid element simply carries voffset = BUFFER_OOB.
// (No 64-bit address arithmetic either.)  soffset must keep valid lanes inside the buffer; it is not bounds-checked.
struct BufferRsrc { __amdgpu_buffer_rsrc_t r; };
constexpr unsigned BUFFER_OOB = 0x80000000u;              // >= any size make_buffer accepts
DREAM_DEVICE BufferRsrc make_buffer(const void *base, size_t bytes) {
    const unsigned n = bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes;
    return {__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000)};
}
DREAM_DEVICE f32x4 buffer_load_x4(BufferRsrc b, unsigned voffset_bytes, unsigned soffset_bytes) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(b.r, voffset_bytes, soffset_bytes, 0);
    return __builtin_bit_cast(f32x4, v);
}

// the same with a compile-time cache policy (aux bits on gfx94x / gfx950: 1 = sc0, 2 = nt, 16 = sc1) -- A/B builds only
template <int AUX>
DREAM_DEVICE f32x4 buffer_load_x4_aux(BufferRsrc b, unsigned voffset_bytes, unsigned soffset_bytes) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(b.r, voffset_bytes, soffset_bytes, AUX);
    return __builtin_bit_cast(f32x4, v);
}

DREAM_DEVICE f32x2 buffer_load_x2(BufferRsrc b, unsigned voffset_bytes, unsigned soffset_bytes) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voffset_bytes, soffset_bytes, 0);
    return __builtin_bit_cast(f32x2, v);
}
DREAM_DEVICE float buffer_load_f32(BufferRsrc b, unsigned voffset_bytes, unsigned soffset_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voffset_bytes, soffset_bytes, 0));
}
// an out-of-range lane (voffset = BUFFER_OOB) stores nothing: masked stores without branches
DREAM_DEVICE void buffer_store_f32(BufferRsrc b, float v, unsigned voffset_bytes, unsigned soffset_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, voffset_bytes, soffset_bytes, 0);
}
// the same with the non-temporal hint (aux bit 1 = nt on gfx94x/gfx950): a write-once stream that should not displace L2 residents
DREAM_DEVICE void buffer_store_f32_nt(BufferRsrc b, float v, unsigned voffset_bytes, unsigned soffset_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, voffset_bytes, soffset_bytes, 2);
}
DREAM_DEVICE void buffer_store_x4(BufferRsrc b, f32x4 v, unsigned voffset_bytes, unsigned soffset_bytes) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), b.r, voffset_bytes, soffset_bytes, 0);
}
// n / d for n < 2^24 with magic = ceil(2^40 / d) (host side: magic_div40): exact, 3 VALU instead of ~40
DREAM_DEVICE int div_magic40(int n, unsigned long long magic) { return (int)(((unsigned long long)(unsigned)n * magic) >> 40); }

// DPP quad_perm [2,2,1,1]: lane 4k + r receives v from lane 4k + {2, 2, 1, 1}[r] (a VALU operand modifier, no LDS traffic)
DREAM_DEVICE float quad_perm_2211(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x5A, 0xF, 0xF, true));
}

// u[e] + s * quad_perm_2211(u[e]) for the four components, fused (one rounding, = fmaf(s, quad_perm_2211(u[e]), u[e]) bit for bit), as
// four v_fmac_f32 with the DPP modifier on the multiplicand.  The compiler does not fold v_mov_b32_dpp into v_fmac / v_pk_fma (it emitted
// 4 + 2 instructions per call), hence the assembly; the s_nop covers the two wait states a DPP read needs after a VALU write of its
// source, which the compiler's hazard recogniser does not see inside an asm block.
DREAM_DEVICE f32x4 fma_quad_perm_2211(f32x4 u, float s) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3])
        : "v"(s));
    return u;
}

// the same with quad_perm [1,0,3,2] (the lane pair's partner)
DREAM_DEVICE f32x4 fma_quad_perm_1032(f32x4 u, float s) {
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3])
        : "v"(s));
    return u;
}

// DPP quad_perm [2,3,0,1]: lanes 0 <-> 2 and 1 <-> 3 of every quad swap
DREAM_DEVICE float quad_perm_2301(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}

// DPP quad_perm [1,0,3,2]: the two lanes of every pair swap
DREAM_DEVICE float quad_perm_1032(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// lane index 0..63 recomputed from the hardware (v_mbcnt): for code that runs long after the kernel's entry and should not pin a
// register with threadIdx.x until then
DREAM_DEVICE int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave index within the workgroup as a provably wave-uniform (SGPR) value
DREAM_DEVICE int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// cross-lane helpers (wave = 64 lanes)
DREAM_DEVICE float  lane_xor(float v, int m)  { return __shfl_xor(v, m, 64); }
DREAM_DEVICE double lane_xor(double v, int m) { return __shfl_xor(v, m, 64); }
DREAM_DEVICE int    lane_xor(int v, int m)    { return __shfl_xor(v, m, 64); }
DREAM_DEVICE int    lane_up(int v, int d)     { return __shfl_up(v, d, 64); }
DREAM_DEVICE unsigned long long wave_ballot(int pred) { return __ballot(pred); }
DREAM_DEVICE int    popcount64(unsigned long long v) { return __popcll(v); }
DREAM_DEVICE bool   wave_all(int pred) { return __ballot(pred) == __ballot(1); }      // wave-uniform

// ---- "last arriver finishes" (a per-channel reduction finalised inside the launch that produced its partial sums) -------------
// The L2s of the eight XCDs are not coherent with one another, and an agent-scope release / acquire FENCE costs a write-back /
// invalidate of the whole L2 (measured in round 4: a GEMM whose 628 wavefronts each fenced once ran 2.5x slower).  So the partial
// sums travel as agent-scope relaxed atomics -- `global_store_dwordx2 ... sc1` writes through to the memory side,
// `global_load_dwordx2 ... sc1` does not hit a stale line -- ordered by hand: publish (coherent_store), wait until the stores are
// acknowledged (publish_wait: s_waitcnt vmcnt(0) as inline asm, which no compiler pass removes), then draw a ticket from a
// device-scope counter (relaxed RMW, performed at the memory side); the wave that draws the last ticket reads everybody's
// partial sums with coherent_load.  Nobody spins: no residency requirement, no deadlock.
DREAM_DEVICE void coherent_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DREAM_DEVICE double coherent_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DREAM_DEVICE void publish_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16 bytes (two doubles) per lane with the sc1 policy (cache-policy bit 4 of the buffer instructions)
struct double2_ { double x, y; };
DREAM_DEVICE double2_ buffer_load_d2_coherent(BufferRsrc b, unsigned voffset_bytes, unsigned soffset_bytes) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(b.r, voffset_bytes, soffset_bytes, 16);
    return __builtin_bit_cast(double2_, v);
}
// lane 0 of the wave draws the ticket; every lane receives it
DREAM_DEVICE unsigned grid_ticket(unsigned *counter) {
    unsigned t = 0;
    if (lane_id() == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
}
DREAM_DEVICE void grid_counter_reset(unsigned *counter) {
    if (lane_id() == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// value of lane 0 in every lane (wave-uniform)
DREAM_DEVICE int wave_bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }

// IEEE fp64 ops that the compiler must not contract into FMAs (bit-exactness with NumPy/SciPy)
DREAM_DEVICE double dmul(double a, double b) { return __dmul_rn(a, b); }
DREAM_DEVICE double dadd(double a, double b) { return __dadd_rn(a, b); }
DREAM_DEVICE double ddiv(double a, double b) { return __ddiv_rn(a, b); }

// amax side channel: wave-reduce max|v| and atomicMax its bit pattern (non-negative floats order like unsigned
// integers) into one device word.  Same-address atomics serialise in L2 (~12 ns each), so a wave first LOOKS at the
// current value (relaxed, L2-served) and only issues the atomic when it would raise it: after the first few waves
// almost nobody does.
DREAM_DEVICE void publish_amax(unsigned *dst, float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, lane_xor(v, m));
    if ((threadIdx.x & 63) == 0) {
        const unsigned bits = __float_as_uint(v);
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

// Position (ty, tx) of tile row m.  Row-major over the TH x TW tile, or -- when the 2x2 max-pool is fused -- window-
// major: m = 4*window + (dy*2 + dx), so the four accumulator registers (r & 3) of a lane hold exactly one pooling
// window.  rcp = ceil(65536 / d) with d = TW (row-major) or TW/2 (pool); exact for m < 512, d < 128.
DREAM_DEVICE void tile_xy(int m, int TW, int rcp, bool pool, int *ty, int *tx) {
    if (!pool) {
        const int y = (m * rcp) >> 16;
        *ty = y;
        *tx = m - y * TW;
    } else {
        const int q = m >> 2, j = m & 3, hw = TW >> 1;
        const int wy = (q * rcp) >> 16, wx = q - wy * hw;
        *ty = 2 * wy + (j >> 1);
        *tx = 2 * wx + (j & 1);
    }
}

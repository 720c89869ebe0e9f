// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stddef.h>

char *dream_err_buf();            // thread-local, 512 bytes
#define DREAM_ERR_LEN 512

#define DREAM_REQUIRE(cond, ...)                                            \
    do {                                                                    \
        if (!(cond)) {                                                      \
            snprintf(dream_err_buf(), DREAM_ERR_LEN, __VA_ARGS__);          \
            return 1;                                                       \
        }                                                                   \
    } while (0)

#define DREAM_HIP_OK(call)                                                  \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            snprintf(dream_err_buf(), DREAM_ERR_LEN, "%s failed: %s (%s:%d)", #call, \
                     hipGetErrorString(e_), __FILE__, __LINE__);            \
            return 2;                                                       \
        }                                                                   \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize (the full 160 KB of LDS) is a property of (device, kernel): set once per pair,
// from whichever host thread launches the kernel on that device first (api.hip).  0 = ok, else the error text is set.
int dream_allow_full_lds(const void *kernel);

// split-K target of the weight-gradient launches planned by this host thread: `full` workgroups scaled by dream_wgrad_set_width()
int dream_wgrad_width();
static inline long wgrad_target_workgroups(long full) {
    const long t = full * dream_wgrad_width() / 100;
    return t < 1 ? 1 : t;
}

#define DREAM_LAUNCH_OK() DREAM_HIP_OK(hipGetLastError())

// Zero / copy device memory (whole 32-bit words) with a KERNEL launch instead of hipMemsetAsync / hipMemcpyAsync.  Every entry point of
// this library may be captured into a hipGraph (the training steps of the data-parallel replicas are), and a memset / memcpy NODE of a
// replayed graph is not reliably ordered with the kernel nodes around it on this runtime (ROCm 7.2, graph packet capture on): after a
// hipDeviceSynchronize() between two replays the stride-2 1x1 data gradient below read its "zeroed" output as it had been left by the
// previous owner of the memory (tools/dp_exchange_probe.py; profiles/r06_dp_exchange_probe.txt).  0 = ok, else the error text is set.
int dream_zero_words(void *dst, size_t nbytes, hipStream_t stream);
int dream_copy_words(void *dst, const void *src, size_t nbytes, hipStream_t stream);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }

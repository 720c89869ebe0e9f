// Library-level entry points of libdream_hip.so (version, error string, device probes).
#include "common.h"
#include "../../include/dream_hip.h"
#include <string.h>
#include <mutex>
#include <set>
#include <utility>

char *dream_err_buf() {
    static thread_local char buf[DREAM_ERR_LEN] = {0};
    return buf;
}

// Width of the weight-gradient launches of THIS host thread, in per cent of the chip (see dream_hip.h: dream_wgrad_set_width).  Thread-local:
// the replicas of the single-process data-parallel path plan their launches concurrently, each from its own thread.
static thread_local int t_wgrad_width = 100;
int dream_wgrad_width() { return t_wgrad_width; }
extern "C" int dream_wgrad_set_width(int percent) {
    DREAM_REQUIRE(percent >= 5 && percent <= 100, "wgrad width %d %% out of range (5 .. 100)", percent);
    t_wgrad_width = percent;
    return 0;
}

extern "C" int dream_hip_abi_version(void) { return DREAM_HIP_ABI_VERSION; }
extern "C" const char *dream_hip_last_error(void) { return dream_err_buf(); }
extern "C" int dream_hip_device_count(int *count) {
    DREAM_REQUIRE(count != nullptr, "null pointer");
    DREAM_HIP_OK(hipGetDeviceCount(count));
    return 0;
}
extern "C" int dream_hip_device_name(int dev, char *buf, size_t buflen) {
    DREAM_REQUIRE(buf != nullptr && buflen > 0, "null buffer");
    hipDeviceProp_t prop;
    DREAM_HIP_OK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

extern "C" int dream_hip_stream_create(int dev, int priority, void **stream) {
    DREAM_REQUIRE(stream != nullptr, "null pointer");
    int saved = 0;
    DREAM_HIP_OK(hipGetDevice(&saved));
    DREAM_HIP_OK(hipSetDevice(dev));
    hipStream_t s = nullptr;
    hipError_t e;
    if (priority == 0) {
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    } else {
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority > 0 ? least : greatest);
    }
    (void)hipSetDevice(saved);
    DREAM_HIP_OK(e);
    *stream = (void *)s;
    return 0;
}

int dream_allow_full_lds(const void *kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    DREAM_HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return 0;
    DREAM_HIP_OK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({dev, kernel});
    return 0;
}

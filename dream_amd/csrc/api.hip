// Library-level entry points of libdream_hip.so (version, error string, device probes).
#include "common.h"
#include "../../include/dream_hip.h"
#include <string.h>
#include <mutex>
#include <map>
#include <set>
#include <utility>

char *dream_err_buf() {
    static thread_local char buf[DREAM_ERR_LEN] = {0};
    return buf;
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

// Fork / join helper for launches that are independent of one another but each too small to fill the chip (the four output phases of a
// transposed conv on small maps): three auxiliary non-blocking streams + events per device, created once.  fork(): the auxiliary streams
// wait for everything queued on `stream` so far; join(): `stream` waits for what has been queued on them since.  Works under stream
// capture (the event edges pull the auxiliary streams into the capture).  The buffers the launches touch belong to the caller's stream:
// nothing is freed or reused before the join, because every later operation on `stream` is ordered behind it.
int dream_aux_streams(void *stream, int fork, hipStream_t out[3]) {
    struct Aux { hipStream_t s[3]; hipEvent_t f, j[3]; };
    static std::mutex mu;
    static std::map<int, Aux> table;
    int dev = 0;
    DREAM_HIP_OK(hipGetDevice(&dev));
    Aux *a;
    // the whole fork / join sequence under the lock: several host threads may drive one device (replicas that share a GPU), and an
    // event re-recorded by another thread between this thread's record and its waits would order the auxiliary streams behind the wrong point
    std::lock_guard<std::mutex> lock(mu);
    {
        auto it = table.find(dev);
        if (it == table.end()) {
            Aux n;
            for (int i = 0; i < 3; ++i) {
                DREAM_HIP_OK(hipStreamCreateWithFlags(&n.s[i], hipStreamNonBlocking));
                DREAM_HIP_OK(hipEventCreateWithFlags(&n.j[i], hipEventDisableTiming));
            }
            DREAM_HIP_OK(hipEventCreateWithFlags(&n.f, hipEventDisableTiming));
            it = table.emplace(dev, n).first;
        }
        a = &it->second;
    }
    if (fork) {
        DREAM_HIP_OK(hipEventRecord(a->f, (hipStream_t)stream));
        for (int i = 0; i < 3; ++i) {
            DREAM_HIP_OK(hipStreamWaitEvent(a->s[i], a->f, 0));
            out[i] = a->s[i];
        }
    } else {
        for (int i = 0; i < 3; ++i) {
            DREAM_HIP_OK(hipEventRecord(a->j[i], a->s[i]));
            DREAM_HIP_OK(hipStreamWaitEvent((hipStream_t)stream, a->j[i], 0));
        }
    }
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

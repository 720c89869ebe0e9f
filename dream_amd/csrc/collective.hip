// Gradient exchange of the single-process data-parallel path (dream_amd/data_parallel.py): one host process
// drives one replica per GPU (what torch.nn.DataParallel does for /root/reference/dream/network.py:244-256,281-284); after the
// replicas' backward passes every GPU holds one flat fp32 gradient buffer of identical layout, and
//
//     dream_allreduce_sum_f32(ndev, devices, bufs, count, streams)
//
// leaves the element-wise sum in every buffer: one RCCL all-reduce over xGMI (ncclCommInitAll group of the listed devices, created
// once per device list and cached; ncclGroupStart / one ncclAllReduce per device / ncclGroupEnd, each ordered on that replica's
// stream), so that every replica can apply the identical optimizer step to its own parameters -- no gather on GPU 0, no parameter
// broadcast.  xGMI is point-to-point: ONE large call per step (88.9 MB vgg_q / 216 MB resnet_h) lets RCCL use all seven links of
// every GPU; per-tensor calls (46 / 318 of them) would be latency-bound.
//
// RCCL is resolved at first use with dlopen (the soname PyTorch-ROCm already loaded is reused; the library itself does not link
// it, so everything else works on a box without RCCL).  Device lists that are not made of distinct GPUs -- gpu_ids = [0, 0], the
// rehearsal of the N-replica path on a one-GPU box, which RCCL refuses -- and a single device take the local path: the buffers
// are summed into the first one and copied back, stream-ordered with events.  DREAM_FORCE_RCCL=1 sends even a one-device list
// through RCCL (used to rehearse the RCCL call sequence on a one-GPU box).
#include "common.h"
#include "../../include/dream_hip.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef void *rcclComm;
typedef int (*fnCommInitAll)(rcclComm *, int, const int *);
typedef int (*fnGroup)(void);
typedef int (*fnAllReduce)(const void *, void *, size_t, int, int, rcclComm, hipStream_t);
typedef const char *(*fnErrorString)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;       // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
    fnCommInitAll comm_init_all = nullptr;
    fnGroup group_start = nullptr, group_end = nullptr;
    fnAllReduce all_reduce = nullptr;
    fnErrorString error_string = nullptr;
    bool tried = false, ok = false;
    char why[256] = "not tried";                     // the first dlopen / dlsym failure (dlerror() clears itself when read)
};

std::mutex g_mu;
Rccl g_rccl;
std::map<std::vector<int>, std::vector<rcclComm>> g_comms;

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    void *h = nullptr;
    bool noted = false;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
        const char *e = dlerror();
        if (!noted) { snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen failed"); noted = true; }
    }
    if (!h) return false;
    g_rccl.comm_init_all = (fnCommInitAll)dlsym(h, "ncclCommInitAll");
    g_rccl.group_start = (fnGroup)dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (fnGroup)dlsym(h, "ncclGroupEnd");
    g_rccl.all_reduce = (fnAllReduce)dlsym(h, "ncclAllReduce");
    g_rccl.error_string = (fnErrorString)dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.comm_init_all && g_rccl.group_start && g_rccl.group_end && g_rccl.all_reduce && g_rccl.error_string;
    snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", g_rccl.ok ? "" : "a required ncclXxx symbol is missing from librccl");
    return g_rccl.ok;
}

bool distinct(int n, const int *devices) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return false;
    return true;
}

bool use_rccl(int n, const int *devices) {
    if (n > 1) return distinct(n, devices);
    const char *force = getenv("DREAM_FORCE_RCCL");
    return force != nullptr && force[0] == '1';
}

// restores the calling thread's device (and closes an open RCCL group) on every exit path
struct DeviceGuard {
    int saved = -1;
    bool group_open = false;
    hipEvent_t ev = nullptr;
    ~DeviceGuard() {
        if (group_open) g_rccl.group_end();
        if (ev) (void)hipEventDestroy(ev);
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

#define DREAM_RCCL_OK(call)                                                                                        \
    do {                                                                                                           \
        const int r_ = (call);                                                                                     \
        if (r_ != 0) {                                                                                             \
            snprintf(dream_err_buf(), DREAM_ERR_LEN, "%s failed: %s (%s:%d)", #call, g_rccl.error_string(r_), __FILE__, __LINE__); \
            return 3;                                                                                              \
        }                                                                                                          \
    } while (0)

int allreduce_rccl(int n, const int *devices, void *const *bufs, size_t count, void *const *streams) {
    std::lock_guard<std::mutex> lock(g_mu);
    DREAM_REQUIRE(load_rccl(), "allreduce: librccl.so could not be loaded (%s)", g_rccl.why);
    const std::vector<int> key(devices, devices + n);
    auto it = g_comms.find(key);
    if (it == g_comms.end()) {
        std::vector<rcclComm> comms((size_t)n, nullptr);
        DREAM_RCCL_OK(g_rccl.comm_init_all(comms.data(), n, devices));
        it = g_comms.emplace(key, comms).first;
    }
    DeviceGuard guard;
    DREAM_HIP_OK(hipGetDevice(&guard.saved));
    DREAM_RCCL_OK(g_rccl.group_start());
    guard.group_open = true;
    for (int i = 0; i < n; ++i)
        DREAM_RCCL_OK(g_rccl.all_reduce(bufs[i], bufs[i], count, kNcclFloat32, kNcclSum, it->second[(size_t)i], (hipStream_t)streams[i]));
    guard.group_open = false;
    DREAM_RCCL_OK(g_rccl.group_end());
    return 0;
}

// buffers that share a GPU (or a lone buffer): sum into bufs[0] in list order, copy the sum back; every stream waits for it
int allreduce_local(int n, const int *devices, void *const *bufs, size_t count, void *const *streams) {
    if (n == 1) return 0;
    DeviceGuard guard;
    DREAM_HIP_OK(hipGetDevice(&guard.saved));
    DREAM_HIP_OK(hipSetDevice(devices[0]));
    hipStream_t s0 = (hipStream_t)streams[0];
    DREAM_HIP_OK(hipEventCreateWithFlags(&guard.ev, hipEventDisableTiming));
    hipEvent_t ev = guard.ev;
    for (int i = 1; i < n; ++i) {
        if (streams[i] != streams[0]) {
            DREAM_HIP_OK(hipEventRecord(ev, (hipStream_t)streams[i]));
            DREAM_HIP_OK(hipStreamWaitEvent(s0, ev, 0));
        }
        if (dream_add_inplace_f32((float *)bufs[0], (const float *)bufs[i], count, s0)) return 2;
    }
    for (int i = 1; i < n; ++i)
        DREAM_HIP_OK(hipMemcpyAsync(bufs[i], bufs[0], count * sizeof(float), hipMemcpyDeviceToDevice, s0));
    DREAM_HIP_OK(hipEventRecord(ev, s0));
    for (int i = 1; i < n; ++i)
        if (streams[i] != streams[0]) DREAM_HIP_OK(hipStreamWaitEvent((hipStream_t)streams[i], ev, 0));
    return 0;
}

}  // namespace

extern "C" int dream_allreduce_uses_rccl(int ndev, const int *devices) {
    return (ndev > 0 && devices != nullptr && use_rccl(ndev, devices)) ? 1 : 0;
}

extern "C" int dream_allreduce_sum_f32(int ndev, const int *devices, void *const *bufs, size_t count, void *const *streams) {
    DREAM_REQUIRE(ndev > 0 && devices && bufs && streams, "allreduce: bad arguments");
    for (int i = 0; i < ndev; ++i) DREAM_REQUIRE(bufs[i] != nullptr, "allreduce: null buffer %d", i);
    if (count == 0) return 0;
    if (use_rccl(ndev, devices)) return allreduce_rccl(ndev, devices, bufs, count, streams);
    for (int i = 1; i < ndev; ++i)
        DREAM_REQUIRE(devices[i] == devices[0], "allreduce: device list mixes repeated and distinct GPUs (%d, %d)", devices[0], devices[i]);
    return allreduce_local(ndev, devices, bufs, count, streams);
}

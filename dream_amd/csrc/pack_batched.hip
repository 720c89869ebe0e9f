// Every packed weight copy of a network refreshed by ONE launch (a training step re-packs each conv weight after the optimizer
// changed it: 216 launches of 2-10 us for ResNet-101 + decoder, 6.8 % of a step at 16 frames per GPU,
// profiles/r02_layer_profile_resnet_h_train16.txt).  jobs: a device-resident table, one entry per (weight tensor, packed layout);
// grid = (workgroups per job, jobs).  The layouts are the one-tensor kernels' (pack_device.h), bit for bit.
#include <dream_cdna4.h>
#include "common.h"
#include "pack_device.h"
#include "../../include/dream_hip.h"

namespace {

__global__ void __launch_bounds__(256) pack_batched_kernel(const dream_pack_job *jobs) {
    const dream_pack_job j = jobs[blockIdx.y];
    switch (j.kind) {
        case DREAM_PACK_CONV1X1: dream_pack::conv1x1(j.src, j.dst, j.cout, j.cin, j.mode, (int)blockIdx.x, (int)gridDim.x); break;
        case DREAM_PACK_WINOGRAD2: dream_pack::winograd2(j.src, j.dst, j.cout, j.cin, j.mode, (int)blockIdx.x, (int)gridDim.x); break;
        case DREAM_PACK_WINOGRAD4: dream_pack::winograd4(j.src, j.dst, j.cout, j.cin, j.mode, (int)blockIdx.x, (int)gridDim.x); break;
        // one phase of a transposed conv (the ResNet decoder): straight from wT, no materialised 3x3 kernels
        case DREAM_PACK_CONVT_WINOGRAD2: dream_pack::winograd2<true>(j.src, j.dst, j.cout, j.cin, j.mode, (int)blockIdx.x, (int)gridDim.x); break;
        case DREAM_PACK_CONVT_WINOGRAD4: dream_pack::winograd4<true>(j.src, j.dst, j.cout, j.cin, j.mode, (int)blockIdx.x, (int)gridDim.x); break;
        default: break;
    }
}

}  // namespace

extern "C" size_t dream_pack_job_bytes(void) { return sizeof(dream_pack_job); }

extern "C" int dream_pack_weights_batched(const dream_pack_job *jobs_device, int njobs, int workgroups_per_job, void *stream) {
    DREAM_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535 && workgroups_per_job > 0, "batched pack: bad arguments");
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)workgroups_per_job, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
    DREAM_LAUNCH_OK();
    return 0;
}

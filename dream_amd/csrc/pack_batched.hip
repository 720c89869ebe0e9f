// Every packed weight copy of a network refreshed by ONE launch (a training step re-packs each conv weight after the optimizer
// changed it: 216 launches of 2-10 us for ResNet-101 + decoder, 6.8 % of a step at 16 frames per GPU,
// profiles/r02_layer_profile_resnet_h_train16.txt).  jobs: a device-resident table, one entry per (weight tensor, packed layout);
// grid = (workgroups per job, jobs).  The layouts are the one-tensor kernels' (pack_device.h), bit for bit.
#include <dream_cdna4.h>
#include "common.h"
#include "pack_device.h"
#include "../../include/dream_hip.h"

namespace {

DREAM_DEVICE void pack_part(const dream_pack_job &j, int part, int nparts) {
    switch (j.kind) {
        case DREAM_PACK_CONV1X1: dream_pack::conv1x1(j.src, j.dst, j.cout, j.cin, j.mode, part, nparts); break;
        case DREAM_PACK_WINOGRAD2: dream_pack::winograd2(j.src, j.dst, j.cout, j.cin, j.mode, part, nparts); break;
        case DREAM_PACK_WINOGRAD4: dream_pack::winograd4(j.src, j.dst, j.cout, j.cin, j.mode, part, nparts); break;
        // one phase of a transposed conv (the ResNet decoder): straight from wT, no materialised 3x3 kernels
        case DREAM_PACK_CONVT_WINOGRAD2: dream_pack::winograd2<true>(j.src, j.dst, j.cout, j.cin, j.mode, part, nparts); break;
        case DREAM_PACK_CONVT_WINOGRAD4: dream_pack::winograd4<true>(j.src, j.dst, j.cout, j.cin, j.mode, part, nparts); break;
        default: break;
    }
}

__global__ void __launch_bounds__(256) pack_batched_kernel(const dream_pack_job *jobs) {
    const dream_pack_job j = jobs[blockIdx.y];
    pack_part(j, (int)blockIdx.x, (int)gridDim.x);
}

// workgroups dealt out by job size: workgroup i = part spans[i].part of spans[i].nparts of job spans[i].job
__global__ void __launch_bounds__(256) pack_spans_kernel(const dream_pack_job *jobs, const dream_pack_span *spans) {
    const dream_pack_span s = spans[blockIdx.x];
    const dream_pack_job j = jobs[s.job];
    pack_part(j, s.part, s.nparts);
}

}  // namespace

extern "C" size_t dream_pack_job_bytes(void) { return sizeof(dream_pack_job); }
extern "C" size_t dream_pack_span_bytes(void) { return sizeof(dream_pack_span); }

extern "C" int dream_pack_weights_spans(const dream_pack_job *jobs_device, const dream_pack_span *spans_device, int nspans, void *stream) {
    DREAM_REQUIRE(jobs_device && spans_device && nspans > 0, "pack by spans: bad arguments");
    hipLaunchKernelGGL(pack_spans_kernel, dim3((unsigned)nspans), dim3(256), 0, (hipStream_t)stream, jobs_device, spans_device);
    DREAM_LAUNCH_OK();
    return 0;
}

extern "C" int dream_pack_weights_batched(const dream_pack_job *jobs_device, int njobs, int workgroups_per_job, void *stream) {
    DREAM_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535 && workgroups_per_job > 0, "batched pack: bad arguments");
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)workgroups_per_job, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
    DREAM_LAUNCH_OK();
    return 0;
}

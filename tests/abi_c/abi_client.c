/* A client of libdream_hip.so that knows nothing about Python or torch: plain C99, include/dream_hip.h, the HIP
 * runtime C API for memory.  It is what a maintainer binding the library from another host language would write
 * first, and it keeps the header honest: this file is compiled with gcc as C.
 *
 *   abi_client symbols                      -> ABI version / variant count, no device needed
 *   abi_client peaks <in.bin> <offset>      -> dream_keypoints_from_belief_maps_f32 on the maps in <in.bin>
 *       in.bin: int32 N, H, W; float32 maps[N*H*W]; float32 expected[N*2]   (bit-exact comparison, exit 0 / 1)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dream_hip.h"

#define CHECK_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));            \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static int run_peaks(const char *path, double offset) {
    FILE *f = fopen(path, "rb");
    int32_t dims[3];
    if (!f || fread(dims, sizeof(int32_t), 3, f) != 3) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    const size_t n = (size_t)dims[0], hw = (size_t)dims[1] * dims[2];
    float *maps = (float *)malloc(n * hw * sizeof(float)), *want = (float *)malloc(n * 2 * sizeof(float));
    float *got = (float *)malloc(n * 2 * sizeof(float));
    if (fread(maps, sizeof(float), n * hw, f) != n * hw || fread(want, sizeof(float), n * 2, f) != n * 2) return 2;
    fclose(f);
    float *d_maps, *d_scratch, *d_kps;
    CHECK_HIP(hipMalloc((void **)&d_maps, n * hw * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_scratch, 2 * n * hw * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_kps, n * 2 * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_maps, maps, n * hw * sizeof(float), hipMemcpyHostToDevice));
    if (dream_keypoints_from_belief_maps_f32(d_maps, d_scratch, d_kps, NULL, (int)n, dims[1], dims[2], offset, NULL) != 0) {
        fprintf(stderr, "dream_keypoints_from_belief_maps_f32: %s\n", dream_hip_last_error());
        return 2;
    }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(got, d_kps, n * 2 * sizeof(float), hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < n * 2; ++i)
        if (memcmp(&got[i], &want[i], sizeof(float)) != 0) {
            if (bad++ < 5) fprintf(stderr, "keypoint value %zu: got %.9g want %.9g\n", i, got[i], want[i]);
        }
    printf("%zu maps of %dx%d: %d mismatching values\n", n, dims[2], dims[1], bad);
    /* bad arguments must come back as an error code + message, never as a crash */
    if (dream_keypoints_from_belief_maps_f32(NULL, d_scratch, d_kps, NULL, (int)n, dims[1], dims[2], offset, NULL) == 0 ||
        strlen(dream_hip_last_error()) == 0) {
        fprintf(stderr, "null-pointer call was not refused\n");
        return 1;
    }
    hipFree(d_maps); hipFree(d_scratch); hipFree(d_kps);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "symbols") == 0) {
        printf("abi %d, %d selectable conv variants, cout pad of 200 = %zu\n", dream_hip_abi_version(),
               dream_conv3x3_num_variants(), dream_conv3x3_cout_pad(200));
        return dream_hip_abi_version() > 0 ? 0 : 1;
    }
    if (argc >= 4 && strcmp(argv[1], "peaks") == 0) return run_peaks(argv[2], atof(argv[3]));
    fprintf(stderr, "usage: %s symbols | peaks <in.bin> <offset>\n", argv[0]);
    return 2;
}

/* rccl_replicate.c -- one-off fan-out of a device image to the other GPUs of the node.
 *
 * The reference has no counterpart (single process, shared memory).  Reads shard embarrassingly, so the
 * only inter-GPU traffic of the whole path is this: ONE ncclBroadcast (RCCL; xGMI point-to-point links)
 * of the flat image from the GPU that built it.  Steady state has no collective.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include "ctr_host.h"
#include "dev_image.h"

int utree_dev_replicate(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n, utree_dev **out) {
    if (!dev0 || !devices || n < 1 || !out || devices[0] != dev0->device) return UTREE_E_ARG;
    out[0] = dev0;
    if (n == 1) return UTREE_OK;
    int rc = UTREE_OK;
    ncclComm_t *comms = (ncclComm_t *)calloc((size_t)n, sizeof(ncclComm_t));
    hipStream_t *streams = (hipStream_t *)calloc((size_t)n, sizeof(hipStream_t));
    void **images = (void **)calloc((size_t)n, sizeof(void *));
    if (!comms || !streams || !images) { rc = UTREE_E_NOMEM; goto done; }
    images[0] = dev0->image;
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
        if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
        if (i && hipMalloc(&images[i], dev0->image_bytes) != hipSuccess) { rc = UTREE_E_NOMEM; goto done; }
    }
    if (ncclCommInitAll(comms, n, devices) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (ncclGroupStart() != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    for (int i = 0; i < n; ++i) {
        hipSetDevice(devices[i]);
        if (ncclBroadcast(images[i], images[i], dev0->image_bytes, ncclUint8, 0, comms[i], streams[i]) != ncclSuccess) rc = UTREE_E_RCCL;
    }
    if (ncclGroupEnd() != ncclSuccess) rc = UTREE_E_RCCL;
    for (int i = 0; i < n; ++i) {
        hipSetDevice(devices[i]);
        if (hipStreamSynchronize(streams[i]) != hipSuccess) rc = UTREE_E_HIP;
    }
    if (rc) goto done;
    for (int i = 1; i < n; ++i) {
        rc = utree_dev_attach(ctr, devices[i], images[i], dev0->image_bytes, &out[i]);
        if (rc) goto done;
        out[i]->owns = 1;                       /* the replica belongs to its handle */
        images[i] = NULL;
    }
done:
    if (comms) for (int i = 0; i < n; ++i) if (comms[i]) ncclCommDestroy(comms[i]);
    if (streams) for (int i = 0; i < n; ++i) if (streams[i]) { hipSetDevice(devices[i]); hipStreamDestroy(streams[i]); }
    if (images) for (int i = 1; i < n; ++i) if (images[i]) { hipSetDevice(devices[i]); hipFree(images[i]); }
    free(comms); free(streams); free(images);
    return rc;
}

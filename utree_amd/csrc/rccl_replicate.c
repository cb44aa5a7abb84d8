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

/* ---- one process per GPU (bench.py under torchrun, or any launcher that gives every rank one device) --------------------
 * The same single broadcast, with a communicator built from a unique id the root hands to the other ranks over the
 * launcher's own control channel (bench.py: torch.distributed's store).  Root: dev0 = its image.  Others: dev0 = NULL; the
 * image size arrives first (an 8-byte broadcast), then the image, which is attached and owned by the returned handle. */
int utree_rccl_unique_id(void *id_out, size_t cap) {
    ncclUniqueId id;
    if (!id_out || cap < sizeof id) return UTREE_E_ARG;
    if (ncclGetUniqueId(&id) != ncclSuccess) return UTREE_E_RCCL;
    memcpy(id_out, &id, sizeof id);
    return UTREE_OK;
}

int utree_dev_replicate_rank(const utree_ctr *ctr, utree_dev *dev0, int device, int rank, int world, int root, const void *id_bytes,
                             size_t id_len, utree_dev **out) {
    ncclUniqueId id;
    if (!out || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !id_bytes || id_len < sizeof id) return UTREE_E_ARG;
    if ((rank == root) != (dev0 != NULL)) return UTREE_E_ARG;
    if (dev0 && dev0->device != device) return UTREE_E_ARG;
    *out = NULL;
    if (world == 1) { *out = dev0; return UTREE_OK; }
    memcpy(&id, id_bytes, sizeof id);
    int rc = UTREE_OK;
    ncclComm_t comm = NULL;
    hipStream_t st = NULL;
    unsigned long long *d_size = NULL, h_size = dev0 ? (unsigned long long)dev0->image_bytes : 0ull;
    void *image = dev0 ? dev0->image : NULL;
    if (hipSetDevice(device) != hipSuccess) return UTREE_E_HIP;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (hipMalloc((void **)&d_size, 8) != hipSuccess) { rc = UTREE_E_NOMEM; goto done; }
    if (hipMemcpyAsync(d_size, &h_size, 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (ncclBroadcast(d_size, d_size, 8, ncclUint8, root, comm, st) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (hipMemcpyAsync(&h_size, d_size, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (h_size < UTREE_IMG_HEADER_BYTES) { rc = UTREE_E_FORMAT; goto done; }
    if (!dev0 && hipMalloc(&image, (size_t)h_size) != hipSuccess) { image = NULL; rc = UTREE_E_NOMEM; goto done; }
    if (ncclBroadcast(image, image, (size_t)h_size, ncclUint8, root, comm, st) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (hipStreamSynchronize(st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (dev0) *out = dev0;
    else {
        rc = utree_dev_attach(ctr, device, image, (size_t)h_size, out);
        if (!rc) { (*out)->owns = 1; image = NULL; }
    }
done:
    if (comm) ncclCommDestroy(comm);
    if (d_size) hipFree(d_size);
    if (st) hipStreamDestroy(st);
    if (!dev0 && image) hipFree(image);
    return rc;
}

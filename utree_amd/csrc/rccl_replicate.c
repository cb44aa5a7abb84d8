/* rccl_replicate.c -- one-off fan-out of a device image to the other GPUs of the node.
 *
 * The reference's counterpart is its team of workers sharing ONE copy of the database in host memory (itree.c:1009-1018);
 * here every GPU holds a replica.  Reads shard embarrassingly, so the only inter-GPU traffic of the whole path is this: the
 * flat image broadcast from the GPU that built it (RCCL; xGMI point-to-point links), in pieces of at most 1 GiB -- one
 * collective's count stays far below 2^31 elements whatever the build of the library.  Steady state has no collective.
 * utree_dev_fanout is what the command line calls: the broadcast, and when it fails every GPU reads the database from the host
 * over PCIe instead (SURVEY 8(e)'s fallback).
 *
 * Rehearsal on a lease with one GPU (UTREE_RCCL_FORCE=1): the early returns for one device / one rank are skipped -- the
 * communicator is built with its one rank, and the image still travels through ncclBroadcast, out of place into a second
 * allocation on the same card, which is attached like any received copy and handed back in out[0] / *out (the caller keeps
 * dev0).  Every line below then runs; what a one-rank communicator cannot show is the transport between two cards.
 * UTREE_TEST_REPLICATE_FAIL=1 makes both entry points return UTREE_E_RCCL after the communicator is up (the fallback's test).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ctr_host.h"
#include "dev_image.h"

#define BCAST_PIECE ((size_t)1 << 30)

static int env_on(const char *name) { const char *e = getenv(name); return e && atoi(e) > 0; }
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static double g_last_bcast_s;
double utree_dev_replicate_seconds(void) { return g_last_bcast_s; }

int utree_dev_replicate(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n, utree_dev **out) {
    if (!dev0 || !devices || n < 1 || !out || devices[0] != dev0->device) return UTREE_E_ARG;
    const int force = env_on("UTREE_RCCL_FORCE");
    out[0] = dev0;
    g_last_bcast_s = 0.0;
    if (n == 1 && !force) return UTREE_OK;
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) return UTREE_E_ARG;   /* RCCL: one rank per device */
    int rc = UTREE_OK;
    const int first_copy = force ? 0 : 1;            /* ranks from here on receive into a fresh allocation and get a new handle */
    const size_t bytes = dev0->image_bytes;
    ncclComm_t *comms = (ncclComm_t *)calloc((size_t)n, sizeof(ncclComm_t));
    hipStream_t *streams = (hipStream_t *)calloc((size_t)n, sizeof(hipStream_t));
    void **recv = (void **)calloc((size_t)n, sizeof(void *));
    utree_dev **made = (utree_dev **)calloc((size_t)n, sizeof(utree_dev *));
    if (!comms || !streams || !recv || !made) { rc = UTREE_E_NOMEM; goto done; }
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
        if (hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
        if (i >= first_copy && hipMalloc(&recv[i], bytes) != hipSuccess) { recv[i] = NULL; rc = UTREE_E_NOMEM; goto done; }
    }
    if (ncclCommInitAll(comms, n, devices) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (env_on("UTREE_TEST_REPLICATE_FAIL")) { rc = UTREE_E_RCCL; goto done; }
    double t0 = now_s();
    for (size_t lo = 0; lo < bytes && !rc; lo += BCAST_PIECE) {
        size_t cnt = bytes - lo < BCAST_PIECE ? bytes - lo : BCAST_PIECE;
        if (ncclGroupStart() != ncclSuccess) { rc = UTREE_E_RCCL; break; }
        for (int i = 0; i < n; ++i) {
            hipSetDevice(devices[i]);
            const char *src = i == 0 ? (const char *)dev0->image + lo : (const char *)recv[i] + lo;      /* only the root's is read */
            char *dst = i >= first_copy ? (char *)recv[i] + lo : (char *)dev0->image + lo;               /* the root: in place, unless rehearsing */
            if (ncclBroadcast(src, dst, cnt, ncclUint8, 0, comms[i], streams[i]) != ncclSuccess) rc = UTREE_E_RCCL;
        }
        if (ncclGroupEnd() != ncclSuccess) rc = UTREE_E_RCCL;
    }
    for (int i = 0; i < n; ++i) {
        hipSetDevice(devices[i]);
        if (hipStreamSynchronize(streams[i]) != hipSuccess) rc = rc ? rc : UTREE_E_HIP;
    }
    g_last_bcast_s = now_s() - t0;
    if (rc) goto done;
    for (int i = first_copy; i < n; ++i) {
        rc = utree_dev_attach(ctr, devices[i], recv[i], bytes, &made[i]);
        if (rc) goto done;
        made[i]->owns = 1;                           /* the replica belongs to its handle */
        recv[i] = NULL;
    }
    for (int i = first_copy; i < n; ++i) out[i] = made[i], made[i] = NULL;
done:
    if (made) for (int i = 0; i < n; ++i) if (made[i]) utree_dev_free(made[i]);   /* nothing half-made is handed back */
    if (comms) for (int i = 0; i < n; ++i) if (comms[i]) ncclCommDestroy(comms[i]);
    if (streams) for (int i = 0; i < n; ++i) if (streams[i]) { hipSetDevice(devices[i]); hipStreamDestroy(streams[i]); }
    if (recv) for (int i = 0; i < n; ++i) if (recv[i]) { hipSetDevice(devices[i]); hipFree(recv[i]); }
    free(comms); free(streams); free(recv); free(made);
    hipSetDevice(devices[0]);
    return rc;
}

/* ---- what the command line does with more than one GPU ------------------------------------------------------------------
 * The broadcast; if it fails, a warning and utree_dev_upload on every other device (the database crosses PCIe once per GPU
 * instead of once).  The two steps are passed in so that the branch can be driven without a GPU (tests/test_host_cpu.py). */
int utree_dev_fanout_with(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n, int fine_bits, utree_dev **out, int *how,
                          utree_replicate_fn replicate, utree_upload_fn upload) {
    if (!devices || n < 1 || !out || !replicate || !upload) return UTREE_E_ARG;
    if (how) *how = UTREE_FANOUT_NONE;
    out[0] = dev0;
    for (int i = 1; i < n; ++i) out[i] = NULL;
    int rc = replicate(ctr, dev0, devices, n, out);
    if (!rc) { if (how && (n > 1 || out[0] != dev0)) *how = UTREE_FANOUT_BROADCAST; return UTREE_OK; }
    if (rc == UTREE_E_ARG) return rc;
    fprintf(stderr, "[utree_amd] warning: RCCL broadcast of the tree failed (%s); every GPU reads it from the host instead\n", utree_strerror(rc));
    out[0] = dev0;
    for (int i = 1; i < n; ++i) {
        out[i] = NULL;
        rc = upload(ctr, devices[i], fine_bits, &out[i]);
        if (rc) {
            fprintf(stderr, "[utree_amd] device %d: %s\n", devices[i], utree_strerror(rc));
            for (int j = 1; j < i; ++j) { utree_dev_free(out[j]); out[j] = NULL; }
            return rc;
        }
    }
    if (how) *how = UTREE_FANOUT_UPLOAD;
    return UTREE_OK;
}

int utree_dev_fanout(const utree_ctr *ctr, utree_dev *dev0, const int *devices, int n, int fine_bits, utree_dev **out, int *how) {
    return utree_dev_fanout_with(ctr, dev0, devices, n, fine_bits, out, how, utree_dev_replicate, utree_dev_upload);
}

/* ---- one process per GPU (bench.py under torchrun, or any launcher that gives every rank one device) --------------------
 * The same broadcast, with a communicator built from a unique id the root hands to the other ranks over the
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
    const int force = env_on("UTREE_RCCL_FORCE");
    *out = NULL;
    g_last_bcast_s = 0.0;
    if (world == 1 && !force) { *out = dev0; return UTREE_OK; }
    memcpy(&id, id_bytes, sizeof id);
    int rc = UTREE_OK;
    ncclComm_t comm = NULL;
    hipStream_t st = NULL;
    unsigned long long *d_size = NULL, h_size = dev0 ? (unsigned long long)dev0->image_bytes : 0ull;
    const int copy = !dev0 || force;                 /* this rank receives into a fresh allocation */
    void *image = NULL;
    if (hipSetDevice(device) != hipSuccess) return UTREE_E_HIP;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (env_on("UTREE_TEST_REPLICATE_FAIL")) { rc = UTREE_E_RCCL; goto done; }
    if (hipMalloc((void **)&d_size, 8) != hipSuccess) { rc = UTREE_E_NOMEM; goto done; }
    if (hipMemcpyAsync(d_size, &h_size, 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (ncclBroadcast(d_size, d_size, 8, ncclUint8, root, comm, st) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    if (hipMemcpyAsync(&h_size, d_size, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    if (h_size < UTREE_IMG_HEADER_BYTES) { rc = UTREE_E_FORMAT; goto done; }
    if (copy && hipMalloc(&image, (size_t)h_size) != hipSuccess) { image = NULL; rc = UTREE_E_NOMEM; goto done; }
    double t0 = now_s();
    for (size_t lo = 0; lo < (size_t)h_size; lo += BCAST_PIECE) {
        size_t cnt = (size_t)h_size - lo < BCAST_PIECE ? (size_t)h_size - lo : BCAST_PIECE;
        const char *src = dev0 ? (const char *)dev0->image + lo : (const char *)image + lo;
        char *dst = copy ? (char *)image + lo : (char *)dev0->image + lo;
        if (ncclBroadcast(src, dst, cnt, ncclUint8, root, comm, st) != ncclSuccess) { rc = UTREE_E_RCCL; goto done; }
    }
    if (hipStreamSynchronize(st) != hipSuccess) { rc = UTREE_E_HIP; goto done; }
    g_last_bcast_s = now_s() - t0;
    if (!copy) *out = dev0;
    else {
        rc = utree_dev_attach(ctr, device, image, (size_t)h_size, out);
        if (!rc) { (*out)->owns = 1; image = NULL; }
    }
done:
    if (comm) ncclCommDestroy(comm);
    if (d_size) hipFree(d_size);
    if (st) hipStreamDestroy(st);
    if (image) hipFree(image);
    return rc;
}

/* TEST-ONLY stand-in for librccl.so (loaded through DSMI_RCCL_LIBRARY by tests/test_gpu_session.py): the ranks are
 * processes on one machine sharing ONE GPU -- something RCCL itself refuses -- and every message travels as a file in
 * $MOCK_RCCL_DIR.  It exists so that the N > 1 loops of danspeech_amd/csrc/comm.hip (plan, slice offsets, grouped sends
 * and receives, gather rows) run on real device buffers on the one-GPU box.  Not part of the product. */
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct mock_comm { int rank, world; char dir[512]; unsigned seq[64][64]; } *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;

static size_t width(ncclDataType_t t) { return t == 4 || t == 5 || t == 8 ? 8 : (t == 0 || t == 1 ? 1 : 4); }   /* int64/uint64/double, int8/uint8, else 4 */

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof *id); snprintf(id->internal, 128, "mock-%d", (int)getpid()); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    const char* dir = getenv("MOCK_RCCL_DIR");
    (void)id;
    if (!dir || nranks > 64) return 1;
    *comm = (ncclComm_t)calloc(1, sizeof(**comm));
    (*comm)->rank = rank; (*comm)->world = nranks;
    snprintf((*comm)->dir, sizeof (*comm)->dir, "%s", dir);
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r ? "mock rccl failure" : "ok"; }
ncclResult_t ncclGroupStart(void) { return 0; }
ncclResult_t ncclGroupEnd(void) { return 0; }

static ncclResult_t put(ncclComm_t c, const void* dev, size_t bytes, int dst, hipStream_t s) {
    char tmp[700], fin[640];
    void* h = malloc(bytes ? bytes : 1);
    FILE* f;
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%u", c->dir, c->rank, dst, c->seq[c->rank][dst]++);
    snprintf(tmp, sizeof tmp, "%s.tmp", fin);
    f = fopen(tmp, "wb");
    if (!f || fwrite(h, 1, bytes, f) != bytes) return 3;
    fclose(f); free(h);
    return rename(tmp, fin) ? 4 : 0;
}
static ncclResult_t get(ncclComm_t c, void* dev, size_t bytes, int src, hipStream_t s) {
    char fin[640];
    void* h = malloc(bytes ? bytes : 1);
    FILE* f = NULL;
    int tries;
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%u", c->dir, src, c->rank, c->seq[src][c->rank]++);
    for (tries = 0; tries < 60000 && !(f = fopen(fin, "rb")); ++tries) usleep(1000);
    if (!f || fread(h, 1, bytes, f) != bytes) return 5;
    fclose(f); unlink(fin);
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(dev, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return 6;
    free(h);
    return 0;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return put(c, buf, count * width(t), peer, s); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return get(c, buf, count * width(t), peer, s); }
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    int q;
    if (c->rank != root) return get(c, recv, count * width(t), root, s);
    for (q = 0; q < c->world; ++q)
        if (q != root) { const ncclResult_t r = put(c, send, count * width(t), q, s); if (r) return r; }
    if (recv != send && hipMemcpy(recv, send, count * width(t), hipMemcpyDeviceToDevice) != hipSuccess) return 7;
    return 0;
}

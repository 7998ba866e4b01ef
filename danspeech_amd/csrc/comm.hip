// Utterance-level data parallelism for hosts without torch.distributed: the native counterpart of
// danspeech_amd/parallel.py (plan_shards, scatter_ragged, gather_texts).  The reference has no distributed code at all
// (SURVEY 2a); clips are independent on this path, so the only exchanges are the input scatter and the result gather --
// point-to-point ncclSend / ncclRecv between the root and every other rank, grouped, no collective inside the model.
//
// RCCL is bound at run time (dlopen in dsmi_comm_unique_id / dsmi_comm_init): libdsmi.so carries no load-time dependency
// on it, and a process that already holds a copy (PyTorch ships its own) keeps using that one.
#include "common.h"
#include "host_logic.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = std::getenv("DSMI_RCCL_LIBRARY");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            r.why = dlerror();
        }
        if (!r.h) return;
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(r.h, name); if (!p) { ok = false; r.why = std::string("missing symbol ") + name; } return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(r.h); r.h = nullptr; }
    });
    return r.h ? &r : nullptr;
}

std::string rccl_why() { return "dlopen(librccl): no usable library (set DSMI_RCCL_LIBRARY to its path)"; }

int pcm_bytes(int dtype) {
    static const int w[6] = {2, 4, 8, 1, 3, 4};               // I16 F32 F64 U8 I24 I32
    const int base = dtype & 15;
    if (dtype < 0 || base > DSMI_PCM_I32 || (dtype & ~(15 | DSMI_PCM_STEREO))) return 0;
    return w[base] * ((dtype & DSMI_PCM_STEREO) ? 2 : 1);
}

}  // namespace

struct dsmi_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    std::string err;
    unsigned char* dev = nullptr; size_t dev_cap = 0;        // scatter payloads on the device; this rank's shard stays valid until the next scatter
    unsigned char* pin = nullptr; size_t pin_cap = 0;        // their pinned staging on the host (root)
    unsigned char* gdev = nullptr; size_t gdev_cap = 0;      // gather rows, device and pinned host
    unsigned char* gpin = nullptr; size_t gpin_cap = 0;
    int64_t* head = nullptr; size_t head_cap = 0;            // device: header (count, sample type) + clip lengths
};

static thread_local std::string g_comm_error;

namespace {

int cfail(dsmi_comm* c, int code, const std::string& msg) { c->err = msg; return code; }

#define COMM_HIP(c, expr)                                                                                        \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return cfail(c, DSMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)
#define COMM_NCCL(c, expr)                                                                                       \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess) return cfail(c, DSMI_ERR_COMM, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

int ensure(dsmi_comm* c, unsigned char** dev, size_t* cap, size_t need, bool pinned) {
    if (need <= *cap) return DSMI_OK;
    if (*dev) { if (pinned) (void)hipHostFree(*dev); else (void)hipFree(*dev); *dev = nullptr; *cap = 0; }
    const size_t want = std::max<size_t>(need + need / 4, 4096);
    if ((pinned ? hipHostMalloc((void**)dev, want, hipHostMallocDefault) : hipMalloc((void**)dev, want)) != hipSuccess)
        return cfail(c, DSMI_ERR_NOMEM, "allocation failed");
    *cap = want;
    return DSMI_OK;
}

}  // namespace

// ---- host-only planning (danspeech_amd/parallel.py plan_shards): clips sorted by length, descending and stable, dealt
// round-robin, so every rank gets a similar length mix in locally descending order (what pack_padded_sequence asks of a
// batch, reference model.py:117).
extern "C" int dsmi_plan_shards(const int64_t* n_samples, int n, int world, int32_t* rank_of, int32_t* slot_of) {
    if (n < 0 || world < 1 || (n > 0 && (!n_samples || !rank_of || !slot_of))) return DSMI_ERR_INVALID;
    dsmi::plan_shards(n_samples, n, world, rank_of, slot_of);
    return DSMI_OK;
}

extern "C" const char* dsmi_comm_last_error(const dsmi_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

extern "C" int dsmi_comm_unique_id(void* id128) {
    if (!id128) { g_comm_error = "bad argument"; return DSMI_ERR_INVALID; }
    Rccl* r = rccl();
    if (!r) { g_comm_error = "RCCL not available: " + rccl_why(); return DSMI_ERR_COMM; }
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc); return DSMI_ERR_COMM; }
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return DSMI_OK;
}

extern "C" int dsmi_comm_init(const void* id128, int rank, int world, int device, dsmi_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { g_comm_error = "bad comm arguments"; return DSMI_ERR_INVALID; }
    Rccl* r = rccl();
    if (!r) { g_comm_error = "RCCL not available: " + rccl_why(); return DSMI_ERR_COMM; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_error = "hipSetDevice failed"; return DSMI_ERR_HIP; }
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    dsmi_comm* c = new dsmi_comm;
    c->rank = rank; c->world = world; c->device = device;
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + r->GetErrorString(rc); delete c; return DSMI_ERR_COMM; }
    *out = c;
    return DSMI_OK;
}

extern "C" void dsmi_comm_destroy(dsmi_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    if (c->dev) (void)hipFree(c->dev);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->gdev) (void)hipFree(c->gdev);
    if (c->gpin) (void)hipHostFree(c->gpin);
    if (c->head) (void)hipFree(c->head);
    delete c;
}

// Root passes the clips; every rank gets its shard back to back in device memory (longest first), the shard's lengths and
// the clips' positions in the root's list.  Three exchanges: header (count, sample type), lengths, payloads.
extern "C" int dsmi_comm_scatter(dsmi_comm* c, int root, const void* const* clips_host, const int64_t* n_samples_host, int pcm_dtype, int n,
                                 const void** shard_pcm_dev, int64_t* shard_n_samples, int32_t* shard_index, int shard_cap,
                                 int* shard_count, int* shard_dtype, int* total_count, void* stream) {
    if (!c) return DSMI_ERR_INVALID;
    // A rank that returned before (or between) the exchanges would leave the others waiting in a broadcast or a receive for
    // ever.  So: what every rank can check identically (root, the header, the lengths) may return at once -- all ranks do;
    // what only THIS rank knows (its output pointers, its shard_cap) is remembered, the rank still takes part in every
    // exchange, and the error is returned at the end; the root's own bad arguments travel in the header.
    if (root < 0 || root >= c->world) return cfail(c, DSMI_ERR_INVALID, "bad scatter arguments");
    int late_rc = DSMI_OK; const char* late_msg = "";
    if (!shard_pcm_dev || !shard_n_samples || !shard_index || !shard_count || shard_cap < 0) { late_rc = DSMI_ERR_INVALID; late_msg = "bad scatter arguments"; }
    const bool is_root = c->rank == root;
    const bool root_bad = is_root && (n < 0 || !pcm_bytes(pcm_dtype) || (n > 0 && (!clips_host || !n_samples_host)));
    Rccl* r = rccl();
    hipStream_t s = (hipStream_t)stream;
    COMM_HIP(c, hipSetDevice(c->device));

    // 1. header
    int64_t hd[2] = {is_root ? (root_bad ? -1 : n) : 0, is_root ? pcm_dtype : 0};
    if (c->head_cap < 2) { if (c->head) (void)hipFree(c->head); c->head = nullptr; COMM_HIP(c, hipMalloc((void**)&c->head, sizeof(int64_t) * 4096)); c->head_cap = 4096; }
    if (c->world > 1) {
        if (is_root) COMM_HIP(c, hipMemcpyAsync(c->head, hd, sizeof(hd), hipMemcpyHostToDevice, s));
        COMM_NCCL(c, r->Broadcast(c->head, c->head, 2, ncclInt64, root, c->comm, s));
        COMM_HIP(c, hipMemcpyAsync(hd, c->head, sizeof(hd), hipMemcpyDeviceToHost, s));
        COMM_HIP(c, hipStreamSynchronize(s));
    }
    if (hd[0] == -1) return cfail(c, DSMI_ERR_INVALID, is_root ? "bad scatter arguments (root)" : "the root's scatter arguments were refused");
    const int total = (int)hd[0], dtype = (int)hd[1], sb = pcm_bytes(dtype);
    if (total < 0 || !sb) return cfail(c, DSMI_ERR_COMM, "malformed scatter header");
    if (total_count) *total_count = total;
    if (shard_dtype) *shard_dtype = dtype;

    // 2. lengths
    std::vector<int64_t> lens((size_t)std::max(total, 1), 0);
    if (is_root) std::copy(n_samples_host, n_samples_host + total, lens.begin());
    if (c->world > 1 && total > 0) {
        if ((size_t)total > c->head_cap) { (void)hipFree(c->head); c->head = nullptr; c->head_cap = 0; COMM_HIP(c, hipMalloc((void**)&c->head, sizeof(int64_t) * (size_t)total * 2)); c->head_cap = (size_t)total * 2; }
        if (is_root) COMM_HIP(c, hipMemcpyAsync(c->head, lens.data(), sizeof(int64_t) * total, hipMemcpyHostToDevice, s));
        COMM_NCCL(c, r->Broadcast(c->head, c->head, (size_t)total, ncclInt64, root, c->comm, s));
        COMM_HIP(c, hipMemcpyAsync(lens.data(), c->head, sizeof(int64_t) * total, hipMemcpyDeviceToHost, s));
        COMM_HIP(c, hipStreamSynchronize(s));
    }
    for (int i = 0; i < total; ++i) if (lens[(size_t)i] < 1) return cfail(c, DSMI_ERR_INVALID, "empty clip");

    // 3. the plan, identical on every rank: rank q's shard = order[q], order[q + world], ...
    const std::vector<int> order = dsmi::length_order(lens.data(), total);
    std::vector<size_t> bytes_of((size_t)c->world, 0);
    for (int k = 0; k < total; ++k) bytes_of[(size_t)(k % c->world)] += (size_t)lens[(size_t)order[(size_t)k]] * sb;
    const int mine = total > c->rank ? (total - c->rank + c->world - 1) / c->world : 0;
    if (!late_rc && mine > shard_cap) { late_rc = DSMI_ERR_CAPACITY; late_msg = "shard_cap smaller than this rank's share of the clips"; }
    if (!late_rc) {
        for (int j = 0; j < mine; ++j) {
            const int i = order[(size_t)(c->rank + j * c->world)];
            shard_index[j] = i; shard_n_samples[j] = lens[(size_t)i];
        }
        *shard_count = mine;
    }

    // 4. payloads: the root stages every rank's shard back to back, uploads once and sends each rank its slice
    size_t my_off = 0, all = 0;
    std::vector<size_t> off((size_t)c->world + 1, 0);
    for (int q = 0; q < c->world; ++q) off[(size_t)q + 1] = off[(size_t)q] + ((bytes_of[(size_t)q] + 15) & ~(size_t)15);
    all = off[(size_t)c->world];
    if (is_root) {
        int rc = ensure(c, &c->pin, &c->pin_cap, all, true); if (rc) return rc;
        rc = ensure(c, &c->dev, &c->dev_cap, all, false); if (rc) return rc;
        for (int q = 0; q < c->world; ++q) {
            size_t at = off[(size_t)q];
            for (int k = q; k < total; k += c->world) {
                const int i = order[(size_t)k];
                const size_t nb = (size_t)lens[(size_t)i] * sb;
                std::memcpy(c->pin + at, clips_host[i], nb);
                at += nb;
            }
        }
        if (all) COMM_HIP(c, hipMemcpyAsync(c->dev, c->pin, all, hipMemcpyHostToDevice, s));
        my_off = off[(size_t)root];
        if (c->world > 1) {
            COMM_NCCL(c, r->GroupStart());
            for (int q = 0; q < c->world; ++q)
                if (q != root && bytes_of[(size_t)q]) COMM_NCCL(c, r->Send(c->dev + off[(size_t)q], bytes_of[(size_t)q], ncclUint8, q, c->comm, s));
            COMM_NCCL(c, r->GroupEnd());
        }
    } else {
        int rc = ensure(c, &c->dev, &c->dev_cap, bytes_of[(size_t)c->rank], false); if (rc) return rc;
        if (bytes_of[(size_t)c->rank]) {
            COMM_NCCL(c, r->GroupStart());
            COMM_NCCL(c, r->Recv(c->dev, bytes_of[(size_t)c->rank], ncclUint8, root, c->comm, s));
            COMM_NCCL(c, r->GroupEnd());
        }
    }
    COMM_HIP(c, hipStreamSynchronize(s));                     // the staging buffer is free again; the shard is in place
    if (late_rc) return cfail(c, late_rc, late_msg);
    *shard_pcm_dev = c->dev ? c->dev + my_off : nullptr;
    return DSMI_OK;
}

// Every rank passes its shard's transcripts ([shard_count][text_stride], NUL-terminated) and their positions in the root's
// list (dsmi_comm_scatter's shard_index); the root gets all `total_count` transcripts in the caller's order.
// Fixed payload per rank: ceil(total / world) rows of (int32 position, text_stride bytes).
extern "C" int dsmi_comm_gather_text(dsmi_comm* c, int root, const char* text, int text_stride, const int32_t* shard_index, int shard_count,
                                     int total_count, char* all_text, void* stream) {
    if (!c) return DSMI_ERR_INVALID;
    const bool is_root = c->rank == root;
    if (root < 0 || root >= c->world || text_stride < 1 || shard_count < 0 || total_count < 0 || (shard_count > 0 && (!text || !shard_index)) ||
        (is_root && total_count > 0 && !all_text))
        return cfail(c, DSMI_ERR_INVALID, "bad gather arguments");
    const int per = (total_count + c->world - 1) / c->world;
    if (shard_count > per) return cfail(c, DSMI_ERR_INVALID, "more transcripts than this rank's share");
    if (total_count == 0) return DSMI_OK;
    Rccl* r = rccl();
    hipStream_t s = (hipStream_t)stream;
    COMM_HIP(c, hipSetDevice(c->device));
    const size_t row = 4 + (size_t)text_stride, block = (((size_t)per * row) + 15) & ~(size_t)15;
    const size_t need = is_root ? block * c->world : block;
    int rc = ensure(c, &c->gpin, &c->gpin_cap, need, true); if (rc) return rc;
    rc = ensure(c, &c->gdev, &c->gdev_cap, need, false); if (rc) return rc;
    unsigned char* mine = c->gpin + (is_root ? block * (size_t)root : 0);
    for (int j = 0; j < per; ++j) {
        const int32_t pos = j < shard_count ? shard_index[j] : -1;
        std::memcpy(mine + (size_t)j * row, &pos, 4);
        if (j < shard_count) std::memcpy(mine + (size_t)j * row + 4, text + (size_t)j * text_stride, (size_t)text_stride);
    }
    if (c->world > 1) {
        if (is_root) {
            COMM_NCCL(c, r->GroupStart());
            for (int q = 0; q < c->world; ++q)
                if (q != root) COMM_NCCL(c, r->Recv(c->gdev + block * (size_t)q, block, ncclUint8, q, c->comm, s));
            COMM_NCCL(c, r->GroupEnd());
            for (int q = 0; q < c->world; ++q)
                if (q != root) COMM_HIP(c, hipMemcpyAsync(c->gpin + block * (size_t)q, c->gdev + block * (size_t)q, block, hipMemcpyDeviceToHost, s));
        } else {
            COMM_HIP(c, hipMemcpyAsync(c->gdev, c->gpin, block, hipMemcpyHostToDevice, s));
            COMM_NCCL(c, r->GroupStart());
            COMM_NCCL(c, r->Send(c->gdev, block, ncclUint8, root, c->comm, s));
            COMM_NCCL(c, r->GroupEnd());
        }
        COMM_HIP(c, hipStreamSynchronize(s));
    }
    if (is_root) {
        for (int q = 0; q < c->world; ++q)
            for (int j = 0; j < per; ++j) {
                const unsigned char* src = c->gpin + block * (size_t)q + (size_t)j * row;
                int32_t pos;
                std::memcpy(&pos, src, 4);
                if (pos < 0) continue;
                if (pos >= total_count) return cfail(c, DSMI_ERR_COMM, "gathered a position outside the batch");
                std::memcpy(all_text + (size_t)pos * text_stride, src + 4, (size_t)text_stride);
                all_text[(size_t)pos * text_stride + text_stride - 1] = 0;
            }
    }
    return DSMI_OK;
}

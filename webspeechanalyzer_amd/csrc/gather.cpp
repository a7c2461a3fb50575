// gather.cpp — the one exchange step of the multi-GPU path inside the C ABI (include/wsa.h, wsa_gather_*): the feature rows of the batches
// of several contexts (one per GPU, one process) are collected in the root context's device memory with ONE grouped RCCL exchange over xGMI.
//
// The reference has no counterpart: it runs one launch per file on one device (ref /root/reference/src/index.js:291); clips are independent
// (all state is per launch, reset_segmentation ref dist/main.js:2 @B24629), so they shard over GPUs with no exchange on the data path and the
// collection of the row tables is the only communication.  Same protocol as webspeechanalyzer_amd/gather.py (the
// torch.distributed form bench.py uses with one process per GPU): the row counts first — here the host already has them, every batch's run
// publishes its counters — then exactly each rank's rows, metadata and features in their own dtypes, received straight into the rank's slice
// of the root's tables; a rank without rows sends nothing.  The root's own rows travel as a send to itself inside the same group.
//
// librccl is loaded on first use (dlopen): hosts that never gather do not pay for it, and libwsa has no link-time dependency on it.
#include <dlfcn.h>
#include <cstring>
#include <string>
#include <vector>
#include <rccl/rccl.h>
#include "api_internal.hpp"

using wsa_api::fail;

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;            // optional
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
        auto sym = [&](const char* n) -> void* { void* p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
        if (!(CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString)) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;
}  // namespace

struct wsa_gather {
    std::vector<wsa_ctx*> ctxs;
    std::vector<ncclComm_t> comms;
    int root = 0;
    int32_t* d_meta = nullptr; double* d_feat = nullptr; uint64_t cap_rows = 0;     // on the root device
    std::vector<uint32_t> rows;                                                      // per rank, of the last gather
    uint32_t total = 0;
    hipStream_t root_stream = nullptr;                                               // where the root's receives of the last gather were enqueued
    std::vector<hipStream_t> own;                                                    // one stream per rank, on that rank's device: what a rank's calls ride on when the caller names none
    bool gathered = false;
    bool broken = false;                                                             // a call inside the group failed: the communicators are not usable any more
};

// the calling thread's current device is the caller's business: every entry point that switches devices puts it back on every exit path
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

#define NCCL_TRY(ctx, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) \
        return fail((ctx), WSA_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } while (0)

// the context a planned batch belongs to (api.hip)
extern "C" wsa_ctx* wsa_batch_ctx_internal(const wsa_batch* b);

extern "C" {

wsa_status wsa_gather_create(wsa_ctx* const* ctxs, int32_t n_ranks, int32_t root, wsa_gather** out) {
    if (!ctxs || !out || n_ranks < 1 || root < 0 || root >= n_ranks) return fail(nullptr, WSA_ERR_INVALID, "wsa_gather_create: bad arguments");
    *out = nullptr;
    for (int i = 0; i < n_ranks; i++) {
        if (!ctxs[i]) return fail(nullptr, WSA_ERR_INVALID, "wsa_gather_create: null context");
        for (int k = 0; k < i; k++) if (ctxs[k]->device == ctxs[i]->device) return fail(ctxs[root], WSA_ERR_INVALID, "wsa_gather_create: two contexts on one device (a rank is a GPU)");
    }
    if (!g_rccl.load()) return fail(ctxs[root], WSA_ERR_NO_DEVICE, g_rccl.err);
    DeviceGuard keep_device;
    wsa_gather* g = new wsa_gather();
    g->ctxs.assign(ctxs, ctxs + n_ranks); g->root = root; g->rows.assign((size_t)n_ranks, 0u);
    std::vector<int> devs((size_t)n_ranks);
    for (int i = 0; i < n_ranks; i++) devs[(size_t)i] = ctxs[i]->device;
    g->comms.assign((size_t)n_ranks, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_ranks, devs.data());          // one process, n GPUs: rank i = ctxs[i]'s device
    if (r != ncclSuccess) { const std::string m = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); delete g; return fail(ctxs[root], WSA_ERR_HIP, m); }
    // one stream per rank on the rank's own device: a NULL stream would be "the null stream of whatever device is current" on the calling thread
    g->own.assign((size_t)n_ranks, nullptr);
    for (int i = 0; i < n_ranks; i++) {
        hipError_t e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->own[(size_t)i], hipStreamNonBlocking);
        if (e != hipSuccess) { const std::string m = std::string("wsa_gather_create: stream on device ") + std::to_string(ctxs[i]->device) + ": " + hipGetErrorString(e); wsa_gather_destroy(g); return fail(ctxs[root], WSA_ERR_HIP, m); }
    }
    *out = g;
    return WSA_OK;
}

void wsa_gather_destroy(wsa_gather* g) {
    if (!g) return;
    DeviceGuard keep_device;
    for (ncclComm_t c : g->comms) if (c) (void)((g->broken && g_rccl.CommAbort) ? g_rccl.CommAbort(c) : g_rccl.CommDestroy(c));
    for (size_t i = 0; i < g->own.size(); i++) if (g->own[i]) { (void)hipSetDevice(g->ctxs[i]->device); (void)hipStreamDestroy(g->own[i]); }
    if (g->d_meta || g->d_feat) { (void)hipSetDevice(g->ctxs[(size_t)g->root]->device); if (g->d_meta) (void)hipFree(g->d_meta); if (g->d_feat) (void)hipFree(g->d_feat); }
    delete g;
}

wsa_status wsa_gather_rows(wsa_gather* g, wsa_batch* const* batches, void* const* streams, wsa_gather_result* out) {
    if (!g || !batches || !out) return WSA_ERR_INVALID;
    const int n = (int)g->ctxs.size();
    wsa_ctx* rc = g->ctxs[(size_t)g->root];
    if (g->broken) return fail(rc, WSA_ERR_HIP, "wsa_gather_rows: an earlier exchange failed inside its RCCL group; destroy this wsa_gather and create a new one");
    DeviceGuard keep_device;
    // a rank's stream: the caller's, or the gather's own stream on that rank's device
    auto stream_of = [&](int r) { hipStream_t s = reinterpret_cast<hipStream_t>(streams ? streams[r] : nullptr); return s ? s : g->own[(size_t)r]; };
    // 1. the counts: every rank's run has published its counters; wsa_batch_result waits for the rank's stream and reads them
    std::vector<wsa_device_result> res((size_t)n);
    uint64_t total = 0;
    for (int r = 0; r < n; r++) {
        if (!batches[r]) return fail(rc, WSA_ERR_INVALID, "wsa_gather_rows: null batch");
        if (wsa_batch_ctx_internal(batches[r]) != g->ctxs[(size_t)r])
            return fail(rc, WSA_ERR_INVALID, std::string("wsa_gather_rows: batch ") + std::to_string(r) + " was not planned on the context of rank " + std::to_string(r));
        const wsa_status st = wsa_batch_result(batches[r], streams ? streams[r] : nullptr, &res[(size_t)r]);
        if (st != WSA_OK) return fail(rc, st, std::string("rank ") + std::to_string(r) + ": " + wsa_last_error(g->ctxs[(size_t)r]));
        g->rows[(size_t)r] = res[(size_t)r].n_rows; total += res[(size_t)r].n_rows;
    }
    if (total > 0xfffffff0ull) return fail(rc, WSA_ERR_INVALID, "wsa_gather_rows: more than 2^32 rows");
    // 2. the root's tables (grown when needed: the tables of the previous gather are freed here, see wsa.h)
    HIP_TRY(rc, hipSetDevice(rc->device));
    if (total > g->cap_rows) {
        // nothing may still be reading the old tables.  The whole root device is waited for, not the stream the last gather rode on: that may have been a
        // caller's stream, which the caller is free to have destroyed since
        if (g->gathered) HIP_TRY(rc, hipDeviceSynchronize());
        if (g->d_meta) (void)hipFree(g->d_meta);
        if (g->d_feat) (void)hipFree(g->d_feat);
        g->d_meta = nullptr; g->d_feat = nullptr; g->cap_rows = 0; g->gathered = false;
        const uint64_t cap = total + total / 4 + 64;
        HIP_TRY(rc, hipMalloc(reinterpret_cast<void**>(&g->d_meta), cap * 8 * sizeof(int32_t)));
        HIP_TRY(rc, hipMalloc(reinterpret_cast<void**>(&g->d_feat), cap * WSA_NFEAT * sizeof(double)));
        g->cap_rows = cap;
    }
    // 3. one grouped exchange: every rank sends exactly its rows to the root, the root receives them into consecutive slices.
    //    Every call of a rank is made with that rank's device current.  A failing call does NOT return from inside the group: the first
    //    error is kept, the group is closed, and the communicators are marked unusable (an exchange that was only partly enqueued leaves
    //    the peers waiting for each other).
    hipStream_t rs = stream_of(g->root);
    std::string first_err;
    auto keep = [&](ncclResult_t r_, const char* what, int rank) {
        if (r_ != ncclSuccess && first_err.empty()) first_err = std::string(what) + " (rank " + std::to_string(rank) + "): " + g_rccl.GetErrorString(r_);
        return r_ == ncclSuccess;
    };
    auto keep_hip = [&](hipError_t e_, const char* what, int rank) {
        if (e_ != hipSuccess && first_err.empty()) first_err = std::string(what) + " (rank " + std::to_string(rank) + "): " + hipGetErrorString(e_);
        return e_ == hipSuccess;
    };
    NCCL_TRY(rc, g_rccl.GroupStart());                                                 // (nothing is open yet if this one fails)
    uint64_t off = 0;
    for (int r = 0; r < n && first_err.empty(); r++) {
        const uint64_t k = g->rows[(size_t)r];
        if (k) {
            hipStream_t s = stream_of(r);
            bool ok = keep_hip(hipSetDevice(g->ctxs[(size_t)r]->device), "hipSetDevice", r);
            ok = ok && keep(g_rccl.Send(res[(size_t)r].d_row_meta, k * 8, ncclInt32, g->root, g->comms[(size_t)r], s), "ncclSend(meta)", r);
            ok = ok && keep(g_rccl.Send(res[(size_t)r].d_row_feat, k * WSA_NFEAT, ncclFloat64, g->root, g->comms[(size_t)r], s), "ncclSend(features)", r);
            ok = ok && keep_hip(hipSetDevice(rc->device), "hipSetDevice", g->root);
            ok = ok && keep(g_rccl.Recv(g->d_meta + off * 8, k * 8, ncclInt32, r, g->comms[(size_t)g->root], rs), "ncclRecv(meta)", r);
            ok = ok && keep(g_rccl.Recv(g->d_feat + off * WSA_NFEAT, k * WSA_NFEAT, ncclFloat64, r, g->comms[(size_t)g->root], rs), "ncclRecv(features)", r);
        }
        off += k;
    }
    const ncclResult_t ge = g_rccl.GroupEnd();                                         // always: an open group would swallow the thread's next RCCL call
    (void)hipSetDevice(rc->device);
    if (!first_err.empty() || ge != ncclSuccess) {
        if (first_err.empty()) first_err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge);
        g->broken = true; g->gathered = false;
        return fail(rc, WSA_ERR_HIP, "wsa_gather_rows: " + first_err);
    }
    g->total = (uint32_t)total; g->root_stream = rs; g->gathered = true;
    out->n_ranks = (uint32_t)n; out->n_rows = (uint32_t)total; out->rows_per_rank = g->rows.data(); out->d_row_meta = g->d_meta; out->d_row_feat = g->d_feat;
    return WSA_OK;
}

wsa_status wsa_gather_copy_rows(wsa_gather* g, int32_t* row_meta, double* row_feat, uint32_t rows_cap) {
    if (!g) return WSA_ERR_INVALID;
    DeviceGuard keep_device;
    wsa_ctx* rc = g->ctxs[(size_t)g->root];
    if (!g->gathered) return fail(rc, WSA_ERR_INVALID, "wsa_gather_copy_rows: no gather yet");
    if (rows_cap < g->total) return fail(rc, WSA_ERR_INVALID, "wsa_gather_copy_rows: buffers too small");
    HIP_TRY(rc, hipSetDevice(rc->device));
    if (g->total) {
        if (row_meta) HIP_TRY(rc, hipMemcpyAsync(row_meta, g->d_meta, (size_t)g->total * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, g->root_stream));
        if (row_feat) HIP_TRY(rc, hipMemcpyAsync(row_feat, g->d_feat, (size_t)g->total * WSA_NFEAT * sizeof(double), hipMemcpyDeviceToHost, g->root_stream));
    }
    HIP_TRY(rc, hipStreamSynchronize(g->root_stream));
    return WSA_OK;
}

}  // extern "C"

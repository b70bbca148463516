// mock_rccl.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl inside ONE process on ONE GPU: the ranks of a communicator
// are host THREADS that share the device, and every collective is a host-side rendezvous followed by device-to-device
// copies on the calling rank's stream.  RCCL itself refuses two ranks on one device, so this is how the multi-rank paths
// of the C++ shard drivers (ssme_pf_shard_run_series, ssme_lw_shard_run_series: grouped all-gathers, fixed-halo and planned
// send/recv exchanges, window-overflow fallback) are exercised on the one-GPU test box.  The drivers resolve RCCL by
// dlsym(RTLD_DEFAULT, ...), so an executable linked against this library gets these symbols.
// Semantics kept: stream ordering on each rank, pairwise matching of sends and receives in issue order, group
// aggregation.  Not kept: any performance property.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
}

namespace {
struct Op { int kind; const void* send; void* recv; size_t count; int peer; };   // kind 0 all-gather, 1 send, 2 recv, 3 all-reduce(max) of int32
struct World {
    int size = 0, arrived = 0, generation = 0, attached = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::vector<Op>> posted;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == size) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct Comm { World* w; int rank; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm* g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;
std::mutex g_registry_mu;
World* g_pending = nullptr;      // the world being assembled by ncclCommInitRank calls

ncclResult_t flush() {
    Comm* c = g_comm;
    if (!c || g_ops.empty()) { g_ops.clear(); return ncclSuccess; }
    World* w = c->w;
    if (hipStreamSynchronize(g_stream) != hipSuccess) return ncclUnhandledCudaError;       // my send buffers are final
    { std::lock_guard<std::mutex> lk(w->mu); w->posted[c->rank] = g_ops; }
    w->barrier();
    for (size_t i = 0; i < g_ops.size(); ++i) {
        const Op& op = g_ops[i];
        if (op.kind == 0) {
            for (int p = 0; p < w->size; ++p) {
                const Op& peer = w->posted[p][i];                                           // same position in every rank's group
                if (hipMemcpyAsync(static_cast<char*>(op.recv) + (size_t)p * op.count * 8, peer.send, op.count * 8, hipMemcpyDeviceToDevice, g_stream) != hipSuccess)
                    return ncclUnhandledCudaError;
            }
        } else if (op.kind == 3) {
            // max of int32 over the ranks: every rank reads every peer's values and writes the maximum to its own output
            std::vector<int32_t> acc(op.count), tmp(op.count);
            for (int p = 0; p < w->size; ++p) {
                const Op& peer = w->posted[p][i];
                if (peer.kind != 3 || peer.count != op.count) return ncclInvalidArgument;
                if (hipMemcpy(tmp.data(), peer.send, op.count * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
                for (size_t k = 0; k < op.count; ++k) acc[k] = (p == 0 || tmp[k] > acc[k]) ? tmp[k] : acc[k];
            }
            if (hipMemcpy(op.recv, acc.data(), op.count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        } else if (op.kind == 2) {
            // my k-th receive from `peer` pairs with its k-th send to me
            int k = 0;
            for (size_t j = 0; j < i; ++j) if (g_ops[j].kind == 2 && g_ops[j].peer == op.peer) ++k;
            const Op* match = nullptr;
            for (const Op& s : w->posted[op.peer]) if (s.kind == 1 && s.peer == c->rank && k-- == 0) { match = &s; break; }
            if (!match || match->count != op.count) return ncclInvalidArgument;
            if (hipMemcpyAsync(op.recv, match->send, op.count * 8, hipMemcpyDeviceToDevice, g_stream) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    if (hipStreamSynchronize(g_stream) != hipSuccess) return ncclUnhandledCudaError;
    w->barrier();                                                                           // nobody reuses a send buffer before every peer has copied
    g_ops.clear();
    return ncclSuccess;
}
ncclResult_t add(Comm* c, hipStream_t s, const Op& op) {
    g_comm = c; g_stream = s;
    g_ops.push_back(op);
    return g_depth ? ncclSuccess : flush();
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 0, sizeof(*id)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(void** comm, int nranks, ncclUniqueId, int rank) {
    std::lock_guard<std::mutex> lk(g_registry_mu);
    if (!g_pending) { g_pending = new World(); g_pending->size = nranks; g_pending->posted.resize(nranks); }
    World* w = g_pending;
    if (++w->attached == nranks) g_pending = nullptr;
    *comm = new Comm{w, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(void* comm) { delete static_cast<Comm*>(comm); return ncclSuccess; }
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --g_depth == 0 ? flush() : ncclSuccess; }
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, void* comm, hipStream_t s) {
    return add(static_cast<Comm*>(comm), s, Op{0, send, recv, count, -1});
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, void* comm, hipStream_t s) {
    if (dt != ncclInt32 || op != ncclMax) return ncclInvalidArgument;            // the one form the drivers use
    return add(static_cast<Comm*>(comm), s, Op{3, send, recv, count, -1});
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t, int peer, void* comm, hipStream_t s) {
    return add(static_cast<Comm*>(comm), s, Op{1, send, nullptr, count, peer});
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t, int peer, void* comm, hipStream_t s) {
    return add(static_cast<Comm*>(comm), s, Op{2, nullptr, recv, count, peer});
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "mock rccl error"; }
}

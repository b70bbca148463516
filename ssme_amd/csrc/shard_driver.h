// shard_driver.h -- C++ host driver of the particle-sharded bootstrap filter over RCCL (included by pf_api.hip).
//
// SURVEY.md section 8e row 2 / BASELINE.json north star: "C++ host code owns ... and calls HIP kernels through a thin
// extern-"C" ABI ... one RCCL [collective] over xGMI per time step for the global log-weight sum and an all-to-all for
// particle redistribution".  One process per GPU; rank g owns B/world consecutive 2048-particle tiles.  Per time step,
// everything on ONE HIP stream and -- on the fast path -- without any host synchronisation:
//     two grouped ncclAllGather (tile sums, tile maxima: 16 bytes per tile), each straight into its final array
//     k_shard_plan (or k_level2_plan + k_shard_window_check above 1024 tiles): every rank's source-tile window [lo, hi] and
//         a device flag if a window leaves the fixed halo
//     grouped ncclSend / ncclRecv of the halo tiles (integer cdf + particles) with the two neighbouring ranks
//     k_filter_step on the rank's tiles, reading its window in place from the halo buffer
// A rank's flag says what ITS OWN workgroups saw (up to 1024 tiles no plan kernel runs, so nothing else knows), and the
// decision to run again must be the same on every rank -- otherwise one rank re-enters the collectives alone.  After the
// time loop the flags are therefore reduced over the ranks (one ncclAllReduce(max) of one int, still on the stream), and
// the reduced flag is what the host reads: if a window ever left the halo on ANY rank (very unbalanced weights), EVERY rank runs
// the series again on the exact path (the plan is downloaded every step and exactly the planned tiles travel, any rank to any rank).
// Results are bit-identical to the unsharded filter on both paths (RNG counters are global particle indices, the level-2
// is the same exact integer arithmetic on the gathered tile sums).
//
// RCCL is resolved at run time from what the process already has loaded (torch's librccl in the Python tests and
// bench.py; `librccl.so` from the loader path otherwise): the library itself does not link against it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and enums only

namespace ssme {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static void* rccl_sym(void*& lib, const char* name) {
    void* p = dlsym(RTLD_DEFAULT, name);                  // the RCCL this process already uses, if any
    if (p) return p;
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return lib ? dlsym(lib, name) : nullptr;
}
static const RcclApi& rccl() {
    static const RcclApi api = [] {
        RcclApi a;
        void* lib = nullptr;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(rccl_sym(lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(rccl_sym(lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(rccl_sym(lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(rccl_sym(lib, "ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(rccl_sym(lib, "ncclAllReduce"));
        a.Send = reinterpret_cast<decltype(a.Send)>(rccl_sym(lib, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(rccl_sym(lib, "ncclRecv"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(rccl_sym(lib, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(rccl_sym(lib, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(rccl_sym(lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return api;
}

// Does every rank's window [lo, hi] stay inside [first own tile - margin, last own tile + margin]?  lo_hi: [world][2]
// (k_shard_plan), or null: the per-tile ranges of k_level2_plan (window = [lo of the rank's first tile, hi of its last]).
__global__ void k_shard_window_check(const int32_t* lo_hi, const int32_t* l2_lo, const int32_t* l2_hi, int world, int Bl, int B, int margin,
                                     int32_t* flag, int32_t* stats /*[2]: max tiles needed left / right of the own range*/) {
    const int g = threadIdx.x;
    if (g >= world) return;
    const int last = (g + 1) * Bl - 1 < B - 1 ? (g + 1) * Bl - 1 : B - 1;       // the last rank may own fewer than Bl tiles
    const int lo = lo_hi ? lo_hi[2 * g] : l2_lo[(size_t)g * Bl];
    const int hi = lo_hi ? lo_hi[2 * g + 1] : l2_hi[last];
    const int left = g * Bl - lo, right = hi - last;
    if (left > margin || right > margin) atomicOr(flag, 1);
    if (left > 0) atomicMax(&stats[0], left);
    if (right > 0) atomicMax(&stats[1], right);
}

}  // namespace ssme

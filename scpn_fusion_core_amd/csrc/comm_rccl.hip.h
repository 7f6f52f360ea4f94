// comm_rccl.hip.h -- the ensemble gather over RCCL / xGMI behind the C ABI (no PyTorch anywhere on the path).
//
// The hot path shards across ensemble members (one process per GPU, no data-path collective); the only exchange step is the
// all-gather of the dense psi blocks at the end of a solve (the reference's only multi-device prototype is the
// additive-Schwarz sketch scpn-fusion-rs/crates/fusion-core/src/mpi_domain.rs:741-770, which this deliberately does not copy).
// librccl is loaded with dlopen at the first scpn_comm_* call, so single-GPU users never need it.  Rendezvous: rank 0 creates
// the 128-byte ncclUniqueId (scpn_comm_unique_id) and the host side passes it to the other ranks over a TCP socket on
// MASTER_ADDR:MASTER_PORT (scpn_fusion_core_amd/distributed.py) -- before any rank touches its GPU.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <vector>

// The handful of RCCL / NCCL ABI types this file needs, declared here so that the library (single-GPU solver included) builds on a
// ROCm install without the RCCL development headers; every entry point is resolved through dlsym at the first scpn_comm_* call.
// The values are NCCL's stable public ABI (nccl.h 2.x == rccl.h): ncclUniqueId = 128 opaque bytes passed BY VALUE to
// ncclCommInitRank, ncclSuccess = 0, ncclFloat64 = 8, ncclMax = 2.  scpn_comm_init checks ncclGetVersion() >= 2.0 at run time.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclDouble = 8;
constexpr ncclRedOp_t ncclMax = 2;

namespace comm {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct State {
    Api api;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = -1, world = 0, device = 0;
    double* d_scalar = nullptr;  // one device double for the scalar all-reduce / barrier
    double* send = nullptr;      // the all-gather's send block (grown only while the communicator's stream is idle)
    size_t send_cap = 0;         // ... in doubles
    std::vector<hipEvent_t> ready;  // one per solver group: "this group's psi is packed" (the gather waits on them, not the host)
};
inline State& state() {
    static State s;
    return s;
}

inline const char* load(Api& a) {  // nullptr on success, else what is missing
    if (a.lib) return nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
    }
    if (!a.lib) return "librccl.so not found (dlopen)";
#define SCPN_RCCL_SYM(field, sym)                                   \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym)); \
    if (!a.field) return "librccl.so lacks " sym;
    SCPN_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    SCPN_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    SCPN_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    SCPN_RCCL_SYM(CommCount, "ncclCommCount")
    SCPN_RCCL_SYM(CommUserRank, "ncclCommUserRank")
    SCPN_RCCL_SYM(GetVersion, "ncclGetVersion")
    SCPN_RCCL_SYM(AllGather, "ncclAllGather")
    SCPN_RCCL_SYM(AllReduce, "ncclAllReduce")
    SCPN_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef SCPN_RCCL_SYM
    return nullptr;
}

}  // namespace comm

// RCCL binding of the one exchange step of the hot path: the all-reduce of the flat outer-gradient buffer between the ranks of a
// node (one process per GPU; the reference gets it from PL's strategy="ddp", main.py:30-38: DDP's gradient all-reduce over NCCL).
// librccl is resolved with dlopen at first use — libmtts has no link-time dependency on it, and inside a torch process the soname
// resolves to the RCCL instance torch already loaded, so both share one library.  The unique id travels through the caller
// (any out-of-band channel: torch.distributed's store, MPI, a file); nothing here talks to the network itself.
#pragma once
#include <cstring>
#include <string>

#include "compat.h"

#if !defined(MTTS_EMU)
#include <dlfcn.h>
#endif

namespace mtts {

#if defined(MTTS_EMU)
// Emulator build (CPU test-suite only): a loop-back "communicator" that behaves like `world` ranks holding IDENTICAL data — the SUM is
// world x the local buffer.  It lets the CPU suite check the exchange logic (every float of the buffer reduced exactly once, whatever
// the bucket order) without RCCL: a bucket that is never sent stays 1 x, one sent twice becomes world^2 x.
__global__ void comm_emu_scale_kernel(float* p, long long n, float s) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] *= s;
}
#endif

constexpr int kNcclUniqueIdBytes = 128;       // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct NcclUniqueId { char internal[kNcclUniqueIdBytes]; };

struct Comm {
    void* lib = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    std::string err;
    // rccl.h prototypes (ncclResult_t = int, ncclDataType_t ncclFloat32 = 7, ncclRedOp_t ncclSum = 0)
    int (*get_id)(NcclUniqueId*) = nullptr;
    int (*init_rank)(void**, int, NcclUniqueId, int) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*destroy)(void*) = nullptr;
    const char* (*err_str)(int) = nullptr;

    int load() {
#if defined(MTTS_EMU)
        return 0;   // (loop-back communicator, see above)
#else
        if (lib) return 0;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("dlopen(librccl.so) failed: ") + dlerror(); return -1; }
        get_id = (int (*)(NcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
        init_rank = (int (*)(void**, int, NcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
        all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
        destroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        err_str = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        if (!get_id || !init_rank || !all_reduce || !destroy) { err = "librccl.so lacks an expected symbol"; return -1; }
        return 0;
#endif
    }
    int fail(const char* what, int rc) {
        err = std::string(what) + ": " + (err_str ? err_str(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
        return -1;
    }
    int unique_id(NcclUniqueId* id) {
        if (load()) return -1;
#if defined(MTTS_EMU)
        memset(id, 0, sizeof(*id));
        return 0;
#else
        const int rc = get_id(id);
        return rc ? fail("ncclGetUniqueId", rc) : 0;
#endif
    }
    int init(const NcclUniqueId& id, int rank_, int world_) {
        if (load()) return -1;
        if (comm) { err = "communicator already initialised"; return -1; }
        if (world_ < 1 || rank_ < 0 || rank_ >= world_) { err = "bad rank / world size"; return -1; }
#if defined(MTTS_EMU)
        (void)id;
        comm = this;
#else
        const int rc = init_rank(&comm, world_, id, rank_);
        if (rc) { comm = nullptr; return fail("ncclCommInitRank", rc); }
#endif
        rank = rank_; world = world_;
        return 0;
    }
    int sum(float* buf, size_t n, hipStream_t stream) {
        if (!comm) { err = "communicator not initialised (mtts_comm_init)"; return -1; }
#if defined(MTTS_EMU)
        MTTS_LAUNCH(comm_emu_scale_kernel, dim3(64), dim3(256), stream, buf, (long long)n, (float)world);
        return 0;
#else
        const int rc = all_reduce(buf, buf, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, stream);
        return rc ? fail("ncclAllReduce", rc) : 0;
#endif
    }
    void release() {
#if !defined(MTTS_EMU)
        if (comm && destroy) destroy(comm);
#endif
        comm = nullptr;
    }
};

}  // namespace mtts

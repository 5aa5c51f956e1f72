// w3_rccl.h — RCCL behind a lazily resolved symbol table (w3_encode_blocks_sharded_device).
// libw3hip.so does not link librccl: a single-GPU host need not have it.  The first multi-device gather dlopen()s
// "librccl.so.1" — the soname of /opt/rocm's library and of the copy PyTorch bundles, so a process that has loaded torch's
// RCCL gets that one back — and resolves the eight entry points used here.  Prototypes as in <rccl/rccl.h> (ROCm 7.2).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

namespace w3rccl {

typedef void *comm_t;                      // ncclComm_t (opaque)
enum { kSuccess = 0, kUint8 = 1, kUint64 = 5 };   // ncclSuccess, ncclUint8, ncclUint64 (rccl.h:52, :460, :464)

struct Api {
    int (*CommInitAll)(comm_t *comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(comm_t comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, comm_t comm, hipStream_t s) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, comm_t comm, hipStream_t s) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, comm_t comm, hipStream_t s) = nullptr;
    const char *(*GetErrorString)(int rc) = nullptr;
    void *handle = nullptr;
    std::string error;
};

static inline std::string &library_override() { static std::string s; return s; }
static inline bool &resolved() { static bool r = false; return r; }

static inline Api *api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        resolved() = true;
        // w3_rccl_library(path): the one library to try instead of the usual sonames (a host whose RCCL lives elsewhere; the tests
        // name a file that does not exist to take the "not available" path on a machine that has RCCL)
        const char *over = library_override().empty() ? nullptr : library_override().c_str();
        const char *usual[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const char *one[] = {over};
        const char *const *names = over && *over ? one : usual;
        const int n_names = over && *over ? 1 : 3;
        for (int i = 0; i < n_names && !a.handle; i++)
            a.handle = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // a copy the process has loaded already (torch's) first
        for (int i = 0; i < n_names && !a.handle; i++)
            a.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!a.handle) {
            const char *e = dlerror();   // (once: the call clears the error it returns)
            a.error = std::string("RCCL not available: ") + (e ? e : "dlopen failed");
            return;
        }
#define W3_RCCL_SYM(field, name)                                                                          \
        do {                                                                                               \
            *(void **)(&a.field) = dlsym(a.handle, name);                                                  \
            if (!a.field && a.error.empty()) a.error = std::string("RCCL symbol missing: ") + name;        \
        } while (0)
        W3_RCCL_SYM(CommInitAll, "ncclCommInitAll");
        W3_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        W3_RCCL_SYM(GroupStart, "ncclGroupStart");
        W3_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        W3_RCCL_SYM(Send, "ncclSend");
        W3_RCCL_SYM(Recv, "ncclRecv");
        W3_RCCL_SYM(AllGather, "ncclAllGather");
        W3_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef W3_RCCL_SYM
    });
    return &a;
}

}  // namespace w3rccl

// Stand-in for <rccl/rccl.h> in the CPU builds of tests/emu. The device library opens RCCL with dlopen and binds six entry
// points with dlsym; here dlopen / dlsym are redirected to an in-process implementation of those six, so that a group of
// members on distinct (emulated) devices runs its ncclReduce branch: one reduce per member inside ncclGroupStart / End,
// each queued on the member's own stream; the root's stream waits for every member's stream up to its call and then sums
// the members' send buffers in rank order into the root's receive buffer.
#pragma once
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../hip/hip_runtime.h"

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
struct EmuNcclComm {
    int tag, rank, nRanks, device;
    struct EmuNcclGroup *group;
};
typedef EmuNcclComm *ncclComm_t;
struct EmuNcclGroup {
    std::vector<EmuNcclComm *> comms;
};

namespace emu_rccl {
struct Call {
    const void *send;
    void *recv;
    size_t count;
    int root;
    EmuNcclComm *comm;
    hipStream_t stream;
};
inline int groupDepth = 0;
inline std::vector<Call> calls;

inline ncclResult_t commInitAll(ncclComm_t *comms, int n, const int *devices) {
    if (!comms || n < 1) return ncclInvalidArgument;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < k; ++j)
            if (devices && devices[k] == devices[j]) return ncclInvalidUsage;   // RCCL refuses a device listed twice
    EmuNcclGroup *g = new EmuNcclGroup;
    for (int k = 0; k < n; ++k) {
        comms[k] = new EmuNcclComm{3, k, n, devices ? devices[k] : k, g};
        g->comms.push_back(comms[k]);
    }
    return ncclSuccess;
}
inline ncclResult_t commDestroy(ncclComm_t c) {
    if (!c || c->tag != 3) return ncclInvalidArgument;
    EmuNcclGroup *g = c->group;
    c->tag = 0;
    size_t live = 0;
    for (EmuNcclComm *m : g->comms) live += m->tag == 3;
    if (live == 0) {
        for (EmuNcclComm *m : g->comms) delete m;
        delete g;
    }
    return ncclSuccess;
}
inline ncclResult_t flush() {
    if (calls.empty()) return ncclSuccess;
    const int n = calls[0].comm->nRanks, root = calls[0].root;
    const size_t count = calls[0].count;
    if ((int)calls.size() != n) { calls.clear(); return ncclInvalidUsage; }   // a collective needs every rank
    std::vector<Call> byRank((size_t)n);
    for (const Call &c : calls) {
        if (c.comm->group != calls[0].comm->group || c.count != count || c.root != root) return ncclInvalidUsage;
        byRank[(size_t)c.comm->rank] = c;
    }
    // the root's part runs after every member's stream has reached its call
    std::vector<hipEvent_t> reached;
    for (int r = 0; r < n; ++r) {
        if (r == root) continue;
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, byRank[(size_t)r].stream);
        hipStreamWaitEvent(byRank[(size_t)root].stream, e, 0);
        reached.push_back(e);
    }
    emu_rt::submit(byRank[(size_t)root].stream, EmuItem{EmuItem::Work, [byRank, n, root, count]() {
                                                           std::vector<float> sum(count, 0.f);
                                                           for (int r = 0; r < n; ++r) {
                                                               const float *s = static_cast<const float *>(byRank[(size_t)r].send);
                                                               for (size_t i = 0; i < count; ++i) sum[i] = r == 0 ? s[i] : sum[i] + s[i];
                                                           }
                                                           std::memcpy(byRank[(size_t)root].recv, sum.data(), count * sizeof(float));
                                                       },
                                                       nullptr, 0});
    // the other members' streams do not continue before the root has read their buffers
    hipEvent_t done;
    hipEventCreate(&done);
    hipEventRecord(done, byRank[(size_t)root].stream);
    for (int r = 0; r < n; ++r)
        if (r != root) hipStreamWaitEvent(byRank[(size_t)r].stream, done, 0);
    emu_rt::retired.insert(emu_rt::retired.end(), reached.begin(), reached.end());
    emu_rt::retired.push_back(done);
    calls.clear();
    return ncclSuccess;
}
inline ncclResult_t groupStart() { ++groupDepth; return ncclSuccess; }
inline ncclResult_t groupEnd() {
    if (groupDepth <= 0) return ncclInvalidUsage;
    return --groupDepth == 0 ? flush() : ncclSuccess;
}
// What a process must not leave behind: a group that was started and never ended, calls that never formed a collective.
struct AtExit {
    ~AtExit() {
        if (groupDepth != 0) simt::fatal("ncclGroupStart without ncclGroupEnd at the end of the process");
    }
};
inline AtExit atExit;
inline ncclResult_t reduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
    static const bool failing = std::getenv("EMU_RCCL_REDUCE_FAILS") != nullptr;   // for the test of the library's error path
    if (failing) return ncclUnhandledCudaError;
    if (!comm || comm->tag != 3 || !send || type != ncclFloat || op != ncclSum || root < 0 || root >= comm->nRanks) return ncclInvalidArgument;
    if (comm->rank == root && !recv) return ncclInvalidArgument;
    calls.push_back(Call{send, recv, count, root, comm, stream});
    return groupDepth > 0 ? ncclSuccess : flush();
}
inline const char *errorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage" : "invalid argument"; }

inline int handle = 0;
inline void *open(const char *name, int flags) {
    if (!name || !std::strstr(name, "rccl")) return ::dlopen(name, flags);
    return (flags & RTLD_NOLOAD) ? nullptr : &handle;   // not loaded yet, then found
}
inline void *sym(void *h, const char *name) {
    if (h != &handle) return ::dlsym(h, name);
    if (!std::strcmp(name, "ncclCommInitAll")) return reinterpret_cast<void *>(&commInitAll);
    if (!std::strcmp(name, "ncclCommDestroy")) return reinterpret_cast<void *>(&commDestroy);
    if (!std::strcmp(name, "ncclReduce")) return reinterpret_cast<void *>(&reduce);
    if (!std::strcmp(name, "ncclGroupStart")) return reinterpret_cast<void *>(&groupStart);
    if (!std::strcmp(name, "ncclGroupEnd")) return reinterpret_cast<void *>(&groupEnd);
    if (!std::strcmp(name, "ncclGetErrorString")) return reinterpret_cast<void *>(&errorString);
    return nullptr;
}
}   // namespace emu_rccl

#define dlopen(name, flags) emu_rccl::open((name), (flags))
#define dlsym(handle, name) emu_rccl::sym((handle), (name))

// One wavefront on the CPU: 64 lanes as 64 fibres (ucontext) on one thread. A lane runs until it reaches a wave-wide
// operation (ballot, wave barrier) or returns; when every live lane has stopped, the operation is evaluated - all lanes
// must have stopped at the same source line, as lanes of a real wave that execute one instruction together - and the
// lanes continue. Memory shared by lanes (the LDS arrays) therefore has the visibility the kernels rely on: what a lane
// wrote before a wave-wide operation is seen by every lane after it. Test infrastructure.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <vector>

namespace simt {

struct Dim3 { unsigned x, y, z; };

struct Lane {
    ucontext_t ctx;
    Dim3 threadIdx_;
    bool done;
    int waitingAt;      // source line of the wave-wide operation the lane stopped at, 0 = none
    bool pred;
    unsigned long long result;
    std::vector<char> stack;
};

struct Launch { Dim3 blockIdx_, blockDim_, gridDim_; };

inline Lane *current = nullptr;
inline Launch launch = {};
inline ucontext_t scheduler;
inline std::function<void()> *body = nullptr;
inline unsigned long long waveOps = 0;   // wave-wide operations evaluated so far (a run-away loop shows here)
inline std::map<int, unsigned long long> opsAtLine;   // the same by source line (negative: wave barriers)

[[noreturn]] inline void unsupported(const char *what) {
    std::fprintf(stderr, "simt: %s is not emulated\n", what);
    std::abort();
}

inline unsigned long long ballot(bool pred, int line) {
    current->pred = pred;
    current->waitingAt = line;
    swapcontext(&current->ctx, &scheduler);
    return current->result;
}

inline void wave_barrier(int line) {
    current->pred = false;
    current->waitingAt = -line;
    swapcontext(&current->ctx, &scheduler);
}

inline void trampoline() {
    (*body)();
    current->done = true;
    swapcontext(&current->ctx, &scheduler);
}

// Runs `fn` on the 64 lanes of wave `wave` of the current block (threadIdx.x = wave * 64 + lane). Returns false when the
// lanes diverged at a wave-wide operation or `maxOps` operations went by without the wave ending.
inline bool run_wave(unsigned wave, std::function<void()> fn, unsigned long long maxOps) {
    std::vector<Lane> lanes(64);
    body = &fn;
    for (unsigned l = 0; l < 64; ++l) {
        Lane &L = lanes[l];
        L.threadIdx_ = Dim3{wave * 64 + l, 0, 0};
        L.done = false;
        L.waitingAt = 0;
        L.stack.resize(256 * 1024);
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = L.stack.data();
        L.ctx.uc_stack.ss_size = L.stack.size();
        L.ctx.uc_link = &scheduler;
        makecontext(&L.ctx, trampoline, 0);
    }
    unsigned long long ops = 0;
    for (;;) {
        unsigned live = 0;
        for (unsigned l = 0; l < 64; ++l) {
            if (lanes[l].done) continue;
            current = &lanes[l];
            swapcontext(&scheduler, &lanes[l].ctx);
            if (!lanes[l].done) ++live;
        }
        if (live == 0) break;
        int at = 0;
        unsigned long long mask = 0;
        for (unsigned l = 0; l < 64; ++l) {
            if (lanes[l].done) continue;
            if (at == 0) at = lanes[l].waitingAt;
            if (lanes[l].waitingAt != at) {
                std::fprintf(stderr, "simt: lanes diverged at a wave-wide operation (lines %d and %d)\n", at, lanes[l].waitingAt);
                return false;
            }
            if (lanes[l].pred) mask |= 1ull << l;
        }
        for (unsigned l = 0; l < 64; ++l) {
            lanes[l].result = mask;
            lanes[l].waitingAt = 0;
        }
        ++waveOps;
        ++opsAtLine[at];
        if (++ops > maxOps) {
            std::fprintf(stderr, "simt: wave still running after %llu wave-wide operations\n", ops);
            return false;
        }
    }
    current = nullptr;
    return true;
}

}   // namespace simt

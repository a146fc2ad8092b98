// A workgroup on the CPU: every work-item is a fibre (ucontext) on one thread, 64 consecutive fibres form a wave. A lane
// runs until it reaches a wave-wide operation (ballot, wave barrier), a workgroup barrier, or returns. A wave-wide
// operation is evaluated when every live lane of the wave has stopped at it - all at the same source line, as lanes of a
// real wave that execute one instruction together; a workgroup barrier opens when every live work-item has reached it.
// Memory shared by lanes (LDS arrays) therefore has the visibility the kernels rely on: what a lane wrote before such an
// operation is seen by the lanes that take part in it afterwards. Workgroups of a grid run one after the other. Test
// infrastructure: a checker for the library's device and host code, never a way to run the library without a GPU.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <vector>

namespace simt {

struct Dim3 {
    unsigned x, y, z;
};

struct Lane {
    ucontext_t ctx;
    Dim3 threadIdx_;
    bool done;
    int waitingAt;   // 0: runnable; > 0: source line of a ballot; < 0: minus the line of a wave barrier; kBlockBarrier + line: __syncthreads
    bool pred;
    unsigned long long result;
};

struct Launch {
    Dim3 blockIdx_, blockDim_, gridDim_;
};

constexpr int kBlockBarrier = 1 << 24;
constexpr size_t kStackBytes = 96 * 1024;   // AddressSanitizer clears the shadow of a fibre's whole stack at every switch: keep it small

inline Lane *current = nullptr;
inline Launch launch = {};
inline ucontext_t scheduler;
inline std::function<void()> *body = nullptr;
inline unsigned long long waveOps = 0;                  // wave-wide operations evaluated so far
inline std::map<int, unsigned long long> opsAtLine;     // the same by source line (negative: wave barriers)
inline std::map<int, unsigned long long> lanesAtLine;   // lanes whose predicate was true, summed over the ballots of a line
inline std::vector<std::unique_ptr<char[]>> stacks;     // fibre stacks, kept between workgroups
inline unsigned char *dynamicLds = nullptr;             // the running workgroup's dynamic LDS

[[noreturn]] inline void fatal(const char *what) {
    std::fprintf(stderr, "simt: %s\n", what);
    std::abort();
}

inline void stop_at(int where, bool pred) {
    current->pred = pred;
    current->waitingAt = where;
    swapcontext(&current->ctx, &scheduler);
}
inline unsigned long long ballot(bool pred, int line) {
    stop_at(line, pred);
    return current->result;
}
inline void wave_barrier(int line) { stop_at(-line, false); }
inline void block_barrier(int line) { stop_at(kBlockBarrier + line, false); }

inline void trampoline() {
    (*body)();
    current->done = true;
    swapcontext(&current->ctx, &scheduler);
}

// Runs `fn` on the work-items [first, first + count) of the current workgroup (count a multiple of 64 or the tail of the
// workgroup). Returns false when lanes diverged at a wave-wide operation, when nobody can make progress, or when `maxOps`
// wave-wide operations went by without the work-items ending.
inline bool run_items(unsigned first, unsigned count, std::function<void()> fn, unsigned long long maxOps) {
    std::vector<Lane> lanes(count);
    while (stacks.size() < count) stacks.emplace_back(new char[kStackBytes]);
    body = &fn;
    for (unsigned l = 0; l < count; ++l) {
        Lane &L = lanes[l];
        L.threadIdx_ = Dim3{first + l, 0, 0};
        L.done = false;
        L.waitingAt = 0;
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = stacks[l].get();
        L.ctx.uc_stack.ss_size = kStackBytes;
        L.ctx.uc_link = &scheduler;
        makecontext(&L.ctx, trampoline, 0);
    }
    unsigned long long ops = 0;
    for (;;) {
        bool progressed = false;
        unsigned live = 0;
        // between two synchronising operations the order in which lanes run must not matter: the second schedule runs them
        // from the last to the first
        static const bool lastFirst = std::getenv("EMU_LAST_WORKGROUP_FIRST") != nullptr;
        for (unsigned i = 0; i < count; ++i) {
            const unsigned l = lastFirst ? count - 1 - i : i;
            if (lanes[l].done) continue;
            if (lanes[l].waitingAt == 0) {
                current = &lanes[l];
                swapcontext(&scheduler, &lanes[l].ctx);
                progressed = true;
            }
            if (!lanes[l].done) ++live;
        }
        if (live == 0) break;
        // wave-wide operations, wave by wave
        for (unsigned w0 = 0; w0 < count; w0 += 64) {
            const unsigned w1 = w0 + 64 < count ? w0 + 64 : count;
            int at = 0;
            bool ready = true, any = false;
            unsigned long long mask = 0;
            for (unsigned l = w0; l < w1; ++l) {
                if (lanes[l].done) continue;
                const int wa = lanes[l].waitingAt;
                if (wa == 0 || wa >= kBlockBarrier) { ready = false; break; }
                if (!any) { at = wa; any = true; }
                if (wa != at) {
                    std::fprintf(stderr, "simt: lanes of one wave stopped at different wave-wide operations (lines %d and %d)\n", at, wa);
                    return false;
                }
                if (lanes[l].pred) mask |= 1ull << (l - w0);
            }
            if (!ready || !any) continue;
            for (unsigned l = w0; l < w1; ++l) {
                lanes[l].result = mask;
                lanes[l].waitingAt = 0;
            }
            ++waveOps;
            ++opsAtLine[at];
            lanesAtLine[at] += (unsigned long long)__builtin_popcountll(mask);
            progressed = true;
            if (++ops > maxOps) {
                std::fprintf(stderr, "simt: work-items still running after %llu wave-wide operations\n", ops);
                return false;
            }
        }
        // workgroup barrier
        {
            int at = 0;
            bool ready = true, any = false;
            for (unsigned l = 0; l < count; ++l) {
                if (lanes[l].done) continue;
                const int wa = lanes[l].waitingAt;
                if (wa < kBlockBarrier) { ready = false; break; }
                if (!any) { at = wa; any = true; }
                if (wa != at) {
                    std::fprintf(stderr, "simt: work-items stopped at different workgroup barriers (lines %d and %d)\n", at - kBlockBarrier, wa - kBlockBarrier);
                    return false;
                }
            }
            if (ready && any) {
                for (unsigned l = 0; l < count; ++l) lanes[l].waitingAt = 0;
                progressed = true;
            }
        }
        if (!progressed) {
            std::fprintf(stderr, "simt: deadlock: part of a workgroup waits at a barrier that the rest never reaches\n");
            return false;
        }
    }
    current = nullptr;
    return true;
}

// One wave of the current workgroup (threadIdx.x = wave * 64 + lane).
inline bool run_wave(unsigned wave, std::function<void()> fn, unsigned long long maxOps) { return run_items(wave * 64, 64, std::move(fn), maxOps); }

// A whole grid: `fn` is the kernel call. Dynamic LDS is a heap block of exactly `ldsBytes`, refilled with a pattern for
// every workgroup (LDS is not zero at the start of a workgroup).
inline bool run_grid(Dim3 grid, Dim3 block, size_t ldsBytes, const std::function<void()> &fn, unsigned long long maxOps = 1ull << 40) {
    if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1) fatal("only one-dimensional launches are emulated");
    launch.gridDim_ = grid;
    launch.blockDim_ = block;
    std::unique_ptr<unsigned char[]> lds(new unsigned char[ldsBytes ? ldsBytes : 1]);
    static const bool lastFirst = std::getenv("EMU_LAST_WORKGROUP_FIRST") != nullptr;   // a grid promises no order among workgroups
    for (unsigned i = 0; i < grid.x; ++i) {
        const unsigned b = lastFirst ? grid.x - 1 - i : i;
        launch.blockIdx_ = Dim3{b, 0, 0};
        for (size_t i = 0; i < ldsBytes; ++i) lds[i] = 0xCD;
        dynamicLds = lds.get();
        if (!run_items(0, block.x, fn, maxOps)) return false;
    }
    dynamicLds = nullptr;
    return true;
}

}   // namespace simt

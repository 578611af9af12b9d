// pgo_comm_local.hpp — the in-process communicator behind pgo_comm_init_local (include/pgo.h): the ranks are handles of ONE process, each driven by its own host thread.
// Included by pgo_solver.hip only.
//
// What it replaces in the reference: nothing — the reference solves on one CPU (ceres::Solve, src/PoseGraphSLAM.cpp:1903).  It is the third transport of the sharded solver
// next to RCCL (one process per GPU) and a caller-supplied collective: a collective here is a KERNEL that reads the peers' device buffers directly (one GPU: the same address
// space; several GPUs of one process: peer access over xGMI), ordered by HIP events between the handles' streams.  The host threads only meet at a barrier so that every
// rank's event has been recorded before a peer waits on it.  No host staging, no copies through pinned memory.
//
// Protocol of collective number k (every rank issues the same collectives in the same order; parity = k & 1):
//   1. wait (stream) on the peers' done[parity] events: their reads of THIS rank's parity buffer in collective k - 2 are finished — the buffer may be overwritten
//   2. fill the parity buffer (all-reduce: a copy of the operand; exchange: the solver's pack kernel), record ready[parity], publish the pointer
//   3. host barrier
//   4. wait (stream) on the peers' ready[parity], launch the reading kernel (sum / max in rank order, or the peers' segments copied into the receive buffer), record done[parity]
// One barrier per collective: double buffering by parity makes the done events of collective k - 2 visible (they were recorded before their owner entered barrier k - 1).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>

namespace pgo_local {

constexpr int MAX_RANKS = 16;

struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    struct Slot {
        bool joined = false;
        int device = 0;
        const double* ptr[2] = {nullptr, nullptr};          // the parity buffer published for the collective in flight
        const int64_t* send_off[2] = {nullptr, nullptr};    // exchange: the publisher's segment bounds (host array, stable while the plan lives)
        hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
        double* stage[2] = {nullptr, nullptr};               // all-reduce: copies of the operand
        size_t stage_cap[2] = {0, 0};
    } slot[MAX_RANKS];

    // all ranks arrive or the group is broken (a rank failed / left, or nobody came for two minutes): false
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&]() { return generation != gen || broken; });
        if (!ok) { broken = true; cv.notify_all(); return false; }
        return !broken || generation != gen;
    }
    void abort() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }
};

}  // namespace pgo_local

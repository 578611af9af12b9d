// Microbenchmark (measurement only): what does a GRID-WIDE barrier cost inside one persistent kernel on MI355X, compared with a kernel boundary?
//   variant A: every workgroup of the launch takes part (all 8 XCDs): agent-scope release/acquire on a counter
//   variant B: only the workgroups that landed on ONE XCD take part (HW_REG_XCC_ID); the others leave at once.  Inside an XCD the 4 MiB L2 is the
//              single point of coherence, so the barrier needs no L2 write-back: data goes through L2 with agent-scope relaxed loads/stores.
// Each round every participant publishes a value, passes the barrier and checks its neighbour's value (visibility is verified, not assumed).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/grid_barrier.hip -o scripts/microbench/grid_barrier ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define XCC_ID_REG ((3 << 11) | (0 << 6) | 20)   // s_getreg_b32 HW_REG_XCC_ID, bits [3:0]

struct Ctl { unsigned arrived; unsigned slots; unsigned counter; unsigned errors; unsigned n_part; unsigned pad[11]; };

template <bool ONE_XCD, int PAYLOAD, int MODE = 0>   // MODE 0: as described above; 1: one XCD, plain loads/stores + agent-scope release/acquire fences; 2: plain accesses, workgroup release + agent acquire (L1 invalidate only)
__global__ __launch_bounds__(256) void k_barrier(Ctl* c, double* buf, int rounds, int want_xcd) {
    __shared__ unsigned sh_slot, sh_n;
    bool part = true;
    if (ONE_XCD) part = ((int)(__builtin_amdgcn_s_getreg(XCC_ID_REG) & 15) == want_xcd);
    if (threadIdx.x == 0) {
        unsigned slot = 0xffffffffu;
        if (part) slot = __hip_atomic_fetch_add(&c->slots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (part) {
            while (__hip_atomic_load(&c->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
            sh_n = __hip_atomic_load(&c->slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sh_slot = slot;
    }
    __syncthreads();
    if (!part) return;
    const unsigned slot = sh_slot, n = sh_n;
    if (slot == 0 && threadIdx.x == 0) c->n_part = n;
    unsigned errs = 0;
    for (int r = 0; r < rounds; ++r) {
        // publish PAYLOAD doubles per thread
#pragma unroll
        for (int k = 0; k < PAYLOAD; ++k) {
            double* p = buf + ((size_t)slot * PAYLOAD + k) * 256 + threadIdx.x;
            if (ONE_XCD && MODE == 0) __hip_atomic_store(p, (double)(r + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = (double)(r + k);
        }
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else if (ONE_XCD) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // waits for the stores to be issued to L2 (vL1D is write-through)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(r + 1) * n;
            if (ONE_XCD) {
                __builtin_amdgcn_s_waitcnt(0);
                __hip_atomic_fetch_add(&c->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&c->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
            } else {
                __hip_atomic_fetch_add(&c->counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&c->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {}
            }
        }
        __syncthreads();
        if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned nb = (slot + 1) % n;
#pragma unroll
        for (int k = 0; k < PAYLOAD; ++k) {
            const double* p = buf + ((size_t)nb * PAYLOAD + k) * 256 + threadIdx.x;
            double v;
            if (ONE_XCD && MODE == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE >= 1) v = *p;
            else v = *(volatile const double*)p;
            // the neighbour may already be one round ahead
            if (v != (double)(r + k) && v != (double)(r + 1 + k)) ++errs;
        }
    }
    if (errs) atomicAdd(&c->errors, errs);
}

template <bool ONE_XCD, int PAYLOAD, int MODE = 0>
void run(int grid, int rounds, int want_xcd) {
    Ctl* c; double* buf; hipMalloc(&c, sizeof(Ctl)); hipMalloc(&buf, (size_t)4096 * PAYLOAD * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; Ctl h{};
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(c, 0, sizeof(Ctl));
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_barrier<ONE_XCD, PAYLOAD, MODE>), dim3(grid), dim3(256), 0, 0, c, buf, rounds, want_xcd);
        hipEventRecord(e1, 0);
        if (hipStreamSynchronize(0) != hipSuccess) { printf("launch failed\n"); exit(1); }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
    }
    printf("mode %d %s grid %4d payload %d doubles/thread: %4u participants, %.2f us per round (publish + barrier + read), visibility errors %u\n",
           MODE, ONE_XCD ? "one XCD " : "all XCDs", grid, PAYLOAD, h.n_part, best * 1e3f / rounds, h.errors);
    hipFree(c); hipFree(buf);
}
int main() {
    const int rounds = 2000;
    run<false, 1>(32, rounds, 0); run<false, 1>(64, rounds, 0); run<false, 1>(128, rounds, 0); run<false, 1>(256, rounds, 0); run<false, 4>(256, rounds, 0);
    run<true, 1>(64, rounds, 0); run<true, 1>(128, rounds, 0); run<true, 1>(256, rounds, 0); run<true, 4>(256, rounds, 0); run<true, 1>(512, rounds, 0); run<true, 1>(256, rounds, 3);
    run<true, 1, 1>(256, rounds, 0); run<true, 4, 1>(256, rounds, 0); run<true, 1, 2>(256, rounds, 0); run<true, 4, 2>(256, rounds, 0); run<true, 4, 2>(128, rounds, 0);
    return 0;
}

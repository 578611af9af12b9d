// Micro-benchmark (development aid, not product): what does a grid-wide barrier inside ONE launch cost on gfx950 against a kernel boundary inside a hipGraph?
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/grid_barrier_bench scripts/microbench/grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, long long deadline) {
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() > deadline) return false;
    }
    return true;
}
// A: one counter, every workgroup adds 1, thread 0 spins
__device__ __forceinline__ bool barrier_counter(unsigned* ctr, unsigned epoch, long long deadline) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);      // agent scope by default for __atomic_thread_fence in HIP device code? use the explicit builtin below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(ctr, epoch * gridDim.x, deadline);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
// B: one flag per workgroup (own cache line group of 4 B, packed), all threads poll their share
__device__ __forceinline__ bool barrier_flags(unsigned* flags, unsigned epoch, long long deadline) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool ok = true;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) ok = ok && spin_until(flags + i, epoch, deadline);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = __syncthreads_and(ok);
    return ok;
}
// C: two levels: groups of 32 workgroups add to their group's counter; the last arriver of a group adds to the top counter; thread 0 spins on the top counter
__device__ __forceinline__ bool barrier_tree(unsigned* ctr, unsigned epoch, long long deadline) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x >> 5, ng = (gridDim.x + 31) >> 5;
        const unsigned members = (g + 1 == ng) ? gridDim.x - (g << 5) : 32u;
        const unsigned old = __hip_atomic_fetch_add(ctr + 64 + g * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch * members) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(ctr, epoch * ng, deadline);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
template <int KIND>
__global__ void barrier_kernel(unsigned* sync, double* data, int rounds, int* fail, long long budget_ticks) {
    const long long deadline = wall_clock64() + budget_ticks;
    const int n = gridDim.x;
    double acc = 0.0;
    for (int r = 1; r <= rounds; ++r) {
        // each workgroup publishes a value, then (after the barrier) reads its right neighbour's — checks visibility across XCDs
        if (threadIdx.x == 0) data[(size_t)(r & 1) * n + blockIdx.x] = (double)r * 1000.0 + blockIdx.x;
        bool ok;
        if (KIND == 0) ok = barrier_counter(sync, (unsigned)r, deadline);
        else if (KIND == 1) ok = barrier_flags(sync + 4096, (unsigned)r, deadline);
        else ok = barrier_tree(sync + 8192, (unsigned)r, deadline);
        if (!ok) { if (threadIdx.x == 0) atomicExch(fail, 1); return; }
        if (threadIdx.x == 0) {
            const int nb = (blockIdx.x + 97) % n;
            const double v = data[(size_t)(r & 1) * n + nb];
            if (v != (double)r * 1000.0 + nb) atomicExch(fail, 2);
            acc += v;
        }
    }
    if (threadIdx.x == 0 && acc == -1.0) data[0] = acc;
}
__global__ void step_kernel(double* data, int r, int n) {
    if (threadIdx.x == 0) {
        const int nb = (blockIdx.x + 97) % n;
        const double v = data[(size_t)((r - 1) & 1) * n + nb];
        data[(size_t)(r & 1) * n + blockIdx.x] = v * 0.0 + (double)r * 1000.0 + blockIdx.x;
    }
}
int main(int argc, char** argv) {
    const int rounds = 400;
    unsigned* sync; double* data; int* fail;
    CHK(hipMalloc(&sync, 1 << 20)); CHK(hipMalloc(&data, 1 << 20)); CHK(hipMalloc(&fail, 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grids[] = {64, 256, 391, 512, 1024};
    for (int g : grids) {
        for (int kind = 0; kind < 3; ++kind) {
            float best = 1e30f; int h_fail = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CHK(hipMemsetAsync(sync, 0, 1 << 20, st)); CHK(hipMemsetAsync(fail, 0, 4, st));
                CHK(hipEventRecord(e0, st));
                const long long budget = 100000000LL / 10 * 2;    // 0.2 s of the 100-MHz clock
                if (kind == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(g), dim3(192), 0, st, sync, data, rounds, fail, budget);
                else if (kind == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(g), dim3(192), 0, st, sync, data, rounds, fail, budget);
                else hipLaunchKernelGGL(barrier_kernel<2>, dim3(g), dim3(192), 0, st, sync, data, rounds, fail, budget);
                CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                CHK(hipMemcpy(&h_fail, fail, 4, hipMemcpyDeviceToHost));
                if (h_fail) break;
            }
            std::printf("grid %4d x 192, %-8s: %.3f us per barrier%s\n", g, kind == 0 ? "counter" : kind == 1 ? "flags" : "tree", best * 1000.0f / rounds,
                        h_fail == 1 ? "  TIMED OUT" : h_fail == 2 ? "  STALE DATA SEEN" : "");
        }
        // the same chain as kernel boundaries inside a hipGraph
        hipGraph_t graph; hipGraphExec_t exec;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int r = 1; r <= rounds; ++r) hipLaunchKernelGGL(step_kernel, dim3(g), dim3(192), 0, st, data, r, g);
        CHK(hipStreamEndCapture(st, &graph)); CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(exec, st)); CHK(hipEventRecord(e1, st)); CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        std::printf("grid %4d x 192, kernel boundary in a hipGraph: %.3f us per kernel\n", g, best * 1000.0f / rounds);
        CHK(hipGraphExecDestroy(exec)); CHK(hipGraphDestroy(graph));
    }
    return 0;
}

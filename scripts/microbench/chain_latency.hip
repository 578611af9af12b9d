// Microbenchmark (measurement only): what does a SMALL dependent kernel cost on MI355X inside a stream of dependent kernels?
//   empty kernel / kernel with a chain of d dependent global loads (pointer chasing over a buffer written by the PREVIOUS kernel, so that every
//   kernel starts on cold, invalidated L2s like the multigrid level kernels) / the same with 3 __syncthreads in between.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/chain_latency.hip -o scripts/microbench/chain_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void k_empty(int* out) { if (threadIdx.x == 9999) out[0] = 1; }
template <int D, bool BAR>
__global__ __launch_bounds__(192) void k_chain(const int* __restrict__ next, int* __restrict__ out, int n) {
    __shared__ int sh[192];
    int i = (blockIdx.x * 192 + threadIdx.x) % n;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        i = next[i];
        if (BAR && d < 3) { sh[threadIdx.x] = i; __syncthreads(); i = sh[(threadIdx.x + 1) % 192]; __syncthreads(); }
    }
    out[blockIdx.x * 192 + threadIdx.x] = i;
}
__global__ void k_touch(int* next, int n, int salt) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) next[i] = (int)(((long long)next[i] + 0) % n); (void)salt; }
template <int D, bool BAR>
float run(int wgs, const int* next, int* out, int n, hipStream_t st, int reps, bool touch, int* next_rw) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_chain<D, BAR>), dim3(wgs), dim3(192), 0, st, next, out, n);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) { if (touch) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, out); hipLaunchKernelGGL((k_chain<D, BAR>), dim3(wgs), dim3(192), 0, st, next, out, n); }
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1); (void)next_rw;
    return ms * 1e3f / reps;
}
int main() {
    const int n = 1 << 22;   // 16 MB of indices: beyond the 4 MiB L2 of an XCD, inside the Infinity Cache
    std::vector<int> h(n); std::iota(h.begin(), h.end(), 0); std::mt19937 rng(1); std::shuffle(h.begin(), h.end(), rng);
    int *next, *out; hipMalloc(&next, n * 4); hipMalloc(&out, 4096 * 192 * 4);
    hipMemcpy(next, h.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st); for (int r = 0; r < 2000; ++r) hipLaunchKernelGGL(k_empty, dim3(400), dim3(192), 0, st, out); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("empty kernel, 400 workgroups, back to back: %.2f us\n", ms * 1e3f / 2000);
    for (int wgs : {40, 400, 2000}) {
        printf("%4d workgroups, dependent loads 1/2/3/4/6/8: %.2f %.2f %.2f %.2f %.2f %.2f us;  with 3 barrier pairs (depth 4/6): %.2f %.2f us\n", wgs,
               run<1, false>(wgs, next, out, n, st, 1000, false, nullptr), run<2, false>(wgs, next, out, n, st, 1000, false, nullptr), run<3, false>(wgs, next, out, n, st, 1000, false, nullptr),
               run<4, false>(wgs, next, out, n, st, 1000, false, nullptr), run<6, false>(wgs, next, out, n, st, 1000, false, nullptr), run<8, false>(wgs, next, out, n, st, 1000, false, nullptr),
               run<4, true>(wgs, next, out, n, st, 1000, false, nullptr), run<6, true>(wgs, next, out, n, st, 1000, false, nullptr));
    }
    return 0;
}

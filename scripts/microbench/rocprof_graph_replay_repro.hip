// rocprof_graph_replay_repro.hip — an attempt at a MINIMAL reproducer for the crash DESIGN.md §10 reports: `rocprofv3 --kernel-trace` (ROCm 7.2) segfaults
// inside hipGraphLaunch on libpgo's C3 timed region once the PCG's end game interleaves EAGER chunks with REPLAYS of a chunk graph that was captured (thread-local capture mode) late
// in an earlier linear system.  This program does only that, with an empty-ish kernel: per "system" a run of eager launches with a pinned-memory poll kernel + event every chunk; in
// system 1 a 24-kernel chunk is captured and instantiated AFTER 192 eager iterations; from then on full chunks are replayed as the graph and short chunks launched eagerly, the host
// waiting on the poll events one chunk behind.
//   hipcc --offload-arch=gfx950 -O2 -o rocprof_graph_replay_repro scripts/microbench/rocprof_graph_replay_repro.hip
//   ./rocprof_graph_replay_repro                                  (plain: prints "ok")
//   rocprofv3 --kernel-trace -d /tmp/tr -o t -- ./rocprof_graph_replay_repro      (the question: does the tool crash here too?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void work_kernel(double* v, int n, int k, const int* stop) {
    if (*stop) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * 0.999 + 1e-3 * k;
}
__global__ void count_kernel(int* flags, int limit) { if (threadIdx.x == 0 && blockIdx.x == 0) { flags[2] += 1; if (flags[2] >= limit) flags[0] = 1; } }
__global__ void poll_kernel(const int* flags, int* host) { if (threadIdx.x < 3) host[threadIdx.x] = flags[threadIdx.x]; }

int main(int argc, char** argv) {
    const int systems = argc > 1 ? std::atoi(argv[1]) : 8, n = 1 << 20;
    const int every = argc > 3 ? std::atoi(argv[3]) : 24;
    const long max_replays = argc > 4 ? std::atol(argv[4]) : -1;      // stop the program cleanly after this many graph replays (-1: run all systems)      // iterations per chunk: 5 kernel nodes each (24 -> a graph of 120 kernel nodes)
    const int variant = argc > 2 ? std::atoi(argv[2]) : 0;      // bit 0: capture the chunk BEFORE any eager launch; bit 1: after the capture every chunk is a full one (no eager chunk between replays);
                                                                 // bit 2: no kernel writes pinned host memory (the poll is a plain event); bit 3: global capture mode; bit 4: no stop-flag argument reads
    hipStream_t st; CHK(hipStreamCreate(&st));
    double* v; int* flags; int* host; hipEvent_t ev[2];
    CHK(hipMalloc(&v, n * sizeof(double))); CHK(hipMemset(v, 0, n * sizeof(double)));
    CHK(hipMalloc(&flags, 4 * sizeof(int)));
    CHK(hipHostMalloc(&host, 2 * 4 * sizeof(int)));
    CHK(hipEventCreate(&ev[0])); CHK(hipEventCreate(&ev[1]));
    hipGraphExec_t exec = nullptr;
    auto iteration = [&](int k) {      // "matvec", "update" (counts the iteration and raises the stop flag at the limit), three "level kernels"
        hipLaunchKernelGGL(work_kernel, dim3(n / 256), dim3(256), 0, st, v, n, k, (const int*)flags);
        hipLaunchKernelGGL(count_kernel, dim3(1), dim3(64), 0, st, flags, 100000000);
        for (int l = 0; l < 3; ++l) hipLaunchKernelGGL(work_kernel, dim3(64 >> l), dim3(256), 0, st, v, n >> (4 + l), k, (const int*)flags);
    };
    long launched_graphs = 0, eager_chunks = 0;
    for (int s = 0; s < systems; ++s) {
        CHK(hipMemsetAsync(flags, 0, 4 * sizeof(int), st));
        const int total = 300 + 37 * s;      // iterations this "PCG" needs
        int k = 0, chunks = 0;
        while (k < total) {
            int chunk = every;
            if (!(variant & 2) && total - k < 2 * every) chunk = ((total - k) / 2 + 4) & ~1;      // end game: short eager chunks
            if (chunk > total - k) chunk = total - k;
            if (!exec && ((variant & 1) ? true : (s >= 1 && k >= 192)) && (k & 1) == 0) {                  // capture once a PCG has run 192 iterations eagerly
                hipGraph_t g = nullptr;
                CHK(hipStreamBeginCapture(st, (variant & 8) ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
                for (int j = 0; j < every; ++j) iteration(2 + j);
                CHK(hipStreamEndCapture(st, &g));
                CHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
                CHK(hipGraphDestroy(g));
            }
            if (exec && chunk == every && (k & 1) == 0) { CHK(hipGraphLaunch(exec, st)); ++launched_graphs; if (max_replays >= 0 && launched_graphs >= max_replays) { CHK(hipStreamSynchronize(st)); std::printf("ok: stopped after %ld graph replays of %d kernel nodes\n", launched_graphs, 5 * every); std::fflush(stdout); CHK(hipGraphExecDestroy(exec)); return 0; } }
            else { for (int j = 0; j < chunk; ++j) iteration(k + j); ++eager_chunks; }
            k += chunk;
            if (!(variant & 4)) hipLaunchKernelGGL(poll_kernel, dim3(1), dim3(64), 0, st, (const int*)flags, host + 4 * (chunks & 1));
            CHK(hipEventRecord(ev[chunks & 1], st));
            if (chunks >= 1) CHK(hipEventSynchronize(ev[(chunks - 1) & 1]));   // one chunk behind
            ++chunks;
        }
        CHK(hipStreamSynchronize(st));
    }
    std::printf("ok: %d systems, %ld graph replays, %ld eager chunks, v[0] = %g\n", systems, launched_graphs, eager_chunks, 0.0);
    std::fflush(stdout);
    if (exec) CHK(hipGraphExecDestroy(exec));
    return 0;
}

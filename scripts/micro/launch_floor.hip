// What a launch costs on THIS box before it does any work (profiles/r04/launch_floor.json): the duration of kernels that do nothing, or
// only write the 80-byte records of an 8 192-instance shard, as HIP events and rocprofv3 see it -- the part of the 12.6 us of the
// 8 192-instance shard and of the 4 us "empty" fp64 re-solve launch of Formulation A that no kernel change can remove.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_floor.hip -o build/launch_floor ; run: build/launch_floor
//        (under rocprofv3 --kernel-trace --stats the per-kernel averages are the profiler's view of the same launches)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
// one 16-byte store per thread: 256 threads x grid x 16 B
__global__ __launch_bounds__(256) void k_store(double2* out) { out[(size_t)blockIdx.x * 256 + threadIdx.x] = double2{1.0, 2.0}; }
// a dependent chain of `n` scalar loads: what a launch-bound kernel with one latency-bound wavefront per SIMD looks like
__global__ __launch_bounds__(256) void k_chain(const int* __restrict__ next, int n, int* out)
{
    int i = (blockIdx.x * 256 + threadIdx.x) & 1023;
    for (int k = 0; k < n; ++k) i = next[i];
    if (i == -1) *out = i;
}

template <typename F> static void measure(const char* name, int grid, F launch, bool last)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 20; ++w) launch();
    hipDeviceSynchronize();
    std::vector<float> iso;
    for (int r = 0; r < 200; ++r) {                         // isolated: one launch between synchronisations
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); iso.push_back(ms);
    }
    std::sort(iso.begin(), iso.end());
    const int T = 200;                                      // a train: per-launch interval of back-to-back launches
    hipEventRecord(a); for (int r = 0; r < T; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float tr; hipEventElapsedTime(&tr, a, b);
    printf("  {\"kernel\": \"%s\", \"grid\": %d, \"isolated_us_median\": %.2f, \"isolated_us_min\": %.2f, \"train_us_per_launch\": %.2f}%s\n", name, grid,
           1e3 * iso[iso.size() / 2], 1e3 * iso[0], 1e3 * tr / T, last ? "" : ",");
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    int* flag; hipMalloc(&flag, 4);
    double2* buf; hipMalloc(&buf, (size_t)4096 * 256 * 16);
    std::vector<int> nx(1024); for (int i = 0; i < 1024; ++i) nx[i] = (i * 389 + 17) & 1023;
    int* next; hipMalloc(&next, 4096); hipMemcpy(next, nx.data(), 4096, hipMemcpyHostToDevice);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("{\"device\": \"%s\", \"cus\": %d, \"launches\": [\n", prop.name, prop.multiProcessorCount);
    measure("k_empty", 1, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, 0, flag); }, false);
    measure("k_empty", 64, [&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, 0, flag); }, false);
    measure("k_empty", 512, [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, 0, flag); }, false);
    measure("k_empty", 2048, [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, 0, flag); }, false);
    measure("k_store_655KB", 160, [&] { hipLaunchKernelGGL(k_store, dim3(160), dim3(256), 0, 0, buf); }, false);      // 8 192 x 80 B
    measure("k_store_5.2MB", 1280, [&] { hipLaunchKernelGGL(k_store, dim3(1280), dim3(256), 0, 0, buf); }, false);    // 65 536 x 80 B
    measure("k_chain_20_loads", 512, [&] { hipLaunchKernelGGL(k_chain, dim3(512), dim3(256), 0, 0, (const int*)next, 20, flag); }, true);
    printf("]}\n");
    return 0;
}

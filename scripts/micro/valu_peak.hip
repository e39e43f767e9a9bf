// Measured VALU peaks of THIS box for the three FMA forms the roofline could be priced against (profiles/r03/valu_peak.json):
//   v_fma_f64 (the fp64 kernels), v_fma_f32 (what the fp32 solve issues: the library is built -fno-slp-vectorize, no v_pk_*),
//   v_pk_fma_f32 (the packed form behind the 157.3 TFLOP/s spec figure).
// Every lane runs CH independent dependent-chains of FMAs from registers: no memory traffic, 8 wavefronts per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_peak.hip -o build/valu_peak ; run: build/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int CH = 16, UNROLL = 8;
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double b, double c)
{
    double a[CH];
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    double s = 0; for (int i = 0; i < CH; ++i) s += a[i];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float b, float c)
{
    float a[CH];
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    float s = 0; for (int i = 0; i < CH; ++i) s += a[i];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk32(float* out, int iters, float b, float c)
{
    float2v a[CH], bb = {b, b}, cc = {c, c};
    for (int i = 0; i < CH; ++i) a[i] = float2v{threadIdx.x * 1e-3f + i, 1.0f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(bb), "v"(cc));
    float s = 0; for (int i = 0; i < CH; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> double time_ms(F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < 7; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float m; hipEventElapsedTime(&m, e0, e1); ms.push_back(m); }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 4096;     // 8 workgroups x 4 wavefronts per CU = 8 per SIMD
    void* buf; hipMalloc(&buf, (size_t)blocks * 256 * 8);
    const double inst = (double)blocks * 4 /*waves*/ * (double)iters * UNROLL * CH;      // wave-instructions per launch
    const double t64 = time_ms([&] { hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(256), 0, 0, (double*)buf, iters, 0.999999, 1e-7); });
    const double t32 = time_ms([&] { hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, (float*)buf, iters, 0.999999f, 1e-7f); });
    const double tpk = time_ms([&] { hipLaunchKernelGGL(k_pk32, dim3(blocks), dim3(256), 0, 0, (float*)buf, iters, 0.999999f, 1e-7f); });
    const double simds = cus * 4.0;
    auto tf = [&](double ms, double flop_per_inst) { return inst * flop_per_inst / (ms * 1e-3) / 1e12; };
    auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / (inst / simds); };          // cycles per wave-instruction per SIMD at the 2.4 GHz spec clock
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": 8, \"wave_instructions_per_launch\": %.0f,\n", p.name, cus, inst);
    printf(" \"v_fma_f64\":    {\"ms\": %.4f, \"tflops\": %.2f, \"cycles_per_wave_instruction_at_2p4GHz\": %.3f},\n", t64, tf(t64, 128), cyc(t64));
    printf(" \"v_fma_f32\":    {\"ms\": %.4f, \"tflops\": %.2f, \"cycles_per_wave_instruction_at_2p4GHz\": %.3f},\n", t32, tf(t32, 128), cyc(t32));
    printf(" \"v_pk_fma_f32\": {\"ms\": %.4f, \"tflops\": %.2f, \"cycles_per_wave_instruction_at_2p4GHz\": %.3f}}\n", tpk, tf(tpk, 256), cyc(tpk));
    hipFree(buf);
    return 0;
}

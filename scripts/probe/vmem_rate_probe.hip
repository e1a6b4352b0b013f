// How many bytes per clock does ONE compute unit pull through its vector memory path, by load width?  The screens read their
// spectra with global_load_dwordx2 (one complex point per lane: the transform's element t + r M / 16 belongs to thread t);
// if the path is bound by instructions rather than bytes, dwordx4 loads (two adjacent points per lane, exchanged between lane
// pairs afterwards) would halve the load phases of the split-transform screens.
//   One workgroup of 1024 threads per CU (the split screens' occupancy) or 256 x 3 (the persistent kernel's), every wave
//   loading the same L2-resident 128 KiB region again and again, 16 loads in flight per wave.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/vmem_rate_probe.hip -o gpurun_out/vmem_rate_probe && gpurun_out/vmem_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int WIDTH>   // dwords per lane per load: 1, 2, 4
__global__ void rate_kernel(const char *__restrict__ base, size_t region, int iters, float *sink, long long *cycles) {
    const int tid = threadIdx.x;
    const char *p = base + (size_t)blockIdx.x % 8 * region;            // (a region per XCD-ish: all L2 hits after the first pass)
    const unsigned lane_off = (unsigned)tid * (unsigned)(4 * WIDTH);   // coalesced: consecutive lanes, consecutive addresses
    const unsigned span = (unsigned)blockDim.x * 4u * WIDTH;           // bytes one load instruction of the workgroup covers
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        f4 v4[16]; f2 v2[16]; float v1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const char *q = p + ((size_t)r * span) % region;           // scalar base, one vector offset
            if (WIDTH == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v4[r]) : "v"(lane_off), "s"(q) : "memory");
            if (WIDTH == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v2[r]) : "v"(lane_off), "s"(q) : "memory");
            if (WIDTH == 1) asm volatile("global_load_dword %0, %1, %2" : "=v"(v1[r]) : "v"(lane_off), "s"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (WIDTH == 4) { asm volatile("" : "+v"(v4[r])); acc += v4[r].x; }
            if (WIDTH == 2) { asm volatile("" : "+v"(v2[r])); acc += v2[r].x; }
            if (WIDTH == 1) { asm volatile("" : "+v"(v1[r])); acc += v1[r]; }
        }
    }
    const long long t1 = clock64();
    if (acc == 1234.5f) sink[0] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WIDTH>
void run(const char *buf, size_t region, int wg, int wgs_per_cu, int cus, float *sink, long long *cyc) {
    const int iters = 2000;
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<WIDTH>, dim3(grid), dim3(wg), 0, 0, buf, region, 50, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<WIDTH>, dim3(grid), dim3(wg), 0, 0, buf, region, iters, sink, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= grid;
    const double bytes_wg = (double)iters * 16 * wg * 4.0 * WIDTH;
    const double per_cu_per_clk = bytes_wg * wgs_per_cu / mean;   // shader-clock cycles (clock64 = s_memtime counts the 100 MHz..? see below)
    printf("dwordx%d  %4d threads x %d per CU: %.3f ms, %.2f TB/s whole chip, %.1f B per clock64 tick per CU (mean %.0f ticks)\n",
           WIDTH, wg, wgs_per_cu, ms, bytes_wg * grid / (ms * 1e-3) / 1e12, per_cu_per_clk, mean);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t region = 128 << 10;
    char *buf; float *sink; long long *cyc;
    hipMalloc(&buf, region * 8 + (1 << 20));
    hipMemset(buf, 0, region * 8 + (1 << 20));
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, sizeof(long long) * cus * 4);
    printf("%s, %d CUs, shader clock %.0f MHz; bytes per SHADER clock per CU = TB/s / CUs / clock\n", prop.gcnArchName, cus, prop.clockRate / 1e3);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int wg = cfg == 0 ? 1024 : 256, per = cfg == 0 ? 1 : 3;
        run<1>(buf, region, wg, per, cus, sink, cyc);
        run<2>(buf, region, wg, per, cus, sink, cyc);
        run<4>(buf, region, wg, per, cus, sink, cyc);
    }
    return 0;
}

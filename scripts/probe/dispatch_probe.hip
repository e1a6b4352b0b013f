// Where do the workgroups of a (128, 1, 128) grid of 512-thread, 64 KB-LDS workgroups go, and what does it cost when
// half of them leave at once?  The config-4 screen launches exactly that grid; with the lazy screen a workgroup whose
// (segment, tile) is masked returns immediately.  Patterns (active workgroups spin ~40 us of the wall clock):
//   A  every (x, z) masked with probability p            (independent)
//   B  masked iff S[x]                                    (whole segments, segment = blockIdx.x)
//   C  masked iff S[(x + z) % 128]                        (whole segments, segment rotated by the tile)
//   D  masked iff S[(x * 37 + z * 11) % 128]              (whole segments, scrambled)
//   E  masked iff S[z]                                    (whole tiles)
// Prints the launch time per pattern and, once, the XCC / SE / CU each workgroup of the first two tiles ran on.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_probe scripts/probe/dispatch_probe.hip && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(512) void work(const unsigned char *S, int pattern, float p, long long ticks, unsigned *where) {
    extern __shared__ char smem[];
    const unsigned x = blockIdx.x, z = blockIdx.z;
    bool masked;
    if (pattern == 0) {
        unsigned h = x * 2654435761u + z * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        masked = (h & 0xffffu) < (unsigned)(p * 65536.0f);
    } else if (pattern == 1) masked = S[x];
    else if (pattern == 2) masked = S[(x + z) % 128];
    else if (pattern == 3) masked = S[(x * 37 + z * 11) % 128];
    else masked = S[z];
    if (where && threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        where[(z * gridDim.x + x) * 2] = xcc;
        where[(z * gridDim.x + x) * 2 + 1] = hwid;
    }
    if (masked) return;
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

int main() {
    const int B = 128, T = 128;
    std::vector<unsigned char> S(128);
    unsigned char *dS; unsigned *dW;
    CHECK(hipMalloc(&dS, 128)); CHECK(hipMalloc(&dW, B * T * 2 * 4));
    CHECK(hipFuncSetAttribute((const void *)work, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const long long ticks = 4000;   // 40 us of the 100 MHz wall clock
    for (float p : {0.0f, 0.25f, 0.5f, 0.75f}) {
        srand(7); int n = 0;
        for (int i = 0; i < 128; ++i) { S[i] = (rand() % 1000) < p * 1000; n += S[i]; }
        CHECK(hipMemcpy(dS, S.data(), 128, hipMemcpyHostToDevice));
        printf("p = %.2f (%d of 128 masked in S):", p, n);
        for (int pat = 0; pat < 5; ++pat) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(work, dim3(B, 1, T), dim3(512), 65 * 1024, 0, dS, pat, p, ticks, (unsigned *)nullptr);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
            }
            printf("  %c %.3f ms", "ABCDE"[pat], best);
        }
        printf("\n");
    }
    hipLaunchKernelGGL(work, dim3(B, 1, T), dim3(512), 65 * 1024, 0, dS, 0, 0.0f, ticks, dW);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> W(B * T * 2);
    CHECK(hipMemcpy(W.data(), dW, W.size() * 4, hipMemcpyDeviceToHost));
    for (int z = 0; z < 3; ++z) {
        printf("tile %d: (xcc, se, cu) of workgroups x = 0 .. 39:\n  ", z);
        for (int x = 0; x < 40; ++x) {
            const unsigned xcc = W[(z * B + x) * 2] & 0xf, hw = W[(z * B + x) * 2 + 1];
            printf("(%u,%u,%u) ", xcc, (hw >> 13) & 7, (hw >> 8) & 15);
        }
        printf("\n");
    }
    // how many distinct x a CU serves over all tiles (is a compute unit tied to a block index?)
    int tied = 0, total = 0;
    for (int x = 0; x < B; ++x) {
        unsigned first = (W[x * 2] & 0xf) << 16 | (W[x * 2 + 1] & 0xff00);
        int same = 0;
        for (int z = 0; z < T; ++z) same += (((W[(z * B + x) * 2] & 0xf) << 16 | (W[(z * B + x) * 2 + 1] & 0xff00)) == first);
        tied += same; total += T;
    }
    printf("workgroups (x, z) that ran on the same (xcc, se, cu) as (x, 0): %d of %d\n", tied, total);
    int same_xcc = 0;
    for (int x = 0; x < B; ++x) for (int z = 0; z < T; ++z) same_xcc += (W[(z * B + x) * 2] & 0xf) == (W[x * 2] & 0xf);
    printf("workgroups (x, z) on the same XCC as (x, 0): %d of %d\n", same_xcc, B * T);
    return 0;
}

// How long does a hand-off INSIDE a launch take -- 16 KB written through (agent-scope relaxed atomic stores = sc1), drained, a flag
// stored; the consumer polls the flag and reads the 16 KB with sc1 loads -- when producer and consumer sit on the SAME XCD
// (workgroups i and i + 8 of a 1-D grid: the XCD is the linear index mod 8) and when they sit on different ones (i and i + 1)?
// Two workgroups of 256 threads play ping-pong ROUNDS times; the others leave at once.  Also: the flag alone (no payload).
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int WORDS = 2048;   // 2048 x 8 bytes = 16 KB
constexpr int ROUNDS = 2000;
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(256) void pingpong(unsigned long long *buf /* 2 x WORDS */, int *flags /* 2 x 32 */, int wa, int wb, int payload,
                                                unsigned long long *out) {
    const int me = blockIdx.x == wa ? 0 : blockIdx.x == wb ? 1 : -1;
    if (me < 0) return;
    const int tid = threadIdx.x;
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= ROUNDS; ++r) {
        if (me == 1) {   // wait for A's message r
            if (tid == 0) for (int spin = 0; ld_agent(&flags[0]) < r && spin < 20000000; ++spin) __builtin_amdgcn_s_sleep(1);   // (bounded: ~1 s)
            __syncthreads();
            if (payload)
                for (int j = tid; j < WORDS; j += 256) acc += __hip_atomic_load(&buf[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // send: my message r (A first, B in reply)
        unsigned long long *mine = buf + me * WORDS;
        if (payload)
            for (int j = tid; j < WORDS; j += 256) __hip_atomic_store(&mine[j], (unsigned long long)(r + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_agent(&flags[me * 32], r);
        if (me == 0) {   // wait for B's reply r
            if (tid == 0) for (int spin = 0; ld_agent(&flags[32]) < r && spin < 20000000; ++spin) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            if (payload)
                for (int j = tid; j < WORDS; j += 256) acc += __hip_atomic_load(&buf[WORDS + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && me == 0) { out[0] = t1 - t0; }
    if (acc == 0x1234567ull) out[1] = acc;   // (keeps the loads)
}
int main() {
    unsigned long long *buf, *out; int *flags;
    hipMalloc(&buf, 2 * WORDS * 8); hipMalloc(&flags, 64 * 4); hipMalloc(&out, 16);
    for (int payload = 0; payload < 2; ++payload)
        for (int pair = 0; pair < 4; ++pair) {
            const int wa = 0, wb = pair == 0 ? 8 : pair == 1 ? 1 : pair == 2 ? 16 : 4;
            unsigned long long best = ~0ull, t;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(flags, 0, 64 * 4); hipMemset(buf, 0, 2 * WORDS * 8);
                hipLaunchKernelGGL(pingpong, dim3(32), dim3(256), 0, 0, buf, flags, wa, wb, payload, out);
                hipDeviceSynchronize();
                hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
                if (t < best) best = t;
            }
            printf("%s, workgroups %d and %2d (%s XCD): %.2f us per one-way hand-off\n", payload ? "16 KB + flag" : "flag only   ", wa, wb,
                   (wb - wa) % 8 == 0 ? "same" : "another", best * 0.01 / ROUNDS / 2);
        }
    return 0;
}

// What does one tap of an exact correlation chain cost, by how the atom sample reaches the multiply-add?  One wavefront,
// L dependent steps, atom samples and window in LDS; wall time by s_memrealtime (100 MHz) around the loop, 20 runs.
//   A  v_fmac with the atom sample BROADCAST from LDS (every lane reads the same address; ds_read_b128 = 4 taps), the
//      window by ds_read2_b32 (2 taps)                                                   -- 1 VALU instruction per tap
//   B  as A, TWO chains (two atoms) interleaved in one wavefront                         -- 2 independent VALU per tap
//   C  v_readlane + v_fmac, the atom samples one per lane in a register (round 2's plain chain)
//   D  v_mfma_f32_16x16x4_f32, operands by ds_read_b32 (the select's matrix-core chain: 16 atoms x 16 lags, 4 taps)
//   E  v_pk_fma_f32: two atoms as the halves of one packed accumulator, both samples by ONE broadcast ds_read_b64 of an
//      interleaved row
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/chain_latency_probe.hip -o gpurun_out/chain_latency_probe && gpurun_out/chain_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
constexpr int L = 4096;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long cyc() { return __builtin_amdgcn_s_memtime(); }   // shader clock

__global__ void kA(const float *d, const float *win, float *out, unsigned long long *ticks) {
    __shared__ __attribute__((aligned(16))) float sd[L], sw[L + 128];
    const int lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < L; j += blockDim.x) sd[j] = d[j];
    for (int j = threadIdx.x; j < L + 128; j += blockDim.x) sw[j] = win[j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    float acc = 0.f;
    const float *r = sw + lane;
    for (int k = 0; k < L; k += 16) {
        f32x4 a[4];
        float w[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4 *>(sd + k + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = r[k + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(w[u], a[u / 4][u % 4], acc);
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) out[lane] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
__global__ void kB(const float *d, const float *win, float *out, unsigned long long *ticks) {
    __shared__ __attribute__((aligned(16))) float sd[2 * L], sw[L + 128];
    const int lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < 2 * L; j += blockDim.x) sd[j] = d[j];
    for (int j = threadIdx.x; j < L + 128; j += blockDim.x) sw[j] = win[j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    float acc0 = 0.f, acc1 = 0.f;
    const float *r = sw + lane;
    for (int k = 0; k < L; k += 16) {
        f32x4 a[4], b[4];
        float w[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = *reinterpret_cast<const f32x4 *>(sd + k + 4 * u); b[u] = *reinterpret_cast<const f32x4 *>(sd + L + k + 4 * u); }
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = r[k + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc0 = __builtin_fmaf(w[u], a[u / 4][u % 4], acc0); acc1 = __builtin_fmaf(w[u], b[u / 4][u % 4], acc1); }
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) { out[lane] = acc0; out[64 + lane] = acc1; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
__global__ void kC(const float *d, const float *win, float *out, unsigned long long *ticks) {
    __shared__ float sw[L + 128];
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < L + 128; j += 64) sw[j] = win[j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    float acc = 0.f;
    const float *r = sw + lane;
    for (int k = 0; k < L; k += 64) {
        const int dv = __float_as_int(d[k + lane]);
#pragma unroll
        for (int q = 0; q < 64; ++q) acc = __builtin_fmaf(r[k + q], __int_as_float(__builtin_amdgcn_readlane(dv, q)), acc);
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) out[lane] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
__global__ void kD(const float *d /* [16][L] */, const float *win, float *out, unsigned long long *ticks) {
    __shared__ float sw[L + 128];
    __shared__ float sa[2 * (L + 2)];   // two distinct rows stand for the 16
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < L + 128; j += 64) sw[j] = win[j];
    for (int i = 0; i < 2; ++i) for (int j = lane; j < L; j += 64) sa[i * (L + 2) + j] = d[i * L + j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *ap = sa + (lane & 1) * (L + 2) + (lane >> 4);
    const float *bp = sw + (lane & 15) + (lane >> 4);
    for (int k = 0; k < L; k += 32) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = ap[k + 4 * u]; b[u] = bp[k + 4 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) out[lane] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
__global__ void kE(const float *d, const float *win, float *out, unsigned long long *ticks) {
    __shared__ __attribute__((aligned(16))) float sd[2 * L], sw[L + 128];   // sd: (d0[k], d1[k]) interleaved
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < L; j += 64) { sd[2 * j] = d[j]; sd[2 * j + 1] = d[L + j]; }
    for (int j = lane; j < L + 128; j += 64) sw[j] = win[j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    f32x2 acc = {0.f, 0.f};
    const float *r = sw + lane;
    for (int k = 0; k < L; k += 16) {
        f32x4 a[8];
        float w[16];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4 *>(sd + 2 * k + 4 * u);   // taps k + 2u, k + 2u + 1 of both atoms
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = r[k + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const f32x2 dd = {a[u / 2][2 * (u % 2)], a[u / 2][2 * (u % 2) + 1]};
            const f32x2 ww = {w[u], w[u]};
            acc = __builtin_elementwise_fma(ww, dd, acc);
        }
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) { out[lane] = acc[0]; out[64 + lane] = acc[1]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
// F: as D, the atom samples straight from GLOBAL memory (8 rows of a dictionary too large for the L2s, each lane 4 bytes per
// instruction), a ring of RING requests in flight; the window from LDS.  No staging, no barrier in the loop.
template <int RING>
__global__ void kF(const float *d /* rows of L floats, 8 of them used (row i % 8) */, const float *win, float *out, unsigned long long *ticks, int rowstride) {
    __shared__ float sw[L + 128];
    const int lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < L + 128; j += blockDim.x) sw[j] = win[j];
    __syncthreads();
    const unsigned long long t0 = now(), c0 = cyc();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *ap = d + (size_t)(lane & 7) * rowstride + (lane >> 4);
    const float *bp = sw + (lane & 15) + (lane >> 4);
    float ring[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) ring[u] = ap[4 * u];
    for (int k = 0; k < L; k += 4 * RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const float a = ring[u];
            const int kn = k + 4 * (RING + u);
            ring[u] = ap[kn < L ? kn : 0];
            const float b = bp[k + 4 * u];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    }
    const unsigned long long c1 = cyc(), t1 = now();
    if (threadIdx.x < 64 && blockIdx.x == 0) out[lane] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
int main(int argc, char **argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 1, NT = argc > 2 ? atoi(argv[2]) : 64;
    printf("%d blocks of %d threads, every wavefront the same chain\n", NB, NT);
    std::vector<float> d(2 * L), w(L + 128), ref0(64), ref1(64), o(128);
    srand(7);
    for (auto &v : d) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : w) v = rand() / (float)RAND_MAX - 0.5f;
    for (int l = 0; l < 64; ++l) {
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < L; ++k) { a0 = fmaf(w[l + k], d[k], a0); a1 = fmaf(w[l + k], d[L + k], a1); }
        ref0[l] = a0; ref1[l] = a1;
    }
    float *dd, *dw, *dout; unsigned long long *dt;
    hipMalloc(&dd, d.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dout, 512); hipMalloc(&dt, 16);
    hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    // (kF: a dictionary of 2048 rows x L floats = 32 MB, the 8 rows used 256 rows apart: nothing of it in the L2s at first touch)
    float *big; hipMalloc(&big, (size_t)2048 * L * 4); hipMemset(big, 0, (size_t)2048 * L * 4);
    for (int i = 0; i < 8; ++i) hipMemcpy(big + (size_t)i * 256 * L, d.data() + (i & 1) * L, L * 4, hipMemcpyHostToDevice);
    struct V { const char *name; int which; };
    for (V v : {V{"A fmac, LDS-broadcast sample       ", 0}, V{"B two fmac chains interleaved      ", 1}, V{"C readlane + fmac                  ", 2},
                V{"D mfma 16x16x4                     ", 3}, V{"E pk_fma, two atoms, one b64 sample", 4},
                V{"F mfma, atoms from global, ring 16 ", 5}, V{"F mfma, atoms from global, ring 32 ", 6}}) {
        unsigned long long best = ~0ull, t[2], bc = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(dout, 0, 512);
            switch (v.which) {
                case 0: hipLaunchKernelGGL(kA, dim3(NB), dim3(NT), 0, 0, dd, dw, dout, dt); break;
                case 1: hipLaunchKernelGGL(kB, dim3(NB), dim3(NT), 0, 0, dd, dw, dout, dt); break;
                case 2: hipLaunchKernelGGL(kC, dim3(NB), dim3(NT), 0, 0, dd, dw, dout, dt); break;
                case 3: hipLaunchKernelGGL(kD, dim3(NB), dim3(NT), 0, 0, dd, dw, dout, dt); break;
                case 4: hipLaunchKernelGGL(kE, dim3(NB), dim3(NT), 0, 0, dd, dw, dout, dt); break;
                case 5: hipLaunchKernelGGL(kF<16>, dim3(NB), dim3(NT), 0, 0, big, dw, dout, dt, 256 * L); break;
                case 6: hipLaunchKernelGGL(kF<32>, dim3(NB), dim3(NT), 0, 0, big, dw, dout, dt, 256 * L); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
            if (t[0] < best) { best = t[0]; bc = t[1]; }
        }
        hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        if (v.which < 3 || v.which == 4) {
            for (int l = 0; l < 64; ++l) bad += o[l] != ref0[l];
            if (v.which == 1 || v.which == 4) for (int l = 0; l < 64; ++l) bad += o[64 + l] != ref1[l];
        }
        printf("%s: %6.1f us for %d taps = %5.2f ns = %5.1f shader cycles per tap (clock %.2f GHz); lanes differing from the fmaf chain: %d%s\n",
               v.name, best * 0.01, L, best * 10.0 / L, (double)bc / L, bc / (best * 10.0), bad, (v.which == 3 || v.which > 4) ? " (not compared)" : "");
    }
    return 0;
}

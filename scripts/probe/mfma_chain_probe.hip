// Is a chain of v_mfma_f32_4x4x1_16b_f32 (K = 1: one product per accumulation) bit-identical to the fmaf chain of the
// oracle, and which lane holds what?  16 blocks of 4 rows x 4 columns: here rows = 4 atoms, columns/blocks = 64 lags.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/mfma_chain_probe.hip -o gpurun_out/mfma_chain_probe && gpurun_out/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int L = 512;
__global__ void chain_mfma(const float *d /* [4][L] */, const float *win /* [64 + L] */, float *out /* [4][64] */) {
    const int lane = threadIdx.x;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < L; ++j) {
        const float a = d[(lane & 3) * L + j];     // A[block][row i = lane % 4]: atom i, the same in every block
        const float b = win[lane + j];             // B[block = lane / 4][col = lane % 4]: lag 4 * block + col = lane
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = acc[i];   // D[block][row i = register][col]: atom i at lag = lane
}
// v_mfma_f32_16x16x4_f32 (K = 4 per instruction): rows = 16 atoms, columns = 16 lags, lane l: A[i = l % 16][k = l / 16],
// B[k = l / 16][j = l % 16], D register r: [i = 4 (l / 16) + r][j = l % 16].  Exact only if the four products are accumulated
// one after the other, fused, in the order k = 0 .. 3.
__global__ void chain_mfma16(const float *d /* [16][L] */, const float *win, float *out /* [16][16] */) {
    const int lane = threadIdx.x;
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < L; j += 4) {
        const float a = d[(lane & 15) * L + j + (lane >> 4)];
        const float b = win[(lane & 15) + j + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}
__global__ void chain_fma16(const float *d, const float *win, float *out) {
    const int lane = threadIdx.x;   // 64 threads: lag = lane % 16, atoms lane / 16 + 4 r
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) + 4 * r;
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(win[(lane & 15) + j], d[i * L + j], acc);
        out[i * 16 + (lane & 15)] = acc;
    }
}
__global__ void chain_fma(const float *d, const float *win, float *out) {
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(win[lane + j], d[i * L + j], acc);
        out[i * 64 + lane] = acc;
    }
}
int main() {
    std::vector<float> d(4 * L), w(64 + L), o1(256), o2(256);
    float *dd, *dw, *do1, *do2;
    hipMalloc(&dd, d.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&do1, 1024); hipMalloc(&do2, 1024);
    int bad_total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        srand(trial + 1);
        const float scale = trial % 4 == 3 ? 1e-20f : (trial % 4 == 2 ? 1e18f : 1.0f);
        for (auto &v : d) v = (rand() / (float)RAND_MAX - (trial % 3 == 0 ? 0.0f : 0.5f)) * (rand() % 17 == 0 ? 0.0f : 1.0f);
        for (auto &v : w) v = (rand() / (float)RAND_MAX - 0.5f) * scale * (rand() % 13 == 0 ? 0.0f : 1.0f);
        if (trial % 5 == 4) for (int j = L - 37; j < L; ++j) for (int i = 0; i < 4; ++i) d[i * L + j] = 0.0f;   // zero tail
        hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        chain_mfma<<<1, 64>>>(dd, dw, do1); chain_fma<<<1, 64>>>(dd, dw, do2);
        hipMemcpy(o1.data(), do1, 1024, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), do2, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += memcmp(&o1[i], &o2[i], 4) != 0;
        if (bad && bad_total < 5) printf("trial %d (scale %g): %d of 256 differ, e.g. %a vs %a\n", trial, scale, bad, o1[0], o2[0]);
        bad_total += bad;
    }
    printf("mfma 4x4x1 chain vs fmaf chain: %d values differ over 200 trials\n", bad_total);
    {
        std::vector<float> d16(16 * L), p1(256), p2(256);
        float *dd16; hipMalloc(&dd16, d16.size() * 4);
        int bad16 = 0;
        for (int trial = 0; trial < 200; ++trial) {
            srand(1000 + trial);
            const float scale = trial % 4 == 3 ? 1e-36f : (trial % 4 == 2 ? 1e18f : 1.0f);   // (1e-36: denormal products)
            for (auto &v : d16) v = (rand() / (float)RAND_MAX - (trial % 3 == 0 ? 0.0f : 0.5f)) * (rand() % 17 == 0 ? 0.0f : 1.0f);
            for (auto &v : w) v = (rand() / (float)RAND_MAX - 0.5f) * scale * (rand() % 13 == 0 ? 0.0f : 1.0f);
            hipMemcpy(dd16, d16.data(), d16.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
            chain_mfma16<<<1, 64>>>(dd16, dw, do1); chain_fma16<<<1, 64>>>(dd16, dw, do2);
            hipMemcpy(p1.data(), do1, 1024, hipMemcpyDeviceToHost); hipMemcpy(p2.data(), do2, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < 256; ++i) bad += memcmp(&p1[i], &p2[i], 4) != 0;
            if (bad && bad16 < 5 * 256) printf("16x16x4 trial %d (scale %g): %d of 256 differ, e.g. %a vs %a\n", trial, scale, bad, p1[0], p2[0]);
            bad16 += bad;
        }
        printf("mfma 16x16x4 chain vs fmaf chain: %d values differ over 200 trials\n", bad16);
        bad_total += bad16;
    }
    return bad_total != 0;
}

// mpcore.hip -- greedy matching-pursuit encoder for MI355X (gfx950 / CDNA4).  C ABI: include/mpcore.h
//
// Path replaced: the loop body of sparse_code, /root/reference/modules/matchingpursuit.py:269-328.
//
// Design:
//   * The A x N feature map is never materialised.  The (atom, lag) plane of every segment is cut
//     into cells of TA atoms x 64 lags; a correlate kernel computes one cell per wavefront on the
//     fp32 matrix cores (v_mfma_f32_32x32x2_f32, which is bit for bit an ascending-k fmaf chain) and
//     keeps only the cell's signed maximum as a 64-bit key (ordered value | inverted flat index).
//   * A select kernel reduces a segment's keys (= torch.max's first-occurrence argmax), records the
//     event, subtracts gain * atom from the residual, and marks the cells the subtraction touched.
//   * MP_PATH_DIRECT recomputes every cell each iteration (what the reference does);
//     MP_PATH_INCREMENTAL recomputes only the marked cells.  A cell's value depends only on the
//     residual samples under it, so both select identical events, bit for bit.
//   * The dictionary tile of a workgroup is staged once into LDS (up to 128 KiB of the 160 KiB),
//     pre-swizzled by a prep kernel so that MFMA B-operands for four k-steps come from one
//     conflict-free ds_read_b128; the residual window is read as a Toeplitz A-operand straight
//     from LDS with ds_read_b32 at immediate offsets.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <atomic>
#include <mutex>
#include <vector>

#include "mpcore.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return fail(MP_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Optional per-kernel timing (bench.py): hipEvents recorded on the launch stream around the
// correlate / select launches of mp_encode_f32.  Off by default; never used while capturing a graph.
// An event between two kernels costs ~10 us of idle GPU (measured: 7.5 vs 6.0 ms per 64-iteration
// encode with three spans per iteration), so bench.py samples every n-th iteration instead of all.
// ------------------------------------------------------------------------------------------------
enum { PROF_CORR_FULL = 0, PROF_CORR_INC = 1, PROF_SELECT = 2, PROF_KINDS = 3 };
struct ProfSpan { hipEvent_t a, b; int kind; };
// Process-wide span list behind a mutex; whether the CURRENT iteration is sampled is per encoding thread (two host
// threads may encode at once), and a thread closes only the span it opened.
std::mutex g_prof_mu;
struct Profiler {
    std::atomic<int> every{0};   // 0 = off, 1 = every iteration, n = iterations k with k % n == 0
    std::atomic<int> skip{0};    // bit q: no spans of kind q (an event between two kernels idles the GPU ~8 us: a timed region
                                 // that only wants its dominant kernel's durations leaves the select spans out)
    std::vector<ProfSpan> spans; // guarded by g_prof_mu
    std::vector<hipEvent_t> pool;
    static bool &armed() { static thread_local bool on = false; return on; }
    static hipEvent_t &open_end() { static thread_local hipEvent_t e = nullptr; return e; }
    void arm(int k) { const int n = every.load(std::memory_order_relaxed); armed() = n > 0 && k % n == 0; }
    hipEvent_t get() {  // caller holds g_prof_mu
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    void begin(int kind, hipStream_t st) {
        open_end() = nullptr;
        if (!armed() || ((skip.load(std::memory_order_relaxed) >> kind) & 1)) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfSpan s{get(), get(), kind};
        if (!s.a || !s.b) return;
        (void)hipEventRecord(s.a, st);
        spans.push_back(s);
        open_end() = s.b;
    }
    void end(hipStream_t st) {
        if (!armed() || !open_end()) return;
        (void)hipEventRecord(open_end(), st);
        open_end() = nullptr;
    }
};
Profiler g_prof;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int LAGS_PER_WAVE = 64;   // one cell = TA atoms x 64 lags, owned by one wavefront
constexpr int WAVES = 4;            // wavefronts per workgroup
constexpr int KC_MAX = 512;         // atom samples staged in LDS per chunk
constexpr int MAXCONT = 32;         // FFT path: inexact contender cells per segment per iteration
constexpr int SPLIT_REFINE_ROUNDS = 4;  // ... times this many rounds where transforms are split (fft_iteration)
constexpr int MP_FLAG_INTERNAL_ONE_STREAM = 1 << 30;  // set by encode_impl: the batch is not split
constexpr int MP_FLAG_INTERNAL_COHERENCE = 1 << 29;   // set by mp_coherence_f32: |.| screen only
// ------------------------------------------------------------------------------------------------
// THE FORM TABLE: every threshold by which MP_PATH_FFT picks its form for a shape and a load, in one place (mp_form_table
// hands it to the host side: mpcore/_native.py::lazy_pays reads its numbers from here).  Fitted on one MI355X
// (scripts/form_sweep.py, small_batches.py, small_batch_forms.py, persist_percu.py, fine_ab.py; the measurements are quoted
// where each entry is used); tests/test_gpu_parity.py::test_default_form_is_within_reach_of_the_best_forced_form holds
// the default to the best forced form on six shapes either side of these lines.  Results never depend on the form.
// ------------------------------------------------------------------------------------------------
struct FormTable {
    // cells per segment (64-lag blocks x 32-atom tiles): which select runs between two screens
    int64_t quarter_max_cells = 16384;    // up to here: the quarter-cell select (sub-cell maxima kept)
    int64_t fused_min_cells = 65536;      // from here: the whole-cell one-kernel select with block summaries
    int64_t persist_max_cells = 65536;    // the one-launch form keeps its quarter maxima up to here (16 B per cell)
    // one launch (persistent) or launch per step, by load = transform points per step = B x ceil(A / 2) x M
    double persist_points_table = 140e6;       // with the coherence table: one launch up to here ... (round 4, after the one-launch
                                               //     form gained most from the round's kernel work -- scripts/form_sweep.py, one launch /
                                               //     per step at 134 M points: 512 x 512 1283 / 1287, 1024 x 512 766 / 673, 256 x 1024
                                               //     1528 / 1328, 1024 x 1024 385 / 389; at 268 M: 777 / 778, 385 / 518.  Round 3: 96 M.)
    double persist_points_table_1024 = 100e6;  // ... 1024-point transforms: up to here AND
    double persist_points_any_batch_1024 = 40e6;  // ... (small loads: one launch at any batch size up to here; 512 x 256, 128 segments =
                                               //     34 M: 1487 / 1418, 256 segments = 67 M: 1477 / 1722.  Round 3: 20 M.)
    int persist_segments_table_1024 = 64;      // ... at most this many segments (form_sweep, one launch / per step, k seg-it/s: 512 x 256
                                               //     64 segments 1296 / 981, 128: 1310 / 1431; 2048 x 256: 64: 456 / 434, 128: 448 / 481;
                                               //     4096 x 256: 16: 177 / 157, 64 = 134 M points: 200 / 238)
    int sub_batch_min_segments = 48;           // below this many segments the launch-per-step side has one stream: one launch wins
    double persist_spectra_bytes = 8.5e6;      // without the table: pair spectra that fit the L2s (A / 2 x M x 8 B) ...
    double persist_points_fit = 200e6;         // ... keep the one-launch form up to here,
    double persist_points_nofit = 40e6;        // larger dictionaries up to here
    // short segments (N <= short_ratio x L: an event dirties half of the lags or more): launch per step, quarter select
    int short_ratio = 4;
    int short_logm_always = 12;                // at 4096-point transforms at every batch size,
    int short_logm_small = 11;                 // at 2048-point transforms
    int short_small_segments = 8;              // ... up to this many segments
    // sub-batches on forked streams (launch-per-step forms)
    int sub_batches = 4;
    // inside the one-launch form
    int persist_two_per_cu_load = 4;           // tasks per CU in flight up to which two workgroups per CU beat three
    int persist_fine_num = 11, persist_fine_den = 2;  // finer tasks while 2 x (their number) <= 11 x CUs (~1400 in flight)
    int persist_fine_max_logm = 10;            // ... at 1024-point transforms only: at 2048 points they lose from 2 segments up since
                                               //     round 4 (scripts/fine_sweep.py, 512 x 512, one / two slots per quarter: 4 segments 1.296
                                               //     / 1.409 ms, 8: 1.387 / 1.477, 16: 1.566 / 1.639, 32: 1.899 / 1.945; 1024 x 512, 16: 2.05
                                               //     / 2.27; 256 x 256: 1: 1.039 / 0.993, 8: 1.116 / 1.103, 64: 2.040 / 1.770)
    int persist_select_workers = 64;           // select workers = min(segments, this) (scripts/persist_sweep.py at the lazy margin 0.9: 48 / 56 / 64 / 80 at 64 segments 2.766 / 2.626 / 2.608 / 2.604 ms, at 128: 4.803 / 4.505 / 4.443 / 4.447; round 3: 48)
    // when the lazy screen is worth its table (host side: _native.lazy_pays)
    int lazy_min_steps = 8, lazy_min_tiles = 4, lazy_always_tiles = 32, lazy_batch_tiles = 384;
    // the lazy screen's margin (a tile is skipped while its widened upper bounds stay below margin x the best clean lower
    // bound; any value is exact): inside the persistent launch 0.9 -- 0.7 at 1024-point transforms -- (scripts/
    // lazy_knob_sweep.py, 0.7 / 0.85 / 0.9 / 0.95, 3 K planted events: 512 x 512, 64 segments 2.742 / 2.605 / 2.574 / 2.554 ms,
    // 128 segments 5.05 / 4.54 / 4.45 / 4.40, K = 128: 5.79 / 5.62 / 5.56 / 5.49; 2048 x 512: 4.56 / 4.12 / 4.02 / 3.91;
    // 256 x 256: 1.753 / 1.782 / 1.791 / 1.796; K / 2 planted and unplanted signals: flat up to 0.95, +16 % at 1.0),
    // between launches 0.95 (round 3: 0.85; scripts/c3_margin.py, 0.85 / 0.9 / 0.95 / 1.0: BASELINE configs[3] 344 / 327 / 313 / 301 ms
    // = 95.4 / 100.2 / 104.7 / 108.9 k, without planted events 249 / - / 246 / 253 at 32 segments; 1024 x 1024, 128 segments: 16.3 /
    // 14.8 / 14.0 / 13.6 ms, with K / 2 planted events 25.6 / 24.4 / 24.3 / 24.2; 512 x 512, 256 segments, K / 2 planted: 14.9 / 13.8 /
    // 12.9 / 13.0)
    float lazy_margin_persistent = 0.9f, lazy_margin_persistent_1024 = 0.7f, lazy_margin_steps = 0.95f;
};
constexpr FormTable FORM{};
constexpr int64_t QUARTER_MAX_CELLS = FORM.quarter_max_cells;
constexpr int64_t FUSED_MIN_CELLS = FORM.fused_min_cells;
constexpr int64_t PERSIST_MAX_CELLS = FORM.persist_max_cells;

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// Geometry shared by host and device
// ------------------------------------------------------------------------------------------------
struct Geom {
    int64_t B, N, A, L;
    int TA;        // atoms per cell
    int NAT;       // atom tiles  = ceil(A / TA)
    int NBLK;      // lag blocks  = ceil(N / 64)
    int KC;        // k chunk (multiple of 16, <= KC_MAX)
    int NCH;       // k chunks
    int Lpp;       // NCH * KC  (dictionary k extent incl. zero padding)
    int64_t Ns;    // residual row stride in floats (zero padded past N)
    int MAXC;      // most 64-lag blocks one subtraction can dirty
    int circ;      // FFT screen: 0 = by atom length (make_fft_geom); else log2 of a CIRCULAR transform of exactly N = 2^circ
                   // points, every lag valid (the coherence table's pass: mp_coherence_f32)
};

Geom make_geom(int64_t B, int64_t N, int64_t A, int64_t L, int TA, int64_t window = 0) {
    Geom g;
    g.B = B; g.N = N; g.A = A; g.L = L; g.TA = TA;
    g.NAT = (int)((A + TA - 1) / TA);
    g.NBLK = (int)((N + LAGS_PER_WAVE - 1) / LAGS_PER_WAVE);
    int Lp = (int)round_up(L, 16);  // k8 groups are consumed in pairs
    g.KC = Lp < KC_MAX ? Lp : KC_MAX;
    g.NCH = (Lp + g.KC - 1) / g.KC;
    g.Lpp = g.NCH * g.KC;
    // rows are readable (as zeros) far enough past N for the longest window any kernel loads
    const int64_t reach = g.Lpp > window ? g.Lpp : window;
    g.Ns = round_up((int64_t)g.NBLK * LAGS_PER_WAVE + reach + 64, 64);
    // a subtraction at lag p changes lags [p-L+1, p+L-1]: at most this many 64-lag blocks
    int nb = (int)((2 * L - 2) / LAGS_PER_WAVE) + 2;
    if (nb > g.NBLK) nb = g.NBLK;
    g.MAXC = nb;
    g.circ = 0;
    return g;
}

// ------------------------------------------------------------------------------------------------
// Keys: max over keys == torch.max (signed value, first flat index on ties)
// ------------------------------------------------------------------------------------------------
__device__ inline unsigned ord_f32(float f) {
    unsigned u = __float_as_uint(f + 0.0f);  // -0.0 -> +0.0 (torch compares them equal)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float unord_f32(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ inline u64 make_key(float v, unsigned flat) {
    return ((u64)ord_f32(v) << 32) | (u64)(0xffffffffu - flat);
}

__device__ inline u64 wave_max_u64(u64 k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)k, off, 64);
        unsigned hi = __shfl_xor((unsigned)(k >> 32), off, 64);
        u64 o = ((u64)hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}

// ------------------------------------------------------------------------------------------------
// unit_norm  (modules/normalization.py:4-6): one wavefront per atom, sum of squares in fp64 in a fixed order
// (64 interleaved partial sums, then their sum) that the CPU oracle reproduces bit for bit.
// ------------------------------------------------------------------------------------------------
// Correctly rounded fp32 square root, whatever the compiler makes of sqrtf / __fsqrt_rn (the latter is
// __ocml_native_sqrt_f32 in this toolchain, and whether it got the 1-ulp v_sqrt_f32 alone or its refinement depended
// on the surrounding code): fp64 sqrt is correctly rounded and 53 >= 2 * 24 + 2 bits make the second rounding exact.
__device__ __forceinline__ float sqrt_rn_f32(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float div_rn_f32(float a, float b) { return (float)((double)a / (double)b); }  // likewise

// sum of squares of a row in fp64 by one wavefront: lane j sums samples j, j + 64, ... in ascending order, then
// every lane adds the 64 partials in ascending order (the oracle's order, oracle/mp_oracle.c::mpo_unit_norm)
__device__ __forceinline__ double row_sum_squares(const float *row, int64_t L, int lane) {
    double p = 0.0;
    for (int64_t k = lane; k < L; k += 64) {
        const double x = (double)row[k];
        p += x * x;
    }
    double s = 0.0;
    for (int j = 0; j < 64; ++j) s += __shfl(p, j, 64);
    return s;
}
__global__ __launch_bounds__(256) void unit_norm_kernel(const float *__restrict__ d, int64_t A, int64_t L, float eps,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wavefront per row
    if (a >= A) return;
    const float *row = d + a * L;
    const double s = row_sum_squares(row, L, lane);
    const float n = sqrt_rn_f32((float)s);
    const float den = __fadd_rn(n, eps);
    float *orow = out + a * L;
    for (int64_t k = lane; k < L; k += 64) orow[k] = div_rn_f32(row[k], den);
}

// ------------------------------------------------------------------------------------------------
// Dictionary image: [atom tile][k chunk][k8 group][k parity][atom in tile][4] floats, so that lane
// (n = lane & 31, h = lane >> 5) of a wavefront reads, with ONE 16-byte LDS load at
//   ((g * 2 + h) * TA + n) * 16 bytes,
// the B operands d[n][8g + h], d[n][8g + 2 + h], d[n][8g + 4 + h], d[n][8g + 6 + h] of the four
// consecutive v_mfma_f32_32x32x2_f32 k-steps of group g -- ascending k, lanes contiguous (no bank
// conflicts).  Atoms >= A and samples >= L are zero (an fma with a zero product is exact).
// ------------------------------------------------------------------------------------------------
__global__ void dict_image_kernel(const float *__restrict__ du, int64_t A, int64_t L, int TA, int KC,
                                  int NCH, int NAT, float *__restrict__ img) {
    int64_t total = (int64_t)NAT * NCH * KC * TA;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int q = (int)(e & 3);
        int64_t r = e >> 2;
        int n = (int)(r % TA); r /= TA;
        int h = (int)(r & 1); r >>= 1;
        int g = (int)(r % (KC / 8)); r /= (KC / 8);
        int c = (int)(r % NCH);
        int tile = (int)(r / NCH);
        int64_t a = (int64_t)tile * TA + n;
        int64_t k = (int64_t)c * KC + g * 8 + q * 2 + h;
        img[e] = (a < A && k < L) ? du[a * L + k] : 0.0f;
    }
}

// `lead` zeros, the signal, zeros to the end of the row (lead = 0 for matching pursuit proper)
__global__ void init_residual_kernel(const float *__restrict__ signal, int64_t N, int64_t Ns, int64_t lead,
                                     float *__restrict__ res) {
    int64_t b = blockIdx.y;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < Ns;
         t += (int64_t)gridDim.x * blockDim.x)
        res[b * Ns + t] = (t >= lead && t - lead < N) ? signal[b * N + t - lead] : 0.0f;
}

__global__ void reverse_rows_kernel(const float *__restrict__ d, int64_t A, int64_t L, float *__restrict__ out) {
    const int64_t total = A * L;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = e / L, k = e % L;
        out[e] = d[a * L + (L - 1 - k)];
    }
}

__global__ void copy_residual_kernel(const float *__restrict__ res_in, int64_t N, int64_t Ns,
                                     float *__restrict__ out, int64_t lead) {
    const float *res = res_in + lead;
    int64_t b = blockIdx.y;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N;
         t += (int64_t)gridDim.x * blockDim.x)
        out[b * N + t] = res[b * Ns + t];
}

// ------------------------------------------------------------------------------------------------
// Shared pieces of the correlate kernels
// ------------------------------------------------------------------------------------------------
// Stage KC x TA floats of dictionary image (already in LDS order) into LDS; all 256 threads call it.
template <bool DMA>
__device__ __forceinline__ void stage_image(const f32x4 *__restrict__ src, float *img_s, int n4, int tid) {
    if (DMA) {
        // LDS-DMA: each wave-instruction moves 64 x 16 B = 1 KiB, LDS address = wave-uniform base +
        // lane * 16 (the image is linear in both memories, so no swizzle is needed).
        const int w = tid >> 6, lane = tid & 63;
        for (int e0 = w * 64; e0 < n4; e0 += 256) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + e0 + lane),
                (__attribute__((address_space(3))) void *)(img_s + (size_t)e0 * 4), 16, 0, 0);
        }
    } else {
        f32x4 *dst = reinterpret_cast<f32x4 *>(img_s);
        for (int e0 = tid; e0 < n4; e0 += 256 * 8) {  // n4 is a multiple of 2048 / (512/KC)... see callers
            f32x4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tmp[u] = (e0 + u * 256 < n4) ? src[e0 + u * 256] : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * 256 < n4) dst[e0 + u * 256] = tmp[u];
        }
    }
}

// One k-chunk of a cell: acc[s][nt] += R[64 lags x KC] * D^T[KC x 32*NT atoms]  (ascending k).
// img_s: staged image chunk; win_s: this wave's residual window (win_s[j] = r[t0 + k0 + j]).
template <int TA>
__device__ __forceinline__ void mfma_chunk(const float *img_s, const float *win_s, int KC, int i, int h,
                                           f32x16 (&acc)[2][TA / 32]) {
    constexpr int NT = TA / 32;
    const f32x4 *bp = reinterpret_cast<const f32x4 *>(img_s) + h * TA + i;
    const float *ap = win_s + i + h;
    const int ngrp = KC / 8;  // even
    // software pipeline, two k8-groups per trip: operands of the next group are in flight while the
    // 8*NT MFMAs (512*NT cycles) of the current one issue.
    f32x4 bA[NT], bB[NT];
    float aA[8], aB[8];
#define MP_LOAD_GROUP(BQ, AQ, G)                                                        \
    {                                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) BQ[nt] = bp[(size_t)(G) * 2 * TA + nt * 32]; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                 \
            AQ[2 * j] = ap[(G) * 8 + 2 * j];                                            \
            AQ[2 * j + 1] = ap[(G) * 8 + 2 * j + 32];                                   \
        }                                                                               \
    }
#define MP_MFMA_GROUP(BQ, AQ)                                                           \
    {                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                 \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                         \
                acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[2 * j], BQ[nt][j], acc[0][nt], 0, 0, 0);     \
                acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AQ[2 * j + 1], BQ[nt][j], acc[1][nt], 0, 0, 0); \
            }                                                                           \
        }                                                                               \
    }
    MP_LOAD_GROUP(bA, aA, 0)
    for (int g = 0; g < ngrp; g += 2) {
        MP_LOAD_GROUP(bB, aB, g + 1)
        MP_MFMA_GROUP(bA, aA)
        MP_LOAD_GROUP(bA, aA, g + 2)  // last trip reads the LDS pad: loaded, never used
        MP_MFMA_GROUP(bB, aB)
        // pin the issue order: a group's LDS reads go out ahead of the PREVIOUS group's MFMAs, so
        // their latency hides under 8*NT x 64 cycles of matrix work
        __builtin_amdgcn_sched_group_barrier(0x100, NT + 4, 0);  // DS reads (ds_read2 pairs)
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NT, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, NT + 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NT, 0);
    }
#undef MP_LOAD_GROUP
#undef MP_MFMA_GROUP
}

// Signed maximum of a finished cell with torch.max's tie rule, as a key (wave-uniform result).
// lane (i, h) holds, for sub-tile (s, nt) and register r:  atom = tile*TA + nt*32 + i,
// lag = t0 + s*32 + (r&3) + 8*(r>>2) + 4*h.  Invalid lanes/lags are overwritten with -inf in acc.
template <int TA>
__device__ __forceinline__ u64 cell_key(f32x16 (&acc)[2][TA / 32], int tile, int64_t t0, int64_t N,
                                        int64_t A, int i, int h) {
    constexpr int NT = TA / 32;
    const bool whole = (t0 + LAGS_PER_WAVE <= N) && ((int64_t)(tile + 1) * TA <= A);  // wave-uniform
    if (!whole) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t atom = (int64_t)tile * TA + nt * 32 + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t lag = t0 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (!((lag < N) && (atom < A))) acc[s][nt][r] = -INFINITY;
                }
            }
    }
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[s][nt][r]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // lowest flat index holding the wave maximum: walk downwards, last write wins
    unsigned best = 0xffffffffu;
#pragma unroll
    for (int nt = NT - 1; nt >= 0; --nt) {
        const int64_t atom = (int64_t)tile * TA + nt * 32 + i;
        const unsigned base = (unsigned)(atom * N + t0 + 4 * h);  // < 2^32 whenever atom < A
        const bool atom_ok = atom < A;
#pragma unroll
        for (int s = 1; s >= 0; --s)
#pragma unroll
            for (int r = 15; r >= 0; --r) {
                const int dl = s * 32 + (r & 3) + 8 * (r >> 2);
                bool hit = acc[s][nt][r] == m;
                if (!whole) hit = hit && atom_ok && (t0 + dl + 4 * h < N);
                best = hit ? base + (unsigned)dl : best;
            }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    return best == 0xffffffffu ? 0ull : make_key(m, best);
}

// ------------------------------------------------------------------------------------------------
// Cell-order map (the local-contrast-norm schedule's dense map, TA = 32): cell (b, blk, tile) keeps its 32 atoms x 64 lags
// as 2048 floats in ACCUMULATOR order -- float4 number (s * 4 + rq) * 64 + lane holds registers r = 4 rq .. 4 rq + 3 of
// sub-tile s of lane (i = lane & 31, h = lane >> 5), i.e. atom i, lags s * 32 + 8 rq + 4 h + (0 .. 3).  A wavefront
// stores its finished cell with eight 1 KiB-contiguous instructions (the [B, A, N] layout takes 64 strided 4-byte
// rows per instruction); readers un-permute with cell_map_offset.
// ------------------------------------------------------------------------------------------------
constexpr int CELL_FLOATS = 32 * LAGS_PER_WAVE;
__host__ __device__ __forceinline__ int cell_map_offset(int i, int dl) {   // atom i of the tile, lag dl of the block
    const int s = dl >> 5, d32 = dl & 31;
    return (((s * 4 + (d32 >> 3)) * 64 + i + 32 * ((d32 >> 2) & 1)) << 2) + (d32 & 3);
}
__device__ __forceinline__ void store_cell_map(float *__restrict__ cell, const f32x16 (&acc)[2][1], int lane) {
    f32x4 *dst = reinterpret_cast<f32x4 *>(cell);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4 q = {acc[s][0][4 * rq], acc[s][0][4 * rq + 1], acc[s][0][4 * rq + 2], acc[s][0][4 * rq + 3]};
            dst[(s * 4 + rq) * 64 + lane] = q;
        }
}

// ------------------------------------------------------------------------------------------------
// Correlate: one wavefront = one cell task (segment b, 64-lag block, atom tile); the 4 wavefronts of a
// workgroup take 4 consecutive tasks of the SAME atom tile (so they share the staged dictionary
// image) from the flattened list  task = b * stride + c,  c-th block of segment b's dirty run
// (stride = blocks per segment for a full pass, max dirty blocks for an incremental one); a
// workgroup may therefore straddle segments, and every wavefront stages its own residual window.
// GEMM view per wavefront: C[64 lags x TA atoms] = R[64 x L] (Toeplitz: R[t][k] = r[t+k])
// times D^T[L x TA], on v_mfma_f32_32x32x2_f32 (A operand = residual, B operand = dictionary).
//   A operand lane map: lane l holds R[row l&31][k l>>5]  -> LDS word  win[(l&31) + (l>>5) + k]
//   B operand lane map: lane l holds D[k l>>5][col l&31]  -> component j of the 16-byte image word
//   C/D: col = lane & 31 (atom), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (lag)
// Every product chain is ascending in k: bit-identical to the oracle's fmaf chain.
// STORE_FM: additionally write the dense map (hooks that need it); keys are still produced.
// ------------------------------------------------------------------------------------------------
// CELLMAP (TA = 32 only): `fm` is the cell-order map above and no keys are written (the LCN schedule has its own).
template <int TA, bool STORE_FM, bool DMA, bool CELLMAP = false>
__global__ __launch_bounds__(256) void correlate_mfma_kernel(
    const float *__restrict__ res, const float *__restrict__ img, const int *__restrict__ dirty,
    u64 *__restrict__ keys, float *__restrict__ fm, int64_t N, int64_t A, int64_t Ns, int64_t B, int NBLK,
    int NAT, int KC, int NCH, int stride) {
    constexpr int NT = TA / 32;  // 32-atom sub-tiles per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *img_s = reinterpret_cast<float *>(smem);  // KC * TA floats
    const int WIN = KC + LAGS_PER_WAVE + 16;         // per-wavefront residual window (+ prefetch pad)
    float *win_all = img_s + (size_t)KC * TA;

    const int tile = blockIdx.y;
    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    const int i = lane & 31;
    const int h = lane >> 5;

    // this wavefront's task (everything here is wave-uniform)
    const int64_t task = (int64_t)blockIdx.x * WAVES + w;
    const int64_t b = task / stride;
    const int c = (int)(task - b * stride);
    int first = 0, count = NBLK;
    bool active = b < B;
    if (active && dirty) {
        first = dirty[2 * b];
        count = dirty[2 * b + 1];
    }
    active = active && (c < count);
    const int blk = first + c;
    const int64_t t0 = (int64_t)blk * LAGS_PER_WAVE;
    float *win_s = win_all + (size_t)w * WIN;

    f32x16 acc[2][NT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][nt][r] = 0.0f;

    const float *res_b = res + (active ? b : 0) * Ns;
    const f32x4 *img_g = reinterpret_cast<const f32x4 *>(img) + (size_t)tile * NCH * (KC * TA / 4);

    for (int c_k = 0; c_k < NCH; ++c_k) {
        if (c_k > 0) __syncthreads();  // previous chunk fully consumed
        // ---- stage the dictionary chunk and this wave's residual window ---------------------------
        stage_image<DMA>(img_g + (size_t)c_k * (KC * TA / 4), img_s, KC * TA / 4, tid);
        if (active) {
            const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(res_b + t0 + (int64_t)c_k * KC);
            f32x4 *wdst = reinterpret_cast<f32x4 *>(win_s);
            const int w4 = (KC + LAGS_PER_WAVE) / 4;
            for (int e = lane; e < w4; e += 64) wdst[e] = wsrc[e];
        }
        __syncthreads();  // hipcc drains vmcnt(0) here while LDS-DMA is outstanding
        if (active) mfma_chunk<TA>(img_s, win_s, KC, i, h, acc);
    }
    if (!active) return;

    if constexpr (CELLMAP) {
        static_assert(!CELLMAP || TA == 32, "the cell-order map is for 32-atom tiles");
        store_cell_map(fm + ((b * NBLK + blk) * NAT + tile) * CELL_FLOATS, acc, lane);
        return;
    }
    const u64 key = cell_key<TA>(acc, tile, t0, N, A, i, h);
    if (lane == 0) keys[(b * NBLK + blk) * NAT + tile] = key;

    if (STORE_FM) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t atom = (int64_t)tile * TA + nt * 32 + i;
                if (atom < A) {
                    float *row = fm + (b * A + atom) * N;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t lag = t0 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (lag < N) row[lag] = acc[s][nt][r];
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent correlate (atoms of at most KC_MAX samples, i.e. one k-chunk): the grid is sized to the
// machine (WGS_PER_CU workgroups per CU) instead of to the problem.  The flattened task list
//   task = tile * LT + lagtask,   lagtask = b * stride + c   (LT = B * stride)
// is cut into equal contiguous ranges, one per workgroup.  A workgroup stages the dictionary image
// of a tile ONCE and its four wavefronts then pull that range's cell tasks from an LDS counter, so a
// wavefront never waits for another between cells and staging is amortised over many cells.
// ------------------------------------------------------------------------------------------------
template <int TA, bool DMA, bool CELLMAP = false>
__global__ __launch_bounds__(256) void correlate_persistent_kernel(
    const float *__restrict__ res, const float *__restrict__ img, const int *__restrict__ dirty,
    u64 *__restrict__ keys, int64_t N, int64_t A, int64_t Ns, int NBLK, int NAT, int KC, int stride,
    int64_t LT, int64_t tasks_per_wg, int stagger, float *__restrict__ cmap /* CELLMAP: the cell-order map; else unused */) {
    constexpr int NT = TA / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *img_s = reinterpret_cast<float *>(smem);  // KC * TA floats
    const int WIN = KC + LAGS_PER_WAVE + 16;
    float *win_all = img_s + (size_t)KC * TA;
    if (stagger) {
        // Two wavefronts share each SIMD's matrix pipe and run the same program: started together
        // they reach their (pipe-idle) epilogues together.  Delay the one in the odd wave slot by
        // about half a cell so that one computes while the other finishes a cell.
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID.WAVE_ID
        if (slot & 1) {
            for (int z = 0; z < stagger; ++z) __builtin_amdgcn_s_sleep(127);
        }
    }
    int *counter = reinterpret_cast<int *>(win_all + (size_t)WAVES * WIN);  // inside the trailing pad

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    const int i = lane & 31;
    const int h = lane >> 5;
    float *win_s = win_all + (size_t)w * WIN;

    const int64_t T = LT * NAT;
    int64_t lo = (int64_t)blockIdx.x * tasks_per_wg;
    int64_t hi = lo + tasks_per_wg;
    if (hi > T) hi = T;

    // Every loop below has a trip bound every wavefront reaches (a persistent kernel must drain).
    for (int seg = 0; seg <= NAT && lo < hi; ++seg) {  // uniform across the workgroup
        const int tile = (int)(lo / LT);
        int64_t seg_end = (int64_t)(tile + 1) * LT;
        if (seg_end > hi) seg_end = hi;
        const int n_here = (int)(seg_end - lo);
        const int64_t lagtask0 = lo - (int64_t)tile * LT;

        stage_image<DMA>(reinterpret_cast<const f32x4 *>(img) + (size_t)tile * (KC * TA / 4), img_s,
                         KC * TA / 4, tid);
        if (tid == 0) *counter = 0;
        __syncthreads();

        // A wavefront always holds its NEXT cell's residual window in registers (3 x 16 B per lane,
        // loaded while the current cell's MFMAs run), so no cell starts with an exposed HBM/L2 round trip.
        constexpr int WQ = 3;  // ceil((KC_MAX + 64) / 4 / 64)
        const int w4 = (KC + LAGS_PER_WAVE) / 4;
        f32x4 pre[WQ];
        int64_t nb = 0;       // segment of the prefetched cell
        int nblk = -1;        // its lag block, -1 = none (queue drained or empty slot)
        int q = n_here;
        auto grab = [&]() {
            int v = 0;
            if (lane == 0) v = atomicAdd(counter, 1);
            q = __builtin_amdgcn_readfirstlane(v);  // scalar from here on: the control flow is uniform
            nblk = -1;
            if (q < n_here) {
                const int64_t lagtask = lagtask0 + q;
                nb = lagtask / stride;
                const int c = (int)(lagtask - nb * stride);
                int first = 0, count = NBLK;
                if (dirty) {
                    first = __builtin_amdgcn_readfirstlane(dirty[2 * nb]);
                    count = __builtin_amdgcn_readfirstlane(dirty[2 * nb + 1]);
                }
                if (c < count) {  // else: an empty slot of this segment's dirty run
                    nblk = first + c;
                    const f32x4 *wsrc =
                        reinterpret_cast<const f32x4 *>(res + nb * Ns + (int64_t)nblk * LAGS_PER_WAVE);
#pragma unroll
                    for (int u = 0; u < WQ; ++u)
                        if (lane + 64 * u < w4) pre[u] = wsrc[lane + 64 * u];
                }
            }
        };
        grab();
        for (int trip = 0; trip <= n_here && q < n_here; ++trip) {
            const int blk = nblk;
            const int64_t b = nb;
            if (blk >= 0) {
                f32x4 *wdst = reinterpret_cast<f32x4 *>(win_s);
#pragma unroll
                for (int u = 0; u < WQ; ++u)
                    if (lane + 64 * u < w4) wdst[lane + 64 * u] = pre[u];
            }
            grab();  // next cell's window goes in flight now, lands while this cell computes
            if (blk >= 0) {
                const int64_t t0 = (int64_t)blk * LAGS_PER_WAVE;
                f32x16 acc[2][NT];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[s][nt][r] = 0.0f;
                mfma_chunk<TA>(img_s, win_s, KC, i, h, acc);
                if constexpr (CELLMAP) {
                    store_cell_map(cmap + ((b * NBLK + blk) * NAT + tile) * CELL_FLOATS, acc, lane);
                } else {
                    const u64 key = cell_key<TA>(acc, tile, t0, N, A, i, h);
                    if (lane == 0) keys[(b * NBLK + blk) * NAT + tile] = key;
                }
            }
        }
        __syncthreads();  // every wavefront is done with this image (and with the counter)
        lo = seg_end;
    }
}

// ------------------------------------------------------------------------------------------------
// Validation kernel (MP_PATH_NAIVE): one thread per lag, plain fmaf chain from global memory,
// cells of 1 atom x 64 lags.  Same keys contract; no MFMA, no LDS.
// ------------------------------------------------------------------------------------------------
__global__ void correlate_naive_kernel(const float *__restrict__ res, const float *__restrict__ du,
                                       const int *__restrict__ dirty, u64 *__restrict__ keys,
                                       float *__restrict__ fm, int64_t N, int64_t A, int64_t L,
                                       int64_t Ns, int NBLK) {
    const int b = blockIdx.z;
    const int64_t atom = blockIdx.y;
    int first = 0, count = NBLK;
    if (dirty) {
        first = dirty[2 * b];
        count = dirty[2 * b + 1];
    }
    if ((int)blockIdx.x >= count) return;
    const int blk = first + blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t lag = (int64_t)blk * LAGS_PER_WAVE + lane;
    const float *r = res + (int64_t)b * Ns + lag;
    const float *d = du + atom * L;
    float acc = 0.0f;
    for (int64_t k = 0; k < L; ++k) acc = __fmaf_rn(r[k], d[k], acc);
    u64 key = lag < N ? make_key(acc, (unsigned)(atom * N + lag)) : 0ull;
    key = wave_max_u64(key);
    if (lane == 0) keys[((int64_t)b * NBLK + blk) * A + atom] = key;
    if (fm && lag < N) fm[((int64_t)b * A + atom) * N + lag] = acc;
}

// ------------------------------------------------------------------------------------------------
// Select: argmax over a segment's cell keys, record the event, subtract gain * atom (cropped at N,
// two roundings like `residual -= d[atom] * value`, matchingpursuit.py:305,:328), mark dirty blocks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_subtract_kernel(
    const u64 *__restrict__ keys, int64_t n_keys, float *__restrict__ res, const float *__restrict__ du,
    int *__restrict__ dirty, int64_t *__restrict__ out_atom, int64_t *__restrict__ out_lag,
    float *__restrict__ out_gain, int64_t N, int64_t L, int64_t Ns, int NBLK, int K, int k,
    const int *__restrict__ cont, const int *__restrict__ ncont, u64 *__restrict__ keys_wb,
    float *__restrict__ ceps_wb, int64_t n_cells, int shift, int square) {
    __shared__ u64 s_key[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const u64 *kb = keys + (int64_t)b * n_keys;
    if (cont) {  // FFT path: the refined cells' exact keys replace the screened ones (eps = 0)
        const int nc = ncont[b];
        if (tid < nc) {
            const int64_t cidx = (int64_t)b * n_cells + cont[b * MAXCONT + tid];
            keys_wb[cidx] = kb[tid];
            ceps_wb[cidx] = 0.0f;
        }
    }
    u64 best = 0;
    for (int64_t e = tid; e < n_keys; e += 256) {
        u64 v = kb[e];
        best = v > best ? v : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = s_key[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) best = s_key[q] > best ? s_key[q] : best;

    const unsigned flat = 0xffffffffu - (unsigned)(best & 0xffffffffull);
    const float gain = unord_f32((unsigned)(best >> 32));
    const int64_t atom = (int64_t)(flat / (u64)N);
    const int64_t lag = (int64_t)(flat % (u64)N);
    if (tid == 0) {
        out_atom[(int64_t)b * K + k] = atom;
        out_lag[(int64_t)b * K + k] = lag;
        out_gain[(int64_t)b * K + k] = gain;
        if (dirty) {
            int64_t lo = lag + shift - L + 1; if (lo < 0) lo = 0;
            int64_t hi = lag + shift + L - 1; if (hi > N - 1) hi = N - 1;
            const int fb = (int)(lo / LAGS_PER_WAVE), lb = (int)(hi / LAGS_PER_WAVE);
            dirty[2 * b] = fb;
            dirty[2 * b + 1] = lb - fb + 1;
        }
    }
    if (keys_wb) {  // FFT path: the screen merges with atomicMax, so the cells it will redo start from 0
        int64_t lo = lag + shift - L + 1; if (lo < 0) lo = 0;
        int64_t hi = lag + shift + L - 1; if (hi > N - 1) hi = N - 1;
        const int fb = (int)(lo / LAGS_PER_WAVE), lb = (int)(hi / LAGS_PER_WAVE);
        const int nat = (int)(n_cells / NBLK);
        u64 *kz = keys_wb + (int64_t)b * n_cells + (int64_t)fb * nat;
        const int nz = (lb - fb + 1) * nat;
        for (int e = tid; e < nz; e += 256) kz[e] = 0ull;
    }
    // matching pursuit: r[lag + s] -= d[atom][s] * gain (shift 0); convolution model (mp.py:61-64):
    // the FORWARD atom lands at output sample lag, i.e. row position lag + L - 1, scaled by gain^2
    const int64_t len = (N - lag) < L ? (N - lag) : L;
    float *r = res + (int64_t)b * Ns + lag + shift;
    const float *d = du + atom * L;
    const float g2 = square ? __fmul_rn(gain, gain) : gain;
    for (int64_t s = tid; s < len; s += 256) r[s] = __fsub_rn(r[s], __fmul_rn(d[s], g2));
}

// ------------------------------------------------------------------------------------------------
// Backward pass of the gradient-trained model's analysis loop (mp.py:54-66; what autograd does to
//   v_i = conv(r_i, atoms)[a_i, t_i],  b_i = v_i^2 * atom_{a_i} placed at t_i (cropped at N),  r_{i+1} = r_i - b_i,
// channels[:, i] = b_i) in ONE launch: one workgroup per segment walks the K steps from the last to the first.
// With G the gradient arriving at the channels and lam = d loss / d r_{i+1}:
//   r_i    = r_{i+1} + b_i                                   (the residual, walked back from r_K)
//   gw[j]  = G[i, t+j] - lam[t+j]                            (d loss / d b_i on the atom's support)
//   dv     = 2 v sum_j gw[j] atom[j]                         (through b_i = v^2 atom)
//   row[j] = v^2 gw[j] + dv r_i[t-j]                         (d loss / d atom_{a_i} from this event)
//   lam[t-j] += dv atom[j]                                    (through v_i = sum_j r_i[t-j] atom[j])
// The recurrence is sequential in i by construction; as ~17 tensor operations per step it was launch-bound
// (4.4 ms of GPU time and 6.9 ms of host time per train step at the config-5 shape).  `rows` [B, K, L] are
// summed into the dictionary gradient by the caller (one index_add); lam ends as d loss / d audio.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_model_backward_kernel(
    const float *__restrict__ atoms, int64_t L, const int64_t *__restrict__ a_idx, const int64_t *__restrict__ t_idx,
    const float *__restrict__ val, int K, const float *__restrict__ res_final, const float *__restrict__ gch,
    int64_t N, float *lam, float *rows, float *r, int windowed) {
    __shared__ float s_red[4];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    float *rb = r + b * N, *lb = lam + b * N;
    // the gradient arriving at the channels: dense [B, K, N], or only each event's own support [B, K, L]
    const float *gb = gch + b * K * (windowed ? L : N);
    for (int64_t n = tid; n < N; n += 256) {
        rb[n] = res_final[b * N + n];
        lb[n] = 0.0f;
    }
    __syncthreads();
    for (int i = K - 1; i >= 0; --i) {
        const int64_t a = a_idx[b * K + i], t = t_idx[b * K + i];
        const float v = val[b * K + i], v2 = v * v;
        const float *d = atoms + a * L;
        float *row = rows + (b * K + i) * L;
        float part = 0.0f;
        for (int64_t j = tid; j < L; j += 256) {
            const int64_t pos = t + j;
            float gw = 0.0f;
            if (pos < N) {
                rb[pos] = rb[pos] + v2 * d[j];
                gw = (windowed ? gb[(int64_t)i * L + j] : gb[(int64_t)i * N + pos]) - lb[pos];
                part += gw * d[j];
            }
            row[j] = gw;  // parked until dv is known (lam[t] is about to change)
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = part;
        __syncthreads();  // r_i complete, every lam[t+j] read
        const float dv = 2.0f * v * (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
        for (int64_t j = tid; j < L; j += 256) {
            const int64_t back = t - j;
            const float rw = back >= 0 ? rb[back] : 0.0f;
            row[j] = v2 * row[j] + dv * rw;
            if (back >= 0) lb[back] = lb[back] + dv * d[j];
        }
        __syncthreads();  // lam updated before the next step reads it; s_red free
    }
}

// ------------------------------------------------------------------------------------------------
// Local contrast norm (modules/matchingpursuit.py:284-294): the step's event is the argmax of
//   lcn[a,t] = fm[a,t] - avg_pool2d(fm, 9x9, stride 1, zero pad 4, count_include_pad)[a,t]
// over the dense (A, N) map, and its gain is the RAW map value there (:294).  avg is ONE sequential fp32 sum
// over the window in row-major order (a' ascending, then t' ascending; zeros outside the map add nothing)
// divided by 81 -- ATen's cpu_avg_pool2d loop, restated in oracle/mp_oracle.c:mpo_encode_lcn.
// The map is kept in HBM in CELL ORDER (above) and only its dirty 64-lag blocks are recomputed each step (the MFMA
// correlate, persistent form, storing cells instead of keys); this kernel then redoes the LCN keys of those blocks and one
// block either side (the box reaches 4 lags across a block edge).  One workgroup = one map cell of 32 atoms x 64 lags,
// staged with its halo (40 x 72 floats) in LDS: the cell itself by 1 KiB-contiguous loads, the halo's 832 values gathered
// from the eight neighbouring cells.  A wavefront's 64 lanes read 64 consecutive floats of one row (row stride 73 words),
// so the LDS reads are conflict-free; the four rows a wavefront walks together share nine of their twelve box rows.
// ------------------------------------------------------------------------------------------------
constexpr int LCN_ROWS = 32 + 8;
constexpr int LCN_W = LAGS_PER_WAVE + 8;

__device__ __forceinline__ float cell_map_at(const float *__restrict__ cmap, int64_t b, int64_t a, int64_t t, int NBLK, int NAT) {
    const int64_t cell = (b * NBLK + (t >> 6)) * NAT + (a >> 5);
    return cmap[cell * CELL_FLOATS + cell_map_offset((int)(a & 31), (int)(t & 63))];
}

__global__ __launch_bounds__(256) void lcn_keys_kernel(const float *__restrict__ cmap, const int *__restrict__ dirty,
                                                       u64 *__restrict__ lkeys, int64_t N, int64_t A, int NBLK,
                                                       int NAT) {
    __shared__ float tile[LCN_ROWS][LCN_W + 1];
    __shared__ u64 s_key[4];
    const int b = blockIdx.z;
    const int atile = blockIdx.y;
    int first = 0, last = NBLK - 1;
    if (dirty) {
        first = dirty[2 * b] - 1;
        last = dirty[2 * b] + dirty[2 * b + 1];
        if (first < 0) first = 0;
        if (last > NBLK - 1) last = NBLK - 1;
    }
    const int blk = first + blockIdx.x;
    if (blk > last) return;  // uniform across the workgroup
    const int tid = threadIdx.x;
    const int64_t a0 = (int64_t)atile * 32 - 4, t0 = (int64_t)blk * LAGS_PER_WAVE - 4;
    // the cell itself: 512 float4 in accumulator order (atoms >= A and lags >= N hold exact zeros: a zero image row / a zero
    // residual make zero chains)
    {
        const f32x4 *cb = reinterpret_cast<const f32x4 *>(cmap + (((int64_t)b * NBLK + blk) * NAT + atile) * CELL_FLOATS);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            const f32x4 q = cb[e];
            const int lane = e & 63, sr = e >> 6;
            const int r = (lane & 31) + 4, c = (sr >> 2) * 32 + 8 * (sr & 3) + 4 * (lane >> 5) + 4;
            tile[r][c] = q[0]; tile[r][c + 1] = q[1]; tile[r][c + 2] = q[2]; tile[r][c + 3] = q[3];
        }
    }
    // the halo: 4 rows above and below (72 wide), 4 columns left and right of the 32 centre rows
    for (int e = tid; e < 8 * LCN_W + 32 * 8; e += 256) {
        int r, c;
        if (e < 8 * LCN_W) {
            r = e / LCN_W;
            c = e - r * LCN_W;
            if (r >= 4) r += 32;
        } else {
            const int e2 = e - 8 * LCN_W;
            r = 4 + (e2 >> 3);
            c = e2 & 7;
            if (c >= 4) c += LAGS_PER_WAVE;
        }
        const int64_t a = a0 + r, t = t0 + c;
        tile[r][c] = (a >= 0 && a < A && t >= 0 && t < N) ? cell_map_at(cmap, b, a, t, NBLK, NAT) : 0.0f;
    }
    __syncthreads();
    const int w = tid >> 6, lane = tid & 63;
    u64 best = 0;
    // A wavefront owns eight rows of the cell (lane = lag).  The eight box sums walk the sixteen tile rows they cover
    // TOGETHER, top to bottom: a row's nine values are read once (144 LDS reads per eight outputs where eight separate
    // walks read 648 -- the kernel was bound by exactly that: 160 ds_read2 per four outputs, 187 us per incremental
    // launch at the headline shape) and added, left to right, to every sum whose box holds the row.  Each sum still sees
    // its 81 values in the reference's order (rows ascending, then lags ascending); two neighbouring sums that both hold the
    // row take the value in ONE v_pk_add_f32 (two IEEE additions, the operand's low half broadcast).
    {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 sum[4];
        float centre[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) sum[p] = f32x2{0.0f, 0.0f};
        const float *col = &tile[w * 8][lane];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            // the row's nine values as register PAIRS (two neighbouring lags per LDS read: ds_read2_b32 fills a pair), so that
            // the packed additions take a value straight from the half of the pair it sits in (op_sel) -- built from single
            // registers every packed add cost a v_mov and a hazard s_nop first: 84 + 82 of the loop's 890 instructions
            f32x2 w[5];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = f32x2{col[rr * (LCN_W + 1) + 2 * q], col[rr * (LCN_W + 1) + 2 * q + 1]};
            w[4] = f32x2{col[rr * (LCN_W + 1) + 8], 0.0f};
            float v[9];
#pragma unroll
            for (int dt = 0; dt < 9; ++dt) v[dt] = (dt & 1) ? w[dt >> 1].y : w[dt >> 1].x;
            if (rr >= 4 && rr < 12) centre[rr - 4] = v[4];
            // lags outer, sums inner: consecutive packed additions then belong to DIFFERENT sums (a packed result needs a
            // wait state before a dependent instruction may read it: four interleaved chains hide it, one chain at a time
            // had an s_nop behind every addition)
#pragma unroll
            for (int dt = 0; dt < 9; ++dt) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int lo = 2 * p, hi = 2 * p + 1;           // outputs (rows of the cell) of this pair of sums
                    const bool in_lo = rr >= lo && rr <= lo + 8, in_hi = rr >= hi && rr <= hi + 8;
                    if (in_lo && in_hi) {
                        if (dt & 1)   // both sums += the HIGH half of the register pair
                            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(sum[p]) : "v"(sum[p]), "v"(w[dt >> 1]));
                        else          // both sums += the LOW half
                            asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(sum[p]) : "v"(sum[p]), "v"(w[dt >> 1]));
                    } else if (in_lo) {
                        sum[p].x = __fadd_rn(sum[p].x, v[dt]);
                    } else if (in_hi) {
                        sum[p].y = __fadd_rn(sum[p].y, v[dt]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ra = w * 8 + j;
            const float sj = (j & 1) ? sum[j >> 1].y : sum[j >> 1].x;
            const float val = __fsub_rn(centre[j], div_rn_f32(sj, 81.0f));
            const int64_t a = a0 + 4 + ra, t = t0 + 4 + lane;
            const u64 key = (a < A && t < N) ? make_key(val, (unsigned)(a * N + t)) : 0ull;
            best = key > best ? key : best;
        }
    }
    best = wave_max_u64(best);
    if (lane == 0) s_key[w] = best;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) best = s_key[q] > best ? s_key[q] : best;
        lkeys[((int64_t)b * NBLK + blk) * NAT + atile] = best;
    }
}

// argmax over a segment's LCN keys; the gain is read from the raw map; then as select_subtract_kernel
__global__ __launch_bounds__(256) void lcn_select_subtract_kernel(
    const u64 *__restrict__ lkeys, int64_t n_keys, const float *__restrict__ cmap, float *__restrict__ res,
    const float *__restrict__ du, int *__restrict__ dirty, int64_t *__restrict__ out_atom,
    int64_t *__restrict__ out_lag, float *__restrict__ out_gain, int64_t N, int64_t A, int64_t L, int64_t Ns,
    int K, int k, int NBLK, int NAT) {
    __shared__ u64 s_key[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const u64 *kb = lkeys + (int64_t)b * n_keys;
    u64 best = 0;
    for (int64_t e = tid; e < n_keys; e += 256) {
        u64 v = kb[e];
        best = v > best ? v : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = s_key[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) best = s_key[q] > best ? s_key[q] : best;
    const unsigned flat = 0xffffffffu - (unsigned)(best & 0xffffffffull);
    const int64_t atom = (int64_t)(flat / (u64)N);
    const int64_t lag = (int64_t)(flat % (u64)N);
    const float gain = cell_map_at(cmap, b, atom, lag, NBLK, NAT);  // torch.gather(fm, index=mx), :294
    __syncthreads();  // every thread has read the map before the residual (not the map) changes: no hazard,
                      // but keep the order explicit for the next launch's readers
    if (tid == 0) {
        out_atom[(int64_t)b * K + k] = atom;
        out_lag[(int64_t)b * K + k] = lag;
        out_gain[(int64_t)b * K + k] = gain;
        int64_t lo = lag - L + 1; if (lo < 0) lo = 0;
        int64_t hi = lag + L - 1; if (hi > N - 1) hi = N - 1;
        const int fb = (int)(lo / LAGS_PER_WAVE), lb = (int)(hi / LAGS_PER_WAVE);
        dirty[2 * b] = fb;
        dirty[2 * b + 1] = lb - fb + 1;
    }
    const int64_t len = (N - lag) < L ? (N - lag) : L;
    float *r = res + (int64_t)b * Ns + lag;
    const float *d = du + atom * L;
    for (int64_t s = tid; s < len; s += 256) r[s] = __fsub_rn(r[s], __fmul_rn(d[s], gain));
}

// ------------------------------------------------------------------------------------------------
// scatter_segments (modules/matchingpursuit.py:20-58): events of a segment may overlap, and the reference adds them
// to the output one after another -- per SAMPLE the order of the additions is the order of the list, and that is all the
// result depends on.  So the kernel is driven by the output: a workgroup owns SCATTER_CHUNK samples of one segment
// (4 per thread, in registers), compacts -- in list order, a block of SCATTER_EB events at a time -- the events of its
// segment that reach into its chunk, and every thread adds the ones covering its samples, in that order, with the same
// two roundings (product, sum).  Round 1's form -- one workgroup per segment walking the WHOLE list, a dependent scalar
// load per event, 64 workgroups on 256 CUs -- took 0.2 ms of a 3.5 ms sparse_code + scatter at the headline shape.
// ------------------------------------------------------------------------------------------------
constexpr int SCATTER_CHUNK = 4096;   // samples per workgroup (1024 threads x 4)
constexpr int SCATTER_EB = 2048;      // events compacted per round
template <bool ROWS>
__global__ __launch_bounds__(1024) void scatter_kernel(const float *__restrict__ rows,
                                                       const int64_t *__restrict__ atom,
                                                       const int64_t *__restrict__ batch,
                                                       const int64_t *__restrict__ lag,
                                                       const float *__restrict__ gain, int64_t n_events,
                                                       const float *__restrict__ du, int64_t L,
                                                       float *__restrict__ out, int64_t N) {
    __shared__ int64_t s_lag[SCATTER_EB];
    __shared__ int64_t s_src[SCATTER_EB];    // row offset (floats) of the event's atom / row
    __shared__ float s_gain[SCATTER_EB];
    __shared__ int s_wave[16];
    __shared__ int s_n;
    const int64_t b = blockIdx.x;
    const int64_t t0 = (int64_t)blockIdx.y * SCATTER_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float *o = out + b * N;
    float acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = t0 + tid + 1024 * i;
        acc[i] = t < N ? o[t] : 0.0f;
    }
    for (int64_t e0 = 0; e0 < n_events; e0 += SCATTER_EB) {
        if (tid == 0) s_n = 0;
        __syncthreads();
        // order-preserving compaction, 1024 events per pass: ballot within the wavefront, wave totals through LDS
        for (int64_t j0 = 0; j0 < SCATTER_EB && e0 + j0 < n_events; j0 += 1024) {
            const int64_t e = e0 + j0 + tid;
            bool hit = false;
            int64_t p = 0;
            if (e < n_events && batch[e] == b) {
                p = lag[e];
                hit = p + L > t0 && p < t0 + SCATTER_CHUNK;
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_wave[w] = __popcll(m);
            __syncthreads();
            int base = s_n;
            for (int q = 0; q < w; ++q) base += s_wave[q];
            if (hit) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                s_lag[pos] = p;
                s_src[pos] = ROWS ? e * L : atom[e] * L;
                s_gain[pos] = ROWS ? 1.0f : gain[e];
            }
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int q = 0; q < 16; ++q) tot += s_wave[q];
                s_n += tot;
            }
            __syncthreads();
        }
        const int n = s_n;
        const float *srcb = ROWS ? rows : du;
        for (int q = 0; q < n; ++q) {
            const int64_t p = s_lag[q];
            const float *src = srcb + s_src[q];
            const float g = s_gain[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t t = t0 + tid + 1024 * i;
                const int64_t s = t - p;
                if (s >= 0 && s < L && t < N) {
                    const float v = ROWS ? src[s] : __fmul_rn(src[s], g);
                    acc[i] = __fadd_rn(acc[i], v);
                }
            }
        }
        __syncthreads();   // the lists are rebuilt by the next round
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t t = t0 + tid + 1024 * i;
        if (t < N) o[t] = acc[i];
    }
}

// gather_segments + sum over instances (modules/matchingpursuit.py:369-378, :400-401)
__global__ void gather_sum_kernel(const float *__restrict__ x, int64_t N, const int64_t *__restrict__ batch,
                                  const int64_t *__restrict__ lag, int64_t n_events, int64_t L,
                                  double *__restrict__ out) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    double acc = 0.0;
    for (int64_t e = 0; e < n_events; ++e) {
        const int64_t t = lag[e] + s;
        if (t >= 0 && t < N) acc += (double)x[batch[e] * N + t];
    }
    out[s] = acc;
}

// ... the same for several groups of events at once (one dependency level of the multi-GPU dictionary update: what is
// all-reduced is then ONE [groups, L] matrix per level): blockIdx.y = group, events off[g] .. off[g + 1] - 1 in order
__global__ void gather_sum_groups_kernel(const float *__restrict__ x, int64_t N, const int64_t *__restrict__ batch,
                                         const int64_t *__restrict__ lag, const int64_t *__restrict__ off, int64_t L,
                                         double *__restrict__ out) {
    const int64_t g = blockIdx.y;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= L) return;
    const int64_t e0 = off[g] - off[0], e1 = off[g + 1] - off[0];   // (offsets may be a slice of a longer table)
    double acc = 0.0;
    for (int64_t e = e0; e < e1; ++e) {
        const int64_t t = lag[e] + s;
        if (t >= 0 && t < N) acc += (double)x[batch[e] * N + t];
    }
    out[g * L + s] = acc;
}

// ------------------------------------------------------------------------------------------------
// One dependency level of dictionary_learning_step across RANKS, in two phases around the one all-reduce of the level's
// [groups, L] window sums (mpcore/matchingpursuit.py::_dictionary_update_by_levels): what the single-process level kernel
// below does per atom, cut where the other ranks' segments come in.  One workgroup per group (= used atom), events
// off[g] .. off[g + 1] - 1 of the event arrays (`off`: the level's slice of the offset table, ABSOLUTE positions in the
// whole event arrays), this rank's segments only.
//   phase A  residual += the atom's rows (:395-396), acc[g] = fp64 sum over its events of the residual windows (:400-401)
//   phase B  residual -= new_atoms[g] * ||row_e|| (:408-415), new_atoms = unit_norm(all-reduced acc) (:403-404)
// An atom none of whose events overlap (overlap[g] == 0 -- almost every atom) needs no staging: its events touch disjoint
// samples, each sample sees one add (0 + row == row exactly) / one subtract; overlapping events of one atom (same
// segment, hence same rank) go through `sparse` event after event, as the reference's scatter sums them first.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dictionary_level_addback_sum_kernel(
    float *residual, float *sparse, int64_t N, int64_t L, const int64_t *__restrict__ off,
    const int64_t *__restrict__ ev_batch, const int64_t *__restrict__ ev_lag, const float *__restrict__ ev_rows,
    const int *__restrict__ overlap, double *__restrict__ acc) {
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t e0 = off[g], e1 = off[g + 1];   // (absolute: the event arrays are the whole tables)
    const bool apart = overlap == nullptr ? false : overlap[g] == 0;
    if (apart) {
        for (int64_t j = tid; j < L; j += 1024) {
            double a = 0.0;
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t t = ev_lag[e] + j;
                if (t < N) {
                    const float v = __fadd_rn(residual[ev_batch[e] * N + t], ev_rows[e * L + j]);
                    residual[ev_batch[e] * N + t] = v;
                    a += (double)v;
                }
            }
            acc[g * L + j] = a;
        }
        return;
    }
    for (int64_t e = e0; e < e1; ++e) {   // stage in `sparse`, event after event (they may overlap), then move
        float *sp = sparse + ev_batch[e] * N + ev_lag[e];
        const float *row = ev_rows + e * L;
        const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
        for (int64_t s = tid; s < len; s += 1024) sp[s] = __fadd_rn(sp[s], row[s]);
        __syncthreads();
    }
    for (int64_t e = e0; e < e1; ++e) {
        const int64_t base = ev_batch[e] * N + ev_lag[e];
        const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
        for (int64_t s = tid; s < len; s += 1024) {
            residual[base + s] = __fadd_rn(residual[base + s], sparse[base + s]);
            sparse[base + s] = 0.0f;  // an overlapping later event then adds an exact zero
        }
        __syncthreads();
    }
    for (int64_t j = tid; j < L; j += 1024) {
        double a = 0.0;
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t t = ev_lag[e] + j;
            if (t < N) a += (double)residual[ev_batch[e] * N + t];
        }
        acc[g * L + j] = a;
    }
}

__global__ __launch_bounds__(1024) void dictionary_level_subtract_kernel(
    float *residual, float *sparse, int64_t N, int64_t L, const int64_t *__restrict__ off,
    const int64_t *__restrict__ ev_batch, const int64_t *__restrict__ ev_lag, const float *__restrict__ ev_norm,
    const int *__restrict__ overlap, const float *__restrict__ new_atoms) {
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t e0 = off[g], e1 = off[g + 1];
    const float *nw = new_atoms + g * L;
    const bool apart = overlap == nullptr ? false : overlap[g] == 0;
    if (apart) {
        for (int64_t j = tid; j < L; j += 1024) {
            const float a = nw[j];
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t t = ev_lag[e] + j;
                if (t < N) residual[ev_batch[e] * N + t] = __fsub_rn(residual[ev_batch[e] * N + t], __fmul_rn(a, ev_norm[e]));
            }
        }
        return;
    }
    for (int64_t e = e0; e < e1; ++e) {
        float *sp = sparse + ev_batch[e] * N + ev_lag[e];
        const float nrm = ev_norm[e];
        const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
        for (int64_t s = tid; s < len; s += 1024) sp[s] = __fadd_rn(sp[s], __fmul_rn(nw[s], nrm));
        __syncthreads();
    }
    for (int64_t e = e0; e < e1; ++e) {
        const int64_t base = ev_batch[e] * N + ev_lag[e];
        const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
        for (int64_t s = tid; s < len; s += 1024) {
            residual[base + s] = __fsub_rn(residual[base + s], sparse[base + s]);
            sparse[base + s] = 0.0f;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The atom-by-atom loop of dictionary_learning_step (modules/matchingpursuit.py:391-415) in ONE launch of ONE
// workgroup: the loop is sequential by construction (every atom's update reads the residual the earlier
// atoms left) and touches n_g * L samples per atom, so it is latency-, not bandwidth-bound -- as ~10 small
// launches per atom it cost 54 ms at the headline shape, here the whole loop is a few ms.
// Per used atom g (events [off[g], off[g+1]) in selection order), exactly what the reference's dense tensors do:
//   sparse = scatter(rows of g);  residual += sparse            (:395-396; overlapping events sum first)
//   acc[j] = sum_e residual[b_e, lag_e + j]  (fp64, event order; zero beyond N)          (:400-401)
//   new    = acc / (||acc|| + 1e-8)           (unit_norm_kernel's arithmetic)                (:403-404)
//   d[order[g]] = new;  sparse = scatter(new * ||row_e||);  residual -= sparse               (:406-415)
// `sparse` is a zeroed [B, N] scratch; the kernel re-zeroes what it touched.
// ------------------------------------------------------------------------------------------------
// one group (= one used atom) of the loop, by the 1024 threads of a workgroup; ends with a barrier
__device__ __forceinline__ void dictionary_update_group(
    const int64_t g, float *residual, float *sparse, int64_t N, float *d_work, int64_t L,
    const int64_t *__restrict__ order, const int64_t *__restrict__ off,
    const int64_t *__restrict__ ev_batch, const int64_t *__restrict__ ev_lag, const float *__restrict__ ev_rows,
    const float *__restrict__ ev_norm, float eps, const int *__restrict__ overlap, int64_t win_cap, float *nw, float *win,
    float &s_den) {
    const int tid = threadIdx.x;
    {
        const int64_t e0 = off[g], e1 = off[g + 1];
        // An atom none of whose events overlap (overlap[g] == 0: the caller checked -- almost every atom) needs no
        // staging in `sparse` and no event-after-event order: 0 + row == row exactly, so adding the rows straight
        // into the residual, all events at once, is the same arithmetic with 5 barriers per atom instead of 4 n + 3.
        const bool apart = overlap && overlap[g] == 0;
        const int64_t span = (e1 - e0) * L;
        // ... and when the atom's windows fit LDS (n L floats) they stay there between the add-back and the
        // subtraction: one global read and one global write per sample, two memory round trips per atom
        const bool in_lds = apart && span <= win_cap;
        if (in_lds) {
            // work units of 64 samples, dealt to the 16 wavefronts (no per-element index division)
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
            const int chunks = (int)((L + 63) / 64), units = (int)(e1 - e0) * chunks;
            for (int u = wv; u < units; u += 16) {
                const int el = u / chunks, sidx = (u - el * chunks) * 64 + ln;
                const int64_t e = e0 + el, t = ev_lag[e] + sidx;
                if (sidx < L)
                    win[(int64_t)el * L + sidx] = t < N ? __fadd_rn(residual[ev_batch[e] * N + t], ev_rows[e * L + sidx]) : 0.0f;
            }
            __syncthreads();
            for (int64_t j = tid; j < L; j += 1024) {
                double acc = 0.0;
                for (int64_t e = 0; e < e1 - e0; ++e) acc += (double)win[e * L + j];
                nw[j] = (float)acc;
            }
            __syncthreads();
            if (tid < 64) {
                const double ss = row_sum_squares(nw, L, tid);
                if (tid == 0) s_den = __fadd_rn(sqrt_rn_f32((float)ss), eps);
            }
            __syncthreads();
            const float den = s_den;
            float *drow = d_work + order[g] * L;
            for (int64_t j = tid; j < L; j += 1024) {
                const float v = div_rn_f32(nw[j], den);
                nw[j] = v;
                drow[j] = v;
            }
            __syncthreads();
            for (int u = wv; u < units; u += 16) {
                const int el = u / chunks, sidx = (u - el * chunks) * 64 + ln;
                const int64_t e = e0 + el, t = ev_lag[e] + sidx;
                if (sidx < L && t < N)
                    residual[ev_batch[e] * N + t] = __fsub_rn(win[(int64_t)el * L + sidx], __fmul_rn(nw[sidx], ev_norm[e]));
            }
            __syncthreads();  // the next atom reads what this one wrote
            return;
        }
        if (apart) {
            for (int64_t idx = tid; idx < span; idx += 1024) {
                const int64_t e = e0 + idx / L, sidx = idx % L, t = ev_lag[e] + sidx;
                if (t < N) residual[ev_batch[e] * N + t] = __fadd_rn(residual[ev_batch[e] * N + t], ev_rows[e * L + sidx]);
            }
            __syncthreads();
        }
        // add the atom's events back: accumulate in `sparse` (event after event: they may overlap), then move
        for (int64_t e = e0; e < e1 && !apart; ++e) {
            float *sp = sparse + ev_batch[e] * N + ev_lag[e];
            const float *row = ev_rows + e * L;
            const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
            for (int64_t s = tid; s < len; s += 1024) sp[s] = __fadd_rn(sp[s], row[s]);
            __syncthreads();
        }
        for (int64_t e = e0; e < e1 && !apart; ++e) {
            const int64_t base = ev_batch[e] * N + ev_lag[e];
            const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
            for (int64_t s = tid; s < len; s += 1024) {
                residual[base + s] = __fadd_rn(residual[base + s], sparse[base + s]);
                sparse[base + s] = 0.0f;  // an overlapping later event then adds an exact zero
            }
            __syncthreads();
        }
        // sum of the residual windows, then the unit-norm atom
        for (int64_t j = tid; j < L; j += 1024) {
            double acc = 0.0;
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t t = ev_lag[e] + j;
                if (t < N) acc += (double)residual[ev_batch[e] * N + t];
            }
            nw[j] = (float)acc;
        }
        __syncthreads();
        if (tid < 64) {  // the first wavefront: unit_norm_kernel's arithmetic
            const double ss = row_sum_squares(nw, L, tid);
            if (tid == 0) s_den = __fadd_rn(sqrt_rn_f32((float)ss), eps);
        }
        __syncthreads();
        const float den = s_den;
        float *drow = d_work + order[g] * L;
        for (int64_t j = tid; j < L; j += 1024) {
            const float v = div_rn_f32(nw[j], den);
            nw[j] = v;
            drow[j] = v;
        }
        __syncthreads();
        // take the new atom out again with the magnitudes the old instances had
        if (apart) {
            for (int64_t idx = tid; idx < span; idx += 1024) {
                const int64_t e = e0 + idx / L, sidx = idx % L, t = ev_lag[e] + sidx;
                if (t < N)
                    residual[ev_batch[e] * N + t] = __fsub_rn(residual[ev_batch[e] * N + t], __fmul_rn(nw[sidx], ev_norm[e]));
            }
            __syncthreads();
        }
        for (int64_t e = e0; e < e1 && !apart; ++e) {
            float *sp = sparse + ev_batch[e] * N + ev_lag[e];
            const float nrm = ev_norm[e];
            const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
            for (int64_t s = tid; s < len; s += 1024) sp[s] = __fadd_rn(sp[s], __fmul_rn(nw[s], nrm));
            __syncthreads();
        }
        for (int64_t e = e0; e < e1 && !apart; ++e) {
            const int64_t base = ev_batch[e] * N + ev_lag[e];
            const int64_t len = N - ev_lag[e] < L ? N - ev_lag[e] : L;
            for (int64_t s = tid; s < len; s += 1024) {
                residual[base + s] = __fsub_rn(residual[base + s], sparse[base + s]);
                sparse[base + s] = 0.0f;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void dictionary_update_kernel(
    float *residual, float *sparse, int64_t N, float *d_work, int64_t L,
    const int64_t *__restrict__ order, const int64_t *__restrict__ off, int64_t n_groups,
    const int64_t *__restrict__ ev_batch, const int64_t *__restrict__ ev_lag, const float *__restrict__ ev_rows,
    const float *__restrict__ ev_norm, float eps, const int *__restrict__ overlap, int64_t win_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *nw = reinterpret_cast<float *>(smem);  // the new atom, L floats
    float *win = nw + L;                          // win_cap floats: an atom's residual windows
    __shared__ float s_den;
    for (int64_t g = 0; g < n_groups; ++g)
        dictionary_update_group(g, residual, sparse, N, d_work, L, order, off, ev_batch, ev_lag, ev_rows, ev_norm, eps,
                                overlap, win_cap, nw, win, s_den);
}

// The same loop spread over the chip.  Atom g's update reads and writes only the samples under its own events, so
// it depends on an EARLIER atom only if one of that atom's events shares a sample with one of its own.  The host
// (mp_dictionary_levels_host) sorts the atoms into levels -- level(g) = 1 + the highest level among the earlier
// atoms g overlaps -- and the atoms of one level, mutually disjoint and with everything they depend on finished,
// run as one launch, one workgroup each.  Every sample still sees the same operations in the same order as in the
// one-workgroup loop: bit-identical (tests/test_gpu_api.py::test_dictionary_update_levels_are_bit_identical).
// Headline shape: ~400 used atoms in ~34 levels -- 34 short launches instead of a 3.2 ms single-workgroup kernel.
__global__ __launch_bounds__(1024) void dictionary_update_level_kernel(
    float *residual, float *sparse, int64_t N, float *d_work, int64_t L,
    const int64_t *__restrict__ order, const int64_t *__restrict__ off, const int64_t *__restrict__ glist,
    const int64_t *__restrict__ ev_batch, const int64_t *__restrict__ ev_lag, const float *__restrict__ ev_rows,
    const float *__restrict__ ev_norm, float eps, const int *__restrict__ overlap, int64_t win_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *nw = reinterpret_cast<float *>(smem);
    float *win = nw + L;
    __shared__ float s_den;
    dictionary_update_group(glist[blockIdx.x], residual, sparse, N, d_work, L, order, off, ev_batch, ev_lag, ev_rows,
                            ev_norm, eps, overlap, win_cap, nw, win, s_den);
}

// ------------------------------------------------------------------------------------------------
// Workspace carving (all offsets multiples of 256 bytes)
// ------------------------------------------------------------------------------------------------
#include "mpfft.inc"
#include "mplazy.inc"
#include "mplevels.inc"

// bytes of the persistent form's control block + ticket lines + queue (mppersist.inc: PersistCtl <= 256, 512 lines of 64,
// B (K - 1) + 2 entries of 128); carve() reserves them, fft_setup clears them
inline size_t persist_ctl_bytes(int64_t B, int K) { return 256 + 512 * 64 + ((size_t)B * (K - 1) + 2) * 128 + (((size_t)B * 4 + 127) / 128) * 128; }  // (+ one floor per segment: lazy screen)

constexpr int MP_FLAG_FFT_PERSISTENT_BIT = 65536;  // (= MP_FLAG_FFT_PERSISTENT, include/mpcore.h)
thread_local int last_schedule = 0;  // mp_last_schedule(): -1 persistent, 1 one stream, n >= 2 sub-batches

struct Workspace {
    float *res;
    float *img;
    float *drev;  // time-reversed atoms (convolution model)
    u64 *keys;
    int *dirty;
    // FFT path only
    cpx *tw, *pspec, *xspec;
    float *wnorm, *ceps;
    int *cont, *ncont, *overflow;
    float *dscale;  // max atom norm (convolution model: atoms are not unit norm)
    float *subk;    // per-quarter-cell screen maxima [B][cells][SUBCELLS]; only for <= QUARTER_MAX_CELLS
    unsigned *bsum; // per-block (upper, lower) bound summaries [B][NBLK][2]; used for >= FUSED_MIN_CELLS
    u64 *ekeys;
    char *pctl;     // persistent schedule: control block + queue entries (mppersist.inc); nullptr if the shape is not eligible
    cpx *xrec;      // ... and the per-(segment, step) window records
    unsigned *skip; // lazy screen, launch-per-step form: [B][4] tile masks (select -> next screen launch)
    float *lfloor;  // ... and the run's floor per segment [B] (persist_floor_*_kernel after step 0)
    unsigned *work; // ... and the compacted list of (segment, tile) screens the masks leave, tile-major: [B * (NAT + 1)] (count first)
    size_t bytes;
};

// K: steps of the encode (sizes the persistent schedule's per-step records; 0 = that schedule is not carved)
Workspace carve(const Geom &g, int path, char *base, int K = 0) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (size_t)round_up((int64_t)n, 256);
        return o;
    };
    const bool naive = path == MP_PATH_NAIVE;
    size_t o_res = take((size_t)g.B * g.Ns * sizeof(float));
    size_t o_img = take(naive ? 0 : (size_t)g.NAT * g.NCH * g.KC * g.TA * sizeof(float));
    size_t cells = naive ? (size_t)g.A : (size_t)g.NAT;
    size_t o_keys = take((size_t)g.B * g.NBLK * cells * sizeof(u64));
    size_t o_dirty = take((size_t)g.B * 2 * sizeof(int));
    size_t o_drev = take((size_t)g.A * g.L * sizeof(float));
    w.tw = w.pspec = w.xspec = nullptr;
    w.wnorm = w.ceps = nullptr;
    w.cont = w.ncont = w.overflow = nullptr;
    w.ekeys = nullptr;
    w.dscale = nullptr;
    w.subk = nullptr;
    w.bsum = nullptr;
    w.pctl = nullptr;
    w.xrec = nullptr;
    w.skip = nullptr;
    w.lfloor = nullptr;
    w.work = nullptr;
    if (path == MP_PATH_FFT) {
        FftGeom f;
        if (make_fft_geom(g, &f)) {
            size_t o_tw = take((size_t)f.M * 2 * sizeof(cpx));  // (split transforms: half-size + full-size table)
            size_t o_ps = take((size_t)g.NAT * f.NPT * f.M * sizeof(cpx));
            size_t o_xs = take((size_t)g.B * f.NW * f.M * sizeof(cpx));
            size_t o_wn = take((size_t)g.B * f.NW * sizeof(float));
            size_t o_ce = take((size_t)g.B * g.NBLK * g.NAT * sizeof(float));
            size_t o_co = take((size_t)g.B * MAXCONT * sizeof(int));
            size_t o_nc = take((size_t)g.B * sizeof(int));
            size_t o_ov = take((size_t)g.B * sizeof(int));
            size_t o_ek = take((size_t)g.B * (MAXCONT + 1) * sizeof(u64));
            size_t o_ds = take(256);
            size_t o_bs = take((size_t)g.B * g.NBLK * 2 * sizeof(unsigned));
            // quarter maxima: for the quarter-cell select kernel (<= QUARTER_MAX_CELLS) and for the persistent form, whose select
            // reads them cell by cell (1024- to 4096-point transforms, <= PERSIST_MAX_CELLS: larger dictionaries / longer segments)
            const bool quarters = (int64_t)g.NBLK * g.NAT <= QUARTER_MAX_CELLS ||
                                  ((int64_t)g.NBLK * g.NAT <= PERSIST_MAX_CELLS && !f.split && f.logM >= 10 && f.logM <= 12 && g.NAT <= 128);
            size_t o_sk = take(quarters ? (size_t)g.B * g.NBLK * g.NAT * SUBCELLS * sizeof(float) : 0);
            size_t o_lz = take((size_t)g.B * 4 * sizeof(unsigned));
            size_t o_lf = take((size_t)g.B * sizeof(float));
            size_t o_wk = take(g.NAT <= 128 ? (size_t)g.B * (g.NAT + 1) * sizeof(unsigned) : 0);
            w.tw = reinterpret_cast<cpx *>(base + o_tw);
            w.pspec = reinterpret_cast<cpx *>(base + o_ps);
            w.xspec = reinterpret_cast<cpx *>(base + o_xs);
            w.wnorm = reinterpret_cast<float *>(base + o_wn);
            w.ceps = reinterpret_cast<float *>(base + o_ce);
            w.cont = reinterpret_cast<int *>(base + o_co);
            w.ncont = reinterpret_cast<int *>(base + o_nc);
            w.overflow = reinterpret_cast<int *>(base + o_ov);
            w.ekeys = reinterpret_cast<u64 *>(base + o_ek);
            w.dscale = reinterpret_cast<float *>(base + o_ds);
            w.bsum = reinterpret_cast<unsigned *>(base + o_bs);
            if (quarters) w.subk = reinterpret_cast<float *>(base + o_sk);
            w.skip = reinterpret_cast<unsigned *>(base + o_lz);
            w.lfloor = reinterpret_cast<float *>(base + o_lf);
            if (g.NAT <= 128) w.work = reinterpret_cast<unsigned *>(base + o_wk);
            // persistent schedule (mppersist.inc): queue + one window record per (segment, step >= 2)
            // (one write-once window record per segment and step: not for batches whose records would pass 2 GiB --
            //  those run launch per step)
            const size_t xrec_bytes = (size_t)g.B * (K > 2 ? K - 2 : 0) * ((size_t)f.M + 16) * sizeof(cpx);
            if (quarters && !f.split && f.logM >= 10 && f.logM <= 12 && K >= 2 && xrec_bytes <= ((size_t)2 << 30)) {
                const size_t qn = (size_t)g.B * (K - 1) + 2;
                size_t o_pc = take(persist_ctl_bytes(g.B, K));  // PersistCtl + ticket lines + PersistEntry[qn] (mppersist.inc)
                size_t o_xr = take(xrec_bytes);
                w.pctl = base + o_pc;
                w.xrec = reinterpret_cast<cpx *>(base + o_xr);
            }
        }
    }
    w.res = reinterpret_cast<float *>(base + o_res);
    w.img = reinterpret_cast<float *>(base + o_img);
    w.keys = reinterpret_cast<u64 *>(base + o_keys);
    w.dirty = reinterpret_cast<int *>(base + o_dirty);
    w.drev = reinterpret_cast<float *>(base + o_drev);
    w.bytes = off;
    return w;
}

int check_shape(int64_t B, int64_t N, int64_t A, int64_t L, int K) {
    if (B < 0 || N <= 0 || A <= 0 || L <= 0 || K < 0) return fail(MP_ERR_ARG, "bad shape%s");
    if ((unsigned long long)A * (unsigned long long)N > 0xfffffffeull)
        return fail(MP_ERR_ARG, "A * N must be < 2^32 - 1%s");
    if (L > (1 << 24)) return fail(MP_ERR_ARG, "atom size too large%s");
    return MP_OK;
}

// image + window + the pad the pipelined loop's last (unused) prefetch may touch
size_t lds_bytes(const Geom &g) {
    return ((size_t)g.KC * g.TA + (size_t)WAVES * (g.KC + LAGS_PER_WAVE + 16)) * sizeof(float) +
           (size_t)(g.TA + 64) * 16;
}

constexpr int MAX_DEVICES = 16;
int current_device() {  // index for the per-device caches below (a device beyond the table shares slot 0: re-set, never wrong)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}

template <int TA, bool STORE_FM, bool DMA>
int launch_correlate_t(const Geom &g, const Workspace &w, const int *dirty, float *fm, hipStream_t st) {
    auto kern = correlate_mfma_kernel<TA, STORE_FM, DMA>;
    const size_t lds = lds_bytes(g);
    static thread_local size_t configured_dev[MAX_DEVICES] = {0};  // the attribute is per device (and set per thread: cheap)
    size_t &configured = configured_dev[current_device()];
    if (lds > configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const int stride = dirty ? g.MAXC : g.NBLK;
    const int64_t tasks = g.B * (int64_t)stride;
    dim3 grid((unsigned)((tasks + WAVES - 1) / WAVES), g.NAT);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, w.res, w.img, dirty, w.keys, fm, g.N, g.A, g.Ns, g.B,
                       g.NBLK, g.NAT, g.KC, g.NCH, stride);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

// 1-D grid for a launch whose workgroups are dealt to the XCDs by hand (workgroup i runs on XCD i mod 8; mpfft.inc:
// fft_screen_kernel): `units` groups of `per` workgroups each, a whole group on one XCD.
inline unsigned xcd_grid(int64_t per, int64_t units) { return (unsigned)(8 * per * ((units + 7) / 8)); }

int num_cus() {
    static std::atomic<int> cus[MAX_DEVICES];  // zero-initialised; racing first calls compute the same value
    std::atomic<int> &slot = cus[current_device()];
    int n = slot.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        slot.store(n, std::memory_order_relaxed);
    }
    return n;
}

template <int TA, bool DMA>
int launch_persistent_t(const Geom &g, const Workspace &w, const int *dirty, int stagger, hipStream_t st) {
    auto kern = correlate_persistent_kernel<TA, DMA>;
    const size_t lds = lds_bytes(g);
    static thread_local size_t configured_dev[MAX_DEVICES] = {0};
    size_t &configured = configured_dev[current_device()];
    if (lds > configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const int stride = dirty ? g.MAXC : g.NBLK;
    const int64_t LT = g.B * (int64_t)stride;
    const int64_t T = LT * g.NAT;
    const int wgs_per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
    int64_t grid = (int64_t)num_cus() * (wgs_per_cu > 2 ? 2 : wgs_per_cu);
    if (grid > (T + WAVES - 1) / WAVES) grid = (T + WAVES - 1) / WAVES;
    const int64_t per = (T + grid - 1) / grid;
    grid = (T + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, w.res, w.img, dirty, w.keys, g.N, g.A,
                       g.Ns, g.NBLK, g.NAT, g.KC, stride, LT, per, stagger, (float *)nullptr);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

// The LCN schedule's correlate (TA = 32): cells into the cell-order map, no keys.  One k-chunk: the persistent form
// (image staged once per workgroup, windows prefetched); longer atoms: one wavefront per cell, chunk by chunk.
int launch_correlate_cellmap(const Geom &g, const Workspace &w, const int *dirty, float *cmap, hipStream_t st) {
    const size_t lds = lds_bytes(g);
    const int stride = dirty ? g.MAXC : g.NBLK;
    static thread_local size_t configured_dev[MAX_DEVICES][2] = {{0}};
    if (g.NCH == 1) {
        auto kern = correlate_persistent_kernel<32, true, true>;
        size_t &configured = configured_dev[current_device()][0];
        if (lds > configured) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = lds;
        }
        const int64_t LT = g.B * (int64_t)stride;
        const int64_t T = LT * g.NAT;
        const int wgs_per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
        int64_t grid = (int64_t)num_cus() * (wgs_per_cu > 2 ? 2 : wgs_per_cu);
        if (grid > (T + WAVES - 1) / WAVES) grid = (T + WAVES - 1) / WAVES;
        const int64_t per = (T + grid - 1) / grid;
        grid = (T + per - 1) / per;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, w.res, w.img, dirty, w.keys, g.N, g.A, g.Ns, g.NBLK,
                           g.NAT, g.KC, stride, LT, per, dirty ? 2 : 0, cmap);
    } else {
        auto kern = correlate_mfma_kernel<32, true, true, true>;
        size_t &configured = configured_dev[current_device()][1];
        if (lds > configured) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = lds;
        }
        const int64_t tasks = g.B * (int64_t)stride;
        hipLaunchKernelGGL(kern, dim3((unsigned)((tasks + WAVES - 1) / WAVES), g.NAT), dim3(256), lds, st, w.res, w.img, dirty,
                           w.keys, cmap, g.N, g.A, g.Ns, g.B, g.NBLK, g.NAT, g.KC, g.NCH, stride);
    }
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

template <bool STORE_FM>
int launch_correlate(const Geom &g, const Workspace &w, const int *dirty, float *fm, int flags,
                     hipStream_t st) {
    const bool dma = !(flags & MP_FLAG_NO_DMA);
    if (!STORE_FM && g.NCH == 1 && !(flags & MP_FLAG_NO_PERSISTENT)) {
        // s_sleep(127) = 8128 cycles; half a cell at one wave per SIMD is ~8k * KC/512 * TA/32 ... cycles
        // measured (scripts/ab_flags.py): -3 % time on incremental launches (all workgroups start
        // together and run only ~8 cells each), no effect on full passes (hundreds of cells drift apart)
        const int stagger = (dirty && !(flags & MP_FLAG_NO_STAGGER)) ? (g.TA == 32 ? 2 : 4) : 0;
        if (g.TA == 32)
            return dma ? launch_persistent_t<32, true>(g, w, dirty, stagger, st)
                       : launch_persistent_t<32, false>(g, w, dirty, stagger, st);
        return dma ? launch_persistent_t<64, true>(g, w, dirty, stagger, st)
                   : launch_persistent_t<64, false>(g, w, dirty, stagger, st);
    }
    if (g.TA == 32)
        return dma ? launch_correlate_t<32, STORE_FM, true>(g, w, dirty, fm, st)
                   : launch_correlate_t<32, STORE_FM, false>(g, w, dirty, fm, st);
    return dma ? launch_correlate_t<64, STORE_FM, true>(g, w, dirty, fm, st)
               : launch_correlate_t<64, STORE_FM, false>(g, w, dirty, fm, st);
}

// Default tile: 32 atoms (64 KiB image, two workgroups per CU so one stages while the other computes);
// measured faster than 64 on MI355X for both the full and the incremental pass.
int tile_atoms(int flags) {
    if (flags & MP_FLAG_TA64) return 64;
    return 32;
}

int launch_naive(const Geom &g, const Workspace &w, const float *du, const int *dirty, int nblk_grid,
                 float *fm, hipStream_t st) {
    dim3 grid(nblk_grid, (unsigned)g.A, (unsigned)g.B);
    hipLaunchKernelGGL(correlate_naive_kernel, grid, dim3(64), 0, st, w.res, du, dirty, w.keys, fm, g.N,
                       g.A, g.L, g.Ns, g.NBLK);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

// How a selected event changes the residual row.  Matching pursuit (modules/matchingpursuit.py:305,:328):
// row[lag + s] -= d[atom][s] * gain.  Convolution model (mp.py:58-64): the feature map is the CONVOLUTION of
// the residual with the raw atoms -- computed here as the correlation of the row (L - 1 leading zeros, then the
// signal) with the time-reversed atoms -- and the update puts the FORWARD atom at output sample `lag`
// (row position lag + L - 1) scaled by gain^2.
struct Rule {
    const float *du_sub;  // atoms used by the subtraction
    int shift;            // row offset of the subtracted atom relative to the lag
    int square;           // scale by gain^2 instead of gain
    int64_t lead;         // leading zeros of every residual row
};

int stage_inputs(const Geom &g, const Workspace &w, int path, const float *signal, const float *du,
                 int64_t lead, hipStream_t st, bool want_image = true) {
    {
        dim3 grid((unsigned)((g.Ns + 255) / 256 < 1024 ? (g.Ns + 255) / 256 : 1024), (unsigned)g.B);
        hipLaunchKernelGGL(init_residual_kernel, grid, dim3(256), 0, st, signal, g.N, g.Ns, lead, w.res);
        HIP_TRY(hipGetLastError());
    }
    if (path != MP_PATH_NAIVE && want_image) {   // the MFMA kernels' dictionary image
        int64_t total = (int64_t)g.NAT * g.NCH * g.KC * g.TA;
        unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(dict_image_kernel, dim3(blocks), dim3(256), 0, st, du, g.A, g.L, g.TA, g.KC,
                           g.NCH, g.NAT, w.img);
        HIP_TRY(hipGetLastError());
    }
    return MP_OK;
}

// ---- FFT path driver ---------------------------------------------------------------------------
Geom make_geom_for(int64_t B, int64_t N, int64_t A, int64_t L, int path, int flags) {
    if (path == MP_PATH_FFT) {
        Geom g0 = make_geom(B, N, A, L, 32);
        FftGeom f;
        if (make_fft_geom(g0, &f)) return make_geom(B, N, A, L, 32, f.M);
        return g0;
    }
    return make_geom(B, N, A, L, tile_atoms(flags));
}

#define MP_FFT_DISPATCH(LOGM, CALL)                                  \
    switch (LOGM) {                                                  \
        case 8: { constexpr int LG = 8; CALL; } break;               \
        case 9: { constexpr int LG = 9; CALL; } break;               \
        case 10: { constexpr int LG = 10; CALL; } break;             \
        case 11: { constexpr int LG = 11; CALL; } break;             \
        case 12: { constexpr int LG = 12; CALL; } break;             \
        case 13: { constexpr int LG = 13; CALL; } break;             \
        case 14: { constexpr int LG = 14; CALL; } break;             \
        default: return fail(MP_ERR_UNSUPPORTED, "FFT size out of range%s"); \
    }

// split transforms (FftGeom::split = log2 of the parts)
#define MP_SPLIT_DISPATCH(SPLIT, CALL)                               \
    switch (SPLIT) {                                                 \
        case 1: { constexpr int Q = 2; CALL; } break;                \
        case 2: { constexpr int Q = 4; CALL; } break;                \
        default: return fail(MP_ERR_UNSUPPORTED, "FFT split out of range%s"); \
    }

template <typename K>
int fft_lds_attr(K kern, size_t bytes) {
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MP_OK;
}

// Tuning hooks (mp_tune): plain process-wide defaults, read with relaxed atomics on the encode path.  The number of
// sub-batches can also be given per call (MP_FLAG_GROUPS(n)), which is what a caller that wants a particular count
// while other threads encode should use.
std::atomic<int> screen_pps_override{0};
std::atomic<int> overlap_groups{4};   // sub-batches when the batch is split over forked streams
std::atomic<float> tau_override{0.f}; // > 0: replaces the model below
std::atomic<int> audit_on{0};         // debug: after every screen recompute the screened cells exactly (mp_audit_read)

// The screen's error bound.  eps(window) must cover |screen value - fp32 fma chain| at every
// (atom, lag) of the window:
//   * the chain's own rounding, RIGOROUSLY: acc_k = (acc_{k-1} + r_k d_k)(1 + delta_k), |delta_k| <= u = 2^-24, so
//     |chain - exact| <= u (1 + u)^L sum_k |partial_k|, and sum_k |partial_k| <= sum_j (L - j) |r_j| |d_j|
//     <= ||r[t .. t+L)|| W_a <= ||window|| W_a (Cauchy-Schwarz), W_a = sqrt(sum_j (L - j)^2 d_a[j]^2): 0.58 L for
//     an atom of even energy, L at worst.  Same-sign atoms on a DC offset are the inputs that come within a small
//     factor of it (partial sums grow linearly; with constant increments the roundings share a sign inside a
//     binade); random data stay near sqrt(L) u.
//   * the three fp32 transforms and the spectrum product, MODELLED: c log2(M) u ||window|| ||d_a|| -- the rounding
//     error of a Stockham FFT grows with the number of butterfly levels and is relative to the RMS of the result,
//     itself <= ||window|| ||d_a||.  With c = 4 this term ALONE exceeds the largest |screen - chain| the audit mode
//     (MP_TUNE_AUDIT) has seen on random, planted, DC-offset / same-sign, transient and 1e-30 / 1e18-amplitude
//     inputs for L = 16 .. 8192 (scripts/screen_audit.py: at most 41 u ||window|| at L = 8192, 24 u at L = 512), and
//     the whole bound was used to at most 0.09 (tests/test_gpu_parity.py::test_screen_error_bound_audit).
//   eps = u ||window|| max_a (1.001 W_a + 4 log2(M) ||d_a||):  the kernels multiply `tau` (= u here) with
//   wnorm = ||window|| * dscale, dscale = the dictionary factor (max_row_norm_kernel).
// mp_tune(MP_TUNE_TAU, x > 0) replaces it by the constant form eps = x ||window|| max_a ||d_a|| (round 1 used 2e-5).
constexpr float FFT_TAU_C = 4.0f;
struct TauModel { float tau, chain_w, fft_w; };
TauModel fft_tau(int logM) {
    const float o = tau_override.load(std::memory_order_relaxed);
    if (o > 0.f) return TauModel{o, 0.f, 1.f};
    return TauModel{5.9604645e-8f, 1.001f, FFT_TAU_C * (float)logM};
}

// Zero `bytes` (a multiple of 4, 4-byte aligned) on the stream, as a kernel: a captured encode then holds kernel nodes
// only (a replayed hipGraph whose persistent launch depends on the cleared queue must not depend on how the runtime
// orders its memset nodes -- seen: correct on the first replay, into fresh zero pages, stale state on the second).
struct ClearList {   // up to 8 ranges cleared by ONE launch
    unsigned *p[8];
    size_t n[8];      // words
    int count = 0;
    void add(void *ptr, size_t bytes) {
        if (ptr && bytes >= 4 && count < 8) { p[count] = static_cast<unsigned *>(ptr); n[count] = bytes / 4; ++count; }
    }
};
__global__ void clear_words_kernel(ClearList c) {
    for (int r = 0; r < c.count; ++r) {
        unsigned *p = c.p[r];
        const size_t n = c.n[r];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
    }
}
std::atomic<int> clear_with_memset{0};   // debug hook (MP_TUNE_CLEAR_MEMSET): hipMemsetAsync per range, as round 2 first had it
int clear_async(const ClearList &c, hipStream_t st) {
    if (clear_with_memset.load(std::memory_order_relaxed)) {   // (scripts/graph_memset_repro.py: what a capture makes of these)
        for (int r = 0; r < c.count; ++r) HIP_TRY(hipMemsetAsync(c.p[r], 0, c.n[r] * 4, st));
        return MP_OK;
    }
    size_t most = 0;
    for (int r = 0; r < c.count; ++r) most = std::max(most, c.n[r]);
    if (!most) return MP_OK;
    const size_t blocks = std::min<size_t>((most + 255) / 256, 4096);
    hipLaunchKernelGGL(clear_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, c);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

// once per encode, whole batch, on the caller's stream: twiddles, pair spectra, cleared keys / flags
int fft_setup(const Geom &g, const Workspace &w, const float *du, int flags, int K, hipStream_t st) {
    FftGeom f;
    if (!make_fft_geom(g, &f))
        return fail(MP_ERR_UNSUPPORTED, "MP_PATH_FFT: atoms longer than 21782 samples need MP_PATH_INCREMENTAL%s");
    if (g.B > 65535) return fail(MP_ERR_ARG, "MP_PATH_FFT: batch > 65535 per call%s");
    const size_t lds = (size_t)(f.M >> f.split) * sizeof(cpx);
    const int64_t n_cells = (int64_t)g.NBLK * g.NAT;
    const int npairs = g.NAT * f.NPT;
    int rc;
    ClearList cl;   // everything the encode needs zeroed, in one launch
    cl.add(w.overflow, (size_t)g.B * sizeof(int));
    cl.add(w.ekeys, (size_t)g.B * (MAXCONT + 1) * sizeof(u64));
    cl.add(w.bsum, (size_t)g.B * g.NBLK * 2 * sizeof(unsigned));
    cl.add(w.keys, (size_t)g.B * n_cells * sizeof(u64));
    cl.add(w.dscale, sizeof(float));
    cl.add(w.skip, (size_t)g.B * 4 * sizeof(unsigned));
    if (w.pctl && K >= 2) cl.add(w.pctl, persist_ctl_bytes(g.B, K));   // the persistent form's control block and queue
    if ((rc = clear_async(cl, st))) return rc;
    if (f.split) {  // long atoms: two or four 2^14-point transforms per M-point transform (mpfft.inc)
        constexpr int LH = SPLIT_LOGH;
        const int H = f.M >> f.split;
        hipLaunchKernelGGL(fft_twiddle_kernel, dim3((H + 255) / 256), dim3(256), 0, st, w.tw, H);
        hipLaunchKernelGGL(fft_twiddle_kernel, dim3((f.M + 255) / 256), dim3(256), 0, st, w.tw + H, f.M);
        HIP_TRY(hipGetLastError());
        MP_SPLIT_DISPATCH(f.split, {
            if ((rc = fft_lds_attr(fft_dict_split_kernel<LH, Q>, lds))) return rc;
            if ((rc = fft_lds_attr(fft_window_split_kernel<LH, Q>, lds))) return rc;
            if ((rc = fft_lds_attr(fft_correlate_split_kernel<LH, Q>, lds))) return rc;
            hipLaunchKernelGGL((fft_dict_split_kernel<LH, Q>), dim3(npairs, Q), dim3(256), lds, st, du, g.A, g.L, w.tw, w.pspec);
        })
        HIP_TRY(hipGetLastError());
        const size_t lds_ref_s = lds_bytes(g);
        if (!(flags & MP_FLAG_NO_DMA)) { if ((rc = fft_lds_attr(fft_refine_kernel<true>, lds_ref_s))) return rc; }
        else { if ((rc = fft_lds_attr(fft_refine_kernel<false>, lds_ref_s))) return rc; }
        return MP_OK;
    }
    hipLaunchKernelGGL(fft_twiddle_kernel, dim3((f.M + 255) / 256), dim3(256), 0, st, w.tw, f.M);
    HIP_TRY(hipGetLastError());
    MP_FFT_DISPATCH(f.logM, {
        if ((rc = fft_lds_attr(fft_dict_kernel<LG>, lds))) return rc;
        if ((rc = fft_lds_attr(fft_window_kernel<LG>, lds))) return rc;
        if ((rc = fft_lds_attr(fft_correlate_kernel<LG>, lds))) return rc;
        hipLaunchKernelGGL(fft_dict_kernel<LG>, dim3(npairs), dim3(256), lds, st, du, g.A, g.L, w.tw, w.pspec);
    })
    HIP_TRY(hipGetLastError());
    const bool dma = !(flags & MP_FLAG_NO_DMA);
    const size_t lds_ref = lds_bytes(g);
    if (dma) { if ((rc = fft_lds_attr(fft_refine_kernel<true>, lds_ref))) return rc; }
    else { if ((rc = fft_lds_attr(fft_refine_kernel<false>, lds_ref))) return rc; }
    return MP_OK;
}

// one matching-pursuit step of a (sub-)batch on stream st
int fft_iteration(const Geom &g, const Workspace &w, const float *du, int K, int k, int flags, int64_t *out_atom,
                  int64_t *out_lag, float *out_gain, const Rule &rule, hipStream_t st, const LazyArgs &lz = LazyArgs()) {
    FftGeom f;
    if (!make_fft_geom(g, &f)) return fail(MP_ERR_UNSUPPORTED, "MP_PATH_FFT: atom too long%s");
    const float tau = fft_tau(f.logM).tau;
    // split transforms: the four-kernel form between screens (the window kernel makes the next spectrum)
    if (f.split) flags = (flags | MP_FLAG_FFT_UNFUSED) & ~(MP_FLAG_FFT_FUSED | MP_FLAG_FFT_QUARTER);
    const size_t lds = (size_t)(f.M >> f.split) * sizeof(cpx);
    const int64_t n_cells = (int64_t)g.NBLK * g.NAT;
    const bool dma = !(flags & MP_FLAG_NO_DMA);
    const size_t lds_ref = lds_bytes(g);
    int rc;
    // Between two screens: select-A, refine and select-B as three small kernels (the refinement of a cell
    // spread over eight workgroups), or ONE kernel per segment when a segment has so many cells that
    // scanning them dominates (measured: config-4 shape 190 vs 315 us per step; headline shape 35 vs 26 us).
    // MP_FLAG_FFT_FUSED / MP_FLAG_FFT_UNFUSED force either form.  A team of workgroups per segment inside
    // one kernel was tried and dropped: agent-scope fences between its members cost 3-13 us each.
    // (step 0 of the persistent form for segments of more than QUARTER_MAX_CELLS cells: the quarter-cell select kernel keeps
    //  a whole segment's keys in registers and stops there -- the fused select takes its place, behind a screen that
    //  still leaves the quarter maxima and block summaries the launch's selects read)
    const bool pstep0_big = (flags & MP_FLAG_FFT_PERSISTENT_BIT) && w.subk && n_cells > QUARTER_MAX_CELLS && f.logM >= 10;
    const bool fused = pstep0_big || (!(flags & MP_FLAG_REFINE_MFMA) && !(flags & MP_FLAG_FFT_UNFUSED) &&
                       ((flags & MP_FLAG_FFT_FUSED) || n_cells >= FUSED_MIN_CELLS ||
                        (n_cells > QUARTER_MAX_CELLS && g.L <= 512)));  // mid sizes: only while a cell's chains are short
    // both forms leave the next step's window spectrum behind when the screen's register transform exists
    // for this size (the stand-alone window kernel then runs before the first step only)
    // ... or, for small segments, ONE kernel that refines only a quarter of a contender cell (needs the screen's
    // per-quarter maxima, hence its register transform at pps = 4)
    // Measured at the headline shape: with the batch on one stream the two-launch form (screen at pps = 8) is
    // 1-2 % ahead, with two sub-batches on forked streams the quarter kernel is 3 % ahead -- so each gets the
    // schedule it is better under (MP_FLAG_FFT_QUARTER / MP_FLAG_FFT_NO_QUARTER force either).
    const bool two_launch_ok = f.logM <= 12 && n_cells <= 16384;
    const bool quarter = w.subk && f.logM >= 10 && n_cells <= QUARTER_MAX_CELLS &&
                         !(flags & (MP_FLAG_REFINE_MFMA | MP_FLAG_FFT_UNFUSED | MP_FLAG_FFT_FUSED | MP_FLAG_FFT_SIMPLE |
                                    MP_FLAG_FFT_NO_QUARTER)) &&
                         ((flags & MP_FLAG_FFT_QUARTER) || !((flags & MP_FLAG_INTERNAL_ONE_STREAM) && two_launch_ok));
    const bool b_tail = !fused && !quarter && !(flags & MP_FLAG_FFT_SIMPLE) && f.logM >= 10 && f.logM <= 12;
    const bool fused_tail = (fused && f.logM >= 10) || b_tail || quarter;
    // (kept up to date by fft_screen_kernel only: not with the plain radix-4 screen)
    // (... and whenever the fused select is to run the lazy screen, which decides on them: any size then -- that is how the
    //  small shapes of the parity suite and the fuzz sweep reach it, with MP_FLAG_FFT_FUSED and a coherence table)
    unsigned *bsum = (fused && (n_cells > QUARTER_MAX_CELLS || lz.mu) && f.logM >= 10 && !(flags & MP_FLAG_FFT_SIMPLE)) ? w.bsum : nullptr;
    if ((flags & MP_FLAG_FFT_PERSISTENT_BIT) && (quarter || pstep0_big)) bsum = w.bsum;  // step 0 of the persistent schedule builds them too
    // The lazy screen on this form: the fused select decides, per segment, which tiles' dirty cells keep
    // their widened bounds -- a [B][4] mask the NEXT screen launch's workgroups leave on (config-4 shape: 8192-point
    // transforms, where the launch-per-step form is work-bound on its screens).  From step 1's select on: the floor it
    // needs comes from the summaries step 0 leaves (below).
    const bool lazy_form = lz.mu != nullptr && fused && !quarter && bsum != nullptr && w.skip != nullptr && !f.split &&
                           f.logM >= 10 && g.NAT <= 128 && !(flags & (MP_FLAG_FFT_SIMPLE | MP_FLAG_INTERNAL_COHERENCE)) &&
                           !audit_on.load(std::memory_order_relaxed);
    {
        const int *dirty = k == 0 ? nullptr : w.dirty;
        const int nw = k == 0 ? f.NW : 1;
        g_prof.begin(PROF_SELECT, st);
        if (f.split) {
            MP_SPLIT_DISPATCH(f.split, {
                hipLaunchKernelGGL((fft_window_split_kernel<SPLIT_LOGH, Q>), dim3(nw, (unsigned)g.B, Q), dim3(1024), lds, st, w.res,
                                   g.Ns, dirty, w.tw, w.xspec, w.wnorm, f.V, f.NW, (const float *)w.dscale);
            })
        } else if (k == 0 || !fused_tail) {
            MP_FFT_DISPATCH(f.logM, {
                hipLaunchKernelGGL(fft_window_kernel<LG>, dim3(nw, (unsigned)g.B), dim3(256), lds, st, w.res, g.Ns,
                                   dirty, w.tw, w.xspec, w.wnorm, f.V, f.NW,
                                   (const float *)w.dscale);
            })
        }
        g_prof.end(st);
        g_prof.begin(k == 0 ? PROF_CORR_FULL : PROF_CORR_INC, st);
        if (f.split && !(flags & MP_FLAG_FFT_SIMPLE)) {
            using C = ScreenCfg<SPLIT_LOGH>;
            int pps = 16 / C::SLOTS;
            const int64_t tasks = (int64_t)nw * g.NAT * g.B;
            while (pps > 1 && tasks * (16 / (C::SLOTS * pps)) < 8 * (int64_t)num_cus()) pps >>= 1;
            if (const int po = screen_pps_override.load(std::memory_order_relaxed); po > 0 && 16 % (C::SLOTS * po) == 0) pps = po;
            const size_t lds_s = ((size_t)C::SLOTS * C::M + C::M / 64 + 64) * sizeof(cpx);
            const bool seg_fast = (size_t)g.NAT * f.NPT * f.M * sizeof(cpx) > (size_t)16 << 20;
            const unsigned gwp = nw * (16 / (C::SLOTS * pps));
            const dim3 grid = seg_fast ? dim3(xcd_grid(g.B, (int64_t)gwp * g.NAT)) : dim3(gwp, g.NAT, (unsigned)g.B);
            auto kern = f.split == 2 ? fft_screen_split4_kernel<SPLIT_LOGH> : fft_screen_split_kernel<SPLIT_LOGH>;
            if ((rc = fft_lds_attr(kern, lds_s))) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(C::WG), lds_s, st, w.xspec, w.pspec,
                               w.tw, dirty, w.wnorm, w.keys, w.ceps, g.N, g.A, g.NBLK, g.NAT, f.V, f.NW, tau, pps,
                               seg_fast ? (int)g.B : 0);
        } else if (f.split) {
            MP_SPLIT_DISPATCH(f.split, {
                hipLaunchKernelGGL((fft_correlate_split_kernel<SPLIT_LOGH, Q>), dim3(Q * nw, g.NAT, (unsigned)g.B), dim3(256), lds, st,
                                   w.xspec, w.pspec, w.tw, dirty, w.wnorm, w.keys, w.ceps, g.N, g.A, g.NBLK, g.NAT, f.V,
                                   f.NW, tau);
            })
        } else
        MP_FFT_DISPATCH(f.logM, {
            if (LG >= 10 && !(flags & MP_FLAG_FFT_SIMPLE)) {
                constexpr int LS = LG >= 10 ? LG : 10;  // (the branch is dead for smaller LG)
                using C = ScreenCfg<LS>;
                // pairs per slot: fewest workgroups that still oversubscribe the machine ~8x
                int pps = 16 / C::SLOTS;
                const int64_t tasks = (int64_t)nw * g.NAT * g.B;
                while (pps > 1 && tasks * (16 / (C::SLOTS * pps)) < 8 * (int64_t)num_cus()) pps >>= 1;
                if (const int po = screen_pps_override.load(std::memory_order_relaxed); po > 0 && 16 % (C::SLOTS * po) == 0) pps = po;
                if (quarter || pstep0_big) pps = 4;  // one slot = one quarter of a tile
                float *subk = (quarter || pstep0_big) ? w.subk : nullptr;
                const size_t lds_s = ((size_t)C::SLOTS * C::M + C::M / 64 + 64 + SCREEN_TAB_CPX) * sizeof(cpx);
                // pair spectra that cannot stay in the L2s: segment-fastest grid order (see the kernel)
                const bool seg_fast = (size_t)g.NAT * f.NPT * f.M * sizeof(cpx) > (size_t)16 << 20;
                const unsigned gwp = nw * (16 / (C::SLOTS * pps));
                // (masked launches, lazy form: the select's masks compacted into a work list -- room for every (segment,
                //  tile) entry, chunks of 16 entries x parts dealt to the XCDs; mplazy.inc: lazy_compact_kernel)
                const bool listed = lazy_form && k >= 2 && w.work != nullptr && g.B <= 65535 &&
                                    lazy_compact.load(std::memory_order_relaxed) != 0;
                const dim3 grid = listed ? dim3(xcd_grid(16 * (int64_t)gwp, ((int64_t)g.B * g.NAT + 15) / 16))
                                  : seg_fast ? dim3(xcd_grid(g.B, (int64_t)gwp * g.NAT)) : dim3(gwp, g.NAT, (unsigned)g.B);
                if (flags & MP_FLAG_INTERNAL_COHERENCE) {   // mp_coherence_f32: cell maxima of |correlation|, nothing after the screen
                    constexpr int LC = LS <= 13 ? LS : 13;  // (coherence_geom: 1024- to 8192-point transforms)
                    auto kabs = (fft_screen_kernel<LC, true>);
                    if ((rc = fft_lds_attr(kabs, lds_s))) return rc;
                    hipLaunchKernelGGL(kabs, grid, dim3(C::WG), lds_s, st, w.xspec, w.pspec, w.tw, dirty, w.wnorm, w.keys, w.ceps,
                                       g.N, g.A, g.NBLK, g.NAT, f.V, f.NW, tau, pps, seg_fast ? (int)g.B : 0, (float *)nullptr,
                                       (unsigned *)nullptr, (const unsigned *)nullptr, (const unsigned *)nullptr);
                    HIP_TRY(hipGetLastError());
                    g_prof.end(st);
                    return MP_OK;
                }
                if ((rc = fft_lds_attr(fft_screen_kernel<LS>, lds_s))) return rc;
                hipLaunchKernelGGL(fft_screen_kernel<LS>, grid, dim3(C::WG), lds_s, st, w.xspec, w.pspec, w.tw, dirty,
                                   w.wnorm, w.keys, w.ceps, g.N, g.A, g.NBLK, g.NAT, f.V, f.NW, tau, pps, seg_fast ? (int)g.B : 0, subk, bsum,
                                   (const unsigned *)(lazy_form && k >= 2 ? w.skip : nullptr),
                                   (const unsigned *)(listed ? w.work : nullptr));
            } else {
                hipLaunchKernelGGL(fft_correlate_kernel<LG>, dim3(nw, g.NAT, (unsigned)g.B), dim3(256), lds, st,
                                   w.xspec, w.pspec, w.tw, dirty, w.wnorm, w.keys, w.ceps, g.N, g.A, g.NBLK,
                                   g.NAT, f.V, f.NW, tau);
            }
        })
        g_prof.end(st);
        HIP_TRY(hipGetLastError());
        if (audit_on.load(std::memory_order_relaxed) & 1) {  // debug: how much of its bound did that screen use?
            const float *subk_a = (quarter && !f.split) ? w.subk : nullptr;
            hipLaunchKernelGGL(fft_audit_kernel, dim3(g.NAT, k == 0 ? g.NBLK : g.MAXC, (unsigned)g.B), dim3(64), 0, st,
                               w.res, du, dirty, w.keys, w.ceps, subk_a, g.N, g.A, g.L, g.Ns, g.NBLK, g.NAT);
            HIP_TRY(hipGetLastError());
        }
        g_prof.begin(PROF_SELECT, st);
        if (quarter) {
            const size_t lds_chain = (size_t)(round_up(g.L, 64) + 128) * sizeof(float);
            MP_FFT_DISPATCH(f.logM, {
                constexpr int LQ = LG >= 10 ? LG : 10;  // (smaller sizes never get here)
                // (behind the chains' window and the twiddles: the transform's buffer, or -- if that takes more -- room for four
                //  groups of wavefronts to refine four contender quarters side by side)
                const size_t stage_floats = select_quarter_stage_floats(g.L, f.M);
                const size_t lds_q = lds_chain + ((size_t)f.M / 64 + 64) * sizeof(cpx) + stage_floats * sizeof(float);
                if ((rc = fft_lds_attr(fft_select_quarter_kernel<LQ>, lds_q))) return rc;
                hipLaunchKernelGGL(fft_select_quarter_kernel<LQ>, dim3((unsigned)g.B), dim3(1024), lds_q, st, w.keys,
                                   w.ceps, w.subk, n_cells, w.res, du, w.dirty, w.overflow, out_atom, out_lag, out_gain,
                                   g.N, g.A, g.L, g.Ns, g.NBLK, g.NAT, K, k, rule.du_sub, rule.shift, rule.square, w.tw,
                                   w.xspec, w.wnorm, f.NW, (const float *)w.dscale, bsum, (int)stage_floats);
            })
        } else if (fused) {
            const size_t lds_chain = (size_t)(round_up(g.L, 64) + 128) * sizeof(float);
            MP_FFT_DISPATCH(f.logM, {
                constexpr int LT = LG >= 10 ? LG : 0;                  // 0: no tail transform
                const size_t lds_f = lds_chain + (LT ? ((size_t)f.M + f.M / 64 + 64) * sizeof(cpx) : 0);
                if ((rc = fft_lds_attr(fft_select_fused_kernel<LT>, lds_f))) return rc;
                hipLaunchKernelGGL(fft_select_fused_kernel<LT>, dim3((unsigned)g.B), dim3(1024), lds_f, st, w.keys,
                                   w.ceps, n_cells, w.res, du, w.dirty, w.overflow, out_atom, out_lag, out_gain,
                                   g.N, g.A, g.L, g.Ns, g.NBLK, g.NAT, K, k, rule.du_sub, rule.shift, rule.square,
                                   w.tw, w.xspec, w.wnorm, f.NW, (const float *)w.dscale, bsum,
                                   lazy_form && k >= 1 ? lz.mu : (const float *)nullptr, tau, lz.margin, lz.reuse,
                                   (const float *)w.lfloor, lazy_form && k >= 1 ? w.skip : (unsigned *)nullptr, lz.force);
            })
            if (lazy_form && k >= 1 && k + 1 < K && w.work != nullptr && g.B <= 65535 && lazy_compact.load(std::memory_order_relaxed) != 0) {
                hipLaunchKernelGGL(lazy_compact_kernel, dim3(1), dim3(1024), 0, st, (const unsigned *)w.skip, (int)g.B, g.NAT, w.work);
                HIP_TRY(hipGetLastError());
            }
            if (lazy_form && k == 0 && K > 2) {
                // the lazy screen's floor: where this run's maxima are expected to end -- the (K + K/16 + 1)-th largest peak
                // among the CELLS after step 0's select (mplazy.inc: lazy_floor_cells_kernel; peaks dominate their own
                // tile's cells `radius` blocks either side)
                const int radius = lazy_radius_for(g.L, 0 /* by atom length: the floor counts cells, not blocks */, K);
                hipLaunchKernelGGL(lazy_floor_cells_kernel, dim3((unsigned)g.B), dim3(1024), 0, st, (const u64 *)w.keys,
                                   (const float *)w.ceps, (const unsigned *)w.bsum, g.NBLK, g.NAT, lazy_rank_for(K), radius,
                                   w.lfloor);
                HIP_TRY(hipGetLastError());
            }
        } else {
            // select-A merged into the refinement launch when select-B is the kernel that clears the slots
            // afterwards and a segment's keys are few enough for every workgroup to scan them
            const bool scan_refine = b_tail && !(flags & (MP_FLAG_REFINE_MFMA | MP_FLAG_FFT_UNFUSED)) && n_cells <= 16384;
            // Split transforms (atoms of 5399 .. 21782 samples): the chains' own rounding bound grows with the atom (W_a ~ 0.58 L
            // u ||window||, section 4b), so on noise-like residuals a step can meet more contender cells than the MAXCONT slots --
            // and an overflow costs the whole segment a second encode on the MFMA schedule (2048 x 16384 atoms on noise: two of
            // four segments, scripts/longest_atoms_time.py).  Up to four rounds of select-A + refine there: a round's exact keys go
            // back into the cells, the next round's lower bound rises with them and lists what is left; a round that finds
            // nothing -- or follows a round that listed all its contenders -- leaves at once (two launches, ~8 us a round, against
            // steps of 0.3 .. 1 ms).
            // (two rounds for two halves -- overflows were rare there --, four for four quarters: a round is two launches)
            const int rounds = !f.split || scan_refine ? 1 : f.split == 1 ? 2 : SPLIT_REFINE_ROUNDS;
            for (int round = 0; round < rounds; ++round) {
            if (scan_refine) {
                const size_t lds_win = (size_t)(round_up(g.L, 64) + 128) * sizeof(float);
                hipLaunchKernelGGL(fft_scan_refine_kernel, dim3((unsigned)g.B, 8), dim3(SR_WG), lds_win, st, w.keys, w.ceps,
                                   n_cells, w.res, du, w.cont, w.ncont, w.ekeys, w.overflow, g.N, g.A, g.L, g.Ns, g.NAT);
            } else
                hipLaunchKernelGGL(fft_select_a_kernel, dim3((unsigned)g.B), dim3(1024), 0, st, w.keys, w.ceps, n_cells,
                                   w.cont, w.ncont, w.ekeys, w.overflow, round + 1 == rounds ? 1 : 0,
                                   round > 0 ? w.keys : (u64 *)nullptr, round > 0 ? w.ceps : (float *)nullptr);
            if (scan_refine) {
            } else if (flags & MP_FLAG_REFINE_MFMA) {
                if (dma)
                    hipLaunchKernelGGL(fft_refine_kernel<true>, dim3((unsigned)g.B, 4), dim3(256), lds_ref, st, w.res,
                                       w.img, w.cont, w.ncont, w.ekeys, g.N, g.A, g.Ns, g.NBLK, g.NAT, g.KC, g.NCH);
                else
                    hipLaunchKernelGGL(fft_refine_kernel<false>, dim3((unsigned)g.B, 4), dim3(256), lds_ref, st, w.res,
                                       w.img, w.cont, w.ncont, w.ekeys, g.N, g.A, g.Ns, g.NBLK, g.NAT, g.KC, g.NCH);
            } else {
                const size_t lds_win = (size_t)(round_up(round_up(g.L, 64) + 128, 64) + 16 * REFINE_ASTR) * sizeof(float);
                if ((rc = fft_lds_attr(fft_refine_chain_kernel, lds_win))) return rc;
                // (contenders side by side while that still fits the chip: workgroups past a segment's count leave at once)
                const unsigned ry = (unsigned)std::min<int64_t>(MAXCONT, std::max<int64_t>(1, 2 * (int64_t)num_cus() / (2 * g.B)));
                hipLaunchKernelGGL(fft_refine_chain_kernel, dim3(2, ry, (unsigned)g.B), dim3(256), lds_win, st, w.res, du,
                                   w.cont, w.ncont, w.ekeys, g.N, g.A, g.L, g.Ns, g.NAT);
            }
            }
            if (b_tail) {
                MP_FFT_DISPATCH(f.logM, {
                    constexpr int LB = (LG >= 10 && LG <= 12) ? LG : 10;  // (other sizes never get here)
                    const size_t lds_b = ((size_t)(1 << LB) + (1 << LB) / 64 + 64) * sizeof(cpx);
                    if ((rc = fft_lds_attr(fft_select_b_kernel<LB>, lds_b))) return rc;
                    hipLaunchKernelGGL(fft_select_b_kernel<LB>, dim3((unsigned)g.B), dim3(256), lds_b, st, w.ekeys,
                                       w.res, rule.du_sub, w.dirty, out_atom, out_lag, out_gain, g.N, g.L, g.Ns, g.NBLK,
                                       g.NAT, K, k, w.cont, w.ncont, w.keys, w.ceps, rule.shift, rule.square, w.tw,
                                       w.xspec, w.wnorm, f.NW, (const float *)w.dscale);
                })
            } else {
                hipLaunchKernelGGL(select_subtract_kernel, dim3((unsigned)g.B), dim3(256), 0, st, w.ekeys,
                                   (int64_t)(MAXCONT + 1), w.res, rule.du_sub, w.dirty, out_atom, out_lag, out_gain,
                                   g.N, g.L, g.Ns, g.NBLK, K, k, w.cont, w.ncont, w.keys, w.ceps, n_cells, rule.shift,
                                   rule.square);
            }
        }
        g_prof.end(st);
        HIP_TRY(hipGetLastError());
    }
    return MP_OK;
}

#include "mppersist.inc"

// ---- sub-batches on forked streams -----------------------------------------------------------------
constexpr int MAX_GROUPS = 4;
struct StreamPool {
    hipStream_t streams[MAX_GROUPS];
    hipEvent_t fork, join[MAX_GROUPS];
    int n_concurrent;  // streams[0 .. n_concurrent) were seen to run kernels side by side (see stream_pool)
};

// busy-wait for about `ticks` of the 100 MHz wall clock (self-test of stream concurrency)
__global__ void spin_kernel(long long ticks, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && ticks < 0) *sink = 1;
}

// Two streams that run kernels side by side finish two spins in about the time of one; streams that share a
// hardware queue take twice as long.  Returns elapsed(two streams) / elapsed(one spin), or a negative value.
float stream_pair_ratio(hipStream_t a, hipStream_t b) {
    hipEvent_t e0, ea, eb;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return -1.f;
    const long long ticks = 4000;  // 40 us
    float one = 0.f, ta = 0.f, tb = 0.f;
    bool ok = hipEventRecord(e0, a) == hipSuccess;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, (int *)nullptr);
    ok = ok && hipEventRecord(ea, a) == hipSuccess && hipEventSynchronize(ea) == hipSuccess &&
         hipEventElapsedTime(&one, e0, ea) == hipSuccess;
    ok = ok && hipEventRecord(e0, a) == hipSuccess && hipStreamWaitEvent(b, e0, 0) == hipSuccess;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, (int *)nullptr);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks, (int *)nullptr);
    ok = ok && hipEventRecord(ea, a) == hipSuccess && hipEventRecord(eb, b) == hipSuccess &&
         hipEventSynchronize(ea) == hipSuccess && hipEventSynchronize(eb) == hipSuccess &&
         hipEventElapsedTime(&ta, e0, ea) == hipSuccess && hipEventElapsedTime(&tb, e0, eb) == hipSuccess;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (!ok || one <= 0.f) return -1.f;
    return (ta > tb ? ta : tb) / one;
}
// One pool per host thread and device, never destroyed; built by mp_init_streams(), or by the first sub-batched
// encode of the thread on the device when that call is not being captured (host-synchronising, a few ms, once).
// ROCm multiplexes streams onto a few hardware queues (least-used queue at creation time), and two streams on one
// queue run their kernels one after the other: sub-batches on such a pair are SLOWER than one stream.  Which
// streams collide depends on what the process created before -- measured: after any hipGraph capture in the
// process the first two streams created here shared a queue and the default schedule fell from 840 k to 590 k
// segment-iterations/s (scripts/after_capture.py).  So the pool is chosen, not assumed: of eight candidates, keep
// those that a 40 us spin test shows running side by side with every stream already kept.
// The test synchronises with the host, so it never runs under a capture: a capturing caller whose thread has no
// pool yet gets nullptr and the encode stays on ONE stream (nothing untested is assumed; call mp_init_streams()
// before capturing to get sub-batches inside the graph).
StreamPool *stream_pool(hipStream_t caller, bool may_build = true) {
    static thread_local StreamPool pools[MAX_DEVICES];
    static thread_local bool ready[MAX_DEVICES] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    if (!ready[dev]) {
        if (!may_build) return nullptr;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(caller, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
        StreamPool &p = pools[dev];
        constexpr int NCAND = 8;
        hipStream_t cand[NCAND];
        for (int c = 0; c < NCAND; ++c)
            if (hipStreamCreateWithFlags(&cand[c], hipStreamNonBlocking) != hipSuccess) return nullptr;
        int kept[MAX_GROUPS], n = 0;
        bool used[NCAND] = {false};
        for (int c = 0; c < NCAND && n < MAX_GROUPS; ++c) {
            bool apart = true;
            for (int k = 0; k < n && apart; ++k) {
                const float r = stream_pair_ratio(cand[kept[k]], cand[c]);
                apart = r > 0.f && r < 1.5f;
            }
            if (apart) { kept[n++] = c; used[c] = true; }
        }
        p.n_concurrent = n;
        for (int c = 0, q = n; c < NCAND; ++c) {  // fill the rest of the pool, drop what is left
            if (used[c]) continue;
            if (q < MAX_GROUPS) kept[q++] = c, used[c] = true;
            else (void)hipStreamDestroy(cand[c]);
        }
        for (int q = 0; q < MAX_GROUPS; ++q) {
            p.streams[q] = cand[kept[q]];
            if (hipEventCreateWithFlags(&p.join[q], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        ready[dev] = true;
    }
    return &pools[dev];
}

// the workspace rows of segments [b0, ...): every per-segment array is indexed [B][...]
Workspace sub_batch(const Workspace &w, const Geom &g, int path, int64_t b0, int64_t cells) {
    Workspace v = w;
    v.res = w.res + b0 * g.Ns;
    v.keys = w.keys + b0 * cells;
    v.dirty = w.dirty + 2 * b0;
    if (path == MP_PATH_FFT && w.tw) {
        FftGeom f;
        make_fft_geom(g, &f);
        v.xspec = w.xspec + (size_t)b0 * f.NW * f.M;
        v.wnorm = w.wnorm + b0 * f.NW;
        v.ceps = w.ceps + b0 * cells;
        v.cont = w.cont + b0 * MAXCONT;
        v.ncont = w.ncont + b0;
        v.overflow = w.overflow + b0;
        v.ekeys = w.ekeys + b0 * (MAXCONT + 1);
        if (w.subk) v.subk = w.subk + b0 * cells * SUBCELLS;
        v.bsum = w.bsum + b0 * g.NBLK * 2;
        if (w.skip) v.skip = w.skip + b0 * 4;
        if (w.lfloor) v.lfloor = w.lfloor + b0;
        if (w.work) v.work = w.work + (size_t)b0 * (g.NAT + 1);   // (each sub-batch its own list: its count, then up to n * NAT entries)
    }
    return v;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int mp_version(void) { return 1; }

const char *mp_last_error(void) { return g_err; }

size_t mp_workspace_bytes(int64_t B, int64_t N, int64_t A, int64_t L, int K, int path) {
    if (check_shape(B, N, A, L, K) != MP_OK) return 0;
    if (path == MP_PATH_FFT) return carve(make_geom_for(B, N, A, L, path, 0), path, nullptr, K).bytes;
    size_t b64 = carve(make_geom(B, N, A, L, 64), path, nullptr).bytes;
    size_t b32 = carve(make_geom(B, N, A, L, 32), path, nullptr).bytes;
    return b64 > b32 ? b64 : b32;
}

// ---- the dictionary's coherence table (lazy screen) ---------------------------------------------------------------------
// Row a of the pseudo-batch is atom a followed by zeros, Mc samples in all (coherence_geom): its CIRCULAR correlation with
// atom b holds every shift of the pair.  One FFT screen of the A rows with |.| maxima (A * A / 2 pair transforms), then
// per (row, tile) the maximum over the blocks of approx + 2 eps (eps bounds |screen - chain|, and the chain's own
// rounding is within eps too).
__global__ void coherence_rows_kernel(const float *__restrict__ d, int64_t A, int64_t L, int64_t Nrow, float *__restrict__ rows) {
    const int64_t a = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < Nrow; j += (int64_t)gridDim.x * blockDim.x)
        rows[a * Nrow + j] = j < L ? d[a * L + j] : 0.0f;
}
__global__ void coherence_reduce_kernel(const u64 *__restrict__ keys, const float *__restrict__ ceps, int NBLK, int NAT,
                                        float *__restrict__ out) {
    const int64_t a = blockIdx.x;
    for (int t = threadIdx.x; t < NAT; t += blockDim.x) {
        float m = 0.0f;
        for (int blk = 0; blk < NBLK; ++blk) {
            const int64_t c = (a * NBLK + blk) * NAT + t;
            const u64 kv = keys[c];
            if (kv) m = fmaxf(m, unord_f32((unsigned)(kv >> 32)) + 2.0f * ceps[c]);
        }
        out[a * NAT + t] = m;
    }
}
// Every shift of a pair of atoms from ONE circular transform of Mc >= 2 L - 1 points: with atom a in samples 0 .. L-1 of an
// otherwise zero row of Mc samples, sum_k row[(t + k) mod Mc] d_b[k] is the pair's correlation at shift +t for t < L and at
// shift -(Mc - t) for t > Mc - L (nothing wraps onto anything: L + L - 1 <= Mc).  The encode's own transform for this atom
// length is 2 - 4 times larger (3 L + 190: a whole dirty run of valid lags per LINEAR correlation): 8192 points for
// 2048-sample atoms, where 4096 do here -- 113 -> ~50 ms for the 4096 x 2048 dictionary, 0.45 -> ~0.2 ms for 512 x 512.
static bool coherence_geom(int64_t A, int64_t L, Geom *g, FftGeom *f) {
    if (A <= 0 || L <= 0) return false;
    Geom enc = make_geom_for(1, 3 * L, A, L, MP_PATH_FFT, 0);
    FftGeom fe;
    if (!make_fft_geom(enc, &fe) || fe.split || fe.logM < 10 || fe.logM > 13) return false;   // (where the lazy screen exists)
    int lg = 10;
    while ((1ll << lg) < 2 * L - 1) ++lg;
    *g = make_geom_for(A, 1ll << lg, A, L, MP_PATH_FFT, 0);
    g->circ = lg;
    return make_fft_geom(*g, f) && !f->split && f->logM == lg;
}
size_t mp_coherence_workspace_bytes(int64_t A, int64_t L) {
    Geom g;
    FftGeom f;
    if (!coherence_geom(A, L, &g, &f)) return 0;
    return carve(g, MP_PATH_FFT, nullptr, 1).bytes + 256 + (size_t)A * g.N * sizeof(float);
}
int mp_coherence_f32(const float *dict_unit, int64_t A, int64_t L, float *out, void *workspace, size_t workspace_bytes,
                     void *stream) {
    Geom g;
    FftGeom f;
    if (!dict_unit || !out || !workspace) return fail(MP_ERR_ARG, "mp_coherence_f32: null argument%s");
    if (!coherence_geom(A, L, &g, &f))
        return fail(MP_ERR_UNSUPPORTED, "mp_coherence_f32: the lazy screen exists for 1024- to 8192-point transforms only%s");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MP_ERR_WORKSPACE, "workspace not 256-byte aligned%s");
    if (workspace_bytes < mp_coherence_workspace_bytes(A, L)) return fail(MP_ERR_WORKSPACE, "workspace too small%s");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace w = carve(g, MP_PATH_FFT, static_cast<char *>(workspace), 1);
    const int64_t Nrow = g.N;   // (= the circular transform's size)
    float *rows = reinterpret_cast<float *>(static_cast<char *>(workspace) + ((w.bytes + 255) / 256) * 256);
    hipLaunchKernelGGL(coherence_rows_kernel, dim3((unsigned)((Nrow + 255) / 256), (unsigned)A), dim3(256), 0, st, dict_unit, A, L,
                       Nrow, rows);
    HIP_TRY(hipGetLastError());
    int rc;
    if ((rc = stage_inputs(g, w, MP_PATH_FFT, rows, dict_unit, 0, st, false))) return rc;
    if ((rc = fft_setup(g, w, dict_unit, 0, 1, st))) return rc;
    const TauModel tm = fft_tau(f.logM);
    hipLaunchKernelGGL(max_row_norm_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, st, dict_unit, A, L, w.dscale, tm.chain_w,
                       tm.fft_w);
    HIP_TRY(hipGetLastError());
    const Rule rule{dict_unit, 0, 0, 0};
    g_prof.arm(-1);
    if ((rc = fft_iteration(g, w, dict_unit, 1, 0, MP_FLAG_INTERNAL_COHERENCE | MP_FLAG_INTERNAL_ONE_STREAM, nullptr, nullptr, nullptr,
                            rule, st)))
        return rc;
    hipLaunchKernelGGL(coherence_reduce_kernel, dim3((unsigned)A), dim3(64), 0, st, w.keys, w.ceps, g.NBLK, g.NAT, out);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_tune(int key, double value) {
    if (key == MP_TUNE_TAU && value >= 0.0) { tau_override.store((float)value); return MP_OK; }  // 0: back to the model
    if (key == MP_TUNE_SCREEN_PPS) { screen_pps_override.store((int)value); return MP_OK; }
    if (key == MP_TUNE_GROUPS && value >= 2 && value <= MAX_GROUPS) { overlap_groups.store((int)value); return MP_OK; }
    if (key == MP_TUNE_AUDIT) { audit_on.store((int)value); return MP_OK; }  // bit 0: screen audit; bit 1: persistent-schedule latencies
    if (key == MP_TUNE_PERSIST_SHARDS && value >= 0) { persist_shards.store((int)value); return MP_OK; }
    if (key == MP_TUNE_PERSIST_WORKERS && value >= 0) { persist_workers.store((int)value); return MP_OK; }
    if (key == MP_TUNE_PERSIST_SELECTS && value >= 0) { persist_selects.store((int)value); return MP_OK; }
    if (key == MP_TUNE_LAZY_MARGIN && value >= 0.0 && value <= 1.0) { persist_margin.store((float)value); return MP_OK; }   // 0: the defaults (0.7 inside the persistent launch, 0.85 between launches)
    if (key == MP_TUNE_LAZY_REUSE && value >= 0 && value <= 4) { persist_reuse.store((int)value); return MP_OK; }
    if (key == MP_TUNE_LAZY_RADIUS && value >= -1 && value <= 64) { persist_radius.store((int)value); return MP_OK; }
    if (key == MP_TUNE_PERSIST_PRESCAN && (value == 0 || value == 1)) { persist_prescan.store((int)value); return MP_OK; }
    if (key == MP_TUNE_LAZY_FORCE && value >= 0 && value < 3) {
        // (random tile masks: the events are WRONG while it is set -- a timing instrument, accepted only from a process that
        //  says so in its environment, so that no product path can switch it on by accident)
        const char *ok = getenv("MP_ALLOW_WRONG_RESULTS");
        if (value > 0 && !(ok && ok[0] == '1')) return fail(MP_ERR_ARG, "mp_tune(MP_TUNE_LAZY_FORCE): set MP_ALLOW_WRONG_RESULTS=1 in the environment%s");
        lazy_force.store((float)value);
        return MP_OK;
    }
    if (key == MP_TUNE_CLEAR_MEMSET && (value == 0 || value == 1)) {
        // (hipMemsetAsync clears captured into a hipGraph replay a stale fill pattern on this runtime -- wrong events under
        //  EncodePlan, profiles/r03_graph_memset_*.txt -- so the switch is a repro instrument behind the same environment gate)
        const char *ok = getenv("MP_ALLOW_WRONG_RESULTS");
        if (value > 0 && !(ok && ok[0] == '1')) return fail(MP_ERR_ARG, "mp_tune(MP_TUNE_CLEAR_MEMSET): set MP_ALLOW_WRONG_RESULTS=1 in the environment%s");
        clear_with_memset.store((int)value);
        return MP_OK;
    }
    if (key == MP_TUNE_LAZY_COMPACT && (value == 0 || value == 1)) { lazy_compact.store((int)value); return MP_OK; }
    if (key == MP_TUNE_PERSIST_FINE && value >= 0 && value <= 2) { persist_fine.store((int)value); return MP_OK; }
    return fail(MP_ERR_ARG, "mp_tune: unknown key or bad value%s");
}

int mp_profile_enable(int on) {
    g_prof.every.store(on < 0 ? 0 : (on & 0xffff));
    g_prof.skip.store(on < 0 ? 0 : ((on >> 16) & 7));
    return MP_OK;
}

int mp_profile_read(double *ms, int64_t *count) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!ms || !count) return fail(MP_ERR_ARG, "mp_profile_read: null output%s");
    for (int q = 0; q < PROF_KINDS; ++q) { ms[q] = 0.0; count[q] = 0; }
    for (ProfSpan &sp : g_prof.spans) {
        HIP_TRY(hipEventSynchronize(sp.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
        ms[sp.kind] += (double)t;
        count[sp.kind] += 1;
        g_prof.pool.push_back(sp.a);
        g_prof.pool.push_back(sp.b);
    }
    g_prof.spans.clear();
    return MP_OK;
}

int mp_unit_norm_f32(const float *d, int64_t A, int64_t L, float eps, float *out, void *stream) {
    if (!d || !out || A <= 0 || L <= 0) return fail(MP_ERR_ARG, "mp_unit_norm_f32: bad arguments%s");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(unit_norm_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, st, d, A, L, eps, out);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

static int encode_impl(const float *signal, int64_t B, int64_t N, const float *dict_in, int64_t A,
                       int64_t L, int K, int path, int flags, int64_t *out_atom, int64_t *out_lag,
                       float *out_gain, float *out_residual, void *workspace, size_t workspace_bytes,
                       void *stream, bool conv_model, const float *coherence = nullptr) {
    int rc = check_shape(B, N, A, L, K);
    if (rc) return rc;
    flags &= ~MP_FLAG_INTERNAL_ONE_STREAM;  // ours to set
    if (path != MP_PATH_DIRECT && path != MP_PATH_INCREMENTAL && path != MP_PATH_NAIVE && path != MP_PATH_FFT)
        return fail(MP_ERR_ARG, "unknown path%s");
    if (path == MP_PATH_NAIVE && (B > 65535 || A > 65535))
        return fail(MP_ERR_ARG, "MP_PATH_NAIVE: B and A must be <= 65535%s");
    if (B == 0) return MP_OK;
    if (!signal || !dict_in || !workspace) return fail(MP_ERR_ARG, "null pointer%s");
    if (K > 0 && (!out_atom || !out_lag || !out_gain)) return fail(MP_ERR_ARG, "null output%s");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MP_ERR_WORKSPACE, "workspace not 256-byte aligned%s");
    Geom g = make_geom_for(B, N, A, L, path, flags);
    Workspace w = carve(g, path, static_cast<char *>(workspace), K);
    if (w.bytes > workspace_bytes) return fail(MP_ERR_WORKSPACE, "workspace too small%s");
    hipStream_t st = static_cast<hipStream_t>(stream);

    // dictionary the correlation runs against, and the residual update rule
    const float *dict_unit = dict_in;
    Rule rule{dict_in, 0, 0, 0};
    if (conv_model) {
        hipLaunchKernelGGL(reverse_rows_kernel, dim3(256), dim3(256), 0, st, dict_in, A, L, w.drev);
        HIP_TRY(hipGetLastError());
        dict_unit = w.drev;
        rule = Rule{dict_in, (int)(L - 1), 1, L - 1};
    }
    // (MP_PATH_FFT reads the dictionary image only when its refinement runs on the MFMA cell code)
    rc = stage_inputs(g, w, path, signal, dict_unit, rule.lead, st, path != MP_PATH_FFT || (flags & MP_FLAG_REFINE_MFMA));
    if (rc) return rc;
    if (path == MP_PATH_FFT && (rc = fft_setup(g, w, dict_unit, flags, K, st))) return rc;
    if (path == MP_PATH_FFT) {
        // the screen's bound |fm| <= ||window|| * max_a ||d_a||: 1 for the unit-norm dictionary this entry point
        // is documented for, but measured rather than trusted (and the convolution model's atoms are raw)
        FftGeom f;
        if (!make_fft_geom(g, &f)) return fail(MP_ERR_UNSUPPORTED, "MP_PATH_FFT: atom too long%s");
        const TauModel tm = fft_tau(f.logM);
        hipLaunchKernelGGL(max_row_norm_kernel, dim3((unsigned)((A + 3) / 4)), dim3(256), 0, st, dict_unit, A, L, w.dscale,
                           tm.chain_w, tm.fft_w);
        HIP_TRY(hipGetLastError());
    }

    // Segments are independent, so the batch can be cut into sub-batches on forked streams -- while one
    // is in its short, latency-bound select kernels the others keep the CUs busy.  Joined back into the
    // caller's stream before returning; fork/join by events is graph-capture safe.
    // Measured (scripts/sub_batches.py, headline dictionary, one synchronised encode at a time; segment-iterations/s
    // with 1 / 2 / 4 sub-batches): 16 segments 409 / 367 / 377 k, 32: 632 / 621 / 631 k, 48: 709 / 759 / 780 k,
    // 64: 814 / 846 / 877 k, 128: 899 / 1041 / 1050 k, 256: 985 / 1176 / 1180 k; +1 % on the incremental MFMA
    // schedule; -4 % at the config-4 shape, whose screens fill the GPU on their own.  Default for MP_PATH_FFT:
    // four sub-batches from 48 segments up when a segment has < 65536 cells (MP_FLAG_NO_OVERLAP turns it off),
    // opt-in elsewhere (MP_FLAG_OVERLAP).  Only on streams seen to run side by side (stream_pool).
    // The persistent schedule (mppersist.inc): step 0 as separate kernels (full-pass screen, quarter select), then
    // steps 1 .. K-1 of the whole batch in one launch of resident workgroups.
    // Default for MP_PATH_FFT at every batch size where it applies
    // (MP_FLAG_FFT_NO_PERSISTENT, or any flag that asks for
    // a particular launch-per-step form or sub-batch count, turns it off; MP_FLAG_FFT_PERSISTENT asks for it at any size).
    // Measured, headline dictionary, eight encodes back to back (scripts/persist_percu.py; k segment-iterations/s,
    // persistent / one stream / sub-batches): 16 segments 415 / 420 / 388, 24: 577 / 519 / 520, 32: 732 / 572 / 647,
    // 48: 930 / 726 / 703, 64: 1097 / 827 / 874, 96: 1118 / 835 / 987, 128: 1124 / 897 / 1056.
    const int forms = MP_FLAG_NO_OVERLAP | MP_FLAG_OVERLAP | MP_FLAG_FFT_NO_QUARTER | MP_FLAG_FFT_QUARTER | MP_FLAG_FFT_FUSED |
                      MP_FLAG_FFT_UNFUSED | MP_FLAG_REFINE_MFMA | MP_FLAG_FFT_SIMPLE | MP_FLAG_FFT_NO_PERSISTENT |
                      (7 << MP_FLAG_GROUPS_SHIFT);
    // (With the select at 15 us -- its chains on the matrix core -- the one-launch form also wins for few segments where an
    //  entry has enough tasks to spread: 512 x 512 dictionary, persistent / one stream: 8 segments 256 / 233 k, 12: 363 /
    //  295, 16: 462 / 407, 20: 551 / 446; 256 x 1024: 212 / 198, 306 / 252, 398 / 309, 481 / 348.  Two tiles -- 64 x 300 --
    //  lose until 20 segments: 331 / 363 at 8, 660 / 687 at 16.  scripts/small_batches.py.)
    // (... and with the select workers working ahead of their screens -- a select is 6 us then -- it wins at every batch size, one
    //  segment included: 512 x 512, persistent / one stream 48 / 41 k at 1 segment, 94 / 75 at 2, 181 / 137 at 4, 250 / 187
    //  at 6; 256 x 1024: 42 / 33, 78 / 62, 145 / 113, 207 / 159; 64 x 300: 54 / 48, 105 / 92, 204 / 186, 302 / 272, 600 /
    //  516 at 12.)
    // ... but not at every LOAD (round 3, scripts/form_sweep.py: six dictionaries x 8 .. 256 segments, with and without the
    // table).  The one-launch form wins while a step is a chain of latencies; once the screens of a step are enough work to
    // fill the chip for long, what counts is the rate at which transforms get done, and there the plain screen kernel (four
    // wavefronts per SIMD, sixteen pairs per workgroup, workgroups of one tile side by side: its pair spectra shared in
    // the L2s) does more than the queue's tasks (three per SIMD, four pairs per slot, tiles interleaved) -- above all where
    // the pair spectra outgrow the L2s (1024 x 1024: 16.8 MB, persistent 190 k segment-iterations/s at ANY batch from 32
    // segments up against 256 - 274 k launch per step).  With the table the launch-per-step side is the fused select with
    // its lazy screen on sub-batches.  Transform points per step = B x ceil(A / 2) x M:
    //   with the table    512 x 512:  67 M (128 segments) persistent 1224 against 1119 k, 134 M (256) 1217 against 1348
    //                    1024 x 512:  67 M (64) 652 / 567,  134 M (128) 681 / 703       256 x 1024: 134 M (256) 1336 / 1299
    //                    1024 x 1024: 67 M (32) 305 / 251,  134 M (64) 318 / 376, 268 M (128) 319 / 498
    //                    2048 x 512:  67 M (32) 286 / 218,  134 M (64) 283 / 337
    //                    1024-point transforms: 512 x 256: 17 M (64) 1296 / 981, 34 M (128) 1310 / 1431; 2048 x 256: 34 M (32)
    //                    411 / 326, 67 M (64) 456 / 434, 134 M (128) 448 / 481; 4096 x 256: 34 M (16) 177 / 157, 134 M (64) 200 / 238
    //                    (below 48 segments the launch-per-step side has one stream and a handful of select workgroups:
    //                    the one-launch form wins there at every load measured)
    //   without           level up to 134 M while the pair spectra fit (<= 8 MB; 2048 x 256 at 268 M: 356 / 397);
    //                    1024 x 1024: 34 M (16) 177 / 178, 67 M 187 / 226; 2048 x 512: 34 M 175 / 164, 67 M 180 / 208
    FftGeom fp;
    bool short_segments = false;
    bool persist_size = make_fft_geom(g, &fp);
    if (persist_size) {
        const double pairs = (double)((A + 1) / 2);
        const double points = (double)B * pairs * fp.M;
        persist_size = (coherence && !conv_model)
                           ? (B < FORM.sub_batch_min_segments ||
                              (fp.logM == 10 ? (points <= FORM.persist_points_any_batch_1024 ||
                                                (points <= FORM.persist_points_table_1024 && B <= FORM.persist_segments_table_1024))
                                             : points <= FORM.persist_points_table))
                           : ((pairs * fp.M * 8.0 <= FORM.persist_spectra_bytes && points <= FORM.persist_points_fit) || points <= FORM.persist_points_nofit);
        // Short segments -- an event dirties half of the segment's lags or more (N <= 4 L: the multiband model's bands are
        // exactly that) -- leave a select nothing to do ahead of its screen and nothing for the lazy screen to skip, and the
        // one-launch form keeps only its 256-thread selects and its hand-offs: launch per step with the quarter select (1024
        // threads, contender quarters side by side) is ahead at 4096-point transforms at every batch size measured (1024 x
        // 1024 atoms, 4096-sample segments, scripts/small_batch_forms.py, k segment-iterations/s one launch / per step: 1
        // segment 22 / 31, 8: 115 / 150, 32: 217 / 279, 64: 223 / 307), at 2048-point transforms up to 8 segments (1024 x 512,
        // 2048 samples: 8: 206 / 233, 16: 359 / 370, 32: 597 / 507), level at 1024 points.
        if (FORM.short_ratio * L >= N && (fp.logM == FORM.short_logm_always || (fp.logM == FORM.short_logm_small && B <= FORM.short_small_segments))) {
            persist_size = false;
            short_segments = true;
        }
    }
    const bool persist = path == MP_PATH_FFT && ((flags & MP_FLAG_FFT_PERSISTENT_BIT) || (persist_size && !(flags & forms))) &&
                         !audit_on.load(std::memory_order_relaxed);  // (the audit checks screens launch by launch)
    // a shape the persistent form would take but for its load, with the table: the fused select, which has the lazy screen
    // (not for short segments: nothing to skip there, and the quarter select is the faster one)
    if (path == MP_PATH_FFT && !persist && coherence && !conv_model && !(flags & forms) && fp.logM >= 10 && fp.logM <= 12 && !fp.split &&
        !short_segments)
        flags |= MP_FLAG_FFT_FUSED;
    if (persist) {
        FftGeom f;
        if (make_fft_geom(g, &f) && persist_eligible(g, f, w, K)) {
            const int f0 = (flags & ~(MP_FLAG_FFT_NO_QUARTER | MP_FLAG_FFT_FUSED | MP_FLAG_FFT_UNFUSED | MP_FLAG_REFINE_MFMA |
                                      MP_FLAG_FFT_SIMPLE)) | MP_FLAG_FFT_QUARTER | MP_FLAG_INTERNAL_ONE_STREAM | MP_FLAG_FFT_PERSISTENT_BIT;
            last_schedule = -1;
            g_prof.arm(0);
            if ((rc = fft_iteration(g, w, dict_unit, K, 0, f0, out_atom, out_lag, out_gain, rule, st))) return rc;
            g_prof.begin(PROF_CORR_INC, st);  // (one span around the whole launch: steps 1 .. K-1)
            const float *mu = conv_model ? nullptr : coherence;
            float *lbfloor = reinterpret_cast<float *>(w.pctl + 256 + 512 * 64 + ((size_t)B * (K - 1) + 2) * 128);
            if (mu) {   // the lazy screen's floor: where this run's maxima are expected to end (mppersist.inc)
                // (rank K + K/16 + 1: now and then one event leaves two peaks; radius 1 + ceil(max(0, L - 512) / 256) blocks)
                const int radius = lazy_radius_for(L, g.NBLK, K);
                if (g.NBLK <= FLOOR_WAVE_MAXBLK)
                    hipLaunchKernelGGL(persist_floor_wave_kernel, dim3((unsigned)B), dim3(64), 0, st, (const unsigned *)w.bsum, g.NBLK,
                                       lazy_rank_for(K), radius, lbfloor);
                else
                    hipLaunchKernelGGL(persist_floor_kernel, dim3((unsigned)B), dim3(256), 0, st, (const unsigned *)w.bsum, g.NBLK,
                                       lazy_rank_for(K), radius, lbfloor);
                HIP_TRY(hipGetLastError());
            }
            rc = launch_persistent(g, f, w, dict_unit, rule, K, fft_tau(f.logM).tau, out_atom, out_lag, out_gain, st, mu, lbfloor);
            g_prof.end(st);
            if (rc) return rc;
            hipLaunchKernelGGL(persist_mark_kernel, dim3((unsigned)B), dim3(64), 0, st, w.pctl, w.overflow, out_gain, (int)B, K);
            HIP_TRY(hipGetLastError());
            if (out_residual) {
                dim3 grid((unsigned)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024), (unsigned)B);
                hipLaunchKernelGGL(copy_residual_kernel, grid, dim3(256), 0, st, w.res, N, g.Ns, out_residual, rule.lead);
                HIP_TRY(hipGetLastError());
            }
            return MP_OK;
        }
    }
    int n_groups = 1;
    // (segments of 65536 cells and more: their screens fill the GPU alone -- sub-batches measured level, 0.878 against
    //  0.879 s at the config-4 shape -- until the lazy screen takes two thirds of the screen's workgroups out: then the
    //  0.15 ms select between two 1.2 ms screens is a tenth of the step, and four sub-batches hide most of it: 0.425 ->
    //  0.398 s, three runs each, scripts/c4_full.py)
    if (K > 0 && !(flags & MP_FLAG_NO_OVERLAP) &&
        (((flags & MP_FLAG_OVERLAP) && B >= 8) ||
         (path == MP_PATH_FFT && B >= FORM.sub_batch_min_segments && ((int64_t)g.NBLK * g.NAT < FORM.fused_min_cells || (coherence && !conv_model))))) {
        const int per_call = (flags >> MP_FLAG_GROUPS_SHIFT) & 7;                  // MP_FLAG_GROUPS(n): this call only
        const int dflt = overlap_groups.load(std::memory_order_relaxed);
        n_groups = per_call >= 2 && per_call <= MAX_GROUPS ? per_call : (dflt >= 2 && dflt <= MAX_GROUPS ? dflt : FORM.sub_batches);
    }
    StreamPool *pool = nullptr;
    if (n_groups > 1) {
        pool = stream_pool(st);  // nullptr under a capture without a pool: one stream, nothing untested assumed
        if (pool && n_groups > pool->n_concurrent) n_groups = pool->n_concurrent;  // never sub-batches on one hardware queue
        if (!pool || n_groups < 2) { n_groups = 1; pool = nullptr; }
    }
    last_schedule = n_groups;
    if (pool) {
        HIP_TRY(hipEventRecord(pool->fork, st));
        for (int q = 0; q < n_groups; ++q) HIP_TRY(hipStreamWaitEvent(pool->streams[q], pool->fork, 0));
    }
    const bool naive = path == MP_PATH_NAIVE;
    const bool incremental = path == MP_PATH_INCREMENTAL;
    const int64_t cells = (int64_t)g.NBLK * (naive ? g.A : g.NAT);
    // the lazy screen where the launch-per-step form has it (fft_iteration: the fused select with block summaries)
    LazyArgs lz;
    if (path == MP_PATH_FFT && coherence && !conv_model) {
        lz.mu = coherence;
        lz.margin = lazy_margin_for(true, 0);
        lz.reuse = lazy_reuse_for(K, true);
        lz.force = lazy_force.load(std::memory_order_relaxed);
    }
    // the K steps; whatever it returns, the forked streams are joined back into the caller's below
    auto run_steps = [&]() -> int {
        for (int k = 0; k < K; ++k) {
            g_prof.arm(k);
            for (int q = 0; q < n_groups; ++q) {
                const int64_t b0 = B * q / n_groups, b1 = B * (q + 1) / n_groups;
                Geom gq = g;
                gq.B = b1 - b0;
                Workspace wq = sub_batch(w, g, path, b0, cells);
                hipStream_t sq = n_groups > 1 ? pool->streams[q] : st;
                int64_t *oa = out_atom + b0 * K, *ol = out_lag + b0 * K;
                float *og = out_gain + b0 * K;
                int rcq;
                if (path == MP_PATH_FFT) {
                    if ((rcq = fft_iteration(gq, wq, dict_unit, K, k,
                                             flags | (n_groups == 1 ? MP_FLAG_INTERNAL_ONE_STREAM : 0), oa, ol, og, rule, sq, lz)))
                        return rcq;
                    continue;
                }
                const bool full = (k == 0) || !incremental;
                const int *dirty = full ? nullptr : wq.dirty;
                g_prof.begin(full ? PROF_CORR_FULL : PROF_CORR_INC, sq);
                if (naive)
                    rcq = launch_naive(gq, wq, dict_unit, dirty, gq.NBLK, nullptr, sq);
                else
                    rcq = launch_correlate<false>(gq, wq, dirty, nullptr, flags, sq);
                g_prof.end(sq);
                if (rcq) return rcq;
                g_prof.begin(PROF_SELECT, sq);
                hipLaunchKernelGGL(select_subtract_kernel, dim3((unsigned)gq.B), dim3(256), 0, sq, wq.keys, cells, wq.res,
                                   rule.du_sub, incremental ? wq.dirty : nullptr, oa, ol, og, N, L, g.Ns, g.NBLK, K, k,
                                   (const int *)nullptr, (const int *)nullptr, (u64 *)nullptr, (float *)nullptr,
                                   (int64_t)0, rule.shift, rule.square);
                g_prof.end(sq);
                HIP_TRY(hipGetLastError());
            }
        }
        return MP_OK;
    };
    rc = run_steps();
    if (n_groups > 1) {  // also after an error: a forked stream left unjoined would break the caller's capture / ordering
        for (int q = 0; q < n_groups; ++q) {
            const hipError_t e1 = hipEventRecord(pool->join[q], pool->streams[q]);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, pool->join[q], 0) : e1;
            if (e2 != hipSuccess && rc == MP_OK) rc = fail(MP_ERR_HIP, "joining the internal streams: %s", hipGetErrorString(e2));
        }
    }
    if (rc) return rc;
    if (path == MP_PATH_FFT && K > 0) {
        hipLaunchKernelGGL(fft_mark_overflow_kernel, dim3((unsigned)B), dim3(64), 0, st, w.overflow, out_gain, K);
        HIP_TRY(hipGetLastError());
    }
    if (out_residual) {
        dim3 grid((unsigned)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024), (unsigned)B);
        hipLaunchKernelGGL(copy_residual_kernel, grid, dim3(256), 0, st, w.res, N, g.Ns, out_residual, rule.lead);
        HIP_TRY(hipGetLastError());
    }
    return MP_OK;
}

int mp_encode_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A,
                  int64_t L, int K, int path, int flags, int64_t *out_atom, int64_t *out_lag,
                  float *out_gain, float *out_residual, void *workspace, size_t workspace_bytes,
                  void *stream) {
    return encode_impl(signal, B, N, dict_unit, A, L, K, path, flags, out_atom, out_lag, out_gain, out_residual,
                       workspace, workspace_bytes, stream, false);
}

int mp_encode_lazy_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A,
                       int64_t L, int K, int path, int flags, const float *coherence, int64_t *out_atom, int64_t *out_lag,
                       float *out_gain, float *out_residual, void *workspace, size_t workspace_bytes,
                       void *stream) {
    return encode_impl(signal, B, N, dict_unit, A, L, K, path, flags, out_atom, out_lag, out_gain, out_residual,
                       workspace, workspace_bytes, stream, false, coherence);
}

int mp_encode_conv_f32(const float *signal, int64_t B, int64_t N, const float *atoms, int64_t A,
                       int64_t L, int K, int path, int flags, int64_t *out_atom, int64_t *out_time,
                       float *out_value, float *out_residual, void *workspace, size_t workspace_bytes,
                       void *stream) {
    return encode_impl(signal, B, N, atoms, A, L, K, path, flags, out_atom, out_time, out_value, out_residual,
                       workspace, workspace_bytes, stream, true);
}

// sparse_code(..., local_contrast_norm=True), modules/matchingpursuit.py:284-294.  Workspace = the direct
// path's + the dense map [B, A, N] + one LCN key per (64-lag block, 16-atom tile).
static size_t lcn_extra(const Geom &g, size_t *o_map, size_t *o_keys, size_t base) {
    size_t off = (size_t)round_up((int64_t)base, 256);
    *o_map = off;   // the cell-order map: [B][NBLK][NAT] cells of 32 x 64 floats
    off += (size_t)round_up((int64_t)((size_t)g.B * g.NBLK * g.NAT * CELL_FLOATS * sizeof(float)), 256);
    *o_keys = off;
    off += (size_t)round_up((int64_t)((size_t)g.B * g.NBLK * g.NAT * sizeof(u64)), 256);
    return off;
}

size_t mp_lcn_workspace_bytes(int64_t B, int64_t N, int64_t A, int64_t L, int K) {
    if (check_shape(B, N, A, L, K) != MP_OK) return 0;
    Geom g = make_geom(B, N, A, L, 32);
    size_t o_map, o_keys;
    return lcn_extra(g, &o_map, &o_keys, carve(g, MP_PATH_INCREMENTAL, nullptr).bytes);
}

int mp_encode_lcn_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A, int64_t L,
                      int K, int64_t *out_atom, int64_t *out_lag, float *out_gain, float *out_residual,
                      void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_shape(B, N, A, L, K);
    if (rc) return rc;
    if (B > 65535 || (A + 31) / 32 > 65535)
        return fail(MP_ERR_ARG, "mp_encode_lcn_f32: B and A / 32 must be <= 65535%s");
    if (B == 0) return MP_OK;
    if (!signal || !dict_unit || !workspace) return fail(MP_ERR_ARG, "null pointer%s");
    if (K > 0 && (!out_atom || !out_lag || !out_gain)) return fail(MP_ERR_ARG, "null output%s");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MP_ERR_WORKSPACE, "workspace not 256-byte aligned%s");
    Geom g = make_geom(B, N, A, L, 32);
    char *base = static_cast<char *>(workspace);
    Workspace w = carve(g, MP_PATH_INCREMENTAL, base);
    size_t o_map, o_keys;
    if (lcn_extra(g, &o_map, &o_keys, w.bytes) > workspace_bytes) return fail(MP_ERR_WORKSPACE, "workspace too small%s");
    float *map = reinterpret_cast<float *>(base + o_map);
    u64 *lkeys = reinterpret_cast<u64 *>(base + o_keys);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = stage_inputs(g, w, MP_PATH_INCREMENTAL, signal, dict_unit, 0, st))) return rc;
    for (int k = 0; k < K; ++k) {
        const int *dirty = k == 0 ? nullptr : w.dirty;
        if ((rc = launch_correlate_cellmap(g, w, dirty, map, st))) return rc;
        const int nb = k == 0 ? g.NBLK : (g.MAXC + 2 < g.NBLK ? g.MAXC + 2 : g.NBLK);
        hipLaunchKernelGGL(lcn_keys_kernel, dim3((unsigned)nb, (unsigned)g.NAT, (unsigned)B), dim3(256), 0, st, map,
                           dirty, lkeys, N, A, g.NBLK, g.NAT);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(lcn_select_subtract_kernel, dim3((unsigned)B), dim3(256), 0, st, lkeys,
                           (int64_t)g.NBLK * g.NAT, map, w.res, dict_unit, w.dirty, out_atom, out_lag, out_gain, N, A, L,
                           g.Ns, K, k, g.NBLK, g.NAT);
        HIP_TRY(hipGetLastError());
    }
    if (out_residual) {
        dim3 grid((unsigned)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024), (unsigned)B);
        hipLaunchKernelGGL(copy_residual_kernel, grid, dim3(256), 0, st, w.res, N, g.Ns, out_residual, (int64_t)0);
        HIP_TRY(hipGetLastError());
    }
    return MP_OK;
}

int mp_feature_map_f32(const float *residual, int64_t B, int64_t N, const float *dict_unit,
                       int64_t A, int64_t L, float *fm, void *workspace, size_t workspace_bytes,
                       void *stream) {
    int rc = check_shape(B, N, A, L, 0);
    if (rc) return rc;
    if (B == 0) return MP_OK;
    if (!residual || !dict_unit || !fm || !workspace) return fail(MP_ERR_ARG, "null pointer%s");
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return fail(MP_ERR_WORKSPACE, "workspace not 256-byte aligned%s");
    Geom g = make_geom(B, N, A, L, 64);
    Workspace w = carve(g, MP_PATH_DIRECT, static_cast<char *>(workspace));
    if (w.bytes > workspace_bytes) return fail(MP_ERR_WORKSPACE, "workspace too small%s");
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_inputs(g, w, MP_PATH_DIRECT, residual, dict_unit, 0, st);
    if (rc) return rc;
    return launch_correlate<true>(g, w, nullptr, fm, 0, st);
}

int mp_fft_c2c_f32(const float *in, float *out, int log2_m, int64_t batch, int inverse, void *workspace,
                   void *stream) {
    if (!in || !out || !workspace || batch < 0 || log2_m < 8 || log2_m > 14)
        return fail(MP_ERR_ARG, "mp_fft_c2c_f32: bad arguments (8 <= log2_m <= 14)%s");
    if (batch == 0) return MP_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int M = 1 << log2_m;
    cpx *tw = static_cast<cpx *>(workspace);
    hipLaunchKernelGGL(fft_twiddle_kernel, dim3((M + 255) / 256), dim3(256), 0, st, tw, M);
    const size_t lds = (size_t)M * sizeof(cpx);
    int rc;
    if (inverse == 2) {  // the screen's register transform (inverse only, M >= 1024)
        if (log2_m < 10) return fail(MP_ERR_ARG, "mp_fft_c2c_f32: the screen transform needs log2_m >= 10%s");
        MP_FFT_DISPATCH(log2_m, {
            constexpr int LS = LG >= 10 ? LG : 10;
            using C = ScreenCfg<LS>;
            const size_t lds_s = ((size_t)C::SLOTS * C::M + C::M / 64 + 64) * sizeof(cpx);
            if ((rc = fft_lds_attr(fft_screen_probe_kernel<LS>, lds_s))) return rc;
            hipLaunchKernelGGL(fft_screen_probe_kernel<LS>, dim3((unsigned)((batch + C::SLOTS - 1) / C::SLOTS)),
                               dim3(C::WG), lds_s, st, reinterpret_cast<const cpx *>(in),
                               reinterpret_cast<cpx *>(out), tw, batch);
        })
        HIP_TRY(hipGetLastError());
        return MP_OK;
    }
    MP_FFT_DISPATCH(log2_m, {
        if ((rc = fft_lds_attr(fft_c2c_kernel<LG>, lds))) return rc;
        hipLaunchKernelGGL(fft_c2c_kernel<LG>, dim3((unsigned)batch), dim3(256), lds, st,
                           reinterpret_cast<const cpx *>(in), reinterpret_cast<cpx *>(out), tw, inverse);
    })
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_scatter_f32(const int64_t *atom, const int64_t *batch, const int64_t *lag, const float *gain,
                   int64_t n_events, const float *dict_unit, int64_t A, int64_t L, float *out,
                   int64_t B, int64_t N, void *stream) {
    if (n_events == 0 || B == 0) return MP_OK;
    if (!atom || !batch || !lag || !gain || !dict_unit || !out || A <= 0 || L <= 0 || N <= 0 || B < 0 ||
        n_events < 0)
        return fail(MP_ERR_ARG, "mp_scatter_f32: bad arguments%s");
    if ((N + SCATTER_CHUNK - 1) / SCATTER_CHUNK > 65535) return fail(MP_ERR_ARG, "mp_scatter_f32: segments longer than 2^28 samples%s");
    hipLaunchKernelGGL(scatter_kernel<false>, dim3((unsigned)B, (unsigned)((N + SCATTER_CHUNK - 1) / SCATTER_CHUNK)), dim3(1024), 0,
                       static_cast<hipStream_t>(stream), nullptr, atom, batch, lag, gain, n_events, dict_unit,
                       L, out, N);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_scatter_rows_f32(const float *rows, const int64_t *batch, const int64_t *lag, int64_t n_events,
                        int64_t L, float *out, int64_t B, int64_t N, void *stream) {
    if (n_events == 0 || B == 0) return MP_OK;
    if (!rows || !batch || !lag || !out || L <= 0 || N <= 0 || B < 0 || n_events < 0)
        return fail(MP_ERR_ARG, "mp_scatter_rows_f32: bad arguments%s");
    if ((N + SCATTER_CHUNK - 1) / SCATTER_CHUNK > 65535) return fail(MP_ERR_ARG, "mp_scatter_rows_f32: segments longer than 2^28 samples%s");
    hipLaunchKernelGGL(scatter_kernel<true>, dim3((unsigned)B, (unsigned)((N + SCATTER_CHUNK - 1) / SCATTER_CHUNK)), dim3(1024), 0,
                       static_cast<hipStream_t>(stream), rows, nullptr, batch, lag, nullptr, n_events, nullptr,
                       L, out, N);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_gather_sum_f32(const float *x, int64_t B, int64_t N, const int64_t *batch, const int64_t *lag,
                      int64_t n_events, int64_t L, double *out, void *stream) {
    if (!x || !out || L <= 0 || N <= 0 || B <= 0 || n_events < 0 || (n_events > 0 && (!batch || !lag)))
        return fail(MP_ERR_ARG, "mp_gather_sum_f32: bad arguments%s");
    hipLaunchKernelGGL(gather_sum_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, N, batch, lag, n_events, L, out);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_gather_sum_groups_f32(const float *x, int64_t B, int64_t N, const int64_t *batch, const int64_t *lag,
                             const int64_t *offsets, int64_t n_groups, int64_t L, double *out, void *stream) {
    if (n_groups == 0) return MP_OK;
    if (!x || !out || !offsets || L <= 0 || N <= 0 || B <= 0 || n_groups < 0 || n_groups > 65535)
        return fail(MP_ERR_ARG, "mp_gather_sum_groups_f32: bad arguments%s");
    hipLaunchKernelGGL(gather_sum_groups_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)n_groups), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, N, batch, lag, offsets, L, out);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_dictionary_level_addback_sum_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, int64_t L,
                                        const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch,
                                        const int64_t *ev_lag, const float *ev_rows, const int *overlap, double *acc,
                                        void *stream) {
    if (n_groups == 0) return MP_OK;
    if (!residual || !sparse_zeroed || !offsets || !acc || B <= 0 || N <= 0 || L <= 0 || n_groups < 0)
        return fail(MP_ERR_ARG, "mp_dictionary_level_addback_sum_f32: bad arguments%s");
    hipLaunchKernelGGL(dictionary_level_addback_sum_kernel, dim3((unsigned)n_groups), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       residual, sparse_zeroed, N, L, offsets, ev_batch, ev_lag, ev_rows, overlap, acc);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_dictionary_level_subtract_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, int64_t L,
                                     const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch,
                                     const int64_t *ev_lag, const float *ev_norm, const int *overlap,
                                     const float *new_atoms, void *stream) {
    if (n_groups == 0) return MP_OK;
    if (!residual || !sparse_zeroed || !offsets || !new_atoms || B <= 0 || N <= 0 || L <= 0 || n_groups < 0)
        return fail(MP_ERR_ARG, "mp_dictionary_level_subtract_f32: bad arguments%s");
    hipLaunchKernelGGL(dictionary_level_subtract_kernel, dim3((unsigned)n_groups), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       residual, sparse_zeroed, N, L, offsets, ev_batch, ev_lag, ev_norm, overlap, new_atoms);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_dictionary_update_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, float *dict_work, int64_t A,
                             int64_t L, const int64_t *order, const int64_t *offsets, int64_t n_groups,
                             const int64_t *ev_batch, const int64_t *ev_lag, const float *ev_rows,
                             const float *ev_norm, float eps, const int *overlap, void *stream) {
    if (n_groups == 0) return MP_OK;
    if (!residual || !sparse_zeroed || !dict_work || !order || !offsets || !ev_batch || !ev_lag || !ev_rows ||
        !ev_norm || B <= 0 || N <= 0 || A <= 0 || L <= 0 || n_groups < 0)
        return fail(MP_ERR_ARG, "mp_dictionary_update_f32: bad arguments%s");
    if ((size_t)L * sizeof(float) > 150 * 1024) return fail(MP_ERR_UNSUPPORTED, "mp_dictionary_update_f32: atom too long for LDS%s");
    // the new atom, and what is left of 144 KiB for the windows of an atom's events
    const int64_t win_cap = overlap ? (int64_t)((144 * 1024 - (size_t)L * sizeof(float)) / sizeof(float)) : 0;
    const size_t lds = ((size_t)L + (size_t)(win_cap > 0 ? win_cap : 0)) * sizeof(float);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dictionary_update_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(dictionary_update_kernel, dim3(1), dim3(1024), lds, static_cast<hipStream_t>(stream), residual,
                       sparse_zeroed, N, dict_work, L, order, offsets, n_groups, ev_batch, ev_lag, ev_rows, ev_norm, eps,
                       overlap, win_cap > 0 ? win_cap : 0);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_dictionary_levels_host(const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch, const int64_t *ev_lag,
                              int64_t n_events, int64_t L, int32_t *level, int32_t *overlap, int64_t *n_levels) {
    switch (mplevels::dictionary_levels(offsets, n_groups, ev_batch, ev_lag, n_events, L, level, overlap, n_levels)) {   // (mplevels.inc: plain host C++)
        case 0: return MP_OK;
        case 2: return fail(MP_ERR_ARG, "mp_dictionary_levels_host: offsets do not cover the events%s");
        default: return fail(MP_ERR_ARG, "mp_dictionary_levels_host: bad arguments%s");
    }
}

int mp_dictionary_update_levels_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, float *dict_work,
                                    int64_t A, int64_t L, const int64_t *order, const int64_t *offsets, int64_t n_groups,
                                    const int64_t *ev_batch, const int64_t *ev_lag, const float *ev_rows,
                                    const float *ev_norm, float eps, const int *overlap, const int64_t *group_list,
                                    const int64_t *level_offsets_host, int64_t n_levels, void *stream) {
    if (n_groups == 0 || n_levels == 0) return MP_OK;
    if (!residual || !sparse_zeroed || !dict_work || !order || !offsets || !ev_batch || !ev_lag || !ev_rows || !ev_norm ||
        !overlap || !group_list || !level_offsets_host || B <= 0 || N <= 0 || A <= 0 || L <= 0 || n_groups < 0 || n_levels < 0)
        return fail(MP_ERR_ARG, "mp_dictionary_update_levels_f32: bad arguments%s");
    if (level_offsets_host[0] != 0 || level_offsets_host[n_levels] != n_groups)
        return fail(MP_ERR_ARG, "mp_dictionary_update_levels_f32: level offsets do not cover the groups%s");
    if ((size_t)L * sizeof(float) > 150 * 1024) return fail(MP_ERR_UNSUPPORTED, "mp_dictionary_update_levels_f32: atom too long for LDS%s");
    const int64_t win_cap = (int64_t)((144 * 1024 - (size_t)L * sizeof(float)) / sizeof(float));
    const size_t lds = ((size_t)L + (size_t)(win_cap > 0 ? win_cap : 0)) * sizeof(float);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(dictionary_update_level_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t lv = 0; lv < n_levels; ++lv) {
        const int64_t g0 = level_offsets_host[lv], g1 = level_offsets_host[lv + 1];
        if (g1 < g0) return fail(MP_ERR_ARG, "mp_dictionary_update_levels_f32: level offsets not ascending%s");
        if (g1 == g0) continue;
        hipLaunchKernelGGL(dictionary_update_level_kernel, dim3((unsigned)(g1 - g0)), dim3(1024), lds,
                           static_cast<hipStream_t>(stream), residual, sparse_zeroed, N, dict_work, L, order, offsets,
                           group_list + g0, ev_batch, ev_lag, ev_rows, ev_norm, eps, overlap, win_cap > 0 ? win_cap : 0);
    }
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

int mp_conv_model_backward_f32(const float *atoms, int64_t A, int64_t L, const int64_t *atom_idx,
                               const int64_t *time_idx, const float *value, int K, const float *residual_final,
                               const float *grad_channels, int windowed, int64_t B, int64_t N, float *grad_audio,
                               float *grad_rows, float *scratch, void *stream) {
    if (B == 0) return MP_OK;
    if (!atoms || !atom_idx || !time_idx || !value || !residual_final || !grad_channels || !grad_audio || !grad_rows ||
        !scratch || A <= 0 || L <= 0 || K < 0 || B < 0 || N <= 0)
        return fail(MP_ERR_ARG, "mp_conv_model_backward_f32: bad arguments%s");
    hipLaunchKernelGGL(conv_model_backward_kernel, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), atoms,
                       L, atom_idx, time_idx, value, K, residual_final, grad_channels, N, grad_audio, grad_rows, scratch,
                       windowed);
    HIP_TRY(hipGetLastError());
    return MP_OK;
}

/* test / diagnosis hook: concurrency ratio of the internal streams q0 and q1 (see stream_pair_ratio) */
int mp_audit_read(float *max_ratio, int64_t *cells, float *max_quarter_ratio, int64_t *over_bound) {
    unsigned h[4] = {0, 0, 0, 0};
    const unsigned zero[4] = {0, 0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_audit), sizeof(h)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_audit), zero, sizeof(zero)));
    float r0, r2;
    memcpy(&r0, &h[0], 4);
    memcpy(&r2, &h[2], 4);
    if (max_ratio) *max_ratio = r0;
    if (cells) *cells = (int64_t)h[1];
    if (max_quarter_ratio) *max_quarter_ratio = r2;
    if (over_bound) *over_bound = (int64_t)h[3];
    return MP_OK;
}

int mp_persist_stats(uint64_t *out8 /* [16] */) {
    if (!out8) return fail(MP_ERR_ARG, "mp_persist_stats: null output%s");
    HIP_TRY(hipDeviceSynchronize());
    uint64_t nine[9];
    HIP_TRY(hipMemcpyFromSymbol(nine, HIP_SYMBOL(g_persist_stats), 9 * sizeof(uint64_t)));
    for (int i = 0; i < 8; ++i) out8[i] = nine[i];
    {   // the selects' phase ticks behind them: [8..12] acquire, scan, quarters + chains, event + window, transform + stores; [13] selects
        const uint64_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyFromSymbol(out8 + 8, HIP_SYMBOL(g_persist_phase), 8 * sizeof(uint64_t)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_persist_phase), zero, sizeof(zero)));
    }
    out8[14] = nine[8];   // screen tasks answered without a transform (lazy screen)
    return MP_OK;
}

int mp_lazy_stats(uint64_t *out2 /* [8] */) {
    if (!out2) return fail(MP_ERR_ARG, "mp_lazy_stats: null output%s");
    const uint64_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_lazy_stats), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lazy_stats), zero, sizeof(zero)));
    return MP_OK;
}

int mp_last_schedule(void) { return last_schedule; }

int mp_form_table(double *out, int capacity) {
    const double v[] = {(double)FORM.quarter_max_cells, (double)FORM.fused_min_cells, (double)FORM.persist_max_cells,
                        FORM.persist_points_table, FORM.persist_points_table_1024, FORM.persist_points_any_batch_1024,
                        (double)FORM.persist_segments_table_1024,
                        (double)FORM.sub_batch_min_segments,
                        FORM.persist_spectra_bytes, FORM.persist_points_fit, FORM.persist_points_nofit, (double)FORM.short_ratio,
                        (double)FORM.short_logm_always, (double)FORM.short_logm_small, (double)FORM.short_small_segments,
                        (double)FORM.sub_batches, (double)FORM.persist_two_per_cu_load,
                        (double)FORM.persist_fine_num / FORM.persist_fine_den, (double)FORM.persist_select_workers,
                        (double)FORM.lazy_min_steps, (double)FORM.lazy_min_tiles, (double)FORM.lazy_always_tiles,
                        (double)FORM.lazy_batch_tiles, (double)FORM.lazy_margin_persistent, (double)FORM.lazy_margin_persistent_1024,
                        (double)FORM.lazy_margin_steps, (double)FORM.persist_fine_max_logm};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    if (out)
        for (int i = 0; i < n && i < capacity; ++i) out[i] = v[i];
    return n;
}

int mp_init_streams(void *stream) {
    StreamPool *p = stream_pool(static_cast<hipStream_t>(stream));
    if (!p) return fail(MP_ERR_HIP, "mp_init_streams: could not build the internal stream pool (called under a capture?)%s");
    return p->n_concurrent;
}

float mp_stream_pair_ratio(int q0, int q1) {
    StreamPool *p = stream_pool(nullptr);
    if (!p || q0 < 0 || q1 < 0 || q0 >= MAX_GROUPS || q1 >= MAX_GROUPS) return -1.f;
    return stream_pair_ratio(p->streams[q0], p->streams[q1]);
}

}  // extern "C"

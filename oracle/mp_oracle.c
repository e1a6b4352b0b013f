/*
 * mp_oracle.c -- CPU restatement of the reference's greedy matching-pursuit path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product path (matching-pursuit_amd/) never
 * links, imports or calls it.
 *
 * What is restated (all citations relative to /root/reference):
 *   unit_norm                  modules/normalization.py:4-6
 *   direct correlation         modules/matchingpursuit.py:275-277  (== modules/conv.py:4-9)
 *   signed first-max argmax    modules/matchingpursuit.py:298-303
 *   scaled-atom subtraction    modules/matchingpursuit.py:304-307, 20-58, 326-328
 *   sparse_code loop           modules/matchingpursuit.py:269-328
 *   scatter_segments (decode)  modules/matchingpursuit.py:20-58
 *   dictionary_learning_step   modules/matchingpursuit.py:348-419
 *   local contrast norm        modules/matchingpursuit.py:284-294  (mpo_encode_lcn)
 * (sparse_feature_map :68-125, sparse_coding_loss :128-146 and the approximate branches of
 *  modules/conv.py:24-47 are restated in numpy on top of these, in oracle/mp_oracle.py.)
 *
 * Arithmetic contract (this is what the HIP kernels are held to, bit for bit):
 *   - fm[a,t] is ONE fp32 fused-multiply-add chain in ascending k:
 *         acc = 0;  for k in 0..L-1: acc = fmaf(r[t+k], d[a,k], acc)
 *     with r[] read as zero at and beyond sample N (the reference zero-pads by L, :275).
 *     The reference leaves the summation order to oneDNN; any fixed order is a valid
 *     restatement, and this one is what v_mfma_f32_32x32x2_f32 computes on gfx950.
 *   - argmax is over the SIGNED value, first occurrence in flat order a*N + t (torch.max).
 *   - subtraction is two roundings, r = r - (d*g), never an fma (the reference forms
 *     `at = d[atom] * value` first, :305) and is cropped at N, gain not renormalised.
 *   - unit_norm: sum of squares accumulated in fp64, 64 interleaved partial sums then their sum, each in
 *     ascending order (products of two floats
 *     are exact in fp64), rounded once to fp32, sqrtf, then x / (n + eps) in fp32.
 *
 * Pinned against the real reference by tests/golden/ (see tests/golden/generate_golden.py):
 * identical (atom, lag) sequences, gains/residuals within 1e-5 relative.
 *
 * Build: see oracle/Makefile  (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MPO_TBLK 64 /* lags held in registers across the k loop */

int mpo_version(void) { return 1; }

int mpo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void mpo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* modules/normalization.py:4-6 : x / (||x||_2 + eps) along the last axis */
int mpo_unit_norm(const float *d, int64_t A, int64_t L, float eps, float *out) {
    for (int64_t a = 0; a < A; ++a) {
        /* 64 interleaved partial sums (partial j: samples j, j + 64, ... in ascending order), then the partials
         * in ascending order: the order a 64-lane wavefront produces, so the GPU kernel can use all its lanes
         * and still match bit for bit */
        double s = 0.0;
        for (int j = 0; j < 64; ++j) {
            double p = 0.0;
            for (int64_t k = j; k < L; k += 64) {
                double x = (double)d[a * L + k];
                p += x * x;
            }
            s += p;
        }
        float n = sqrtf((float)s);
        float den = n + eps;
        for (int64_t k = 0; k < L; ++k) out[a * L + k] = d[a * L + k] / den;
    }
    return 0;
}

/* One row of the feature map for lags [t0, t0+MPO_TBLK): fma chain in ascending k.
 * rp is the zero-padded residual (length >= N + L + MPO_TBLK). */
static inline void corr_block(const float *rp, const float *da, int64_t L, int64_t t0, float *acc) {
    for (int j = 0; j < MPO_TBLK; ++j) acc[j] = 0.0f;
    const float *r0 = rp + t0;
    for (int64_t k = 0; k < L; ++k) {
        const float dk = da[k];
        const float *rk = r0 + k;
        for (int j = 0; j < MPO_TBLK; ++j) acc[j] = fmaf(rk[j], dk, acc[j]);
    }
}

typedef struct {
    float v;      /* best value */
    int64_t idx;  /* flat index a*N + t of the best value */
    float v2;     /* second-best value (diagnostic: near-tie gap) */
} best_t;

static inline void best_update(best_t *b, float v, int64_t idx) {
    /* torch.max: first occurrence wins => strictly greater, or equal at a lower index. */
    if (v > b->v || (v == b->v && idx < b->idx)) {
        b->v2 = b->v;
        b->v = v;
        b->idx = idx;
    } else if (v > b->v2) {
        b->v2 = v;
    }
}

/* Feature map of one segment + its argmax.  fm (A*N floats) may be NULL. */
static void correlate_argmax(const float *rp, int64_t N, const float *du, int64_t A, int64_t L,
                             float *fm, best_t *out) {
    best_t g = {-INFINITY, INT64_MAX, -INFINITY};
#pragma omp parallel
    {
        best_t loc = {-INFINITY, INT64_MAX, -INFINITY};
        float acc[MPO_TBLK];
#pragma omp for schedule(static) nowait
        for (int64_t a = 0; a < A; ++a) {
            const float *da = du + a * L;
            for (int64_t t0 = 0; t0 < N; t0 += MPO_TBLK) {
                corr_block(rp, da, L, t0, acc);
                int64_t lim = N - t0 < MPO_TBLK ? N - t0 : MPO_TBLK;
                for (int64_t j = 0; j < lim; ++j) {
                    float v = acc[j] + 0.0f; /* -0.0 -> +0.0, as torch compares them equal */
                    best_update(&loc, v, a * N + t0 + j);
                    if (fm) fm[a * N + t0 + j] = acc[j];
                }
            }
        }
#pragma omp critical
        {
            if (loc.v > g.v || (loc.v == g.v && loc.idx < g.idx)) {
                float other = g.v > loc.v2 ? g.v : loc.v2;
                g.v2 = other;
                g.v = loc.v;
                g.idx = loc.idx;
            } else {
                if (loc.v > g.v2) g.v2 = loc.v;
            }
        }
    }
    *out = g;
}

/* Dense feature map of a batch: fm[b,a,t]  (modules/conv.py:4-9 torch_conv). */
int mpo_feature_map(const float *residual, int64_t B, int64_t N, const float *du, int64_t A,
                    int64_t L, float *fm) {
    float *rp = (float *)calloc((size_t)(N + L + MPO_TBLK), sizeof(float));
    if (!rp) return -1;
    for (int64_t b = 0; b < B; ++b) {
        memcpy(rp, residual + b * N, (size_t)N * sizeof(float));
        best_t best;
        correlate_argmax(rp, N, du, A, L, fm + b * A * N, &best);
    }
    free(rp);
    return 0;
}

/*
 * sparse_code (modules/matchingpursuit.py:229-345), direct-correlation branch, C == 1.
 *   signal   [B,N]      dict_unit [A,L] (already unit_norm'ed, :254)
 *   atom/lag [B,K] int64, gain [B,K] fp32 in SELECTION order (step-major per segment)
 *   residual [B,N] out (may be NULL); top2 [B,K,2] = (best, second best) out (may be NULL)
 */
int mpo_encode(const float *signal, int64_t B, int64_t N, const float *du, int64_t A, int64_t L,
               int K, int64_t *atom, int64_t *lag, float *gain, float *residual, float *top2) {
    if (B < 0 || N <= 0 || A <= 0 || L <= 0 || K < 0) return -2;
    float *rp = (float *)calloc((size_t)(N + L + MPO_TBLK), sizeof(float));
    if (!rp) return -1;
    for (int64_t b = 0; b < B; ++b) {
        memcpy(rp, signal + b * N, (size_t)N * sizeof(float)); /* residual = signal.clone() :256 */
        for (int k = 0; k < K; ++k) {
            best_t best;
            correlate_argmax(rp, N, du, A, L, NULL, &best); /* :275-277, :298-299 */
            int64_t ai = best.idx / N;                       /* :302 */
            int64_t p = best.idx % N;                        /* :303 */
            float g = best.v;
            atom[b * K + k] = ai;
            lag[b * K + k] = p;
            gain[b * K + k] = g;
            if (top2) {
                top2[(b * K + k) * 2 + 0] = best.v;
                top2[(b * K + k) * 2 + 1] = best.v2;
            }
            /* at = d[atom] * value (:305); residual -= scatter(at) cropped to N (:326-328) */
            int64_t len = N - p < L ? N - p : L;
            const float *da = du + ai * L;
            for (int64_t i = 0; i < len; ++i) {
                float at = da[i] * g;
                rp[p + i] = rp[p + i] - at;
            }
        }
        if (residual) memcpy(residual + b * N, rp, (size_t)N * sizeof(float));
    }
    free(rp);
    return 0;
}

/*
 * sparse_code(..., local_contrast_norm=True) (modules/matchingpursuit.py:284-294): the step's event is the
 * argmax of  fm - avg_pool2d(fm, 9x9, stride 1, pad 4)  over the (A, N) map (zero padding, every window
 * divided by 81: count_include_pad), and its gain is the RAW map value there (:294).
 *   avg[a,t] = (sum over a' = a-4..a+4 ascending, t' = t-4..t+4 ascending of fm[a',t'], ONE sequential
 *              fp32 sum, cells outside the map skipped) / 81.0f     -- ATen's cpu_avg_pool2d loop order
 *   lcn[a,t] = fm[a,t] - avg[a,t]        (fp32)
 * Same outputs as mpo_encode (no top2).
 */
int mpo_encode_lcn(const float *signal, int64_t B, int64_t N, const float *du, int64_t A, int64_t L,
                   int K, int64_t *atom, int64_t *lag, float *gain, float *residual) {
    if (B < 0 || N <= 0 || A <= 0 || L <= 0 || K < 0) return -2;
    float *rp = (float *)calloc((size_t)(N + L + MPO_TBLK), sizeof(float));
    float *fm = (float *)malloc((size_t)A * N * sizeof(float));
    if (!rp || !fm) { free(rp); free(fm); return -1; }
    for (int64_t b = 0; b < B; ++b) {
        memcpy(rp, signal + b * N, (size_t)N * sizeof(float));
        for (int k = 0; k < K; ++k) {
            best_t raw;
            correlate_argmax(rp, N, du, A, L, fm, &raw);
            best_t g = {-INFINITY, INT64_MAX, -INFINITY};
#pragma omp parallel
            {
                best_t loc = {-INFINITY, INT64_MAX, -INFINITY};
#pragma omp for schedule(static) nowait
                for (int64_t a = 0; a < A; ++a) {
                    int64_t a0 = a - 4 < 0 ? 0 : a - 4, a1 = a + 5 > A ? A : a + 5;
                    for (int64_t t = 0; t < N; ++t) {
                        int64_t t0 = t - 4 < 0 ? 0 : t - 4, t1 = t + 5 > N ? N : t + 5;
                        float sum = 0.0f;
                        for (int64_t ia = a0; ia < a1; ++ia)
                            for (int64_t it = t0; it < t1; ++it) sum += fm[ia * N + it];
                        float v = fm[a * N + t] - sum / 81.0f;
                        best_update(&loc, v + 0.0f, a * N + t);
                    }
                }
#pragma omp critical
                {
                    if (loc.v > g.v || (loc.v == g.v && loc.idx < g.idx)) g = loc;
                }
            }
            int64_t ai = g.idx / N, p = g.idx % N;
            float val = fm[g.idx]; /* torch.gather(fm, index=mx) :294 */
            atom[b * K + k] = ai;
            lag[b * K + k] = p;
            gain[b * K + k] = val;
            int64_t len = N - p < L ? N - p : L;
            const float *da = du + ai * L;
            for (int64_t i = 0; i < len; ++i) {
                float at = da[i] * val;
                rp[p + i] = rp[p + i] - at;
            }
        }
        if (residual) memcpy(residual + b * N, rp, (size_t)N * sizeof(float));
    }
    free(rp);
    free(fm);
    return 0;
}

/*
 * scatter_segments (modules/matchingpursuit.py:20-58), channels == 1:
 *   out[batch[e], lag[e] + i] += rows[e, i]   for i < L, cropped to [0, N), events in order.
 */
int mpo_scatter_rows(const float *rows, const int64_t *batch, const int64_t *lag, int64_t n_events,
                     int64_t L, float *out, int64_t B, int64_t N) {
    for (int64_t e = 0; e < n_events; ++e) {
        int64_t j = batch[e], p = lag[e];
        if (j < 0 || j >= B) return -2;
        for (int64_t i = 0; i < L; ++i) {
            int64_t t = p + i;
            if (t >= 0 && t < N) out[j * N + t] += rows[e * L + i];
        }
    }
    return 0;
}

/* Decoder form: rows[e] = dict_unit[atom[e]] * gain[e]. */
int mpo_scatter(const int64_t *atom, const int64_t *batch, const int64_t *lag, const float *gain,
                int64_t n_events, const float *du, int64_t A, int64_t L, float *out, int64_t B,
                int64_t N) {
    for (int64_t e = 0; e < n_events; ++e) {
        int64_t j = batch[e], p = lag[e], a = atom[e];
        if (j < 0 || j >= B || a < 0 || a >= A) return -2;
        float g = gain[e];
        for (int64_t i = 0; i < L; ++i) {
            int64_t t = p + i;
            if (t >= 0 && t < N) out[j * N + t] += du[a * L + i] * g;
        }
    }
    return 0;
}

/*
 * dictionary_learning_step (modules/matchingpursuit.py:348-419), C == 1.
 *   d_raw [A,L] in, d_out [A,L] out (the caller's d is not modified, :365).
 * Faithful to the reference, including that `residual` here starts from the ORIGINAL
 * signal (:367) and is not the encoder's residual.
 */
int mpo_dictionary_learning_step(const float *signal, int64_t B, int64_t N, const float *d_raw,
                                 int64_t A, int64_t L, int K, float *d_out) {
    int64_t E = B * (int64_t)K;
    float *d = (float *)malloc((size_t)(A * L) * sizeof(float));
    float *res = (float *)malloc((size_t)(B * N) * sizeof(float));
    int64_t *atom = (int64_t *)malloc((size_t)(E ? E : 1) * sizeof(int64_t));
    int64_t *lag = (int64_t *)malloc((size_t)(E ? E : 1) * sizeof(int64_t));
    float *gain = (float *)malloc((size_t)(E ? E : 1) * sizeof(float));
    int64_t *order = (int64_t *)malloc((size_t)(A) * sizeof(int64_t));
    char *seen = (char *)calloc((size_t)A, 1);
    double *acc = (double *)malloc((size_t)L * sizeof(double));
    float *newatom = (float *)malloc((size_t)L * sizeof(float));
    float *anorm = (float *)malloc((size_t)(E ? E : 1) * sizeof(float));
    float *sparse = (float *)malloc((size_t)((B * N) > 0 ? (B * N) : 1) * sizeof(float));
    if (!d || !res || !atom || !lag || !gain || !order || !seen || !acc || !newatom || !anorm || !sparse)
        return -1;

    mpo_unit_norm(d_raw, A, L, 1e-8f, d);                 /* :365 */
    memcpy(res, signal, (size_t)(B * N) * sizeof(float)); /* :367 */
    int rc = mpo_encode(signal, B, N, d, A, L, K, atom, lag, gain, NULL, NULL); /* :380 */
    if (rc) return rc;

    /* instances.keys(): first-selection order, iterating steps then batch (:269, :311, :321) */
    int64_t n_order = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t b = 0; b < B; ++b) {
            int64_t a = atom[b * K + k];
            if (!seen[a]) {
                seen[a] = 1;
                order[n_order++] = a;
            }
        }
    /* torch.norm(x[3]) of a = d[atom]*value, computed BEFORE d[atom] is replaced (:408-411
     * evaluates lazily, but x[3] was materialised at encode time). */
    for (int k = 0; k < K; ++k)
        for (int64_t b = 0; b < B; ++b) {
            int64_t e = b * K + k;
            const float *da = d + atom[e] * L;
            double s = 0.0;
            for (int64_t i = 0; i < L; ++i) {
                float at = da[i] * gain[e];
                s += (double)at * (double)at;
            }
            anorm[e] = sqrtf((float)s);
        }
    /* per-event scaled atoms as materialised at encode time */
    float *ev_rows = (float *)malloc((size_t)(E ? E : 1) * (size_t)L * sizeof(float));
    if (!ev_rows) return -1;
    for (int64_t e = 0; e < E; ++e)
        for (int64_t i = 0; i < L; ++i) ev_rows[e * L + i] = d[atom[e] * L + i] * gain[e];

    for (int64_t oi = 0; oi < n_order; ++oi) { /* :391 */
        int64_t index = order[oi];
        /* sparse = scatter(inst): zeros += a, events in inst order (steps, then batch);
         * residual += sparse  (:395-396) */
        memset(sparse, 0, (size_t)(B * N) * sizeof(float));
        for (int k = 0; k < K; ++k)
            for (int64_t b = 0; b < B; ++b) {
                int64_t e = b * K + k;
                if (atom[e] != index) continue;
                int64_t p = lag[e];
                for (int64_t i = 0; i < L && p + i < N; ++i) sparse[b * N + p + i] += ev_rows[e * L + i];
            }
        for (int64_t q = 0; q < B * N; ++q) res[q] += sparse[q];
        /* new_atom = sum of residual windows (:400-401), zero beyond N (:370-371) */
        for (int64_t i = 0; i < L; ++i) acc[i] = 0.0;
        for (int k = 0; k < K; ++k)
            for (int64_t b = 0; b < B; ++b) {
                int64_t e = b * K + k;
                if (atom[e] != index) continue;
                int64_t p = lag[e];
                for (int64_t i = 0; i < L && p + i < N; ++i) acc[i] += (double)res[b * N + p + i];
            }
        for (int64_t i = 0; i < L; ++i) newatom[i] = (float)acc[i];
        mpo_unit_norm(newatom, 1, L, 1e-8f, newatom); /* :403-404 */
        memcpy(d + index * L, newatom, (size_t)L * sizeof(float)); /* :406 */
        /* sparse = scatter(new_atom * ||a||); residual = residual - sparse  (:408-415) */
        memset(sparse, 0, (size_t)(B * N) * sizeof(float));
        for (int k = 0; k < K; ++k)
            for (int64_t b = 0; b < B; ++b) {
                int64_t e = b * K + k;
                if (atom[e] != index) continue;
                int64_t p = lag[e];
                for (int64_t i = 0; i < L && p + i < N; ++i) {
                    float v = newatom[i] * anorm[e];
                    sparse[b * N + p + i] += v;
                }
            }
        for (int64_t q = 0; q < B * N; ++q) res[q] = res[q] - sparse[q];
    }
    mpo_unit_norm(d, A, L, 1e-8f, d_out); /* :417 */

    free(d); free(res); free(atom); free(lag); free(gain); free(order); free(seen);
    free(acc); free(newatom); free(anorm); free(ev_rows); free(sparse);
    return 0;
}

/*
 * mpcore.h -- C ABI of libmpcore.so, the MI355X (gfx950) greedy matching-pursuit encoder.
 *
 * The reference (JohnVinyard/matching-pursuit) has no FFI: its boundary for this path is a
 * set of Python functions over torch tensors.  Each entry point below replaces the torch
 * ops one of those functions spends its time in; the citation says which (paths relative
 * to the reference checkout).  The Python mirror of the reference surface that calls these
 * through ctypes is matching-pursuit_amd/mpcore/ (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned, contiguous memory unless stated;
 *     nothing is allocated or freed on the caller's behalf (workspace is caller-supplied,
 *     so PyTorch's caching allocator owns all memory);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream) and may be captured into a hipGraph: no host synchronisation inside.
 *     ONE exception, once per host thread and device: the first mp_encode_f32 that splits its
 *     batch over internal streams builds and tests that stream pool (a few ms, synchronises
 *     with the host) -- unless the call is being captured, in which case it stays on one
 *     stream instead.  mp_init_streams() does that step explicitly, ahead of time;
 *     mp_profile_read / mp_audit_read (measurement, debug) synchronise by definition;
 *   - return value: 0 = ok, negative = error; mp_last_error() (thread-local) says why;
 *   - all floating data is fp32; indices are int64 at the boundary (torch's index type);
 *   - C == 1 (mono) only.
 */
#ifndef MPCORE_H
#define MPCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_ERR_ARG (-1)       /* bad shape / null pointer / unsupported size            */
#define MP_ERR_WORKSPACE (-2) /* workspace too small or misaligned                      */
#define MP_ERR_HIP (-3)       /* a HIP runtime call failed (message has hipGetErrorString) */
#define MP_ERR_UNSUPPORTED (-4)

/* How the per-iteration correlation is scheduled.  All paths select bit-identical events. */
#define MP_PATH_DIRECT 0      /* full A x N correlation every iteration (what the reference
                                 recomputes each step, modules/matchingpursuit.py:275-277)  */
#define MP_PATH_FFT 1         /* FFT correlation (modules/conv.py:11-53) as an overlap-save SCREEN,
                                 the few cells that can hold the maximum re-evaluated exactly:
                                 same events as MP_PATH_DIRECT.  A segment whose screen
                                 overflowed (> 32 inexact contender cells in one step; > 64 / 128
                                 where the transforms run as two halves / four quarters) gets
                                 out_gain[b, :] = NaN: re-encode it with MP_PATH_INCREMENTAL.
                                 Atoms of up to 21782 samples (a 3 L + 190-point transform whole
                                 in LDS up to 5398, as two 2^14-point halves up to 10859, as four
                                 quarters beyond); longer atoms: MP_ERR_UNSUPPORTED           */
#define MP_PATH_INCREMENTAL 2 /* full correlation once, then only the lags an event touched
                                 ([p-L+1, p+L-1]); untouched block maxima are reused        */
#define MP_PATH_NAIVE 8       /* one-thread-per-lag fmaf chain, no MFMA: validation only    */

/* mp_encode_f32 `flags` (tuning / A-B switches; results never depend on them) */
#define MP_FLAG_NO_DMA 1 /* stage the dictionary tile through registers instead of LDS-DMA */
#define MP_FLAG_TA32 2   /* 32-atom workgroup tiles (64 KiB of LDS, two workgroups per CU): default */
#define MP_FLAG_TA64 4   /* 64-atom workgroup tiles (128 KiB of LDS, one workgroup per CU)         */
#define MP_FLAG_NO_STAGGER 16 /* do not delay odd wave slots by half a cell on incremental launches */
#define MP_FLAG_REFINE_MFMA 32 /* MP_PATH_FFT: refine contender cells on the MFMA cell code instead of VALU chains */
#define MP_FLAG_FFT_SIMPLE 64 /* MP_PATH_FFT: plain radix-4 screen kernel instead of the register radix-16 one */
/* (128 and 256 were two dominated screen variants, removed: do not reuse) */
#define MP_FLAG_FFT_UNFUSED 512 /* MP_PATH_FFT: select-A / refine / select-B as separate kernels plus the window kernel */
#define MP_FLAG_FFT_FUSED 1024  /* MP_PATH_FFT: the whole-cell one-kernel select (default only for >= 65536 cells per
                                   segment)                                                                        */
#define MP_FLAG_OVERLAP 2048    /* sub-batches on forked internal streams (joined before returning); default for
                                   MP_PATH_FFT from 48 segments of < 65536 cells up                               */
#define MP_FLAG_NO_OVERLAP 4096 /* never split the batch                                                          */
#define MP_FLAG_FFT_NO_QUARTER 8192 /* MP_PATH_FFT: segments of <= 16384 cells through scan+refine / select-B instead
                                       of the one-kernel quarter-cell select (default when the batch is split)    */
#define MP_FLAG_FFT_QUARTER 16384   /* MP_PATH_FFT: the quarter-cell select also when the batch stays on one stream */
#define MP_FLAG_FFT_PERSISTENT 65536 /* MP_PATH_FFT: steps 1 .. K-1 of the whole batch in ONE launch of resident workgroups that
                                        pull screen tasks from a queue while select workers serve the segments whose screens
                                        are complete (csrc/mppersist.inc).  Default at every batch size where it applies
                                        (no split transforms, <= 65536 cells per segment and <= 4096 atoms, 1024 <= M <= 4096, <= 2 GiB of window records); this flag
                                        asks for it at any batch size; shapes it does not cover use the other forms   */
#define MP_FLAG_FFT_NO_PERSISTENT 131072 /* MP_PATH_FFT: launch-per-step kernels (sub-batches on forked streams from 48 segments) */
#define MP_FLAG_GROUPS_SHIFT 20
#define MP_FLAG_GROUPS(n) (((n) & 7) << MP_FLAG_GROUPS_SHIFT) /* this call: n (2..4) sub-batches where the batch is split */
#define MP_FLAG_NO_PERSISTENT 8 /* one workgroup per 4 cells instead of machine-sized persistent grid */

int mp_version(void);
const char *mp_last_error(void);

/*
 * Measurement hook for bench.py (not part of the reference surface): while enabled, mp_encode_f32
 * brackets kernel launches with hipEvents on the launch stream.  every = 0: off; 1: every iteration;
 * n > 1: only iterations k with k % n == 0 (an event between two kernels idles the GPU for ~10 us, so
 * sampling keeps the timed region honest).  mp_profile_read synchronises, then returns total
 * milliseconds and span counts in ms[3] / count[3] = { full correlate, incremental correlate,
 * everything else (window transform, select, refine, subtract) }, and resets.
 * Bits 16 .. 18 of `every`: kinds to leave OUT (bit 16 + q = kind q of the list above) -- a timed region that wants its
 * dominant kernel's durations only records no spans around the selects.
 * Host pointers.  Do not enable while capturing a hipGraph.
 */
int mp_profile_enable(int every);

/* Tuning hook (process-wide; results never depend on it as long as tau stays above the transform error):
 *   MP_TUNE_TAU         the FFT screen's error bound per unit of window norm; 0 (default) = the model
 *                       tau(L, M) = (1.01 L + 4 log2 M) 2^-24: L 2^-24 bounds the fp32 chain's own rounding
 *                       rigorously, the log2 M term models the transforms (csrc/mpcore.hip::fft_tau)
 *   MP_TUNE_SCREEN_PPS  atom pairs per transform slot in the screen kernel (0 = heuristic)
 *   MP_TUNE_GROUPS      process-wide default number of sub-batches when a batch is split over forked
 *                       streams (2..4, default 4); MP_FLAG_GROUPS(n) sets it for one call only
 *   MP_TUNE_AUDIT       debug: 1 = after every FFT screen recompute the screened cells exactly and record
 *                       |screen - exact| / eps (mp_audit_read); as slow as MP_PATH_DIRECT                */
#define MP_TUNE_TAU 1
#define MP_TUNE_SCREEN_PPS 2
#define MP_TUNE_GROUPS 3
#define MP_TUNE_AUDIT 4
#define MP_TUNE_PERSIST_SHARDS 6 /* MP_FLAG_FFT_PERSISTENT: number of ticket counters the screen workers are split over
                                    (0 = heuristic: one per ~512 workers)                                          */
#define MP_TUNE_PERSIST_WORKERS 7 /* persistent form: workgroups of the launch (0 = heuristic: 2 or 3 per CU by the
                                     number of screen tasks the batch can have in flight)                          */
#define MP_TUNE_PERSIST_SELECTS 8 /* persistent form: how many of them are select workers (0 = min(segments, 64))    */
#define MP_TUNE_LAZY_REUSE 12     /* lazy screen: how often a cell's bound may be widened before the cell is screened again the next
                                     time it is dirty: 1 .. 4 (4 = no cap); 0 (default) = by form: inside the persistent launch 4
                                     up to 96 steps and 1 beyond, between launches (the launch-per-step form) 4               */
#define MP_TUNE_LAZY_MARGIN 10    /* lazy screen (mp_encode_lazy_f32): a tile is skipped when its dirty cells' widened upper
                                     bounds stay below margin x the best lower bound of the untouched blocks (0 < margin <= 1;
                                     0 = the defaults: 0.9 inside the persistent launch (0.7 at 1024-point transforms), 0.95 between launches; any value is
                                     exact, smaller = fewer skips and fewer stale contenders)                                */
#define MP_TUNE_LAZY_RADIUS 13    /* lazy screen: the run's floor is the (K + K/16 + 1)-th largest PEAK among the blocks' lower bounds after step
                                     0 -- a block counts if it is the best within this many blocks either side; 0 (default) =
                                     by atom length, 1 + ceil(max(0, L - 512) / 256); -1 = every block counts (tests: the floor comes out too high).  Smaller = more skips, and stale contenders
                                     (overflow marks) on signals whose maxima collapse within the run                          */
#define MP_TUNE_PERSIST_PRESCAN 14 /* persistent form: a select worker that holds an entry whose screen is still running scans the
                                     blocks that screen does not touch -- and refines their contenders -- meanwhile (1, default; 0: off) */
#define MP_TUNE_LAZY_FORCE 16      /* TIMING EXPERIMENTS ONLY -- the events are WRONG while it is set: the launch-per-step lazy
                                     screen's tile masks are drawn at random (1.p: every (segment, tile) skipped with probability
                                     p; 2.p: every segment skips all its tiles with probability p; 0, default: off); refused unless the process has
                                     MP_ALLOW_WRONG_RESULTS=1 in its environment.  How the
                                     screen's time follows the share and the pattern of skipped workgroups: DESIGN.md 4d       */
#define MP_TUNE_PERSIST_FINE 18    /* persistent form: how a tile quarter's four atom pairs are walked -- 1 = by one slot of a workgroup (a task is
                                     four transforms long), 2 = by two neighbouring slots, two pairs each (twice the tasks, half as
                                     long: what small batches want), 0 (default) = 2 while every finer task still finds a workgroup
                                     of its own.  Same results either way; 4096-point transforms have one slot per workgroup       */
#define MP_TUNE_LAZY_COMPACT 17    /* launch-per-step lazy screen: 1 (default) = a masked screen launch runs from the masks' compacted work list
                                     (one small kernel per step builds it); 0 = every workgroup of the full grid looks its mask up and
                                     returns if it is set.  Same results either way                                              */
#define MP_TUNE_CLEAR_MEMSET 15    /* debug: 1 = the encode's clears are hipMemsetAsync calls instead of one kernel launch (what a
                                     stream capture makes of memset nodes: scripts/graph_memset_repro.py); 0 (default).
                                     Replayed from a hipGraph those memset nodes leave WRONG events on this runtime, so 1 is refused
                                     unless the process has MP_ALLOW_WRONG_RESULTS=1 in its environment, like MP_TUNE_LAZY_FORCE   */
int mp_tune(int key, double value);

/* The form table: every threshold by which MP_PATH_FFT picks its form (one launch / launch per step, which select, sub-
 * batches, when the lazy screen pays), as doubles in the order of csrc/mpcore.hip::FormTable -- the one place they live;
 * the host side reads them from here (mpcore/_native.py::form_table, lazy_pays).  Writes up to `capacity` values to `out`
 * (may be NULL) and returns how many there are. */
int mp_form_table(double *out, int capacity);
int mp_profile_read(double *ms, int64_t *count);

/* Device bytes mp_encode_f32 needs in `workspace` for this problem (0 on bad arguments).
 * What is in it: the padded residual [B][~N + L], the dictionary image, one 8-byte key and one 4-byte
 * bound per cell ([B][N / 64][A / 32]: 12 B per 2048 map values); MP_PATH_FFT adds the pair spectra (A / 2 transforms of M
 * points x 8 B: 4 MiB at 512 x 512, 134 MB at 4096 x 2048), the full pass's window spectra ([B][ceil(N / V)][M] x 8 B),
 * 16 B of quarter maxima per cell for segments of up to 16384 cells (65536 where the persistent form applies) -- and, where the shape takes the persistent form
 * (1024- to 4096-point transforms, <= 65536 cells per segment, K >= 2), ONE WINDOW RECORD PER SEGMENT AND STEP:
 * B (K - 2) (M + 16) x 8 B -- 62 MiB at the headline shape, 0.54 GB for 128 segments x 256 steps of 2048-point
 * transforms, capped at 2 GiB (a batch whose records would pass the cap runs launch per step and gets none).  The figure
 * depends on K for that reason.  All of it is scratch: nothing in it outlives the call. */
size_t mp_workspace_bytes(int64_t B, int64_t N, int64_t A, int64_t L, int K, int path);

/*
 * unit_norm along the last axis: out = d / (||d||_2 + eps).
 * Replaces modules/normalization.py:4-6 as called at modules/matchingpursuit.py:254,365,417.
 * d, out: [A, L].  out may alias d.
 */
int mp_unit_norm_f32(const float *d, int64_t A, int64_t L, float eps, float *out, void *stream);

/*
 * K steps of greedy matching pursuit on a batch of independent segments.
 * Replaces the loop body of sparse_code, modules/matchingpursuit.py:269-328: correlation of
 * the running residual with every atom (:275-277), signed first-max argmax over atom x lag
 * (:298-303), scaled-atom subtraction cropped at N (:304-307, :326-328).
 *
 *   signal     [B, N]   in   (not modified; the reference clones it, :256)
 *   dict_unit  [A, L]   in   unit-normed dictionary (mp_unit_norm_f32)
 *   out_atom   [B, K]   out  atom index of step k of segment b   (selection order)
 *   out_lag    [B, K]   out  lag (sample position) of that event
 *   out_gain   [B, K]   out  value of the feature map at the argmax (the event's gain)
 *   out_residual [B, N] out  residual after K steps, or NULL
 *   workspace           >= mp_workspace_bytes(...) bytes, 256-byte aligned
 * Asynchronous on `stream`.  MP_PATH_FFT may run parts of a large batch on internal streams forked from and
 * joined back into `stream` by events (MP_FLAG_NO_OVERLAP: never); the caller sees ordinary stream order.
 */
int mp_encode_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A,
                  int64_t L, int K, int path, int flags, int64_t *out_atom, int64_t *out_lag,
                  float *out_gain, float *out_residual, void *workspace, size_t workspace_bytes,
                  void *stream);

/*
 * The analysis loop of the reference's gradient-trained model, mp.py:54-66 (MatchingPursuit.forward):
 * per step, spec = CONVOLUTION of the residual with the raw atoms cropped to N
 * (modules/transfer.py:548-569 fft_convolve), top-1 over atom x time (modules/sparse.py:46-89 with
 * n_to_keep = 1), and residual -= value^2 * atom placed at that time (mp.py:61-64: the value enters once
 * through the one-hot atom selector and once through the time impulse).
 *   signal [B, N], atoms [A, L] RAW (not normalised), out_atom/out_time [B, K], out_value [B, K] = the
 *   feature-map value v of each step (the event channel is v^2 * atom), out_residual [B, N] or NULL.
 * Same schedules (`path`), workspace and conventions as mp_encode_f32; values are exact fma chains.
 */
int mp_encode_conv_f32(const float *signal, int64_t B, int64_t N, const float *atoms, int64_t A,
                       int64_t L, int K, int path, int flags, int64_t *out_atom, int64_t *out_time,
                       float *out_value, float *out_residual, void *workspace, size_t workspace_bytes,
                       void *stream);

/*
 * mp_encode_f32 with the reference's local-contrast-norm selection rule,
 * sparse_code(..., local_contrast_norm=True), modules/matchingpursuit.py:284-294 (and
 * dictionary_learning_step(local_constrast_norm=True), :361-363): each step selects the argmax of
 *   fm - avg_pool2d(fm, (9, 9), stride 1, padding 4)      over the dense (A, N) map
 * (zero padding, every window divided by 81) and reports the RAW map value there as the gain (:294).
 * The box mean is one sequential fp32 sum in row-major window order, as ATen computes it on the CPU, so
 * selections are bit-identical to the reference's.  The dense map lives in the workspace -- B x ceil(N / 64) x ceil(A / 32)
 * cells of 32 atoms x 64 lags, 8 KiB each, in the matrix core's accumulator order (about 4 B A N bytes: size the batch
 * accordingly) -- and only the 64-lag blocks an event dirties are recomputed per step (exactly: every value under a 9 x 9
 * box enters the rule).
 * Arguments as mp_encode_f32; workspace >= mp_lcn_workspace_bytes(...), 256-byte aligned; B <= 65535.
 */
size_t mp_lcn_workspace_bytes(int64_t B, int64_t N, int64_t A, int64_t L, int K);
int mp_encode_lcn_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A, int64_t L,
                      int K, int64_t *out_atom, int64_t *out_lag, float *out_gain, float *out_residual,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * Dense feature map fm[b, a, t] = sum_k r[b, t+k] * d[a, k]  (r zero beyond N), t in [0, N).
 * Replaces F.conv1d(F.pad(residual,(0,L)), d.view(A,1,L))[..., :N]
 * (modules/matchingpursuit.py:275-277, :90-92; modules/conv.py:4-9).  Serves the hooks that
 * need the dense map (visit_key_point, extract_atom_embedding, sparse_feature_map).
 *   residual [B, N], dict_unit [A, L], fm [B, A, N]; workspace as for mp_encode_f32(K = 0).
 */
int mp_feature_map_f32(const float *residual, int64_t B, int64_t N, const float *dict_unit,
                       int64_t A, int64_t L, float *fm, void *workspace, size_t workspace_bytes,
                       void *stream);

/*
 * Test / diagnosis hook.  mp_encode_f32 runs sub-batches on internal streams that a spin test at their creation
 * showed running kernels side by side (streams multiplexed onto one hardware queue would serialise them).  This
 * repeats the test for internal streams q0, q1 in [0, 4): elapsed(two 40 us spins, one per stream) / elapsed(one
 * spin) -- about 1 side by side, about 2 one after the other; negative on error.  Synchronises with the host.
 */
float mp_stream_pair_ratio(int q0, int q1);

/* Build and test the calling thread's internal stream pool on the current device now (host-synchronising, a
 * few ms; idempotent) instead of inside the first sub-batched mp_encode_f32.  Call it before capturing an
 * encode into a hipGraph if the graph should contain sub-batches.  Returns the number of internal streams seen
 * to run side by side (>= 1; sub-batches need >= 2), or a negative error (e.g. `stream` is being captured). */
int mp_init_streams(void *stream);

/*
 * mp_encode_f32 with the dictionary's COHERENCE TABLE (the lazy screen of MP_PATH_FFT's persistent form).
 * coherence[a * NAT + t], NAT = ceil(A / 32): an upper bound on |sum_j d_a[j] d_b[j + s]| over the 32 atoms b of tile t and
 * all shifts s, for the unit-norm dictionary passed (the caller computes it -- e.g. from mp_feature_map_f32 of the atoms --
 * and may keep it as long as the dictionary does not change; it must be an upper bound in exact arithmetic).  After an
 * event (a, g) a cell changes by at most |g| coherence: tiles none of whose dirty cells can then reach the best lower bound
 * of the untouched blocks keep their (widened) bounds and skip their transforms.  Same events bit for bit; NULL, or any
 * form other than the persistent one, is mp_encode_f32.  (Not for the convolution model.)
 */
int mp_encode_lazy_f32(const float *signal, int64_t B, int64_t N, const float *dict_unit, int64_t A, int64_t L, int K,
                       int path, int flags, const float *coherence, int64_t *out_atom, int64_t *out_lag, float *out_gain,
                       float *out_residual, void *workspace, size_t workspace_bytes, void *stream);

/* The coherence table mp_encode_lazy_f32 takes, computed on the device: out[A][ceil(A / 32)] (one full-pass FFT screen of
 * the atoms against the dictionary: ~0.3 ms at 512 x 512, ~90 ms at 4096 x 2048).  Exists where the lazy screen does
 * (1024- to 8192-point transforms: 63 <= L <= 2667); mp_coherence_workspace_bytes returns 0 otherwise.  Which encodes use
 * the table: the persistent form (its select decides inside the launch), and the launch-per-step form whose select is
 * the fused whole-cell kernel with block summaries (>= 65536 cells per segment, or > 16384 with L <= 512 -- BASELINE
 * configs[3]'s shape): there the select leaves a tile mask the next screen launch obeys.  Other forms ignore it. */
size_t mp_coherence_workspace_bytes(int64_t A, int64_t L);
int mp_coherence_f32(const float *dict_unit, int64_t A, int64_t L, float *out, void *workspace, size_t workspace_bytes,
                     void *stream);

/* Which form the calling thread's last mp_encode_f32 / mp_encode_conv_f32 took: -1 = the persistent form (step 0, then
 * one launch for steps 1 .. K-1), 1 = one kernel sequence per step on the caller's stream, n >= 2 = n sub-batches on
 * forked internal streams; 0 before the first encode.  No device work. */
int mp_last_schedule(void);

/* Debug: statistics of the last MP_FLAG_FFT_PERSISTENT launch on the current device, summed over its workgroups --
 * out16[0..2] = 100 MHz wall-clock ticks spent idle (polling the queue), in screen tasks, in selects; [3..5] = tasks,
 * selects, polls; [6] = error flag, [7] = segments finished; [8..12] = ticks of the selects' phases (acquire, scan,
 * quarters + chains, event + next window, transform + stores), [13] = selects counted, [14] = screen tasks answered without
 * a transform (lazy screen).  Synchronises the device. */
int mp_persist_stats(uint64_t *out16);

/* Debug: the lazy screen of the launch-per-step form since the last read, summed over encodes -- out8[0] = (segment,
 * tile) screens skipped (their dirty cells kept widened bounds), out8[1] = (segment, tile) decisions made; selects that
 * decided nothing because out8[2] a static condition failed (an atom cropped at the segment's end, a contender overflow),
 * out8[3] more than two contender cells had been refined (near-ties), out8[4] the run's floor is 0, out8[5] there is no
 * clean lower bound or the next window is all zeros.  Beside them, how much exact refinement the screens left to do:
 * out8[6] = contender cells refined (whole cells, 32 chains each) by the launch-per-step fused select, out8[7] = contender
 * quarter-cells (8 chains each) refined by the persistent form's selects.  Synchronises the device; resets the counters. */
int mp_lazy_stats(uint64_t *out8);

/* Debug (MP_TUNE_AUDIT): the largest |screen - exact| / eps over all cells audited since the last read, their
 * number, the same for quarter-cell maxima, and how many exceeded 1 (must be 0).  Synchronises the device;
 * resets the counters.  Any output pointer may be NULL. */
int mp_audit_read(float *max_ratio, int64_t *cells, float *max_quarter_ratio, int64_t *over_bound);

/*
 * Test hook: batched complex FFT of 2^log2_m points (8 <= log2_m <= 14), unscaled -- the transforms
 * MP_PATH_FFT is built on.  inverse = 0: forward, 1: inverse (radix-4 Stockham in LDS: window and
 * dictionary spectra), 2: inverse through the screen's mixed-radix register transform (log2_m >= 10).
 * in/out: [batch, 2^log2_m] interleaved (re, im) fp32; workspace >= 8 * 2^log2_m bytes (twiddle table).
 */
int mp_fft_c2c_f32(const float *in, float *out, int log2_m, int64_t batch, int inverse, void *workspace,
                   void *stream);

/*
 * Decoder: out[batch[e], lag[e] + i] += dict_unit[atom[e], i] * gain[e], i < L, cropped to
 * [0, N); events of one segment are applied in array order (bitwise reproducible).
 * Replaces scatter_segments over encoder events, modules/matchingpursuit.py:20-58.
 *   atom, batch, lag [n_events] int64; gain [n_events]; out [B, N] (accumulated into).
 */
int mp_scatter_f32(const int64_t *atom, const int64_t *batch, const int64_t *lag, const float *gain,
                   int64_t n_events, const float *dict_unit, int64_t A, int64_t L, float *out,
                   int64_t B, int64_t N, void *stream);

/*
 * General scatter_segments: out[batch[e], lag[e] + i] += rows[e, i]  (rows [n_events, L]).
 * Replaces modules/matchingpursuit.py:20-58 for arbitrary event payloads (e.g. the
 * re-scaled new atoms of dictionary_learning_step, :408-414).
 */
int mp_scatter_rows_f32(const float *rows, const int64_t *batch, const int64_t *lag,
                        int64_t n_events, int64_t L, float *out, int64_t B, int64_t N,
                        void *stream);

/*
 * gather_segments + sum over instances (modules/matchingpursuit.py:369-378, :400-401):
 *   out[i] = sum_e x[batch[e], lag[e] + i]   i < L, x read as zero beyond N,
 * accumulated in fp64 in event order, rounded once to fp32.  This [L] vector is the only
 * cross-segment quantity of dictionary_learning_step, i.e. what a multi-GPU run all-reduces.
 */
int mp_gather_sum_f32(const float *x, int64_t B, int64_t N, const int64_t *batch,
                      const int64_t *lag, int64_t n_events, int64_t L, double *out, void *stream);

/*
 * The same for n_groups groups of events in one launch: group g owns events [offsets[g] - offsets[0],
 * offsets[g + 1] - offsets[0]) of batch / lag (`offsets` may be a slice of a longer table: only differences count),
 * out[g * L + i] = its window sum, fp64.  One dependency level of the multi-GPU dictionary_learning_step
 * (modules/matchingpursuit.py:391-415 by levels: atoms of one level share no sample, so their add-backs, window sums
 * and re-subtractions commute) -- what the ranks all-reduce is this ONE [n_groups, L] matrix per level instead of one
 * [L] vector per atom.  n_groups <= 65535.
 */
int mp_gather_sum_groups_f32(const float *x, int64_t B, int64_t N, const int64_t *batch, const int64_t *lag,
                             const int64_t *offsets, int64_t n_groups, int64_t L, double *out, void *stream);

/*
 * One dependency level of the multi-rank dictionary_learning_step in two launches around the level's all-reduce
 * (replaces, per level, modules/matchingpursuit.py:395-396 + :400-401 and :408-415 for all atoms of the level at once;
 * the single-device form of the same level is inside mp_dictionary_update_levels_f32).  Group g (one used atom) owns the
 * events [offsets[g], offsets[g + 1]) of the event arrays -- absolute positions: `offsets` is the level's slice of the
 * whole table, the event arrays are passed whole -- all of THIS rank's segments (possibly none); overlap[g] != 0
 * (or overlap == NULL) = two of its events may share a sample: they are staged in `sparse_zeroed` ([B, N], zero on entry
 * and on return) event after event.
 *   addback_sum:  residual += the events' rows;  acc[g, :] = fp64 sum over the events of residual[batch, lag : lag + L]
 *   subtract:     residual -= new_atoms[g, :] * ev_norm[e]  for every event e of g   (new_atoms [n_groups, L])
 * The caller all-reduces acc over the ranks, normalises (mp_unit_norm_f32) and passes the result back as new_atoms.
 */
int mp_dictionary_level_addback_sum_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, int64_t L,
                                        const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch,
                                        const int64_t *ev_lag, const float *ev_rows, const int *overlap, double *acc,
                                        void *stream);
int mp_dictionary_level_subtract_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, int64_t L,
                                     const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch,
                                     const int64_t *ev_lag, const float *ev_norm, const int *overlap,
                                     const float *new_atoms, void *stream);

/*
 * Backward pass of mp_encode_conv_f32, i.e. of the analysis loop of the reference's gradient-trained model
 * (mp.py:54-66: what loss.backward() at mp.py:104 does to it), in one launch: given the events of the forward pass
 * (atom_idx, time_idx, value [B, K]), the residual it ended with [B, N] and the gradient arriving at the K event
 * channels [B, K, N] (windowed = 0) -- or only at each channel's own support, [B, K, L] with entry j of event i
 * belonging to sample time_idx + j (windowed = 1: a loss that works on the events never forms the dense
 * channels) -- it returns grad_audio [B, N] and, per event, the row grad_rows [B, K, L] to be added to the
 * gradient of atoms[atom_idx] (the caller sums rows of equal atoms: one index_add).  scratch: [B, N] floats.
 */
int mp_conv_model_backward_f32(const float *atoms, int64_t A, int64_t L, const int64_t *atom_idx,
                               const int64_t *time_idx, const float *value, int K, const float *residual_final,
                               const float *grad_channels, int windowed, int64_t B, int64_t N, float *grad_audio,
                               float *grad_rows, float *scratch, void *stream);

/*
 * The atom-by-atom update loop of dictionary_learning_step (modules/matchingpursuit.py:391-415) in one
 * launch.  Events are grouped by atom: group g (atom order[g], in first-selection order) owns events
 * [offsets[g], offsets[g+1]) of ev_batch / ev_lag / ev_rows [E, L] (= d[atom] * value as materialised at
 * encode time) / ev_norm [E] (= ||row||).  residual [B, N] starts as the ORIGINAL signal (:367) and is
 * updated in place; dict_work [A, L] receives the new unit-norm atoms; sparse_zeroed is a [B, N] scratch that
 * must be zero on entry and is zero again on return.  overlap [n_groups] or NULL: overlap[g] == 0 promises that
 * no two events of group g share a sample (same segment, lags less than L apart), which lets the kernel apply
 * the group's events all at once (same arithmetic, an eighth of the barriers); NULL or nonzero: one by one.
 * Single-device form; a multi-GPU run needs the per-atom all-reduce of mp_gather_sum_f32 between the two halves
 * and keeps the step-by-step form.
 */
int mp_dictionary_update_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, float *dict_work, int64_t A,
                             int64_t L, const int64_t *order, const int64_t *offsets, int64_t n_groups,
                             const int64_t *ev_batch, const int64_t *ev_lag, const float *ev_rows,
                             const float *ev_norm, float eps, const int *overlap, void *stream);

/*
 * The same loop spread over the chip (single device).  An atom's update touches only the samples under its own
 * events, so it depends on an earlier atom only where their events share a sample.
 *
 * mp_dictionary_levels_host -- HOST pointers, no device work: from the grouped events (offsets [n_groups + 1],
 * ev_batch / ev_lag [n_events], as above but in host memory) computes level[g] = 0 for a group that overlaps no
 * earlier group, else 1 + the highest level among the earlier groups it overlaps; overlap[g] = 1 if two events of
 * group g itself share a sample; *n_levels = highest level + 1.  O(E log E + overlapping pairs).
 *
 * mp_dictionary_update_levels_f32 -- as mp_dictionary_update_f32, one launch per level and one workgroup per
 * atom: group_list (DEVICE, [n_groups]) lists the groups sorted by level (ascending group index inside a level),
 * level_offsets_host (HOST, [n_levels + 1]) delimits the levels in it; overlap is the device copy of the array
 * above.  Every sample sees the same operations in the same order as in the one-launch form: bit-identical.
 */
int mp_dictionary_levels_host(const int64_t *offsets, int64_t n_groups, const int64_t *ev_batch, const int64_t *ev_lag,
                              int64_t n_events, int64_t L, int32_t *level, int32_t *overlap, int64_t *n_levels);
int mp_dictionary_update_levels_f32(float *residual, float *sparse_zeroed, int64_t B, int64_t N, float *dict_work,
                                    int64_t A, int64_t L, const int64_t *order, const int64_t *offsets, int64_t n_groups,
                                    const int64_t *ev_batch, const int64_t *ev_lag, const float *ev_rows,
                                    const float *ev_norm, float eps, const int *overlap, const int64_t *group_list,
                                    const int64_t *level_offsets_host, int64_t n_levels, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCORE_H */

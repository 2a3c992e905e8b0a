/*
 * mmwgpu.h -- C ABI of libmmwgpu.so: MI355X (gfx950) range-Doppler-angle + CFAR hot path.
 *
 * Drop-in boundary for davidmhunt/mmwave_radar_processing.  The reference has no FFI
 * (SURVEY.md section 8b): its boundary is the Python class protocol
 * `processor.process(adc_cube, **kw)` / `detector.detect(x)`.  This header is what a
 * ctypes binding for that protocol binds; every entry point cites the reference
 * call (file:line, relative to mmwave_radar_processing/) whose NumPy arithmetic it replaces.
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every function returns an int
 *     status (MMW_OK == 0, negative == error); mmw_last_error() gives the text
 *     of the calling thread's last failure.
 *   - complex64 = interleaved (re, im) float pairs.  Cube layout is the reference's:
 *     [frame][virtRx V][sample S][chirp C], C-order, chirp fastest
 *     (processors/_processor.py:58).
 *   - Pointers named d_* are DEVICE pointers obtained from mmw_malloc; h_* are host.
 *   - All compute entry points enqueue on the context's HIP stream and return
 *     without synchronising unless they take a host output pointer; call
 *     mmw_sync() before reading device results through mmw_memcpy_d2h (which
 *     synchronises itself).
 *   - One context per host thread; contexts are not thread-safe.
 */
#ifndef MMWGPU_H
#define MMWGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMW_OK                 0
#define MMW_ERR_INVALID       -1   /* bad argument (shape, null pointer, unsupported size)   */
#define MMW_ERR_HIP           -2   /* HIP runtime error, see mmw_last_error()               */
#define MMW_ERR_NOMEM         -3   /* device allocation failed                              */
#define MMW_ERR_TRUNCATED     -4   /* detection output capacity too small; counts are exact */
#define MMW_ERR_UNSUPPORTED   -5   /* valid request this build has no kernel for            */

typedef struct mmw_ctx mmw_ctx;

/* CFAR kinds: string keys of detectors/detector_registry.py:15-27 */
#define MMW_CFAR_CA 0
#define MMW_CFAR_OS 1
#define MMW_CFAR_GO 2
#define MMW_CFAR_SO 3

/* ---------------------------------------------------------------- lifecycle */
const char *mmw_version(void);
/* Bumped whenever an exported signature changes: a binding checks it at load (the argtypes of a ctypes binding are
 * hard-coded, so a library of another revision would reinterpret ints as device pointers). */
#define MMWGPU_ABI_VERSION 5
int mmw_abi_version(void);
const char *mmw_last_error(void);
int mmw_device_count(int *count);
/* Device name and gcnArchName into caller buffers (for fail-loud checks and reports). */
int mmw_device_info(int device, char *name, int name_len, char *arch, int arch_len,
                    int *num_cu, size_t *total_mem);
int mmw_ctx_create(mmw_ctx **out, int device);
int mmw_ctx_destroy(mmw_ctx *ctx);
int mmw_sync(mmw_ctx *ctx);

/* ---------------------------------------------------------------- memory */
int mmw_malloc(mmw_ctx *ctx, void **d_ptr, size_t bytes);
int mmw_free(mmw_ctx *ctx, void *d_ptr);
int mmw_memcpy_h2d(mmw_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int mmw_memcpy_d2h(mmw_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int mmw_memset(mmw_ctx *ctx, void *d_dst, int value, size_t bytes);

/* ---------------------------------------------------------------- host streaming
 * For frame loops whose cubes arrive on the host (the reference's scripts/test_vel_estimation.py:145-151): pinned staging
 * blocks, asynchronous copies on the compute queue or on a separate copy queue, and events to order the two, so that
 * chunk k + 1 is uploaded while chunk k is processed (batch.FramePipeline.stream).
 *   mmw_host_alloc / mmw_host_free   pinned host memory owned by the context
 *   mmw_memcpy_async                 to_host = 0: host -> device, 1: device -> host; queue = MMW_QUEUE_*; returns at once
 *   mmw_event_create / _destroy / _record(queue) / mmw_queue_wait_event(queue, event) / mmw_event_sync(event) */
#define MMW_QUEUE_COMPUTE 0
#define MMW_QUEUE_COPY    1
int mmw_host_alloc(mmw_ctx *ctx, void **h_ptr, size_t bytes);
int mmw_host_free(mmw_ctx *ctx, void *h_ptr);
int mmw_memcpy_async(mmw_ctx *ctx, void *dst, const void *src, size_t bytes, int to_host, int queue);
int mmw_event_create(mmw_ctx *ctx, void **event);
int mmw_event_destroy(mmw_ctx *ctx, void *event);
int mmw_event_record(mmw_ctx *ctx, void *event, int queue);
int mmw_queue_wait_event(mmw_ctx *ctx, int queue, void *event);
int mmw_event_sync(mmw_ctx *ctx, void *event);

/* ---------------------------------------------------------------- timing (HIP events on the ctx stream) */
int mmw_timer_start(mmw_ctx *ctx);
int mmw_timer_stop(mmw_ctx *ctx, float *elapsed_ms);   /* synchronises on the stop event */

/* ---------------------------------------------------------------- input staging
 * mmw_synth_cubes: fill d_cubes[n_frames][V][S][C] complex64 with the synthetic
 *   point-target + noise workload directly in HBM (counter-based RNG; integer-valued
 *   I/Q like SURVEY.md 8d).  Benchmark input only -- parity tests download these
 *   cubes and hand the SAME bytes to the oracle.
 * mmw_virtual_array_reformat: raw [F][num_rx][S][num_tx*loops] -> [F][num_rx*num_tx][S][loops]
 *   replaces VirtualArrayReformatter.process (processors/virtual_array_reformater.py:44-65). */
int mmw_synth_cubes(mmw_ctx *ctx, void *d_cubes, int n_frames, int V, int S, int C,
                    uint64_t seed0, int num_targets, float noise_sigma);
int mmw_virtual_array_reformat(mmw_ctx *ctx, const void *d_raw, void *d_virt, int n_frames,
                               int num_rx, int num_tx, int S, int loops);
/* mmw_virtual_array_reformat_i16: the same from int16 I/Q samples, raw [F][num_rx][S][num_tx*loops][2] (I, Q) ->
 *   complex64 [F][num_rx*num_tx][S][loops] (half the bytes over PCIe and HBM).  NO UPSTREAM ORACLE for this sample
 *   layout: the reference receives complex cubes from the cpsl_datasets reader, which is not part of its tree
 *   (SURVEY.md F3); the layout here is "the raw cube of mmw_virtual_array_reformat with int16 I/Q pairs". */
int mmw_virtual_array_reformat_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_virt, int n_frames,
                                   int num_rx, int num_tx, int S, int loops);

/* ---------------------------------------------------------------- FFT chain
 * mmw_range_doppler: d_out[F][V][S][C] c64 = fftshift_C( FFT_S FFT_C( hann(S) hann(C) x ) )
 *   replaces RangeDopplerProcessor.process (processors/range_doppler_resp.py:49-110) and
 *   RangeDopplerDetector._compute_range_doppler_response (range_doppler_detection/range_doppler_detector.py:62-80).
 *   If d_mag_f32 != NULL also writes |out| [F][V][S][C] float32 (return_magnitude=True path).
 * mmw_range_doppler_mag64: double-precision |RD| of ONE virtual antenna, d_mag[F][S][C] float64.
 *   This is the CFAR input plane (range_doppler_detector.py:78 uses rx 0 only); computed
 *   end-to-end in float64 so detection indices are bit-exact against the float64 reference.
 * mmw_angle_fft: d_out[F][A][S][C] c64 = fftshift_A FFT_A( zero-pad_{V->A}( hann(V) rd ) )
 *   last stage of RangeAngleProcessorDBSEnhanced.compute_3d_windowed_fft
 *   (processors/range_angle_resp_dbs_enhanced.py:175-196).  With MMW_ANGLE_MAGNITUDE the output is
 *   float32 |.| [F][A][S][C] instead (perform/process_dbs_enhanced :293).
 * mmw_chain3d: mmw_range_doppler followed by mmw_angle_fft, chunked so the RD intermediate
 *   stays cache-resident; d_rd may be NULL (internal scratch) or [F][V][S][C] to keep it.
 *   Replaces compute_3d_windowed_fft (:137-198) end to end. */
int mmw_range_doppler(mmw_ctx *ctx, const void *d_cubes, void *d_out, void *d_mag_f32,
                      int n_frames, int V, int S, int C);
int mmw_range_doppler_mag64(mmw_ctx *ctx, const void *d_cubes, double *d_mag,
                            int n_frames, int V, int S, int C, int rx_idx);
/* flags of mmw_angle_fft / mmw_chain3d (0 = complex64 output, Hann(V) window, fftshift over angle) */
#define MMW_ANGLE_MAGNITUDE 1   /* float32 |.| output                                                    */
#define MMW_ANGLE_NO_WINDOW 2   /* no antenna window (DopplerAzimuthProcessor on "ods" geometry)         */
#define MMW_ANGLE_NO_SHIFT  4   /* no fftshift over the angle axis (shift_angle=False)                   */
int mmw_angle_fft(mmw_ctx *ctx, const void *d_rd, void *d_out, int n_frames,
                  int V, int S, int C, int A, int flags);
int mmw_chain3d(mmw_ctx *ctx, const void *d_cubes, void *d_rd, void *d_out, int n_frames,
                int V, int S, int C, int A, int flags);
/* Raw-cube variants: d_raw[F][num_rx][S][num_tx * loops] c64 as the DCA1000 delivers it; the virtual-array
 *   de-interleave of VirtualArrayReformatter.process (virtual antenna tx * num_rx + rx = every num_tx-th chirp from
 *   tx, processors/virtual_array_reformater.py:53-63) is folded into the first kernel's loads, so
 *   mmw_range_doppler_raw == mmw_virtual_array_reformat + mmw_range_doppler and
 *   mmw_chain3d_raw == mmw_virtual_array_reformat + mmw_chain3d without the extra pass over HBM
 *   (V = num_rx * num_tx, C = loops; outputs as for the non-raw calls). */
int mmw_range_doppler_raw(mmw_ctx *ctx, const void *d_raw, void *d_out, int n_frames, int num_rx, int num_tx,
                          int S, int loops);
int mmw_chain3d_raw(mmw_ctx *ctx, const void *d_raw, void *d_rd, void *d_out, int n_frames, int num_rx, int num_tx,
                    int S, int loops, int A, int flags);
/* int16 (I, Q) raw cubes [F][num_rx][S][num_tx * loops][2]: conversion and de-interleave inside the first kernel's loads
 * (256 x 128 planes and every plane shape of the shipped cfgs; other shapes convert first), no complex64 cube in between.
 * NO UPSTREAM ORACLE for the sample layout (the reference's reader, cpsl_datasets, is not in its tree): defined as
 * mmw_virtual_array_reformat_i16 + the virtual-array entry point, and bit-identical to that. */
int mmw_range_doppler_raw_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_out, int n_frames, int num_rx, int num_tx,
                              int S, int loops);
int mmw_chain3d_raw_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_rd, void *d_out, int n_frames, int num_rx,
                        int num_tx, int S, int loops, int A, int flags);
/* mmw_dbs_gather: d_out[F][S][n_out] float32 = d_mag[F][h_ang_idx[i]][s][h_vel_idx[i]] -- the Doppler-beam-
 *   sharpening column pick of perform_dbs_sharpen (processors/range_angle_resp_dbs_enhanced.py:216-263); the
 *   nearest-bin index tables are computed by the host from its angle / velocity bin tables. */
int mmw_dbs_gather(mmw_ctx *ctx, const float *d_mag, const int *h_ang_idx, const int *h_vel_idx, float *d_out,
                   int n_frames, int A, int S, int C, int n_out);
/* mmw_mean_over_range: d_out[F][C][A] float32 = mean over range rows [s_lo, s_hi) of d_mag[F][A][S][C];
 *   with mmw_chain3d(flags | MAGNITUDE) this is DopplerAzimuthProcessor.process, coarse path
 *   (processors/doppler_azimuth_resp.py:84-128,296-334,419-491): range FFT -> range-window mask ->
 *   2-D FFT over (chirp, antenna) -> |.| -> mean over the kept range bins. */
int mmw_mean_over_range(mmw_ctx *ctx, const float *d_mag, float *d_out, int n_frames,
                        int A, int S, int C, int s_lo, int s_hi);
/* mmw_doppler_azimuth: d_out[F][C][A] float32 = mean over range rows [s_lo, s_hi) of
 *   | fftshift_A FFT_A( pad_{V->A}( hann(V) RD[v][s][c] ) ) |, RD = the windowed range-Doppler spectrum of mmw_range_doppler:
 *   DopplerAzimuthProcessor.process, coarse path (processors/doppler_azimuth_resp.py:84-128,296-334,419-491) in one
 *   call; the [A][S][C] magnitude cube is never written (the angle FFT, |.| and the range mean are one kernel).
 *   flags: MMW_ANGLE_NO_WINDOW, MMW_ANGLE_NO_SHIFT.  Same result as mmw_chain3d(MAGNITUDE) + mmw_mean_over_range. */
int mmw_doppler_azimuth(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int A,
                        int s_lo, int s_hi, int flags);
/* mmw_doppler_azimuth_zoom: d_out[F][M][A] float32 = mean over range rows [s_lo, s_hi) of
 *   | fftshift_A FFT_A( pad_{V->A}( hann(V)[v] sum_{c < n_used} R[v][s][c] exp(-j 2 pi c h_freq[k]) ) ) |,
 *   R = FFT_S( hann(S) hann(C) x ), h_freq[M] in cycles per chirp (host array; NaN = a bin the reference fills
 *   with zeros).  This is DopplerAzimuthProcessor.process(use_precise_fft=True)
 *   (processors/doppler_azimuth_resp.py:130-163,208-294,477-489), whose two scipy.signal.ZoomFFT calls over the
 *   chirp axis evaluate exactly these sums; the caller derives h_freq from the velocity range (:148-155,254-283).
 *   flags: MMW_ANGLE_NO_WINDOW, MMW_ANGLE_NO_SHIFT. */
int mmw_doppler_azimuth_zoom(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int A,
                             int s_lo, int s_hi, int n_used, const double *h_freq, int M, int flags);

/* mmw_range_profile: d_out[F][S] float32 = mean_rx | FFT_S( hann(S) x[:, :, chirp] ) |
 *   replaces RangeProcessor.coarse_fft (processors/range_resp.py:32-57).
 * mmw_range_angle: d_out[F][S][A] float32 = | fftshift_A fft2( pad_A( (win) x[rx, :, chirp].T ) ) |
 *   replaces RangeAngleProcessor.process (processors/range_angle_resp.py:55-122); the window is
 *   applied over ALL V antennas before the subset h_rx[n_rx] is taken (:96-101); n_rx == 0 = all. */
int mmw_range_profile(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames,
                      int V, int S, int C, int chirp_idx);
/* float64 variant feeding the 1-D range CFAR of RangeDopplerDetectorSequential
 * (processors/range_doppler_detection/range_doppler_detector_sequential.py:84-91). */
int mmw_range_profile_f64(mmw_ctx *ctx, const void *d_cubes, double *d_out, int n_frames,
                          int V, int S, int C, int chirp_idx);
/* mmw_range_zoom: d_out[F][m] float32 = mean_rx | sum_n hann(S)[n] x[rx][n][chirp] exp(-j 2 pi n (f0 + k df)) |,
 *   k = 0..m-1, frequencies in cycles per sample: the zoomed range profile of RangeProcessor.zoom_fft
 *   (processors/range_resp.py:59-102), which the reference evaluates with scipy.signal.ZoomFFT. */
int mmw_range_zoom(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C,
                   int chirp_idx, int m, double f0_cycles_per_sample, double df_cycles_per_sample);
int mmw_range_angle(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames,
                    int V, int S, int C, int A, int chirp_idx, const int *h_rx, int n_rx,
                    int perform_windowing);

/* ---------------------------------------------------------------- CFAR detectors (float64, like the reference)
 * mmw_cfar2d: thresholds/noise [F][R][D] float64 and det mask [F][R][D] uint8 for
 *   BaseCFAR2D.detect + Ca/OsCFAR2D._compute_thresholds (detectors/base.py:208-230,
 *   ca_cfar.py:85-155, os_cfar.py:134-195).  CA: scale = alpha = N(pfa^(-1/N)-1) computed by the
 *   caller (base.py:281-293), k_rank ignored.  OS: scale = alpha, k_rank 1-based (os_cfar.py:131-132).
 *   Valid region only; elsewhere threshold = +inf, noise = 0; decision X > T strict.
 *   d_thr / d_noise may be NULL when only the mask is wanted.
 * mmw_cfar1d: same for Ca/Os/Go/SoCFAR1D (ca_cfar.py:11-77, os_cfar.py:29-86, go_so_cfar.py:11-123)
 *   over n_rows independent rows of length L.
 * mmw_compact2d: ordered (row-major, == np.where, base.py:229-230) compaction of the mask into
 *   d_dets[F][cap][2] int32 (row, col) and d_counts[F] int32.  Counts are exact even when
 *   a frame overflows cap (status MMW_ERR_TRUNCATED is reported by mmw_detect_* wrappers). */
int mmw_cfar2d(mmw_ctx *ctx, const double *d_X, double *d_thr, double *d_noise, uint8_t *d_mask,
               int n_frames, int R, int D, int kind, int train_r, int train_d,
               int guard_r, int guard_d, double scale, int k_rank);
int mmw_cfar1d(mmw_ctx *ctx, const double *d_x, double *d_thr, double *d_noise, uint8_t *d_mask,
               int n_rows, int L, int kind, int num_train, int num_guard, double scale, int k_rank);
int mmw_compact2d(mmw_ctx *ctx, const uint8_t *d_mask, int32_t *d_dets, int32_t *d_counts,
                  int n_frames, int R, int D, int cap);

/* mmw_detect_batch: the detection pipeline of RangeDopplerDetector2D for a batch of frames in one call:
 *   d_rd[F][V][S][C] c64 (mmw_range_doppler) and, for antenna 0, d_mag64[F][S][C] -> 2-D CFAR mask -> ordered
 *   detections d_dets[F][cap][2] / d_counts[F]
 *   (range_doppler_detection/range_doppler_detector.py:45-80 + range_doppler_detector_2d.py:49-65 per frame).
 *   d_l1 (may be NULL): [F][V] float32, the per-plane norms of mmw_plane_l1 for a following
 *   mmw_angle_argmax_exact -- produced inside the range-Doppler kernel where it can, so the cube is not read again. */
int mmw_detect_batch(mmw_ctx *ctx, const void *d_cubes, void *d_rd, double *d_mag64, uint8_t *d_mask,
                     int32_t *d_dets, int32_t *d_counts, float *d_l1, int n_frames, int V, int S, int C, int cfar_kind,
                     int train_r, int train_d, int guard_r, int guard_d, double scale, int k_rank, int cap);

/* mmw_detect_points: BASELINE configs[2] in one call -- range-Doppler of every antenna (float32, d_rd / d_l1 as above),
 *   CFAR on antenna 0, ordered detections and the azimuth / elevation argmax bins of every detection
 *   (range_doppler_detector.py:62-80, detectors/ca_cfar.py:85-155, point_cloud_generator.py:143-214), with the SAME
 *   results as the float64 path of mmw_detect_batch + mmw_angle_argmax_exact: the CFAR decision is screened on the
 *   float32 plane with a worst-case rounding-error band (mmw_detect.h); the few cells inside the band are decided from
 *   float64 DFT sums of the input cube.  One kernel launch per frame batch behind the range-Doppler kernel instead of six.
 *   d_mag32 (may be NULL): [F][S][C] float32 |RD| of antenna 0.  h_az / h_el: antenna lists (n == 0: that estimate
 *   is skipped and its index buffer may be NULL); d_az_idx / d_el_idx [F][cap] int32.
 *   d_counts[f] is the exact detection count (may exceed cap), or -1 for a frame the screening pass cannot decide
 *   (non-finite samples in antenna 0, or more undecided cells than its list holds): run mmw_detect_batch +
 *   mmw_angle_argmax_exact on such a frame.
 *   h_stats (may be NULL; passing it synchronises): [0] frames with undecided cells, [1] undecided cells, [2] frames
 *   returned with count -1, [3] / [4] azimuth / elevation detections re-evaluated in float64.
 *   MMW_ERR_UNSUPPORTED (nothing launched) when mmw_detect_points_supported(...) == 0: CA-CFAR only, S * C float32
 *   magnitudes must fit the LDS, at most 8 antennas per list -- at most 16 with A = 64 angle bins (the reference's
 *   az_el_fft_size): then the angle estimates are launches of their own in the call's tail, one lane per detection, over
 *   the cells the screening kernel copied aside (the "late argmax", the default whenever A = 64).
 *   The call's tail (exact cells, list insertion, float64 refinement) stays on side queues when the call returns: the
 *   range-Doppler launch of a following mmw_detect_points runs beside it; every other entry point of the context
 *   (mmw_sync, the copies, any other kernel) joins it first, so results are complete whenever they can be observed
 *   through this API.  Context option MMW_DETECT_DEFER_TAIL=0: each call joins its own tail. */
int mmw_detect_points_supported(int S, int C, int cfar_kind, int train_r, int train_d, int guard_r, int guard_d,
                                int n_az, int n_el, int A);
int mmw_detect_points(mmw_ctx *ctx, const void *d_cubes, void *d_rd, float *d_l1, float *d_mag32, int32_t *d_dets,
                      int32_t *d_counts, int32_t *d_az_idx, int32_t *d_el_idx, int n_frames, int V, int S, int C,
                      int cfar_kind, int train_r, int train_d, int guard_r, int guard_d, double scale, int k_rank, int cap,
                      const int *h_az, int n_az, int shift_az, const int *h_el, int n_el, int shift_el, int A,
                      int *h_stats);

/* ---------------------------------------------------------------- point cloud
 * mmw_angle_argmax: for each detection (r, v) of frame f gather rd[f][ant[i]][r][v], zero-pad to A,
 *   FFT, optional fftshift, |.|, first-max argmax (a NaN magnitude wins, as in np.argmax) -> d_idx[F][cap] int32.
 *   replaces PointCloudGenerator._compute_angle_estimation (processors/point_cloud_generator.py:143-214).
 *   float32 arithmetic on the float32 cube: where the reference's two best float64 magnitudes are within ~1e-6 of
 *   each other the index may differ -- the exact variant below removes that.
 * mmw_plane_l1: d_l1[F][V] float32 = sum over each plane of hann(S) hann(C) (|re| + |im|): the scale of the rounding-
 *   error bound the exact variant uses (computed once per batch, shared by the azimuth and elevation calls).
 * mmw_angle_argmax_exact: same result contract as the reference's complex128 computation (:186-206).  The float32
 *   pass bounds how far its magnitudes can be from the float64 ones (from d_l1 and the gathered cells: the WORST-CASE
 *   rounding bound of the kernel that produced the cube -- a proof; the context option MMW_ARGMAX_BOUND_DIV=8 restores
 *   round 3's empirical eighth of it); a detection whose winner is not certainly the float64 one --
 *   best and second-best closer than twice
 *   that bound and, for lists of up to 8 antennas, some bin also failing the pairwise form of the test that treats the two
 *   bins' errors as the same cell errors seen through two steering vectors -- is re-evaluated in float64 from the raw cube
 *   d_cubes (its range-Doppler cells as direct float64 2-D DFT sums, then the float64 angle DFT + argmax).
 *   (mmw_detect_points uses the worst-case bound throughout.)
 *   h_n_refined (may be NULL): number of re-evaluated detections; passing it makes the call synchronise.
 * mmw_angle_argmax_cells64: the float64 angle DFT + first-max argmax for rows of n_ant complex128 cells the caller
 *   gathered itself (a caller-supplied complex128 range-Doppler cube, :168-178); d_cells [n_rows][n_ant]. */
int mmw_angle_argmax(mmw_ctx *ctx, const void *d_rd, const int32_t *d_dets, const int32_t *d_counts,
                     int32_t *d_idx, int n_frames, int V, int S, int C, int cap,
                     const int *h_ant, int n_ant, int A, int shift);
int mmw_plane_l1(mmw_ctx *ctx, const void *d_cubes, float *d_l1, int n_frames, int V, int S, int C);
int mmw_angle_argmax_exact(mmw_ctx *ctx, const void *d_cubes, const float *d_l1, const void *d_rd,
                           const int32_t *d_dets, const int32_t *d_counts, int32_t *d_idx, int n_frames,
                           int V, int S, int C, int cap, const int *h_ant, int n_ant, int A, int shift,
                           int *h_n_refined);
int mmw_angle_argmax_cells64(mmw_ctx *ctx, const void *d_cells, int32_t *d_idx, int n_rows, int n_ant, int A,
                             int shift);

/* ---------------------------------------------------------------- beamformers
 * mmw_bartlett: delay-and-sum steering-matrix contraction on MFMA for a batch of frames,
 *   Y[f][s][t] = FFT_S( hann(S) * sum_e X[f][s][e] hamming(E)[e] exp(j 2 pi d_t . p_{f,e} / lambda) )
 *   d_X [F][S][E] c64, d_P [F][3][E] float64 (each frame's array geometry), d_dirs [3][T] float64,
 *   d_out [F][S][T] c64.
 *   replaces compute_synthetic_response / compute_response_at_steering_angle
 *   (processors/simple_synthetic_array_beamformer_processor_multiFrame.py:499-585), whose Python loop over steering
 *   angles (:543-585) becomes one complex GEMM per frame.
 * mmw_capon: MVDR spectrum on a V-element half-wavelength ULA for a batch of frames; NO upstream implementation
 *   exists -- definition in DESIGN.md; d_X [F][V][R][K] c64 (K snapshots per range bin),
 *   d_out [F][R][T] float32. */
int mmw_bartlett(mmw_ctx *ctx, const void *d_X, const double *d_P, const double *d_dirs,
                 void *d_out, int n_frames, int S, int E, int T, double lambda_m);
int mmw_capon(mmw_ctx *ctx, const void *d_X, const double *h_thetas, float *d_out,
              int n_frames, int V, int R, int K, int T, double delta);

/* ---------------------------------------------------------------- element-wise helpers */
int mmw_abs_c64(mmw_ctx *ctx, const void *d_in, float *d_out, size_t n);
/* d_out[i] = (double) d_in[i], n floats (a complex64 array of m elements: n = 2 m): lets a caller that owes the reference's
 *   complex128 / float64 dtypes (range_angle_resp_dbs_enhanced.py:196, range_doppler_resp.py:103) widen on the device and
 *   download the result directly instead of converting on a host core. */
int mmw_widen_f32_f64(mmw_ctx *ctx, const float *d_in, double *d_out, size_t n);

/* ---------------------------------------------------------------- diagnostics
 * In-situ HBM ceiling on the same device: mode 0 = 16-B/lane copy, 1 = write only, 2 = read only,
 * grid-stride over `blocks` workgroups of 256 (0 = 8 per CU).  Used by tools/kbench.py to state what
 * a known-good streaming kernel reaches next to the hot-path kernels. */
/* mmw_diag_set_option: a tuning / test switch of THIS context (the names INTEGRATION.md lists; the same names are read
 *   from the environment when the context has no value of its own).  value == INT_MIN removes the context's value. */
int mmw_diag_set_option(mmw_ctx *ctx, const char *name, int value);
int mmw_diag_membw(mmw_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int mode, int blocks);
/* Matrix-core peak of this device, measured: back-to-back MFMAs on independent accumulators with register operands,
 * two waves per SIMD.  kind 0 = v_mfma_f32_32x32x2_f32 (the Bartlett GEMM's instruction), 1 = v_mfma_f64_16x16x4_f64
 * (the Capon covariance).  The figure the beamformer kernels' TFLOP/s are divided by in tools/kbench.py.
 * kinds 2 / 3: the kind-0 stream with 8 / 16 independent v_fma_f32 after every MFMA (rate still counts the MFMAs only);
 * kinds 4 / 5: the same with ONE wave per SIMD -- how much vector work hides under float32 MFMAs (it does not);
 * kinds 6 / 7: the contrast case, v_mfma_f32_32x32x16_bf16 alone / with 8 v_fma_f32 after every MFMA (tools/pmc_coexec.sh
 * collects SQ_VALU_MFMA_COEXEC_CYCLES for all of them). */
int mmw_diag_mfma_peak(mmw_ctx *ctx, int kind, double *tflops);
/* Which range-Doppler kernel mmw_range_doppler picks for an S x C plane, without touching a device (host logic
 * only): plan[0] = 0 fused 256x128 | 1 LDS-resident power of two | 2 mixed radix | 3 generic two-kernel path |
 * 4 single pass with half / three quarters of the plane carried in registers (planes of 32768 / 65536 cells);
 * for the mixed-radix kernel plan[1..7] = register class, has a run-time-radix level, S1, S2, C1, C2 (S = S1 S2,
 * C = C1 C2), dynamic LDS bytes.  float64 != 0 asks for the float64 CFAR-plane variant. */
int mmw_diag_rd_plan(int S, int C, int float64, int plan[8]);
/* Schedule mmw_chain3d(d_rd = NULL) would use for this batch on this context (host logic + environment knobs only):
 * plan[0] = 1 overlapped (range-Doppler || angle on two queues) / 0 serial, plan[1] = frames per kernel launch,
 * plan[2] = ring depth of range-Doppler chunks in flight, plan[3] = CUs of the range-Doppler queue (0 = unmasked),
 * plan[4] = range-Doppler planes transformed per frame (V, or V - 2 when the zero-weight end antennas of the
 * Hann(V) window are skipped), plan[5] = chain calls of this context re-run on the event schedule after a
 * hand-off timeout so far, plan[6] = 1 for the device-synchronised form (ONE range-Doppler
 * launch and ONE angle launch per call, handing frames over through counters in device memory; plan[1] is then the
 * whole batch), plan[7] = frames in its ring of range-Doppler cubes.  bench.py derives its bytes-per-launch from
 * this instead of restating the rule. */
int mmw_diag_chain_plan(mmw_ctx *ctx, int n_frames, int V, int S, int C, int A, int flags, int plan[8]);

/* Host logic without a device (planning only; what tests/cpp/host_sanitize.cpp runs under ASan / UBSan):
 * mmw_diag_chain_plan_nodev = mmw_diag_chain_plan for a device of num_cu CUs (raw != 0: mmw_chain3d_raw);
 * mmw_diag_detect_plan: banding of mmw_detect_points -- plan[0] supported, [1] workgroups per frame (1), [2] plane rows one
 *   band's loads can carry (band + halo rows), [3] band rows, [4] band pitch, [5] LDS bytes, [6] compile-time window, [7] rounding-error budget of the range-Doppler
 *   kernel for this plane in units of 2^-24 of the plane's L1 norm;
 * mmw_diag_czt_runs: the uniform runs (offset, length, zero-filled) a zoom frequency list of M bins is cut into. */
int mmw_diag_chain_plan_nodev(int num_cu, int raw, int n_frames, int V, int S, int C, int A, int flags, int plan[8]);
int mmw_diag_detect_plan(int S, int C, int cfar_kind, int train_r, int train_d, int guard_r, int guard_d, int n_az,
                         int n_el, int A, int plan[8]);
int mmw_diag_czt_runs(const double *h_freq, int M, int n_used, int *h_runs, int cap, int *n_runs);

/* ---------------------------------------------------------------- per-kernel timing hook for bench.py
 * Average duration (ms) of the most recent launch group of the named kernel family measured
 * with HIP event pairs on the launching queue (no host sync): "rd", "angle", "cfar", ...
 * mmw_profile_enable(ctx, n): 0 = off, 1 = time every launch group, n > 1 = every n-th per family
 * (an event pair costs a few microseconds of queue time, so bench.py samples). */
int mmw_profile_enable(mmw_ctx *ctx, int on);
int mmw_profile_get(mmw_ctx *ctx, const char *family, float *total_ms, int *launches);
int mmw_profile_reset(mmw_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MMWGPU_H */

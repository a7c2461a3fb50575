/*
 * wsa.h — C ABI of libwsa: the MI355X (gfx950) implementation of the formantanalyzer hot path
 *         PCM -> Hann -> FFT -> mel -> u32 frame -> peak scan -> voiced state machine / noise gate
 *         -> formant tracking -> segment finalize -> (syllables) -> 53-feature vectors.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface
 * it stands in for; "ref" = /root/reference/dist/main.js (line 2, byte offset "@B<n>") which
 * bundles formantanalyzer@1.1.6, and /root/reference/src/index.js (its caller).
 * INTEGRATION.md shows the N-API / ctypes bindings over this header.
 *
 * Conventions: plain C types only; every function returns a wsa_status (0 = OK); the library
 * never falls back to a CPU path — if no gfx950 device / code object is available, wsa_create
 * fails with WSA_ERR_NO_DEVICE.  A context is not thread-safe; distinct contexts are.
 */
#ifndef WSA_H
#define WSA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WSA_ABI_VERSION 5       /* 5: wsa_queue_create / wsa_queue_destroy (a HIP stream for hosts without a HIP binding: batches of one device overlap instead of queueing on the null stream); 4: wsa_host_alloc / wsa_host_free (page-locked clip memory), wsa_gather_rows runs every rank on its own device and stream; 3: wsa_gather_* (multi-GPU collection over RCCL); 2: wsa_stream_rows gained stream_cuts, formants, utt_*, track_*, WSA_FLAG_STREAM_CUT, d_spectra always set, default output_level 4 */
#define WSA_NFEAT 53            /* ref src/localstore.js:7 process_exp_features_len[5] == [13] == 53 */
#define WSA_NUTT 264            /* utterance features of output_level 11 (ref @B107902: 15 histograms) */

typedef enum {
    WSA_OK = 0,
    WSA_ERR_INVALID = 1,        /* bad argument / unsupported configuration */
    WSA_ERR_NO_DEVICE = 2,      /* no gfx950 GPU or HIP runtime failure at create */
    WSA_ERR_HIP = 3,            /* HIP runtime error (see wsa_last_error) */
    WSA_ERR_CAPACITY = 4,       /* a device-side arena overflowed (reported, never silently wrong) */
    WSA_ERR_BUSY = 5            /* "Error: Already playing" (ref @B4554) */
} wsa_status;

/*
 * Configuration = the reference's settings object: defaults ref @B2965, merged by
 * configure(cfg) ref @B3292, forwarded to the worklet ref @B6726 and to reset_segmentation
 * ref @B24629.  Units as in the reference (Hz, ms, dB).  wsa_config_default() fills the
 * reference defaults.
 */
typedef struct {
    int32_t spec_type;          /* 1 mel bands (hot path), 2 power bins, 3 magnitude bins */
    int32_t output_level;       /* 5 = segment features, 13 = syllable features; 4 / 10 = segment / syllable formant
                                   frames (rows carry the indices, d_formants the [len][9] frames); 11 = level 10 plus the
                                   264 utterance features after every result (d_utt_*); 12 = one row per syllable whose
                                   first 23 feature slots hold the polynomial coefficients (slot 23 = 1 marks a syllable on
                                   which numeric.uncmin threw in the reference: its make_coeffs then returns only the rows
                                   before it, so a host reports the segment's features up to that row, the times in full); 3: indices only;
                                   1,2: u32 spectrum frames only — the back end is not run, any band count) */
    double  f_min, f_max;       /* Hz */
    int32_t N_fft_bins, N_mel_bins;
    double  window_width, window_step;     /* ms */
    double  pause_length, min_seg_length;  /* ms */
    int32_t auto_noise_gate;
    double  voiced_max_dB, voiced_min_dB;
    double  pre_norm_gain, high_f_emph;
} wsa_config;

typedef struct wsa_ctx wsa_ctx;         /* one per (config, device) — the module-level state of ref inner module 1 */
typedef struct wsa_batch wsa_batch;     /* a planned batch shape: n_clips x n_samples[] at one sample rate */

void wsa_config_default(wsa_config *cfg);                                 /* ref @B2965 */
int  wsa_abi_version(void);

/* ref: module load + configure() (@B3292).  device = HIP device ordinal. */
wsa_status wsa_create(const wsa_config *cfg, int32_t device, wsa_ctx **out);
void       wsa_destroy(wsa_ctx *ctx);
const char *wsa_last_error(const wsa_ctx *ctx);   /* ctx may be NULL: last create error */

/* Geometry the front end derives from (config, fs): what ref reset_nodes (@B6992) hands the worklet. */
typedef struct {
    int32_t nfft, win, hop, bands, kmax;
} wsa_geometry;
wsa_status wsa_geometry_for(const wsa_ctx *ctx, double fs, wsa_geometry *out);
/* centre frequency of each output band — the `{bins_Hz: [...]}` message of the worklet (ref @B8380) */
wsa_status wsa_bins_hz(const wsa_ctx *ctx, double fs, double *out, int32_t n);

/*
 * Plan a batch: n_clips independent clips ("launches" in the reference's terms: each clip is one
 * LaunchAudioNodes(1, buffer, cb, labels, offline=true, test_play=false) run, ref @B4469), clip i
 * having n_samples[i] mono float32 samples at `fs`.  Allocates all device work space; nothing is
 * allocated in wsa_batch_run.
 */
wsa_status wsa_batch_create(wsa_ctx *ctx, uint32_t n_clips, const uint32_t *n_samples, double fs, wsa_batch **out);
/*
 * The same with a sample-rate conversion in front: the clips the caller hands to wsa_batch_run / wsa_batch_run_host
 * are at fs_in (n_samples_in[i] samples), the analysis runs on their conversion to fs_out (wsa_resample_length
 * samples each) — what the browser's decodeAudioData does to a file before the reference sees it (its offline path
 * decodes into `new OfflineAudioContext(1, 48e6, 48e3)`, ref dist/main.js:2 @B18769: always 48 kHz).  The converter is
 * specified here, not by the reference (RS-1, DESIGN.md: 32-tap windowed sinc, 32 sub-sample offsets, parity unpinned).
 * The converted PCM lives in the batch; wsa_batch_copy_pcm hands it out.
 */
wsa_status wsa_batch_create_resampled(wsa_ctx *ctx, uint32_t n_clips, const uint32_t *n_samples_in, double fs_in, double fs_out, wsa_batch **out);
uint64_t   wsa_resample_length(uint64_t n_in, double fs_in, double fs_out);      /* trunc(n_in * fs_out / fs_in) */
/* converted PCM of a resampling batch after a run: pcm [n_clips][stride] floats, stride >= the longest converted clip */
wsa_status wsa_batch_copy_pcm(wsa_batch *b, void *stream, float *pcm, uint64_t stride);
void       wsa_batch_destroy(wsa_batch *b);

/*
 * Run the whole hot path on device-resident PCM.  d_pcm is a device pointer; clip i starts at
 * d_pcm + i * clip_stride (floats).  `stream` is a hipStream_t (NULL = default stream).  Fully
 * asynchronous: only kernel launches / async copies are enqueued, so the call can be captured in
 * a hipGraph and timed with events on `stream`.
 * Stands in for: worklet process() (not in ref tree) -> port.onmessage -> spectrum_push (@B8752,
 * @B30392) -> D() (@B25717) -> O() (@B27088) for every frame of every clip.
 */
wsa_status wsa_batch_run(wsa_batch *b, const float *d_pcm, uint64_t clip_stride, void *stream);

/* Same, PCM in host memory (clip i at pcm[i]); copies H2D on `stream` first (PCIe-inclusive path). */
wsa_status wsa_batch_run_host(wsa_batch *b, const float *const *pcm, void *stream);
/* Same with 16-bit PCM as a WAV file holds it (ref: the file's ArrayBuffer goes in as it is, src/index.js:291): clip i = its samples
 * interleaved over `channels[i]` channels (NULL: all mono), channel 0 is analysed; n_samples[i] of the plan counts frames of one
 * channel.  Half the PCIe bytes of the float path; the conversion x / 32768 runs on the device and is exact in fp32, so the rows equal
 * those of wsa_batch_run_host on the converted floats bit for bit.  Uploads are issued clip by clip on `stream` in front of the kernels. */
wsa_status wsa_batch_run_host_i16(wsa_batch *b, const int16_t *const *pcm, const uint32_t *channels, void *stream);
/* Page-locked host memory for the clips of the two entry points above (no counterpart in the reference: the app hands the browser the file's
 * ArrayBuffer, src/index.js:291).  A copy out of ordinary (pageable) memory is staged by the runtime through its own pinned buffers on the calling
 * thread; a clip that already lies in memory from wsa_host_alloc goes to the device by DMA at the link's rate.  Hosts that read many files
 * allocate their clip buffers here (the Node host: allocPinned).  Any clip pointer is accepted by the run functions either way.
 * wsa_host_free(NULL) is a no-op.  The copy out of page-locked memory is a true asynchronous DMA: until the run's stream has completed, clip buffers
 * must stay allocated AND unmodified (a host that refills a slab for the next batch waits for wsa_batch_result of the one in flight first, or
 * alternates between two slabs).  Clips that lie back to back inside ONE wsa_host_alloc allocation (and keep the device layout's alignment) travel as
 * one copy; copies are never merged across two allocations, however close they lie. */
wsa_status wsa_host_alloc(wsa_ctx *ctx, uint64_t bytes, void **out);
void       wsa_host_free(void *p);
/* A stream of the library's own on the context's device (a non-blocking hipStream_t), for hosts that have no HIP binding to make one (the Node addon):
 * what every `stream` argument of this header accepts.  Runs handed different queues overlap on the device — the upload of one batch under the kernels
 * of another (the reference's app loops over files one launch at a time, src/index.js:277-296; a host that does the same over batches keeps the PCIe
 * link busy that way) — where runs on NULL, the device's null stream, queue up behind each other.  Destroy a queue only when nothing runs on it. */
wsa_status wsa_queue_create(wsa_ctx *ctx, void **stream);
void       wsa_queue_destroy(wsa_ctx *ctx, void *stream);

/*
 * Results of the last run (device resident, compacted in (clip, si[, syllable]) order — the order
 * in which the reference would have invoked callback(si, label, seg_time, features), ref @B28869).
 *
 * rows:     one per callback feature vector: level 5 one per reported segment, level 13 one per
 *           syllable.  meta columns (int32):
 *             [0] clip  [1] si (callback index within the clip, ref `callbacks_processed-1`)
 *             [2] t_start frame  [3] t_len frames   -> seg_time as the reference computes it
 *                 level 5:  [t_start*step_s, (t_len+1)*step_s]                       (ref @B31504)
 *                 level 13: [((t_start)*step_s).toFixed(3), ((t_len+1)*step_s).toFixed(3)]
 *                           with t_start = segments_ci[si][0] + syl_start            (ref @B31114)
 *             [4] own segment index in segments_ci  [5] syllable index in its segment (level 13, else 0)
 *             [6] own start frame (segment, or segment start + syllable start)  [7] own length
 *           features: double[53] per row (ref @B33436 order).
 * segments: every [start,len] the reference pushes to segments_ci (ref @B27190), including the
 *           ones whose straighten step throws (flag -1): int32 [n_segments][4] =
 *           {clip, start, len, flag} with flag 1 = reported, 0 = no feature rows, -1 = dropped.
 */
typedef struct {
    uint32_t n_clips, n_rows, n_segments, n_frames_total;
    uint32_t status_flags;                /* WSA_FLAG_* */
    const int32_t  *d_row_meta;           /* device [n_rows][8] */
    const double   *d_row_feat;           /* device [n_rows][53] */
    const int32_t  *d_segments;           /* device [n_segments][4] */
    const uint32_t *d_clip_row_off;       /* device [n_clips+1] */
    const uint32_t *d_clip_seg_off;       /* device [n_clips+1] */
    const uint32_t *d_spectra;            /* device [n_frames_total][bands] u32 frames (the worklet's output) */
    const uint32_t *d_clip_frame_off;     /* device [n_clips+1] */
    const float    *d_formants;           /* levels 4 / 10 (else NULL): device [n_frames_total][9] f32 — the straightened
                                             frames (3 x bin, band energy, width; ref @B35074) of every reported segment,
                                             frame d of the segment of a row at d_clip_frame_off[clip] + meta[6] + d
                                             (level 4: the row's segment; level 10: the row's syllable, meta[7] frames) */
    /* level 11 (else 0 / NULL): one entry per callback `callback(0, label, Y(), get_utterance_features(...))` of the
     * reference's dispatcher (ref @B28869, @B107902), i.e. one per result, each over everything the clip produced so far */
    uint32_t        n_utterance_rows;
    const int32_t  *d_utt_meta;           /* device [n][4] = {clip, result index, t_start frame, t_sum frames}:
                                             seg_time = [t_start*step_s, (t_sum+1)*step_s]  (ref Y() @B31330) */
    const double   *d_utt_feat;           /* device [n][264] */
    const uint32_t *d_clip_utt_off;       /* device [n_clips+1] */
} wsa_device_result;

/* Synchronises `stream`, reads the counters back and fills `out` (pointers stay valid until the
 * next run on this batch or wsa_batch_destroy). */
wsa_status wsa_batch_result(wsa_batch *b, void *stream, wsa_device_result *out);

/* Copy results to caller-provided host buffers (any pointer may be NULL to skip it).
 * Capacities in elements of the respective row type; fails with WSA_ERR_INVALID if too small. */
wsa_status wsa_batch_copy_rows(wsa_batch *b, void *stream, int32_t *row_meta, double *row_feat, uint32_t rows_cap,
                               int32_t *segments, uint32_t seg_cap, uint32_t *clip_row_off, uint32_t *clip_seg_off);
wsa_status wsa_batch_copy_spectra(wsa_batch *b, void *stream, uint32_t *spectra, uint64_t cap_words, uint32_t *clip_frame_off);
/* The u32 frames (the worklet's messages, ref @B8568) are handed from the front end to the peak scan through an array in HBM, so
 * they are always available after a run (wsa_batch_copy_spectra, d_spectra); this call is kept for hosts written against ABI
 * version 1, where a fused front end could keep them on the chip, and has no effect. */
wsa_status wsa_batch_keep_spectra(wsa_batch *b, int32_t on);
/* Number of times results were fetched only after a second pass of the back end with the full-size tracker table (the
 * default tracker variant keeps its active-track table in LDS and reports an overflow; see wsa_batch_result).  A hipGraph
 * captured from wsa_batch_run keeps replaying the variant it was captured with: re-capture after this number changed. */
wsa_status wsa_batch_backend_reruns(const wsa_batch *b, uint32_t *out);
/* levels 4 / 10: the whole d_formants table ([n_frames_total][9] floats; rows of frames outside reported segments are unspecified) */
wsa_status wsa_batch_copy_formants(wsa_batch *b, void *stream, float *formants, uint64_t cap_frames);
/* level 3: the ranked raw formant tracks of every segment — what the reference's callback receives as its third
 * argument (`s.push(get_ranked_formants())`, ref dist/main.js:2 @B28273; dispatched by `b(e, label, s[e])` @B30132).
 * Segments in the order of d_segments.  seg_off [n_segments + 1][2] = offsets of a segment's points / ranked ids;
 * points [n_points][8] in arrival order = {track id, bin | width << 8, band energy (f64: lo, hi word), start bin,
 * amplitude (u32), filing index (the reference's frame list entry), end bin} — the per-point entries of the 18-field
 * track record (@B35952: [7] frames, [8] starts, [9] ends, [10] bins, [11] amps, [12] energies; every other field is a
 * function of them); ranked [n_ranked] = ids of the tracks with count >= 2 and mean bin >= 7, ascending by mean bin (@B35670). */
typedef struct { uint32_t n_segments; uint64_t n_points, n_ranked; } wsa_tracks_info;
wsa_status wsa_batch_tracks_info(wsa_batch *b, void *stream, wsa_tracks_info *out);
wsa_status wsa_batch_copy_tracks(wsa_batch *b, void *stream, uint64_t *seg_off, int32_t *points, uint64_t cap_points, int32_t *ranked, uint64_t cap_ranked);
/* level 11: the utterance-feature entries (any pointer may be NULL); cap_rows in entries */
wsa_status wsa_batch_copy_utterance(wsa_batch *b, void *stream, int32_t *utt_meta, double *utt_feat, uint32_t cap_rows, uint32_t *clip_utt_off);

/* Capacity bounds of a planned batch (so callers can size buffers before running). */
typedef struct {
    uint32_t n_clips, n_frames_total, max_frames_per_clip, bands;
    uint32_t rows_cap, segments_cap;      /* worst-case totals */
    uint64_t workspace_bytes;             /* device memory held by the batch */
} wsa_batch_info;
wsa_status wsa_batch_get_info(const wsa_batch *b, wsa_batch_info *out);

/* Per-stage device time of the last run in ms, measured with HIP events on the run's stream
 * (valid after wsa_batch_result): [0] front end (PCM->u32), [1] peak candidates + gate + span order,
 * [2] tracker (formant tracking + finalize), [3] compaction. */
wsa_status wsa_batch_stage_ms(wsa_batch *b, float out[4]);

/* Stage events are recorded on the run's stream by default; switch them off before capturing a run
 * into a hipGraph. */
wsa_status wsa_batch_enable_timing(wsa_batch *b, int32_t on);

/* Per-frame state trace of the sequential stage (tests / debugging): 12 doubles per frame =
 * c_ci, c_started, no_fm_segs, ctx_max, noise_floor, n_peaks, argmax bin, h, d, g (the values the
 * reference's frame loop holds just before `c_ci++`, ref @B26985) and the tracker's non-formant /
 * formant energy accumulators (ref @B35952 `s`, `c`). */
wsa_status wsa_batch_enable_trace(wsa_batch *b, int32_t on);
wsa_status wsa_batch_copy_trace(wsa_batch *b, void *stream, double *out, uint64_t cap_rows);

/* Run only the front end (PCM -> u32 frames), for front-end parity tests and profiling. */
wsa_status wsa_batch_run_frontend(wsa_batch *b, const float *d_pcm, uint64_t clip_stride, void *stream);
/* Run only the back end on caller-supplied u32 frames laid out like d_spectra (device pointer). */
wsa_status wsa_batch_run_backend(wsa_batch *b, const uint32_t *d_spectra, void *stream);

/*
 * ---- Collection of the feature rows of several GPUs (BASELINE config 4: clips sharded per GPU, "a single RCCL gather over xGMI").
 * No counterpart in the reference: it runs one launch per file on one device (ref src/index.js:291) and all state is per launch
 * (reset_segmentation, ref dist/main.js:2 @B24629), so clips shard over GPUs with nothing exchanged on the data path; collecting the row
 * tables is the only communication.  One PROCESS drives the GPUs here (the Node host: configure({devices: [...]})): one context per
 * GPU = one rank, one planned batch per context holding that rank's clips.  After wsa_batch_run on every batch, wsa_gather_rows reads the
 * ranks' row counts (every run publishes them) and moves exactly each rank's rows — [rows][8] int32 metadata and [rows][53] double
 * features, in their own types — into the root context's device memory with ONE grouped RCCL exchange (ncclSend / ncclRecv over the
 * direct xGMI links; the root's own rows as a send to itself in the same group): rows of rank 0, rank 1, ... back to back, each rank's
 * rows in its own (clip, callback) order with meta[0] = the clip's index inside its rank.  Then one copy from the root device
 * (wsa_gather_copy_rows) instead of one per GPU.  librccl is loaded on first use; WSA_ERR_NO_DEVICE if it is not there.
 * With one process per GPU (bench.py under torch.distributed) the same protocol runs as webspeechanalyzer_amd/gather.py.
 */
typedef struct wsa_gather wsa_gather;
typedef struct {
    uint32_t n_ranks, n_rows;             /* n_rows = sum of rows_per_rank */
    const uint32_t *rows_per_rank;        /* host [n_ranks], valid until the next gather */
    const int32_t  *d_row_meta;           /* root device [n_rows][8] */
    const double   *d_row_feat;           /* root device [n_rows][53] */
} wsa_gather_result;
/* ctxs[i] = rank i (distinct devices); root = index of the context whose device collects */
wsa_status wsa_gather_create(wsa_ctx *const *ctxs, int32_t n_ranks, int32_t root, wsa_gather **out);
void       wsa_gather_destroy(wsa_gather *g);
/* batches[i] = rank i's batch, planned on ctxs[i] (checked) and run on streams[i].  Waits for every rank's run, then enqueues the
 * exchange, every rank's calls with that rank's device current: on streams[i], or — where streams or streams[i] is NULL — on a stream
 * of the gather's own on rank i's device (never "the null stream of whatever device is current").  The tables are complete when the
 * root's stream is (wsa_gather_copy_rows waits for it).  d_row_meta / d_row_feat / rows_per_rank belong to the wsa_gather and are
 * valid until the NEXT wsa_gather_rows on it (which may free and re-allocate the tables) or wsa_gather_destroy.  If a call inside the
 * RCCL group fails, the group is still closed, the first error is returned (wsa_last_error of the root context) and the wsa_gather
 * refuses further use: destroy it and create a new one.  A wsa_gather is not thread-safe (one gather at a time per object). */
wsa_status wsa_gather_rows(wsa_gather *g, wsa_batch *const *batches, void *const *streams, wsa_gather_result *out);
/* Waits for the last gather and copies its tables to the host (either pointer may be NULL); rows_cap in rows. */
wsa_status wsa_gather_copy_rows(wsa_gather *g, int32_t *row_meta, double *row_feat, uint32_t rows_cap);

/*
 * ---- Streams: n_streams concurrent launches advancing in lock step (BASELINE config "streaming").
 * Stands in for the reference's online path: the worklet's process() per frame -> port message ->
 * spectrum_push (ref @B8752, @B30392) with the module-level segmenter / tracker state carried from
 * frame to frame, the callback fired when a segment closes (ref @B28869), and StopAudioNodes ->
 * segment_truncate (ref @B5699, @B30757) at the end of a stream.  Every step hands each stream
 * frames_per_step * hop new samples; results are identical to running the whole signal as one clip.
 * A step only enqueues work (H2D of the control words, kernels, D2H of the step's rows) and is captured
 * into a hipGraph after the first call (wsa_stream_enable_graph), so a step is one graph launch.
 */
typedef struct wsa_stream wsa_stream;

/* status_flags of wsa_device_result / wsa_stream_rows */
#define WSA_FLAG_CAPACITY   1u   /* a device-side arena overflowed: results invalid (the call also returns WSA_ERR_CAPACITY) */
#define WSA_FLAG_STREAM_CUT 8u   /* streams: some stream's voiced span reached max_span_frames in this step and was cut (stream_cuts tells which);
                                    that stream's rows deviate from one uninterrupted run until its next pause — not an error */

#define WSA_STREAM_ACTIVE 1u    /* the stream has samples in this step */
#define WSA_STREAM_START  2u    /* fresh launch state before this step's frames (ref reset_segmentation @B24629) */
#define WSA_STREAM_STOP   4u    /* segment_truncate after this step's frames (ref @B30757): flushes the open segment */

/* max_span_frames: longest voiced span (frames between two segmenter resets) a stream may hold, rounded
 * up to a power of two.  A source that does not pause for that long is cut there as if it had been stopped and started again
 * (segment_truncate, ref @B30757): the segment so far is reported, tracking starts afresh — for THAT stream only, and
 * counted in wsa_stream_rows.stream_cuts; its results deviate from one uninterrupted run until its next pause. */
wsa_status wsa_stream_create(wsa_ctx *ctx, uint32_t n_streams, double fs, uint32_t frames_per_step,
                             uint32_t max_span_frames, wsa_stream **out);
void       wsa_stream_destroy(wsa_stream *st);
uint32_t   wsa_stream_samples_per_step(const wsa_stream *st);        /* frames_per_step * hop */

/* One step on device-resident PCM: stream i's new samples at d_pcm + i * stream_stride.  ctl = host array
 * of n_streams control bytes (WSA_STREAM_*), or NULL: START|ACTIVE on the first step, ACTIVE afterwards. */
wsa_status wsa_stream_step(wsa_stream *st, const float *d_pcm, uint64_t stream_stride, const uint8_t *ctl, void *stream);
/* Same with the samples in the stream object's own pinned host buffer ([n_streams][samples_per_step]
 * floats): fill it, call this; the H2D copy is part of the step (and of its graph).  The device reads the buffer while the
 * step runs: refill it only after wsa_stream_collect has returned for that step. */
float     *wsa_stream_host_input(wsa_stream *st);
wsa_status wsa_stream_step_host(wsa_stream *st, const uint8_t *ctl, void *stream);

/* Rows of the last step, in (stream, callback) order; meta / feature layout as wsa_device_result with
 * [0] = stream index and [1] = callback index since the stream's START.  Host memory owned by the stream
 * object, valid until the next step. */
typedef struct {
    uint32_t n_rows, n_segments, status_flags;
    const int32_t *row_meta;       /* [n_rows][8] */
    const double  *row_feat;       /* [n_rows][53] */
    const int32_t *segments;       /* [n_segments][4] = {stream, start, len, flag} */
    const uint32_t *stream_cuts;   /* [n_streams] spans cut at max_span_frames since the stream's START */
    /* levels 4 / 10 (else NULL): the straightened frames (3 x bin, band energy, width; ref @B35074) of row r's segment (level 4) or
     * syllable (level 10): meta[7] frames of 9 floats starting at formants + 9 * row_formant_off[r] ([n_rows + 1] offsets) */
    const float    *formants;
    const uint32_t *row_formant_off;
    /* level 11 (else 0 / NULL): one entry per result of this step, as wsa_device_result's utterance tables — utt_meta [n][4] =
     * {stream, result index since START, first segment's start, sum of the segment lengths so far}, utt_feat [n][264]: the histograms over
     * everything the stream has produced since its START (ref @B107902), carried on the device from step to step */
    uint32_t        n_utterance_rows;
    const int32_t  *utt_meta;
    const double   *utt_feat;
    /* level 3 (else 0 / NULL): the ranked raw tracks of this step's segments, laid out as wsa_batch_copy_tracks lays them out — track_off
     * [n_segments + 1][2] = first point / first ranked id of segment k, track_points [n_track_points][8], track_ranked [n_track_ranked] */
    uint64_t        n_track_points, n_track_ranked;
    const uint64_t *track_off;
    const int32_t  *track_points;
    const int32_t  *track_ranked;
} wsa_stream_rows;
wsa_status wsa_stream_collect(wsa_stream *st, void *stream, wsa_stream_rows *out);   /* synchronises `stream` */
wsa_status wsa_stream_enable_graph(wsa_stream *st, int32_t on);
/* n_steps timed steps (measurement helper): step k copies feed[k mod feed_steps] ([n_streams][samples_per_step] floats; NULL: the
 * input buffer stays as it is) into the pinned input buffer, then wsa_stream_step_host + wsa_stream_collect are timed with the
 * host's monotonic clock; out_us[k] = microseconds of step k, *rows_total = feature rows produced (may be NULL). */
wsa_status wsa_stream_time_steps(wsa_stream *st, uint32_t n_steps, const float *feed, uint32_t feed_steps, void *stream, double *out_us, uint64_t *rows_total);

#ifdef __cplusplus
}
#endif
#endif

/*
 * wsa_napi.c — thin N-API binding of the C ABI (include/wsa.h) for the JavaScript host
 * (webspeechanalyzer_amd/js/formantanalyzer.js).  Raw node_api.h, N-API >= 4 (async work).
 *
 * Exposes exactly what the host needs:
 *   abiVersion() -> number
 *   defaults() -> config object                                  (wsa_config_default, ref @B2965)
 *   create(config, device) -> external ctx                        (wsa_create)
 *   destroy(ctx)
 *   geometry(ctx, fs) -> {nfft, win, hop, bands, kmax}            (wsa_geometry_for)
 *   binsHz(ctx, fs) -> Float64Array                               (wsa_bins_hz, ref @B8380)
 *   processBatch(ctx, clips: Float32Array[], fs[, output_level[, analysis_rate]]) -> Promise<{meta, feat, segments, rowOff, segOff, stageMs[, formants, frameOff][, trackOff, trackPoints, trackRanked]}>
 *       runs wsa_batch_create / wsa_batch_run_host / wsa_batch_copy_rows on a worker thread
 *       (napi_async_work) so the JS thread stays free; the promise settles on the JS main thread.
 *   streamOpen(ctx, nStreams, fs, framesPerStep, maxSpanFrames) -> external stream      (wsa_stream_create)
 *   streamInput(stream) -> Float32Array over the pinned [nStreams][samplesPerStep] input buffer (no copy)
 *   streamStep(stream, ctl: Uint8Array | null) -> {meta, feat, segments}   (wsa_stream_step_host + wsa_stream_collect;
 *       one hipGraph launch, well under a millisecond, so it runs on the calling thread)
 *   streamClose(stream)
 * Rejections carry the library's error string.  No compute happens in this file.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../../include/wsa.h"

#define NAPI_OK(env, call) do { if ((call) != napi_ok) { napi_throw_error((env), NULL, "N-API call failed: " #call); return NULL; } } while (0)

static const char *CFG_INT[] = {"spec_type", "output_level", "N_fft_bins", "N_mel_bins", "auto_noise_gate"};
static const char *CFG_DBL[] = {"f_min", "f_max", "window_width", "window_step", "pause_length", "min_seg_length",
                                "voiced_max_dB", "voiced_min_dB", "pre_norm_gain", "high_f_emph"};

static int32_t *cfg_int(wsa_config *c, int i) {
    switch (i) { case 0: return &c->spec_type; case 1: return &c->output_level; case 2: return &c->N_fft_bins;
                 case 3: return &c->N_mel_bins; default: return &c->auto_noise_gate; }
}
static double *cfg_dbl(wsa_config *c, int i) {
    switch (i) { case 0: return &c->f_min; case 1: return &c->f_max; case 2: return &c->window_width; case 3: return &c->window_step;
                 case 4: return &c->pause_length; case 5: return &c->min_seg_length; case 6: return &c->voiced_max_dB;
                 case 7: return &c->voiced_min_dB; case 8: return &c->pre_norm_gain; default: return &c->high_f_emph; }
}

static napi_value config_to_js(napi_env env, const wsa_config *c) {
    napi_value o, v;
    NAPI_OK(env, napi_create_object(env, &o));
    for (int i = 0; i < 5; i++) {
        if (i == 4) { NAPI_OK(env, napi_get_boolean(env, *cfg_int((wsa_config *)c, i) != 0, &v)); }
        else NAPI_OK(env, napi_create_int32(env, *cfg_int((wsa_config *)c, i), &v));
        NAPI_OK(env, napi_set_named_property(env, o, CFG_INT[i], v));
    }
    for (int i = 0; i < 10; i++) {
        NAPI_OK(env, napi_create_double(env, *cfg_dbl((wsa_config *)c, i), &v));
        NAPI_OK(env, napi_set_named_property(env, o, CFG_DBL[i], v));
    }
    return o;
}

static int js_to_config(napi_env env, napi_value o, wsa_config *c) {
    wsa_config_default(c);
    for (int i = 0; i < 5; i++) {
        bool has; napi_value v; napi_valuetype t;
        if (napi_has_named_property(env, o, CFG_INT[i], &has) != napi_ok || !has) continue;
        if (napi_get_named_property(env, o, CFG_INT[i], &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok) return 0;
        if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); *cfg_int(c, i) = b ? 1 : 0; }
        else if (t == napi_number) { double d; napi_get_value_double(env, v, &d); *cfg_int(c, i) = (int32_t)d; }
    }
    for (int i = 0; i < 10; i++) {
        bool has; napi_value v; napi_valuetype t;
        if (napi_has_named_property(env, o, CFG_DBL[i], &has) != napi_ok || !has) continue;
        if (napi_get_named_property(env, o, CFG_DBL[i], &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok) return 0;
        if (t == napi_number) napi_get_value_double(env, v, cfg_dbl(c, i));
    }
    return 1;
}

static napi_value fn_abi_version(napi_env env, napi_callback_info info) {
    napi_value v; NAPI_OK(env, napi_create_int32(env, wsa_abi_version(), &v)); return v;
}
static napi_value fn_defaults(napi_env env, napi_callback_info info) {
    wsa_config c; wsa_config_default(&c); return config_to_js(env, &c);
}

/* What JS holds for a context: a box that outlives wsa_destroy, so that a handle used after destroy() (or destroyed twice) finds NULL
 * instead of freed memory, and that counts the batches in flight and the open streams created from the context — destroy() refuses
 * while any of them is alive (their wsa_batch / wsa_stream objects point into the context). */
typedef struct {
    wsa_ctx *ctx; uint32_t children;
    /* the planned batch of the last processBatch call: a call with the same clip lengths and rates reuses it (planning a 1024-clip
     * batch allocates GBs of work space: ~3 ms and more); taken out of the box while a job uses it, dropped by destroy() */
    wsa_batch *plan; uint32_t plan_n; uint32_t *plan_ns; double plan_fs, plan_fs_out;
    /* the context's own HIP stream (wsa_queue_create, made by the first processBatch): every job of the context runs on it, so that the jobs of TWO
     * contexts on one device overlap — the upload of one batch under the kernels of the other — instead of queueing on the device's null stream */
    void *queue;
} ctx_box;
static void box_drop_plan(ctx_box *b) {
    if (b->plan) wsa_batch_destroy(b->plan);
    free(b->plan_ns); b->plan = NULL; b->plan_ns = NULL; b->plan_n = 0;
}
/* the communicator of gatherRows (one at a time; rebuilt when the set of contexts changes, dropped before any of its contexts is destroyed) */
static wsa_gather *g_gather = NULL; static wsa_ctx **g_gather_ctxs = NULL; static uint32_t g_gather_n = 0;
/* gatherRows jobs run on libuv worker threads, destroy() on the JS thread: every access to the three words above holds this lock
 * (a wsa_gather is not thread-safe either: one exchange at a time) */
static pthread_mutex_t g_gather_lock = PTHREAD_MUTEX_INITIALIZER;
static void gather_drop(void) {          /* caller holds g_gather_lock */
    if (g_gather) wsa_gather_destroy(g_gather);
    free(g_gather_ctxs); g_gather = NULL; g_gather_ctxs = NULL; g_gather_n = 0;
}
static void ctx_finalize(napi_env env, void *data, void *hint) {          /* the JS handle is gone */
    ctx_box *b = (ctx_box *)data;
    if (!b->ctx && b->children == 0) free(b);        /* a context nobody destroyed stays (explicit destroy() only: the finalizer may run at process exit, after the HIP runtime) */
}

static napi_value fn_create(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_config c; int32_t device = 0;
    if (argc < 1 || !js_to_config(env, argv[0], &c)) { napi_throw_type_error(env, NULL, "create(config, device)"); return NULL; }
    if (argc > 1) napi_get_value_int32(env, argv[1], &device);
    wsa_ctx *ctx = NULL;
    const wsa_status st = wsa_create(&c, device, &ctx);
    if (st != WSA_OK) { napi_throw_error(env, NULL, wsa_last_error(NULL)); return NULL; }
    ctx_box *box = calloc(1, sizeof *box);
    box->ctx = ctx;
    napi_value ext; NAPI_OK(env, napi_create_external(env, box, ctx_finalize, NULL, &ext));
    return ext;
}
static ctx_box *get_box(napi_env env, napi_value v) {
    void *p = NULL; if (napi_get_value_external(env, v, &p) != napi_ok) return NULL; return (ctx_box *)p;
}
static wsa_ctx *get_ctx(napi_env env, napi_value v) {                      /* NULL once destroy() has run */
    ctx_box *b = get_box(env, v); return b ? b->ctx : NULL;
}
static napi_value fn_destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    ctx_box *b = argc ? get_box(env, argv[0]) : NULL;
    if (b && b->ctx) {
        if (b->children) { napi_throw_error(env, NULL, "context still has batches in flight or open streams"); return NULL; }
        pthread_mutex_lock(&g_gather_lock);
        for (uint32_t i = 0; i < g_gather_n; i++) if (g_gather_ctxs[i] == b->ctx) { gather_drop(); break; }      /* the communicator goes before its contexts */
        pthread_mutex_unlock(&g_gather_lock);
        box_drop_plan(b);
        if (b->queue) { wsa_queue_destroy(b->ctx, b->queue); b->queue = NULL; }
        wsa_destroy(b->ctx); b->ctx = NULL;
    }
    return NULL;
}
/* allocPinned(ctx, bytes) -> ArrayBuffer over page-locked host memory (wsa_host_alloc): clips read into views of it reach the device by DMA at the
 * link's rate instead of through the runtime's staging copies.  The memory is released when the ArrayBuffer is collected. */
/* every page-locked buffer handed to JS has a record: freePinned(ab) releases the memory at a moment of the caller's choosing (hipHostFree synchronises the device:
 * inside the garbage collector's finalizer that stall hits the event loop whenever V8 decides) and detaches the ArrayBuffer; the finalizer then finds nothing left to do */
typedef struct pinned_rec { void *p; uint64_t bytes; int freed; struct pinned_rec *next; } pinned_rec;
static pinned_rec *g_pinned = NULL;                 /* JS thread only */
static void pinned_release(napi_env env, pinned_rec *r) {
    if (r->freed) return;
    int64_t now = 0;
    napi_adjust_external_memory(env, -(int64_t)r->bytes, &now);      /* V8 was told about the bytes at allocation: page-locked slabs do create GC pressure */
    wsa_host_free(r->p);
    r->freed = 1;
}
static void pinned_finalize(napi_env env, void *data, void *hint) {
    pinned_rec *r = (pinned_rec *)hint;
    pinned_release(env, r);
    for (pinned_rec **q = &g_pinned; *q; q = &(*q)->next) if (*q == r) { *q = r->next; break; }
    free(r);
}
static napi_value fn_alloc_pinned(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2]; double bytes = 0;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_ctx *ctx = argc ? get_ctx(env, argv[0]) : NULL;
    if (!ctx || argc < 2 || napi_get_value_double(env, argv[1], &bytes) != napi_ok || bytes < 0 || bytes > 68719476736.0) { napi_throw_type_error(env, NULL, "allocPinned(ctx, bytes)"); return NULL; }
    void *p = NULL;
    if (wsa_host_alloc(ctx, (uint64_t)bytes, &p) != WSA_OK) { napi_throw_error(env, NULL, wsa_last_error(ctx)); return NULL; }
    pinned_rec *r = calloc(1, sizeof *r);
    if (!r) { wsa_host_free(p); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    r->p = p; r->bytes = (uint64_t)bytes;
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, pinned_finalize, r, &ab) != napi_ok) { wsa_host_free(p); free(r); napi_throw_error(env, NULL, "napi_create_external_arraybuffer failed"); return NULL; }
    r->next = g_pinned; g_pinned = r;
    { int64_t now = 0; napi_adjust_external_memory(env, (int64_t)bytes, &now); }
    return ab;
}
/* freePinned(ab): the page-locked memory behind an ArrayBuffer of allocPinned goes back NOW (no run may still be reading it) and the ArrayBuffer is detached:
 * views on it have length 0 from here on.  Returns true when it was such a buffer and still allocated. */
static napi_value fn_free_pinned(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1]; void *data = NULL; size_t len = 0; bool is_ab = false;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1 || napi_is_arraybuffer(env, argv[0], &is_ab) != napi_ok || !is_ab || napi_get_arraybuffer_info(env, argv[0], &data, &len) != napi_ok) { napi_throw_type_error(env, NULL, "freePinned(arrayBuffer)"); return NULL; }
    bool done = false;
    for (pinned_rec *r = g_pinned; r; r = r->next) if (r->p == data && !r->freed) { pinned_release(env, r); done = true; break; }
    if (done) (void)napi_detach_arraybuffer(env, argv[0]);
    napi_value v; NAPI_OK(env, napi_get_boolean(env, done, &v)); return v;
}
static napi_value fn_geometry(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2]; double fs = 0;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_ctx *ctx = argc ? get_ctx(env, argv[0]) : NULL;
    if (!ctx || argc < 2 || napi_get_value_double(env, argv[1], &fs) != napi_ok) { napi_throw_type_error(env, NULL, "geometry(ctx, fs)"); return NULL; }
    wsa_geometry g;
    if (wsa_geometry_for(ctx, fs, &g) != WSA_OK) { napi_throw_error(env, NULL, wsa_last_error(ctx)); return NULL; }
    napi_value o, v; NAPI_OK(env, napi_create_object(env, &o));
    const char *names[5] = {"nfft", "win", "hop", "bands", "kmax"}; const int32_t vals[5] = {g.nfft, g.win, g.hop, g.bands, g.kmax};
    for (int i = 0; i < 5; i++) { NAPI_OK(env, napi_create_int32(env, vals[i], &v)); NAPI_OK(env, napi_set_named_property(env, o, names[i], v)); }
    return o;
}
static napi_value fn_bins_hz(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2]; double fs = 0;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_ctx *ctx = argc ? get_ctx(env, argv[0]) : NULL;
    if (!ctx || argc < 2 || napi_get_value_double(env, argv[1], &fs) != napi_ok) { napi_throw_type_error(env, NULL, "binsHz(ctx, fs)"); return NULL; }
    wsa_geometry g;
    if (wsa_geometry_for(ctx, fs, &g) != WSA_OK) { napi_throw_error(env, NULL, wsa_last_error(ctx)); return NULL; }
    napi_value ab, ta; void *data = NULL;
    NAPI_OK(env, napi_create_arraybuffer(env, sizeof(double) * (size_t)g.bands, &data, &ab));
    if (wsa_bins_hz(ctx, fs, (double *)data, g.bands) != WSA_OK) { napi_throw_error(env, NULL, wsa_last_error(ctx)); return NULL; }
    NAPI_OK(env, napi_create_typedarray(env, napi_float64_array, (size_t)g.bands, ab, 0, &ta));
    return ta;
}

/* ---- processBatch: async work ---- */
typedef struct {
    napi_async_work work; napi_deferred deferred;
    wsa_ctx *ctx; double fs; double fs_out;      /* fs_out != fs: convert in front (wsa_batch_create_resampled) */
    uint32_t n_clips; uint32_t *n_samples; const float **pcm; napi_ref *clip_refs;
    int is_i16; uint32_t *channels;   /* Int16Array clips (pcm[] then holds int16 pointers): wsa_batch_run_host_i16 */
    wsa_batch *plan; int plan_reused; /* taken from / returned to the box on the JS thread */
    /* results */
    wsa_status st; char err[512];
    uint32_t n_rows, n_segs; int32_t *meta; double *feat; int32_t *segs; uint32_t *row_off, *seg_off; float stage_ms[4];
    uint32_t n_frames; float *formants; uint32_t *frame_off;      /* levels 4 / 10 */
    uint32_t n_utt; int32_t *utt_meta; double *utt_feat; uint32_t *utt_off;   /* level 11 */
    int defer_rows;               /* the rows stay on the device (gatherRows collects them from all shards with one RCCL exchange) */
    int level; uint32_t trk_segs; uint64_t trk_np, trk_nr; uint64_t *trk_off; int32_t *trk_pts, *trk_rank;   /* level 3 */
    ctx_box *box;                 /* the JS handle's box: one child while the job runs */
    void *queue;                  /* the context's stream (ctx_box.queue) */
} job_t;

static void job_execute(napi_env env, void *data) {
    job_t *j = (job_t *)data;
    wsa_batch *b = j->plan;
    if (!j->box->queue) {           /* (one job at a time uses a context, so nobody else looks at the box's stream now) */
        j->st = wsa_queue_create(j->ctx, &j->box->queue);
        if (j->st != WSA_OK) { snprintf(j->err, sizeof j->err, "%s", wsa_last_error(j->ctx)); return; }
    }
    j->queue = j->box->queue;
    if (!b) {
        j->st = (j->fs_out > 0 && j->fs_out != j->fs) ? wsa_batch_create_resampled(j->ctx, j->n_clips, j->n_samples, j->fs, j->fs_out, &b)
                                                      : wsa_batch_create(j->ctx, j->n_clips, j->n_samples, j->fs, &b);
        if (j->st != WSA_OK) { snprintf(j->err, sizeof j->err, "%s", wsa_last_error(j->ctx)); return; }
        j->plan = b;
    }
    do {
        j->st = j->is_i16 ? wsa_batch_run_host_i16(b, (const int16_t *const *)j->pcm, j->channels, j->queue) : wsa_batch_run_host(b, j->pcm, j->queue);
        if (j->st != WSA_OK) break;
        wsa_device_result r;
        j->st = wsa_batch_result(b, j->queue, &r);
        if (j->st != WSA_OK) break;
        j->n_rows = r.n_rows; j->n_segs = r.n_segments;
        j->meta = malloc(sizeof(int32_t) * 8 * (size_t)(r.n_rows ? r.n_rows : 1));
        j->feat = malloc(sizeof(double) * WSA_NFEAT * (size_t)(r.n_rows ? r.n_rows : 1));
        j->segs = malloc(sizeof(int32_t) * 4 * (size_t)(r.n_segments ? r.n_segments : 1));
        j->row_off = malloc(sizeof(uint32_t) * ((size_t)j->n_clips + 1));
        j->seg_off = malloc(sizeof(uint32_t) * ((size_t)j->n_clips + 1));
        if (!j->meta || !j->feat || !j->segs || !j->row_off || !j->seg_off) { j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
        j->st = wsa_batch_copy_rows(b, j->queue, j->defer_rows ? NULL : j->meta, j->defer_rows ? NULL : j->feat, r.n_rows ? r.n_rows : 1, j->segs, r.n_segments ? r.n_segments : 1, j->row_off, j->seg_off);
        if (j->st != WSA_OK) break;
        if (r.d_formants) {                                   /* levels 4 / 10: the straightened frames */
            j->n_frames = r.n_frames_total;
            j->formants = malloc(sizeof(float) * 9 * (size_t)(r.n_frames_total ? r.n_frames_total : 1));
            j->frame_off = malloc(sizeof(uint32_t) * ((size_t)j->n_clips + 1));
            if (!j->formants || !j->frame_off) { j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
            j->st = wsa_batch_copy_formants(b, j->queue, j->formants, r.n_frames_total ? r.n_frames_total : 1);
            if (j->st != WSA_OK) break;
            j->st = wsa_batch_copy_spectra(b, j->queue, NULL, 0, j->frame_off);
            if (j->st != WSA_OK) break;
        }
        if (r.d_utt_feat) {                                   /* level 11: utterance features after every result */
            j->n_utt = r.n_utterance_rows;
            j->utt_meta = malloc(sizeof(int32_t) * 4 * (size_t)(j->n_utt ? j->n_utt : 1));
            j->utt_feat = malloc(sizeof(double) * WSA_NUTT * (size_t)(j->n_utt ? j->n_utt : 1));
            j->utt_off = malloc(sizeof(uint32_t) * ((size_t)j->n_clips + 1));
            if (!j->utt_meta || !j->utt_feat || !j->utt_off) { j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
            j->st = wsa_batch_copy_utterance(b, j->queue, j->utt_meta, j->utt_feat, j->n_utt ? j->n_utt : 1, j->utt_off);
            if (j->st != WSA_OK) break;
        }
        if (j->level == 3) {                                  /* level 3: the ranked raw tracks (points + ranked ids per segment) */
            wsa_tracks_info ti;
            j->st = wsa_batch_tracks_info(b, j->queue, &ti);
            if (j->st != WSA_OK) break;
            j->trk_segs = ti.n_segments; j->trk_np = ti.n_points; j->trk_nr = ti.n_ranked;
            j->trk_off = malloc(sizeof(uint64_t) * 2 * ((size_t)ti.n_segments + 1));
            j->trk_pts = malloc(sizeof(int32_t) * 8 * (size_t)(ti.n_points ? ti.n_points : 1));
            j->trk_rank = malloc(sizeof(int32_t) * (size_t)(ti.n_ranked ? ti.n_ranked : 1));
            if (!j->trk_off || !j->trk_pts || !j->trk_rank) { j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
            j->st = wsa_batch_copy_tracks(b, j->queue, j->trk_off, j->trk_pts, ti.n_points, j->trk_rank, ti.n_ranked);
            if (j->st != WSA_OK) break;
        }
        wsa_batch_stage_ms(b, j->stage_ms);
    } while (0);
    if (j->st != WSA_OK) snprintf(j->err, sizeof j->err, "%s", wsa_last_error(j->ctx));
    /* the plan goes back to the box in job_complete (JS thread) */
}

static napi_value make_typed(napi_env env, napi_typedarray_type type, const void *src, size_t count, size_t elt) {
    napi_value ab, ta; void *data = NULL;
    if (napi_create_arraybuffer(env, count * elt, &data, &ab) != napi_ok) return NULL;
    if (count) memcpy(data, src, count * elt);
    if (napi_create_typedarray(env, type, count, ab, 0, &ta) != napi_ok) return NULL;
    return ta;
}

static void job_complete(napi_env env, napi_status status, void *data) {
    job_t *j = (job_t *)data;
    if (j->box && j->box->children) j->box->children--;
    if (j->plan) {                       /* keep the plan for the next call of the same shape (one entry; a failed run drops it) */
        if (j->box && j->box->ctx && j->st == WSA_OK && !j->box->plan) {
            j->box->plan = j->plan; j->box->plan_n = j->n_clips; j->box->plan_fs = j->fs; j->box->plan_fs_out = j->fs_out;
            j->box->plan_ns = j->n_samples; j->n_samples = NULL;
        } else wsa_batch_destroy(j->plan);
        j->plan = NULL;
    }
    for (uint32_t i = 0; i < j->n_clips; i++) napi_delete_reference(env, j->clip_refs[i]);
    if (status != napi_ok || j->st != WSA_OK) {
        napi_value msg;
        napi_create_string_utf8(env, j->st != WSA_OK ? j->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_reject_deferred(env, j->deferred, msg);          /* the reference rejects with strings (ref @B4554) */
    } else {
        napi_value o;
        napi_create_object(env, &o);
        if (!j->defer_rows) {
            napi_set_named_property(env, o, "meta", make_typed(env, napi_int32_array, j->meta, (size_t)j->n_rows * 8, 4));
            napi_set_named_property(env, o, "feat", make_typed(env, napi_float64_array, j->feat, (size_t)j->n_rows * WSA_NFEAT, 8));
        }
        napi_set_named_property(env, o, "segments", make_typed(env, napi_int32_array, j->segs, (size_t)j->n_segs * 4, 4));
        napi_set_named_property(env, o, "rowOff", make_typed(env, napi_uint32_array, j->row_off, (size_t)j->n_clips + 1, 4));
        napi_set_named_property(env, o, "segOff", make_typed(env, napi_uint32_array, j->seg_off, (size_t)j->n_clips + 1, 4));
        napi_set_named_property(env, o, "stageMs", make_typed(env, napi_float32_array, j->stage_ms, 4, 4));
        if (j->utt_feat) {
            napi_set_named_property(env, o, "uttMeta", make_typed(env, napi_int32_array, j->utt_meta, (size_t)j->n_utt * 4, 4));
            napi_set_named_property(env, o, "uttFeat", make_typed(env, napi_float64_array, j->utt_feat, (size_t)j->n_utt * WSA_NUTT, 8));
            napi_set_named_property(env, o, "uttOff", make_typed(env, napi_uint32_array, j->utt_off, (size_t)j->n_clips + 1, 4));
        }
        if (j->trk_off) {
            /* offsets as doubles (exact below 2^53): [n_segments + 1][2] = first point / first ranked id of a segment */
            const size_t n = 2 * ((size_t)j->trk_segs + 1);
            double *od = malloc(sizeof(double) * n);
            for (size_t i = 0; i < n; i++) od[i] = (double)j->trk_off[i];
            napi_set_named_property(env, o, "trackOff", make_typed(env, napi_float64_array, od, n, 8));
            free(od);
            napi_set_named_property(env, o, "trackPoints", make_typed(env, napi_int32_array, j->trk_pts, (size_t)j->trk_np * 8, 4));
            napi_set_named_property(env, o, "trackRanked", make_typed(env, napi_int32_array, j->trk_rank, (size_t)j->trk_nr, 4));
        }
        if (j->formants) {
            napi_set_named_property(env, o, "formants", make_typed(env, napi_float32_array, j->formants, (size_t)j->n_frames * 9, 4));
            napi_set_named_property(env, o, "frameOff", make_typed(env, napi_uint32_array, j->frame_off, (size_t)j->n_clips + 1, 4));
        }
        napi_resolve_deferred(env, j->deferred, o);
    }
    napi_delete_async_work(env, j->work);
    free(j->meta); free(j->feat); free(j->segs); free(j->row_off); free(j->seg_off); free(j->formants); free(j->frame_off); free(j->utt_meta); free(j->utt_feat); free(j->utt_off); free(j->trk_off); free(j->trk_pts); free(j->trk_rank);
    free(j->n_samples); free((void *)j->pcm); free(j->clip_refs); free(j->channels); free(j);
}

static napi_value fn_process_batch(napi_env env, napi_callback_info info) {
    size_t argc = 7; napi_value argv[7];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_ctx *ctx = argc ? get_ctx(env, argv[0]) : NULL;
    bool is_arr = false; double fs = 0; uint32_t n = 0;
    if (!ctx || argc < 3 || napi_is_array(env, argv[1], &is_arr) != napi_ok || !is_arr ||
        napi_get_value_double(env, argv[2], &fs) != napi_ok || napi_get_array_length(env, argv[1], &n) != napi_ok) {
        napi_throw_type_error(env, NULL, "processBatch(ctx, Float32Array[] | Int16Array[], fs[, level[, analysisRate[, channels[, deferRows]]]])"); return NULL;
    }
    job_t *j = calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    if (argc >= 7) { bool d = false; if (napi_get_value_bool(env, argv[6], &d) == napi_ok) j->defer_rows = d ? 1 : 0; }
    j->ctx = ctx; j->fs = fs; j->n_clips = n; j->box = get_box(env, argv[0]);
    if (argc >= 4) { int32_t lv = 0; if (napi_get_value_int32(env, argv[3], &lv) == napi_ok) j->level = lv; }
    if (argc >= 5) { double fo = 0; if (napi_get_value_double(env, argv[4], &fo) == napi_ok) j->fs_out = fo; }           /* analysis rate */   /* the ctx's output_level: 3 adds the raw tracks */
    j->n_samples = calloc(n ? n : 1, sizeof(uint32_t)); j->pcm = calloc(n ? n : 1, sizeof(float *)); j->clip_refs = calloc(n ? n : 1, sizeof(napi_ref));
    if (!j->n_samples || !j->pcm || !j->clip_refs) { free(j->n_samples); free((void *)j->pcm); free(j->clip_refs); free(j); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    /* clips: all Float32Array (mono floats) or all Int16Array (16-bit PCM as a WAV file holds it, interleaved over channels[i] channels
     * given by the optional 6th argument, a Uint32Array; channel 0 is analysed and the conversion runs on the device) */
    uint32_t *chan = NULL; size_t chan_len = 0;
    if (argc >= 6) {
        napi_typedarray_type ct; void *cd = NULL; bool cta = false;
        if (napi_is_typedarray(env, argv[5], &cta) == napi_ok && cta && napi_get_typedarray_info(env, argv[5], &ct, &chan_len, &cd, NULL, NULL) == napi_ok && ct == napi_uint32_array) chan = (uint32_t *)cd;
    }
    for (uint32_t i = 0; i < n; i++) {
        napi_value el; napi_typedarray_type tt; size_t len; void *data; bool is_ta = false;
        const char *bad = NULL;
        if (napi_get_element(env, argv[1], i, &el) != napi_ok || napi_is_typedarray(env, el, &is_ta) != napi_ok || !is_ta ||
            napi_get_typedarray_info(env, el, &tt, &len, &data, NULL, NULL) != napi_ok || (tt != napi_float32_array && tt != napi_int16_array)) bad = "every clip must be a Float32Array or an Int16Array";
        else if (i > 0 && (tt == napi_int16_array) != (j->is_i16 != 0)) bad = "the clips of one batch must be of one kind";
        if (bad) {
            for (uint32_t k = 0; k < i; k++) napi_delete_reference(env, j->clip_refs[k]);
            free(j->n_samples); free((void *)j->pcm); free(j->clip_refs); free(j->channels); free(j);
            napi_throw_type_error(env, NULL, bad); return NULL;
        }
        if (i == 0) { j->is_i16 = tt == napi_int16_array; if (j->is_i16) j->channels = calloc(n, sizeof(uint32_t)); }
        uint32_t ch = 1;
        if (j->is_i16) { ch = (chan && i < chan_len && chan[i] >= 1) ? chan[i] : 1; j->channels[i] = ch; }
        j->n_samples[i] = (uint32_t)(len / ch); j->pcm[i] = (const float *)data;
        napi_create_reference(env, el, 1, &j->clip_refs[i]);      /* keep the PCM alive while the worker reads it */
    }
    /* a plan of exactly this shape waiting in the box?  (one job at a time uses a context: the plan leaves the box while it runs) */
    if (j->box->plan) {
        int same = j->box->plan_n == n && j->box->plan_fs == j->fs && j->box->plan_fs_out == j->fs_out;
        for (uint32_t i = 0; same && i < n; i++) same = j->box->plan_ns[i] == j->n_samples[i];
        if (same) { j->plan = j->box->plan; j->plan_reused = 1; j->box->plan = NULL; free(j->box->plan_ns); j->box->plan_ns = NULL; j->box->plan_n = 0; }
        else box_drop_plan(j->box);
    }
    napi_value promise, name;
    NAPI_OK(env, napi_create_promise(env, &j->deferred, &promise));
    NAPI_OK(env, napi_create_string_utf8(env, "wsa.processBatch", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(env, napi_create_async_work(env, NULL, name, job_execute, job_complete, j, &j->work));
    NAPI_OK(env, napi_queue_async_work(env, j->work));
    j->box->children++;                                          /* until job_complete */
    return promise;
}

/* ---- gatherRows: the rows of the contexts' last batches (processBatch(..., deferRows = true)) collected on the first context's device by one
 * RCCL exchange (wsa_gather_rows) and copied to the host once; resolves {meta, feat, rowsPerRank} ---- */
typedef struct {
    napi_async_work work; napi_deferred deferred;
    uint32_t n; ctx_box **boxes; wsa_ctx **ctxs; wsa_batch **plans;
    wsa_status st; char err[512];
    uint32_t n_rows; uint32_t *per; int32_t *meta; double *feat;
} gjob_t;
static void gjob_execute_locked(gjob_t *j);
static void gjob_execute(napi_env env, void *data) {
    pthread_mutex_lock(&g_gather_lock);
    gjob_execute_locked((gjob_t *)data);
    pthread_mutex_unlock(&g_gather_lock);
}
static void gjob_execute_locked(gjob_t *j) {
    int same = g_gather && g_gather_n == j->n;
    for (uint32_t i = 0; same && i < j->n; i++) same = g_gather_ctxs[i] == j->ctxs[i];
    if (!same) {
        gather_drop();
        j->st = wsa_gather_create(j->ctxs, (int32_t)j->n, 0, &g_gather);
        if (j->st != WSA_OK) { snprintf(j->err, sizeof j->err, "%s", wsa_last_error(j->ctxs[0])); g_gather = NULL; return; }
        g_gather_ctxs = malloc(sizeof(wsa_ctx *) * j->n); g_gather_n = j->n;
        if (!g_gather_ctxs) { gather_drop(); j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
        memcpy(g_gather_ctxs, j->ctxs, sizeof(wsa_ctx *) * j->n);
    }
    wsa_gather_result r;
    j->st = wsa_gather_rows(g_gather, j->plans, NULL, &r);
    if (j->st == WSA_OK) {
        j->n_rows = r.n_rows;
        j->per = malloc(sizeof(uint32_t) * j->n);
        j->meta = malloc(sizeof(int32_t) * 8 * (size_t)(r.n_rows ? r.n_rows : 1));
        j->feat = malloc(sizeof(double) * WSA_NFEAT * (size_t)(r.n_rows ? r.n_rows : 1));
        if (!j->per || !j->meta || !j->feat) { j->st = WSA_ERR_INVALID; snprintf(j->err, sizeof j->err, "out of memory"); return; }
        memcpy(j->per, r.rows_per_rank, sizeof(uint32_t) * j->n);
        j->st = wsa_gather_copy_rows(g_gather, j->meta, j->feat, r.n_rows ? r.n_rows : 1);
    }
    if (j->st != WSA_OK) { snprintf(j->err, sizeof j->err, "%s", wsa_last_error(j->ctxs[0])); gather_drop(); }      /* a failed exchange leaves the communicators unusable: the next job builds new ones */
}
static void gjob_complete(napi_env env, napi_status status, void *data) {
    gjob_t *j = (gjob_t *)data;
    for (uint32_t i = 0; i < j->n; i++) if (j->boxes[i]->children) j->boxes[i]->children--;
    if (status != napi_ok || j->st != WSA_OK) {
        napi_value msg;
        napi_create_string_utf8(env, j->st != WSA_OK ? j->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_reject_deferred(env, j->deferred, msg);
    } else {
        napi_value o; napi_create_object(env, &o);
        napi_set_named_property(env, o, "meta", make_typed(env, napi_int32_array, j->meta, (size_t)j->n_rows * 8, 4));
        napi_set_named_property(env, o, "feat", make_typed(env, napi_float64_array, j->feat, (size_t)j->n_rows * WSA_NFEAT, 8));
        napi_set_named_property(env, o, "rowsPerRank", make_typed(env, napi_uint32_array, j->per, j->n, 4));
        napi_resolve_deferred(env, j->deferred, o);
    }
    napi_delete_async_work(env, j->work);
    free(j->per); free(j->meta); free(j->feat); free(j->boxes); free(j->ctxs); free(j->plans); free(j);
}
static napi_value fn_gather_rows(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    bool is_arr = false; uint32_t n = 0;
    if (argc < 1 || napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[0], &n) != napi_ok || n < 1) {
        napi_throw_type_error(env, NULL, "gatherRows(ctx[])"); return NULL;
    }
    gjob_t *j = calloc(1, sizeof *j);
    if (j) { j->n = n; j->boxes = calloc(n, sizeof(ctx_box *)); j->ctxs = calloc(n, sizeof(wsa_ctx *)); j->plans = calloc(n, sizeof(wsa_batch *)); }
    if (!j || !j->boxes || !j->ctxs || !j->plans) { if (j) { free(j->boxes); free(j->ctxs); free(j->plans); free(j); } napi_throw_error(env, NULL, "out of memory"); return NULL; }
    for (uint32_t i = 0; i < n; i++) {
        napi_value el;
        ctx_box *b = napi_get_element(env, argv[0], i, &el) == napi_ok ? get_box(env, el) : NULL;
        if (!b || !b->ctx || !b->plan) {       /* the rows to collect are those of the plan the context's last processBatch left in its box */
            free(j->boxes); free(j->ctxs); free(j->plans); free(j);
            napi_throw_error(env, NULL, "gatherRows: every context needs a finished processBatch(..., deferRows = true)"); return NULL;
        }
        j->boxes[i] = b; j->ctxs[i] = b->ctx; j->plans[i] = b->plan;
    }
    napi_value promise, name;
    NAPI_OK(env, napi_create_promise(env, &j->deferred, &promise));
    NAPI_OK(env, napi_create_string_utf8(env, "wsa.gatherRows", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(env, napi_create_async_work(env, NULL, name, gjob_execute, gjob_complete, j, &j->work));
    NAPI_OK(env, napi_queue_async_work(env, j->work));
    for (uint32_t i = 0; i < n; i++) j->boxes[i]->children++;      /* the contexts (and their plans) stay until gjob_complete */
    return promise;
}

/* ---- streams ---- */
typedef struct { wsa_stream *st; wsa_ctx *ctx; ctx_box *box; uint32_t n, sps; napi_ref input_ref; } stream_t;   /* input_ref: the ArrayBuffer over the pinned input, detached at close */
static void stream_finalize(napi_env env, void *data, void *hint) { /* explicit streamClose() only */ }
static stream_t *get_stream(napi_env env, napi_value v) {
    void *p = NULL; if (napi_get_value_external(env, v, &p) != napi_ok) return NULL; return (stream_t *)p;
}
static napi_value fn_stream_open(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5]; double fs = 0; uint32_t n = 0, fps = 1, span = 1024;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    wsa_ctx *ctx = argc ? get_ctx(env, argv[0]) : NULL;
    if (!ctx || argc < 3 || napi_get_value_uint32(env, argv[1], &n) != napi_ok || napi_get_value_double(env, argv[2], &fs) != napi_ok) {
        napi_throw_type_error(env, NULL, "streamOpen(ctx, nStreams, fs, framesPerStep, maxSpanFrames)"); return NULL;
    }
    if (argc > 3) napi_get_value_uint32(env, argv[3], &fps);
    if (argc > 4) napi_get_value_uint32(env, argv[4], &span);
    stream_t *h = calloc(1, sizeof *h);
    h->ctx = ctx; h->n = n; h->box = get_box(env, argv[0]);
    if (wsa_stream_create(ctx, n, fs, fps, span, &h->st) != WSA_OK) { free(h); napi_throw_error(env, NULL, wsa_last_error(ctx)); return NULL; }
    h->box->children++;                                          /* until streamClose */
    wsa_stream_enable_graph(h->st, 1);
    h->sps = wsa_stream_samples_per_step(h->st);
    napi_value ext; NAPI_OK(env, napi_create_external(env, h, stream_finalize, NULL, &ext));
    return ext;
}
static napi_value fn_stream_input(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    stream_t *h = argc ? get_stream(env, argv[0]) : NULL;
    if (!h || !h->st) { napi_throw_type_error(env, NULL, "streamInput(stream)"); return NULL; }
    const size_t count = (size_t)h->n * h->sps;
    napi_value ab, ta;
    if (h->input_ref) {                      /* one ArrayBuffer per stream object: hand the same one out again */
        NAPI_OK(env, napi_get_reference_value(env, h->input_ref, &ab));
    } else {
        NAPI_OK(env, napi_create_external_arraybuffer(env, wsa_stream_host_input(h->st), count * sizeof(float), NULL, NULL, &ab));
        NAPI_OK(env, napi_create_reference(env, ab, 1, &h->input_ref));
    }
    NAPI_OK(env, napi_create_typedarray(env, napi_float32_array, count, ab, 0, &ta));
    return ta;
}
static napi_value fn_stream_step(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    stream_t *h = argc ? get_stream(env, argv[0]) : NULL;
    if (!h || !h->st) { napi_throw_type_error(env, NULL, "streamStep(stream, ctl)"); return NULL; }
    const uint8_t *ctl = NULL;
    if (argc > 1) {
        bool is_ta = false; napi_typedarray_type tt; size_t len; void *data;
        if (napi_is_typedarray(env, argv[1], &is_ta) == napi_ok && is_ta) {
            if (napi_get_typedarray_info(env, argv[1], &tt, &len, &data, NULL, NULL) != napi_ok || tt != napi_uint8_array || len != h->n) {
                napi_throw_type_error(env, NULL, "ctl must be a Uint8Array with one byte per stream"); return NULL;
            }
            ctl = (const uint8_t *)data;
        }
    }
    wsa_stream_rows r;
    if (wsa_stream_step_host(h->st, ctl, NULL) != WSA_OK || wsa_stream_collect(h->st, NULL, &r) != WSA_OK) {
        napi_throw_error(env, NULL, wsa_last_error(h->ctx)); return NULL;
    }
    napi_value o;
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "meta", make_typed(env, napi_int32_array, r.row_meta, (size_t)r.n_rows * 8, 4));
    napi_set_named_property(env, o, "feat", make_typed(env, napi_float64_array, r.row_feat, (size_t)r.n_rows * WSA_NFEAT, 8));
    napi_set_named_property(env, o, "segments", make_typed(env, napi_int32_array, r.segments, (size_t)r.n_segments * 4, 4));
    napi_set_named_property(env, o, "cuts", make_typed(env, napi_uint32_array, r.stream_cuts, (size_t)h->n, 4));
    if (r.utt_feat) {                               /* level 11: one 264-vector per result of the step */
        napi_set_named_property(env, o, "uttMeta", make_typed(env, napi_int32_array, r.utt_meta, (size_t)r.n_utterance_rows * 4, 4));
        napi_set_named_property(env, o, "uttFeat", make_typed(env, napi_float64_array, r.utt_feat, (size_t)r.n_utterance_rows * WSA_NUTT, 8));
    }
    if (r.formants && r.row_formant_off) {          /* levels 4 / 10: the straightened frames of the rows */
        napi_set_named_property(env, o, "formantOff", make_typed(env, napi_uint32_array, r.row_formant_off, (size_t)r.n_rows + 1, 4));
        napi_set_named_property(env, o, "formants", make_typed(env, napi_float32_array, r.formants, (size_t)r.row_formant_off[r.n_rows] * 9, 4));
    }
    if (r.track_off) {                              /* level 3: ranked raw tracks of the step's segments, as LaunchBatch's trackOff / trackPoints / trackRanked */
        const size_t n = 2 * ((size_t)r.n_segments + 1);
        double *od = malloc(sizeof(double) * n);
        for (size_t i = 0; i < n; i++) od[i] = (double)r.track_off[i];
        napi_set_named_property(env, o, "trackOff", make_typed(env, napi_float64_array, od, n, 8));
        free(od);
        napi_set_named_property(env, o, "trackPoints", make_typed(env, napi_int32_array, r.track_points, (size_t)r.n_track_points * 8, 4));
        napi_set_named_property(env, o, "trackRanked", make_typed(env, napi_int32_array, r.track_ranked, (size_t)r.n_track_ranked, 4));
    }
    { napi_value fl; napi_create_uint32(env, r.status_flags, &fl); napi_set_named_property(env, o, "flags", fl); }   /* WSA_FLAG_* (8: a span was cut in this step) */
    return o;
}
static napi_value fn_stream_close(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    stream_t *h = argc ? get_stream(env, argv[0]) : NULL;
    if (h && h->st) {
        if (h->input_ref) {                  /* the pinned buffer goes away with the stream object: views on it must not outlive it */
            napi_value ab;
            if (napi_get_reference_value(env, h->input_ref, &ab) == napi_ok && ab) napi_detach_arraybuffer(env, ab);
            napi_delete_reference(env, h->input_ref); h->input_ref = NULL;
        }
        wsa_stream_destroy(h->st); h->st = NULL;
        if (h->box && h->box->children) h->box->children--;
    }
    return NULL;
}

NAPI_MODULE_INIT() {
    /* the structures below follow the header this file was compiled against: refuse a libwsa.so of another ABI version */
    if (wsa_abi_version() != WSA_ABI_VERSION) { napi_throw_error(env, NULL, "libwsa.so ABI version differs from the one wsa_napi.node was built against (include/wsa.h): rebuild"); return NULL; }
    const struct { const char *name; napi_callback fn; } fns[] = {
        {"abiVersion", fn_abi_version}, {"freePinned", fn_free_pinned}, {"defaults", fn_defaults}, {"create", fn_create}, {"destroy", fn_destroy},
        {"geometry", fn_geometry}, {"allocPinned", fn_alloc_pinned}, {"binsHz", fn_bins_hz}, {"processBatch", fn_process_batch}, {"gatherRows", fn_gather_rows},
        {"streamOpen", fn_stream_open}, {"streamInput", fn_stream_input}, {"streamStep", fn_stream_step}, {"streamClose", fn_stream_close}};
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

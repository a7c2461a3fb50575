// api.hip — the C ABI of include/wsa.h: context, batch plans, stage launches, result tables.
// Host-side mirror of the reference's module-level state (ref dist/main.js:2 inner module 1,
// @B2750-5843: config object, LaunchAudioNodes orchestration) for the batch use of the hot path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <system_error>
#include <thread>
#include "wsa_internal.hpp"

using namespace wsa;

#include "api_internal.hpp"

thread_local std::string wsa_api::g_create_error;
using wsa_api::fail;

struct wsa_batch {
    wsa_ctx* ctx = nullptr;
    uint32_t n_clips = 0;
    double fs = 0;
    std::vector<uint32_t> n_samples, n_frames, frame_off;
    uint32_t total_frames = 0, max_frames = 0, max_samples = 0;
    FePlanHost plan;
    int seg_cap = 0, row_cap = 0, tcap = 0, pcap = 0, fcap = 0, n_waves = 0;
    size_t ws_stride = 0, dev_bytes = 0;
    // device memory
    std::vector<void*> allocs;
    float *d_window = nullptr, *d_mel_w = nullptr, *d_emph = nullptr;
    float2 *d_tw_n2 = nullptr, *d_tw_64 = nullptr, *d_tw_nfft = nullptr, *d_tw_m = nullptr;
    int32_t *d_mel_k0 = nullptr, *d_mel_cnt = nullptr, *d_mel_off = nullptr;
    uint32_t *d_n_frames = nullptr, *d_frame_off = nullptr, *d_spec = nullptr;
    RecPtrs rec = {nullptr, nullptr, nullptr};      // frame records (wsa_internal.hpp)
    char* d_ws = nullptr;
    int32_t *d_seg_i = nullptr, *d_meta_pool = nullptr, *d_seg = nullptr, *d_meta = nullptr, *d_fr_info = nullptr;
    double *d_seg_d = nullptr, *d_feat_pool = nullptr, *d_feat = nullptr, *d_fr_v = nullptr, *d_fr_fl = nullptr;
    uint2* d_order = nullptr;            // spans sorted by length, longest first (launch_span_order)
    uint2* d_redo = nullptr;             // spans the paired tracker variant hands to the one-span kernel
    char* d_pool = nullptr; double* d_span_hdr = nullptr; bool split = false; size_t pool_bpf = 0;   // split finalize: span regions (pool_bpf bytes per frame) + a header per span
    hipStream_t up_stream[8] = {}; hipEvent_t up_event[8] = {}, up_start = nullptr; bool up_ready = false;   // upload_clips
    int16_t* d_i16 = nullptr; uint64_t i16_cap = 0; uint64_t* d_i16_off = nullptr; uint32_t *d_i16_ch = nullptr, *d_i16_ns = nullptr;   // wsa_batch_run_host_i16: upload buffer (own allocation, grows) + clip tables
    std::vector<uint64_t> h_i16_off; std::vector<uint32_t> h_i16_ch;
    bool pair = false;                   // tracker: two spans per wave (tracker_kernel_pair)
    bool published = false;              // the last back-end run's compaction kernel has handed the result counters to the host itself
    Tuning tune;                         // tuning / test switches, read from the environment when the batch is planned
    uint32_t* d_span_hist = nullptr; uint2* d_span_key = nullptr;       // the gate's part of that sort: bucket counts, {bucket, rank} per segment
    uint32_t *d_seg_count = nullptr, *d_clip_rows = nullptr, *d_counters = nullptr, *d_row_off = nullptr, *d_seg_off = nullptr, *d_totals = nullptr;
    float* d_pcm_own = nullptr;
    float* d_formants = nullptr;            // levels 4 / 10 / 11: [total_frames][9]
    // sample-rate conversion in front of the path (wsa_batch_create_resampled): input lengths / rate, offset-kernel table, converted PCM
    bool rs_on = false; double fs_in = 0; std::vector<uint32_t> n_samples_in; uint32_t max_samples_in = 0;
    uint32_t *d_rs_n_in = nullptr, *d_rs_n_out = nullptr; float *d_rs_table = nullptr, *d_rs_pcm = nullptr; uint64_t rs_stride = 0;
    int4* d_trk_pts = nullptr; int32_t* d_trk_rank = nullptr; int32_t* d_trk_seg = nullptr;      // level 3: raw-track pools (TrParams)
    bool counters_clean = false, end_clears = false, capturing = false, ever_captured = false;      // the fused compaction of the previous run has left the counters cleared: the next run launches no clear kernel
    char* d_trk_stage = nullptr; size_t trk_stage_cap = 0; std::vector<uint64_t> h_trk_desc;        // level 3: wsa_batch_copy_tracks gathers through this
    std::vector<int32_t> h_trk_seg; std::vector<uint32_t> h_seg_count;                          // level 3: host copies for wsa_batch_copy_tracks
    float* d_sums = nullptr; double* d_coef_ws = nullptr;    // level 12
    int32_t* d_utt_meta = nullptr; double* d_utt_feat = nullptr; uint32_t* d_utt_off = nullptr;   // level 11
    uint32_t res_utt = 0;
    double* d_trace = nullptr;
    uint32_t* h_totals = nullptr;           // pinned + mapped: rows, segs, flags, utterance results — written by batch_publish_kernel at the end of every run
    uint32_t* h_totals_dev = nullptr;       // ... as the device sees it
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timing = true, ran = false, full_table = false;
    uint32_t reruns = 0;
    uint32_t res_rows = 0, res_segs = 0, res_flags = 0;
    const uint32_t* spec_in_use = nullptr;
};

template <typename T>
static bool dev_alloc(wsa_batch* b, T** p, size_t count) {
    const size_t bytes = (count ? count : 1) * sizeof(T);
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) return false;
    b->allocs.push_back(q); b->dev_bytes += bytes;
    *p = reinterpret_cast<T*>(q);
    return true;
}
template <typename T, typename U>
static bool dev_upload(wsa_batch* b, T** p, const std::vector<U>& v) {
    static_assert(sizeof(U) <= sizeof(T) && sizeof(T) % sizeof(U) == 0, "upload type");
    const size_t count = v.size() * sizeof(U) / sizeof(T);
    if (!dev_alloc(b, p, count)) return false;
    if (!v.empty() && hipMemcpy(*p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice) != hipSuccess) return false;
    return true;
}

namespace wsa {
Tuning Tuning::from_env() {
    Tuning t;
    // A library that a host process embeds does not listen to the environment: the switches below (tuning experiments and the equivalence tests'
    // hooks; tools/README.md) are read only when WSA_TUNING_ENV=1 says so — tests/conftest.py and the tools/ scripts set it.
    { const char* on = std::getenv("WSA_TUNING_ENV"); if (!on || on[0] != '1') return t; }
    auto num = [](const char* name, int dflt) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; };
    t.dbg = num("WSA_DBG", 0);
    t.no_pair = std::getenv("WSA_NO_PAIR") != nullptr; t.no_split = std::getenv("WSA_NO_SPLIT") != nullptr; t.no_quad = std::getenv("WSA_NO_QUAD") != nullptr; t.quad = std::getenv("WSA_QUAD") != nullptr; t.no_fuse = std::getenv("WSA_NO_FUSE") != nullptr;
    t.fe_fat = std::getenv("WSA_FE_FAT") != nullptr; t.peaks_lanes = std::getenv("WSA_PEAKS_LANES") != nullptr;
    t.full_table = num("WSA_FULL_TABLE", -1);
    t.tracker_wpc = num("WSA_TRACKER_WPC", 0); t.fin_wpc = num("WSA_FIN_WPC", 0); t.fpw = num("WSA_FPW", 0);
    t.fe_wg_per_cu = num("WSA_FE_WGS", 0); t.fe_no_queue = std::getenv("WSA_FE_NO_QUEUE") != nullptr; t.peaks_wpc = num("WSA_PEAKS_WPC", 0); t.peaks_w = num("WSA_PEAKS_W", 0); t.upload_threads = num("WSA_UPLOAD_THREADS", 0);
    t.rs_s = num("WSA_RS_S", 0); t.rs_j = num("WSA_RS_J", 0); t.rs_c = num("WSA_RS_C", 0);
    return t;
}
// run prologue: work-queue counters and totals back to zero.  A kernel, not hipMemsetAsync: memset / memcpy nodes of a
// captured graph did not replay reliably on ROCm 7.2 / gfx950 (see stream_api.hip), a kernel node does.
// int16 PCM as uploaded (clip c at in + off[c] values, interleaved over ch[c] channels) -> float32 channel 0 of every clip, x / 32768 (exact)
__global__ void pcm_i16_to_f32_kernel(const int16_t* in, const uint64_t* off, const uint32_t* ch, const uint32_t* ns, float* out, uint64_t stride) {
    const uint32_t c = blockIdx.y;
    const uint32_t n = ns[c], k = ch[c];
    const int16_t* src = in + off[c];
    float* dst = out + (uint64_t)c * stride;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (float)src[(uint64_t)i * k] * (1.0f / 32768.0f);
}
// run epilogue: the four result counters go to the batch's mapped pinned words, so that the host reads them behind a stream synchronize.  (Three
// 4-byte hipMemcpyAsync D2H were three blit-kernel dispatches: behind the other batches' kernels each waited 50 - 300 us for a free slot on a full
// GPU, ~430 us per step before the host heard that the batch was done — with three batches in flight the step was bound by that round trip.)
__global__ void batch_publish_kernel(const uint32_t* totals, const uint32_t* counters, uint32_t* host) {
    if (threadIdx.x == 0) {
        host[0] = totals[0]; host[1] = totals[1]; host[2] = counters[1]; host[3] = totals[3];
        __threadfence_system();
    }
}
__global__ void batch_clear_kernel(uint32_t* counters, uint32_t* totals, uint32_t* span_hist) {
    if (threadIdx.x < 16) counters[threadIdx.x] = 0;
    if (threadIdx.x < 4) totals[threadIdx.x] = 0;
    if (span_hist) for (int b = threadIdx.x; b < SPAN_BUCKETS; b += blockDim.x) span_hist[b] = 0;
}
}  // namespace wsa

extern "C" {

int wsa_abi_version(void) { return WSA_ABI_VERSION; }

void wsa_config_default(wsa_config* c) {          // ref @B2965 (output_level 4 there as well: "Segment Formants")
    c->spec_type = 1; c->output_level = 4; c->f_min = 50; c->f_max = 4000; c->N_fft_bins = 256; c->N_mel_bins = 128;
    c->window_width = 25; c->window_step = 25; c->pause_length = 200; c->min_seg_length = 50; c->auto_noise_gate = 1;
    c->voiced_max_dB = 100; c->voiced_min_dB = 10; c->pre_norm_gain = 1000; c->high_f_emph = 0;
}

const char* wsa_last_error(const wsa_ctx* ctx) { return ctx ? ctx->err.c_str() : wsa_api::g_create_error.c_str(); }

wsa_status wsa_create(const wsa_config* cfg, int32_t device, wsa_ctx** out) {
    if (!cfg || !out) return fail(nullptr, WSA_ERR_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, WSA_ERR_NO_DEVICE, "no HIP device visible (libwsa has no CPU path)");
    if (device < 0 || device >= n) return fail(nullptr, WSA_ERR_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, WSA_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, WSA_ERR_NO_DEVICE, std::string("libwsa is built for gfx950 only; device is ") + prop.gcnArchName);
    const int lv = cfg->output_level;
    if (!(lv == 1 || lv == 2 || lv == 3 || lv == 4 || lv == 5 || lv == 10 || lv == 11 || lv == 12 || lv == 13))
        return fail(nullptr, WSA_ERR_INVALID, "output_level must be 1, 2 (spectrum frames only), 3, 4, 5, 10, 11, 12 or 13");
    if (!(cfg->window_step > 0) || !(cfg->window_width > 0)) return fail(nullptr, WSA_ERR_INVALID, "window_width / window_step must be positive");
    wsa_ctx* c = new wsa_ctx();
    c->cfg = *cfg; c->device = device; c->n_cu = prop.multiProcessorCount;
    *out = c;
    return WSA_OK;
}

void wsa_destroy(wsa_ctx* ctx) { delete ctx; }

wsa_status wsa_geometry_for(const wsa_ctx* ctx, double fs, wsa_geometry* out) {
    if (!ctx || !out) return WSA_ERR_INVALID;
    FePlanHost p; std::string err;
    if (!build_fe_plan(ctx->cfg, fs, p, err)) return fail(const_cast<wsa_ctx*>(ctx), WSA_ERR_INVALID, err);
    out->nfft = p.nfft; out->win = p.win; out->hop = p.hop; out->bands = p.bands; out->kmax = p.kmax;
    return WSA_OK;
}

wsa_status wsa_bins_hz(const wsa_ctx* ctx, double fs, double* out, int32_t n) {
    if (!ctx || !out) return WSA_ERR_INVALID;
    FePlanHost p; std::string err;
    if (!build_fe_plan(ctx->cfg, fs, p, err)) return fail(const_cast<wsa_ctx*>(ctx), WSA_ERR_INVALID, err);
    if (n < p.bands) return fail(const_cast<wsa_ctx*>(ctx), WSA_ERR_INVALID, "bins_hz buffer too small");
    for (int i = 0; i < p.bands; i++) out[i] = p.bins_hz[i];
    return WSA_OK;
}

void wsa_batch_destroy(wsa_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    for (void* p : b->allocs) (void)hipFree(p);
    if (b->d_trk_stage) (void)hipFree(b->d_trk_stage);
    if (b->d_i16) (void)hipFree(b->d_i16);
    for (auto& st : b->up_stream) if (st) (void)hipStreamDestroy(st);
    for (auto& e : b->up_event) if (e) (void)hipEventDestroy(e);
    if (b->up_start) (void)hipEventDestroy(b->up_start);
    if (b->h_totals) (void)hipHostFree(b->h_totals);
    for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
    delete b;
}

static wsa_status batch_create_impl(wsa_ctx* ctx, uint32_t n_clips, const uint32_t* n_samples, double fs, const uint32_t* n_samples_in, double fs_in, wsa_batch** out) {
    if (!ctx || !out || (n_clips && !n_samples)) return fail(ctx, WSA_ERR_INVALID, "null argument");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    wsa_batch* b = new wsa_batch();
    b->ctx = ctx; b->n_clips = n_clips; b->fs = fs;
    b->tune = Tuning::from_env();
    std::string err;
    if (!build_fe_plan(ctx->cfg, fs, b->plan, err)) { delete b; return fail(ctx, WSA_ERR_INVALID, err); }
    const FePlanHost& P = b->plan;
    if (!fe_supported_R(P.R, P.three)) { delete b; return fail(ctx, WSA_ERR_INVALID, "unsupported FFT length: NFFT = the smallest of {2^k, 3 * 2^k} >= max(window, fs * N_fft_bins / f_max) must lie in 256 .. 8192 or be 3 * 2^k up to 12288"); }
    {   // the front-end kernel keeps its tables (window, twiddles, mel taps / power rows) in LDS: a geometry that needs more than a workgroup may have is refused here, not at the first launch
        const size_t need = fe_lds_required(P, b->tune.fe_fat);
        if (need > 160 * 1024) { delete b; return fail(ctx, WSA_ERR_INVALID, "this window / band setting needs " + std::to_string(need) + " bytes of LDS for the front end's tables (limit 163840): shorten the window or lower f_max / N_fft_bins"); }
    }
    b->n_samples.assign(n_samples, n_samples + n_clips);
    b->n_frames.resize(n_clips); b->frame_off.resize(n_clips + 1);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n_clips; i++) {
        const uint32_t ns = n_samples[i];
        const uint32_t nf = ns < (uint32_t)P.win ? 0u : (ns - (uint32_t)P.win) / (uint32_t)P.hop + 1u;     // FE-1 F1: tail dropped
        b->n_frames[i] = nf; b->frame_off[i] = (uint32_t)tot; tot += nf;
        if (nf > b->max_frames) b->max_frames = nf;
        if (ns > b->max_samples) b->max_samples = ns;
    }
    if (tot * CAND_CAP > 0xfffffff0ull) { delete b; return fail(ctx, WSA_ERR_INVALID, "batch has too many frames"); }     // candidate indices are 32-bit
    b->frame_off[n_clips] = (uint32_t)tot; b->total_frames = (uint32_t)tot;

    // capacity bounds (DESIGN.md "capacities"): nothing below can overflow for any input
    const wsa_config& c = ctx->cfg;
    const double breaker = c.pause_length > 2 * c.window_step ? c.pause_length / c.window_step : 250 / c.window_step;
    const double min_frames = std::trunc(c.min_seg_length / c.window_step);
    const int period = (int)min_frames + 1 + (int)std::floor(breaker);
    b->fcap = (int)b->max_frames + 2;
    b->seg_cap = (int)b->max_frames / (period > 0 ? period : 1) + 2;
    b->row_cap = (c.output_level == 10 || c.output_level == 11 || c.output_level == 12 || c.output_level == 13) ? (int)b->max_frames / 2 + 2 : b->seg_cap;
    b->tcap = ((P.bands + 1) / 2) * b->fcap;
    b->pcap = b->tcap;
    b->ws_stride = tracker_ws_bytes(b->tcap, b->pcap, b->fcap, c.output_level == 3);
    // two spans per wave (two work spaces each) wherever the paired tracker variant applies: its bit map of peak bins covers 128 bands,
    // level 3 and the per-frame trace keep the one-span kernel (WSA_NO_PAIR=1: test hook)
    b->pair = c.output_level != 3 && P.bands <= 128 && !b->tune.no_pair;
    // split finalize (the paired accumulate and the finalize as two kernels; WSA_NO_SPLIT=1: test hook for the one-kernel variant): the spans' tracks and points
    // live in regions of one pool (4.7 KB per frame of the batch) instead of per-wave work spaces
    b->pool_bpf = tracker_pool_bpf();
    b->split = b->pair && !b->tune.no_split && (size_t)b->total_frames * b->pool_bpf <= ((size_t)48 << 30);      // (a plan of more than ~10 M frames keeps the per-wave work spaces: bounded memory)
    const size_t budget = (size_t)(b->pair ? 16 : 8) << 30;
    size_t waves = budget / ((b->ws_stride ? b->ws_stride : 1) * (b->pair ? 2 : 1));
    size_t wpc = 16;                                          // tracker waves per CU = what the default variant's registers and LDS allow (tuning knob WSA_TRACKER_WPC)
    if (b->tune.tracker_wpc >= 1 && b->tune.tracker_wpc <= 32) wpc = (size_t)b->tune.tracker_wpc;
    const size_t want = (size_t)ctx->n_cu * wpc;
    if (waves > want) waves = want;
    if (waves > (size_t)n_clips * (size_t)b->seg_cap) waves = (size_t)n_clips * (size_t)b->seg_cap;
    if (b->pair && waves > ((size_t)n_clips * (size_t)b->seg_cap + 1) / 2 && n_clips * (size_t)b->seg_cap > 256) waves = ((size_t)n_clips * (size_t)b->seg_cap + 1) / 2;   // a wave takes two spans (the redo kernel runs on up to 1024 of them)
    if (waves < 1) waves = 1;
    b->n_waves = (int)waves;

    bool ok = true;
    ok = ok && dev_upload(b, &b->d_window, P.window) && dev_upload(b, &b->d_tw_n2, P.tw_n2) && dev_upload(b, &b->d_tw_m, P.tw_m) && dev_upload(b, &b->d_tw_64, P.tw_64)
            && dev_upload(b, &b->d_tw_nfft, P.tw_nfft) && dev_upload(b, &b->d_mel_k0, P.mel_k0) && dev_upload(b, &b->d_mel_cnt, P.mel_cnt)
            && dev_upload(b, &b->d_mel_off, P.mel_off) && dev_upload(b, &b->d_mel_w, P.mel_w) && dev_upload(b, &b->d_emph, P.emph)
            && dev_upload(b, &b->d_n_frames, b->n_frames) && dev_upload(b, &b->d_frame_off, b->frame_off);
    ok = ok && dev_alloc(b, &b->d_spec, (size_t)b->total_frames * P.bands);
    if (ok && c.output_level > 2 && b->split) {
        // split finalize: the span pool (4.7 KB per frame of the batch) and the span headers first — when the device cannot give them (several planned batches,
        // a cached plan, a smaller device), the plan falls back to the one-kernel paired tracker with its per-wave work spaces instead of failing
        size_t free_b = 0, total_b = 0;
        const size_t pool_bytes = (size_t)b->total_frames * b->pool_bpf;
        const bool room = hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b > pool_bytes + ((size_t)2 << 30);
        if (!(room && dev_alloc(b, &b->d_pool, pool_bytes) && dev_alloc(b, &b->d_span_hdr, (size_t)n_clips * b->seg_cap * 8))) {
            (void)hipGetLastError();
            b->split = false; b->d_pool = nullptr; b->d_span_hdr = nullptr;
        }
    }
    if (c.output_level > 2) {
        const size_t ncand = (size_t)b->total_frames * CAND_CAP;
        ok = ok && dev_alloc(b, &b->rec.hdr, (size_t)b->total_frames) && dev_alloc(b, &b->rec.amp, ncand) && dev_alloc(b, &b->rec.ent, ncand)
                && dev_alloc(b, &b->d_ws, b->ws_stride * (size_t)b->n_waves * (b->pair && !b->split ? 2 : 1)) && dev_alloc(b, &b->d_redo, (size_t)n_clips * b->seg_cap)
                && dev_alloc(b, &b->d_seg_i, (size_t)n_clips * b->seg_cap * 8) && dev_alloc(b, &b->d_seg_d, (size_t)n_clips * b->seg_cap * 2)
                && dev_alloc(b, &b->d_seg_count, (size_t)n_clips) && dev_alloc(b, &b->d_clip_rows, (size_t)n_clips) && dev_alloc(b, &b->d_order, (size_t)n_clips * b->seg_cap)
                && dev_alloc(b, &b->d_span_hist, (size_t)SPAN_BUCKETS) && dev_alloc(b, &b->d_span_key, (size_t)n_clips * b->seg_cap)
                && dev_alloc(b, &b->d_fr_info, (size_t)b->total_frames) && dev_alloc(b, &b->d_fr_v, (size_t)b->total_frames)
                && dev_alloc(b, &b->d_fr_fl, (size_t)b->total_frames)
                && dev_alloc(b, &b->d_meta_pool, (size_t)n_clips * b->row_cap * 8) && dev_alloc(b, &b->d_feat_pool, (size_t)n_clips * b->row_cap * WSA_NFEAT)
                && dev_alloc(b, &b->d_seg, (size_t)n_clips * b->seg_cap * 4) && dev_alloc(b, &b->d_meta, (size_t)n_clips * b->row_cap * 8)
                && dev_alloc(b, &b->d_feat, (size_t)n_clips * b->row_cap * WSA_NFEAT);
        if (c.output_level == 4 || c.output_level == 10 || c.output_level == 11 || c.output_level == 12) ok = ok && dev_alloc(b, &b->d_formants, (size_t)b->total_frames * 9);
        if (c.output_level == 3) ok = ok && dev_alloc(b, &b->d_trk_pts, ((size_t)b->total_frames + 1) * 64 * 2) && dev_alloc(b, &b->d_trk_rank, ((size_t)b->total_frames + 1) * 64)
                                         && dev_alloc(b, &b->d_trk_seg, (size_t)n_clips * b->seg_cap * 4);
        if (c.output_level == 12) ok = ok && dev_alloc(b, &b->d_sums, (size_t)b->total_frames) && dev_alloc(b, &b->d_coef_ws, (size_t)b->total_frames * 8);
        if (c.output_level == 11)
            ok = ok && dev_alloc(b, &b->d_utt_meta, (size_t)n_clips * b->seg_cap * 4) && dev_alloc(b, &b->d_utt_feat, (size_t)n_clips * b->seg_cap * WSA_NUTT)
                    && dev_alloc(b, &b->d_utt_off, (size_t)n_clips + 1);
    }
    ok = ok && dev_alloc(b, &b->d_counters, 16) && dev_alloc(b, &b->d_row_off, (size_t)n_clips + 1) && dev_alloc(b, &b->d_seg_off, (size_t)n_clips + 1)
            && dev_alloc(b, &b->d_totals, 4);
    if (ok) ok = hipHostMalloc(reinterpret_cast<void**>(&b->h_totals), 8 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess
                 && hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_totals_dev), b->h_totals, 0) == hipSuccess;
    if (ok) std::memset(b->h_totals, 0, 8 * sizeof(uint32_t));
    for (auto& e : b->ev) if (ok) ok = hipEventCreate(&e) == hipSuccess;
    if (b->tune.full_table >= 0) b->full_table = b->tune.full_table != 0;       // test hook: start with the worst-case tracker variant
    if (ok && n_samples_in) {              // K0 in front: the caller's PCM is at fs_in, everything planned above works on the converted clips
        b->rs_on = true; b->fs_in = fs_in; b->n_samples_in.assign(n_samples_in, n_samples_in + n_clips);
        for (uint32_t i = 0; i < n_clips; i++) if (n_samples_in[i] > b->max_samples_in) b->max_samples_in = n_samples_in[i];
        std::vector<float> K0, K; build_resample_table(fs_in, fs, K0); resample_table_image(K0, K);
        b->rs_stride = ((uint64_t)b->max_samples + 3u) & ~3ull;
        ok = dev_upload(b, &b->d_rs_table, K) && dev_upload(b, &b->d_rs_n_in, b->n_samples_in) && dev_upload(b, &b->d_rs_n_out, b->n_samples)
             && dev_alloc(b, &b->d_rs_pcm, (size_t)n_clips * b->rs_stride + 4);
    }
    if (!ok) {
        const std::string m = std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError());
        wsa_batch_destroy(b);
        return fail(ctx, WSA_ERR_HIP, m);
    }
    *out = b;
    return WSA_OK;
}

wsa_status wsa_batch_create(wsa_ctx* ctx, uint32_t n_clips, const uint32_t* n_samples, double fs, wsa_batch** out) {
    return batch_create_impl(ctx, n_clips, n_samples, fs, nullptr, 0, out);
}

uint64_t wsa_resample_length(uint64_t n_in, double fs_in, double fs_out) { return fs_in > 0 && fs_out > 0 ? resample_length(n_in, fs_in, fs_out) : 0; }

wsa_status wsa_batch_create_resampled(wsa_ctx* ctx, uint32_t n_clips, const uint32_t* n_samples_in, double fs_in, double fs_out, wsa_batch** out) {
    if (!ctx || !out || (n_clips && !n_samples_in)) return fail(ctx, WSA_ERR_INVALID, "null argument");
    if (!(fs_in > 0) || !(fs_out > 0) || fs_in / fs_out > 16 || fs_out / fs_in > 16) return fail(ctx, WSA_ERR_INVALID, "sample rates must be positive and at most a factor 16 apart");
    std::vector<uint32_t> n_out(n_clips);
    for (uint32_t i = 0; i < n_clips; i++) {
        const uint64_t n = resample_length(n_samples_in[i], fs_in, fs_out);
        if (n > 0xfffffff0ull) return fail(ctx, WSA_ERR_INVALID, "a converted clip would exceed 2^32 samples");
        n_out[i] = (uint32_t)n;
    }
    return batch_create_impl(ctx, n_clips, n_out.data(), fs_out, n_samples_in, fs_in, out);
}

wsa_status wsa_batch_get_info(const wsa_batch* b, wsa_batch_info* o) {
    if (!b || !o) return WSA_ERR_INVALID;
    o->n_clips = b->n_clips; o->n_frames_total = b->total_frames; o->max_frames_per_clip = b->max_frames; o->bands = (uint32_t)b->plan.bands;
    o->rows_cap = b->n_clips * (uint32_t)b->row_cap; o->segments_cap = b->n_clips * (uint32_t)b->seg_cap; o->workspace_bytes = b->dev_bytes;
    return WSA_OK;
}

static void fill_fe(const wsa_batch* b, const float* d_pcm, uint64_t stride, FeParams& p) {
    const FePlanHost& P = b->plan;
    p.pcm = d_pcm; p.clip_stride = stride; p.n_frames = b->d_n_frames; p.frame_off = b->d_frame_off; p.spec = b->d_spec;
    p.win = P.win; p.hop = P.hop; p.kmax = P.kmax; p.bands = P.bands; p.spec_type = P.spec_type; p.mel_total = (int)P.mel_w.size();
    p.mel_max_taps = 0; for (int32_t c_ : P.mel_cnt) if (c_ > p.mel_max_taps) p.mel_max_taps = c_;
    p.mel_max_taps_lo = 0; for (size_t i_ = 0; i_ < P.mel_cnt.size() && i_ < 64; i_++) if (P.mel_cnt[i_] > p.mel_max_taps_lo) p.mel_max_taps_lo = P.mel_cnt[i_];
    p.frames_per_wave = b->tune.fpw > 0 ? b->tune.fpw : 25; p.pcm_off = nullptr;
    p.fat = b->tune.fe_fat ? 1 : 0; p.wg_per_cu = b->tune.fe_wg_per_cu;
    p.queue = b->tune.fe_no_queue ? nullptr : b->d_counters + 8; p.chunks_per_clip = 0; p.n_chunks = 0; p.n_cu = b->ctx->n_cu;      // (counters[8]: the front end's chunk queue, zeroed by batch_clear_kernel)
    p.window = b->d_window; p.tw_n2 = b->d_tw_n2; p.tw_64 = b->d_tw_64; p.tw_nfft = b->d_tw_nfft; p.tw_m = b->d_tw_m;
    p.mel_k0 = b->d_mel_k0; p.mel_cnt = b->d_mel_cnt; p.mel_off = b->d_mel_off; p.mel_w = b->d_mel_w; p.emph = b->d_emph; p.gain = P.gain;
}

static wsa_status run_backend_stages(wsa_batch* b, const uint32_t* d_spec, bool skip_peaks, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    const wsa_config& c = ctx->cfg;
    b->published = false;
    if (c.output_level <= 2) return WSA_OK;
    const int dbg = b->tune.dbg;
    {
        hipStream_t cs = s;
        uint32_t* counters = b->d_counters + 4;             // [0] largest per-clip segment count
        uint32_t* shared = b->d_counters;                   // [1] flags
        PkParams pk; pk.spec = d_spec; pk.rec = b->rec; pk.frame0 = 0; pk.total_frames = b->total_frames; pk.bands = b->plan.bands;
        pk.stream_state = nullptr; pk.n_frames = nullptr; pk.step_frames = 0; pk.ring = 0; pk.flags = shared + 1; pk.dbg = (b->tune.dbg >> 20) & 0xff; pk.lanes_only = b->tune.peaks_lanes ? 1 : 0;      // (WSA_DBG bits 20 .. 27: the peak scan's what-if switches, TUNING=1 builds only)
        pk.wpc = b->tune.peaks_wpc; pk.round_bins = b->tune.peaks_w;
        if (!skip_peaks) launch_peaks(pk, cs);        // (a rerun of the back end finds the frame records in place)
        GateParams g;
        g.rec = b->rec; g.n_frames = b->d_n_frames; g.frame_off = b->d_frame_off; g.clip0 = 0; g.n_clips = b->n_clips;
        const int klevel = (c.output_level == 11 || c.output_level == 12) ? 10 : c.output_level;      // levels 11 / 12 store what level 10 stores (ref @B27713, @B27240)
        g.level = klevel;
        g.max_voiced_bin = (int)std::trunc(0.7 * b->plan.bands);                                   // ref @B25136
        g.breaker = c.pause_length > 2 * c.window_step ? c.pause_length / c.window_step : 250 / c.window_step;   // ref @B25188
        g.min_frames = std::trunc(c.min_seg_length / c.window_step);                               // ref @B25218
        g.auto_gate = c.auto_noise_gate ? 1 : 0;
        if (g.auto_gate) { g.ctx_max0 = 50; g.floor0 = 2; }                                        // ref @B25471
        else { g.ctx_max0 = std::pow(10.0, c.voiced_max_dB / 20); g.floor0 = std::pow(10.0, c.voiced_min_dB / 20); }
        g.fr_info = b->d_fr_info; g.fr_v = b->d_fr_v; g.fr_fl = b->d_fr_fl;
        g.seg_i = b->d_seg_i; g.seg_d = b->d_seg_d; g.seg_cap = b->seg_cap; g.seg_count = b->d_seg_count;
        g.clip_rows = b->d_clip_rows; g.counters = counters; g.shared = shared; g.trace = b->d_trace; g.dbg = dbg; g.strided = 1;
        const bool ordered = !(dbg & 8192);                         // WSA_DBG bit 8192: (clip, segment) enumeration instead of the length-sorted order
        g.span_hist = ordered ? b->d_span_hist : nullptr; g.span_key = b->d_span_key;
        g.state = nullptr; g.ctl = nullptr; g.ring = 0; g.step_frames = 0; g.prio = 1;
        launch_gate(g, cs);
        TrParams t;
        t.rec = b->rec; t.frame_off = b->d_frame_off; t.level = klevel;
        t.fr_info = b->d_fr_info; t.fr_v = b->d_fr_v; t.fr_fl = b->d_fr_fl;
        t.seg_i = b->d_seg_i; t.seg_d = b->d_seg_d; t.seg_cap = b->seg_cap; t.seg_count = b->d_seg_count; t.n_clips = b->n_clips; t.counters = counters; t.shared = shared;
        t.ws = b->d_ws; t.ws_stride = b->ws_stride; t.tcap = b->tcap; t.pcap = b->pcap; t.fcap = b->fcap;
        t.row_meta = b->d_meta_pool; t.row_feat = b->d_feat_pool; t.row_cap = (uint32_t)b->row_cap; t.clip_rows = b->d_clip_rows; t.trace = b->d_trace;
        t.dbg = dbg; t.ring_mask = 0xffffffffu; t.formants = b->d_formants; t.sums = b->d_sums; t.trk_pts = b->d_trk_pts; t.trk_rank = b->d_trk_rank; t.trk_seg = b->d_trk_seg;
        t.order = ordered ? b->d_order : nullptr; t.order_cnt = 1; t.redo = b->d_redo; t.redo_count = counters + 2;
        t.fin_waves = b->tune.fin_wpc >= 1 && b->tune.fin_wpc <= 32 ? ctx->n_cu * b->tune.fin_wpc : 0;
        // four spans per wave where the batch has spans enough to keep every wave slot busy that way (the 12 500-clip shard: -5 % per step); a 1024-clip batch
        // has ~6 000 spans = 1 500 such waves on 4 096 slots, and the longer waves cost more than the instructions they save (WSA_QUAD=1 / WSA_NO_QUAD=1 force it)
        t.quad = b->split && !b->tune.no_quad && (b->tune.quad || b->total_frames >= 1200000u) ? 1 : 0;
        t.quad_waves = (int)std::min<size_t>((size_t)b->n_waves, ((size_t)b->n_clips * (size_t)b->seg_cap + 3) / 4 + 1);
        t.pool = b->split ? b->d_pool : nullptr; t.pool_bpf = (uint32_t)b->pool_bpf; t.span_hdr = b->split ? b->d_span_hdr : nullptr;
        if (t.order) launch_span_order(t, b->d_span_hist, b->d_span_key, b->d_order, counters, cs);
        if (b->timing) HIP_TRY(ctx, hipEventRecord(b->ev[2], s));          // stage 1 = peak scan + gate + span order, stage 2 = tracker
        launch_tracker(t, b->n_waves, b->full_table, b->pair, cs);
    }
    if (b->timing) HIP_TRY(ctx, hipEventRecord(b->ev[3], s));
    CompactParams cp;
    cp.n_clips = b->n_clips; cp.seg_cap = b->seg_cap; cp.level = (c.output_level == 11 || c.output_level == 12) ? 10 : c.output_level;
    cp.seg_i = b->d_seg_i; cp.seg_count = b->d_seg_count; cp.row_meta_in = b->d_meta_pool; cp.row_feat_in = b->d_feat_pool;
    cp.seg_out = b->d_seg; cp.row_meta_out = b->d_meta; cp.row_feat_out = b->d_feat;
    cp.clip_row_off = b->d_row_off; cp.clip_seg_off = b->d_seg_off; cp.totals = b->d_totals; cp.carry = nullptr; cp.ctl = nullptr;
    // levels 11 / 12 run a kernel behind the compaction that adds to the result counters: they keep the separate publish at the end of the run
    cp.clip_rows = b->d_clip_rows; cp.flags = b->d_counters + 1; cp.fused = b->tune.no_fuse ? 0 : 1;
    cp.host = (c.output_level == 11 || c.output_level == 12) ? nullptr : b->h_totals_dev;
    b->published = compact_is_fused(cp) && cp.host != nullptr;
    // ... and, outside a stream capture, it leaves the counters cleared for the batch's next run (a captured run keeps its own clear kernel: a graph must not depend on what ran before it)
    const bool self_clear = b->published && !b->capturing;
    cp.clr_counters = self_clear ? b->d_counters : nullptr; cp.clr_hist = self_clear ? b->d_span_hist : nullptr;
    b->end_clears = self_clear;
    launch_compact(cp, s);
    if (c.output_level == 11) {
        UttParams u;
        u.n_clips = b->n_clips; u.segments = b->d_seg; u.row_meta = b->d_meta; u.clip_seg_off = b->d_seg_off; u.clip_row_off = b->d_row_off;
        u.frame_off = b->d_frame_off; u.formants = b->d_formants; u.clip_utt_off = b->d_utt_off; u.utt_meta = b->d_utt_meta; u.utt_feat = b->d_utt_feat;
        u.totals = b->d_totals; u.state = nullptr; u.carry = nullptr; u.ctl = nullptr; u.ring_mask = 0xffffffffu;
        launch_utterance(u, s);
    }
    if (c.output_level == 12) {
        CoefParams q;
        q.row_meta = b->d_meta; q.row_feat = b->d_feat; q.frame_off = b->d_frame_off; q.totals = b->d_totals; q.formants = b->d_formants; q.sums = b->d_sums;
        q.ws = b->d_coef_ws; q.total_frames = b->total_frames; q.shared = b->d_counters; q.ring_mask = 0xffffffffu; q.scratch_stride = 0;
        launch_coeffs(q, b->n_clips * (uint32_t)b->row_cap, s);
    }
    HIP_TRY(ctx, hipGetLastError());
    return WSA_OK;
}

static wsa_status run_impl(wsa_batch* b, const float* d_pcm, uint64_t stride, const uint32_t* d_spec_in, bool fe, bool be, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        b->capturing = cap != hipStreamCaptureStatusNone;
        if (b->capturing) b->ever_captured = true;
    }
    // (the previous run's last kernel has cleared the counters already when it was the fused compaction: b->counters_clean.  A batch that has ever been
    //  captured into a graph clears in front of every run: a replay is a run this function does not see)
    if (!b->counters_clean || b->ever_captured) hipLaunchKernelGGL(batch_clear_kernel, dim3(1), dim3(256), 0, s, b->d_counters, b->d_totals, b->d_span_hist);
    b->counters_clean = false; b->end_clears = false;
    if (b->timing) HIP_TRY(ctx, hipEventRecord(b->ev[0], s));
    const uint32_t* spec = d_spec_in ? d_spec_in : b->d_spec;
    if (fe) {
        if (!d_pcm && b->total_frames) return fail(ctx, WSA_ERR_INVALID, "null PCM pointer");
        if (stride < (b->rs_on ? b->max_samples_in : b->max_samples) && b->n_clips > 1) return fail(ctx, WSA_ERR_INVALID, "clip_stride smaller than the longest clip");
        if (b->rs_on) {                      // K0: the caller's PCM (fs_in) -> the batch's own buffer at the analysis rate
            RsParams r; r.in = d_pcm; r.stride_in = stride; r.out = b->d_rs_pcm; r.stride_out = b->rs_stride;
            r.n_in = b->d_rs_n_in; r.n_out = b->d_rs_n_out; r.table = b->d_rs_table; r.ratio = b->fs_in / b->fs; r.S = b->tune.rs_s > 0 ? b->tune.rs_s : resample_stride(b->fs_in, b->fs);
            r.J = b->tune.rs_j > 0 ? b->tune.rs_j : resample_outputs_per_lane(r.S, r.ratio); r.span = resample_span(r.ratio, r.S, r.J); r.chunks = b->tune.rs_c > 0 ? b->tune.rs_c : 1;       // (more runs per block never won: tools/resample_sweep.sh)
            launch_resample(r, b->n_clips, b->max_samples, s);
            HIP_TRY(ctx, hipGetLastError());
            d_pcm = b->d_rs_pcm; stride = b->rs_stride;
        }
        FeParams p; fill_fe(b, d_pcm, stride, p);
        launch_frontend(p, (int)b->n_clips, (int)b->max_frames, b->plan.R, b->plan.three, s);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (b->timing) HIP_TRY(ctx, hipEventRecord(b->ev[1], s));
    if (be) {
        const wsa_status st = run_backend_stages(b, spec, false, s);
        if (st != WSA_OK) return st;
    } else if (b->timing) { HIP_TRY(ctx, hipEventRecord(b->ev[2], s)); HIP_TRY(ctx, hipEventRecord(b->ev[3], s)); }
    if (!(be && b->published)) hipLaunchKernelGGL(batch_publish_kernel, dim3(1), dim3(64), 0, s, b->d_totals, b->d_counters, b->h_totals_dev);
    if (b->timing) HIP_TRY(ctx, hipEventRecord(b->ev[4], s));
    b->ran = true; b->spec_in_use = spec;
    b->counters_clean = be && b->end_clears && !b->capturing;
    return WSA_OK;
}

// tuning (not part of wsa.h): the batch's 16 device counters as the last run left them
int wsa_debug_batch_counters(wsa_batch* b, uint32_t* out16) {
    if (!b || !out16) return WSA_ERR_INVALID;
    return hipMemcpy(out16, b->d_counters, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? WSA_OK : WSA_ERR_HIP;
}

wsa_status wsa_batch_run(wsa_batch* b, const float* d_pcm, uint64_t clip_stride, void* stream) {
    if (!b) return WSA_ERR_INVALID;
    return run_impl(b, d_pcm, clip_stride, nullptr, true, true, reinterpret_cast<hipStream_t>(stream));
}
wsa_status wsa_batch_run_frontend(wsa_batch* b, const float* d_pcm, uint64_t clip_stride, void* stream) {
    if (!b) return WSA_ERR_INVALID;
    return run_impl(b, d_pcm, clip_stride, nullptr, true, false, reinterpret_cast<hipStream_t>(stream));
}
wsa_status wsa_batch_run_backend(wsa_batch* b, const uint32_t* d_spectra, void* stream) {
    if (!b || !d_spectra) return WSA_ERR_INVALID;
    return run_impl(b, nullptr, 0, d_spectra, false, true, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
// The page-locked slabs wsa_host_alloc handed out (base -> bytes): upload_clips merges copies only INSIDE one of them — two allocations that happen to be
// adjacent in the address space are still two registrations, and one copy across their seam is not something the runtime has to accept.
static std::mutex g_slab_mu;
static std::map<uintptr_t, size_t> g_slabs;
static bool slab_of(const void* p, uintptr_t* base, size_t* size) {
    std::lock_guard<std::mutex> lk(g_slab_mu);
    auto it = g_slabs.upper_bound(reinterpret_cast<uintptr_t>(p));
    if (it == g_slabs.begin()) return false;
    --it;
    if (reinterpret_cast<uintptr_t>(p) >= it->first + it->second) return false;
    *base = it->first; *size = it->second;
    return true;
}
// Host -> device copies of many clips.  A copy out of pageable memory is staged by the runtime on the calling thread (~16 GB/s here),
// so large uploads are spread over a few threads (3: 2 .. 4 measure alike, 6 and 8 lose a quarter; WSA_UPLOAD_THREADS) with a stream each; `s` then waits for all of them.  dst(i) / src(i) / bytes(i) per clip.
template <typename DST, typename SRC, typename LEN>
static wsa_status upload_clips(wsa_batch* b, uint32_t n, DST dst, SRC src, LEN bytes, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) total += bytes(i);
    constexpr int UP_MAX = 8;
    int UP_THREADS = 3;
    if (b->tune.upload_threads >= 1 && b->tune.upload_threads <= UP_MAX) UP_THREADS = b->tune.upload_threads;      // tuning knob; 1 = copies on the calling thread
    // Page-locked sources (wsa_host_alloc, or memory the caller registered): the copy is a DMA out of the caller's buffer, no staging thread is needed, and
    // what costs is the number of copies (~10 us of submission each: 1024 clips = 10 ms) — clips that lie back to back in host AND device memory (views into
    // one pinned slab, lengths that keep the device layout's alignment) travel as ONE copy.
    // Every clip is classified (a slab of wsa_host_alloc: a map lookup; anything else: one attribute query): only when ALL of them are page-locked is this
    // path taken — a batch with pageable clips in the middle keeps the worker threads below, which copy page-locked clips just as well.
    {
        bool pinned = n > 0;
        std::vector<uintptr_t> slab_end(n, 0);          // end of the wsa_host_alloc slab clip i lies in (0: page-locked by other means — never merged)
        for (uint32_t i = 0; i < n && pinned; i++) {
            if (!bytes(i)) continue;
            uintptr_t base = 0; size_t size = 0;
            if (slab_of(src(i), &base, &size)) { slab_end[i] = base + size; continue; }
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, src(i)) != hipSuccess) { (void)hipGetLastError(); pinned = false; }
            else pinned = at.type == hipMemoryTypeHost;
        }
        if (pinned) {
            uint32_t i = 0;
            while (i < n) {
                const char* s0 = reinterpret_cast<const char*>(src(i)); char* d0 = reinterpret_cast<char*>(dst(i));
                size_t len = bytes(i); uint32_t j = i + 1;
                // a run grows while the next clip follows back to back on both sides AND still ends inside the slab the run started in
                while (slab_end[i] && j < n && reinterpret_cast<const char*>(src(j)) == s0 + len && reinterpret_cast<char*>(dst(j)) == d0 + len
                       && reinterpret_cast<uintptr_t>(s0) + len + bytes(j) <= slab_end[i]) { len += bytes(j); j++; }
                if (len) HIP_TRY(ctx, hipMemcpyAsync(d0, s0, len, hipMemcpyHostToDevice, s));
                i = j;
            }
            return WSA_OK;
        }
    }
    if (n < 16 || total < ((uint64_t)32 << 20) || UP_THREADS == 1) {
        for (uint32_t i = 0; i < n; i++) if (bytes(i)) HIP_TRY(ctx, hipMemcpyAsync(dst(i), src(i), bytes(i), hipMemcpyHostToDevice, s));
        return WSA_OK;
    }
    if (!b->up_ready) {
        for (int t = 0; t < UP_MAX; t++) { HIP_TRY(ctx, hipStreamCreateWithFlags(&b->up_stream[t], hipStreamNonBlocking)); HIP_TRY(ctx, hipEventCreateWithFlags(&b->up_event[t], hipEventDisableTiming)); }
        HIP_TRY(ctx, hipEventCreateWithFlags(&b->up_start, hipEventDisableTiming));
        b->up_ready = true;
    }
    // the upload streams start behind what is already queued on `s` (an earlier run may still read the buffers)
    HIP_TRY(ctx, hipEventRecord(b->up_start, s));
    hipError_t err[UP_MAX];
    std::thread th[UP_MAX];
    // contiguous ranges of equal byte counts
    uint32_t first_[UP_MAX + 1]; first_[0] = 0;
    { uint64_t acc = 0; int t = 1; for (uint32_t i = 0; i < n && t < UP_THREADS; i++) { acc += bytes(i); if (acc >= total * t / UP_THREADS) first_[t++] = i + 1; } while (t <= UP_THREADS) first_[t++] = n; }
    first_[UP_THREADS] = n;
    int started = 0;
    for (int t = 0; t < UP_THREADS; t++) {
        err[t] = hipSuccess;
        try {
        th[t] = std::thread([&, t]() {
            hipError_t e = hipSetDevice(ctx->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(b->up_stream[t], b->up_start, 0);
            for (uint32_t i = first_[t]; i < first_[t + 1] && e == hipSuccess; i++) if (bytes(i)) e = hipMemcpyAsync(dst(i), src(i), bytes(i), hipMemcpyHostToDevice, b->up_stream[t]);
            if (e == hipSuccess) e = hipEventRecord(b->up_event[t], b->up_stream[t]);
            err[t] = e;
        });
        } catch (const std::system_error&) { break; }         // no thread to be had: the calling thread takes what is left below
        started++;
    }
    for (int t = 0; t < started; t++) th[t].join();
    hipError_t first = hipSuccess;
    for (int t = started; t < UP_THREADS && first == hipSuccess; t++)       // (ranges whose thread could not be started)
        for (uint32_t i = first_[t]; i < first_[t + 1] && first == hipSuccess; i++) if (bytes(i)) first = hipMemcpyAsync(dst(i), src(i), bytes(i), hipMemcpyHostToDevice, s);
    // `s` waits for every worker's copies, also when one of them failed: none of them may still be writing the batch's buffers when the caller sees the error
    for (int t = 0; t < started; t++) {
        if (err[t] != hipSuccess) { if (first == hipSuccess) first = err[t]; (void)hipStreamSynchronize(b->up_stream[t]); }
        else { const hipError_t e = hipStreamWaitEvent(s, b->up_event[t], 0); if (e != hipSuccess && first == hipSuccess) first = e; }
    }
    HIP_TRY(ctx, first);
    return WSA_OK;
}

extern "C" {
wsa_status wsa_batch_run_host(wsa_batch* b, const float* const* pcm, void* stream) {
    if (!b || (!pcm && b->n_clips)) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const std::vector<uint32_t>& ns_host = b->rs_on ? b->n_samples_in : b->n_samples;      // what the caller holds: clips at the input rate
    const uint64_t stride = ((b->rs_on ? b->max_samples_in : b->max_samples) + 3u) & ~3ull;
    if (!b->d_pcm_own && !dev_alloc(b, &b->d_pcm_own, (size_t)b->n_clips * stride)) return fail(ctx, WSA_ERR_HIP, "PCM staging allocation failed");
    {
        const wsa_status st = upload_clips(b, b->n_clips, [&](uint32_t i) { return b->d_pcm_own + (size_t)i * stride; }, [&](uint32_t i) { return pcm[i]; },
                                           [&](uint32_t i) { return (size_t)ns_host[i] * sizeof(float); }, s);
        if (st != WSA_OK) return st;
    }
    return run_impl(b, b->d_pcm_own, stride, nullptr, true, true, s);
}

wsa_status wsa_batch_run_host_i16(wsa_batch* b, const int16_t* const* pcm, const uint32_t* channels, void* stream) {
    if (!b || (!pcm && b->n_clips)) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const std::vector<uint32_t>& ns_host = b->rs_on ? b->n_samples_in : b->n_samples;
    const uint64_t stride = ((b->rs_on ? b->max_samples_in : b->max_samples) + 3u) & ~3ull;
    if (!b->d_pcm_own && !dev_alloc(b, &b->d_pcm_own, (size_t)b->n_clips * stride)) return fail(ctx, WSA_ERR_HIP, "PCM staging allocation failed");
    // clip offsets inside one int16 upload buffer (8-byte aligned starts), grown when a run needs more than the last one
    std::vector<uint64_t> off(b->n_clips + 1, 0); std::vector<uint32_t> ch(b->n_clips ? b->n_clips : 1, 1u);
    for (uint32_t i = 0; i < b->n_clips; i++) {
        ch[i] = channels ? channels[i] : 1u;
        if (ch[i] < 1 || ch[i] > 64) return fail(ctx, WSA_ERR_INVALID, "channels must be 1 .. 64");
        off[i + 1] = off[i] + (((uint64_t)ns_host[i] * ch[i] + 3u) & ~3ull);
    }
    if (off[b->n_clips] > b->i16_cap) {
        if (b->d_i16) { (void)hipFree(b->d_i16); b->d_i16 = nullptr; b->i16_cap = 0; }
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&b->d_i16), (size_t)off[b->n_clips] * sizeof(int16_t) + 16));
        b->i16_cap = off[b->n_clips];
    }
    if (!b->d_i16_off) {
        if (!dev_alloc(b, &b->d_i16_off, (size_t)b->n_clips + 1) || !dev_alloc(b, &b->d_i16_ch, (size_t)b->n_clips + 1) || !dev_alloc(b, &b->d_i16_ns, (size_t)b->n_clips + 1))
            return fail(ctx, WSA_ERR_HIP, "int16 staging allocation failed");
    }
    // (the three small tables travel from buffers the batch owns: the copies are asynchronous)
    b->h_i16_off = off; b->h_i16_ch = ch;
    HIP_TRY(ctx, hipMemcpyAsync(b->d_i16_off, b->h_i16_off.data(), ((size_t)b->n_clips + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(b->d_i16_ch, b->h_i16_ch.data(), (size_t)(b->n_clips ? b->n_clips : 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(b->d_i16_ns, ns_host.data(), (size_t)b->n_clips * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    {
        const wsa_status st = upload_clips(b, b->n_clips, [&](uint32_t i) { return b->d_i16 + off[i]; }, [&](uint32_t i) { return pcm[i]; },
                                           [&](uint32_t i) { return (size_t)ns_host[i] * ch[i] * sizeof(int16_t); }, s);
        if (st != WSA_OK) return st;
    }
    if (b->n_clips) {
        const uint32_t mx = b->rs_on ? b->max_samples_in : b->max_samples;
        hipLaunchKernelGGL(pcm_i16_to_f32_kernel, dim3((mx + 1023) / 1024 ? (mx + 1023) / 1024 : 1, b->n_clips), dim3(256), 0, s, b->d_i16, b->d_i16_off, b->d_i16_ch, b->d_i16_ns, b->d_pcm_own, stride);
        HIP_TRY(ctx, hipGetLastError());
    }
    return run_impl(b, b->d_pcm_own, stride, nullptr, true, true, s);
}

static wsa_status fetch_totals(wsa_batch* b, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    if (!b->ran) return fail(ctx, WSA_ERR_INVALID, "no run on this batch yet");
    {   // always re-read: a captured graph may have re-run the batch without wsa_batch_run being called again
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(s));             // the run's last kernel has written the counters into the mapped pinned words
        const volatile uint32_t* ht = b->h_totals;
        b->res_rows = ht[0]; b->res_segs = ht[1]; b->res_flags = ht[2]; b->res_utt = ht[3];
        if (b->res_flags & 2u) {
            // the fast tracker variant ran out of LDS active-track slots: rerun the back end (frame
            // records are still in place) with the worst-case table, for this and all later runs.
            // Unconditional on full_table: a hipGraph captured before the switch keeps replaying the fast
            // variant and may overflow again (wsa.h: re-capture after wsa_batch_backend_reruns() changed).
            b->full_table = true; b->reruns++;
            hipLaunchKernelGGL(batch_clear_kernel, dim3(1), dim3(256), 0, s, b->d_counters, b->d_totals, b->d_span_hist);
            const bool tm = b->timing; b->timing = false;
            const wsa_status st = run_backend_stages(b, b->spec_in_use, true, s);
            b->timing = tm;
            if (st != WSA_OK) return st;
            if (!b->published) hipLaunchKernelGGL(batch_publish_kernel, dim3(1), dim3(64), 0, s, b->d_totals, b->d_counters, b->h_totals_dev);
            HIP_TRY(ctx, hipStreamSynchronize(s));
            b->res_rows = ht[0]; b->res_segs = ht[1]; b->res_flags = ht[2]; b->res_utt = ht[3];
        }
    }
    if (b->res_flags & 1u) return fail(ctx, WSA_ERR_CAPACITY, "a device-side arena overflowed; results are invalid");
    return WSA_OK;
}

wsa_status wsa_host_alloc(wsa_ctx* ctx, uint64_t bytes, void** out) {
    if (!ctx || !out) return WSA_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, bytes ? (size_t)bytes : 1, hipHostMallocPortable));
    { std::lock_guard<std::mutex> lk(g_slab_mu); g_slabs[reinterpret_cast<uintptr_t>(*out)] = bytes ? (size_t)bytes : 1; }
    return WSA_OK;
}
void wsa_host_free(void* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(g_slab_mu); g_slabs.erase(reinterpret_cast<uintptr_t>(p)); }
    (void)hipHostFree(p);
}

wsa_status wsa_queue_create(wsa_ctx* ctx, void** stream) {
    if (!ctx || !stream) return WSA_ERR_INVALID;
    *stream = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return WSA_OK;
}
void wsa_queue_destroy(wsa_ctx* ctx, void* stream) {
    if (!stream) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
}

// (gather.cpp checks that a rank's batch belongs to the rank's context)
wsa_ctx* wsa_batch_ctx_internal(const wsa_batch* b) { return b ? b->ctx : nullptr; }

wsa_status wsa_batch_result(wsa_batch* b, void* stream, wsa_device_result* o) {
    if (!b || !o) return WSA_ERR_INVALID;
    const wsa_status st = fetch_totals(b, reinterpret_cast<hipStream_t>(stream));
    o->n_clips = b->n_clips; o->n_rows = b->res_rows; o->n_segments = b->res_segs; o->n_frames_total = b->total_frames;
    o->status_flags = b->res_flags;
    o->d_row_meta = b->d_meta; o->d_row_feat = b->d_feat; o->d_segments = b->d_seg; o->d_clip_row_off = b->d_row_off; o->d_clip_seg_off = b->d_seg_off;
    o->d_spectra = b->spec_in_use; o->d_clip_frame_off = b->d_frame_off; o->d_formants = b->d_formants;
    o->n_utterance_rows = b->d_utt_feat ? b->res_utt : 0; o->d_utt_meta = b->d_utt_meta; o->d_utt_feat = b->d_utt_feat; o->d_clip_utt_off = b->d_utt_off;
    return st;
}

wsa_status wsa_batch_copy_rows(wsa_batch* b, void* stream, int32_t* row_meta, double* row_feat, uint32_t rows_cap,
                               int32_t* segments, uint32_t seg_cap, uint32_t* clip_row_off, uint32_t* clip_seg_off) {
    if (!b) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const wsa_status st = fetch_totals(b, s);
    if (st != WSA_OK) return st;
    if ((row_meta || row_feat) && rows_cap < b->res_rows) return fail(ctx, WSA_ERR_INVALID, "row buffer too small");
    if (segments && seg_cap < b->res_segs) return fail(ctx, WSA_ERR_INVALID, "segment buffer too small");
    if (row_meta && b->res_rows) HIP_TRY(ctx, hipMemcpyAsync(row_meta, b->d_meta, (size_t)b->res_rows * 8 * sizeof(int32_t), hipMemcpyDefault, s));
    if (row_feat && b->res_rows) HIP_TRY(ctx, hipMemcpyAsync(row_feat, b->d_feat, (size_t)b->res_rows * WSA_NFEAT * sizeof(double), hipMemcpyDefault, s));
    if (segments && b->res_segs) HIP_TRY(ctx, hipMemcpyAsync(segments, b->d_seg, (size_t)b->res_segs * 4 * sizeof(int32_t), hipMemcpyDefault, s));
    if (clip_row_off) HIP_TRY(ctx, hipMemcpyAsync(clip_row_off, b->d_row_off, ((size_t)b->n_clips + 1) * sizeof(uint32_t), hipMemcpyDefault, s));
    if (clip_seg_off) HIP_TRY(ctx, hipMemcpyAsync(clip_seg_off, b->d_seg_off, ((size_t)b->n_clips + 1) * sizeof(uint32_t), hipMemcpyDefault, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_copy_spectra(wsa_batch* b, void* stream, uint32_t* spectra, uint64_t cap_words, uint32_t* clip_frame_off) {
    if (!b) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!b->ran) return fail(ctx, WSA_ERR_INVALID, "no run on this batch yet");
    if (spectra && !b->spec_in_use) return fail(ctx, WSA_ERR_INVALID, "the u32 frames of this run were not stored: call wsa_batch_keep_spectra(batch, 1) before the run");
    const uint64_t words = (uint64_t)b->total_frames * (uint32_t)b->plan.bands;
    if (spectra && cap_words < words) return fail(ctx, WSA_ERR_INVALID, "spectra buffer too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (spectra && words) HIP_TRY(ctx, hipMemcpyAsync(spectra, b->spec_in_use, words * sizeof(uint32_t), hipMemcpyDefault, s));
    if (clip_frame_off) std::memcpy(clip_frame_off, b->frame_off.data(), ((size_t)b->n_clips + 1) * sizeof(uint32_t));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_copy_utterance(wsa_batch* b, void* stream, int32_t* utt_meta, double* utt_feat, uint32_t cap_rows, uint32_t* clip_utt_off) {
    if (!b) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!b->d_utt_feat) return fail(ctx, WSA_ERR_INVALID, "no utterance features: output_level must be 11");
    const wsa_status st = fetch_totals(b, s);
    if (st != WSA_OK) return st;
    if ((utt_meta || utt_feat) && cap_rows < b->res_utt) return fail(ctx, WSA_ERR_INVALID, "utterance buffer too small");
    if (utt_meta && b->res_utt) HIP_TRY(ctx, hipMemcpyAsync(utt_meta, b->d_utt_meta, (size_t)b->res_utt * 4 * sizeof(int32_t), hipMemcpyDefault, s));
    if (utt_feat && b->res_utt) HIP_TRY(ctx, hipMemcpyAsync(utt_feat, b->d_utt_feat, (size_t)b->res_utt * WSA_NUTT * sizeof(double), hipMemcpyDefault, s));
    if (clip_utt_off) HIP_TRY(ctx, hipMemcpyAsync(clip_utt_off, b->d_utt_off, ((size_t)b->n_clips + 1) * sizeof(uint32_t), hipMemcpyDefault, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_copy_formants(wsa_batch* b, void* stream, float* formants, uint64_t cap_frames) {
    if (!b || !formants) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!b->ran || !b->d_formants) return fail(ctx, WSA_ERR_INVALID, "no formant frames: output_level must be 4, 10, 11 or 12 and the batch must have run");
    if (cap_frames < b->total_frames) return fail(ctx, WSA_ERR_INVALID, "formant buffer too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (b->total_frames) HIP_TRY(ctx, hipMemcpyAsync(formants, b->d_formants, (size_t)b->total_frames * 9 * sizeof(float), hipMemcpyDefault, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}


// ---- level 3: the ranked raw tracks of every segment (ref @B28273 `s.push(i)`, dispatched at @B30132)
static wsa_status fetch_track_tables(wsa_batch* b, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    if (!b->d_trk_seg) return fail(ctx, WSA_ERR_INVALID, "no raw tracks: output_level must be 3");
    const wsa_status st = fetch_totals(b, s);
    if (st != WSA_OK) return st;
    b->h_trk_seg.resize((size_t)b->n_clips * b->seg_cap * 4); b->h_seg_count.resize(b->n_clips);
    if (b->n_clips) {
        HIP_TRY(ctx, hipMemcpyAsync(b->h_trk_seg.data(), b->d_trk_seg, b->h_trk_seg.size() * sizeof(int32_t), hipMemcpyDefault, s));
        HIP_TRY(ctx, hipMemcpyAsync(b->h_seg_count.data(), b->d_seg_count, b->h_seg_count.size() * sizeof(uint32_t), hipMemcpyDefault, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    return WSA_OK;
}

wsa_status wsa_batch_tracks_info(wsa_batch* b, void* stream, wsa_tracks_info* out) {
    if (!b || !out) return WSA_ERR_INVALID;
    const wsa_status st = fetch_track_tables(b, reinterpret_cast<hipStream_t>(stream));
    if (st != WSA_OK) return st;
    uint64_t np = 0, nr = 0; uint32_t ns = 0;
    for (uint32_t c = 0; c < b->n_clips; c++)
        for (uint32_t k = 0; k < b->h_seg_count[c]; k++) { const int32_t* t = &b->h_trk_seg[((size_t)c * b->seg_cap + k) * 4]; np += (uint32_t)t[1]; nr += (uint32_t)t[2]; ns++; }
    out->n_segments = ns; out->n_points = np; out->n_ranked = nr;
    return WSA_OK;
}

wsa_status wsa_batch_copy_tracks(wsa_batch* b, void* stream, uint64_t* seg_off, int32_t* points, uint64_t cap_points, int32_t* ranked, uint64_t cap_ranked) {
    if (!b || !seg_off || !points || !ranked) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const wsa_status st = fetch_track_tables(b, s);
    if (st != WSA_OK) return st;
    // one gather kernel into a staging buffer and two copies (a copy per segment was ~12 000 small copies for a 1024-clip batch)
    uint64_t np = 0, nr = 0; uint32_t ns = 0;
    b->h_trk_desc.clear();
    for (uint32_t c = 0; c < b->n_clips; c++)
        for (uint32_t k = 0; k < b->h_seg_count[c]; k++) {
            const int32_t* t = &b->h_trk_seg[((size_t)c * b->seg_cap + k) * 4];
            const uint64_t pool0 = (uint64_t)(uint32_t)t[0] | ((uint64_t)(uint32_t)t[3] << 32);
            const uint32_t n_pt = (uint32_t)t[1], nq = (uint32_t)t[2];
            if (np + n_pt > cap_points || nr + nq > cap_ranked) return fail(ctx, WSA_ERR_INVALID, "track buffers too small");
            seg_off[2 * ns] = np; seg_off[2 * ns + 1] = nr;
            const uint64_t d[6] = {pool0, 0ull, n_pt, nq, np, nr};
            b->h_trk_desc.insert(b->h_trk_desc.end(), d, d + 6);
            np += n_pt; nr += nq; ns++;
        }
    seg_off[2 * ns] = np; seg_off[2 * ns + 1] = nr;
    if (np + nr) {
        const size_t o_pts = ((size_t)ns * 6 * sizeof(uint64_t) + 255) & ~(size_t)255, o_rank = o_pts + (size_t)np * 8 * sizeof(int32_t), need = o_rank + (size_t)nr * sizeof(int32_t);
        if (need > b->trk_stage_cap) {
            if (b->d_trk_stage) { (void)hipFree(b->d_trk_stage); b->d_trk_stage = nullptr; b->trk_stage_cap = 0; }
            if (hipMalloc(reinterpret_cast<void**>(&b->d_trk_stage), need + need / 4) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, WSA_ERR_HIP, "no device memory for the raw-track staging buffer"); }
            b->trk_stage_cap = need + need / 4;
        }
        HIP_TRY(ctx, hipMemcpyAsync(b->d_trk_stage, b->h_trk_desc.data(), (size_t)ns * 6 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        launch_gather_tracks(reinterpret_cast<const uint64_t*>(b->d_trk_stage), ns, 1ull << 63, b->d_trk_pts, b->d_trk_rank,
                             reinterpret_cast<int4*>(b->d_trk_stage + o_pts), reinterpret_cast<int32_t*>(b->d_trk_stage + o_rank), s);
        HIP_TRY(ctx, hipGetLastError());
        if (np) HIP_TRY(ctx, hipMemcpyAsync(points, b->d_trk_stage + o_pts, (size_t)np * 8 * sizeof(int32_t), hipMemcpyDefault, s));
        if (nr) HIP_TRY(ctx, hipMemcpyAsync(ranked, b->d_trk_stage + o_rank, (size_t)nr * sizeof(int32_t), hipMemcpyDefault, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_copy_pcm(wsa_batch* b, void* stream, float* pcm, uint64_t stride) {
    if (!b || !pcm) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!b->rs_on || !b->ran) return fail(ctx, WSA_ERR_INVALID, "no converted PCM: the batch must come from wsa_batch_create_resampled and must have run");
    if (stride < b->max_samples) return fail(ctx, WSA_ERR_INVALID, "stride smaller than the longest converted clip");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < b->n_clips; i++)
        if (b->n_samples[i]) HIP_TRY(ctx, hipMemcpyAsync(pcm + (size_t)i * stride, b->d_rs_pcm + (size_t)i * b->rs_stride, (size_t)b->n_samples[i] * sizeof(float), hipMemcpyDefault, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_stage_ms(wsa_batch* b, float out[4]) {
    if (!b || !out) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    if (!b->ran || !b->timing) return fail(ctx, WSA_ERR_INVALID, "no timed run on this batch");
    HIP_TRY(ctx, hipEventSynchronize(b->ev[4]));
    for (int i = 0; i < 4; i++) HIP_TRY(ctx, hipEventElapsedTime(&out[i], b->ev[i], b->ev[i + 1]));
    return WSA_OK;
}

wsa_status wsa_batch_enable_trace(wsa_batch* b, int32_t on) {
    if (!b) return WSA_ERR_INVALID;
    if (on && !b->d_trace) {
        HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
        const size_t rows = b->total_frames ? b->total_frames : 1;      // an empty batch still gets a valid (unused) buffer
        if (!dev_alloc(b, &b->d_trace, rows * 12)) return fail(b->ctx, WSA_ERR_HIP, "trace allocation failed");
        HIP_TRY(b->ctx, hipMemset(b->d_trace, 0, rows * 12 * sizeof(double)));
    }
    if (!on) b->d_trace = nullptr;       // the allocation stays owned by the batch
    return WSA_OK;
}

wsa_status wsa_batch_copy_trace(wsa_batch* b, void* stream, double* out, uint64_t cap_rows) {
    if (!b || !out) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!b->d_trace || !b->ran) return fail(ctx, WSA_ERR_INVALID, "trace not enabled or no run yet");
    if (cap_rows < b->total_frames) return fail(ctx, WSA_ERR_INVALID, "trace buffer too small");
    if (b->total_frames) HIP_TRY(ctx, hipMemcpyAsync(out, b->d_trace, (size_t)b->total_frames * 12 * sizeof(double), hipMemcpyDefault, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return WSA_OK;
}

wsa_status wsa_batch_keep_spectra(wsa_batch* b, int32_t on) {
    if (!b) return WSA_ERR_INVALID;
    (void)on;                                          // the front end hands the u32 frames to the peak scan through the array: they are always there
    return WSA_OK;
}

wsa_status wsa_batch_backend_reruns(const wsa_batch* b, uint32_t* out) {
    if (!b || !out) return WSA_ERR_INVALID;
    *out = b->reruns;
    return WSA_OK;
}

wsa_status wsa_batch_enable_timing(wsa_batch* b, int32_t on) {
    if (!b) return WSA_ERR_INVALID;
    b->timing = on != 0;
    return WSA_OK;
}

}  // extern "C"

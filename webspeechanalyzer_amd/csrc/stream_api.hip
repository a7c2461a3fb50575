// stream_api.hip — the wsa_stream_* part of include/wsa.h: n_streams concurrent launches that advance
// in lock step, one hipGraph launch per step.
//
// Stands in for the reference's online path (ref dist/main.js:2): worklet process() -> port message ->
// spectrum_push (@B8752, @B30392) once per frame with module-level state, callbacks as segments close
// (@B28869), StopAudioNodes -> segment_truncate (@B5699, @B30757).  The kernels are the batch ones:
//   front end   frontend.hip on the step's samples (n_frames = frames of this step per stream)
//   peaks       peaks.hip, records written into per-stream rings (slot = absolute frame & (ring - 1))
//   gate        gate.hip gate_kernel_t<true>: state in HBM between steps, segments of this step only
//   tracker     tracker.hip over the spans that closed in this step (ring-indexed frames)
//   compaction  tracker.hip, callback index / segments_ci history carried in HBM
// A span (frames between two segmenter resets) stays in its stream's ring until it closes, so results
// are those of one clip holding the whole signal; tests/test_gpu_stream.py checks exactly that.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "api_internal.hpp"

using namespace wsa;
using wsa_api::fail;

struct wsa_stream {
    wsa_ctx* ctx = nullptr;
    uint32_t n = 0, F = 0, ring = 0;
    std::vector<uint32_t> seen_cuts;        // stream_cuts as of the previous collect (WSA_FLAG_STREAM_CUT)
    double fs = 0;
    FePlanHost plan;
    Tuning tune;                            // test switches, read from the environment when the stream set is planned
    uint32_t hist = 0, q = 1, step_samples = 0, stage_stride = 0;     // hist = (q - 1) * hop samples of history, q = ceil(win / hop)
    int seg_cap = 0, row_cap = 0, tcap = 0, pcap = 0, fcap = 0, n_waves = 0;
    size_t ws_stride = 0;
    std::vector<void*> allocs;
    float *d_window = nullptr, *d_mel_w = nullptr, *d_emph = nullptr, *d_stage = nullptr, *d_pcm_in = nullptr;
    float2 *d_tw_n2 = nullptr, *d_tw_64 = nullptr, *d_tw_nfft = nullptr, *d_tw_m = nullptr;
    int32_t *d_mel_k0 = nullptr, *d_mel_cnt = nullptr, *d_mel_off = nullptr;
    uint32_t *d_ctl = nullptr;              // [3][n]: n_frames, pcm_off, ctl bits
    uint32_t *d_frame_off = nullptr, *d_ring_off = nullptr, *d_spec = nullptr;
    RecPtrs rec = {nullptr, nullptr, nullptr};      // frame records of the ring slots
    int4* d_trk_pts = nullptr; int32_t *d_trk_rank = nullptr, *d_trk_seg = nullptr;      // level 3: raw-track pools (per stream a ring of ring x 64 entries), per segment {pool offset lo, points, ranked, offset hi}
    std::vector<uint64_t> x_trk_off; std::vector<int32_t> x_trk_pts, x_trk_rank, x_trk_seg;
    uint32_t *d_utt_state = nullptr, *d_utt_off = nullptr; int32_t* d_utt_meta = nullptr; double* d_utt_feat = nullptr;      // level 11: per-stream histogram state, this step's results
    std::vector<int32_t> x_utt_meta; std::vector<double> x_utt_feat;
    float* d_sums = nullptr; double* d_coef_ws = nullptr;      // level 12: per-frame energy sums of straighten (ring), scratch of the four fits per syllable
    float* d_formants = nullptr;            // levels 4 / 10 / 12: straightened frames of the segments, per stream a ring [ring][9] indexed like the frame records
    std::vector<float> x_formants; std::vector<uint32_t> x_formant_off;      // ... of the rows of the last step, gathered at collect
    char* d_collect = nullptr; size_t collect_cap = 0;                        // collect's staging (levels 3 / 4 / 10): the step's pieces out of the rings, gathered by one kernel, fetched by one copy
    std::vector<uint64_t> x_trk_desc;
    double *d_state = nullptr, *d_fr_v = nullptr, *d_fr_fl = nullptr, *d_seg_d = nullptr, *d_feat_pool = nullptr, *d_feat = nullptr;
    int32_t *d_tr_state = nullptr, *d_fr_span = nullptr; char* d_tr_act = nullptr;      // incremental tracker: state of every stream between steps
    int32_t *d_fr_info = nullptr, *d_seg_i = nullptr, *d_meta_pool = nullptr, *d_meta = nullptr, *d_seg = nullptr, *d_carry = nullptr;
    uint32_t *d_seg_count = nullptr, *d_clip_rows = nullptr, *d_counters = nullptr, *d_row_off = nullptr, *d_seg_off = nullptr, *d_totals = nullptr;
    char* d_ws = nullptr;
    // pinned host side
    uint32_t* h_ctl = nullptr;              // [3][n]
    float* h_pcm = nullptr;                 // [n][step_samples]
    float* h_pcm_dev = nullptr;             // the same buffers as the device sees them
    uint32_t *h_ctl_dev = nullptr, *h_totals_dev = nullptr; int32_t *h_meta_dev = nullptr, *h_seg_dev = nullptr; double* h_feat_dev = nullptr;
    uint32_t* h_totals = nullptr;           // rows, segs, lost, flags, then per stream the spans cut at the ring's capacity
    hipStream_t last_stream = nullptr;      // where the previous step was enqueued
    int32_t *h_meta = nullptr, *h_seg = nullptr;
    double* h_feat = nullptr;
    uint32_t d2h_rows = 0, d2h_segs = 0, rows_cap = 0, segs_cap = 0;
    std::vector<int32_t> x_meta, x_seg; std::vector<double> x_feat;      // overflow of the fixed D2H window
    std::vector<uint32_t> warm;             // frames still to skip after START (windows reaching before time zero)
    uint64_t steps = 0;
    bool graph_on = false, stepped = false;
    hipGraphExec_t gexec = nullptr;
    hipStream_t own = nullptr; hipEvent_t ev_in = nullptr;   // the legacy NULL stream cannot be captured: steps given stream 0 run on `own`
    const float* g_pcm = nullptr; uint64_t g_stride = 0; hipStream_t g_stream = nullptr; bool g_host = false;
};

template <typename T>
static bool s_alloc(wsa_stream* b, T** p, size_t count, bool zero = false) {
    const size_t bytes = (count ? count : 1) * sizeof(T);
    void* qv = nullptr;
    if (hipMalloc(&qv, bytes) != hipSuccess) return false;
    b->allocs.push_back(qv);
    if (zero && hipMemset(qv, 0, bytes) != hipSuccess) return false;
    *p = reinterpret_cast<T*>(qv);
    return true;
}
template <typename T, typename U>
static bool s_upload(wsa_stream* b, T** p, const std::vector<U>& v) {
    const size_t count = v.size() * sizeof(U) / sizeof(T);
    if (!s_alloc(b, p, count)) return false;
    return v.empty() || hipMemcpy(*p, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice) == hipSuccess;
}

namespace wsa {
// history shuffle for overlapping windows: stage[s] = [last `hist` samples of the previous stage | new samples]
__global__ __launch_bounds__(256) void stream_stage_kernel(float* stage, uint32_t stage_stride, const float* pcm, uint64_t pcm_stride,
                                                           const uint32_t* ctl_bits, uint32_t hist, uint32_t step_samples) {
    extern __shared__ float s_hist[];
    const uint32_t s = blockIdx.x;
    if (!(ctl_bits[s] & 4u)) return;                       // bit 2 of the device control word: stream active in this step
    float* st = stage + (uint64_t)s * stage_stride;
    for (uint32_t i = threadIdx.x; i < hist; i += 256) s_hist[i] = st[step_samples + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < hist; i += 256) st[i] = s_hist[i];
    const float* src = pcm + (uint64_t)s * pcm_stride;
    for (uint32_t i = threadIdx.x; i < step_samples; i += 256) st[hist + i] = src[i];
}
// host -> device through the mapped pinned buffer (a kernel node: H2D memcpy nodes of more than a few KB from
// pinned memory faulted inside captured graphs on ROCm 7.2 / gfx950, the same copy outside a graph was fine)
__global__ __launch_bounds__(256) void stream_pull_kernel(float* dst, const float* __restrict__ src, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    else for (size_t k = i; k < n; k++) dst[k] = src[k];
}
// step prologue: control words host -> device (mapped pinned memory), counters cleared
// (level 3: the per (stream, k of this step) table of the segments' track pools starts every step empty, so that an entry the step did not write reads as
//  "no tracks", not as a previous step's pool offsets — cleared HERE, by a kernel: a memset node inside the captured step did not replay reliably on this ROCm)
__global__ __launch_bounds__(256) void stream_begin_kernel(uint32_t* d_ctl, const uint32_t* __restrict__ h_ctl, uint32_t words, uint32_t* counters, uint32_t* totals,
                                                           int32_t* trk_seg, uint32_t trk_seg_words) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < words) d_ctl[i] = h_ctl[i];
    if (i < 8) counters[i] = 0;
    if (i < 4) totals[i] = 0;
    for (uint32_t k = i; k < trk_seg_words; k += gridDim.x * 256) trk_seg[k] = 0;
}
// step epilogue: this step's totals and rows device -> host (mapped pinned memory); only what exists is sent
__global__ __launch_bounds__(256) void stream_push_kernel(const uint32_t* __restrict__ totals, const uint32_t* __restrict__ shared,
                                                          const int32_t* __restrict__ meta, const double* __restrict__ feat, const int32_t* __restrict__ seg,
                                                          uint32_t* h_totals, int32_t* h_meta, double* h_feat, int32_t* h_seg, uint32_t cap_rows, uint32_t cap_segs,
                                                          const double* __restrict__ state, uint32_t n_streams) {
    const uint32_t rows = min(totals[0], cap_rows), segs = min(totals[1], cap_segs);
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    for (uint32_t i = tid; i < rows * 8; i += nth) h_meta[i] = meta[i];
    for (uint32_t i = tid; i < rows * WSA_NFEAT; i += nth) h_feat[i] = feat[i];
    for (uint32_t i = tid; i < segs * 4; i += nth) h_seg[i] = seg[i];
    if (tid == 0) { h_totals[0] = totals[0]; h_totals[1] = totals[1]; h_totals[2] = totals[2]; h_totals[3] = shared[1]; }
    for (uint32_t i = tid; i < n_streams; i += nth) h_totals[4 + i] = (uint32_t)state[(uint64_t)i * GATE_STATE + 12];     // spans cut at the ring's capacity
}
}  // namespace wsa

extern "C" {

void wsa_stream_destroy(wsa_stream* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    if (b->gexec) (void)hipGraphExecDestroy(b->gexec);
    if (b->own) (void)hipStreamDestroy(b->own);
    if (b->ev_in) (void)hipEventDestroy(b->ev_in);
    for (void* p : b->allocs) (void)hipFree(p);
    if (b->d_collect) (void)hipFree(b->d_collect);
    for (void* p : {(void*)b->h_ctl, (void*)b->h_pcm, (void*)b->h_totals, (void*)b->h_meta, (void*)b->h_seg, (void*)b->h_feat}) if (p) (void)hipHostFree(p);
    delete b;
}

wsa_status wsa_stream_create(wsa_ctx* ctx, uint32_t n_streams, double fs, uint32_t frames_per_step, uint32_t max_span_frames, wsa_stream** out) {
    if (!ctx || !out || n_streams == 0 || frames_per_step == 0) return fail(ctx, WSA_ERR_INVALID, "bad stream arguments");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const wsa_config& c = ctx->cfg;
    if (!(c.output_level == 5 || c.output_level == 13 || c.output_level == 4 || c.output_level == 10 || c.output_level == 12 || c.output_level == 11 || c.output_level == 3))
        return fail(ctx, WSA_ERR_INVALID, "streams support output_level 3, 4, 5, 10, 11, 12 and 13");
    wsa_stream* b = new wsa_stream();
    b->tune = Tuning::from_env();
    b->ctx = ctx; b->n = n_streams; b->F = frames_per_step; b->fs = fs;
    std::string err;
    if (!build_fe_plan(c, fs, b->plan, err)) { delete b; return fail(ctx, WSA_ERR_INVALID, err); }
    const FePlanHost& P = b->plan;
    if (!fe_supported_R(P.R, P.three)) { delete b; return fail(ctx, WSA_ERR_INVALID, "unsupported FFT length for this sample rate / band setting"); }
    if (const size_t need = fe_lds_required(P, b->tune.fe_fat); need > 160 * 1024) { delete b; return fail(ctx, WSA_ERR_INVALID, "this window / band setting needs " + std::to_string(need) + " bytes of LDS for the front end's tables (limit 163840)"); }
    b->q = (uint32_t)((P.win + P.hop - 1) / P.hop);
    b->hist = (b->q - 1) * (uint32_t)P.hop;
    b->step_samples = b->F * (uint32_t)P.hop;
    b->stage_stride = (b->hist + b->step_samples + (uint32_t)P.win + 3u) & ~3u;
    uint32_t want = max_span_frames ? max_span_frames : 1024u;
    if (want < 2 * b->F + 64) want = 2 * b->F + 64;
    uint32_t ring = 64; while (ring < want + b->F) ring <<= 1;
    b->ring = ring;
    const double breaker = c.pause_length > 2 * c.window_step ? c.pause_length / c.window_step : 250 / c.window_step;
    const double min_frames = std::trunc(c.min_seg_length / c.window_step);
    const int period = (int)min_frames + 1 + (int)std::floor(breaker);
    b->fcap = (int)ring + 2;
    b->seg_cap = (int)b->F / (period > 0 ? period : 1) + 3;
    b->row_cap = (c.output_level == 10 || c.output_level == 11 || c.output_level == 12 || c.output_level == 13) ? (int)(ring + b->F) / 2 + 4 : b->seg_cap;
    b->tcap = ((P.bands + 1) / 2) * b->fcap; b->pcap = b->tcap;
    b->ws_stride = tracker_ws_bytes(b->tcap, b->pcap, b->fcap, c.output_level == 3);
    size_t waves = ((size_t)2 << 30) / (b->ws_stride ? b->ws_stride : 1);
    if (waves > (size_t)ctx->n_cu) waves = (size_t)ctx->n_cu;
    if (waves > (size_t)n_streams * (size_t)b->seg_cap) waves = (size_t)n_streams * (size_t)b->seg_cap;
    if (waves < 1) waves = 1;
    b->n_waves = (int)waves;
    b->rows_cap = n_streams * (uint32_t)b->row_cap; b->segs_cap = n_streams * (uint32_t)b->seg_cap;
    b->d2h_rows = b->rows_cap < 1024u ? b->rows_cap : 1024u;
    b->d2h_segs = b->segs_cap < 1024u ? b->segs_cap : 1024u;
    b->warm.assign(n_streams, 0);

    std::vector<uint32_t> foff(n_streams + 1);
    for (uint32_t i = 0; i <= n_streams; i++) foff[i] = i * b->F;
    std::vector<uint32_t> roff(n_streams + 1);
    for (uint32_t i = 0; i <= n_streams; i++) roff[i] = i * ring;
    const size_t nfr_ring = (size_t)n_streams * ring;
    bool ok = s_upload(b, &b->d_window, P.window) && s_upload(b, &b->d_tw_n2, P.tw_n2) && s_upload(b, &b->d_tw_m, P.tw_m) && s_upload(b, &b->d_tw_64, P.tw_64)
           && s_upload(b, &b->d_tw_nfft, P.tw_nfft) && s_upload(b, &b->d_mel_k0, P.mel_k0) && s_upload(b, &b->d_mel_cnt, P.mel_cnt)
           && s_upload(b, &b->d_mel_off, P.mel_off) && s_upload(b, &b->d_mel_w, P.mel_w) && s_upload(b, &b->d_emph, P.emph)
           && s_upload(b, &b->d_frame_off, foff) && s_upload(b, &b->d_ring_off, roff)
           && s_alloc(b, &b->d_ctl, (size_t)3 * n_streams, true) && s_alloc(b, &b->d_spec, (size_t)n_streams * b->F * P.bands)
           && s_alloc(b, &b->rec.hdr, nfr_ring) && s_alloc(b, &b->rec.amp, nfr_ring * CAND_CAP) && s_alloc(b, &b->rec.ent, nfr_ring * CAND_CAP) && s_alloc(b, &b->d_state, (size_t)n_streams * GATE_STATE, true)
           && s_alloc(b, &b->d_fr_info, nfr_ring) && s_alloc(b, &b->d_fr_v, nfr_ring) && s_alloc(b, &b->d_fr_fl, nfr_ring)
           && s_alloc(b, &b->d_seg_i, (size_t)n_streams * b->seg_cap * 8) && s_alloc(b, &b->d_seg_d, (size_t)n_streams * b->seg_cap * 2)
           && s_alloc(b, &b->d_seg_count, (size_t)n_streams, true) && s_alloc(b, &b->d_clip_rows, (size_t)n_streams, true)
           && ((c.output_level != 4 && c.output_level != 10 && c.output_level != 12 && c.output_level != 11) || s_alloc(b, &b->d_formants, (size_t)n_streams * b->ring * 9, true))
           && (c.output_level != 3 || (s_alloc(b, &b->d_trk_pts, (size_t)n_streams * b->ring * 64 * 2) && s_alloc(b, &b->d_trk_rank, (size_t)n_streams * b->ring * 64)
                                       && s_alloc(b, &b->d_trk_seg, (size_t)n_streams * b->seg_cap * 4, true)))
           && (c.output_level != 11 || (s_alloc(b, &b->d_utt_state, (size_t)n_streams * UTT_STATE_WORDS, true) && s_alloc(b, &b->d_utt_off, (size_t)n_streams + 1)
                                        && s_alloc(b, &b->d_utt_meta, (size_t)b->segs_cap * 4) && s_alloc(b, &b->d_utt_feat, (size_t)b->segs_cap * WSA_NUTT)))
           && (c.output_level != 12 || (s_alloc(b, &b->d_sums, (size_t)n_streams * b->ring, true) && s_alloc(b, &b->d_coef_ws, (size_t)8 * n_streams * 2 * b->ring)))
           && s_alloc(b, &b->d_meta_pool, (size_t)b->rows_cap * 8) && s_alloc(b, &b->d_feat_pool, (size_t)b->rows_cap * WSA_NFEAT)
           && s_alloc(b, &b->d_meta, (size_t)b->rows_cap * 8) && s_alloc(b, &b->d_feat, (size_t)b->rows_cap * WSA_NFEAT)
           && s_alloc(b, &b->d_seg, (size_t)b->segs_cap * 4) && s_alloc(b, &b->d_carry, (size_t)n_streams * CARRY_WORDS, true)
           && s_alloc(b, &b->d_counters, 8, true) && s_alloc(b, &b->d_row_off, (size_t)n_streams + 1) && s_alloc(b, &b->d_seg_off, (size_t)n_streams + 1)
           && s_alloc(b, &b->d_totals, 4, true) && s_alloc(b, &b->d_ws, b->ws_stride * (size_t)n_streams, true)      /* one tracker work space per stream (its filing generations start at zero) */
           && s_alloc(b, &b->d_tr_state, (size_t)n_streams * TR_STATE_WORDS, true) && s_alloc(b, &b->d_tr_act, (size_t)n_streams * TR_ACT_BYTES, true)
           && s_alloc(b, &b->d_fr_span, nfr_ring, true)
           && s_alloc(b, &b->d_pcm_in, (size_t)n_streams * b->step_samples);
    if (ok && b->hist) ok = s_alloc(b, &b->d_stage, (size_t)n_streams * b->stage_stride, true);
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&b->h_ctl), (size_t)3 * n_streams * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess
            && hipHostMalloc(reinterpret_cast<void**>(&b->h_pcm), (size_t)n_streams * b->step_samples * sizeof(float), hipHostMallocMapped) == hipSuccess
            && hipHostMalloc(reinterpret_cast<void**>(&b->h_totals), (4 + (size_t)n_streams) * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess
            && hipHostMalloc(reinterpret_cast<void**>(&b->h_meta), (size_t)(b->d2h_rows ? b->d2h_rows : 1) * 8 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess
            && hipHostMalloc(reinterpret_cast<void**>(&b->h_feat), (size_t)(b->d2h_rows ? b->d2h_rows : 1) * WSA_NFEAT * sizeof(double), hipHostMallocMapped) == hipSuccess
            && hipHostMalloc(reinterpret_cast<void**>(&b->h_seg), (size_t)(b->d2h_segs ? b->d2h_segs : 1) * 4 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&b->own, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        const std::string m = std::string("stream allocation failed: ") + hipGetErrorString(hipGetLastError());
        wsa_stream_destroy(b);
        return fail(ctx, WSA_ERR_HIP, m);
    }
    std::memset(b->h_pcm, 0, (size_t)n_streams * b->step_samples * sizeof(float));
    std::memset(b->h_totals, 0, (4 + (size_t)n_streams) * sizeof(uint32_t));
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_pcm_dev), b->h_pcm, 0) != hipSuccess
        || hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_ctl_dev), b->h_ctl, 0) != hipSuccess
        || hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_totals_dev), b->h_totals, 0) != hipSuccess
        || hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_meta_dev), b->h_meta, 0) != hipSuccess
        || hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_feat_dev), b->h_feat, 0) != hipSuccess
        || hipHostGetDevicePointer(reinterpret_cast<void**>(&b->h_seg_dev), b->h_seg, 0) != hipSuccess) {
        wsa_stream_destroy(b);
        return fail(ctx, WSA_ERR_HIP, "hipHostGetDevicePointer failed for a pinned stream buffer");
    }
    *out = b;
    return WSA_OK;
}

uint32_t wsa_stream_samples_per_step(const wsa_stream* b) { return b ? b->step_samples : 0; }
float* wsa_stream_host_input(wsa_stream* b) { return b ? b->h_pcm : nullptr; }

wsa_status wsa_stream_enable_graph(wsa_stream* b, int32_t on) {
    if (!b) return WSA_ERR_INVALID;
    b->graph_on = on != 0;
    if (!on && b->gexec) { (void)hipGraphExecDestroy(b->gexec); b->gexec = nullptr; }
    return WSA_OK;
}

// everything one step puts on the stream (this is what the graph holds)
static wsa_status enqueue_step(wsa_stream* b, const float* d_pcm, uint64_t stride, bool host_in, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    const wsa_config& c = ctx->cfg;
    const FePlanHost& P = b->plan;
    const uint32_t n = b->n;
    // no memcpy / memset nodes: everything that crosses PCIe goes through mapped pinned buffers, moved by kernels
    hipLaunchKernelGGL(stream_begin_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, s, b->d_ctl, b->h_ctl_dev, 3 * n, b->d_counters, b->d_totals,
                       b->d_trk_seg, b->d_trk_seg ? (uint32_t)((size_t)n * b->seg_cap * 4) : 0u);
    if (host_in) {
        const size_t cnt = (size_t)n * b->step_samples;
        hipLaunchKernelGGL(stream_pull_kernel, dim3((unsigned)((cnt / 4 + 256) / 256)), dim3(256), 0, s, b->d_pcm_in, b->h_pcm_dev, cnt);
        d_pcm = b->d_pcm_in; stride = b->step_samples;
    }
    const uint32_t *d_nfr = b->d_ctl, *d_off = b->d_ctl + n, *d_bits = b->d_ctl + 2 * n;
    GateParams g;
    g.auto_gate = c.auto_noise_gate ? 1 : 0;
    if (g.auto_gate) { g.ctx_max0 = 50; g.floor0 = 2; }                                            // ref @B25471
    else { g.ctx_max0 = std::pow(10.0, c.voiced_max_dB / 20); g.floor0 = std::pow(10.0, c.voiced_min_dB / 20); }
    launch_stream_prepare(b->d_state, b->d_carry, b->d_tr_state, d_bits, n, g.ctx_max0, g.floor0, s);
    if (b->hist) {
        hipLaunchKernelGGL(stream_stage_kernel, dim3(n), dim3(256), (size_t)b->hist * sizeof(float), s,
                           b->d_stage, b->stage_stride, d_pcm, stride, d_bits, b->hist, b->step_samples);
        d_pcm = b->d_stage; stride = b->stage_stride;
    }
    FeParams p;
    p.pcm = d_pcm; p.clip_stride = stride; p.n_frames = d_nfr; p.frame_off = b->d_frame_off; p.spec = b->d_spec;
    p.win = P.win; p.hop = P.hop; p.kmax = P.kmax; p.bands = P.bands; p.spec_type = P.spec_type; p.mel_total = (int)P.mel_w.size();
    p.mel_max_taps = 0; for (int32_t c_ : P.mel_cnt) if (c_ > p.mel_max_taps) p.mel_max_taps = c_;
    p.mel_max_taps_lo = 0; for (size_t i_ = 0; i_ < P.mel_cnt.size() && i_ < 64; i_++) if (P.mel_cnt[i_] > p.mel_max_taps_lo) p.mel_max_taps_lo = P.mel_cnt[i_];
    p.frames_per_wave = (int)((b->F + 3) / 4); if (p.frames_per_wave > 25) p.frames_per_wave = 25;
    p.window = b->d_window; p.tw_n2 = b->d_tw_n2; p.tw_64 = b->d_tw_64; p.tw_nfft = b->d_tw_nfft; p.tw_m = b->d_tw_m;
    p.mel_k0 = b->d_mel_k0; p.mel_cnt = b->d_mel_cnt; p.mel_off = b->d_mel_off; p.mel_w = b->d_mel_w; p.emph = b->d_emph; p.gain = P.gain;
    p.pcm_off = d_off; p.fat = b->tune.fe_fat ? 1 : 0; p.wg_per_cu = 0; p.queue = nullptr; p.chunks_per_clip = 0; p.n_chunks = 0; p.n_cu = 0;
    launch_frontend(p, (int)n, (int)b->F, P.R, P.three, s);
    PkParams pk;
    pk.spec = b->d_spec; pk.rec = b->rec; pk.frame0 = 0; pk.total_frames = n * b->F; pk.bands = P.bands;
    pk.stream_state = b->d_state; pk.n_frames = d_nfr; pk.step_frames = b->F; pk.ring = b->ring; pk.flags = b->d_counters + 1; pk.dbg = 0; pk.lanes_only = b->tune.peaks_lanes ? 1 : 0; pk.wpc = 0; pk.round_bins = b->tune.peaks_w;
    launch_peaks(pk, s);
    g.rec = b->rec; g.n_frames = d_nfr; g.frame_off = nullptr; g.clip0 = 0; g.n_clips = n;
    const int klevel = (c.output_level == 12 || c.output_level == 11) ? 10 : c.output_level;      // levels 11 / 12 store what level 10 stores (12: + the energy sums; ref @B27713, @B27240)
    g.level = klevel;
    g.max_voiced_bin = (int)std::trunc(0.7 * P.bands);                                             // ref @B25136
    g.breaker = c.pause_length > 2 * c.window_step ? c.pause_length / c.window_step : 250 / c.window_step;   // ref @B25188
    g.min_frames = std::trunc(c.min_seg_length / c.window_step);                                   // ref @B25218
    g.fr_info = b->d_fr_info; g.fr_v = b->d_fr_v; g.fr_fl = b->d_fr_fl;
    g.seg_i = b->d_seg_i; g.seg_d = b->d_seg_d; g.seg_cap = b->seg_cap; g.seg_count = b->d_seg_count;
    g.clip_rows = b->d_clip_rows; g.counters = b->d_counters + 4; g.shared = b->d_counters; g.trace = nullptr; g.dbg = 0; g.strided = 1; g.span_hist = nullptr; g.span_key = nullptr;
    g.state = b->d_state; g.ctl = d_bits; g.ring = b->ring; g.step_frames = b->F; g.fr_span = b->d_fr_span; g.prio = 0;
    launch_gate_stream(g, s);
    TrParams t;
    t.rec = b->rec; t.frame_off = b->d_ring_off; t.level = klevel;
    t.fr_info = b->d_fr_info; t.fr_v = b->d_fr_v; t.fr_fl = b->d_fr_fl;
    t.seg_i = b->d_seg_i; t.seg_d = b->d_seg_d; t.seg_cap = b->seg_cap; t.seg_count = b->d_seg_count; t.n_clips = n; t.counters = b->d_counters + 4; t.shared = b->d_counters;
    t.ws = b->d_ws; t.ws_stride = b->ws_stride; t.tcap = b->tcap; t.pcap = b->pcap; t.fcap = b->fcap;
    t.row_meta = b->d_meta_pool; t.row_feat = b->d_feat_pool; t.row_cap = (uint32_t)b->row_cap; t.clip_rows = b->d_clip_rows; t.trace = nullptr; t.dbg = 0;
    t.pool = nullptr; t.pool_bpf = 0; t.span_hdr = nullptr; t.fin_waves = 0; t.quad = 0; t.quad_waves = 0;
    t.ring_mask = b->ring - 1; t.formants = b->d_formants; t.sums = b->d_sums; t.trk_pts = b->d_trk_pts; t.trk_rank = b->d_trk_rank; t.trk_seg = b->d_trk_seg; t.order = nullptr; t.order_cnt = 1; t.redo = nullptr; t.redo_count = nullptr;
    t.st_state = b->d_tr_state; t.st_act = b->d_tr_act; t.fr_span = b->d_fr_span; t.n_frames_step = d_nfr; t.gate_state = b->d_state;
    launch_tracker_stream(t, n, s);       // one wave per stream: this step's frames go into the stream's tracker state, closed segments are finalized
    CompactParams cp;
    cp.n_clips = n; cp.seg_cap = b->seg_cap; cp.level = klevel;
    cp.seg_i = b->d_seg_i; cp.seg_count = b->d_seg_count; cp.row_meta_in = b->d_meta_pool; cp.row_feat_in = b->d_feat_pool;
    cp.seg_out = b->d_seg; cp.row_meta_out = b->d_meta; cp.row_feat_out = b->d_feat;
    cp.clip_row_off = b->d_row_off; cp.clip_seg_off = b->d_seg_off; cp.totals = b->d_totals; cp.carry = b->d_carry; cp.ctl = d_bits; cp.clip_rows = nullptr; cp.flags = nullptr; cp.host = nullptr; cp.fused = 0; cp.clr_counters = nullptr; cp.clr_hist = nullptr;
    launch_compact(cp, s);
    HIP_TRY(ctx, hipGetLastError());
    if (c.output_level == 12) {            // K5 on the step's syllable rows: four polynomial fits each, frames and energy sums out of the rings
        CoefParams q;
        q.row_meta = b->d_meta; q.row_feat = b->d_feat; q.frame_off = b->d_ring_off; q.totals = b->d_totals; q.formants = b->d_formants; q.sums = b->d_sums;
        q.ws = b->d_coef_ws; q.total_frames = n * 2 * b->ring; q.shared = b->d_counters; q.ring_mask = b->ring - 1; q.scratch_stride = 2 * b->ring;
        launch_coeffs(q, b->rows_cap, s);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (c.output_level == 11) {            // K4 on the step's results: the launch's histograms are carried per stream
        UttParams u;
        u.n_clips = n; u.segments = b->d_seg; u.row_meta = b->d_meta; u.clip_seg_off = b->d_seg_off; u.clip_row_off = b->d_row_off;
        u.frame_off = b->d_ring_off; u.formants = b->d_formants; u.clip_utt_off = b->d_utt_off; u.utt_meta = b->d_utt_meta; u.utt_feat = b->d_utt_feat;
        u.totals = b->d_totals; u.state = b->d_utt_state; u.carry = b->d_carry; u.ctl = d_bits; u.ring_mask = b->ring - 1;
        launch_utterance(u, s);
        HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(stream_push_kernel, dim3(16), dim3(256), 0, s, b->d_totals, b->d_counters, b->d_meta, b->d_feat, b->d_seg,
                       b->h_totals_dev, b->h_meta_dev, b->h_feat_dev, b->h_seg_dev, b->d2h_rows, b->d2h_segs, b->d_state, b->n);
    HIP_TRY(ctx, hipGetLastError());
    return WSA_OK;
}

static wsa_status step_impl(wsa_stream* b, const float* d_pcm, uint64_t stride, bool host_in, const uint8_t* ctl, hipStream_t s) {
    wsa_ctx* ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!s) {                      // NULL stream: run on the object's own stream, after what the NULL stream holds now
        HIP_TRY(ctx, hipEventRecord(b->ev_in, nullptr));
        s = b->own;
        HIP_TRY(ctx, hipStreamWaitEvent(s, b->ev_in, 0));
    }
    if (!host_in && !d_pcm) return fail(ctx, WSA_ERR_INVALID, "null PCM pointer");
    if (!host_in && stride < b->step_samples && b->n > 1) return fail(ctx, WSA_ERR_INVALID, "stream_stride smaller than samples_per_step");
    // the pinned control words are read by the step's first H2D copy: the previous step must be done
    if (b->stepped) HIP_TRY(ctx, hipStreamSynchronize(b->last_stream ? b->last_stream : s));      // ... on whichever stream it ran
    b->last_stream = s;
    const uint32_t n = b->n, F = b->F;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t cb = ctl ? ctl[i] : (b->steps == 0 ? (WSA_STREAM_ACTIVE | WSA_STREAM_START) : WSA_STREAM_ACTIVE);
        uint32_t bits = 0, nfr = 0, off = 0;
        if (cb & WSA_STREAM_START) { b->warm[i] = b->q - 1; bits |= 1u; }
        if (cb & WSA_STREAM_ACTIVE) {
            const uint32_t skip = b->warm[i] < F ? b->warm[i] : F;
            b->warm[i] -= skip; nfr = F - skip; off = skip * (uint32_t)b->plan.hop; bits |= 4u;
        }
        if (cb & WSA_STREAM_STOP) bits |= 2u;
        b->h_ctl[i] = nfr; b->h_ctl[n + i] = off; b->h_ctl[2 * n + i] = bits;
    }
    if (b->graph_on && b->steps >= 1) {
        if (b->gexec && (b->g_pcm != d_pcm || b->g_stride != stride || b->g_stream != s || b->g_host != host_in)) { (void)hipGraphExecDestroy(b->gexec); b->gexec = nullptr; }
        if (!b->gexec) {
            hipGraph_t graph = nullptr;
            HIP_TRY(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const wsa_status st = enqueue_step(b, d_pcm, stride, host_in, s);
            const hipError_t e = hipStreamEndCapture(s, &graph);
            if (st != WSA_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
            if (e != hipSuccess) return fail(ctx, WSA_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            const hipError_t e2 = hipGraphInstantiate(&b->gexec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e2 != hipSuccess) { b->gexec = nullptr; return fail(ctx, WSA_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
            b->g_pcm = d_pcm; b->g_stride = stride; b->g_stream = s; b->g_host = host_in;
        }
        HIP_TRY(ctx, hipGraphLaunch(b->gexec, s));
    } else {
        const wsa_status st = enqueue_step(b, d_pcm, stride, host_in, s);
        if (st != WSA_OK) return st;
    }
    b->steps++; b->stepped = true;
    return WSA_OK;
}

wsa_status wsa_stream_step(wsa_stream* b, const float* d_pcm, uint64_t stream_stride, const uint8_t* ctl, void* stream) {
    if (!b) return WSA_ERR_INVALID;
    return step_impl(b, d_pcm, stream_stride, false, ctl, reinterpret_cast<hipStream_t>(stream));
}
wsa_status wsa_stream_step_host(wsa_stream* b, const uint8_t* ctl, void* stream) {
    if (!b) return WSA_ERR_INVALID;
    return step_impl(b, nullptr, 0, true, ctl, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- collect helpers (levels 3 / 4 / 10): a step's rows / segments point into the streams' rings; one kernel gathers the pieces (a span
// may wrap around its ring) into a staging buffer in the order the host hands them out, and one copy fetches them — instead of a
// synchronous copy per piece (dozens per step at many streams).
__global__ __launch_bounds__(64) void stream_gather_formants_kernel(const int32_t* meta, uint32_t rows, const float* formants, uint32_t ring, float* out) {
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    uint32_t off = 0;                                        // frames of the rows in front of this one (rows per step: tens)
    for (uint32_t q = lane; q < r; q += 64) off += (uint32_t)meta[8 * q + 7];
    for (int d = 32; d > 0; d >>= 1) off += (uint32_t)__shfl_xor((int)off, d, 64);
    const uint32_t sidx = (uint32_t)meta[8 * r], f0 = (uint32_t)meta[8 * r + 6], len = (uint32_t)meta[8 * r + 7];
    for (uint32_t i = lane; i < len * 9u; i += 64) {
        const uint32_t fr = i / 9u, c = i - fr * 9u;
        out[(size_t)(off + fr) * 9 + c] = formants[((size_t)sidx * ring + ((f0 + fr) & (ring - 1))) * 9 + c];
    }
}
namespace wsa {
// level 3, batch and streams: the segments' raw-track pieces out of the pools into one staging buffer in the order the host hands them out.
// desc per segment: {first pool entry (absolute), the base of its pool region, points, ranked ids, points / ranked ids of the segments in front};
// a stream's pool is a ring of `region` entries (a batch's is not: region = 2^63)
__global__ __launch_bounds__(256) void gather_tracks_kernel(const uint64_t* desc, uint64_t region, const int4* pts, const int32_t* rank, int4* out_pts, int32_t* out_rank) {
    const uint64_t* d = desc + 6 * (size_t)blockIdx.x;
    const uint64_t pool0 = d[0], base = d[1], n_pt = d[2], nq = d[3], np = d[4], nr = d[5];
    const uint64_t off0 = pool0 - base;
    for (uint64_t i = threadIdx.x; i < 2 * n_pt; i += 256) out_pts[2 * np + i] = pts[2 * (base + (off0 + (i >> 1)) % region) + (i & 1)];
    for (uint64_t i = threadIdx.x; i < nq; i += 256) out_rank[nr + i] = rank[base + (off0 + i) % region];
}
void launch_gather_tracks(const uint64_t* desc, uint32_t n_segments, uint64_t region, const int4* pts, const int32_t* rank, int4* out_pts, int32_t* out_rank, hipStream_t s) {
    if (n_segments) hipLaunchKernelGGL(gather_tracks_kernel, dim3(n_segments), dim3(256), 0, s, desc, region, pts, rank, out_pts, out_rank);
}
}  // namespace wsa

static bool collect_stage(wsa_stream* b, size_t bytes) {
    if (bytes <= b->collect_cap) return true;
    if (b->d_collect) { (void)hipFree(b->d_collect); b->d_collect = nullptr; b->collect_cap = 0; }
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipMalloc(reinterpret_cast<void**>(&b->d_collect), want) != hipSuccess) { (void)hipGetLastError(); return false; }
    b->collect_cap = want;
    return true;
}

extern "C" {

wsa_status wsa_stream_collect(wsa_stream* b, void* stream, wsa_stream_rows* o) {
    if (!b || !o) return WSA_ERR_INVALID;
    wsa_ctx* ctx = b->ctx;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (!s) s = b->own;
    if (!b->stepped) return fail(ctx, WSA_ERR_INVALID, "no step on this stream object yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    const uint32_t rows = b->h_totals[0], segs = b->h_totals[1];
    o->n_rows = rows; o->n_segments = segs; o->status_flags = (b->h_totals[3] & 1u) | (b->h_totals[2] ? 1u : 0u);
    o->row_meta = b->h_meta; o->row_feat = b->h_feat; o->segments = b->h_seg; o->stream_cuts = b->h_totals + 4;
    // a span cut at the ring's capacity in this step is flagged for hosts that do not look at the per-stream counters
    if (b->seen_cuts.size() != b->n) b->seen_cuts.assign(b->n, 0u);
    for (uint32_t i = 0; i < b->n; i++) if (b->h_totals[4 + i] != b->seen_cuts[i]) { o->status_flags |= WSA_FLAG_STREAM_CUT; b->seen_cuts[i] = b->h_totals[4 + i]; }
    if (rows > b->d2h_rows) {                 // more rows than the fixed window of the step: fetch them all
        b->x_meta.resize((size_t)rows * 8); b->x_feat.resize((size_t)rows * WSA_NFEAT);
        HIP_TRY(ctx, hipMemcpy(b->x_meta.data(), b->d_meta, (size_t)rows * 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(b->x_feat.data(), b->d_feat, (size_t)rows * WSA_NFEAT * sizeof(double), hipMemcpyDeviceToHost));
        o->row_meta = b->x_meta.data(); o->row_feat = b->x_feat.data();
    }
    if (segs > b->d2h_segs) {
        b->x_seg.resize((size_t)segs * 4);
        HIP_TRY(ctx, hipMemcpy(b->x_seg.data(), b->d_seg, (size_t)segs * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
        o->segments = b->x_seg.data();
    }
    o->formants = nullptr; o->row_formant_off = nullptr;
    o->n_track_points = 0; o->n_track_ranked = 0; o->track_off = nullptr; o->track_points = nullptr; o->track_ranked = nullptr;
    if (b->d_trk_pts) {
        // level 3: the ranked raw tracks of every segment of this step (as wsa_batch_copy_tracks: offsets [n_segments + 1][2], points [8 ints], ranked
        // track ids), unwrapped out of the stream's pool ring.  Segments arrive in (stream, k) order; the device table is per (stream, k of this step)
        const int32_t* sgm = o->segments;
        b->x_trk_seg.resize((size_t)b->n * b->seg_cap * 4);
        if (segs) HIP_TRY(ctx, hipMemcpy(b->x_trk_seg.data(), b->d_trk_seg, b->x_trk_seg.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        b->x_trk_off.assign(2 * ((size_t)segs + 1), 0); b->x_trk_desc.resize(6 * (size_t)segs + 1);
        uint64_t np = 0, nr = 0; uint32_t kk = 0; int32_t last_stream = -1;
        const uint64_t region = (uint64_t)b->ring * 64;
        for (uint32_t q = 0; q < segs; q++) {
            const int32_t sidx = sgm[4 * q];
            kk = sidx == last_stream ? kk + 1 : 0; last_stream = sidx;
            const int32_t* t = &b->x_trk_seg[((size_t)sidx * b->seg_cap + kk) * 4];
            const uint64_t pool0 = (uint64_t)(uint32_t)t[0] | ((uint64_t)(uint32_t)t[3] << 32);
            const uint32_t n_pt = (uint32_t)t[1], nq = (uint32_t)t[2];
            b->x_trk_off[2 * q] = np; b->x_trk_off[2 * q + 1] = nr;
            uint64_t* d = &b->x_trk_desc[6 * (size_t)q];
            d[0] = pool0; d[1] = (uint64_t)sidx * region; d[2] = n_pt; d[3] = nq; d[4] = np; d[5] = nr;
            np += n_pt; nr += nq;
        }
        b->x_trk_pts.resize((size_t)np * 8 + 8); b->x_trk_rank.resize((size_t)nr + 1);
        if (np + nr) {
            // staging: [descriptors][points: 8 ints each][ranked ids]
            const size_t o_pts = ((size_t)segs * 6 * sizeof(uint64_t) + 255) & ~(size_t)255, o_rank = o_pts + (size_t)np * 8 * sizeof(int32_t);
            if (!collect_stage(b, o_rank + (size_t)nr * sizeof(int32_t))) return fail(ctx, WSA_ERR_HIP, "no device memory for the collect staging buffer");
            HIP_TRY(ctx, hipMemcpyAsync(b->d_collect, b->x_trk_desc.data(), (size_t)segs * 6 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            launch_gather_tracks(reinterpret_cast<const uint64_t*>(b->d_collect), segs, region, b->d_trk_pts, b->d_trk_rank,
                                 reinterpret_cast<int4*>(b->d_collect + o_pts), reinterpret_cast<int32_t*>(b->d_collect + o_rank), s);
            HIP_TRY(ctx, hipGetLastError());
            if (np) HIP_TRY(ctx, hipMemcpyAsync(b->x_trk_pts.data(), b->d_collect + o_pts, (size_t)np * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            if (nr) HIP_TRY(ctx, hipMemcpyAsync(b->x_trk_rank.data(), b->d_collect + o_rank, (size_t)nr * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
        }
        b->x_trk_off[2 * (size_t)segs] = np; b->x_trk_off[2 * (size_t)segs + 1] = nr;
        o->n_track_points = np; o->n_track_ranked = nr; o->track_off = b->x_trk_off.data(); o->track_points = b->x_trk_pts.data(); o->track_ranked = b->x_trk_rank.data();
    }
    o->n_utterance_rows = 0; o->utt_meta = nullptr; o->utt_feat = nullptr;
    if (b->d_utt_state) {
        // level 11: one 264-vector per result of this step (in (stream, result) order: the segments of the step that produced a result entry)
        uint32_t nu = 0;
        for (uint32_t k = 0; k < segs; k++) nu += o->segments[4 * k + 3] >= 0 ? 1u : 0u;
        b->x_utt_meta.resize((size_t)nu * 4 + 1); b->x_utt_feat.resize((size_t)nu * WSA_NUTT + 1);
        if (nu) {
            HIP_TRY(ctx, hipMemcpy(b->x_utt_meta.data(), b->d_utt_meta, (size_t)nu * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(b->x_utt_feat.data(), b->d_utt_feat, (size_t)nu * WSA_NUTT * sizeof(double), hipMemcpyDeviceToHost));
        }
        o->n_utterance_rows = nu; o->utt_meta = b->x_utt_meta.data(); o->utt_feat = b->x_utt_feat.data();
    }
    if (b->d_formants && !b->d_sums && !b->d_utt_state) {
        // levels 4 / 10: the straightened frames of every row's segment / syllable (meta[6] = first frame since the stream's START,
        // meta[7] frames) come out of the stream's ring: one gather kernel and one copy at collect time (not part of the graph)
        const int32_t* m = o->row_meta;
        b->x_formant_off.resize((size_t)rows + 1);
        size_t tot = 0;
        for (uint32_t r = 0; r < rows; r++) { b->x_formant_off[r] = (uint32_t)tot; tot += (size_t)m[8 * r + 7]; }
        b->x_formant_off[rows] = (uint32_t)tot;
        b->x_formants.resize(tot * 9 + 1);
        if (tot) {
            // the rows' table on the device is the one the host holds (b->d_meta: compacted rows of this step)
            if (!collect_stage(b, tot * 9 * sizeof(float))) return fail(ctx, WSA_ERR_HIP, "no device memory for the collect staging buffer");
            hipLaunchKernelGGL(stream_gather_formants_kernel, dim3(rows), dim3(64), 0, s, b->d_meta, rows, b->d_formants, b->ring, reinterpret_cast<float*>(b->d_collect));
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(b->x_formants.data(), b->d_collect, tot * 9 * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
        }
        o->formants = b->x_formants.data(); o->row_formant_off = b->x_formant_off.data();
    }
    if (o->status_flags & 1u)
        return fail(ctx, WSA_ERR_CAPACITY, "a device-side arena overflowed; results are invalid (step "
                    + std::to_string(b->steps) + ", flags " + std::to_string(b->h_totals[3]) + ", history " + std::to_string(b->h_totals[2]) + ")");
    return WSA_OK;
}

// Timed steps for the latency figure of BASELINE config 5: step k copies feed[k mod feed_steps] (n_streams x samples_per_step floats,
// the audio "arriving") into the pinned input buffer — outside the timed region — then times wsa_stream_step_host +
// wsa_stream_collect with the host's monotonic clock and notes the microseconds (no interpreter between the two calls).
wsa_status wsa_stream_time_steps(wsa_stream* b, uint32_t n_steps, const float* feed, uint32_t feed_steps, void* stream, double* out_us, uint64_t* rows_total) {
    if (!b || !out_us || (feed && feed_steps == 0)) return WSA_ERR_INVALID;
    const size_t words = (size_t)b->n * b->step_samples;
    uint64_t rows = 0;
    for (uint32_t k = 0; k < n_steps; k++) {
        if (feed) std::memcpy(b->h_pcm, feed + (size_t)(k % feed_steps) * words, words * sizeof(float));
        const auto t0 = std::chrono::steady_clock::now();
        wsa_status st = wsa_stream_step_host(b, nullptr, stream);
        wsa_stream_rows r;
        if (st == WSA_OK) st = wsa_stream_collect(b, stream, &r);
        const auto t1 = std::chrono::steady_clock::now();
        if (st != WSA_OK) return st;
        out_us[k] = std::chrono::duration<double, std::micro>(t1 - t0).count();
        rows += r.n_rows;
    }
    if (rows_total) *rows_total = rows;
    return WSA_OK;
}

}  // extern "C"

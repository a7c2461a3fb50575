// wsa_internal.hpp — shared declarations of libwsa (host plan + device kernel parameter blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/wsa.h"

namespace wsa {

// ---- front-end plan: everything (config, fs) determines; tables computed on the host in fp64 and
// rounded once to fp32 (DESIGN.md "FE-1").  Stands in for what the reference's worklet derives
// from its config message (ref dist/main.js:2 @B6726).
struct FePlanHost {
    int win = 0, hop = 0, nfft = 0, n2 = 0, R = 0, kmax = 0, bands = 0, spec_type = 1;
    int three = 0, M = 0;               // N2 = 3 M (three) or N2 = M; M = 64 R
    std::vector<float> window;          // win
    std::vector<float> tw_n2;           // 2*n2   W_N2^j = (cos, -sin)
    std::vector<float> tw_m;            // 2*M    W_M^j (only when three)
    std::vector<float> tw_64;           // 2*64
    std::vector<float> tw_nfft;         // 2*(kmax+1)
    std::vector<int32_t> mel_k0, mel_cnt, mel_off;
    std::vector<float> mel_w;           // 0.25 * triangle weights, flat
    std::vector<float> emph;            // bands
    float gain = 0;
    std::vector<double> bins_hz;        // bands
};
bool build_fe_plan(const wsa_config& cfg, double fs, FePlanHost& out, std::string& err);

// ---- tuning and test switches (tools/README.md).  The environment is read ONCE per planned batch / stream set (wsa_batch_create,
// wsa_stream_create) and the values travel in the plan: no launch path looks at the environment.
struct Tuning {
    int dbg = 0;                        // WSA_DBG
    bool no_pair = false, no_split = false, fe_fat = false, peaks_lanes = false;      // WSA_NO_PAIR, WSA_NO_SPLIT, WSA_FE_FAT, WSA_PEAKS_LANES
    bool no_quad = false, quad = false; // WSA_NO_QUAD / WSA_QUAD: two / four spans per wave in the split tracker's tracking kernel whatever the batch size
    bool no_fuse = false;               // WSA_NO_FUSE: the separate scan / gather / publish kernels at the end of a run instead of the fused compaction
    int full_table = -1;                // WSA_FULL_TABLE (-1: not set)
    int tracker_wpc = 0, fin_wpc = 0;   // WSA_TRACKER_WPC, WSA_FIN_WPC: waves per CU of the tracking / finalize kernels (0: default)
    int fpw = 0;                        // WSA_FPW: frames per front-end wave (0: default)
    int fe_wg_per_cu = 0;               // WSA_FE_WGS: workgroups per CU of the persistent 1024-point front end (1 .. 4; 0: default); -1 .. -3: one chunk per workgroup, capped by LDS padding
    bool fe_no_queue = false;           // WSA_FE_NO_QUEUE: one chunk per workgroup instead of the persistent launch
    int peaks_wpc = 0;                  // WSA_PEAKS_WPC: cap on the peak scan's waves per CU (same mechanism)
    int peaks_w = 0;                    // WSA_PEAKS_W: bins per round of the lane-per-frame peak scan, 16 or 32 (0: default)
    int upload_threads = 0;             // WSA_UPLOAD_THREADS (0: default)
    int rs_s = 0, rs_j = 0, rs_c = 0;   // WSA_RS_S / WSA_RS_J / WSA_RS_C: the rate converter's outputs per block row / per lane and run, runs per block (0: default)
    static Tuning from_env();
};

struct FeParams {
    const float* pcm; uint64_t clip_stride;
    const uint32_t* n_frames;           // [n_clips]
    const uint32_t* frame_off;          // [n_clips+1]
    uint32_t* spec;                     // [total_frames][bands]
    int win, hop, kmax, bands, spec_type, frames_per_wave, mel_total, mel_max_taps;
    int mel_max_taps_lo;                // ... of the bands below 64 (the lower band of every lane)
    const float* window; const float2* tw_n2; const float2* tw_64; const float2* tw_nfft;
    const float2* tw_m;                 // W_M^j of the three M-point transforms behind the radix-3 stage (NFFT = 3 * 2^k), else nullptr
    const int32_t* mel_k0; const int32_t* mel_cnt; const int32_t* mel_off; const float* mel_w;
    const float* emph; float gain;
    const uint32_t* pcm_off;            // optional per-clip sample offset into the clip's PCM (streaming warm-up), or nullptr
    int fat, wg_per_cu;                 // host side only (Tuning::fe_fat, fe_wg_per_cu)
    // persistent launch of the 1024-point kernel: workgroups take chunks of 4 x frames_per_wave frames of a clip from this counter (zeroed before the
    // launch) until all n_chunks = chunks_per_clip x clips are handed out; nullptr: one chunk per workgroup (grid = chunks)
    uint32_t* queue; uint32_t chunks_per_clip, n_chunks; int n_cu;
};

// ---- per-frame peak candidates (output of the parallel half of the reference's frame loop D(), ref @B25827).
// Frame records = one 16-byte header per frame + a candidate table in structure-of-arrays form:
//   hdr[frame]  .x = low word of g (g = sum e[1..B-1] < 2^40), .y = high byte of g | n << 8 | bin of the largest candidate << 16,
//               .z = amplitude of the largest candidate (end-of-spectrum emission excluded, first one on ties; 0 if none),
//               .w = index of the frame's first candidate in the table
//   candidate c amp[c] = e[l];  ent[c] = { i | s << 8 | l << 16 | (end-of-spectrum emission) << 24 (shoulders already shrunk),
//               low word of P[i-1] = sum e[0..i-1], low word of P[s] = sum e[0..s], their high bytes (P[i-1] | P[s] << 8) } — exact
//               prefix sums below 2^40: any merged band sum e[st..en] is one subtraction of two of them.
// 20 bytes per candidate, a frame's candidates contiguous (the gate reads only hdr + amp).  Every frame (ring slot of a stream) has
// its own CAND_CAP entries; consumers find a frame's table through hdr.w.
constexpr int CAND_CAP = 64;               // candidates per frame (all a spectrum of <= 128 bands can have)
struct RecPtrs { uint4* hdr; uint32_t* amp; uint4* ent; };
struct PkParams {
    const uint32_t* spec; RecPtrs rec; uint32_t frame0, total_frames; int bands;   // frames [frame0, frame0 + total_frames)
    // streaming (stream_state != nullptr): spec holds step_frames frames per stream; frame j of stream s goes to
    // record slot s * ring + ((frames the stream has seen so far + j) & (ring - 1)); frames j >= n_frames[s] are skipped
    const double* stream_state; const uint32_t* n_frames; uint32_t step_frames, ring;
    uint32_t* flags;                    // bit 0 is raised when a frame holds more than CAND_CAP candidates (only possible above 128 bands)
    int dbg;                            // tuning experiments (TUNING=1 builds, wsa_debug_peaks_time): 1 no emission, 2 no state machine, 4 no mask pass
    int lanes_only, wpc;                // host side only (Tuning::peaks_lanes, peaks_wpc)
    int round_bins;                     // host side only: bins per round of the lane-per-frame kernel, 16 or 32 (0: default)
};

// ---- sequential half, split in two (DESIGN.md "back end"):
//   K2a gate kernel    one wavefront per CLIP: candidate gating, voiced state machine, auto noise gate.
//                      None of it depends on the formant tracks, so it runs ahead and emits (a) per frame
//                      what accumulate_fm needs and (b) the list of segments that reach a finalize.
//   K2b tracker kernel one wavefront per SEGMENT span: the tracker is cleared at every reset_segment,
//                      so spans are independent of each other and run in parallel.
struct GateParams {
    RecPtrs rec;
    const uint32_t* n_frames; const uint32_t* frame_off; uint32_t clip0, n_clips;     // clips [clip0, clip0 + n_clips)
    int level, max_voiced_bin; double breaker, min_frames; int auto_gate; double ctx_max0, floor0;   // ref @B24629
    int32_t* fr_info;                   // per frame: -1 = accumulate_fm not called, else filing index | stale << 30
    double* fr_v;                       // per frame: noise floor the peak scan used (`v` at frame start)
    double* fr_fl;                      // per frame: noise floor handed to accumulate_fm (after the gate)
    int32_t* seg_i; double* seg_d; int seg_cap;   // per clip [seg_cap][8] / [seg_cap][2], see SEG_* below
    uint32_t* seg_count;                // [n_clips]
    uint32_t* clip_rows;                // [n_clips] rows handed out of the clip's part of the row pool (zeroed here, bumped by the tracker)
    uint32_t* counters;                 // [0] largest number of segments any clip holds (the tracker's enumeration bound)
    uint32_t* shared;                   // batch-wide [1] flags (bit0 capacity overflow)
    double* trace; int dbg;
    // span order (batch): every finalized segment takes a rank inside the bucket of its span length (longest first) — an atomic on
    // span_hist[bucket] — and notes {bucket, rank} in span_key[clip * seg_cap + segment]; launch_span_order turns them into the sorted list.  nullptr: off
    uint32_t* span_hist; uint2* span_key;
    int strided;                        // 1: frame slot's candidates start at slot * CAND_CAP (what the peak scans write); 0: a producer that packs the tables (none at present)
    // streaming (gate_stream_kernel): per-stream state carried from step to step, ring-indexed per-frame arrays
    double* state;                      // [n_streams][GATE_STATE]
    const uint32_t* ctl;                // [n_streams] bit0: fresh stream (launch state) before this step, bit1: segment_truncate after it
    uint32_t ring, step_frames;         // ring = frames of history per stream (power of two)
    int32_t* fr_span;                   // streams: per frame (ring-indexed) the first frame of the span the frame's accumulate_fm call belongs to, or nullptr
    int prio;                           // 1: the batch gate kernel raises its wave priority
};
enum { GATE_STATE = 16 };               // doubles per stream: cur_frame, no_fm, c_ci, c_started, ctx_max, floor, last_max, last_floor, w, T, k, span_begin, spans cut at the ring's capacity
enum { SEG_START = 0, SEG_LEN = 1, SEG_FBEGIN = 2, SEG_FEND = 3, SEG_CCI = 4, SEG_FLAG = 5, SEG_NROWS = 6, SEG_ROW0 = 7 };

struct TrParams {
    RecPtrs rec;
    const uint32_t* frame_off;
    int level;
    const int32_t* fr_info; const double* fr_v; const double* fr_fl;
    int32_t* seg_i; const double* seg_d; int seg_cap;
    const uint32_t* seg_count; uint32_t n_clips; const uint32_t* counters; uint32_t* shared;
    char* ws; uint64_t ws_stride; int tcap, pcap, fcap;
    int32_t* row_meta; double* row_feat; uint32_t row_cap; uint32_t* clip_rows;     // row pool: row_cap rows per clip, filled in completion order
    double* trace;
    int dbg;                            // tuning experiments only (WSA_DBG)
    uint32_t ring_mask;                 // 0xffffffff for a batch; ring - 1 when frames live in per-stream rings
    float* formants;                    // levels 4 / 10: [total_frames][9] f32 straightened frames, or nullptr
    // incremental streaming (tracker_kernel_stream): tracker state of every stream between steps
    int32_t* st_state; char* st_act;    // [n_streams][TR_STATE_WORDS] counters + accumulators, [n_streams][TR_ACT_BYTES] the active-track table
    const int32_t* fr_span; const uint32_t* n_frames_step; const double* gate_state;   // GateParams::fr_span, frames of this step, GateParams::state
    int4* trk_pts; int32_t* trk_rank; int32_t* trk_seg;   // level 3: point pool [frames * 64][2 x int4], ranked track ids [frames * 64], per segment {pool offset lo, points, ranked, offset hi}
    const uint2* order;                 // batch: spans sorted by length (launch_span_order), counters[1] of them; nullptr: enumerate (clip, segment)
    float* sums;                        // level 12: [total_frames] f32 per-frame energy sum of straighten (ref sums[d][1]), or nullptr
    // paired spans (tracker_kernel_pair): spans the lock-step variant declines (more than 32 accepted peaks in a frame, more than 64 live tracks)
    // are listed here and redone by the one-span-per-wave kernel, which then runs with order = redo, order_cnt = 2
    uint2* redo; uint32_t* redo_count;
    int order_cnt;                      // `order` holds counters[order_cnt] entries
    // split finalize (tracker_kernel_pair_acc + tracker_kernel_finalize): a span's tracks and points live in ITS region of `pool` — pool_bpf bytes per frame
    // of the batch, the region of a span starts at its first frame — and the accumulate kernel leaves span_hdr[(clip * seg_cap + segment) * 8] =
    // {tracks, points, stale index, stale points, sum g, sum E, 1 (finalize) | 2 (arena overflow) | 0 (on the redo list)} for the finalize kernel
    char* pool; uint32_t pool_bpf; double* span_hdr;
    int fin_waves;                      // host side only: grid of the finalize kernel (0: 2 x the tracking kernel's; Tuning::fin_wpc)
    int quad, quad_waves;               // host side only: four spans per wave in the tracking kernel of the split tracker, its grid
};

struct CompactParams {
    uint32_t n_clips; int seg_cap, level;
    const int32_t* seg_i; const uint32_t* seg_count;
    const int32_t* row_meta_in; const double* row_feat_in;
    int32_t* seg_out; int32_t* row_meta_out; double* row_feat_out;
    uint32_t* clip_row_off; uint32_t* clip_seg_off; uint32_t* totals;   // totals[0]=rows, [1]=segs
    // streaming: the callback index and the segments_ci history continue across steps
    int32_t* carry;                     // [n_streams][CARRY_WORDS]: segments so far, results so far, last CARRY_HIST [start, len]
    const uint32_t* ctl;                // as GateParams::ctl
    // fused form (batches; compact_gather_kernel<true>): per-clip row counters as the tracker left them, the flag word, the host's mapped result words (or nullptr)
    const uint32_t* clip_rows; const uint32_t* flags; uint32_t* host; int fused;
    // fused form only: the run's last wave leaves the batch's 16 counters, its totals and its span histogram cleared for the next run (nullptr: the caller launches its clear kernel)
    uint32_t* clr_counters; uint32_t* clr_hist;
};
bool compact_is_fused(const CompactParams& p);
enum { CARRY_HIST = 32, CARRY_WORDS = 2 + 2 * CARRY_HIST };

// ---- K4 utterance features (output_level 11): reads the compacted level-10 products
struct UttParams {
    uint32_t n_clips;
    const int32_t* segments; const int32_t* row_meta;                       // compacted tables (K3 output)
    const uint32_t* clip_seg_off; const uint32_t* clip_row_off; const uint32_t* frame_off;
    const float* formants;
    uint32_t* clip_utt_off;             // [n_clips + 1]
    int32_t* utt_meta; double* utt_feat; // [results][4] = {clip, k, first start, sum of lengths}, [results][264]
    uint32_t* totals;                   // totals[3] = number of results
    // streams (state != nullptr): the tables hold this step's segments / rows only and everything the reference accumulates over a launch is
    // carried per stream: [UTT_STATE_WORDS] = 264 histogram bins, 16 ghost counters, results / segments so far, prev_end, tsum, first start;
    // carry = CompactParams::carry (segments_ci history, already advanced over this step), ctl as GateParams::ctl, frames in rings
    uint32_t* state; const int32_t* carry; const uint32_t* ctl; uint32_t ring_mask;
};
enum { UTT_STATE_WORDS = 288 };

// ---- K0 sample-rate conversion (spec RS-1): out[clip][n] from in[clip][...], one lane per output sample
constexpr int RS_TAPS = 32, RS_OFFS = 32;
struct RsParams {
    const float* in; uint64_t stride_in; float* out; uint64_t stride_out;
    const uint32_t* n_in; const uint32_t* n_out;     // [n_clips] samples per clip before / after
    int chunks;                                       // runs of S * J outputs a block converts one after the other
    const float* table; double ratio; int span, S, J; // the LDS image of the [33][32] offset kernels (resample_table_image), fs_in / fs_out, inputs a block of outputs touches, outputs per block row, outputs per lane
};
void build_resample_table(double fs_in, double fs_out, std::vector<float>& K);
void resample_table_image(const std::vector<float>& K, std::vector<float>& img);
uint64_t resample_length(uint64_t n_in, double fs_in, double fs_out);
int resample_stride(double fs_in, double fs_out);
int resample_span(double ratio, int S, int J);
int resample_outputs_per_lane(int S, double ratio);
void launch_resample(const RsParams& p, uint32_t n_clips, uint64_t max_out, hipStream_t s);

void launch_frontend(const FeParams& p, int n_clips, int max_frames, int R, int three, hipStream_t s);
bool fe_supported_R(int R, int three);  // packed FFT length 64 R, R in {2, 4, 8, 16, 32, 64}, or 3 * 64 R, R in {1, 2, 4, 8, 16, 32}
size_t fe_lds_required(const FePlanHost& P, bool fat);      // dynamic LDS of the front-end kernel this geometry selects (limit: 160 KB per workgroup)
void launch_peaks(const PkParams& p, hipStream_t s);
void launch_peaks_mode(const PkParams& p, int mode, hipStream_t s);   // 1: lane-per-frame kernel, 2: wave-per-frame kernel (tests)
void launch_gate(const GateParams& p, hipStream_t s);
void launch_gate_stream(const GateParams& p, hipStream_t s);
void launch_stream_prepare(double* state, int32_t* carry, int32_t* tr_state, const uint32_t* ctl, uint32_t n, double ctx_max0, double floor0, hipStream_t s);
void launch_tracker(const TrParams& p, int n_waves, bool full_table, bool pair, hipStream_t s);
enum { SPAN_BUCKETS = 2048 };          // span lengths 0 .. 2047+ frames, bucket = SPAN_BUCKETS - 1 - min(frames, SPAN_BUCKETS - 1)
void launch_span_order(const TrParams& p, uint32_t* span_hist, const uint2* span_key, uint2* order, uint32_t* counters, hipStream_t s);
void launch_tracker_stream(const TrParams& p, uint32_t n_streams, hipStream_t s);
// level 3: gathers the segments' raw-track pieces out of the pools (desc: 6 words per segment, stream_api.hip) into dense tables
void launch_gather_tracks(const uint64_t* desc, uint32_t n_segments, uint64_t region, const int4* pts, const int32_t* rank, int4* out_pts, int32_t* out_rank, hipStream_t s);
enum { TR_STATE_WORDS = 16, TR_ACT_MAX = 320, TR_ACT_BYTES = TR_ACT_MAX * 44 };
void launch_compact(const CompactParams& p, hipStream_t s);
void launch_utterance(const UttParams& p, hipStream_t s);

// ---- K5 polynomial coefficients (output_level 12): reads the compacted level-10 rows + frames + energy sums
struct CoefParams {
    const int32_t* row_meta; double* row_feat;      // compacted rows: 23 numbers go to row_feat[row][0..22]
    const uint32_t* frame_off; const uint32_t* totals;   // totals[0] = number of rows
    const float* formants; const float* sums;        // [total_frames][9], [total_frames]
    double* ws; uint32_t total_frames;               // scratch: 8 x total_frames doubles (points of the four fits)
    uint32_t* shared;                                // flags (bit 2: a fit hit numeric's "gradient fails" path)
    // streams: the frames live in per-stream rings (frame f of stream c at frame_off[c] + (f & ring_mask)); a syllable's scratch rows are
    // then taken from a per-stream region of scratch_stride (= 2 x ring) rows, where they do not wrap.  Batches: ring_mask = ~0, scratch_stride = 0
    uint32_t ring_mask, scratch_stride;
};
void launch_coeffs(const CoefParams& p, uint32_t rows_cap, hipStream_t s);
size_t tracker_ws_bytes(int tcap, int pcap, int fcap, bool raw_tracks);
size_t tracker_pool_bpf();

}  // namespace wsa

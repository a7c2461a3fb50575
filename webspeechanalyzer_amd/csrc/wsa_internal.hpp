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
    std::vector<float> window;          // win
    std::vector<float> tw_n2;           // 2*n2   W_N2^j = (cos, -sin)
    std::vector<float> tw_64;           // 2*64
    std::vector<float> tw_nfft;         // 2*(kmax+1)
    std::vector<int32_t> mel_k0, mel_cnt, mel_off;
    std::vector<float> mel_w;           // 0.25 * triangle weights, flat
    std::vector<float> emph;            // bands
    float gain = 0;
    std::vector<double> bins_hz;        // bands
};
bool build_fe_plan(const wsa_config& cfg, double fs, FePlanHost& out, std::string& err);

struct FeParams {
    const float* pcm; uint64_t clip_stride;
    const uint32_t* n_frames;           // [n_clips]
    const uint32_t* frame_off;          // [n_clips+1]
    uint32_t* spec;                     // [total_frames][bands]
    int win, hop, kmax, bands, spec_type, frames_per_wave, mel_total;
    const float* window; const float2* tw_n2; const float2* tw_64; const float2* tw_nfft;
    const int32_t* mel_k0; const int32_t* mel_cnt; const int32_t* mel_off; const float* mel_w;
    const float* emph; float gain;
};

// ---- per-frame peak candidates (output of the parallel half of the reference's frame loop D(),
// ref @B25827); peak word = i | s<<8 | l<<16 | (end-of-spectrum emission)<<24.
// frame record (u32 words, stride rec_stride = 4 + 6*64): [0..1] g (f64: sum e[1..B-1]), [2] n,
// [4..68) peak words, [68..132) amplitudes e[l], [132..260) f64 prefix sums at i (sum e[0..i-1]),
// [260..388) f64 prefix sums past s (sum e[0..s]) — so any merged band sum is one subtraction.
struct PkParams {
    const uint32_t* spec; uint32_t* rec; uint32_t total_frames; int bands, rec_stride;
};

// ---- tracker (sequential half, one wavefront per clip)
struct TrParams {
    const uint32_t* rec;                // frame records written by K1b
    const uint32_t* n_frames; const uint32_t* frame_off;
    uint32_t n_clips; int bands, rec_stride, level;
    // segmenter constants (ref reset_segmentation @B24629)
    int max_voiced_bin; double breaker, min_frames; int auto_gate; double ctx_max0, floor0;
    // per-resident-wave work space
    char* ws; uint64_t ws_stride;
    int tcap, pcap, fcap;               // tracks, points, frames per clip capacity
    // outputs with fixed per-clip strides (compacted afterwards)
    int32_t* seg_out;  int seg_cap;     // [n_clips][seg_cap][4]  {start,len,flag,nrows}
    int32_t* row_meta; double* row_feat; int row_cap;   // [n_clips][row_cap][8], [..][53]
    uint32_t* counts;                   // [n_clips][2] {n_seg, n_rows}
    uint32_t* flags;                    // [1] bit0 capacity overflow
    double* trace;                      // optional [total_frames][12] per-frame state (tests / debugging)
};

struct CompactParams {
    uint32_t n_clips; int seg_cap, row_cap, level;
    const int32_t* seg_in; const int32_t* row_meta_in; const double* row_feat_in; const uint32_t* counts;
    int32_t* seg_out; int32_t* row_meta_out; double* row_feat_out;
    uint32_t* clip_row_off; uint32_t* clip_seg_off; uint32_t* totals;   // totals[0]=rows, [1]=segs
};

void launch_frontend(const FeParams& p, int n_clips, int max_frames, int R, hipStream_t s);
void launch_peaks(const PkParams& p, hipStream_t s);
void launch_tracker(const TrParams& p, int n_waves, hipStream_t s);
void launch_compact(const CompactParams& p, hipStream_t s);
size_t tracker_ws_bytes(int tcap, int pcap, int fcap);

}  // namespace wsa

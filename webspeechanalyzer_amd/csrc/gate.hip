// gate.hip — K2a: per-clip sequential pass that does NOT need the formant tracks, ONE WAVEFRONT PER CLIP.
//
// Stands in for (ref = /root/reference/dist/main.js line 2, byte offsets):
//   frame loop D() after the peak scan            @B25717: candidate acceptance `e[l] > v`, n / d / h / p
//   start test / voiced test / counters           @B26527, @B26646
//   auto noise gate C(h)                          @B28506
//   the decision part of finalize O(e)            @B27088-27190 (len, start, `u.push([start,len])`)
//   reset_segment L(e)                            @B25649
// In the reference these run interleaved with accumulate_fm, but nothing here reads tracker state:
// accumulate_fm only ever writes tracks, and tracks are only read at finalize.  So this pass runs
// ahead over the whole clip and leaves, per frame, the arguments accumulate_fm was called with
// (filing index incl. the stale first-frame quirk, both noise floors), and per finalized segment the
// span of frames whose tracks it owns.  tracker.hip then processes all spans of all clips in parallel.
#include "wsa_internal.hpp"
#include "wave_ops.hpp"
#include "jsmath_device.hpp"
#include "gate_floor.hpp"

// (the scalar loops below name m0 as clobbered: v_writelane with a scalar value AND a scalar lane number needs the lane in m0 — the
//  constant bus feeds one SGPR per instruction; the compiler flags a clobbered "reserved register", which is the point)
#pragma clang diagnostic ignored "-Winline-asm"

namespace wsa {


// ST = false: a whole clip per wave (batch).  ST = true: one STEP of a stream per wave — the state is
// loaded from / stored to p.state, frame numbers are absolute (frames since the stream's launch) and
// index the per-stream rings modulo p.ring, the segment table holds this step's segments only, and the
// closing segment_truncate runs only when the host asks for it (ctl bit 1).
template <bool ST>
__global__ __launch_bounds__(64) void gate_kernel_t(GateParams p) {
    const int lane = threadIdx.x;
    for (uint32_t clip = p.clip0 + blockIdx.x; clip < p.clip0 + p.n_clips; clip += gridDim.x) {
        const uint32_t nfr = p.n_frames[clip];
        const uint32_t foff = ST ? clip * p.ring : p.frame_off[clip];
        const uint32_t fmask = ST ? p.ring - 1 : 0xffffffffu;
        int32_t* seg_i = p.seg_i + (uint64_t)clip * p.seg_cap * 8;
        double* seg_d = p.seg_d + (uint64_t)clip * p.seg_cap * 2;

        // ---- launch state (ref reset_segmentation @B24629)
        int cur_frame = 0, no_fm = 0, c_ci = 0, c_started = -1;
        double ctx_max = p.ctx_max0, floor_ = p.floor0, last_max = p.ctx_max0, last_floor = p.floor0;
        double gw = 0, gT = 0, gk = 0;               // gate counters w, T, k
        int nseg = 0, span_begin = 0, cuts = 0;
        bool overflow = false;
        double* st = ST ? p.state + (uint64_t)clip * GATE_STATE : nullptr;
        if (ST) {
            cur_frame = (int)st[0]; no_fm = (int)st[1]; c_ci = (int)st[2]; c_started = (int)st[3];
            ctx_max = st[4]; floor_ = st[5]; last_max = st[6]; last_floor = st[7]; gw = st[8]; gT = st[9]; gk = st[10];
            span_begin = (int)st[11]; cuts = (int)st[12];
        }
        const uint32_t fbase = (uint32_t)cur_frame;      // absolute number of this step's first frame (0 for a batch)

        auto finalize = [&](int e_arg, int f_end) __attribute__((always_inline)) {      // ref @B27088
            const int len = e_arg - no_fm;
            if (!((double)len > p.min_frames && c_started >= 2)) return;
            if (nseg >= p.seg_cap) { overflow = true; return; }
            if (lane == 0) {
                int32_t* sg = seg_i + 8 * nseg;
                sg[SEG_START] = cur_frame - len; sg[SEG_LEN] = len; sg[SEG_FBEGIN] = span_begin; sg[SEG_FEND] = f_end;
                sg[SEG_CCI] = c_ci; sg[SEG_FLAG] = p.level == 3 ? 1 : 0; sg[SEG_NROWS] = 0; sg[SEG_ROW0] = 0;
                seg_d[2 * nseg] = ctx_max; seg_d[2 * nseg + 1] = floor_;
                if (!ST && p.span_hist) {            // the span's place in the tracker's longest-first order
                    const uint32_t bkt = (uint32_t)(SPAN_BUCKETS - 1) - min((uint32_t)(f_end - span_begin), (uint32_t)(SPAN_BUCKETS - 1));
                    p.span_key[(uint64_t)clip * p.seg_cap + nseg] = make_uint2(bkt, atomicAdd(&p.span_hist[bkt], 1u));
                }
            }
            nseg++;
        };
        bool gate_reset = false;
        auto noise_gate = [&](double h) __attribute__((always_inline)) {               // ref @B28506
            gw++;
            if (h > ctx_max || (gw > 40 && h > 2 * floor_)) {
                if (h >= ctx_max) { gw = 0; last_max = ctx_max = h; }
                else if (h > last_max / 100) { ctx_max -= trunc(ctx_max / 8); gw = 35; }
                const double y = ctx_max, t = jsm::log10(y);
                double v;
                if (t > 7) v = trunc(jsm::pow_pos(10, t - 3) / 20);
                else if (t > 6) v = trunc(jsm::pow_pos(10, t - 3) / 2);
                else if (t > 4) v = trunc(jsm::pow_pos(10, t - 2) / 2);
                else if (t > 2) v = trunc(jsm::pow_pos(10, t / 3));
                else if (t > 1) v = trunc(y / 10);
                else v = 1;
                floor_ = v; last_floor = v;
                if (gk > 0 && gT / gk < 30 * v) { c_ci = 0; c_started = 0; no_fm = 0; gate_reset = true; gk = 0; gT = 0; }   // L(0)
                gT += ctx_max; gk += 1;
            } else if (floor_ > 10 && floor_ > last_floor / 10 && gw > 20) {
                floor_ -= trunc(last_floor / 20);
                if (floor_ < 10) floor_ = 10;
            }
        };

        // Frame records are read in blocks of 64 frames: lane j fetches the header (g, n) of frame blk + j — one load
        // instruction per block, the next block's issued a block ahead — and v_readlane hands a frame its header.
        // Candidate amplitudes (lane = candidate) are requested a group of GF frames ahead of their
        // use.  The per-frame outputs are collected in lane j and leave as three coalesced stores per block, so that no
        // store sits in front of the next frame's `s_waitcnt` (stores count in vmcnt on this ISA).
        constexpr int GF = 4;
        const uint32_t fend = fbase + nfr;
        // (the header words stay as loaded until a frame consumes them: an ALU op on a prefetched value waits for the load on the spot)
        auto load_hdr_blk = [&](uint32_t blk, uint4& h) __attribute__((always_inline)) {
            const uint32_t f = min(blk + (uint32_t)lane, fend - 1);          // branch-free: lanes past the end read the last frame
            h = p.rec.hdr[foff + (f & fmask)];
        };
        // (only the amplitude of a candidate is needed here: which candidate is the largest is in the frame's header)
        auto load_ent = [&](uint32_t f, int n, uint32_t cbase, uint32_t& w) __attribute__((always_inline)) {
            w = 0u;
            if (f < fend && lane < n) w = p.rec.amp[cbase + (uint32_t)lane];
        };
        uint4 hd = make_uint4(0u, 0u, 0u, 0u), hd2 = hd;       // .x g low word, .y g high byte | n << 8 | largest candidate's bin << 16, .z its amplitude, .w table index
        uint32_t e_ent[GF], x_ent[GF];
        if (nfr > 0) {
            load_hdr_blk(fbase, hd);
#pragma unroll
            for (int k = 0; k < GF; k++) load_ent(fbase + k, (read_lane_i32((int)hd.y, k) >> 8) & 0xff, (uint32_t)read_lane_i32((int)hd.w, k), e_ent[k]);
        }
        for (uint32_t blk = fbase; blk < fend; blk += 64) {
          if (blk + 64 < fend) load_hdr_blk(blk + 64, hd2);
          int o_info = -1, o_span = 0; double o_v = 0, o_fl = 0;
          const int nblk = (int)min(64u, fend - blk);
          for (int j0 = 0; j0 < nblk; j0 += GF) {
            // entries of the next group (its headers are in this block's lanes, or in the next block's)
#pragma unroll
            for (int k = 0; k < GF; k++) {
                const int jn = j0 + GF + k;
                const int ny = jn < 64 ? read_lane_i32((int)hd.y, jn & 63) : read_lane_i32((int)hd2.y, jn & 63);
                const uint32_t nb = (uint32_t)(jn < 64 ? read_lane_i32((int)hd.w, jn & 63) : read_lane_i32((int)hd2.w, jn & 63));
                load_ent(blk + (uint32_t)jn, (ny >> 8) & 0xff, nb, x_ent[k]);
            }
#pragma unroll
            for (int k = 0; k < GF; k++) {
            const int jf = j0 + k;
            if (jf >= nblk) break;
            const uint32_t f = blk + (uint32_t)jf;
            const int hnw = read_lane_i32((int)hd.y, jf);
            const int ncand = (hnw >> 8) & 0xff;
            const double g = (double)(hnw & 0xff) * 4294967296.0 + (double)(uint32_t)read_lane_i32((int)hd.x, jf);      // exact: g < 2^40
            const uint32_t amp = e_ent[k];

            cur_frame++;
            const int t_idx = c_ci;                                  // captured before the start test (quirk 1)
            const double v = floor_;
            // ---- accept candidates (ref @B25827: `e[l] > v`), n, d, h, p
            const bool acc = lane < ncand && (double)amp > v;
            const int n = __popcll(__ballot(acc));
            const double d = wave_sum_int40(acc ? (uint64_t)amp : 0ull);
            // h / p: the largest accepted candidate other than the end-of-spectrum one, first one on ties — the peak scan
            // found the largest candidate already (record header); it is accepted whenever it exceeds h = 2v >= v
            const uint32_t mx = (uint32_t)read_lane_i32((int)hd.z, jf);
            double h = 2 * v; int pbin = 0;
            if (n > 0 && (double)mx > h) { h = mx; pbin = (hnw >> 16) & 0xff; }
            // ---- start test (ref @B26527)
            bool reset_before_acc = false;
            if (c_started < 0) {
                // ref: r = d>h ? h*(n-1)/(d-h) : 0;  r > 4.  All operands are integers below 2^38, so the rounded
                // quotient exceeds 4 exactly when h*(n-1) > 4*(d-h) (the gap to 4 is >= 1/(d-h) >> one ulp).
                // (h = 2v is an integer only under the auto noise gate; a fixed gate keeps the division.)
                const bool r_gt4 = p.auto_gate ? (d > h && h * (n - 1) > 4 * (d - h)) : ((d > h ? h * (n - 1) / (d - h) : 0) > 4);
                if (n > 0 && pbin > 7 && pbin < p.max_voiced_bin && n > 4 && r_gt4) {
                    c_ci = 0; c_started = 0; no_fm = 0; reset_before_acc = true; span_begin = (int)f;        // L(0)
                } else no_fm++;
            }
            bool do_reset = false;
            int info = -1;
            if (c_started >= 0) {                                    // ref @B26646
                if (n == 0 || pbin < 7 || pbin >= p.max_voiced_bin || (n > 3 && g - d > 0 && 10 * d < g - d)) {     // ref: d/(g-d) < .1 — exact: no integer ratio lies within 1e-17 of 1/10; d/0 = Infinity is not < .1
                    no_fm++;
                    if (c_started < 2) c_started--;
                    else if ((double)no_fm >= p.breaker) { finalize(c_ci + 1, (int)f + 1); do_reset = true; }
                    else if (p.auto_gate) { gate_reset = false; noise_gate(h); if (gate_reset) span_begin = (int)f + 1; }
                } else {
                    if (p.auto_gate) { gate_reset = false; noise_gate(h); if (gate_reset) { reset_before_acc = true; span_begin = (int)f; } }
                    info = t_idx | (reset_before_acc ? (1 << 30) : 0);      // accumulate_fm(e, peaks, t_idx, g, floor_)
                    if (c_started < 2) c_started++; else no_fm = 0;
                }
            }
            if (lane == jf) { o_info = info; o_v = v; o_fl = floor_; o_span = span_begin; }
            if (lane == 0) {
                if (!ST && p.trace && !(p.dbg & 16)) {
                    double* tr = p.trace + ((uint64_t)foff + f) * 12;
                    tr[0] = c_ci; tr[1] = c_started; tr[2] = no_fm; tr[3] = ctx_max; tr[4] = floor_; tr[5] = n; tr[6] = pbin;
                    tr[7] = h; tr[8] = d; tr[9] = g; tr[10] = 0; tr[11] = 0;
                }
            }
            c_ci++;
            if (do_reset) { c_ci = 0; c_started = -1; no_fm = 0; span_begin = (int)f + 1; }   // L(-1) in the Promise .then (quirk 8)
            // a started span must stay inside the ring together with the frames of one more step: a speaker who does not pause for
            // max_span_frames is cut there as if the source had been stopped and restarted (segment_truncate, ref @B30757: O(c_ci), L(1)) —
            // this stream's results differ from the reference's from here to the next pause, every other stream is untouched; counted in st[12]
            if (ST && c_started >= 0 && f + 1 - (uint32_t)span_begin + p.step_frames > p.ring) {
                finalize(c_ci, (int)f + 1);
                c_ci = 0; c_started = 1; no_fm = 0; span_begin = (int)f + 1; cuts++;
            }
            }
#pragma unroll
            for (int k = 0; k < GF; k++) e_ent[k] = x_ent[k];
          }
          if (lane < nblk) {
              const uint32_t fi = foff + ((blk + (uint32_t)lane) & fmask);
              p.fr_info[fi] = o_info; p.fr_v[fi] = o_v; p.fr_fl[fi] = o_fl;
              if (ST && p.fr_span) p.fr_span[fi] = o_span;
          }
          hd = hd2;
        }
        // ---- end of input: segment_truncate (ref @B30757) -> O(c_ci) -> L(1)
        if (!ST || (p.ctl[clip] & 2u)) {
            finalize(c_ci, (int)fend);
            c_ci = 0; c_started = 1; no_fm = 0; span_begin = (int)fend;       // L(1)
        }
        if (ST && lane == 0) {
            st[0] = cur_frame; st[1] = no_fm; st[2] = c_ci; st[3] = c_started; st[4] = ctx_max; st[5] = floor_; st[6] = last_max;
            st[7] = last_floor; st[8] = gw; st[9] = gT; st[10] = gk; st[11] = span_begin; st[12] = cuts;
        }
        // the tracker enumerates (clip, segment) pairs itself: it needs the per-clip counts and their maximum
        if (lane == 0) {
            p.seg_count[clip] = (uint32_t)nseg; p.clip_rows[clip] = 0;
            if (nseg > 0) atomicMax(&p.counters[0], (uint32_t)nseg);
            if (overflow) atomicOr(&p.shared[1], 1u);
        }
    }
}

// ---- batch + auto noise gate: the same state machine in integer arithmetic, ONE WAVEFRONT PER CLIP -------------------
// Under the auto gate every quantity of the frame loop is an integer (launch state ctx_max = 50, floor = 2 @B25471; amplitudes are
// u32; every update is a trunc, a sum or a copy), so the per-frame decisions run on the scalar unit:
//   floor, ctx_max, last_max, last_floor, h     u32          g, d, T      u64 (< 2^53, exact as doubles in the reference)
//   r = h(n-1)/(d-h) > 4        <=>  h(n-1) > 4(d-h)          (integers: the rounded quotient cannot reach 4 from either side)
//   d/(g-d) < .1, g-d > 0       <=>  11 d < g
//   floor > last_floor/10       <=>  floor > last_floor div 10 (the f64 quotient is >= 0.1 away from any integer it is not equal to)
//   h > last_max/100            <=>  100 h > last_max
//   trunc(last_floor/20), trunc(ctx_max/8)   integer division / shift
//   no_fm >= breaker            <=>  no_fm >= ceil(breaker)
//   T/k < 30 v                  <=>  T < 30 v k               (k < 2^20: the quotient stays >= 1/k >> one ulp of 30 v <= 2^23 away; else f64)
// d (sum of the accepted amplitudes: a cross-lane reduction) is only computed where a decision needs it: the voiced test's
// `d/(g-d) < .1` is false without looking whenever the largest accepted candidate alone has 11 mx >= g (d >= mx), which covers
// all but ~1 % of the frames of speech-like input; the start test needs it only for a frame that passed its other clauses.
// A wave alone on its SIMD issues an instruction every ~10 cycles whatever its kind, so the loop is written for instruction COUNT:
//   * what does not depend on the state is evaluated for the 64 frames of a block at once (lane = frame) and kept as 64-bit masks in
//     scalar registers — bin of the largest candidate out of the voiced range / inside the start range, 11 mx < g;
//   * the two steady states (no segment open; segment running) have a short straight-line path; every frame on which something
//     happens (start test passing its cheap clauses, first frames of a segment, ctx_max moving, a pause running out, d needed)
//     goes through the general path, which is the reference's frame body term by term;
//   * the floor law v(ctx_max) is integer arithmetic except on the few y where the f64 evaluation's last bit matters (gate_floor.hpp).
// The first GATE_STAGE candidate amplitudes of 64 frames are staged through LDS one block ahead (lane = candidate when read back; entries past
// a frame's candidate count are zeroed on the way, so `amp > floor` needs no count mask); a frame with more candidates than that (1 in 10^3) reads
// its row from global memory when it is looked at.  Headers live one per lane.  (Staging all 64 table slots of every frame cost 64 VGPRs and 16 KB
// of LDS per wave — 152 VGPRs: a gate wave then does not fit the 128 registers per SIMD that three front-end workgroups per CU leave free, and
// the gate, which every later stage of its batch waits for, had to wait for the other batches' front ends to drain.)
template <bool TRACE>       // TRACE: every frame through the general path, per-frame state trace written if p.trace (tests; WSA_DBG bit 4096)
__global__ __launch_bounds__(64) void gate_kernel_auto(GateParams p) {
    constexpr int GS = 16;                               // candidates per frame that are staged
    __shared__ uint32_t s_amp[64 * GS];                  // one block: the next one waits in registers until this one has been walked
    static_assert(CAND_CAP == 64 && GS % 4 == 0, "one LDS row per frame, one lane per candidate");
    if (p.prio) __builtin_amdgcn_s_setprio(3);          // one wave per clip, every later stage of the batch waits for it: its instructions go first
    const int lane = threadIdx.x;
    const int br_i = p.breaker >= 2147483647.0 ? 2147483647 : (int)ceil(p.breaker);
    const int maxvb = p.max_voiced_bin;
    const bool want_trace = TRACE && p.trace && !(p.dbg & 16);
    const uint64_t lt_l = lanemask_lt(lane);
    for (uint32_t clip = p.clip0 + blockIdx.x; clip < p.clip0 + p.n_clips; clip += gridDim.x) {
        const uint32_t nfr = p.n_frames[clip];
        const uint32_t foff = p.frame_off[clip];
        int32_t* seg_i = p.seg_i + (uint64_t)clip * p.seg_cap * 8;
        double* seg_d = p.seg_d + (uint64_t)clip * p.seg_cap * 2;
        int no_fm = 0, c_ci = 0, c_started = -1, gw = 0;
        uint32_t ctx_max = (uint32_t)p.ctx_max0, floor_ = (uint32_t)p.floor0, last_max = ctx_max;       // (last_floor lives on as dec20 and thr_b)
        uint32_t dec20 = floor_ / 20u, thr_b = max(10u, floor_ / 10u);     // trunc(last_floor/20); the floor decays while floor > max(10, last_floor div 10)
        uint64_t gT = 0; uint32_t gk = 0;
        int nseg = 0, span_begin = 0;
        bool overflow = false;

        auto finalize = [&](int e_arg, int f_end, int cur_frame) __attribute__((always_inline)) {      // ref @B27088
            const int len = e_arg - no_fm;
            if (!((double)len > p.min_frames && c_started >= 2)) return;
            if (nseg >= p.seg_cap) { overflow = true; return; }
            if (lane == 0) {
                int32_t* sg = seg_i + 8 * nseg;
                sg[SEG_START] = cur_frame - len; sg[SEG_LEN] = len; sg[SEG_FBEGIN] = span_begin; sg[SEG_FEND] = f_end;
                sg[SEG_CCI] = c_ci; sg[SEG_FLAG] = p.level == 3 ? 1 : 0; sg[SEG_NROWS] = 0; sg[SEG_ROW0] = 0;
                seg_d[2 * nseg] = (double)ctx_max; seg_d[2 * nseg + 1] = (double)floor_;
                if (p.span_hist) {                   // the span's place in the tracker's longest-first order
                    const uint32_t bkt = (uint32_t)(SPAN_BUCKETS - 1) - min((uint32_t)(f_end - span_begin), (uint32_t)(SPAN_BUCKETS - 1));
                    p.span_key[(uint64_t)clip * p.seg_cap + nseg] = make_uint2(bkt, atomicAdd(&p.span_hist[bkt], 1u));
                }
            }
            nseg++;
        };
        // rows (frames) lane>>2 + 16i, candidates 4(lane&3).. of a 64-frame block: 4 x 16-byte loads per lane; 16-byte pieces
        // wholly past the frame's candidate count (or past the clip) are not fetched, the tail of the last piece is zeroed
        uint4 stg[GS / 4];
        auto stage_load = [&](uint32_t blk, const uint4& hdr_of_lane) __attribute__((always_inline)) {
            const int c0 = 4 * (lane & (GS / 4 - 1));
#pragma unroll
            for (int i = 0; i < GS / 4; i++) {
                const int row = (256 / GS) * i + (lane / (GS / 4));
                const int nc = (__builtin_amdgcn_ds_bpermute(row << 2, (int)hdr_of_lane.y) >> 8) & 0xff;
                uint4 w = make_uint4(0u, 0u, 0u, 0u);
                if (blk + (uint32_t)row < nfr && c0 < nc) w = *reinterpret_cast<const uint4*>(p.rec.amp + ((uint64_t)(foff + blk + (uint32_t)row)) * CAND_CAP + c0);
                if (c0 + 1 >= nc) w.y = 0u;
                if (c0 + 2 >= nc) w.z = 0u;
                if (c0 + 3 >= nc) w.w = 0u;
                stg[i] = w;
            }
        };
        auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < GS / 4; i++) *reinterpret_cast<uint4*>(&s_amp[((256 / GS) * i + (lane / (GS / 4))) * GS + 4 * (lane & (GS / 4 - 1))]) = stg[i];
        };
        uint4 hd = make_uint4(0u, 0u, 0u, 0u), hd2 = hd;
        if (nfr > 0) {
            hd = p.rec.hdr[foff + min((uint32_t)lane, nfr - 1)];
            stage_load(0, hd);
            stage_store();
        }
        for (uint32_t blk = 0; blk < nfr; blk += 64) {
            wsync();                                            // this block's rows are in s_amp
            if (blk + 64 < nfr) { hd2 = p.rec.hdr[foff + min(blk + 64 + (uint32_t)lane, nfr - 1)]; stage_load(blk + 64, hd2); }
            // state-independent clauses of the block's frames (lane = frame), one flag word per frame
            const int mxbin_l = (int)((hd.y >> 16) & 0xffu);
            const uint32_t flags_l = ((mxbin_l < 7 || mxbin_l >= maxvb) ? 1u : 0u)                 // the voiced test's `p < 7 || p >= max_voiced_bin` for p = that bin
                                   | (11ull * (uint64_t)hd.z < (((uint64_t)(hd.y & 0xffu) << 32) | hd.x) ? 2u : 0u)     // 11 mx < g: d must be looked at
                                   | ((mxbin_l > 7 && mxbin_l < maxvb) ? 4u : 0u);              // the start test's `p > 7 && p < max_voiced_bin`
            int o_info = -1; uint32_t o_fl = 0;
            const uint32_t floor_in = floor_;
            const int nblk = (int)min(64u, nfr - blk);
            // lane j of the block's output registers (v at frame start = floor after the frame before: shifted in at the block's end)
            // (v_writelane with a scalar value AND a scalar lane number needs the lane in m0: the constant bus feeds one SGPR per instruction.
            //  m0 is named as clobbered — the compiler flags that as "reserved register", which is the point: nothing may live there across)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            auto put = [&](int j_, int info_) __attribute__((always_inline)) {
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0" : "+v"(o_info), "+v"(o_fl) : "s"(j_), "s"(info_), "s"(floor_) : "m0");
            };
#pragma clang diagnostic pop
            // amplitude of candidate `lane` of frame j_ of this block (0 past the frame's count)
            auto amp_row = [&](int j_) __attribute__((always_inline)) -> uint32_t {
                uint32_t a = lane < GS ? s_amp[j_ * GS + lane] : 0u;
                const int nc = (read_lane_i32((int)hd.y, j_) >> 8) & 0xff;
                if (nc > GS) a = lane < nc ? p.rec.amp[((uint64_t)(foff + blk) + (uint32_t)j_) * CAND_CAP + (uint32_t)lane] : 0u;      // (uniform branch; rare)
                return a;
            };
            auto count_accepted = [&](int j_) __attribute__((always_inline)) -> int {      // ref @B25827 `e[l] > v`
                return __popcll(__ballot(amp_row(j_) > floor_));
            };
            for (int j = 0; j < nblk; j++) {
                if (!TRACE) {
                    // the two steady states as tight loops; they leave at the first frame on which something happens
                    if (c_started >= 2) {
                        // segment running: every frame calls the gate; the loop leaves for the general path when the pause runs out, d is
                        // needed, or the gate's T / k test resets the segment (before anything is modified: the general path replays the frame).
                        // n only matters for `n > 3` of the d clause: the largest candidate is accepted when it exceeds 2 floor, so n > 0 there
                        for (; j < nblk; j++) {
                            // The frames on which NOTHING but the slow floor decay happens (no ctx_max event, d not needed, pause not run out), all at once with
                            // lane = frame: between two events the floor follows a closed form — it loses dec20 on every frame whose gate counter exceeds 20,
                            // S times at most (until it is down at thr_b) —, so every frame of the block knows the floor it would see if nothing happened before
                            // it, tests the loop's exit conditions against that, and the first lane that raises one ends the run (what the lanes behind it
                            // computed is void).  ~70 instructions per RUN (a run is ~40 frames of speech-like input) where the scalar loop spent 31 per frame.
                            {
                                const int k = lane - j;                                             // frames since the start of the run
                                const int d0 = max(20 - gw, 0);                                     // the run's first frame on which the floor may decay (w > 20 after the increment)
                                const uint32_t room = floor_ > thr_b ? floor_ - thr_b : 0u;
                                const int S = __popcll(__ballot((uint64_t)(uint32_t)lane * (uint64_t)dec20 < (uint64_t)room));      // decay steps left: floor - s dec20 > thr_b (dec20 = 0: never ends, changes nothing)
                                const uint32_t lo10 = S > 0 ? 10u : 0u;                             // (`if (floor < 10) floor = 10` belongs to a decay step)
                                const int sk = min(max(k - d0, 0), S), sk1 = min(max(k + 1 - d0, 0), S);
                                const uint32_t fk = max(floor_ - (uint32_t)sk * dec20, lo10), fk1 = max(floor_ - (uint32_t)sk1 * dec20, lo10);      // floor before / after frame k
                                const uint32_t mxl = hd.z, v2 = fk << 1;
                                const bool strong = mxl > v2, voiced = strong && !(flags_l & 1u);
                                const uint64_t run_m = ~0ull << j;
                                const uint64_t prev = __ballot(voiced) & run_m & lt_l;              // voiced frames of the run before this one
                                const int nofm_b = prev ? lane - 64 + __clzll((long long)prev) : no_fm + k;      // no_fm_segs in front of frame k
                                const bool ex = max(mxl, v2) > ctx_max                              // h > ctx_max: the gate's ctx_max branch
                                             || (strong && gw + k >= 40)                            // w > 40 after the increment and h > 2 floor: ctx_max branch
                                             || (voiced && (flags_l & 2u))                          // 11 mx < g: n and perhaps d must be looked at
                                             || (!voiced && nofm_b >= br_i - 1);                    // the pause runs out: finalize
                                const uint64_t exm = __ballot(ex) & run_m;
                                const int E = min(exm ? __ffsll((long long)exm) - 1 : 64, nblk);
                                if (E > j) {
                                    if (k >= 0 && lane < E) { o_info = voiced ? c_ci + k : -1; o_fl = fk1; }
                                    floor_ = (uint32_t)read_lane_i32((int)fk1, E - 1);
                                    no_fm = read_lane_i32(voiced ? 0 : nofm_b + 1, E - 1);
                                    c_ci += E - j; gw += E - j; j = E;
                                }
                                if (j >= nblk) break;
                            }
                            const uint32_t mx_ = (uint32_t)read_lane_i32((int)hd.z, j), fl_ = (uint32_t)read_lane_i32((int)flags_l, j);
                            const bool strong_ = mx_ > 2u * floor_;
                            const uint32_t h_ = max(mx_, 2u * floor_);
                            bool unv_;
                            if (strong_ && !(fl_ & 1u)) {
                                if ((fl_ & 2u) && count_accepted(j) > 3) break;
                                unv_ = false;
                            } else {
                                if (no_fm + 1 >= br_i) break;
                                unv_ = true;
                            }
                            if (h_ > ctx_max || (gw >= 40 && strong_)) {             // ref @B28506: ctx_max moves, the floor is set anew
                                uint32_t nctx = ctx_max, nlast = last_max; int ngw = gw + 1;
                                if (h_ >= ctx_max) { ngw = 0; nlast = nctx = h_; }
                                else if (100ull * (uint64_t)h_ > (uint64_t)last_max) { nctx -= nctx >> 3; ngw = 35; }
                                const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)floor_law(nctx));
                                if (gk >= (1u << 20) || (gk > 0 && gT < 30ull * (uint64_t)nv * (uint64_t)gk)) break;
                                ctx_max = nctx; last_max = nlast; gw = ngw;
                                floor_ = nv; dec20 = nv / 20u; thr_b = max(10u, nv / 10u);
                                gT += nctx; gk += 1;
                            } else {
                                gw++;
                                if (gw > 20 && floor_ > thr_b) floor_ = max(floor_ - dec20, 10u);
                            }
                            put(j, unv_ ? -1 : c_ci);
                            no_fm = unv_ ? no_fm + 1 : 0;
                            c_ci++;
                        }
                    } else if (c_started < 0) {
                        // no segment open: the start test fails on one of its cheap clauses (p = 0 unless the largest candidate exceeds 2 floor)
                        for (; j < nblk; j++) {
                            {   // (as above, lane = frame: the run ends in front of the first frame whose start test passes `largest candidate above 2 floor, its bin inside the start range`)
                                const uint64_t exm = __ballot(hd.z > 2u * floor_ && (flags_l & 4u)) & (~0ull << j);        // n > 4 must be looked at
                                const int E = min(exm ? __ffsll((long long)exm) - 1 : 64, nblk);
                                if (E > j) {
                                    if (lane >= j && lane < E) { o_info = -1; o_fl = floor_; }
                                    no_fm += E - j; c_ci += E - j; j = E;
                                }
                                if (j >= nblk) break;
                            }
                            const uint32_t mx_ = (uint32_t)read_lane_i32((int)hd.z, j), fl_ = (uint32_t)read_lane_i32((int)flags_l, j);
                            if (mx_ > 2u * floor_ && (fl_ & 4u) && count_accepted(j) > 4) break;
                            put(j, -1);
                            no_fm++; c_ci++;
                        }
                    }
                    if (j >= nblk) break;
                }
                // ---------------- general path: the reference's frame body
                const uint32_t f = blk + (uint32_t)j;
                const uint32_t amp = amp_row(j);
                const uint32_t mx = (uint32_t)read_lane_i32((int)hd.z, j);
                const uint32_t v = floor_;
                // ---- accept candidates (ref @B25827: `e[l] > v`): n; h / p from the header's largest candidate (accepted whenever it exceeds h = 2v >= v)
                const int n = __popcll(__ballot(amp > v));
                const bool strong = mx > 2u * v;                 // then that candidate is accepted: n > 0
                const uint32_t h = strong ? mx : 2u * v;
                int info = -1;
                {
                    // ---------------- general path: the reference's frame body
                    const uint32_t hx = (uint32_t)read_lane_i32((int)hd.x, j), hy = (uint32_t)read_lane_i32((int)hd.y, j);
                    const uint64_t g = ((uint64_t)(hy & 0xffu) << 32) | hx;
                    const int pbin = strong ? (int)((hy >> 16) & 0xffu) : 0;
                    const int t_idx = c_ci;                                  // captured before the start test (quirk 1)
                    uint64_t d = 0; bool have_d = false;
                    auto exact_d = [&]() __attribute__((always_inline)) {
                        if (have_d) return;
                        const uint32_t a = amp > v ? amp : 0u;
                        d = ((uint64_t)wave_sum_u32(a >> 20) << 20) + (uint64_t)wave_sum_u32(a & 0xfffffu);
                        have_d = true;
                    };
                    if (want_trace) exact_d();
                    // ---- start test (ref @B26527)
                    bool reset_before_acc = false;
                    if (c_started < 0) {
                        bool start = false;
                        if (n > 4 && pbin > 7 && pbin < maxvb) { exact_d(); start = d > (uint64_t)h && (uint64_t)h * (uint64_t)(n - 1) > 4ull * (d - (uint64_t)h); }
                        if (start) { c_ci = 0; c_started = 0; no_fm = 0; reset_before_acc = true; span_begin = (int)f; }        // L(0)
                        else no_fm++;
                    }
                    bool do_reset = false, gate_reset = false;
                    auto noise_gate = [&]() __attribute__((always_inline)) {                       // ref @B28506, argument h
                        gw++;
                        if (h > ctx_max || (gw > 40 && h > 2u * floor_)) {
                            if (h >= ctx_max) { gw = 0; last_max = ctx_max = h; }
                            else if (100ull * (uint64_t)h > (uint64_t)last_max) { ctx_max -= ctx_max >> 3; gw = 35; }
                            const uint32_t nv = (uint32_t)__builtin_amdgcn_readfirstlane((int)floor_law(ctx_max));
                            floor_ = nv; dec20 = nv / 20u; thr_b = max(10u, nv / 10u);
                            const bool below = gk < (1u << 20) ? gT < 30ull * (uint64_t)nv * (uint64_t)gk : (double)gT / (double)gk < 30.0 * (double)nv;
                            if (gk > 0 && below) { c_ci = 0; c_started = 0; no_fm = 0; gate_reset = true; gk = 0; gT = 0; }   // L(0)
                            gT += ctx_max; gk += 1;
                        } else if (floor_ > thr_b && gw > 20) {
                            floor_ = max(floor_ - dec20, 10u);
                        }
                    };
                    if (c_started >= 0) {                                    // ref @B26646
                        bool unv = n == 0 || pbin < 7 || pbin >= maxvb;
                        if (!unv && n > 3 && 11ull * (uint64_t)mx < g) { exact_d(); unv = 11ull * d < g; }      // else d >= mx decides: 11 d >= g
                        if (unv) {
                            no_fm++;
                            if (c_started < 2) c_started--;
                            else if (no_fm >= br_i) { finalize(c_ci + 1, (int)f + 1, (int)f + 1); do_reset = true; }
                            else { noise_gate(); if (gate_reset) span_begin = (int)f + 1; }
                        } else {
                            noise_gate();
                            if (gate_reset) { reset_before_acc = true; span_begin = (int)f; }
                            info = t_idx | (reset_before_acc ? (1 << 30) : 0);      // accumulate_fm(e, peaks, t_idx, g, floor_)
                            if (c_started < 2) c_started++; else no_fm = 0;
                        }
                    }
                    if (want_trace && lane == 0) {
                        double* tr = p.trace + ((uint64_t)foff + f) * 12;
                        tr[0] = c_ci; tr[1] = c_started; tr[2] = no_fm; tr[3] = (double)ctx_max; tr[4] = (double)floor_; tr[5] = n; tr[6] = pbin;
                        tr[7] = (double)h; tr[8] = (double)d; tr[9] = (double)g; tr[10] = 0; tr[11] = 0;
                    }
                    c_ci++;
                    if (do_reset) { c_ci = 0; c_started = -1; no_fm = 0; span_begin = (int)f + 1; }   // L(-1) in the Promise .then (quirk 8)
                }
                put(j, info);
            }
            const uint32_t o_sh = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, (int)o_fl);   // all lanes active: lane L receives lane L-1
            if (lane < nblk) {
                const uint32_t fi = foff + blk + (uint32_t)lane;
                const uint32_t o_v = lane == 0 ? floor_in : o_sh;
                p.fr_info[fi] = o_info; p.fr_v[fi] = (double)o_v; p.fr_fl[fi] = (double)o_fl;
            }
            wsync();
            if (blk + 64 < nfr) stage_store();
            hd = hd2;
        }
        // ---- end of input: segment_truncate (ref @B30757) -> O(c_ci) -> L(1)
        finalize(c_ci, (int)nfr, (int)nfr);
        if (lane == 0) {
            p.seg_count[clip] = (uint32_t)nseg; p.clip_rows[clip] = 0;
            if (nseg > 0) atomicMax(&p.counters[0], (uint32_t)nseg);
            if (overflow) atomicOr(&p.shared[1], 1u);
        }
        wsync();
    }
}

// streaming: applied before the step's kernels — a fresh stream (ctl bit 0) starts from the launch state
// (ref reset_segmentation @B24629) with an empty callback history
__global__ void stream_prepare_kernel(double* state, int32_t* carry, int32_t* tr_state, const uint32_t* ctl, uint32_t n, double ctx_max0, double floor0) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !(ctl[s] & 1u)) return;
    double* st = state + (uint64_t)s * GATE_STATE;
    st[0] = 0; st[1] = 0; st[2] = 0; st[3] = -1; st[4] = ctx_max0; st[5] = floor0; st[6] = ctx_max0; st[7] = floor0;
    st[8] = 0; st[9] = 0; st[10] = 0; st[11] = 0; st[12] = 0;
    for (int i = 0; i < CARRY_WORDS; i++) carry[(uint64_t)s * CARRY_WORDS + i] = 0;
    if (tr_state) {          // the tracker starts empty; its generation counter (word 6) keeps running so that stale filing slots never match
        int32_t* t = tr_state + (uint64_t)s * TR_STATE_WORDS;
        t[0] = 0; t[1] = 0; t[2] = 0; t[3] = -1; t[4] = 0; t[5] = -2; t[8] = 0; t[9] = 0; t[10] = 0; t[11] = 0;
    }
}

void launch_gate(const GateParams& p, hipStream_t s) {
    if (p.n_clips == 0) return;
    // WSA_DBG bit 2048: the general (f64, lane = candidate) kernel also under the auto gate
    if (p.auto_gate && p.strided && !(p.dbg & 2048)) {
        if ((p.trace && !(p.dbg & 16)) || (p.dbg & 4096)) hipLaunchKernelGGL(gate_kernel_auto<true>, dim3(p.n_clips), dim3(64), 0, s, p);
        else hipLaunchKernelGGL(gate_kernel_auto<false>, dim3(p.n_clips), dim3(64), 0, s, p);
    }
    else hipLaunchKernelGGL(gate_kernel_t<false>, dim3(p.n_clips), dim3(64), 0, s, p);
}

void launch_stream_prepare(double* state, int32_t* carry, int32_t* tr_state, const uint32_t* ctl, uint32_t n, double ctx_max0, double floor0, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(stream_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, s, state, carry, tr_state, ctl, n, ctx_max0, floor0);
}

void launch_gate_stream(const GateParams& p, hipStream_t s) {
    if (p.n_clips == 0) return;
    hipLaunchKernelGGL(gate_kernel_t<true>, dim3(p.n_clips), dim3(64), 0, s, p);
}

}  // namespace wsa

// frontend.hip — K1: batched PCM -> Hann window -> 1024-point real FFT -> power -> mel bands ->
// emphasis / gain -> Uint32 frame, for gfx950 (MI355X).
//
// Stands in for the reference's "spectrum-processor" AudioWorklet (source not in the reference
// tree: fetched from unpkg at run time, ref dist/main.js:2 @B6480; configured @B6726; its output,
// one Uint32Array(spec_bands) per frame, is consumed @B8568).  Arithmetic = specification FE-1
// (DESIGN.md): bit-exact with oracle/frontend.c.  Compiled with -ffp-contract=off: only the
// explicit __builtin_fmaf calls fuse.
//
// Mapping: ONE WAVEFRONT PER FRAME, everything in registers except two LDS transposes.
//   N2 = 512 packed complex points = 64 lanes x 8 points.  DIF radices [8, 8, 8]:
//   pass 1  lane m holds z[64a + m], a = 0..7           (only a < AZ are non-zero: window <= 128 AZ samples)
//   X1      LDS transpose: lane (a', c) gets b = 0..7 of sub-FFT a'      (row stride 72: conflict-free)
//   pass 2  radix 8 over b, twiddle W_64^{c b'}
//   X2      LDS transpose inside 8-lane groups: lane (a', b') gets c = 0..7  (row stride 9)
//   pass 3  radix 8 over c  ->  lane holds Z[k0 + 64 c'], k0 = a' + 8 b'
//   split   real-FFT recombination with the partner lane (k0 <-> 64 - k0) through ds_bpermute
//   mel     4|X|^2 to LDS, each lane sums its bands (fmaf chain, ascending bin), u32 store (coalesced)
// Twiddles, the window and the split factors are loop-invariant per lane and live in VGPRs for all
// frames a wave processes.  PCM is read exactly once with 512-byte-per-instruction coalesced loads.
#include "fe_common.hpp"
#include <algorithm>
#include <cstdlib>

namespace wsa {

// NR = rows of 64 bins the split produces (kmax / 64 + 1 <= NR), MW = mel taps per band kept in registers (wider bands take the LDS loop),
// T1L = inter-pass twiddles W_512^{m a'} read from LDS instead of 14 registers: <4, 5, 8, true> is the BASELINE geometry (16 kHz, 128 mel bands
// up to 4 kHz) at 4 waves per SIMD
// MWL = taps kept for the lane's LOWER band (bands 0..63 of a mel bank are the narrow ones: 4 taps at the baseline geometry, 8 for the upper half)
// AF = leading 64-point blocks of packed input that lie inside the window for every lane (win >= 128 AF): their samples need no select
// where power bin k of the 1024-point kernel sits in its LDS row: the split's lanes hold k0 = (lane >> 3) + 8 (lane & 7), so lanes l and l + 4 of a half-wave
// would store to the same bank (k0 mod 32); swapping bit 2 where bit 5 is set spreads the 32 lanes of a half-wave over the 32 banks.  The readers' tap
// addresses are loop-invariant registers either way.
__device__ __forceinline__ int psw(int k) { return k ^ ((k >> 3) & 4); }
// (the lean instantiations are held to 128 VGPRs = 4 waves per SIMD: what does not fit are three addresses of the general split path, which the
//  baseline geometry never runs)
#ifndef WSA_FE_X1REG
#define WSA_FE_X1REG 1
#endif
// ---- the first transpose without LDS (baseline instantiation).  v[b] of lane (hi3 = h, lo3) <- v[h] of lane (hi3 = b, lo3): the register index and lane bits
// 3 .. 5 change places, one bit pair per stage.  Lane bit 5 / 4 <-> register bit 2 / 1 are gfx950's v_permlane32_swap / v_permlane16_swap (the upper half of
// one register against the lower half of another; the odd 16-lane rows of one against the even rows of another): one instruction per register pair, no
// selects.  Lane bit 3 <-> register bit 0 has no swap instruction: row_ror:8 (lane ^ 8 inside a row) as the DPP operand of a v_cndmask per register.
__device__ __forceinline__ void fe_swap32(v2f& a, v2f& b) {
    const auto rx = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, (float)a.x), __builtin_bit_cast(unsigned, (float)b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, (float)a.y), __builtin_bit_cast(unsigned, (float)b.y), false, false);
    a.x = __builtin_bit_cast(float, (unsigned)rx[0]); b.x = __builtin_bit_cast(float, (unsigned)rx[1]);
    a.y = __builtin_bit_cast(float, (unsigned)ry[0]); b.y = __builtin_bit_cast(float, (unsigned)ry[1]);
}
__device__ __forceinline__ void fe_swap16(v2f& a, v2f& b) {
    const auto rx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, (float)a.x), __builtin_bit_cast(unsigned, (float)b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, (float)a.y), __builtin_bit_cast(unsigned, (float)b.y), false, false);
    a.x = __builtin_bit_cast(float, (unsigned)rx[0]); b.x = __builtin_bit_cast(float, (unsigned)rx[1]);
    a.y = __builtin_bit_cast(float, (unsigned)ry[0]); b.y = __builtin_bit_cast(float, (unsigned)ry[1]);
}
// lanes 8 .. 15 of every row (bank mask 0xC) take src from lane ^ 8, the others keep old — and the mirror image (bank mask 0x3): v_mov_b32_dpp row_ror:8, no select
__device__ __forceinline__ float fe_ror8_hi(float old, float src) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0xc, false)); }
__device__ __forceinline__ float fe_ror8_lo(float old, float src) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0x3, false)); }
__device__ __forceinline__ void fe_transpose_hi3(v2f (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 4; k++) fe_swap32(v[k], v[k + 4]);
#pragma unroll
    for (int k = 0; k < 8; k++) if (!(k & 2)) fe_swap16(v[k], v[k + 2]);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const float ax = v[k].x, ay = v[k].y, bx = v[k + 1].x, by = v[k + 1].y;
        v[k].x = fe_ror8_hi(ax, bx); v[k].y = fe_ror8_hi(ay, by);
        v[k + 1].x = fe_ror8_lo(bx, ax); v[k + 1].y = fe_ror8_lo(by, ay);
    }
}

template <int AZ, int NR, int MW, bool T1L, int MWL = MW, int AF = 0>
__global__ __launch_bounds__(256, (T1L && MWL == 4) ? 4 : 1) void fe_kernel_r8(FeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (the wave number through v_readfirstlane: the compiler then knows the frame counter, the frame's addresses and the loop tests are uniform — scalar code)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ uint32_t s_next[2];                         // the persistent launch's hand-off of the next chunk number (two slots: one barrier per chunk)
    // ---- LDS carve-up: [mel_w | mel_k0 | mel_cnt | mel_off | emph] shared, then per wave X + P
    float* s_melw = reinterpret_cast<float*>(smem);
    int* s_k0 = reinterpret_cast<int*>(s_melw + ((p.mel_total + 3) & ~3));
    int* s_cnt = s_k0 + p.bands;
    int* s_off = s_cnt + p.bands;
    float* s_emph = reinterpret_cast<float*>(s_off + p.bands);
    const int shared_words = ((p.mel_total + 3) & ~3) + 4 * p.bands;
    // (power rows: whole rows of 64 bins — the split stores every lane's row entry, no test against kmax)
    v2f* X = reinterpret_cast<v2f*>(smem + (size_t)((shared_words + 3) & ~3) * 4) + (size_t)wave * XBUF;
    // the power rows share the wave's exchange buffer: they are written behind the last read of the second transpose and read before the next frame's first store
    // (5 KB less per workgroup: beside two of them a CU then holds five 20 KB peak-scan waves of the other batches instead of four)
    float* P = reinterpret_cast<float*>(X);
    static_assert(XBUF * 8 >= 64 * 9 * 4, "the power rows fit the exchange buffer");
    v2f* s_tw1 = reinterpret_cast<v2f*>(reinterpret_cast<float2*>(smem + (size_t)((shared_words + 3) & ~3) * 4) + 4 * XBUF);   // [7][64]
    if (T1L) for (int i = threadIdx.x; i < 7 * 64; i += 256) s_tw1[i] = to_v2f(p.tw_n2[(i & 63) * ((i >> 6) + 1)]);
    // (the baseline geometry's pipelined loop keeps the power taps of the frame before in registers across the head of the next frame: the split's five twiddles
    //  W_1024^k of the lane make room — read from this table beside the partner values, same round trip)
    v2f* s_tws = s_tw1 + 7 * 64;                                                                                               // [5][64]
    if (AF == 3 && NR == 5 && T1L) for (int i = threadIdx.x; i < 5 * 64; i += 256) { const int l = i & 63, k = (l >> 3) + 8 * (l & 7) + 64 * (i >> 6); s_tws[i] = to_v2f(k <= p.kmax ? p.tw_nfft[k] : make_float2(0.f, 0.f)); }

    for (int i = threadIdx.x; i < p.mel_total; i += 256) s_melw[i] = p.mel_w[i];
    for (int i = threadIdx.x; i < p.bands; i += 256) {
        s_emph[i] = p.emph[i];
        if (p.spec_type == 1) { s_k0[i] = p.mel_k0[i]; s_cnt[i] = p.mel_cnt[i]; s_off[i] = p.mel_off[i]; }
    }
    // chunk = 4 x frames_per_wave frames of one clip.  Persistent launch (p.queue): fewer workgroups than chunks, each takes chunk after chunk from a
    // device counter — the number of the chunk after this one is requested before the chunk is worked on, so its latency is never waited for — and the per-lane
    // constants below are set up once per workgroup instead of once per chunk.  Otherwise chunk = blockIdx.x.
    uint32_t chunk = blockIdx.x, nxt = 0;
    if (p.queue) {
        if (threadIdx.x == 0) { chunk = atomicAdd(p.queue, 1u); s_next[0] = chunk; }
    }
    __syncthreads();
    if (p.queue) chunk = s_next[0];
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    int phase = 0;

    // ---- loop-invariant per-lane constants (registers)
    v2f tw1[8], tw2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) { if (!T1L) tw1[k] = to_v2f(p.tw_n2[lane * k]); tw2[k] = to_v2f(p.tw_64[(lane & 7) * k]); }
    v2f wn[AZ];
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        wn[a].x = n < p.win ? p.window[n] : 0.0f;
        wn[a].y = n + 1 < p.win ? p.window[n + 1] : 0.0f;
    }
    v2f ss; ss.x = 0.70710678118654752440f; ss.y = 0.70710678118654752440f;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int k0 = hi3 + 8 * lo3;                         // this lane ends up holding Z[k0 + 64 c']
    const int k0p = (64 - k0) & 63;
    const int partner = ((k0p & 7) << 3) | (k0p >> 3);    // lane holding Z[k0p + 64 c']
    // BASE: the baseline geometry's own instantiation (AF == 3: 16 kHz, 400-sample window, 128 mel bands up to 4 kHz).  launch_frontend selects it only when
    // the split has exactly five rows, all 128 bands exist and every band's taps fit the registers, so the loop-invariant tests of the frame loop
    // (rows, mel_fast, all_bands: a scalar compare + branch each, ~20 instructions per frame) are compile-time constants there
    constexpr bool BASE = AF == 3 && NR == 5 && T1L;
    const int nrow = BASE ? 5 : p.kmax / 64 + 1;          // rows c' with some k <= kmax (<= NR)
    v2f tws[NR];
#pragma unroll
    for (int c = 0; c < NR; c++) {
        const int k = k0 + 64 * c;
        tws[c] = to_v2f((k <= p.kmax) ? p.tw_nfft[k] : make_float2(0.f, 0.f));       // (dead in the BASE instantiation: its loop reads s_tws)
    }

    // mel taps of this lane's two bands (m = lane, lane + 64): loop invariant, zero padded.  A padded
    // tap contributes fmaf(0, P, e) = e exactly, so the fixed-length chain equals the FE-1 chain.
    float mw[2][MW]; int mk[2], mn[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int m = lane + 64 * q;
        mk[q] = 0; mn[q] = 0;
#pragma unroll
        for (int j = 0; j < MW; j++) mw[q][j] = 0.f;
        if (p.spec_type == 1 && m < p.bands) {
            mk[q] = s_k0[m]; mn[q] = s_cnt[m];
#pragma unroll
            for (int j = 0; j < MW; j++) if (j < mn[q]) mw[q][j] = s_melw[s_off[m] + j];
        }
    }
    const bool mel_fast = BASE || (p.spec_type == 1 && p.bands <= 128 && __all(mn[0] <= MWL && mn[1] <= MW));
    // emphasis factors of the lane's two bands: loop invariant (two registers instead of two LDS reads per frame)
    float emph_r[2];
#pragma unroll
    for (int q = 0; q < 2; q++) { const int m = lane + 64 * q; emph_r[q] = s_emph[m < p.bands ? m : 0]; }
    const int pmax = p.kmax;                                    // padded taps read a valid P slot
    const bool all_bands = BASE || p.bands == 128;              // both of a lane's bands exist: the stores need no lane test

    // PCM of frame f+1 is requested before frame f is transformed (lane m takes complex points 64a + m)
    // branch-free: every lane reads one 8-byte pair inside the window (index clamped to win - 2), the lane that
    // owns the last sample of an odd window takes the pair's second half; samples at n >= win are replaced by 0
    uint32_t ld_idx[AZ]; uint64_t ld_v0[AZ], ld_v1[AZ], ld_odd[AZ];     // lane masks (scalar registers); the lane's BYTE offset as an unsigned 32-bit word: uniform base + 32-bit lane offset is one address mode, no 64-bit vector arithmetic per load
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        ld_idx[a] = (uint32_t)max(min(n, p.win - 2), 0) * 4u;
        ld_v0[a] = __ballot(n < p.win); ld_v1[a] = __ballot(n + 1 < p.win); ld_odd[a] = __ballot(n == p.win - 1);
    }
    // the loaded pairs stay untouched until the next iteration consumes them (anything computed from them here
    // would make the loop wait for the loads at once and lose the prefetch)
    for (; chunk < p.n_chunks;) {
    // (the quotient comes out of the vector unit — integer division is a float-reciprocal sequence there — and everything derived from it would stay in vector
    //  registers: the clip's frame count a vector load, the frame's output address a v_readfirstlane pair with its wait states in front of every store)
    const uint32_t clip = __builtin_amdgcn_readfirstlane(chunk / p.chunks_per_clip), cx = chunk - clip * p.chunks_per_clip;
    const uint32_t nfr = p.n_frames[clip];
    const uint32_t f_begin = (cx * 4u + (uint32_t)wave) * (uint32_t)p.frames_per_wave;
    uint32_t f_end = f_begin + (uint32_t)p.frames_per_wave;
    if (f_end > nfr) f_end = nfr;
    // (both offsets through v_readfirstlane: uniform values the compiler then keeps in scalar registers, so that a frame's addresses are scalar arithmetic and the
    //  loads / stores take the "scalar base + 32-bit lane offset" form; the pointers themselves stay derived from the kernel arguments — global address space)
    const float* clip_pcm = p.pcm + uniform_u64((uint64_t)clip * p.clip_stride + (p.pcm_off ? p.pcm_off[clip] : 0u));
    uint32_t* out_base = p.spec + uniform_u64((uint64_t)p.frame_off[clip] * (uint32_t)p.bands);
    // buffer addressing: the wave's first frame is the base of a raw buffer descriptor (four scalar registers, set up per chunk), a frame is a 32-bit SCALAR byte
    // offset from it and a lane a 32-bit vector byte offset — no 64-bit vector address arithmetic per load / store (five v_lshl_add_u64 and eight registers before)
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(clip_pcm + (uint64_t)f_begin * (uint32_t)p.hop), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out_base + (uint64_t)f_begin * (uint32_t)p.bands, 0, 0x7fffffff, 0x00020000);
    auto load_pcm = [&](uint32_t f, v2f (&x)[AZ]) __attribute__((always_inline)) {
        const uint32_t so = (f - f_begin) * (uint32_t)p.hop * 4u;
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r_in, ld_idx[a], so, 0);
            x[a] = __builtin_bit_cast(v2f, q);          // (the whole vector: element-wise bit casts of q.x / q.y compile to ONE dword load copied into both halves on this compiler)
        }
    };
    // (the next chunk's number is requested right behind the first frame's samples: the two round trips overlap, and it is only looked at behind the chunk)
    if (f_begin < f_end) {
    v2f xin[AZ];
    load_pcm(f_begin, xin);
    if (p.queue && threadIdx.x == 0) nxt = atomicAdd(p.queue, 1u);

    if constexpr (BASE) {
        // ---- the baseline geometry's loop, software-pipelined by one stage.  A wave issues nothing while it waits for an LDS read, and a frame has five such
        //      round trips one after the other (two transposes, the split's partner values, the power taps, the twiddles); here the END of frame f — power taps
        //      -> mel sums -> store — is taken apart: its reads are requested right behind the power rows' stores, then the HEAD of frame f + 1 (window, first
        //      radix-8 pass, twiddles) runs while they are on their way, the first transpose of f + 1 is issued, and only then the mel sums of f are formed —
        //      while that transpose is on ITS way.  Two of the round trips are covered by the other frame's arithmetic.  The power rows share the exchange buffer
        //      (P aliases X): the taps of f are READ before the transpose of f + 1 is WRITTEN, and a wave's LDS instructions execute in order.  Same operations
        //      on the same operands as the plain loop below: bit-identical frames.
        v2f v[8]; float pv[2][MW];
        auto head = [&](uint32_t f) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < 8; a++) { v[a].x = 0.f; v[a].y = 0.f; }
#pragma unroll
            for (int a = 0; a < AZ; a++) {
                v2f x = xin[a];
                if (a >= AF) { x.x = sel_mask(0.f, sel_mask(xin[a].x, xin[a].y, ld_odd[a]), ld_v0[a]); x.y = sel_mask(0.f, xin[a].y, ld_v1[a]); }
                v[a] = pk_mul(x, wn[a]);
            }
            load_pcm(min(f + 1, f_end - 1), xin);          // (the last frame is requested again: no branch)
#pragma unroll
            for (int k = 1; k < 8; k++) tw1[k] = s_tw1[(k - 1) * 64 + lane];
            radix8_pk<AZ>(v, ss);
            pk_cmul7(v, tw1);
        };
        auto tail = [&](uint32_t fp, int q) __attribute__((always_inline)) {
            const uint32_t so_out = (fp - f_begin) * (uint32_t)p.bands * 4u;
            {
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < MW; j++) if (q == 1 || j < MWL) e = __builtin_fmaf(mw[q][j], pv[q][j], e);
                e = e * emph_r[q];
                e = e * p.gain;
                __builtin_amdgcn_raw_buffer_store_b32(to_u32(e), r_out, (uint32_t)(lane + 64 * q) * 4u, so_out, 0);
            }
        };
        head(f_begin);
        for (uint32_t f = f_begin; f < f_end; f++) {
            // X1 of frame f
#if WSA_FE_X1REG
            fe_transpose_hi3(v);
#else
#pragma unroll
            for (int k = 0; k < 8; k++) X[k * XROW + lane] = v[k];
            wave_lds_sync();
#pragma unroll
            for (int b = 0; b < 8; b++) v[b] = X[hi3 * XROW + 8 * b + lo3];
            wave_lds_sync();
#endif
            if (f > f_begin) tail(f - 1, 1);                // the upper band's mel sum and store of the frame before, behind this frame's first transpose
            radix8_pk<8>(v, ss);
            pk_cmul7(v, tw2);
#pragma unroll
            for (int k = 0; k < 8; k++) X[hi3 * XROW + k * 9 + lo3] = v[k];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = X[hi3 * XROW + lo3 * 9 + c];
            wave_lds_sync();
            if (f > f_begin) tail(f - 1, 0);                // ... the lower band's behind the second
            radix8_pk<8>(v, ss);
            {
                v2f za[5], zb[5], tw5[5]; float pw[5];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const v2f src = v[7 - c];
                    zb[c].x = __shfl(src.x, partner, 64);
                    zb[c].y = __shfl(src.y, partner, 64);
                    if (k0 == 0) zb[c] = v[(8 - c) & 7];
                    za[c] = v[c]; tw5[c] = s_tws[c * 64 + lane];
                }
                pk_split5(za, zb, tw5, pw);
#pragma unroll
                for (int c = 0; c < 5; c++) P[psw(k0) + 64 * c] = pw[c];
            }
            wave_lds_sync();
            // the power taps of frame f are requested ...
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int j = 0; j < MW; j++) if (q == 1 || j < MWL) { const int k = mk[q] + j; pv[q][j] = P[psw(k <= pmax ? k : pmax)]; }
            wave_lds_sync();
            // ... and the head of frame f + 1 runs while they arrive (behind the last frame: the same frame's samples once more, nobody looks at the result)
            head(f + 1);
        }
        tail(f_end - 1, 1); tail(f_end - 1, 0);
    } else
    for (uint32_t f = f_begin; f < f_end; f++) {
        // ---- window (F1-F3)
        v2f v[8];
#pragma unroll
        for (int a = 0; a < 8; a++) { v[a].x = 0.f; v[a].y = 0.f; }
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            v2f x = xin[a];                                // samples at n >= win are 0; an odd window's last sample sits in .y
            if (a >= AF) { x.x = sel_mask(0.f, sel_mask(xin[a].x, xin[a].y, ld_odd[a]), ld_v0[a]); x.y = sel_mask(0.f, xin[a].y, ld_v1[a]); }
            v[a] = pk_mul(x, wn[a]);
        }
        load_pcm(min(f + 1, f_end - 1), xin);              // (the last frame is requested twice: no branch)
        // ---- pass 1: radix 8 over a, twiddle W_512^{m a'}
        if (T1L) {
#pragma unroll
            for (int k = 1; k < 8; k++) tw1[k] = s_tw1[(k - 1) * 64 + lane];
        }
        radix8_pk<AZ>(v, ss);
        pk_cmul7(v, tw1);
        // ---- X1: [a'][m] -> lane (a', c) reads b = 0..7
#pragma unroll
        for (int k = 0; k < 8; k++) X[k * XROW + lane] = v[k];
        wave_lds_sync();
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = X[hi3 * XROW + 8 * b + lo3];
        wave_lds_sync();
        // ---- pass 2: radix 8 over b, twiddle W_64^{c b'}
        radix8_pk<8>(v, ss);
        pk_cmul7(v, tw2);
        // ---- X2: [a'][b'][c] (row stride 9) -> lane (a', b') reads c = 0..7
#pragma unroll
        for (int k = 0; k < 8; k++) X[hi3 * XROW + k * 9 + lo3] = v[k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = X[hi3 * XROW + lo3 * 9 + c];
        wave_lds_sync();
        // ---- pass 3: radix 8 over c -> v[c'] = Z[k0 + 64 c']
        radix8_pk<8>(v, ss);
        // ---- real-FFT split + 4x power (F4): X[k] from Z[k] and conj(Z[512 - k])
        if (NR == 5 && nrow == 5) {
            // the five rows of the baseline geometry in one block (pk_split5): partner values first, then the five chains interleaved
            v2f za[5], zb[5], tw5[5]; float pw[5];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const v2f src = v[7 - c];
                zb[c].x = __shfl(src.x, partner, 64);
                zb[c].y = __shfl(src.y, partner, 64);
                if (k0 == 0) zb[c] = v[(8 - c) & 7];
                za[c] = v[c]; tw5[c] = tws[c < NR ? c : 0];
            }
            pk_split5(za, zb, tw5, pw);
#pragma unroll
            for (int c = 0; c < 5; c++) P[psw(k0) + 64 * c] = pw[c];          // (psw only looks at bits 2 and 5)
        } else
#pragma unroll
        for (int c = 0; c < NR; c++) {
            if (c < nrow) {
                // partner value Z[512 - k]: general lanes: partner lane's register 7 - c;
                // the k0 == 0 lane pairs with itself: register (8 - c) & 7
                v2f zb;
                if (c < 8) {
                    const v2f src = v[7 - c];
                    zb.x = __shfl(src.x, partner, 64);
                    zb.y = __shfl(src.y, partner, 64);
                } else { zb.x = 0.f; zb.y = 0.f; }
                if (k0 == 0) zb = v[(8 - c) & 7];
                const v2f za = v[c & 7];                   // c == 8 only for k0 == 0: Z[512] = Z[0]
                const v2f e = pk_add_conj(za, zb), o = pk_sub_conj(za, zb);      // za +- conj(zb)
                const v2f t = pk_cmul(o, tws[c]);
                const v2f xx = pk_add_mi(e, t);            // (e.x + t.y, e.y - t.x)
                const int k = k0 + 64 * c;
                P[psw(k)] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);   // (bins above kmax land in the row's padding: nobody reads them)
            }
        }
        wave_lds_sync();
        // ---- bands (F5-F8)
        const uint32_t so_out = (f - f_begin) * (uint32_t)p.bands * 4u;
        auto store_band = [&](int m, float e) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b32(to_u32(e), r_out, (uint32_t)m * 4u, so_out, 0); };
        if (mel_fast) {
            // the power values of BOTH bands are requested before the first multiply-add: one LDS round trip per frame for the taps instead of one per band
            // (the wave issues nothing while it waits for a read; the two bands' chains are independent)
            float pv[2][MW];
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int j = 0; j < MW; j++) if (q == 1 || j < MWL) { const int k = mk[q] + j; pv[q][j] = P[psw(k <= pmax ? k : pmax)]; }
            wave_lds_sync();
            float e2[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < MW; j++) if (q == 1 || j < MWL) e = __builtin_fmaf(mw[q][j], pv[q][j], e);       // (a dropped tap is a zero weight: fmaf(0, P, e) = e)
                e = e * emph_r[q];
                e2[q] = e * p.gain;
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int m = lane + 64 * q;
                if (all_bands) store_band(m, e2[q]); else if (m < p.bands) store_band(m, e2[q]);
            }
        } else
        for (int m = lane; m < p.bands; m += 64) {
            float e;
            if (p.spec_type == 1) {
                e = 0.f;
                const int kb = s_k0[m], n = s_cnt[m];
                const float* w = s_melw + s_off[m];
                for (int j = 0; j < n; j++) e = __builtin_fmaf(w[j], P[psw(kb + j)], e);
            } else {
                e = 0.25f * P[psw(m)];
                if (p.spec_type == 3) e = __builtin_sqrtf(e)  /* correctly rounded (the __fsqrt_rn intrinsic is the raw 1-ulp v_sqrt_f32) */;
            }
            e = e * s_emph[m];
            e = e * p.gain;
            store_band(m, e);
        }
        wave_lds_sync();
    }
    } else if (p.queue && threadIdx.x == 0) nxt = atomicAdd(p.queue, 1u);
    if (!p.queue) break;
    phase ^= 1;
    if (threadIdx.x == 0) s_next[phase] = nxt;
    __syncthreads();
    chunk = __builtin_amdgcn_readfirstlane(s_next[phase]);
    }
}

// =====================================================================================================
// General FFT length: N2 = 64 R packed complex points, R in {2, 4, 16, 32} (NFFT 256 / 512 / 2048 / 4096:
// 4 kHz .. 64 kHz audio at the default band settings).  Same mapping, wave per frame:
//   pass 1  lane m holds z[64a + m], a < R (a < AZ non-zero); radix-R butterfly in registers (log2 R
//           radix-2 stages, FE-1 twiddle forms, pruned where an input is a structural zero), then the
//           inter-pass twiddle W_N2^{m a'} from a per-lane table in LDS ([a' - 1][lane]: conflict free)
//   then the R sub-FFTs of 64 points, eight at a time (group g = sub-FFTs 8g .. 8g + 7), each group
//   exactly the [8, 8] tail of the 1024-point kernel (X1, pass 2, X2, pass 3):
//           lane (al, b') ends up holding Z[(8g + al) + R b' + 8R c'], c' = 0..7, in z[8g + c'].
//   split   partner of (a', b', c') is ((R - a') mod R, 7 - b', 7 - c') (a' != 0), i.e. a fixed partner
//           lane whose value sits in group R/8 - g - (al > 0): a select between two registers, then
//           ds_bpermute.  Split twiddles W_NFFT^k come from LDS.
// For R < 8 only the lanes with al < R carry sub-FFTs.
// t * (-i) = (t.y, -t.x) as one packed multiply by (1, -1)
__device__ __forceinline__ v2f pk_mi(v2f t, v2f one_mone) {
#if WSA_FE_PK_ASM
    v2f d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(t), "v"(one_mone)); return d;
#else
    v2f d; d.x = t.y; d.y = -t.x; return d;
#endif
}

template <int R, int NZ>
__device__ __forceinline__ void radix_r(v2f (&v)[R], const float2* __restrict__ tw64, const v2f ss, const v2f one_mone) {
    // FE-1 butterfly (oracle/frontend.c butterfly()): in place, bit-reversed result, then reordered.
    // Leading non-zero inputs of a block of size 2h: min(NZ, 2h) (no loop-carried state: the loops must unroll
    // completely or the register arrays turn into indexed moves).
#pragma unroll
    for (int h = R / 2; h >= 1; h >>= 1) {
        const int nz = NZ < 2 * h ? NZ : 2 * h;
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int j = 0; j < h; j++) {
                const int a = blk + j, b = a + h;
                v2f t;
                if (j + h < nz) { const v2f u = v[a], w = v[b]; v[a] = pk_add(u, w); t = pk_sub(u, w); }
                else if (j < nz) t = v[a];            // w == 0: u + 0 = u, u - 0 = u
                else continue;                        // both structural zeros
                if (j == 0) v[b] = t;
                else if (2 * j == h) v[b] = pk_mi(t, one_mone);
                else if (4 * j == h) v[b] = pk_mul_w8(t, ss);
                else if (4 * j == 3 * h) v[b] = pk_mul_w83(t, ss);
                else v[b] = pk_cmul(t, to_v2f(tw64[j * 32 / h]));
            }
        }
    }
    constexpr int P = R == 2 ? 1 : R == 4 ? 2 : R == 8 ? 3 : R == 16 ? 4 : R == 32 ? 5 : 6;
    v2f y[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        int r = 0;
#pragma unroll
        for (int i = 0; i < P; i++) r |= ((k >> i) & 1) << (P - 1 - i);
        y[k] = v[r];
    }
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = y[k];
}

struct FeLdsLayout { size_t melw, twl, tws, wn, mwp, wave0, xbytes, pbytes, total; };
__host__ __device__ inline FeLdsLayout fe_lds_layout_rx(int mel_total, int bands, int kmax, int R, int AZ = 0, int MW = 0) {
    FeLdsLayout L;
    const size_t shared_words = (size_t)((mel_total + 3) & ~3) + 4 * (size_t)bands;
    L.melw = 0;
    L.twl = ((shared_words + 3) & ~(size_t)3) * 4;
    L.tws = L.twl + (size_t)(R - 1) * 64 * 8;
    L.wn = L.tws + (size_t)((kmax + 2) & ~1) * 8;                 // window pairs [AZ][64] v2f
    L.mwp = L.wn + (size_t)AZ * 64 * 8;                           // mel taps of the lane's two bands, zero padded: [2][MW][64] f32
    L.wave0 = L.mwp + (size_t)2 * MW * 64 * 4;
    L.xbytes = (size_t)XBUF * 8;
    L.pbytes = (size_t)((kmax + 1 + 3) & ~3) * 4;
    L.total = L.wave0 + 4 * (L.xbytes + L.pbytes);
    return L;
}

template <int R, int AZ, int MW>
__global__ __launch_bounds__(256) void fe_kernel_rx(FeParams p) {
    constexpr int NG = (R + 7) / 8;            // groups of eight 64-point sub-FFTs
    constexpr int AL = R < 8 ? R : 8;          // sub-FFTs in a group
    constexpr int N2 = 64 * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: scalar frame counter and addresses)
    const int clip = blockIdx.y;
    const FeLdsLayout L = fe_lds_layout_rx(p.mel_total, p.bands, p.kmax, R, AZ, MW);
    float* s_melw = reinterpret_cast<float*>(smem);
    int* s_k0 = reinterpret_cast<int*>(s_melw + ((p.mel_total + 3) & ~3));
    int* s_cnt = s_k0 + p.bands;
    int* s_off = s_cnt + p.bands;
    float* s_emph = reinterpret_cast<float*>(s_off + p.bands);
    v2f* s_twl = reinterpret_cast<v2f*>(smem + L.twl);
    v2f* s_tws = reinterpret_cast<v2f*>(smem + L.tws);
    v2f* s_wn = reinterpret_cast<v2f*>(smem + L.wn);
    float* s_mwp = reinterpret_cast<float*>(smem + L.mwp);
    v2f* X = reinterpret_cast<v2f*>(smem + L.wave0 + (size_t)wave * L.xbytes);
    float* P = reinterpret_cast<float*>(smem + L.wave0 + 4 * L.xbytes + (size_t)wave * L.pbytes);

    // loop invariants that do not fit the register file next to 2 R data registers live in LDS, one
    // conflict-free [..][lane] row per use: inter-pass twiddles, split twiddles, window pairs, mel taps
    for (int i = threadIdx.x; i < p.mel_total; i += 256) s_melw[i] = p.mel_w[i];
    for (int i = threadIdx.x; i < p.bands; i += 256) {
        s_emph[i] = p.emph[i];
        if (p.spec_type == 1) { s_k0[i] = p.mel_k0[i]; s_cnt[i] = p.mel_cnt[i]; s_off[i] = p.mel_off[i]; }
    }
    for (int i = threadIdx.x; i < (R - 1) * 64; i += 256) s_twl[i] = to_v2f(p.tw_n2[(i & 63) * ((i >> 6) + 1)]);
    for (int i = threadIdx.x; i <= p.kmax; i += 256) s_tws[i] = to_v2f(p.tw_nfft[i]);
    for (int i = threadIdx.x; i < AZ * 64; i += 256) {
        const int n = 2 * (64 * (i >> 6) + (i & 63));
        v2f w; w.x = n < p.win ? p.window[n] : 0.0f; w.y = n + 1 < p.win ? p.window[n + 1] : 0.0f;
        s_wn[i] = w;
    }
    __syncthreads();
    bool taps_fit = true;
    for (int i = threadIdx.x; i < 2 * MW * 64; i += 256) {           // [q][j][lane]: tap j of band lane + 64 q, zero padded
        const int ln = i & 63, j = (i >> 6) % MW, q = (i >> 6) / MW, m = ln + 64 * q;
        float w = 0.f;
        if (p.spec_type == 1 && m < p.bands) { if (j < s_cnt[m]) w = s_melw[s_off[m] + j]; if (j == 0 && s_cnt[m] > MW) taps_fit = false; }
        s_mwp[i] = w;
    }
    const bool mel_fast = p.spec_type == 1 && p.bands <= 128 && __syncthreads_and(taps_fit);

    const uint32_t nfr = p.n_frames[clip];
    const uint32_t f_begin = (uint32_t)(blockIdx.x * 4 + wave) * (uint32_t)p.frames_per_wave;
    if (f_begin >= nfr) return;
    uint32_t f_end = f_begin + (uint32_t)p.frames_per_wave;
    if (f_end > nfr) f_end = nfr;

    v2f tw2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) tw2[k] = to_v2f(p.tw_64[(lane & 7) * k]);
    v2f ss; ss.x = 0.70710678118654752440f; ss.y = 0.70710678118654752440f;
    v2f one_mone; one_mone.x = 1.0f; one_mone.y = -1.0f;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const bool act = hi3 < AL;                                   // lane carries a sub-FFT (always for R >= 8)
    // partner lanes of the real split (see header): al > 0: ((R - al) mod 8, 7 - b'); al == 0: group 0 pairs
    // b' with 8 - b' (lane 0 with itself), later groups with 7 - b'
    const int part_hi = ((((R - hi3) & 7) << 3) | (7 - lo3)) & 63;
    const int part_g0 = hi3 > 0 ? part_hi : ((8 - lo3) & 7);
    const int part_gn = hi3 > 0 ? part_hi : (7 - lo3);
    int mk[2];
#pragma unroll
    for (int q = 0; q < 2; q++) { const int m = lane + 64 * q; mk[q] = (p.spec_type == 1 && m < p.bands) ? s_k0[m] : 0; }
    const int pmax = p.kmax;

    const float* clip_pcm = p.pcm + (uint64_t)clip * p.clip_stride + (p.pcm_off ? p.pcm_off[clip] : 0u);
    uint32_t* out_base = p.spec + (uint64_t)p.frame_off[clip] * (uint32_t)p.bands;

    // branch-free prefetch (see fe_kernel_r8): clamped 8-byte pairs, untouched until the next iteration
    int ld_idx[AZ]; bool ld_v0[AZ], ld_v1[AZ], ld_odd[AZ];
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        ld_idx[a] = min(n, p.win - 2);
        ld_v0[a] = n < p.win; ld_v1[a] = n + 1 < p.win; ld_odd[a] = n == p.win - 1;
    }
    auto load_pcm = [&](uint32_t f, v2f (&x)[AZ]) __attribute__((always_inline)) {
        const float* fr = clip_pcm + (uint64_t)f * (uint32_t)p.hop;
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            const pcm2 q = *reinterpret_cast<const pcm2*>(fr + ld_idx[a]);
            x[a].x = q.x; x[a].y = q.y;
        }
    };
    v2f xin[AZ];
    load_pcm(f_begin, xin);

    for (uint32_t f = f_begin; f < f_end; f++) {
        v2f v[R];
#pragma unroll
        for (int a = 0; a < R; a++) { v[a].x = 0.f; v[a].y = 0.f; }
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            v2f x;
            x.x = ld_v0[a] ? (ld_odd[a] ? xin[a].y : xin[a].x) : 0.f;
            x.y = ld_v1[a] ? xin[a].y : 0.f;
            v[a] = pk_mul(x, s_wn[a * 64 + lane]);
        }
        if (f + 1 < f_end) load_pcm(f + 1, xin);
        // ---- pass 1: radix R over a, twiddle W_N2^{m a'}
        radix_r<R, AZ>(v, p.tw_64, ss, one_mone);
#pragma unroll
        for (int k = 1; k < R; k++) v[k] = pk_cmul(v[k], s_twl[(k - 1) * 64 + lane]);
        // ---- the R sub-FFTs of 64 points, eight at a time
        v2f z[NG * 8];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            v2f u[8];
#pragma unroll
            for (int k = 0; k < AL; k++) X[k * XROW + lane] = v[8 * g + k];
            wave_lds_sync();
#pragma unroll
            for (int b = 0; b < 8; b++) { if (act) u[b] = X[hi3 * XROW + 8 * b + lo3]; else { u[b].x = 0.f; u[b].y = 0.f; } }
            wave_lds_sync();
            radix8_pk<8>(u, ss);
#pragma unroll
            for (int k = 1; k < 8; k++) u[k] = pk_cmul(u[k], tw2[k]);
#pragma unroll
            for (int k = 0; k < 8; k++) X[hi3 * XROW + k * 9 + lo3] = u[k];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; c++) u[c] = X[hi3 * XROW + lo3 * 9 + c];
            wave_lds_sync();
            radix8_pk<8>(u, ss);
#pragma unroll
            for (int c = 0; c < 8; c++) z[8 * g + c] = u[c];
        }
        // ---- real-FFT split + 4x power: X[k] from Z[k] and conj(Z[N2 - k])
#pragma unroll
        for (int g = 0; g < NG; g++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (8 * g + 8 * R * c <= p.kmax) {                       // smallest k of this row (uniform)
                    const v2f s_hi = z[8 * (NG - 1 - g) + 7 - c];        // partner's group when al > 0
                    const v2f s_lo = z[8 * ((NG - g) % NG) + 7 - c];     // ... when al == 0
                    v2f src; src.x = hi3 > 0 ? s_hi.x : s_lo.x; src.y = hi3 > 0 ? s_hi.y : s_lo.y;
                    const int partner = g == 0 ? part_g0 : part_gn;
                    v2f zb;
                    zb.x = __shfl(src.x, partner, 64);
                    zb.y = __shfl(src.y, partner, 64);
                    if (g == 0 && lane == 0) zb = z[(8 - c) & 7];        // k = 8R c pairs with 8R (8 - c)
                    const v2f za = z[8 * g + c];
                    const int k = 8 * g + hi3 + R * lo3 + 8 * R * c;
                    const v2f tw = s_tws[k <= p.kmax ? k : 0];
                    const v2f e = pk_add_conj(za, zb), o = pk_sub_conj(za, zb);
                    const v2f t = pk_cmul(o, tw);
                    const v2f xx = pk_add_mi(e, t);
                    if (act && k <= p.kmax) P[k] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
                }
            }
        }
        if (p.kmax == N2 && lane == 0) {                                 // X[N2] from Z[0] alone
            const v2f za = z[0];
            const v2f e = pk_add_conj(za, za), o = pk_sub_conj(za, za);
            const v2f t = pk_cmul(o, s_tws[N2]);
            const v2f xx = pk_add_mi(e, t);
            P[N2] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
        }
        wave_lds_sync();
        // ---- bands (F5-F8)
        uint32_t* out = out_base + (uint64_t)f * (uint32_t)p.bands;
        if (mel_fast) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int m = lane + 64 * q;
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < MW; j++) { const int k = mk[q] + j; e = __builtin_fmaf(s_mwp[(q * MW + j) * 64 + lane], P[k <= pmax ? k : pmax], e); }
                e = e * s_emph[m < p.bands ? m : 0];
                e = e * p.gain;
                if (m < p.bands) out[m] = to_u32(e);
            }
        } else
        for (int m = lane; m < p.bands; m += 64) {
            float e;
            if (p.spec_type == 1) {
                e = 0.f;
                const int kb = s_k0[m], n = s_cnt[m];
                const float* w = s_melw + s_off[m];
                for (int j = 0; j < n; j++) e = __builtin_fmaf(w[j], P[kb + j], e);
            } else {
                e = 0.25f * P[m];
                if (p.spec_type == 3) e = __builtin_sqrtf(e)  /* correctly rounded (the __fsqrt_rn intrinsic is the raw 1-ulp v_sqrt_f32) */;
            }
            e = e * s_emph[m];
            e = e * p.gain;
            out[m] = to_u32(e);
        }
        wave_lds_sync();
    }
}


// =====================================================================================================
// NFFT = 3 * 2^k (FE-1 F2: 3072 points at 44.1 / 48 kHz, the rate of the reference's offline path, ref @B18765):
// N2 = 3 M packed complex points, M = 64 RM.  Same mapping, wave per frame:
//   stage 0  lane m holds z[64 a + m], a < 3 RM (a < AZ non-zero); a = j RM + aM, j = the third of the packed frame.
//            Radix-3 butterfly over j in registers (oracle/frontend.c radix3(): t = x1 + x2, y0 = x0 + t, m = fma(-1/2, t, x0),
//            s = fl32(sqrt(3)/2) (x1 - x2), y1 = m - i s, y2 = m + i s; thirds that are structural zeros are pruned), then the
//            twiddle W_N2^{(64 aM + m) k3} from a per-lane LDS table
//   then     for k3 = 0, 1, 2 the M-point transform exactly as in fe_kernel_rx<RM>: radix-RM pass, twiddle W_M^{m a'}, the RM
//            sub-FFTs of 64 points eight at a time through the [8, 8] tail; lane (al, b') ends up with
//            Z_k3[(8g + al) + RM b' + 8 RM c'] in z[k3][8g + c'], and bin k = 3 k' + k3 of the N2-point transform is Z_k3[k']
//   split    Z[N2 - k]: k3 = 0 pairs with itself (k' <-> M - k': the partner rule of fe_kernel_rx); k3 = 1 pairs with k3 = 2 at
//            k'' = M - 1 - k' = (RM - 1 - a') + RM (7 - b') + 8 RM (7 - c'): group NG - 1 - g, register 7 - c', lane (AL - 1 - al, 7 - b')
struct FeLdsLayout3 { size_t tw3, twl, tws, wn, mwp, wave0, xbytes, pbytes, total; };
__host__ __device__ inline FeLdsLayout3 fe_lds_layout_r3(int mel_total, int bands, int kmax, int RM, int AZ, int MW, int pad_k = 0) {
    if (pad_k > kmax) kmax = pad_k;                               // (the AF > 0 instantiation: split twiddles and power rows up to the last bin of the rows it keeps)
    FeLdsLayout3 L;
    const size_t shared_words = (size_t)((mel_total + 3) & ~3) + 4 * (size_t)bands;
    L.tw3 = ((shared_words + 3) & ~(size_t)3) * 4;
    L.twl = L.tw3 + (size_t)2 * RM * 64 * 8;                      // W_N2^{(64 aM + lane) k3}: [k3 - 1][aM][lane] v2f
    L.tws = L.twl + (size_t)(RM > 1 ? RM - 1 : 1) * 64 * 8;       // W_M^{lane k}: [k - 1][lane] v2f
    L.wn = L.tws + (size_t)((kmax + 2) & ~1) * 8;
    L.mwp = L.wn + (size_t)AZ * 64 * 8;
    L.wave0 = L.mwp + (size_t)2 * MW * 64 * 4;
    L.xbytes = (size_t)XBUF * 8;
    L.pbytes = (size_t)((kmax + 1 + 3) & ~3) * 4;
    L.total = L.wave0 + 4 * (L.xbytes + L.pbytes);
    return L;
}

// CR = rows c of the output registers the power spectrum reaches into (3 (8 RM c) <= kmax): the instantiation for the usual band limit
// (f_max well below Nyquist) keeps only those and their split partners 7 - c alive after the last radix-8 stage
// AF > 0 (round 5; the 48 kHz instantiation — a 25 ms window at 44.1 kHz is 1102 samples, under the 9 x 128 this needs, and stays on the AF = 0 kernel): the first AF 64-point blocks of packed input lie inside the window for every lane, so only the blocks
// behind them select samples — branch-free, on lane masks in scalar registers (the bool form compiled to two nested exec-mask branches per block with the masks
// spilled to vector lanes, and to ONE LDS round trip per block for the window: ten serial round trips per frame) —, the window values live in registers, the
// samples and bands travel by buffer addressing (32-bit scalar frame offset + 32-bit lane offset), every output row the template keeps exists (no test
// against kmax per row; launch_frontend checks it) and the power rows are padded (no test per lane)
template <int RM, int AZ, int MW, int CR = 8, int AF = 0>
__global__ __launch_bounds__(256) void fe_kernel_r3(FeParams p) {
    constexpr bool LN = AF > 0;
    constexpr int NG = (RM + 7) / 8;           // groups of eight 64-point sub-FFTs per M-point transform
    constexpr int AL = RM < 8 ? RM : 8;        // sub-FFTs in a group
    constexpr int M = 64 * RM, N2 = 3 * M;
    constexpr int NZM = AZ < RM ? AZ : RM;     // leading non-zero inputs of every M-point transform
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (uniform: scalar frame counter and addresses)
    __shared__ uint32_t s_next[2];              // persistent launch (p.queue): the hand-off of the next chunk's number, as in fe_kernel_r8
    const FeLdsLayout3 L = fe_lds_layout_r3(p.mel_total, p.bands, p.kmax, RM, AZ, MW, AF > 0 ? 3 * 8 * RM * CR + 2 : 0);
    float* s_melw = reinterpret_cast<float*>(smem);
    int* s_k0 = reinterpret_cast<int*>(s_melw + ((p.mel_total + 3) & ~3));
    int* s_cnt = s_k0 + p.bands;
    int* s_off = s_cnt + p.bands;
    float* s_emph = reinterpret_cast<float*>(s_off + p.bands);
    v2f* s_tw3 = reinterpret_cast<v2f*>(smem + L.tw3);
    v2f* s_twl = reinterpret_cast<v2f*>(smem + L.twl);
    v2f* s_tws = reinterpret_cast<v2f*>(smem + L.tws);
    v2f* s_wn = reinterpret_cast<v2f*>(smem + L.wn);
    float* s_mwp = reinterpret_cast<float*>(smem + L.mwp);
    v2f* X = reinterpret_cast<v2f*>(smem + L.wave0 + (size_t)wave * L.xbytes);
    float* P = reinterpret_cast<float*>(smem + L.wave0 + 4 * L.xbytes + (size_t)wave * L.pbytes);

    for (int i = threadIdx.x; i < p.mel_total; i += 256) s_melw[i] = p.mel_w[i];
    for (int i = threadIdx.x; i < p.bands; i += 256) {
        s_emph[i] = p.emph[i];
        if (p.spec_type == 1) { s_k0[i] = p.mel_k0[i]; s_cnt[i] = p.mel_cnt[i]; s_off[i] = p.mel_off[i]; }
    }
    for (int i = threadIdx.x; i < 2 * RM * 64; i += 256) {
        const int ln = i & 63, aM = (i >> 6) % RM, k3 = (i >> 6) / RM + 1;
        s_tw3[i] = to_v2f(p.tw_n2[(64 * aM + ln) * k3]);
    }
    for (int i = threadIdx.x; i < (RM - 1) * 64; i += 256) s_twl[i] = to_v2f(p.tw_m[(i & 63) * ((i >> 6) + 1)]);
    for (int i = threadIdx.x; i <= (LN ? 3 * 8 * RM * CR + 2 : p.kmax); i += 256) s_tws[i] = to_v2f(i <= p.kmax ? p.tw_nfft[i] : make_float2(0.f, 0.f));
    for (int i = threadIdx.x; i < AZ * 64; i += 256) {
        const int n = 2 * (64 * (i >> 6) + (i & 63));
        v2f w; w.x = n < p.win ? p.window[n] : 0.0f; w.y = n + 1 < p.win ? p.window[n + 1] : 0.0f;
        s_wn[i] = w;
    }
    __syncthreads();
    bool taps_fit = true;
    for (int i = threadIdx.x; i < 2 * MW * 64; i += 256) {
        const int ln = i & 63, j = (i >> 6) % MW, q = (i >> 6) / MW, m = ln + 64 * q;
        float w = 0.f;
        if (p.spec_type == 1 && m < p.bands) { if (j < s_cnt[m]) w = s_melw[s_off[m] + j]; if (j == 0 && s_cnt[m] > MW) taps_fit = false; }
        s_mwp[i] = w;
    }
    const bool mel_fast = p.spec_type == 1 && p.bands <= 128 && __syncthreads_and(taps_fit);

    v2f tw2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) tw2[k] = to_v2f(p.tw_64[(lane & 7) * k]);
    v2f ss; ss.x = 0.70710678118654752440f; ss.y = 0.70710678118654752440f;
    v2f one_mone; one_mone.x = 1.0f; one_mone.y = -1.0f;
    v2f mhalf; mhalf.x = -0.5f; mhalf.y = -0.5f;
    v2f c3; c3.x = 0.86602540378443864676f; c3.y = 0.86602540378443864676f;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const bool act = hi3 < AL;
    // partner lanes of the real split (see the header): transform 0 as in fe_kernel_rx, transforms 1 <-> 2 mirrored
    const int part_hi = ((((RM - hi3) & 7) << 3) | (7 - lo3)) & 63;
    const int part_g0 = hi3 > 0 ? part_hi : ((8 - lo3) & 7);
    const int part_gn = hi3 > 0 ? part_hi : (7 - lo3);
    const int part_x = ((((AL - 1 - hi3) & 7) << 3) | (7 - lo3)) & 63;
    int mk[2];
#pragma unroll
    for (int q = 0; q < 2; q++) { const int m = lane + 64 * q; mk[q] = (p.spec_type == 1 && m < p.bands) ? s_k0[m] : 0; }
    const int pmax = p.kmax;

    int ld_idx[AZ]; bool ld_v0[AZ], ld_v1[AZ], ld_odd[AZ];
    uint32_t ld_off[AZ]; uint64_t m_v0[AZ], m_v1[AZ], m_odd[AZ];           // LN: byte offsets, lane masks (scalar registers)
#pragma unroll
    for (int a = 0; a < AZ; a++) {
        const int n = 2 * (64 * a + lane);
        ld_idx[a] = min(n, p.win - 2);
        ld_v0[a] = n < p.win; ld_v1[a] = n + 1 < p.win; ld_odd[a] = n == p.win - 1;
        ld_off[a] = (uint32_t)max(ld_idx[a], 0) * 4u;
        m_v0[a] = __ballot(ld_v0[a]); m_v1[a] = __ballot(ld_v1[a]); m_odd[a] = __ballot(ld_odd[a]);
    }
    v2f wn_r[LN ? AZ : 1];
    if (LN) {
#pragma unroll
        for (int a = 0; a < AZ; a++) wn_r[LN ? a : 0] = s_wn[a * 64 + lane];
    }
    // chunk = 4 x frames_per_wave frames of one clip: blockIdx (x = chunk of the clip, y = clip), or — persistent launch, p.queue — chunk after chunk from a
    // device counter (the tables above are filled once per workgroup instead of once per 100 frames)
    const uint32_t cpc = p.queue ? (uint32_t)p.chunks_per_clip : gridDim.x;
    uint32_t chunk = blockIdx.y * gridDim.x + blockIdx.x, nxt = 0;
    if (p.queue) {
        if (threadIdx.x == 0) s_next[0] = atomicAdd(p.queue, 1u);
        __syncthreads();
        chunk = s_next[0];
    }
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    int phase = 0;
    for (; !p.queue || chunk < p.n_chunks;) {
    const uint32_t clip = __builtin_amdgcn_readfirstlane(chunk / cpc), cx = chunk - clip * cpc;
    const uint32_t nfr = p.n_frames[clip];
    const uint32_t f_begin = (cx * 4u + (uint32_t)wave) * (uint32_t)p.frames_per_wave;
    uint32_t f_end = f_begin + (uint32_t)p.frames_per_wave;
    if (f_end > nfr) f_end = nfr;
    const float* clip_pcm = p.pcm + uniform_u64((uint64_t)clip * p.clip_stride + (p.pcm_off ? p.pcm_off[clip] : 0u));
    uint32_t* out_base = p.spec + uniform_u64((uint64_t)p.frame_off[clip] * (uint32_t)p.bands);
    if (f_begin < f_end) {
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(clip_pcm + (uint64_t)f_begin * (uint32_t)p.hop), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(out_base + (uint64_t)f_begin * (uint32_t)p.bands, 0, 0x7fffffff, 0x00020000);
    auto load_pcm = [&](uint32_t f, v2f (&x)[AZ]) __attribute__((always_inline)) {
        if (LN) {
            const uint32_t so = (f - f_begin) * (uint32_t)p.hop * 4u;
#pragma unroll
            for (int a = 0; a < AZ; a++) x[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r_in, ld_off[a], so, 0));
            return;
        }
        const float* fr = clip_pcm + (uint64_t)f * (uint32_t)p.hop;
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            const pcm2 q = *reinterpret_cast<const pcm2*>(fr + ld_idx[a]);
            x[a].x = q.x; x[a].y = q.y;
        }
    };
    v2f xin[AZ];
    load_pcm(f_begin, xin);
    if (p.queue && threadIdx.x == 0) nxt = atomicAdd(p.queue, 1u);          // (requested behind the first samples: the two round trips overlap)

    for (uint32_t f = f_begin; f < f_end; f++) {
        // ---- window: the frame's samples become x·w in place (samples outside the window are zeros of the table)
#pragma unroll
        for (int a = 0; a < AZ; a++) {
            if (LN) {
                v2f u = xin[a];
                if (a >= AF) { u.x = sel_mask(0.f, sel_mask(xin[a].x, xin[a].y, m_odd[a]), m_v0[a]); u.y = sel_mask(0.f, xin[a].y, m_v1[a]); }
                xin[a] = pk_mul(u, wn_r[LN ? a : 0]);
                continue;
            }
            v2f u;
            u.x = ld_v0[a] ? (ld_odd[a] ? xin[a].y : xin[a].x) : 0.f;
            u.y = ld_v1[a] ? xin[a].y : 0.f;
            xin[a] = pk_mul(u, s_wn[a * 64 + lane]);
        }
        // ---- per output residue k3: the radix-3 stage over the thirds j of the packed frame (a = j RM + aM), recomputed from the windowed
        //      samples for each k3 — the three transforms' inputs are never alive together (48 registers less than holding w[3][RM];
        //      the price is t / d / m / s computed twice) —, then the M-point transform
        v2f z[3][NG * 8];
#pragma unroll
        for (int k3 = 0; k3 < 3; k3++) {
            v2f w[RM];
#pragma unroll
            for (int aM = 0; aM < RM; aM++) {
                v2f zero; zero.x = 0.f; zero.y = 0.f;
                if (aM >= AZ) { w[aM] = zero; continue; }                                       // all three thirds are structural zeros
                const v2f x0 = xin[aM];
                if (RM + aM >= AZ) { w[aM] = k3 == 0 ? x0 : pk_cmul(x0, s_tw3[((k3 - 1) * RM + aM) * 64 + lane]); continue; }      // x1 = x2 = 0: y0 = y1 = y2 = x0
                const v2f x1 = xin[RM + aM];
                v2f t, d;
                if (2 * RM + aM >= AZ) { t = x1; d = x1; }                                        // x2 = 0
                else { const v2f x2 = xin[2 * RM + aM]; t = pk_add(x1, x2); d = pk_sub(x1, x2); }
                if (k3 == 0) { w[aM] = pk_add(x0, t); continue; }
                const v2f m = pk_fma(mhalf, t, x0);
                const v2f sq = pk_mul(c3, d);
                const v2f y = k3 == 1 ? pk_add_mi(m, sq) : pk_sub_mi(m, sq);                      // m - i s, m + i s
                w[aM] = pk_cmul(y, s_tw3[((k3 - 1) * RM + aM) * 64 + lane]);
            }
            if (k3 == 2) { if (LN) load_pcm(min(f + 1, f_end - 1), xin); else if (f + 1 < f_end) load_pcm(f + 1, xin); }          // the windowed samples are dead: the next frame's take their registers
            if constexpr (RM > 1) {
                radix_r<RM, NZM>(w, p.tw_64, ss, one_mone);
#pragma unroll
                for (int k = 1; k < RM; k++) w[k] = pk_cmul(w[k], s_twl[(k - 1) * 64 + lane]);
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                v2f u[8];
                if constexpr (WSA_FE_X1REG && LN && AL == 8) {       // (every lane takes part: AL = 8 rows of 8)
#pragma unroll
                    for (int k = 0; k < 8; k++) u[k] = w[8 * g + k];
                    fe_transpose_hi3(u);
                } else {
#pragma unroll
                for (int k = 0; k < AL; k++) X[k * XROW + lane] = w[8 * g + k];
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < 8; b++) { if (act) u[b] = X[hi3 * XROW + 8 * b + lo3]; else { u[b].x = 0.f; u[b].y = 0.f; } }
                wave_lds_sync();
                }
                radix8_pk<8>(u, ss);
#pragma unroll
                for (int k = 1; k < 8; k++) u[k] = pk_cmul(u[k], tw2[k]);
#pragma unroll
                for (int k = 0; k < 8; k++) X[hi3 * XROW + k * 9 + lo3] = u[k];
                wave_lds_sync();
#pragma unroll
                for (int c = 0; c < 8; c++) u[c] = X[hi3 * XROW + lo3 * 9 + c];
                wave_lds_sync();
                radix8_pk<8>(u, ss);
#pragma unroll
                for (int c = 0; c < 8; c++) z[k3][8 * g + c] = u[c];
            }
        }
        // ---- real-FFT split + 4x power: X[k] from Z[k] and conj(Z[N2 - k]), k = 3 k' + k3
#pragma unroll
        for (int k3 = 0; k3 < 3; k3++) {
#pragma unroll
            for (int g = 0; g < NG; g++) {
#pragma unroll
                for (int c = 0; c < CR; c++) {
                    if (LN || 3 * (8 * g + 8 * RM * c) + k3 <= p.kmax) {        // smallest k of this row (uniform; LN: every kept row exists)
                        v2f src; int partner;
                        if (k3 == 0) {
                            const v2f s_hi = z[0][8 * (NG - 1 - g) + 7 - c];     // partner's group when al > 0
                            const v2f s_lo = z[0][8 * ((NG - g) % NG) + 7 - c];  // ... when al == 0
                            src.x = hi3 > 0 ? s_hi.x : s_lo.x; src.y = hi3 > 0 ? s_hi.y : s_lo.y;
                            partner = g == 0 ? part_g0 : part_gn;
                        } else {
                            src = z[3 - k3][8 * (NG - 1 - g) + 7 - c];
                            partner = part_x;
                        }
                        v2f zb;
                        zb.x = __shfl(src.x, partner, 64);
                        zb.y = __shfl(src.y, partner, 64);
                        if (k3 == 0 && g == 0 && lane == 0) zb = z[0][(8 - c) & 7];   // k' = 8 RM c pairs with 8 RM (8 - c)
                        const v2f za = z[k3][8 * g + c];
                        const int k = 3 * (8 * g + hi3 + RM * lo3 + 8 * RM * c) + k3;
                        const v2f tw = s_tws[(LN || k <= p.kmax) ? k : 0];          // (LN: the table and the power rows are padded to the rows the template keeps)
                        const v2f e = pk_add_conj(za, zb), o = pk_sub_conj(za, zb);
                        const v2f t = pk_cmul(o, tw);
                        const v2f xx = pk_add_mi(e, t);
                        if (LN) { if (act) P[k] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y); }
                        else if (act && k <= p.kmax) P[k] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
                    }
                }
            }
        }
        if (p.kmax == N2 && lane == 0) {                                 // X[N2] from Z[0] alone
            const v2f za = z[0][0];
            const v2f e = pk_add_conj(za, za), o = pk_sub_conj(za, za);
            const v2f t = pk_cmul(o, s_tws[N2]);
            const v2f xx = pk_add_mi(e, t);
            P[N2] = __builtin_fmaf(xx.x, xx.x, xx.y * xx.y);
        }
        wave_lds_sync();
        // ---- bands (F5-F8)
        uint32_t* out = out_base + (uint64_t)f * (uint32_t)p.bands;
        const uint32_t so_out = (f - f_begin) * (uint32_t)p.bands * 4u;
        if (mel_fast) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int m = lane + 64 * q;
                // (taps and weights of the band requested before the first multiply-add)
                float pvq[MW], wq[MW];
#pragma unroll
                for (int j = 0; j < MW; j++) { const int k = mk[q] + j; pvq[j] = P[k <= pmax ? k : pmax]; wq[j] = s_mwp[(q * MW + j) * 64 + lane]; }
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < MW; j++) e = __builtin_fmaf(wq[j], pvq[j], e);
                e = e * s_emph[m < p.bands ? m : 0];
                e = e * p.gain;
                if (LN) { if (m < p.bands) __builtin_amdgcn_raw_buffer_store_b32(to_u32(e), r_out, (uint32_t)m * 4u, so_out, 0); }
                else if (m < p.bands) out[m] = to_u32(e);
            }
        } else
        for (int m = lane; m < p.bands; m += 64) {
            float e;
            if (p.spec_type == 1) {
                e = 0.f;
                const int kb = s_k0[m], n = s_cnt[m];
                const float* wt = s_melw + s_off[m];
                for (int j = 0; j < n; j++) e = __builtin_fmaf(wt[j], P[kb + j], e);
            } else {
                e = 0.25f * P[m];
                if (p.spec_type == 3) e = __builtin_sqrtf(e);
            }
            e = e * s_emph[m];
            e = e * p.gain;
            out[m] = to_u32(e);
        }
        wave_lds_sync();
    }
    } else if (p.queue && threadIdx.x == 0) nxt = atomicAdd(p.queue, 1u);
    if (!p.queue) break;
    phase ^= 1;
    if (threadIdx.x == 0) s_next[phase] = nxt;
    __syncthreads();
    chunk = __builtin_amdgcn_readfirstlane(s_next[phase]);
    }
}

size_t fe_lds_bytes(const FeParams& p) {                  // the 1024-point kernel
    const size_t shared_words = (size_t)((p.mel_total + 3) & ~3) + 4 * (size_t)p.bands;
    // (the power rows live in the exchange buffers)
    return ((shared_words + 3) & ~(size_t)3) * 4 + 4 * XBUF * sizeof(float2) + 7 * 64 * 8 + 5 * 64 * 8;      // + the W_512 table of the <.., true> variants + the split twiddles of the baseline instantiation
}

bool fe_supported_R(int R, int three) { return three ? (R == 1 || R == 2 || R == 4 || R == 8 || R == 16 || R == 32) : (R == 2 || R == 4 || R == 8 || R == 16 || R == 32 || R == 64); }

// fe_lds_required(): the launch functions below run "dry" — they note the dynamic LDS their instantiation would ask for and launch nothing
static thread_local size_t* fe_dry = nullptr;

template <int R, int AZ, int MW>
static void launch_rx(const FeParams& p, dim3 grid, size_t, hipStream_t s) {
    const size_t lds = fe_lds_layout_rx(p.mel_total, p.bands, p.kmax, R, AZ, MW).total;
    if (fe_dry) { *fe_dry = lds; return; }
    // dynamic LDS up to the device limit (160 KB per workgroup on gfx950) needs no opt-in on ROCm; a configuration
    // that asks for more fails the launch and is reported through hipGetLastError by the caller
    hipLaunchKernelGGL((fe_kernel_rx<R, AZ, MW>), grid, dim3(256), lds, s, p);
}

template <int RM, int AZ, int MW, int CR = 8, int AF = 0>
static void launch_r3(const FeParams& p, dim3 grid, hipStream_t s) {
    const size_t lds = fe_lds_layout_r3(p.mel_total, p.bands, p.kmax, RM, AZ, MW, AF > 0 ? 3 * 8 * RM * CR + 2 : 0).total;
    if (fe_dry) { *fe_dry = lds; return; }
    FeParams q = p;
    if (AF > 0 && p.queue) {          // (p.queue is null under WSA_FE_NO_QUEUE: Tuning::fe_no_queue, read when the batch was planned)
        // persistent launch (batches): n_cu x WSA_FE_WGS workgroups (default 2: the kernel's 162 VGPRs admit three) take the chunks from the queue — the ~30 KB of
        // tables are filled once per workgroup, and the other batches' back-end kernels find room beside it
        q.chunks_per_clip = grid.x; q.n_chunks = grid.x * grid.y;
        const unsigned want = (unsigned)(p.n_cu > 0 ? p.n_cu : 256) * (unsigned)(p.wg_per_cu >= 1 && p.wg_per_cu <= 3 ? p.wg_per_cu : 2);
        if (want < q.n_chunks) grid = dim3(want, 1, 1); else q.queue = nullptr;
    } else q.queue = nullptr;
    hipLaunchKernelGGL((fe_kernel_r3<RM, AZ, MW, CR, AF>), grid, dim3(256), lds, s, q);
}

void launch_frontend(const FeParams& p, int n_clips, int max_frames, int R, int three, hipStream_t s) {
    if (n_clips <= 0 || max_frames <= 0) return;
    const int frames_per_block = 4 * p.frames_per_wave;
    dim3 grid((max_frames + frames_per_block - 1) / frames_per_block, n_clips, 1);
    // the 1024-point kernel: chunks through a queue when the caller provides one (persistent launch), else one per workgroup.  Default TWO workgroups per CU
    // (8 of a CU's 16 wave slots at this kernel's 128 VGPRs, 68 of its 160 KB of LDS): alone the kernel then runs 0.35 instead of 0.27 ms, but a caller that
    // keeps several batches in flight gets the peak scan / gate / tracker of the other batches onto the same CUs beside it — the front end is bound by the
    // LDS pipe, they by instruction issue — and the pipelined step of the 1024-clip batch went from 0.67 to 0.60 ms (profiles/r04_notes.md); 4 restores the old occupancy
    FeParams q8 = p;
    q8.chunks_per_clip = grid.x; q8.n_chunks = grid.x * (uint32_t)n_clips;
    dim3 grid8(q8.n_chunks, 1, 1);
    if (q8.queue) { const uint32_t want = (uint32_t)(q8.n_cu > 0 ? q8.n_cu : 256) * (uint32_t)(p.wg_per_cu >= 1 && p.wg_per_cu <= 4 ? p.wg_per_cu : 2); if (want < grid8.x) grid8.x = want; else q8.queue = nullptr; }
    size_t lds = fe_lds_bytes(p);
    // co-residency experiments (Tuning::fe_wg_per_cu): dynamic LDS padded so that k workgroups fit a CU's 160 KB and k + 1 do not
    // (negative values: the cap by padding; positive ones cap the persistent launch's grid below, which leaves the LDS to the other kernels)
    if (p.wg_per_cu <= -1 && p.wg_per_cu >= -3) lds = std::max(lds, ((size_t)163840 / (size_t)(-p.wg_per_cu + 1) + 256) & ~(size_t)255);
    const int az = (p.win + 127) / 128;       // non-zero 64-point blocks of packed input
    if (three) {                              // NFFT = 3 * 64 R * 2: the smallest instantiation whose non-zero blocks cover the window
        if (R == 1) { if (az <= 2) launch_r3<1, 2, 12>(p, grid, s); else launch_r3<1, 3, 12>(p, grid, s); }
        else if (R == 2) { if (az <= 3) launch_r3<2, 3, 12>(p, grid, s); else launch_r3<2, 6, 12>(p, grid, s); }
        else if (R == 4) {
            if (az <= 5 && p.kmax < 3 * 8 * 4 * 3 && !p.fat) launch_r3<4, 5, 14, 3>(p, grid, s);          // 1536 points = 22.05 kHz: rows c = 0 .. 2
            else if (az <= 5) launch_r3<4, 5, 14>(p, grid, s); else if (az <= 8) launch_r3<4, 8, 14>(p, grid, s); else launch_r3<4, 12, 14>(p, grid, s);
        }
        else if (R == 8) {
            // (3072 points = 44.1 / 48 kHz with the default 4 kHz band limit: bins <= 383 sit in rows c = 0, 1 of the output registers)
            // ... and, at the reference's own geometry (25 ms window at 48 kHz — the offline context's rate; 1102 samples at 44.1 kHz fall short —: at least nine blocks inside the window, both kept rows of every residue
            // exist), the instantiation with the window in registers, mask selects and buffer addressing (AF = 9)
            if (az <= 10 && p.kmax < 3 * 8 * 8 * 2 && p.kmax >= 3 * 8 * 8 + 2 && p.win >= 128 * 9 && !p.fat) launch_r3<8, 10, 14, 2, 9>(p, grid, s);
            else if (az <= 10 && p.kmax < 3 * 8 * 8 * 2 && !p.fat) launch_r3<8, 10, 14, 2>(p, grid, s);
            else if (az <= 10) launch_r3<8, 10, 14>(p, grid, s); else if (az <= 16) launch_r3<8, 16, 14>(p, grid, s); else launch_r3<8, 24, 14>(p, grid, s);
        }
        else if (R == 16) { if (az <= 20) launch_r3<16, 20, 14>(p, grid, s); else if (az <= 32) launch_r3<16, 32, 14>(p, grid, s); else launch_r3<16, 48, 14>(p, grid, s); }
        else if (R == 32) launch_r3<32, 96, 14>(p, grid, s);
        return;
    }
    // (the general kernel instantiated at R = 8 keeps its invariants in LDS, needs 114 VGPRs = 4 waves per SIMD and
    // is slower than the register-resident one below: 0.368 vs 0.345 ms — the kernel is VALU + LDS throughput bound)
    if (R == 8 && fe_dry) { *fe_dry = lds; return; }
    if (R == 8) {
        // the lean instantiation (4 waves per SIMD) serves what it can hold: <= 5 rows of bins, <= 8 taps per band (launch argument mel_max_taps)
        const bool lean = p.kmax / 64 + 1 <= 5 && p.mel_max_taps <= 8 && p.spec_type == 1 && !p.fat;
        if (az <= 2) hipLaunchKernelGGL((fe_kernel_r8<2, 9, MELW, false>), grid8, dim3(256), lds, s, q8);
        else if (az <= 4 && lean && p.mel_max_taps_lo <= 4 && p.win >= 384 && p.bands == 128 && p.kmax / 64 + 1 == 5)
            hipLaunchKernelGGL((fe_kernel_r8<4, 5, 8, true, 4, 3>), grid8, dim3(256), lds, s, q8);      // the baseline geometry (16 kHz: 400-sample window, 128 bands, five rows of bins): BASE
        else if (az <= 4 && lean && p.mel_max_taps_lo <= 4) hipLaunchKernelGGL((fe_kernel_r8<4, 5, 8, true, 4>), grid8, dim3(256), lds, s, q8);
        else if (az <= 4 && lean) hipLaunchKernelGGL((fe_kernel_r8<4, 5, 8, true>), grid8, dim3(256), lds, s, q8);
        else if (az <= 4) hipLaunchKernelGGL((fe_kernel_r8<4, 9, MELW, false>), grid8, dim3(256), lds, s, q8);
        else hipLaunchKernelGGL((fe_kernel_r8<8, 9, MELW, false>), grid8, dim3(256), lds, s, q8);
    } else if (R == 2) launch_rx<2, 2, 12>(p, grid, lds, s);
    else if (R == 4) { if (az <= 2) launch_rx<4, 2, 12>(p, grid, lds, s); else launch_rx<4, 4, 12>(p, grid, lds, s); }
    else if (R == 16) {
        if (az <= 5) launch_rx<16, 5, 14>(p, grid, lds, s);
        else if (az <= 8) launch_rx<16, 8, 14>(p, grid, lds, s);
        else launch_rx<16, 16, 14>(p, grid, lds, s);
    } else if (R == 32) {
        if (az <= 10) launch_rx<32, 10, 14>(p, grid, lds, s);
        else if (az <= 16) launch_rx<32, 16, 14>(p, grid, lds, s);
        else launch_rx<32, 32, 14>(p, grid, lds, s);
    } else if (R == 64) {
        launch_rx<64, 64, 14>(p, grid, lds, s);
    }
}

// dynamic LDS the front-end kernel of this geometry asks for (a workgroup may have 160 KB on gfx950; wsa_batch_create / wsa_stream_create refuse settings beyond that)
size_t fe_lds_required(const FePlanHost& P, bool fat) {
    FeParams p{};
    p.win = P.win; p.hop = P.hop; p.kmax = P.kmax; p.bands = P.bands; p.spec_type = P.spec_type; p.mel_total = (int)P.mel_w.size();
    for (int32_t c_ : P.mel_cnt) if (c_ > p.mel_max_taps) p.mel_max_taps = c_;
    for (size_t i_ = 0; i_ < P.mel_cnt.size() && i_ < 64; i_++) if (P.mel_cnt[i_] > p.mel_max_taps_lo) p.mel_max_taps_lo = P.mel_cnt[i_];
    p.frames_per_wave = 25; p.fat = fat ? 1 : 0;
    size_t need = 0;
    fe_dry = &need;
    launch_frontend(p, 1, 1, P.R, P.three, nullptr);
    fe_dry = nullptr;
    return need;
}

}  // namespace wsa

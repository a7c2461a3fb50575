// formantanalyzer.js — drop-in for `require('formantanalyzer')` (formantanalyzer@1.1.6, the module
// /root/reference/src/index.js:9 loads) with the hot path running on an MI355X through libwsa.
//
// Same four exports, same argument meaning, same callback shapes, same string rejections as the
// reference's inner module 1 (ref dist/main.js:2 @B2750-5843):
//   configure(cfg)                                                         ref @B3292
//   LaunchAudioNodes(context_source, source_obj, callback, file_labels=[], offline=false,
//                    test_play=true, play_offset=null, play_duration=null) -> Promise<true>   ref @B4469
//   StopAudioNodes(reason)                                                 ref @B5699
//   set_predicted_label_for_segment(si, idx, label)                        ref @B21711
// Node has no Web Audio, so context_source 1 (file) takes the audio as a WAV ArrayBuffer/Buffer, a
// Float32Array (+ sampleRate option) or {pcm, sampleRate}; sources 2 (element) and 3 (microphone)
// reject with "Invalid audio source" like the reference does for anything it cannot play.
// Extension for batch work (BASELINE configs 2-4): LaunchBatch(clips, callback, labels).
'use strict';
const path = require('path');

let native = null;
function addon() {
  if (!native) {
    // no fallback: without the addon + libwsa.so (+ a gfx950 GPU at create time) this throws
    native = require(path.join(__dirname, '..', 'lib', 'wsa_napi.node'));
  }
  return native;
}

// ref @B2965 — the library's own defaults (output_level 4 there; plot keys accepted and ignored)
const settings = {
  plot_enable: false, spec_type: 1, output_level: 4, plot_len: 200, f_min: 50, f_max: 4000,
  N_fft_bins: 256, N_mel_bins: 128, window_width: 25, window_step: 25, pause_length: 200,
  min_seg_length: 50, auto_noise_gate: true, voiced_max_dB: 100, voiced_min_dB: 10, plot_lag: 1,
  pre_norm_gain: 1000, high_f_emph: 0, plot_canvas: null, canvas_width: 200, canvas_height: 100,
  sample_rate: 16000, device: 0,          // ours: rate assumed for raw Float32Array input; GPU ordinal
  devices: null,                          // ours: GPU ordinals a LaunchBatch is sharded over (contiguous clip shards, one context and one worker thread each)
  gather: null,                           // ours: collect the shards' feature rows on the first device with one RCCL exchange and copy them to the host once
                                          // (levels 5 / 13; null = whenever more than one device is configured, true / false = always / never)
  resample_to: 0,                         // ours: analysis rate the audio is converted to first (0 = analyse at its own rate); 48000 = what the
                                          // reference's offline path gets from the browser (OfflineAudioContext at 48 kHz, ref @B18769)
};

// ref @B3292: truthy-merge, except the five keys tested with `null !==` (0 / false are honoured,
// and an absent key overwrites with undefined — the app always passes all keys, src/index.js:370-390)
function configure(e) {
  if (null !== e.spec_type) settings.spec_type = e.spec_type;
  if (e.output_level) settings.output_level = e.output_level;
  if (null !== e.f_min) settings.f_min = e.f_min;
  if (e.f_max) settings.f_max = e.f_max;
  if (e.N_fft_bins) settings.N_fft_bins = e.N_fft_bins;
  if (e.N_mel_bins) settings.N_mel_bins = e.N_mel_bins;
  if (e.window_width) settings.window_width = e.window_width;
  if (e.window_step) settings.window_step = e.window_step;
  if (e.pre_norm_gain) settings.pre_norm_gain = e.pre_norm_gain;
  if (null !== e.high_f_emph) settings.high_f_emph = e.high_f_emph;
  if (e.pause_length) settings.pause_length = e.pause_length;
  if (e.min_seg_length) settings.min_seg_length = e.min_seg_length;
  if (null !== e.auto_noise_gate) settings.auto_noise_gate = e.auto_noise_gate;
  if (e.voiced_max_dB) settings.voiced_max_dB = e.voiced_max_dB;
  if (null !== e.voiced_min_dB) settings.voiced_min_dB = e.voiced_min_dB;
  if (e.sample_rate) settings.sample_rate = e.sample_rate;
  if (e.device !== undefined && e.device !== null) settings.device = e.device;
  if (e.devices !== undefined) settings.devices = Array.isArray(e.devices) && e.devices.length > 0 ? e.devices.slice() : null;
  if (e.resample_to !== undefined && e.resample_to !== null) settings.resample_to = e.resample_to;
  if (e.gather !== undefined) settings.gather = e.gather === null ? null : !!e.gather;
  settings.plot_enable = false;            // no canvas under Node
}

function native_config() {
  const c = {};
  for (const k of ['spec_type', 'output_level', 'f_min', 'f_max', 'N_fft_bins', 'N_mel_bins', 'window_width',
    'window_step', 'pause_length', 'min_seg_length', 'auto_noise_gate', 'voiced_max_dB', 'voiced_min_dB',
    'pre_norm_gain', 'high_f_emph']) c[k] = settings[k];
  return c;
}

// ---- minimal RIFF/WAVE reader (PCM 8/16/24/32-bit and float32, also WAVE_FORMAT_EXTENSIBLE); channel 0 is analysed —
// what a worklet reading inputs[0][0] sees (the reference hands the decoded buffer straight to its worklet node, ref @B20010;
// the worklet itself is not in the tree, so this choice belongs to the front-end specification FE-1)
const LITTLE_ENDIAN = new Uint8Array(new Uint16Array([1]).buffer)[0] === 1;
function decode_wav(buf) {
  const b = Buffer.isBuffer(buf) ? buf : Buffer.from(buf);
  if (b.length < 44 || b.toString('ascii', 0, 4) !== 'RIFF' || b.toString('ascii', 8, 12) !== 'WAVE') throw 'Unable to decode audio data';
  let pos = 12, fmt = null, data = null;
  while (pos + 8 <= b.length) {
    const id = b.toString('ascii', pos, pos + 4), len = b.readUInt32LE(pos + 4);
    if (id === 'fmt ') {
      fmt = { tag: b.readUInt16LE(pos + 8), ch: b.readUInt16LE(pos + 10), rate: b.readUInt32LE(pos + 12), bits: b.readUInt16LE(pos + 22) };
      if (fmt.tag === 0xfffe && len >= 26) fmt.tag = b.readUInt16LE(pos + 8 + 24);          // extensible: first word of the SubFormat GUID
    }
    else if (id === 'data') { data = b.subarray(pos + 8, Math.min(b.length, pos + 8 + len)); break; }
    pos += 8 + len + (len & 1);
  }
  if (!fmt || !data) throw 'Unable to decode audio data';
  const bytes = fmt.bits >> 3, n = Math.floor(data.length / (bytes * fmt.ch));
  if (fmt.ch < 1 || !(fmt.tag === 1 || fmt.tag === 3)) throw 'Unable to decode audio data';
  if (fmt.tag === 1 && fmt.bits === 16 && LITTLE_ENDIAN) {
    // 16-bit PCM (what WAV files usually hold) is not decoded here at all: the data chunk goes to the device as it is — interleaved
    // channels and all, half the PCIe bytes of floats — and libwsa converts channel 0 there (x / 32768, exact in fp32)
    const bytes_used = n * fmt.ch * 2;
    const aligned = (data.byteOffset & 1) === 0 ? data : Buffer.from(data.subarray(0, bytes_used));      // an odd offset needs one memcpy
    return { pcm16: new Int16Array(aligned.buffer, aligned.byteOffset, n * fmt.ch), channels: fmt.ch, sampleRate: fmt.rate };
  }
  const pcm = new Float32Array(n);
  for (let i = 0; i < n; i++) {
    const o = i * fmt.ch * bytes;                      // channel 0
    let v;
    if (fmt.tag === 3 && fmt.bits === 32) v = data.readFloatLE(o);
    else if (fmt.tag === 3 && fmt.bits === 64) v = data.readDoubleLE(o);
    else if (fmt.bits === 16) v = data.readInt16LE(o) / 32768;
    else if (fmt.bits === 8) v = (data.readUInt8(o) - 128) / 128;
    else if (fmt.bits === 24) v = data.readIntLE(o, 3) / 8388608;
    else if (fmt.bits === 32) v = data.readInt32LE(o) / 2147483648;
    else throw 'Unable to decode audio data';
    pcm[i] = v;
  }
  return { pcm, sampleRate: fmt.rate };
}

// a clip = { pcm: Float32Array (mono), sampleRate } or { pcm16: Int16Array (interleaved over `channels`), channels, sampleRate }
function to_pcm(source_obj) {
  if (source_obj instanceof Float32Array) return { pcm: source_obj, sampleRate: settings.sample_rate };
  if (source_obj instanceof Int16Array) return { pcm16: source_obj, channels: 1, sampleRate: settings.sample_rate };
  if (source_obj && source_obj.pcm instanceof Float32Array) return { pcm: source_obj.pcm, sampleRate: source_obj.sampleRate || settings.sample_rate };
  if (source_obj && source_obj.pcm16 instanceof Int16Array) return { pcm16: source_obj.pcm16, channels: source_obj.channels || 1, sampleRate: source_obj.sampleRate || settings.sample_rate };
  if (source_obj instanceof ArrayBuffer || Buffer.isBuffer(source_obj) || ArrayBuffer.isView(source_obj)) return decode_wav(source_obj);
  throw 'Invalid audio source';
}

function clip_floats(c) {              // channel 0 of an int16 clip as floats (only batches that mix the two kinds need it)
  if (c.pcm) return c.pcm;
  const ch = c.channels, n = Math.floor(c.pcm16.length / ch), x = new Float32Array(n);
  for (let i = 0; i < n; i++) x[i] = c.pcm16[i * ch] / 32768;
  return x;
}
function clip_slice(c, a, b) {         // samples [a, b) of a clip (bufferSource.start(0, offset, duration))
  if (c.pcm) return { pcm: c.pcm.subarray(a, Math.min(b, c.pcm.length)), sampleRate: c.sampleRate };
  const ch = c.channels, n = Math.floor(c.pcm16.length / ch);
  return { pcm16: c.pcm16.subarray(a * ch, Math.min(b, n) * ch), channels: ch, sampleRate: c.sampleRate };
}
function clip_length(c) { return c.pcm ? c.pcm.length : Math.floor(c.pcm16.length / c.channels); }

// ---- contexts are kept across launches (creating one is cheap, but the planned batch the addon keeps with it is not: GBs of
// work space); a launch with other settings or devices replaces them, shutdown() releases them
let ctx_cache = { key: null, ctxs: [] };
function contexts_for(nat, devs) {
  const key = JSON.stringify([native_config(), devs]);
  if (ctx_cache.key !== key) {
    drop_contexts(nat);
    const ctxs = [];
    try { for (const d of devs) ctxs.push(nat.create(native_config(), d)); }
    catch (e) { for (const c of ctxs) { try { nat.destroy(c); } catch (e2) { /* first error wins */ } } throw e; }
    ctx_cache = { key, ctxs };
  }
  return ctx_cache.ctxs;
}
function drop_contexts(nat) {
  for (const c of ctx_cache.ctxs) { try { nat.destroy(c); } catch (e) { /* still in use by a failed launch's stragglers: left to process exit */ } }
  ctx_cache = { key: null, ctxs: [] };
}
function shutdown() { if (native && !playing) { drop_contexts(native); drop_pipe_contexts(native); } }

// ---- page-locked clip memory (ours; no counterpart in the reference, which hands the browser a file's ArrayBuffer, src/index.js:291).  A clip that
// lies in an ArrayBuffer from allocPinned goes to the GPU by DMA at the PCIe link's rate; a clip in ordinary memory is staged by the runtime
// through its own pinned buffers first (about half that rate).  A host that reads many files reads them into views of such buffers:
//     const ab = fa.allocPinned(n * 2); fs.readSync(fd, Buffer.from(ab), ...); clips.push({ pcm16: new Int16Array(ab), channels: 1, sampleRate })
// LaunchBatch / LaunchAudioNodes accept pinned and ordinary clips alike.  The memory goes back when the ArrayBuffer is collected.
function allocPinned(bytes) {
  const nat = addon();
  const ctxs = contexts_for(nat, settings.devices ? settings.devices.slice() : [settings.device]);
  return nat.allocPinned(ctxs[0], bytes);
}

// the page-locked memory behind an ArrayBuffer of allocPinned goes back now — at a moment the caller chooses, not whenever the garbage collector finalizes the buffer
// (releasing page-locked memory synchronises the device) — and the ArrayBuffer is detached.  No launch may still be reading it.
function freePinned(ab) { return addon().freePinned(ab); }

// ---- module state: one analysis at a time, like the reference's global nodes (ref @B4554)
let playing = false, stop_requested = false;
let labels_per_segment = [];
const open_streams = new Set();

// rows of one clip -> the reference's callback sequence (ref dispatcher P() @B28869)
function dispatch(res, clip, callback, label) {
  const level = settings.output_level, step = settings.window_step / 1e3;
  const a = res.rowOff[clip], b = res.rowOff[clip + 1];
  const feat = (r) => Array.from(res.feat.subarray(r * 53, r * 53 + 53));
  if (level === 5) {
    for (let r = a; r < b; r++) {
      if (stop_requested) return;
      const m = res.meta.subarray(r * 8, r * 8 + 8);
      callback(m[1], label, [m[2] * step, (m[3] + 1) * step], feat(r));                       // ref @B29622, @B31504
    }
  } else if (level === 13 || level === 12) {
    const nf = level === 12 ? 23 : 53;          // level 12: the polynomial coefficients of make_coeffs (ref @B34150)
    let r = a;
    while (r < b) {
      if (stop_requested) return;
      const si = res.meta[r * 8 + 1];
      const times = [], feats = [];
      let cut = false;
      while (r < b && res.meta[r * 8 + 1] === si) {
        const m = res.meta.subarray(r * 8, r * 8 + 8);
        times.push([(m[2] * step).toFixed(3), ((m[3] + 1) * step).toFixed(3)]);              // ref @B31114
        // level 12: a syllable on which numeric.uncmin threw ends the segment's list (ref make_coeffs' try / catch @B34150)
        if (nf === 23 && res.feat[r * 53 + 23] !== 0) cut = true;
        if (!cut) feats.push(nf === 53 ? feat(r) : Array.from(res.feat.subarray(r * 53, r * 53 + nf)));
        r++;
      }
      if (feats.length > 0) callback(si, label, times, feats);                                // ref @B29138 (`p[e].length>0`)
    }
  } else if (level === 11) {
    // utterance features: after every result the 264 histogram bins over everything so far, callback index 0 (ref @B28869)
    for (let k = res.uttOff[clip]; k < res.uttOff[clip + 1]; k++) {
      if (stop_requested) return;
      const m = res.uttMeta.subarray(k * 4, k * 4 + 4);
      callback(0, label, [m[2] * step, (m[3] + 1) * step], Array.from(res.uttFeat.subarray(k * 264, k * 264 + 264)));     // ref Y() @B31330
    }
  } else if (level === 4 || level === 10) {
    // the straightened frames themselves: Float32Array(9) per frame = 3 x (bin, band energy, width), ref @B35074
    const base = res.frameOff[clip];
    const frames = (m) => { const o = [], a0 = (base + m[6]) * 9; for (let d = 0; d < m[7]; d++) o.push(res.formants.slice(a0 + 9 * d, a0 + 9 * d + 9)); return o; };
    let r = a;
    while (r < b) {
      if (stop_requested) return;
      const m = res.meta.subarray(r * 8, r * 8 + 8);
      if (level === 4) { callback(m[1], label, [m[2] * step, (m[3] + 1) * step], frames(m)); r++; continue; }          // ref @B28124
      const si = m[1], times = [], syl = [];
      while (r < b && res.meta[r * 8 + 1] === si) {
        const q = res.meta.subarray(r * 8, r * 8 + 8);
        times.push([(q[2] * step).toFixed(3), ((q[3] + 1) * step).toFixed(3)]);
        syl.push(frames(q)); r++;
      }
      callback(si, label, times, syl);                                                                                  // ref @B27713
    }
  } else if (level === 3) {
    // the ranked raw tracks of every segment, three-argument callback (ref @B28273 `s.push(i)`, @B30132 `b(e, label, s[e])`)
    for (let k = res.segOff[clip]; k < res.segOff[clip + 1]; k++) {
      if (stop_requested) return;
      const tr = tracks_of_segment(res, k);
      if (tr.length > 0) callback(k - res.segOff[clip], label, tr);
    }
  } else {
    throw 'output_level ' + level + ' is not available through this build (3, 4, 5, 10, 11, 12 and 13 are)';
  }
}

// level 3: the 18-field track records of accumulate_fm (ref @B35952; field map SURVEY.md App. A) rebuilt from the per-point
// entries libwsa hands out (include/wsa.h wsa_batch_copy_tracks): every other field is a function of the six point arrays
function tracks_of_segment(res, k) {
  const p0 = res.trackOff[2 * k], p1 = res.trackOff[2 * k + 2], r0 = res.trackOff[2 * k + 1], r1 = res.trackOff[2 * k + 3];
  const per = new Map();
  const f64 = new Float64Array(1), i32 = new Int32Array(f64.buffer);
  for (let q = p0; q < p1; q++) {
    const t = res.trackPoints[8 * q];
    if (!per.has(t)) per.set(t, []);
    per.get(t).push(q);
  }
  const out = [];
  for (let r = r0; r < r1; r++) {
    const P = per.get(res.trackRanked[r]);
    const frames = [], starts = [], ends = [], bins = [], amps = [], energies = [];
    let sE = 0, sEb = 0, sW = 0;
    for (const q of P) {
      const w = res.trackPoints.subarray(8 * q, 8 * q + 8);
      i32[0] = w[2]; i32[1] = w[3];
      const be = f64[0], bin = w[1] & 0xff;
      frames.push(w[6]); starts.push(w[4]); ends.push(w[7]); bins.push(bin); amps.push(w[5] >>> 0); energies.push(be);
      sE += be; sEb += be * bin; sW += w[7] - w[4] + 1;                                      // ref @B36776 [13] [15] [17]
    }
    const h = P.length - 1, pb = bins[h];                                                   // velocity of the last update, ref @B36624
    const vel = h === 0 ? 0 : (h === 1 ? pb - bins[0] : (h === 2 ? ((pb - bins[1]) + (bins[0] - bins[1])) / 2
      : ((pb - bins[h - 1]) + (bins[h - 2] - bins[h - 1]) + (bins[h - 3] - bins[h - 2])) / 3));
    out.push([starts[h], ends[h], frames[h], frames[h], vel, pb, amps[h], frames, starts, ends, bins, amps, energies, sE, P.length, sEb, 0, sW]);
  }
  return out;
}

// contiguous block partition of n clips over k shards (the first n % k shards take one more)
function shard_ranges(n, k) {
  const out = [], q = Math.floor(n / k), r = n % k;
  for (let i = 0, a = 0; i < k; i++) { const b = a + q + (i < r ? 1 : 0); out.push([a, b]); a = b; }
  return out;
}

async function run(clips, callback, labels_of, test_play) {
  const nat = addon();
  if (playing) throw 'Error: Already playing';                                               // ref @B4554
  playing = true; stop_requested = false; labels_per_segment = [];
  try {
    const rates = new Set(clips.map((c) => c.sampleRate));
    if (rates.size !== 1) throw 'All clips of one launch must share a sample rate';
    const fs = clips[0].sampleRate;
    const fs_an = settings.resample_to > 0 ? settings.resample_to : fs;         // the rate the analysis runs at (K0 converts in front, spec RS-1)
    // clips are independent launches (SURVEY.md 8e): shard them contiguously over the configured devices, one context each; every
    // shard is one napi_async_work, i.e. its own worker thread (contexts are not thread-safe, distinct contexts are)
    const devs = settings.devices && clips.length > 1 ? settings.devices.slice(0, clips.length) : [settings.device];
    const shards = shard_ranges(clips.length, devs.length);
    const ctxs = contexts_for(nat, devs);
    const g = nat.geometry(ctxs[0], fs_an);
    const bands = settings.spec_type === 1 ? settings.N_mel_bins : settings.N_fft_bins;
    if (g.bands !== bands) throw 'Bins count mismatch: ' + g.bands + ', ' + bands;              // ref @B8568 check
    // 16-bit clips travel as they are (Int16Array + channel counts); a batch that mixes them with float clips is sent as floats
    const all16 = clips.every((c) => c.pcm16);
    // the feature rows of all shards in one piece: they stay on their devices, one grouped RCCL send / receive over xGMI moves them to the first
    // device (include/wsa.h wsa_gather_rows) and ONE copy brings them to the host; a shard's small tables (segments, offsets) come with its own job
    const gather = (settings.output_level === 5 || settings.output_level === 13) && (settings.gather === null ? devs.length > 1 && new Set(devs).size === devs.length : settings.gather);      // (a rank is a GPU: contexts that share a device are not gathered)
    const job = ([a, b], i) => {
      const part = clips.slice(a, b);
      return all16 ? nat.processBatch(ctxs[i], part.map((c) => c.pcm16), fs, settings.output_level, fs_an, Uint32Array.from(part, (c) => c.channels), gather)
        : nat.processBatch(ctxs[i], part.map(clip_floats), fs, settings.output_level, fs_an, undefined, gather);
    };
    // every shard runs to its end before anything else happens (a context with work in flight must not be touched), then the
    // first failure, if any, is what the launch rejects with
    const settled = await Promise.allSettled(shards.map(job));
    const failed = settled.find((r) => r.status === 'rejected');
    if (failed) { drop_contexts(nat); throw failed.reason; }
    const results = settled.map((r) => r.value);
    if (gather) {
      let all;
      try { all = await nat.gatherRows(ctxs); } catch (e) { drop_contexts(nat); throw e; }
      for (let i = 0, off = 0; i < results.length; i++) {        // a shard's rows = its slice of the gathered tables (views, no copy)
        const k = all.rowsPerRank[i];
        results[i].meta = all.meta.subarray(off * 8, (off + k) * 8); results[i].feat = all.feat.subarray(off * 53, (off + k) * 53);
        off += k;
      }
    }
    // StopAudioNodes while the work was in flight: the reference tears the nodes down at the next frame and resolves (ref @B8851) —
    // nothing is dispatched any more, the launch still resolves
    if (!test_play && callback) {                                                              // ref @B24762: silent when test_play
      for (let i = 0; i < shards.length && !stop_requested; i++)
        for (let c = shards[i][0]; c < shards[i][1] && !stop_requested; c++) dispatch(results[i], c - shards[i][0], callback, labels_of(c));
    }
    return results;
  } finally {
    playing = false;
  }
}

function LaunchAudioNodes(context_source, source_obj = null, callback = null, file_labels = [], offline = false,
  test_play = true, play_offset = null, play_duration = null) {
  return new Promise((resolve, reject) => {
    if (playing) { reject('Error: Already playing'); return; }
    let clip;
    try {
      if (context_source !== 1 || !source_obj) throw 'Invalid audio source';                  // ref @B5698
      clip = to_pcm(source_obj);
      if (play_offset || play_duration) {                                                     // bufferSource.start(0, offset, duration)
        const a = Math.max(0, Math.floor((play_offset || 0) * clip.sampleRate)), len = clip_length(clip);
        const b = play_duration ? Math.min(len, a + Math.floor(play_duration * clip.sampleRate)) : len;
        clip = clip_slice(clip, a, b);
      }
    } catch (e) { reject(typeof e === 'string' ? e : String(e.message || e)); return; }
    run([clip], callback, () => file_labels, test_play).then(() => resolve(true), (e) => reject(typeof e === 'string' ? e : String(e.message || e)));
  });
}

// extension: many independent clips in one GPU batch; callbacks are delivered clip by clip, in
// segment order, as callback(si, labels[clip], seg_time, features, clip_index)
function LaunchBatch(clips, callback = null, labels = [], test_play = false) {
  return new Promise((resolve, reject) => {
    let list;
    try { list = clips.map(to_pcm); } catch (e) { reject(typeof e === 'string' ? e : String(e.message || e)); return; }
    let current = 0;
    const cb = callback ? (si, label, t, f) => callback(si, label, t, f, current) : null;
    const labels_of = (c) => { current = c; return labels[c] || []; };
    run(list, cb, labels_of, test_play).then((rs) => resolve({ rows: rs.reduce((t, r) => t + r.meta.length / 8, 0), segments: rs.reduce((t, r) => t + r.segments.length / 4, 0),
      stageMs: Array.from(rs[0].stageMs), shards: rs.length, stopped: stop_requested }),
      (e) => reject(typeof e === 'string' ? e : String(e.message || e)));
  });
}

// extension: a SEQUENCE of batches, software-pipelined through two contexts on one device (each with its own planned batch and HIP stream, napi/wsa_napi.c):
// while batch k's kernels run, batch k + 1 uploads, and batch k - 1's callbacks are delivered on the JS thread.  The reference's app walks its files one launch
// at a time, the next one when the promise resolves (src/index.js:277-296) — a LaunchBatch per group of files, awaited one after the other, pays upload +
// kernels + callbacks in a row; this keeps the PCIe link busy instead (bench_host.js: i16p_sustained).  Callbacks arrive batch by batch, clip by clip, in segment
// order, as callback(si, labels[batch][clip], seg_time, features, clip_index, batch_index) — for every batch exactly what LaunchBatch delivers for it.
let pipe_cache = { key: null, ctxs: [] };
function pipe_contexts(nat) {
  const key = JSON.stringify([native_config(), settings.device]);
  if (pipe_cache.key !== key) {
    drop_pipe_contexts(nat);
    const ctxs = [];
    try { for (let i = 0; i < 2; i++) ctxs.push(nat.create(native_config(), settings.device)); }
    catch (e) { for (const c of ctxs) { try { nat.destroy(c); } catch (e2) { /* first error wins */ } } throw e; }
    pipe_cache = { key, ctxs };
  }
  return pipe_cache.ctxs;
}
function drop_pipe_contexts(nat) {
  for (const c of pipe_cache.ctxs) { try { nat.destroy(c); } catch (e) { /* a failed launch's straggler still uses it: left to process exit */ } }
  pipe_cache = { key: null, ctxs: [] };
}
async function run_batches(batches, callback, labels, test_play) {
  const nat = addon();
  if (playing) throw 'Error: Already playing';                                               // ref @B4554
  playing = true; stop_requested = false; labels_per_segment = [];
  try {
    const ctxs = pipe_contexts(nat);
    const lists = batches.map((b) => b.map(to_pcm));
    const bands = settings.spec_type === 1 ? settings.N_mel_bins : settings.N_fft_bins;
    const start = (k) => {
      const clips = lists[k];
      const rates = new Set(clips.map((c) => c.sampleRate));
      if (rates.size !== 1) throw 'All clips of one launch must share a sample rate';
      const fs = clips[0].sampleRate, fs_an = settings.resample_to > 0 ? settings.resample_to : fs;
      const g = nat.geometry(ctxs[k % 2], fs_an);
      if (g.bands !== bands) throw 'Bins count mismatch: ' + g.bands + ', ' + bands;          // ref @B8568 check
      const all16 = clips.every((c) => c.pcm16);
      return all16 ? nat.processBatch(ctxs[k % 2], clips.map((c) => c.pcm16), fs, settings.output_level, fs_an, Uint32Array.from(clips, (c) => c.channels), false)
        : nat.processBatch(ctxs[k % 2], clips.map(clip_floats), fs, settings.output_level, fs_an, undefined, false);
    };
    let rows = 0, segments = 0, done = 0;
    let next = lists.length > 0 ? start(0) : null;
    for (let k = 0; k < lists.length; k++) {
      const mine = next;
      // batch k + 1 goes to the other context now (its previous job, batch k - 1, was awaited one turn ago); a failure to start it must not leave batch k unawaited
      let start_err = null;
      next = null;
      if (k + 1 < lists.length && !stop_requested) { try { next = start(k + 1); } catch (e) { start_err = e; } }
      let res;
      try { res = await mine; }
      catch (e) { if (next) { try { await next; } catch (e2) { /* the first failure is reported */ } } drop_pipe_contexts(nat); throw e; }
      if (start_err) { drop_pipe_contexts(nat); throw start_err; }
      rows += res.meta.length / 8; segments += res.segments.length / 4; done++;
      if (!test_play && callback && !stop_requested) {                                        // ref @B24762: silent when test_play
        const lb = labels[k] || [];
        for (let c = 0; c < lists[k].length && !stop_requested; c++) {
          const cb = (si, label, t, f) => callback(si, label, t, f, c, k);
          dispatch(res, c, cb, lb[c] || []);
        }
      }
      if (stop_requested && next) { try { await next; } catch (e) { /* stopping */ } next = null; done++; break; }
    }
    return { rows, segments, batches: done, stopped: stop_requested };
  } finally {
    playing = false;
  }
}
function LaunchBatches(batches, callback = null, labels = [], test_play = false) {
  return new Promise((resolve, reject) => {
    if (!Array.isArray(batches) || batches.some((b) => !Array.isArray(b) || b.length === 0)) { reject('LaunchBatches(batches: clip[][], callback, labels[][][])'); return; }
    run_batches(batches, callback, labels, test_play).then(resolve, (e) => reject(typeof e === 'string' ? e : String(e.message || e)));
  });
}

// extension: n concurrent real-time streams in lock step (the reference's online path — worklet frame ->
// spectrum_push with carried state -> callback as a segment closes, ref @B8752 / @B28869 — for many
// sources at once).  Returns {input, samplesPerStep, push(ctl), close()}: write each stream's next
// samplesPerStep samples into input (a Float32Array over the pinned [n][samplesPerStep] buffer), call
// push(); callbacks fire as callback(si, labels[stream], seg_time, features, stream) for every segment that
// closed in that step.  ctl = Uint8Array of STREAM_ACTIVE | STREAM_START | STREAM_STOP per stream, or
// omitted (all streams start on the first push and stay active).
const STREAM_ACTIVE = 1, STREAM_START = 2, STREAM_STOP = 4;
function StreamOpen(n_streams, sample_rate, callback = null, labels = [], frames_per_step = 1, max_span_frames = 1024) {
  const nat = addon();
  const level = settings.output_level, step = settings.window_step / 1e3;
  if (![3, 4, 5, 10, 11, 12, 13].includes(level)) throw 'output_level ' + level + ' is not available for streams through this build (3, 4, 5, 10, 11, 12 and 13 are)';
  const ctx = nat.create(native_config(), settings.device);
  let st;
  try {
    const g = nat.geometry(ctx, sample_rate);
    const bands = settings.spec_type === 1 ? settings.N_mel_bins : settings.N_fft_bins;
    if (g.bands !== bands) throw 'Bins count mismatch: ' + g.bands + ', ' + bands;              // ref @B8568 check
    st = nat.streamOpen(ctx, n_streams, sample_rate, frames_per_step, max_span_frames);
  } catch (e) { nat.destroy(ctx); throw (typeof e === 'string' ? e : String(e.message || e)); }
  const input = nat.streamInput(st);
  let open = true, started = false;
  const stopped = new Uint8Array(n_streams);          // streams that have had their segment_truncate since their last START
  const seg_seen = new Uint32Array(n_streams);        // level 3: segments a stream has closed since its last START (the callback index, ref @B28273)
  const feat = (res, r) => Array.from(res.feat.subarray(r * 53, r * 53 + 53));
  const deliver = (res) => {
    const rows = res.meta.length / 8;
    if (level === 3) {
      // the ranked raw tracks of every segment that closed (ref @B28273, @B30132: no fourth argument); the stream is the fifth argument as at every level
      for (let k = 0; k < res.segments.length / 4; k++) {
        const s = res.segments[4 * k], tr = callback ? tracks_of_segment(res, k) : [];
        if (tr.length > 0) callback(seg_seen[s], labels[s] || [], tr, undefined, s);
        seg_seen[s]++;
      }
    } else if (callback && level === 11) {
      // utterance features: after every result the 264 histogram bins over everything the source has produced so far, callback index 0 (ref @B28869)
      for (let k = 0; k < res.uttMeta.length / 4; k++) {
        const m = res.uttMeta.subarray(k * 4, k * 4 + 4);
        callback(0, labels[m[0]] || [], [m[2] * step, (m[3] + 1) * step], Array.from(res.uttFeat.subarray(k * 264, k * 264 + 264)), m[0]);   // ref Y() @B31330
      }
    } else if (callback) {
      let r = 0;
      while (r < rows) {
        const s = res.meta[r * 8], si = res.meta[r * 8 + 1];
        // levels 4 / 10: the straightened frames of the row's segment / syllable, Float32Array(9) per frame (ref @B35074)
        const frames = (k) => { const o = []; for (let q = res.formantOff[k]; q < res.formantOff[k + 1]; q++) o.push(res.formants.slice(9 * q, 9 * q + 9)); return o; };
        if (level === 5 || level === 4) {
          callback(si, labels[s] || [], [res.meta[r * 8 + 2] * step, (res.meta[r * 8 + 3] + 1) * step], level === 5 ? feat(res, r) : frames(r), s);   // ref @B29622, @B31504, @B28124
          r++;
        } else if (level === 10) {
          const times = [], syl = [];
          while (r < rows && res.meta[r * 8] === s && res.meta[r * 8 + 1] === si) {
            times.push([(res.meta[r * 8 + 2] * step).toFixed(3), ((res.meta[r * 8 + 3] + 1) * step).toFixed(3)]);
            syl.push(frames(r)); r++;
          }
          callback(si, labels[s] || [], times, syl, s);                                                                        // ref @B27713
        } else {
          // level 13: 53 features per syllable; level 12: the 23 polynomial numbers, the list ending at a syllable on which numeric threw
          const times = [], feats = [];
          let cut = false;
          while (r < rows && res.meta[r * 8] === s && res.meta[r * 8 + 1] === si) {
            times.push([(res.meta[r * 8 + 2] * step).toFixed(3), ((res.meta[r * 8 + 3] + 1) * step).toFixed(3)]);             // ref @B31114
            if (level === 12 && res.feat[r * 53 + 23] !== 0) cut = true;
            if (!cut) feats.push(level === 12 ? Array.from(res.feat.subarray(r * 53, r * 53 + 23)) : feat(res, r));
            r++;
          }
          if (feats.length > 0) callback(si, labels[s] || [], times, feats, s);                                                // ref @B29138 (`p[e].length>0`)
        }
      }
    }
    return { rows, segments: res.segments.length / 4, cuts: res.cuts, cut: (res.flags & 8) !== 0 };   // cut: some stream's span reached max_span_frames in this step (WSA_FLAG_STREAM_CUT)
  };
  const handle = {
    input, samplesPerStep: input.length / n_streams, stopPending: false,
    push(ctl = null) {
      if (!open) throw 'stream closed';
      // StopAudioNodes (ref @B5699 -> disconnect_nodes @B21559): the frame in flight is still pushed, then every source is truncated
      // (segment_truncate, ref @B8851 / @B30757) — the open segments are flushed and reported, the object closes
      const c = new Uint8Array(n_streams);
      for (let i = 0; i < n_streams; i++) {
        c[i] = ctl ? ctl[i] : (STREAM_ACTIVE | (started ? 0 : STREAM_START));
        if (handle.stopPending && !stopped[i]) c[i] |= STREAM_STOP;
        if (c[i] & STREAM_START) { stopped[i] = 0; seg_seen[i] = 0; }
        if (c[i] & STREAM_STOP) stopped[i] = 1;
      }
      started = true;
      const out = deliver(nat.streamStep(st, c));
      if (handle.stopPending) handle.close(false);
      return out;
    },
    // flush = true: sources that have not been stopped yet get their segment_truncate first (a step without new frames), so that the
    // segment a source is in the middle of is reported like the reference reports it when its nodes are disconnected
    close(flush = true) {
      if (!open) return { rows: 0, segments: 0 };
      let out = { rows: 0, segments: 0 };
      if (flush && started && stopped.some((v) => !v)) {
        const c = new Uint8Array(n_streams);
        for (let i = 0; i < n_streams; i++) if (!stopped[i]) { c[i] = STREAM_STOP; stopped[i] = 1; }
        out = deliver(nat.streamStep(st, c));
      }
      open = false; open_streams.delete(handle);
      nat.streamClose(st); nat.destroy(ctx);           // detaches `input`: the pinned buffer is gone
      return out;
    },
  };
  open_streams.add(handle);
  return handle;
}

// ref @B5699 -> disconnect_nodes @B21559 (`0 != audioPlaying && (audioPlaying = -1)`): only a running analysis is affected.  A batch
// launch stops dispatching and resolves; open stream objects flush their sources at their next push() and close.
function StopAudioNodes(reason = 'no reason') {
  if (playing) stop_requested = true;
  for (const h of open_streams) h.stopPending = true;
}

function set_predicted_label_for_segment(si, idx, label) {                                    // ref @B31711
  if (!labels_per_segment[si]) labels_per_segment[si] = [];
  while (labels_per_segment[si].length < idx) labels_per_segment[si].push(-1);
  labels_per_segment[si][idx] = label;
}

module.exports = { configure, LaunchAudioNodes, StopAudioNodes, set_predicted_label_for_segment, LaunchBatch, LaunchBatches,
  StreamOpen, STREAM_ACTIVE, STREAM_START, STREAM_STOP, shutdown, allocPinned, freePinned,
  _settings: settings, _decode_wav: decode_wav, _clip_floats: clip_floats };

// node_runner.js — test helper: drives the JS drop-in (webspeechanalyzer_amd/js/formantanalyzer.js) the
// way src/index.js drives formantanalyzer (configure + LaunchAudioNodes + callback), prints JSON.
// usage: node node_runner.js job.json     job = {level, clips:[{file, kind:"f32"|"wav", fs}], batch:bool}
'use strict';
const fs = require('fs');
const path = require('path');
const fa = require(path.join(__dirname, '..', 'webspeechanalyzer_amd', 'js', 'formantanalyzer.js'));
const { FeatureDB } = require(path.join(__dirname, '..', 'webspeechanalyzer_amd', 'js', 'featuredb.js'));

async function main() {
  const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const cfg = Object.assign({ spec_type: 1, output_level: job.level, f_min: 50, f_max: 4000, N_fft_bins: 256, N_mel_bins: 128,
    window_width: 25, window_step: 25, pause_length: 200, min_seg_length: 50, auto_noise_gate: true, voiced_max_dB: 100,
    voiced_min_dB: 10, pre_norm_gain: 1000, high_f_emph: 0 }, job.config || {});
  fa.configure(cfg);
  const load = (c) => {
    const b = fs.readFileSync(c.file);
    if (c.kind === 'wav') return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength);     // ArrayBuffer, as the app passes it
    return { pcm: new Float32Array(b.buffer, b.byteOffset, b.byteLength / 4), sampleRate: c.fs };
  };
  // job.pinned: the float clips are moved into page-locked memory from allocPinned — "each": a buffer per clip, "slab": views into one buffer
  const pin = (clips) => {
    if (!job.pinned) return clips;
    if (job.pinned === 'each') return clips.map((c) => { const a = new Float32Array(fa.allocPinned(c.pcm.length * 4)); a.set(c.pcm); return { pcm: a, sampleRate: c.sampleRate }; });
    const total = clips.reduce((s, c) => s + c.pcm.length, 0), slab = fa.allocPinned(total * 4);
    let o = 0;
    return clips.map((c) => { const a = new Float32Array(slab, o * 4, c.pcm.length); a.set(c.pcm); o += c.pcm.length; return { pcm: a, sampleRate: c.sampleRate }; });
  };
  const out = [];
  if (job.stream) {
    // extension StreamOpen: all clips as concurrent streams, fed frames_per_step frames at a time
    const clips = job.clips.map(load), n = clips.length;
    const per = clips.map(() => []);
    const st = fa.StreamOpen(n, clips[0].sampleRate, (si, label, t, f, s) => per[s].push([si, label, t, f]), clips.map((c, i) => ['s' + i]), job.stream.frames_per_step);
    const sps = st.samplesPerStep, steps = Math.floor(Math.min(...clips.map((c) => c.pcm.length)) / sps);
    const ctl = new Uint8Array(n);
    let used = steps * sps, closed_error = null, detached = null;
    for (let k = 0; k < steps; k++) {
      // job.stream.stop_at: StopAudioNodes() is called before that push (the frame in flight still goes through, then every source is
      // truncated and the object closes); job.stream.close_at: close() after that many pushes, without any STOP from the caller
      if (job.stream.stop_at === k) fa.StopAudioNodes('test');
      if (job.stream.close_at === k) { used = k * sps; break; }
      try {
        for (let i = 0; i < n; i++) st.input.set(clips[i].pcm.subarray(k * sps, (k + 1) * sps), i * sps);
        const plain = job.stream.stop_at !== undefined || job.stream.close_at !== undefined;
        ctl.fill(fa.STREAM_ACTIVE | (k === 0 ? fa.STREAM_START : 0) | (!plain && k === steps - 1 ? fa.STREAM_STOP : 0));
        st.push(ctl);
      } catch (e) { closed_error = String(e.message || e); break; }
      if (job.stream.stop_at === k) used = (k + 1) * sps;
    }
    st.close();
    try { detached = st.input.length; } catch (e) { detached = 'throws'; }
    process.stdout.write(JSON.stringify({ used, per, closed_error, input_length_after_close: detached }));
    return;
  }
  if (job.batches) {
    // extension LaunchBatches: the clips dealt into job.batches consecutive batches, pipelined through two contexts; per clip the callbacks it received
    const all = pin(job.clips.map(load)), nb = job.batches, per = all.map(() => []);
    const size = Math.ceil(all.length / nb), groups = [], labels = [], first = [];
    for (let k = 0; k < nb; k++) { const a = k * size, b = Math.min(all.length, a + size); if (b > a) { groups.push(all.slice(a, b)); labels.push(all.slice(a, b).map((c, i) => ['clip' + (a + i)])); first.push(a); } }
    const order = [];
    const info = await fa.LaunchBatches(groups, (si, label, t, f, clip, batch) => { per[first[batch] + clip].push([si, label, t, f]); order.push(batch); }, labels);
    process.stdout.write(JSON.stringify({ per, info, in_order: order.every((b, i) => i === 0 || order[i - 1] <= b) }));
    return;
  }
  if (job.batch) {
    const per = job.clips.map(() => []);
    const info = await fa.LaunchBatch(pin(job.clips.map(load)), (si, label, t, f, clip) => per[clip].push([si, label, t, f]), job.clips.map((c, i) => ['clip' + i]));
    if (job.want_info) { process.stdout.write(JSON.stringify({ per, info })); return; }
    out.push(...per);
  } else {
    // job.featuredb: collect like the app does (src/index.js:36 call_backed -> StoreFeatures) and export the DB files
    const db = job.featuredb ? new FeatureDB() : null;
    const collect = db ? db.callback(job.level, 1) : null;
    for (const c of job.clips) {
      const calls = [];
      const busy = [];
      const lbl = db ? [path.basename(c.file)] : ['lbl'];
      if (job.stop_before) fa.StopAudioNodes('nothing is playing');           // no effect: only a running analysis is stopped (ref @B21559)
      const p = fa.LaunchAudioNodes(1, load(c), (si, label, t, f) => {
        calls.push(f === undefined ? [si, label, t] : [si, label, t, JSON.parse(JSON.stringify(f))]); if (collect) collect(si, label, t, f);
        if (job.stop_after && calls.length === job.stop_after) fa.StopAudioNodes('test');
      }, lbl, true, false);
      if (job.stop_in_flight) fa.StopAudioNodes('test');                      // while the GPU work is in flight: nothing is dispatched, the launch resolves
      if (job.check_busy) await fa.LaunchAudioNodes(1, load(c), null, [], true, true).catch((e) => busy.push(e));
      const r = await p;
      out.push({ resolved: r, calls, busy });
    }
    if (db) { process.stdout.write(JSON.stringify({ clips: out, db_json: db.Download_DB(1, 'JSON'), db_csv: db.Download_DB(1, 'CSV') })); return; }
  }
  process.stdout.write(JSON.stringify(out));
}
main().catch((e) => { console.error('ERROR', e); process.exit(1); });

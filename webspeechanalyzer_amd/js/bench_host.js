// bench_host.js — throughput THROUGH THE JAVASCRIPT HOST: LaunchBatch(clips) -> N-API -> libwsa -> callbacks, PCIe and
// marshalling included (the number a Node application sees; bench.py measures the HBM-resident kernels).
//   node bench_host.js [clips=256] [seconds=10] [level=5] [kind=i16|f32|i16p]   (i16: 16-bit PCM clips, what WAV files hold; f32: Float32Array clips;
//   i16p: 16-bit clips as views into ONE page-locked slab from allocPinned, as a host that reads its files into such a slab holds them;
//   i16ps: SUSTAINED — `batches` (7th argument, default 12) such batches back to back through LaunchBatches, which pipelines them over two contexts: batch k + 1 uploads
//   while batch k computes and batch k - 1's callbacks run; two slabs are filled in turn, as a host that reads the next files while the GPU works would do)
'use strict';
const fa = require('./formantanalyzer.js');

function synth(n, fs, seed) {          // harmonic complex with moving resonances, syllabic envelope and pauses
  let s = seed >>> 0; const rnd = () => ((s = (s * 1664525 + 1013904223) >>> 0) / 4294967296);
  const x = new Float32Array(n), f0 = 100 + 120 * rnd();
  const F = [300 + 600 * rnd(), 900 + 1500 * rnd(), 2400 + 1100 * rnd()], rate = 3 + 3 * rnd(), ph = rnd() * 6.28;
  for (let i = 0; i < n; i++) {
    const t = i / fs, env = Math.max(0, Math.sin(2 * Math.PI * rate * t + ph)) * (Math.sin(2 * Math.PI * 0.6 * t + ph) > -0.5 ? 1 : 0);
    let v = 0;
    for (let h = 1; h * f0 < 3800; h++) {
      const f = h * f0; let g = 0;
      for (let k = 0; k < 3; k++) { const d = (f - F[k] * (1 + 0.1 * Math.sin(2 * Math.PI * 0.9 * t + k))) / (90 + 40 * k); g += Math.exp(-d * d); }
      v += g * Math.sin(2 * Math.PI * f * t);
    }
    x[i] = 0.12 * v * env + 0.003 * (rnd() - 0.5);
  }
  return x;
}

async function main() {
  const nclips = parseInt(process.argv[2] || '256'), seconds = parseFloat(process.argv[3] || '10'), level = parseInt(process.argv[4] || '5'), kind = process.argv[5] || 'i16';
  const fs = 16000, ns = Math.floor(seconds * fs);
  const distinct = Math.min(nclips, 16), base = [];
  for (let i = 0; i < distinct; i++) base.push(synth(ns, fs, 1234 + i));
  const base16 = base.map((x) => Int16Array.from(x, (v) => Math.max(-32768, Math.min(32767, Math.round(v * 32768)))));
  fa.configure({ spec_type: 1, output_level: level, f_min: 50, f_max: 4000, N_fft_bins: 256, N_mel_bins: 128, window_width: 25, window_step: 25,
    pause_length: 200, min_seg_length: 50, auto_noise_gate: true, voiced_max_dB: 100, voiced_min_dB: 10, pre_norm_gain: 1000, high_f_emph: 0 });
  const clips = [];
  const slab = kind === 'i16p' ? fa.allocPinned(nclips * ns * 2) : null;       // one page-locked slab, the clips views into it: back to back, they travel as one DMA
  for (let i = 0; i < nclips; i++) {
    if (kind === 'i16p') { const a = new Int16Array(slab, i * ns * 2, ns); a.set(base16[i % distinct]); clips.push({ pcm16: a, channels: 1, sampleRate: fs }); }
    else clips.push(kind === 'i16' ? { pcm16: base16[i % distinct], channels: 1, sampleRate: fs } : { pcm: base[i % distinct], sampleRate: fs });
  }
  let calls = 0;
  const cb = () => { calls++; };
  if (kind === 'i16ps') {
    const nb = parseInt(process.argv[6] || '12');
    const slabs = [fa.allocPinned(nclips * ns * 2), fa.allocPinned(nclips * ns * 2)];
    const sets = slabs.map((sl) => { const cs = []; for (let i = 0; i < nclips; i++) { const a = new Int16Array(sl, i * ns * 2, ns); a.set(base16[i % distinct]); cs.push({ pcm16: a, channels: 1, sampleRate: fs }); } return cs; });
    const seq = (n) => Array.from({ length: n }, (_, k) => sets[k % 2]);
    await fa.LaunchBatches(seq(2).map((b) => b.slice(0, Math.min(8, nclips))), cb, []);      // warm-up (library load, first launches)
    await fa.LaunchBatches(seq(2), cb, []);                                                  // both contexts plan their batch
    calls = 0;
    // one at a time, awaited (what a loop of LaunchBatch calls gives), then the pipelined sequence
    let t0 = process.hrtime.bigint();
    let rows1 = 0;
    for (let k = 0; k < 4; k++) rows1 += (await fa.LaunchBatch(sets[k % 2], cb, [])).rows;
    const serial = Number(process.hrtime.bigint() - t0) / 1e9 / 4;
    let best = Infinity, rows = 0;
    for (let r = 0; r < 3; r++) {
      t0 = process.hrtime.bigint();
      const res = await fa.LaunchBatches(seq(nb), cb, []);
      best = Math.min(best, Number(process.hrtime.bigint() - t0) / 1e9 / nb); rows = res.rows / nb;
    }
    const frames = nclips * (Math.floor((ns - 400) / 400) + 1);
    console.log(JSON.stringify({ metric: '53-feat frames/sec through the Node host, sustained over back-to-back batches (PCIe + N-API inclusive)', value: frames / best, unit: 'frames/s',
      clips: nclips, seconds, level, kind, frames, rows, batches: nb, best_s: best, one_at_a_time_s: serial, rows_equal: Math.round(rows) === rows1 / 4, node: process.version }));
    fa.shutdown();
    return;
  }
  await fa.LaunchBatch(clips.slice(0, Math.min(8, nclips)), cb, []);             // warm-up (library load, first launches)
  calls = 0;
  const reps = 5; let best = Infinity, rows = 0;
  for (let r = 0; r < reps; r++) {
    const t0 = process.hrtime.bigint();
    const res = await fa.LaunchBatch(clips, cb, []);
    const dt = Number(process.hrtime.bigint() - t0) / 1e9;
    best = Math.min(best, dt); rows = res.rows;
  }
  const frames = nclips * (Math.floor((ns - 400) / 400) + 1);
  console.log(JSON.stringify({ metric: '53-feat frames/sec through the Node host (PCIe + N-API inclusive)', value: frames / best, unit: 'frames/s',
    clips: nclips, seconds, level, kind, frames, rows, callbacks_per_run: calls / reps, best_s: best, node: process.version }));
  fa.shutdown();
}
main().catch((e) => { console.error('ERROR', e); process.exit(1); });

// run.js — command-line driver of the JS oracle (test infrastructure).  node run.js job.json
//   job = {mode:"fe",  pcm, fs, settings, out}              PCM (raw f32 file) -> u32 frames (raw file)
//       | {mode:"be",  spectra, frames, cfg}                u32 frames (raw file) -> callbacks JSON on stdout
//       | {mode:"e2e", pcm | wav, fs, settings}             whole path -> callbacks JSON on stdout (settings.resample_to: RS-1 conversion first)
//       | {mode:"rs",  pcm, fs, to, out}                    RS-1: PCM (raw f32 file) at fs -> raw f32 file at `to`
//       | {mode:"time", clips, seconds, fs, settings}       synthetic clips, prints {frames, ms}
'use strict';
const fs = require('fs');
const o = require('./wsa_oracle.js');
const num = (x) => (Number.isFinite(x) ? x : String(x));
const cbJSON = (cbs, level) => level === 3 ? cbs : cbs.map((c) => [c[0], c[1], c[2], (level === 5 || level === 11) ? Array.from(c[3], num) : c[3].map((v) => Array.from(v, num))]);
const f32 = (file) => { const b = fs.readFileSync(file); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };

function decodeWav(buf) {          // PCM16/24/32 + float32 RIFF, channel 0 (ref: decodeAudioData(...).getChannelData(0), dist/main.js:2 @B5391)
  const dv = new DataView(buf.buffer, buf.byteOffset, buf.byteLength);
  let p = 12, fmt = null;
  while (p + 8 <= dv.byteLength) {
    const id = String.fromCharCode(dv.getUint8(p), dv.getUint8(p + 1), dv.getUint8(p + 2), dv.getUint8(p + 3)), sz = dv.getUint32(p + 4, true);
    if (id === 'fmt ') fmt = { tag: dv.getUint16(p + 8, true), ch: dv.getUint16(p + 10, true), rate: dv.getUint32(p + 12, true), bits: dv.getUint16(p + 22, true) };
    else if (id === 'data') {
      const bps = fmt.bits / 8, n = Math.floor(Math.min(sz, dv.byteLength - p - 8) / (bps * fmt.ch)), out = new Float32Array(n);
      for (let i = 0; i < n; i++) {
        const q = p + 8 + i * bps * fmt.ch;
        out[i] = fmt.tag === 3 ? dv.getFloat32(q, true) : fmt.bits === 16 ? dv.getInt16(q, true) / 32768 : fmt.bits === 24 ? ((dv.getInt8(q + 2) << 16) | (dv.getUint8(q + 1) << 8) | dv.getUint8(q)) / 8388608 : fmt.bits === 32 ? dv.getInt32(q, true) / 2147483648 : (dv.getUint8(q) - 128) / 128;
      }
      return { pcm: out, fs: fmt.rate };
    }
    p += 8 + sz + (sz & 1);
  }
  throw new Error('Unable to decode audio data');
}

const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const settings = Object.assign({}, o.DEFAULTS, job.settings || {});
if (job.mode === 'fe') {
  const fe = new o.FrontEnd(Object.assign({ fs: job.fs }, settings));
  const s = fe.run(f32(job.pcm));
  fs.writeFileSync(job.out, Buffer.from(s.buffer, s.byteOffset, s.byteLength));
  process.stdout.write(JSON.stringify({ nfft: fe.nfft, win: fe.win, hop: fe.hop, bands: fe.bands, kmax: fe.kmax, frames: fe.n_frames(f32(job.pcm).length), bins_hz: Array.from(fe.bins_hz) }));
} else if (job.mode === 'be') {
  const b = fs.readFileSync(job.spectra), spec = new Uint32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength));
  const sg = o.runBackend(spec, job.frames, job.cfg);
  process.stdout.write(JSON.stringify({ segments_ci: sg.segs.map((s) => [s.start, s.len]), flags: sg.segs.map((s) => s.flag), callbacks: cbJSON(sg.callbacks(), job.cfg.level) }));
} else if (job.mode === 'e2e') {
  const src = job.wav ? decodeWav(fs.readFileSync(job.wav)) : { pcm: f32(job.pcm), fs: job.fs };
  const r = o.analyze(src.pcm, src.fs, settings);
  process.stdout.write(JSON.stringify({ fs: src.fs, samples: src.pcm.length, nfft: r.fe.nfft, frames: r.spec.length / r.fe.bands, segments_ci: r.sg.segs.map((s) => [s.start, s.len]),
    flags: r.sg.segs.map((s) => s.flag), callbacks: cbJSON(r.sg.callbacks(), settings.output_level) }));
} else if (job.mode === 'rs') {
  const y = o.resample(f32(job.pcm), job.fs, job.to);
  fs.writeFileSync(job.out, Buffer.from(y.buffer, y.byteOffset, y.byteLength));
  process.stdout.write(JSON.stringify({ samples: y.length }));
} else if (job.mode === 'time') {
  const clips = job.files.map(f32);
  const t0 = process.hrtime.bigint(); let frames = 0, rows = 0;
  for (const pcm of clips) { const r = o.analyze(pcm, job.fs, settings); frames += r.fe.n_frames(pcm.length); rows += r.sg.callbacks().length; }
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  process.stdout.write(JSON.stringify({ frames, rows, ms }));
}

// ref_driver.js — fixture generator helper. TEST INFRASTRUCTURE, build-container only.
//
// Loads the reference's own segmenter / formant-tracker / feature code (webpack module 584 =
// formantanalyzer@1.1.6) out of /root/reference/dist/main.js AT RUN TIME (nothing of the bundle is
// copied into this repository), drives it with u32 spectrum frames read from a binary file and
// writes what the reference produced (segments_ci, syllables_ci, callbacks) as JSON.
// /root/reference does not exist on the GPU box: this script is only ever run by
// tests/golden/gen/make_golden.py inside the build container; its outputs are the committed
// fixtures under tests/golden/.
//
// Recipe follows SURVEY.md §8(c): slice bytes [100,114174) of dist/main.js, expose the inner
// webpack require, shim `window`/`document`, push ONE frame per macrotask (the reference resets
// segment state in a Promise microtask, dist/main.js:2 @B26783), wait >10 ms after
// segment_truncate (@B30757).
//
// usage: node ref_driver.js job.json out.json
//   job.json = {"bundle": "/root/reference/dist/main.js",
//               "clips": [{"spectra": "x.bin", "frames": F, "bands": B, "level": 5,
//                          "window_step": 25, "pause_length": 200, "min_seg_length": 50,
//                          "auto_noise_gate": true, "voiced_max_dB": 100, "voiced_min_dB": 10,
//                          "trace": false}, ...]}
'use strict';
const fs = require('fs');

function load_reference(bundle_path) {
  const b = fs.readFileSync(bundle_path);
  let src = b.slice(100, 114174).toString('latin1');
  src = src.slice(src.indexOf('function(module)'));
  if (src.indexOf('n(n.s=1)') < 0) throw new Error('bundle layout changed');
  src = src.replace('n(n.s=1)', '(globalThis.__fa_require=n,n(n.s=1))');
  // per-frame state trace hook (end of the frame loop body in D(), @B26985)
  const hook = 'o.c_ci++,f=null';
  if (src.indexOf(hook) < 0) throw new Error('trace hook site not found');
  src = src.replace(hook,
    '(globalThis.__fa_trace&&globalThis.__fa_trace(o,y,v,n,p,h,d,g)),' + hook);
  // the match score `_` (@B37340) is module-private: hand it out through a global (function-level fixture G2)
  const score_decl = 'function _(e,t,n,r,a,i,o,l){let s=0;';
  if (src.indexOf(score_decl) < 0) throw new Error('score function not found');
  src = src.replace(score_decl, 'globalThis.__fa_score=function(){return _.apply(null,arguments)};' + score_decl);
  global.window = { setTimeout: setTimeout, screen: {} };
  global.document = { getElementById: () => ({}) };
  const mod = { exports: {} };
  (0, eval)('(' + src + ')')(mod);
  const req = globalThis.__fa_require;
  return { seg: req(3), fm: req(4), stats: req(0) };
}

function enc(x) {  // JSON-safe deep copy; non-finite numbers become strings
  if (typeof x === 'number') return Number.isFinite(x) ? x : String(x);
  if (x === null || x === undefined) return null;
  if (typeof x === 'string' || typeof x === 'boolean') return x;
  if (ArrayBuffer.isView(x)) return Array.from(x, enc);
  if (Array.isArray(x)) return x.map(enc);
  return String(x);
}

const tick = () => new Promise(r => setImmediate(r));
const sleep = ms => new Promise(r => setTimeout(r, ms));

async function run_clip(ref, c) {
  const raw = fs.readFileSync(c.spectra);
  const all = new Uint32Array(raw.buffer, raw.byteOffset, c.frames * c.bands);
  const calls = [];
  const cb = function () { calls.push(enc(Array.prototype.slice.call(arguments))); };
  const trace = [];
  globalThis.__fa_trace = c.trace ? (o, y, v, n, p, h, d, g) => {
    trace.push([o.c_ci, o.c_started, o.no_fm_segs, y, v, n, p, h, d, g]);
  } : null;
  const log = console.log; console.log = () => {};   // "seg_size ... ignored" chatter
  try {
    await ref.seg.reset_segmentation(c.level, c.bands, 200, c.window_step, c.pause_length,
      c.min_seg_length, c.auto_noise_gate, c.voiced_max_dB, c.voiced_min_dB, cb, false, []);
    for (let f = 0; f < c.frames; f++) {
      // copy: the reference keeps/mutates nothing of the frame, but a fresh array per frame is
      // what the worklet delivers (@B8568)
      ref.seg.spectrum_push(Uint32Array.from(all.subarray(f * c.bands, (f + 1) * c.bands)), f);
      await tick();
    }
    ref.seg.segment_truncate();
    await sleep(25);
    await tick();
  } finally { console.log = log; }
  const segs = [];
  for (let i = 0; ; i++) { const s = ref.seg.get_segments_ci(i); if (!s) break; segs.push(enc(s)); }
  const syls = [];
  if (c.level >= 10) {
    for (let i = 0; i < segs.length; i++) {
      try { syls.push(enc(ref.seg.get_syllables_ci(i))); } catch (e) { syls.push(null); }
    }
  }
  const out = { segments_ci: segs, syllables_ci: syls, callbacks: calls };
  if (c.trace) out.trace = enc(trace);
  return out;
}

// function-level cases: formant_features (@B32369) on hand-built [len][9] Float32 frames
function run_fn(ref, c) {
  if (c.fn === 'formant_features') {
    ref.fm.clear_fm();                       // module accumulators c = s = 0 -> feature[2] = NaN
    const fr = c.fr.map(r => Float32Array.from(r));
    return enc(ref.fm.formant_features(fr, c.ctx_max, c.floor));
  }
  if (c.fn === 'score') {                     // args: [gap, dist, track length, track bin, peak bin, track amp, peak amp, velocity] per row
    const buf = Buffer.alloc(8);
    return c.args.map(a => { buf.writeDoubleBE(globalThis.__fa_score.apply(null, a)); return buf.toString('hex'); });
  }
  throw new Error('unknown fn ' + c.fn);
}

async function main() {
  const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const ref = load_reference(job.bundle);
  const results = [];
  for (const c of job.clips) results.push(c.fn ? run_fn(ref, c) : await run_clip(ref, c));
  fs.writeFileSync(process.argv[3], JSON.stringify({ node: process.version, results }));
}
main().catch(e => { console.error(e); process.exit(1); });

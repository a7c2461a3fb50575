// featuredb_scenarios.js — the scenarios both sides of the feature-DB parity check run (TEST INFRASTRUCTURE).
// `api` is either the reference's own functions (tests/golden/gen/make_featuredb_golden.js, build container only) or
// webspeechanalyzer_amd/js/featuredb.js wrapped to the same shape (tests/js/featuredb_check.js):
//   api.reset(label_heads)  api.callback(level, db_id) -> fn(si, label, time, incoming)   api.download(db, type, only_selected)
//   api.load_json(db, text) api.load_labels(text)  api.update_true(seg, label, val, clear) api.update_pred(seg, label, val)
// Inputs: the callbacks the reference produced for the committed back-end fixtures (tests/golden/backend_expected.json).
'use strict';

function as_incoming(level, feats) {
  // the library hands level 10 syllables as arrays of Float32Array(9) frames (ref dist/main.js:2 @B35074: fp32 storage)
  if (level === 10) return feats.map(syl => syl.map(fr => Float32Array.from(fr)));
  return feats;
}

function feed(api, cases, level, db_id, names) {
  let n = 0;
  cases.filter(c => c.level === level && c.callbacks.length > 0).slice(0, names.length).forEach((c, ci) => {
    const cb = api.callback(level, db_id);
    for (const call of c.callbacks) {
      // fixture callback = [si, label, time, features] (tests/golden/gen/ref_driver.js)
      cb(call[0], [names[ci]], call[2], as_incoming(level, JSON.parse(JSON.stringify(call[3]))));
      n++;
    }
  });
  return n;
}

function run(api0, golden_cases) {
  // an export that throws in the reference (e.g. CSV rows of unlabeled samples behind a labeled first one,
  // ref src/localstore.js:955) has to throw the same kind of error here
  const api = Object.assign({}, api0, { download(db, type, sel) { try { return api0.download(db, type, sel); } catch (e) { return 'THROWS ' + e.name; } } });
  const out = {};
  const names = ['0001_01_F_N.wav', 'clip two.wav', 'c.wav', 'd.wav'];
  // 1. plain collection at every level the app stores, no labels
  for (const level of [5, 13, 12, 11, 10]) {
    api.reset([[], []]);
    const n = feed(api, golden_cases, level, 1, names);
    out['L' + level + '_calls'] = n;
    out['L' + level + '_json'] = api.download(1, 'JSON', false);
    out['L' + level + '_csv'] = api.download(1, 'CSV', false);
  }
  // 2. labels: index file, label heads, true / predicted labels set, selection
  const heads = [[{ emotion: ['A', 'H'] }, { sex: ['*'] }], ['V', 'A']];
  api.reset(heads);
  api.load_labels(JSON.stringify([
    { i: '0001_01_F_N.wav', emo: 'A', sex: 'F', spkr: 1, U: 0, E: 1, R: 2, V: 0.25, A: 0.5, D: 0.75 },
    { i: 'c.wav', emo: 'S', sex: 'M', spkr: 2, U: 1, E: 0, R: 0, V: 0.1, A: 0.2, D: 0.3 }]));
  feed(api, golden_cases, 5, 7, names);
  api.update_true('7#clip two.wav#0', 'emotion', 'H', false);
  api.update_true('7#clip two.wav#0', 'V', '40', false);
  api.update_true('7#0001_01_F_N.wav#0', 'emotion', 'H', true);       // clears a differing label
  api.update_pred('7#0001_01_F_N.wav#0', 'emotion', 'A');
  api.update_pred('7#clip two.wav#0', 'A', 0.3);
  out.lab_json_early = api.download(7, 'JSON', false);            // the updates above ran before the app had read its label heads
  out.lab_csv_partial = api.download(7, 'CSV', false);            // 'd.wav' has no labels: the reference's CSV writer throws
  api.update_true('7#clip two.wav#0', 'emotion', 'H', false);     // the same updates with the heads in effect
  api.update_true('7#clip two.wav#0', 'V', '40', false);
  api.update_true('7#0001_01_F_N.wav#0', 'emotion', 'H', true);
  api.update_pred('7#0001_01_F_N.wav#0', 'emotion', 'A');
  api.update_pred('7#clip two.wav#0', 'A', 0.3);
  api.update_true('7#d.wav#0', 'emotion', 'A', false);
  for (const k of api.keys(7)) if (k.indexOf('#d.wav#') > 0 || k.indexOf('#clip two.wav#') > 0) api.update_true(k, 'sex', 'F', false);
  out.lab_json = api.download(7, 'JSON', false);
  out.lab_csv = api.download(7, 'CSV', false);
  out.lab_json_sel = api.download(7, 'JSON', true);
  out.lab_csv_sel = api.download(7, 'CSV', true);
  for (const k of api.keys(7)) api.update_pred(k, 'emotion', 'H');       // every sample labeled and predicted: the CSV writer gets through
  out.full_json = api.download(7, 'JSON', false);
  out.full_csv = api.download(7, 'CSV', false);
  out.full_csv_sel = api.download(7, 'CSV', true);
  // 3. import of the exported file into another DB id, export again
  api.load_json(9, out.full_json);
  out.reload_json = api.download(9, 'JSON', false);
  out.reload_csv = api.download(9, 'CSV', false);
  // 4. an empty DB
  out.empty = api.download(1234, 'JSON', false);
  return out;
}

module.exports = { run };

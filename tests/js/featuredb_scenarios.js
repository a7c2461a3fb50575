// featuredb_scenarios.js — the scenarios both sides of the feature-DB file-format check run (TEST INFRASTRUCTURE).
// `api` is either the reference app's own storage / export code (tests/golden/gen/make_featuredb_golden.js, build
// container only) or webspeechanalyzer_amd/js/featuredb.js wrapped to the same shape (tests/js/featuredb_check.js):
//   api.reset()   api.callback(level, db_id) -> fn(si, label, time, incoming)   api.download(db, 'JSON' | 'CSV')
//   api.load_json(db, text)
// Inputs: the callbacks the reference produced for the committed back-end fixtures (tests/golden/backend_expected.json)
// and hand-written data_<db>.json texts (labels are DATA of the file format; editing them is the app's UI and not covered).
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

function vec(n, a) { const v = []; for (let i = 0; i < n; i++) v.push(i === 3 ? 0.1 + a : (i * 7 + a) % 11 - 2.5); return v; }

function run(api0, golden_cases) {
  // an export that throws in the reference (CSV rows of unlabeled samples behind a labeled first one,
  // ref src/localstore.js:955) has to throw the same kind of error here
  const api = Object.assign({}, api0, { download(db, type) { try { return api0.download(db, type); } catch (e) { return 'THROWS ' + e.name; } } });
  const out = {};
  const names = ['0001_01_F_N.wav', 'clip two.wav', 'c.wav', 'd.wav'];
  // 1. plain collection at every level the app stores
  for (const level of [5, 13, 12, 11, 10]) {
    api.reset();
    const n = feed(api, golden_cases, level, 1, names);
    out['L' + level + '_calls'] = n;
    out['L' + level + '_json'] = api.download(1, 'JSON');
    out['L' + level + '_csv'] = api.download(1, 'CSV');
  }
  // 2. the same file stored twice keeps its place and takes the new values; a vector of the wrong length is refused
  api.reset();
  feed(api, golden_cases, 5, 2, names);
  const again = api.callback(5, 2);
  again(0, [names[0]], [1.5, 0.25], vec(53, 1));
  again(1, [names[0]], [2.5, 0.5], vec(52, 2));                     // 52 numbers: not a level-5 vector
  again(0, ['new.wav'], [0, 0.125], vec(53, 3));
  out.restore_json = api.download(2, 'JSON');
  out.restore_csv = api.download(2, 'CSV');
  // 3. import of a labeled data file (labels are part of the file format), export in both formats, import of the export
  const T = (emo, sex, v, a) => [{ emotion: emo, sex: sex }, { V: v, A: a }];
  const labeled = [
    { file: 'a.wav', seg: '0', time: [0.5, 1.25], features: vec(53, 0), origin: null, true: T('H', 'F', 40, 0.5), pred: [{ emotion: 'A' }, { A: 0.3 }] },
    { file: 'a.wav', seg: '1', time: [2, 0.75], features: vec(53, 1), origin: [{ x: 1 }, {}], true: T('A', 'F', 10, 0.25), pred: [{ emotion: 'H' }, {}] },
    { file: 'b c.wav', seg: '0.01', time: ['0.125', '0.500'], features: vec(53, 2), origin: null, true: [{ emotion: null }, {}], pred: [{}, { A: 1 }] },
  ];
  api.reset();
  api.load_json(9, JSON.stringify(labeled));
  out.labeled_json = api.download(9, 'JSON');
  out.labeled_csv = api.download(9, 'CSV');
  api.load_json(10, out.labeled_json);
  out.relabeled_json = api.download(10, 'JSON');
  out.relabeled_csv = api.download(10, 'CSV');
  // a later sample without the labels of the first one: the CSV writer cannot produce the line
  api.reset();
  api.load_json(11, JSON.stringify(labeled.concat([{ file: 'z.wav', seg: '0', time: [0, 1], features: vec(53, 4), origin: null, true: null, pred: null }])));
  out.partial_json = api.download(11, 'JSON');
  out.partial_csv = api.download(11, 'CSV');
  // label fields that are not pairs of plain objects are dropped on import
  api.reset();
  api.load_json(12, JSON.stringify([{ file: 'q.wav', seg: '3', time: [1, 2], features: vec(53, 5), true: [{ e: 1 }, 7], pred: 'x' },
                                    { file: 'q.wav', seg: '3', time: [3, 4], features: vec(53, 6) }]));
  out.loose_json = api.download(12, 'JSON');
  out.loose_csv = api.download(12, 'CSV');
  // 4. texts that are not a data file leave the DB empty; an empty DB writes nothing
  api.reset();
  api.load_json(13, JSON.stringify([{ seg: '0', features: [1] }]));
  api.load_json(13, JSON.stringify({ file: 'x', features: [] }));
  out.invalid = api.download(13, 'JSON');
  out.empty = api.download(1234, 'JSON');
  out.empty_csv = api.download(1234, 'CSV');
  return out;
}

module.exports = { run };

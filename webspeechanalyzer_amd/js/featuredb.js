// featuredb.js — reader / writer of the feature-DB FILE FORMAT the WebSpeechAnalyzer app exchanges with its labeling and
// NN pages (`data_<db>.json`, `data_<db>.csv`), for the rows this library's callbacks deliver (SURVEY.md §8f item 3).
//
// Written from the format, not from the app's functions.  The format (observed on files the reference app writes, cited
// where the app defines it):
//
//   sample      one stored feature vector.  Identity = (db id, file name, part tag); the app's storage key for it is
//               "<db>#<file>#<part>" (ref src/localstore.js:41-42).  part tag = the callback index `si` as a JS number
//               string; syllable k of a callback is `si + k / 100` (ref src/index.js:49, :65, :87); utterance rows
//               (output_level 11) always use 0 (ref src/index.js:41).  Storing an identity again replaces the sample in place.
//   vector      plain numbers; its length is fixed by the output level (ref src/localstore.js:7): 5 -> 53, 10 -> 9,
//               11 -> 264, 12 -> 23, 13 -> 53.  A vector of another length is refused.
//               Level 10 delivers frames, not a vector: the stored row is the per-column mean of the syllable's
//               Float32Array(9) frames, summed in fp32 in frame order and divided (in fp32) only where the sum is
//               non-zero (ref src/index.js:76-86).
//   time        what the callback got: [t0, duration] numbers (levels 5, 11) or two toFixed(3) strings (10, 12, 13).
//   labels      `true` / `pred`: null, or a pair [categorical {name: value}, ordinal {name: value}] — carried through
//               import and export untouched.  `origin` (labels looked up in the app's label index by file name) is
//               always null here: the label index, label editing and label-based selection are UI state of the app.
//   JSON file   JSON.stringify of [{file, seg, time, features, origin, true, pred}, ...] in insertion order
//               (ref src/localstore.js:877-887).  An empty DB writes no file (null).
//   CSV file    header `file,seg,t0,td,` + `true_<name>,` for the categorical then ordinal names of the FIRST sample's
//               `true` pair + `pred_<name>,` likewise + `x0,` .. `x<n-1>,` (n = length of the first vector), CRLF; one
//               line per sample with the same cells, every cell followed by a comma, CRLF (ref src/localstore.js:895-975).
//               The first sample decides the label columns: a later sample without the pair the header was built from
//               cannot be written — the app's writer dies with a TypeError there, and so does this one.
//   import      the JSON file above: an array whose first element has `file` and `features`; `true` / `pred` are kept
//               only when they are pairs of plain objects (ref src/localstore.js:1125-1175).
// Parity: tests/golden/featuredb_expected.json holds the file texts the reference app's own code wrote for the scenarios
// of tests/js/featuredb_scenarios.js; tests/js/featuredb_check.js compares byte for byte.
'use strict';

const VECTOR_LENGTH = { 5: 53, 10: 9, 11: 264, 12: 23, 13: 53 };

const plain_object = (v) => Boolean(v) && typeof v === 'object' && v.constructor === Object;
const label_pair = (v) => (Array.isArray(v) && plain_object(v[0]) && plain_object(v[1]) ? v : null);

class Sample {
  constructor(file, part, time, vector, truth, guess) {
    this.file = file; this.part = part; this.time = time; this.vector = vector;
    this.truth = truth || null; this.guess = guess || null;
  }
  // the object the JSON file holds for this sample (key order is part of the format)
  wire() { return { file: this.file, seg: this.part, time: this.time, features: this.vector, origin: null, true: this.truth, pred: this.guess }; }
}

// label columns of a CSV file: [[which pair member, name], ...] in header order
function label_columns(pair) {
  const cols = [];
  if (pair) for (const side of [0, 1]) if (pair[side]) for (const name of Object.keys(pair[side])) cols.push([side, name]);
  return cols;
}

class FeatureDB {
  constructor() {
    this.tables = new Map();          // db id (as string) -> Map(part key -> Sample), insertion ordered
    this.refused = [];                // human-readable notes on vectors that were not stored
  }

  static key(db, file, part) { return String(db) + '#' + file + '#' + String(part); }

  table(db, create) {
    const id = String(db);
    if (!this.tables.has(id) && create) this.tables.set(id, new Map());
    return this.tables.get(id) || null;
  }

  has_file(db, file) { const t = this.table(db, false); return Boolean(t && t.has(FeatureDB.key(db, file, 0))); }

  samples(db) { const t = this.table(db, false); return t ? Array.from(t.values()) : []; }

  // one vector of `level` for (file, part); returns whether it was taken
  put(level, db, file, part, time, vector) {
    const want = VECTOR_LENGTH[level];
    if (want === undefined || vector.length !== want) {
      this.refused.push('level ' + level + ': vector of ' + vector.length + ' numbers for ' + file + ' part ' + part + ' (expected ' + want + ')');
      return false;
    }
    // through JSON like the app's string storage: typed arrays become index-keyed objects only if handed in as such,
    // -0 becomes 0, NaN / Infinity become null
    const s = new Sample(file, String(part), JSON.parse(JSON.stringify(time)), JSON.parse(JSON.stringify(vector)), null, null);
    this.table(db, true).set(FeatureDB.key(db, file, part), s);
    return true;
  }

  // the collecting callback for FormantAnalyzer.LaunchAudioNodes / LaunchBatch at `level`: (si, labels, time, payload);
  // labels[0] is the file name the app passes as the clip's label list (ref src/index.js:291)
  collector(level, db) {
    const self = this;
    return function collect(si, labels, time, payload) {
      const file = labels[0];
      if (level === 5) self.put(5, db, file, si, time, payload);
      else if (level === 11) self.put(11, db, file, 0, time, payload);
      else if (level === 12 || level === 13) payload.forEach((vec, k) => self.put(level, db, file, si + k / 100, time[k], vec));
      else if (level === 10) payload.forEach((frames, k) => self.put(10, db, file, si + k / 100, time[k], FeatureDB.frame_mean(frames)));
      // level 4 (segment formant frames) is not collected by the app (ref src/index.js:95-98)
    };
  }
  callback(level, db) { return this.collector(level, db); }

  // per-column mean of a syllable's frames in the arithmetic of the app: fp32 running sums in frame order
  static frame_mean(frames) {
    const acc = Float32Array.from(frames[0]);
    for (let f = 1; f < frames.length; f++) for (let c = 0; c < acc.length; c++) acc[c] += frames[f][c];
    for (let c = 0; c < acc.length; c++) if (acc[c] != 0) acc[c] /= frames.length;
    return Array.from(acc);
  }

  to_json(db) {
    const all = this.samples(db);
    return all.length ? JSON.stringify(all.map((s) => s.wire())) : null;
  }

  to_csv(db) {
    const all = this.samples(db);
    if (!all.length) return null;
    const tcols = label_columns(all[0].truth), pcols = label_columns(all[0].guess);
    const cell = (pair, side, name) => {
      if (pair === null) throw new TypeError('sample without the label pair the first sample has');
      return pair[side] ? String(pair[side][name]) : '';
    };
    const lines = [];
    lines.push(['file', 'seg', 't0', 'td'].concat(tcols.map((c) => 'true_' + c[1]), pcols.map((c) => 'pred_' + c[1]),
                                                  all[0].vector.map((_, k) => 'x' + k)));
    for (const s of all)
      lines.push([s.file, s.part, String(s.time[0]), String(s.time[1])].concat(
        tcols.map((c) => cell(s.truth, c[0], c[1])), pcols.map((c) => cell(s.guess, c[0], c[1])), s.vector.map(String)));
    return lines.map((cells) => cells.join(',') + ',\r\n').join('');
  }

  // text of a data_<db>.json file -> samples of `db`; returns how many were read (0: not such a file)
  from_json(db, text) {
    let rows;
    try { rows = JSON.parse(text); } catch (e) { return 0; }
    if (!rows || !rows[0] || !rows[0].file || !rows[0].features) return 0;
    const t = this.table(db, true);
    for (const r of rows) t.set(FeatureDB.key(db, r.file, r.seg), new Sample(r.file, String(r.seg), r.time, r.features, label_pair(r.true), label_pair(r.pred)));
    return rows.length;
  }

  // names of the app's own entry points for these three operations (ref src/localstore.js:843, :1125)
  Download_DB(db, kind) { return kind === 'CSV' ? this.to_csv(db) : this.to_json(db); }
  Load_JSON_Data(db, text) { return this.from_json(db, text); }
}

module.exports = { FeatureDB, VECTOR_LENGTH };

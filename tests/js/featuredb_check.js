// featuredb_check.js — runs webspeechanalyzer_amd/js/featuredb.js through tests/js/featuredb_scenarios.js and compares
// every produced file text with what the reference's own code produced (tests/golden/featuredb_expected.json).
'use strict';
const fs = require('fs');
const path = require('path');
const ROOT = path.join(__dirname, '..', '..');
const { FeatureDB } = require(path.join(ROOT, 'webspeechanalyzer_amd', 'js', 'featuredb.js'));
let db = null;
const api = {
  reset(heads) { db = new FeatureDB(new Map(), heads); },
  callback(level, db_id) { return db.callback(level, db_id); },
  download(d, type, sel) { return db.Download_DB(d, type, sel); },
  load_json(d, text) { db.Load_JSON_Data(d, text); },
  load_labels(text) { db.Load_JSON_Labels_file(text); },
  update_true(seg, label, val, clear) { db.update_true_label(seg, label, val, clear); },
  update_pred(seg, label, val) { return db.update_pred_label(seg, label, val); },
  keys(d) { return JSON.parse(db.store.get('_a_' + String(d)) || '[]'); },
};
const cases = JSON.parse(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'backend_expected.json'), 'utf8')).cases;
const want = JSON.parse(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'featuredb_expected.json'), 'utf8')).expected;
const got = require('./featuredb_scenarios.js').run(api, cases);
const bad = [];
for (const k of Object.keys(want)) if (got[k] !== want[k]) bad.push(k);
for (const k of Object.keys(got)) if (!(k in want)) bad.push(k + ' (not in the fixture)');
console.log(JSON.stringify({ checked: Object.keys(want).length, mismatches: bad }));
process.exit(bad.length ? 1 : 0);

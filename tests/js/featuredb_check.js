// featuredb_check.js — runs webspeechanalyzer_amd/js/featuredb.js through tests/js/featuredb_scenarios.js and compares
// every produced file text with what the reference's own code produced (tests/golden/featuredb_expected.json).
'use strict';
const fs = require('fs');
const path = require('path');
const ROOT = path.join(__dirname, '..', '..');
const { FeatureDB } = require(path.join(ROOT, 'webspeechanalyzer_amd', 'js', 'featuredb.js'));
let db = null;
const api = {
  reset() { db = new FeatureDB(); },
  callback(level, db_id) { return db.collector(level, db_id); },
  download(d, type) { return type === 'CSV' ? db.to_csv(d) : db.to_json(d); },
  load_json(d, text) { db.from_json(d, text); },
};
const cases = JSON.parse(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'backend_expected.json'), 'utf8')).cases;
const want = JSON.parse(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'featuredb_expected.json'), 'utf8')).expected;
const got = require('./featuredb_scenarios.js').run(api, cases);
const bad = [];
for (const k of Object.keys(want)) if (got[k] !== want[k]) bad.push(k);
for (const k of Object.keys(got)) if (!(k in want)) bad.push(k + ' (not in the fixture)');
console.log(JSON.stringify({ checked: Object.keys(want).length, mismatches: bad }));
process.exit(bad.length ? 1 : 0);

// make_featuredb_golden.js — fixture generator.  TEST INFRASTRUCTURE, build container only.
//
// Runs the reference app's OWN storage / export code — /root/reference/src/localstore.js, src/labeling.js and the
// callback `call_backed` of src/index.js, read from there AT RUN TIME (nothing of them is copied into this repository) —
// under Node with a stub `window.localStorage` / `document`, on the scenarios of tests/js/featuredb_scenarios.js, and
// writes the file texts it produces to tests/golden/featuredb_expected.json.
// usage: node tests/golden/gen/make_featuredb_golden.js   (needs /root/reference; outputs are the committed fixture)
'use strict';
const fs = require('fs');
const path = require('path');
const REF = '/root/reference/src';
const ROOT = path.join(__dirname, '..', '..', '..');

function strip_exports(src) { return src.replace(/^export\s+(async\s+)?function/gm, '$1function'); }

let store = new Map();
let dom = {};
let captured = null;
const window_ = {
  localStorage: { getItem: k => (store.has(k) ? store.get(k) : null), setItem: (k, v) => { store.set(k, String(v)); },
                  removeItem: k => { store.delete(k); }, clear: () => store.clear() },
  URL: { createObjectURL: () => 'blob:x' }, navigator: {},
};
const document_ = {
  getElementById: id => { if (!dom[id]) dom[id] = { value: '', textContent: '', innerHTML: '', style: {} }; return dom[id]; },
  createElement: () => ({ click() {} }), body: { appendChild() {}, removeChild() {} },
};
function Blob_(parts) { captured = parts[0]; }
const quiet = { log() {}, warn() {}, error() {} };

// labeling.js
const lab = new Function('document', 'alert', 'console',
  strip_exports(fs.readFileSync(path.join(REF, 'labeling.js'), 'utf8')) + '\nreturn { Load_JSON_Labels_file, label_from_filename };')(document_, () => {}, quiet);
// localstore.js (its `require('./labeling.js')` resolves to the object above; timers fire at once)
const ls_src = strip_exports(fs.readFileSync(path.join(REF, 'localstore.js'), 'utf8'));
const ls = new Function('window', 'document', 'alert', 'Blob', 'require', 'setTimeout', 'console',
  ls_src + '\nreturn { StoreFeatures, collect_db_data, Download_DB, Load_JSON_Data };')(
  window_, document_, () => {}, Blob_, () => lab, (f) => f(), quiet);
// call_backed of index.js
const idx = fs.readFileSync(path.join(REF, 'index.js'), 'utf8');
const a = idx.indexOf('async function call_backed'), b = idx.indexOf('function callback_after_pred');
if (a < 0 || b < a) throw new Error('call_backed not found');
const make_cb = new Function('settings', 'storage_mod', 'pred_mod', 'console', idx.slice(a, b) + '\nreturn call_backed;');

const api = {
  reset() { store = new Map(); dom = {}; document_.getElementById('class_labels').value = '[]'; document_.getElementById('ordinal_labels').value = '[]'; },
  callback(level, db_id) { return make_cb({ output_level: level, collect: true, DB_ID: db_id, plot_enable: false, predict_en: false }, ls, {}, quiet); },
  download(db, type) { captured = null; ls.Download_DB(db, type, false); return captured; },
  load_json(db, text) { ls.Load_JSON_Data(db, text); },
};
const cases = JSON.parse(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'backend_expected.json'), 'utf8')).cases;
const out = require(path.join(ROOT, 'tests', 'js', 'featuredb_scenarios.js')).run(api, cases);
fs.writeFileSync(path.join(ROOT, 'tests', 'golden', 'featuredb_expected.json'), JSON.stringify({
  generator: 'tests/golden/gen/make_featuredb_golden.js', node: process.version,
  reference: 'src/localstore.js, src/labeling.js, src/index.js (call_backed) of /root/reference, run under Node with a stub DOM', expected: out }, null, 1));
for (const k of Object.keys(out)) console.log(k, out[k] === null ? 'null' : (typeof out[k] === 'string' ? out[k].length + ' chars' : out[k]));

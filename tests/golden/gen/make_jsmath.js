// make_jsmath.js — writes Node's own Math.log10 / Math.pow results (as hex bit patterns) for the
// argument ranges the reference's noise gate and feature code use.  Build-container helper;
// the output tests/golden/jsmath_v8.json pins oracle/jsmath.c and the device-side port.
// usage: node make_jsmath.js out.json [big]
'use strict';
const fs = require('fs');
const buf = new DataView(new ArrayBuffer(8));
const hex = x => { buf.setFloat64(0, x); return buf.getUint32(0).toString(16).padStart(8, '0') + buf.getUint32(4).toString(16).padStart(8, '0'); };
let s = 0x1234567n;
const rnd = () => { s = (s * 6364136223846793005n + 1442695040888963407n) & 0xffffffffffffffffn; return Number(s >> 11n) / 9007199254740992; };
const big = process.argv[3] === 'big';
const log10_in = [], pow_in = [];
// every integer the gate can hit exactly on a truncation boundary (SURVEY.md §8a a5)
for (let m = 501; m * 20000 < 4294967296; m += big ? 1 : 1499) log10_in.push(m * 20000);
for (let m = 500; m <= 5000; m += big ? 1 : 61) log10_in.push(m * 2000);
for (let m = 50; m <= 5000; m += big ? 1 : 37) log10_in.push(m * 200);
for (let m = 4; m <= 22; m++) log10_in.push(m * m * m);
for (let e = 0; e <= 9; e++) log10_in.push(Math.pow(10, e));
for (let i = 0; i < (big ? 400000 : 600); i++) log10_in.push(Math.floor(Math.pow(2, 32 * rnd())) + 1);
for (let i = 0; i < (big ? 100000 : 300); i++) log10_in.push(Math.fround(Math.pow(2, 40 * rnd() - 2)));  // fp32 band energies
const out = { node: process.version, v8: process.versions.v8, log10: [], pow: [] };
for (const y of log10_in) {
  const t = Math.log10(y);
  out.log10.push([hex(y), hex(t)]);
  for (const a of [t - 3, t - 2, t / 3]) pow_in.push([10, a]);
}
for (let i = 0; i < (big ? 200000 : 500); i++) pow_in.push([10, 20 * rnd() - 10]);
for (let i = 0; i < (big ? 200000 : 500); i++) pow_in.push([Math.pow(2, 60 * rnd() - 30), 8 * rnd() - 4]);
for (let i = 0; i < 200; i++) pow_in.push([10 * rnd() - 5, Math.floor(12 * rnd()) - 4]);
for (const [x, y] of pow_in) out.pow.push([hex(x), hex(y), hex(Math.pow(x, y))]);
fs.writeFileSync(process.argv[2], JSON.stringify(out));

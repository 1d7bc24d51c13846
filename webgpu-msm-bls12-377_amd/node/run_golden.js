'use strict';
// Test driver: node run_golden.js <case.bin> <n>  -> prints {"x": "...", "y": "..."} (decimal),
// the way the harness compares results (toString equality, src/ui/Benchmark.tsx:41-48).
const fs = require('fs');
const { compute_msm, compute_msm_edwards, set_bases, compute_msm_fixed_base, version } = require('./compute_msm.js');

// node run_golden.js <ed case.bin> <n> ed  -> the Edwards-BLS12 twin on an Edwards golden case (64-byte points)
if (process.argv[4] === 'ed') {
  const blob = fs.readFileSync(process.argv[2]);
  const n = parseInt(process.argv[3], 10);
  const r = compute_msm_edwards(blob.slice(0, 64 * n), blob.slice(64 * n, 96 * n));
  const empty = compute_msm_edwards(Buffer.alloc(0), Buffer.alloc(0));
  console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString(), empty_x: empty.x.toString(), empty_y: empty.y.toString() }));
  process.exit(0);
}

(async () => {
  const blob = fs.readFileSync(process.argv[2]);
  const n = parseInt(process.argv[3], 10);
  const points = blob.slice(0, 96 * n);
  const scalars = blob.slice(96 * n, 128 * n);
  const r = await compute_msm(points, scalars, false);
  // second form: BigIntPoint[] / bigint[]
  const le = (b) => BigInt('0x' + Buffer.from(b).reverse().toString('hex'));
  const pts = [];
  const ks = [];
  for (let i = 0; i < n; i++) {
    pts.push({ x: le(points.slice(96 * i, 96 * i + 48)), y: le(points.slice(96 * i + 48, 96 * i + 96)), z: BigInt(1) });
    ks.push(le(scalars.slice(32 * i, 32 * i + 32)));
  }
  const r2 = await compute_msm(pts, ks, false);
  if (r2.x !== r.x || r2.y !== r.y) throw new Error('BigIntPoint[] form disagrees with Buffer form');
  // third form: U32ArrayPoint[] / Uint32Array[] -- most-significant word first (src/reference/webgpu/utils.ts:41-61)
  const u32 = (v, bits) => {
    const words = new Uint32Array(bits / 32);
    for (let i = words.length - 1; i >= 0; i--) {
      words[i] = Number(v & BigInt(0xffffffff));
      v >>= BigInt(32);
    }
    return words;
  };
  const pts32 = pts.map((p) => ({ x: u32(p.x, 384), y: u32(p.y, 384) }));
  const ks32 = ks.map((k) => u32(k, 256));
  const r3 = await compute_msm(pts32, ks32, false);
  if (r3.x !== r.x || r3.y !== r.y) throw new Error('U32ArrayPoint[] form disagrees with Buffer form');
  // fixed-base form: the same points resident, the same scalars -> the same result; a prefix of the scalars against the
  // resident set agrees with the plain call on that prefix
  set_bases(points);
  const r4 = compute_msm_fixed_base(scalars);
  if (r4.x !== r.x || r4.y !== r.y) throw new Error('fixed-base form disagrees with the plain call');
  if (n > 1) {
    const half = n >> 1;
    const a = compute_msm_fixed_base(scalars.slice(0, 32 * half));
    const b = await compute_msm(points.slice(0, 96 * half), scalars.slice(0, 32 * half), false);
    if (a.x !== b.x || a.y !== b.y) throw new Error('fixed-base prefix disagrees with the plain call');
  }
  const empty = await compute_msm(Buffer.alloc(0), Buffer.alloc(0), false);
  console.log(JSON.stringify({ forms: 3, x: r.x.toString(), y: r.y.toString(), empty_x: empty.x.toString(), empty_y: empty.y.toString(), version: version() }));
})().catch((e) => {
  console.error(String(e));
  process.exit(1);
});

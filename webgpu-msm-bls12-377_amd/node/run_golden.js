'use strict';
// Test driver: node run_golden.js <case.bin> <n>  -> prints {"x": "...", "y": "..."} (decimal),
// the way the harness compares results (toString equality, src/ui/Benchmark.tsx:41-48).
const fs = require('fs');
const { compute_msm, version } = require('./compute_msm.js');

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
  const empty = await compute_msm(Buffer.alloc(0), Buffer.alloc(0), false);
  console.log(JSON.stringify({ forms: 3, x: r.x.toString(), y: r.y.toString(), empty_x: empty.x.toString(), empty_y: empty.y.toString(), version: version() }));
})().catch((e) => {
  console.error(String(e));
  process.exit(1);
});

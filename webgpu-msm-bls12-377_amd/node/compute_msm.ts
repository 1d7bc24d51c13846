/**
 * compute_msm for src/submission: drop-in replacement of the WebGPU implementation
 * (/root/reference/src/submission/submission.ts:85-327).  Copy this file over
 * src/submission/submission.ts (or re-export compute_msm from it): the harness callers
 * (src/ui/AllBenchmarks.tsx:149-158, src/ui/Benchmark.tsx:32,
 * src/submission/miscellaneous/full_benchmarks.ts:62,99) need no change.
 *
 * The five WebGPU stage drivers and the BigInt CPU tail are replaced by one call into the
 * N-API shim (msm377_napi.node) over the C ABI in include/msm377.h; see INTEGRATION.md.
 */
import { BigIntPoint, U32ArrayPoint } from "../reference/types";

// eslint-disable-next-line @typescript-eslint/no-var-requires
const addon: {
  computeMsm(points: Buffer, scalars: Buffer): Promise<Buffer>;
  computeMsmSync(points: Buffer, scalars: Buffer): Buffer;
  computeEdMsmSync(points: Buffer, scalars: Buffer): Buffer;
  setBasesSync(points: Buffer): void;
  fixedBaseMsmSync(scalars: Buffer): Buffer;
  version(): string;
} = require("./msm377/build/msm377_napi.node");

const leBufferToBigInt = (buf: Buffer): bigint =>
  BigInt("0x" + Buffer.from(buf).reverse().toString("hex"));

const bigIntToBufferLE = (v: bigint, bytes: number): Buffer =>
  Buffer.from(v.toString(16).padStart(bytes * 2, "0"), "hex").reverse();

// Most-significant-first u32 words: src/reference/webgpu/utils.ts:49-61
const u32WordsToBigInt = (words: Uint32Array): bigint => {
  let v = BigInt(0);
  for (const w of words) v = (v << BigInt(32)) | BigInt(w >>> 0);
  return v;
};

const toBigInt = (v: bigint | Uint32Array): bigint =>
  typeof v === "bigint" ? v : u32WordsToBigInt(v);

export const pointsToBuffer = (
  baseAffinePoints: BigIntPoint[] | U32ArrayPoint[] | Buffer,
): Buffer => {
  if (Buffer.isBuffer(baseAffinePoints)) return baseAffinePoints;
  const parts: Buffer[] = [];
  for (const pt of baseAffinePoints as (BigIntPoint | U32ArrayPoint)[]) {
    parts.push(bigIntToBufferLE(toBigInt(pt.x), 48));
    parts.push(bigIntToBufferLE(toBigInt(pt.y), 48));
  }
  return Buffer.concat(parts);
};

export const scalarsToBuffer = (
  scalars: bigint[] | Uint32Array[] | Buffer,
): Buffer => {
  if (Buffer.isBuffer(scalars)) return scalars;
  return Buffer.concat(
    (scalars as (bigint | Uint32Array)[]).map((s) => bigIntToBufferLE(toBigInt(s), 32)),
  );
};

export const compute_msm = async (
  baseAffinePoints: BigIntPoint[] | U32ArrayPoint[] | Buffer,
  scalars: bigint[] | Uint32Array[] | Buffer,
  log_result = true,
  force_recompile = false,
): Promise<{ x: bigint; y: bigint }> => {
  void force_recompile; // kernels are compiled ahead of time for gfx950
  const scalarsBuf = scalarsToBuffer(scalars);
  const input_size = scalarsBuf.length / 32;

  if (input_size === 0) {
    return { x: BigInt(0), y: BigInt(1) };
  }

  const pointsBuf = pointsToBuffer(baseAffinePoints);
  const out = await addon.computeMsm(pointsBuf, scalarsBuf);
  const r = {
    x: leBufferToBigInt(out.subarray(0, 48) as Buffer),
    y: leBufferToBigInt(out.subarray(48, 96) as Buffer),
  };
  if (log_result) {
    console.log(r);
  }
  return r;
};

// The Edwards-BLS12 twin (BASELINE.json config 3; the reference's orphaned Edwards shaders,
// src/submission/miscellaneous/wgsl/add_points_any_a.template.wgsl:24-71): 64-byte points x || y, 32-byte
// little-endian each (README.md:299-301); the neutral element is (0, 1).
export const compute_msm_edwards = (points: Buffer, scalars: Buffer): { x: bigint; y: bigint } => {
  if (scalars.length === 0) {
    return { x: BigInt(0), y: BigInt(1) };
  }
  const out: Buffer = addon.computeEdMsmSync(points, scalars);
  return { x: leBufferToBigInt(out.subarray(0, 32) as Buffer), y: leBufferToBigInt(out.subarray(32, 64) as Buffer) };
};

// Fixed-base batches (BASELINE.json config 5): convert and keep a base set in HBM once, then any number of MSMs of
// n <= its size against it (msm377_g1_set_bases / msm377_g1_msm_fixed_base).
export const set_bases = (baseAffinePoints: BigIntPoint[] | U32ArrayPoint[] | Buffer): void => {
  addon.setBasesSync(pointsToBuffer(baseAffinePoints));
};

export const compute_msm_fixed_base = (scalars: bigint[] | Uint32Array[] | Buffer): { x: bigint; y: bigint } => {
  const scalarsBuf = scalarsToBuffer(scalars);
  if (scalarsBuf.length === 0) {
    return { x: BigInt(0), y: BigInt(1) };
  }
  const out: Buffer = addon.fixedBaseMsmSync(scalarsBuf);
  return { x: leBufferToBigInt(out.subarray(0, 48) as Buffer), y: leBufferToBigInt(out.subarray(48, 96) as Buffer) };
};

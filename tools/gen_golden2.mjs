// Golden-vector generator, second file (round 2; our tooling): wire-format codecs the first file does not cover and the public
// clearCofactor.  Runs the REAL reference (type-stripped copy under /tmp, tools/strip_ts.py) and prints one JSON document.
// Driver: tools/gen_golden.py -> tests/golden/ref_vectors2.json.gz        node tools/gen_golden2.mjs /tmp/nbls_ref
import { createHash } from 'crypto';
import { pathToFileURL } from 'url';
import path from 'path';

const refDir = process.argv[2] || '/tmp/nbls_ref';

async function main() {
  const bls = await import(pathToFileURL(path.join(refDir, 'index.mjs')).href);
  const { PointG1, PointG2, Fp, Fp2, CURVE } = bls;
  const hex = (u8) => Buffer.from(u8).toString('hex');
  let ctr = 0;
  const rnd = (tag, bytes) => {
    let out = Buffer.alloc(0);
    while (out.length < bytes) {
      const c = Buffer.alloc(4); c.writeUInt32BE(ctr++);
      out = Buffer.concat([out, createHash('sha256').update('nbls-golden-v2').update(tag).update(c).digest()]);
    }
    return out.slice(0, bytes);
  };
  const rScalar = (tag) => (BigInt('0x' + rnd(tag, 40).toString('hex')) % (CURVE.r - 1n)) + 1n;
  const b48 = (v) => v.toString(16).padStart(96, '0');
  const g1aff = (P) => { const [x, y] = P.toAffine(); return b48(x.value) + b48(y.value); };
  const g2aff = (Q) => { const [x, y] = Q.toAffine(); return b48(x.c0.value) + b48(x.c1.value) + b48(y.c0.value) + b48(y.c1.value); };
  const tryG2 = (fn, bytes) => {
    try { const Q = fn(Uint8Array.from(bytes)); return Q.isZero() ? { result: 'zero', aff: null } : { result: 'ok', aff: g2aff(Q) }; } catch (e) { return { result: e.message, aff: null }; }
  };
  const tryG1 = (bytes) => {
    try { const P = PointG1.fromHex(Uint8Array.from(bytes)); return P.isZero() ? { result: 'zero', aff: null } : { result: 'ok', aff: g1aff(P) }; } catch (e) { return { result: e.message, aff: null }; }
  };
  const out = {};

  // a point of E'(Fp2) outside the prime-order subgroup: x random until x^3 + b is a square (PointG2.fromHex does not check the subgroup)
  const offGroup = [];
  while (offGroup.length < 4) {
    const raw = rnd('off', 96); raw[0] = (raw[0] & 0x0f) | 0x80 | ((offGroup.length & 1) << 5); raw[48] &= 0x0f;
    const r = tryG2((b) => PointG2.fromHex(b), raw);
    if (r.result === 'ok') offGroup.push({ hex: hex(raw), ...r });
  }

  // ---- PointG2.fromHex on 96 bytes (index.ts:532-562): flag rules, no subgroup check, sign by the S bit
  const fh = [];
  for (let i = 0; i < 6; i++) {
    const Q = PointG2.BASE.multiplyUnsafe(rScalar('fh'));
    const c = Q.toRawBytes(true);
    fh.push({ hex: hex(c), ...tryG2((b) => PointG2.fromHex(b), c) });
    const flipped = Uint8Array.from(c); flipped[0] ^= 0x20;                      // the other root: a valid encoding of -Q
    fh.push({ hex: hex(flipped), ...tryG2((b) => PointG2.fromHex(b), flipped) });
  }
  for (const o of offGroup) fh.push(o);
  {
    const Q = PointG2.BASE.multiplyUnsafe(rScalar('fh2'));
    const c = Q.toRawBytes(true);
    for (const m of [0x00, 0x20, 0x40, 0x60, 0xa0, 0xc0, 0xe0]) {                // every flag combination on a valid x
      const v = Uint8Array.from(c); v[0] = (v[0] & 0x1f) | m;
      fh.push({ hex: hex(v), ...tryG2((b) => PointG2.fromHex(b), v) });
    }
    const inf = new Uint8Array(96); inf[0] = 0xc0;
    fh.push({ hex: hex(inf), ...tryG2((b) => PointG2.fromHex(b), inf) });
    for (const pos of [1, 47, 48, 95]) { const v = Uint8Array.from(inf); v[pos] = 1; fh.push({ hex: hex(v), ...tryG2((b) => PointG2.fromHex(b), v) }); }
    const v0 = Uint8Array.from(inf); v0[0] = 0xc1; fh.push({ hex: hex(v0), ...tryG2((b) => PointG2.fromHex(b), v0) });
    for (let i = 0; i < 6; i++) {                                                 // random x: about half have no square root
      const raw = rnd('fh3', 96); raw[0] = (raw[0] & 0x0f) | 0x80 | ((i & 1) << 5); raw[48] &= 0x0f;
      fh.push({ hex: hex(raw), ...tryG2((b) => PointG2.fromHex(b), raw) });
    }
  }
  out.g2_fromhex96 = fh;

  // ---- PointG2.fromSignature on 192 bytes (index.ts:500-530): z1 = first 96 bytes, z2 = last 96 bytes as big-endian integers
  const fs = [];
  for (let i = 0; i < 6; i++) {
    const Q = PointG2.BASE.multiplyUnsafe(rScalar('fs'));
    const s = Q.toSignature();                                                   // z1' (48 B, flags + x.c1) || x.c0 (48 B)
    const pad = i % 2 ? rnd('fsp', 48) : Buffer.alloc(48);                       // the high half of z1 is ignored by the reference (z1 mod 2^383)
    // z2 is reduced mod p by the Fp constructor: x.c0 + k p for a small k
    const z2 = BigInt('0x' + hex(s.slice(48))) + BigInt(i) * CURVE.P;
    const v = Buffer.concat([pad, Buffer.from(s.slice(0, 48)), Buffer.from(z2.toString(16).padStart(192, '0'), 'hex')]);
    fs.push({ hex: hex(v), ...tryG2((b) => PointG2.fromSignature(b), v) });
  }
  {
    const v = Buffer.alloc(192); v[48] = 0xc0; fs.push({ hex: hex(v), ...tryG2((b) => PointG2.fromSignature(b), v) });                  // infinity
    for (const o of offGroup) { const r = Buffer.from(o.hex, 'hex'); const v2 = Buffer.concat([Buffer.alloc(48), r.slice(0, 48), Buffer.alloc(48), r.slice(48)]); fs.push({ hex: hex(v2), ...tryG2((b) => PointG2.fromSignature(b), v2) }); }
    for (let i = 0; i < 4; i++) { const raw = rnd('fs2', 192); raw[48] = (raw[48] & 0x0f) | 0x80 | ((i & 1) << 5); fs.push({ hex: hex(raw), ...tryG2((b) => PointG2.fromSignature(b), raw) }); }
  }
  out.g2_fromsig192 = fs;

  // ---- uncompressed forms: PointG1.fromHex(96 B) (index.ts:317-321), PointG2.fromHex(192 B) (index.ts:563-575), toHex(false) / toHex(true)
  const raw1 = [], raw2 = [];
  for (let i = 0; i < 6; i++) {
    const P = PointG1.BASE.multiplyUnsafe(rScalar('u1')), Q = PointG2.BASE.multiplyUnsafe(rScalar('u2'));
    raw1.push({ hex: P.toHex(false), compressed: P.toHex(true), ...tryG1(P.toRawBytes(false)) });
    raw2.push({ hex: Q.toHex(false), compressed: Q.toHex(true), ...tryG2((b) => PointG2.fromHex(b), Q.toRawBytes(false)) });
  }
  raw1.push({ hex: PointG1.ZERO.toHex(false), compressed: PointG1.ZERO.toHex(true), ...tryG1(PointG1.ZERO.toRawBytes(false)) });
  raw2.push({ hex: PointG2.ZERO.toHex(false), compressed: PointG2.ZERO.toHex(true), ...tryG2((b) => PointG2.fromHex(b), PointG2.ZERO.toRawBytes(false)) });
  for (let i = 0; i < 6; i++) {
    // random coordinates (not on the curve), a coordinate above p (reduced by the Fp constructor), the infinity flag over garbage
    const a = rnd('u3', 96); a[0] &= 0x0f; a[48] &= 0x0f; if (i === 4) a[0] |= 0x40;
    raw1.push({ hex: hex(a), compressed: null, ...tryG1(a) });
    const c = rnd('u4', 192); for (const k of [0, 48, 96, 144]) c[k] &= 0x0f; if (i === 4) c[0] |= 0x40;
    raw2.push({ hex: hex(c), compressed: null, ...tryG2((b) => PointG2.fromHex(b), c) });
  }
  {
    const P = PointG1.BASE.multiplyUnsafe(rScalar('u5')); const [x, y] = P.toAffine();
    const v = Buffer.from(b48(x.value + CURVE.P) + b48(y.value), 'hex');        // x + p < 2^383: same point after reduction
    raw1.push({ hex: hex(v), compressed: null, ...tryG1(v) });
    const Q = PointG2.BASE.multiplyUnsafe(rScalar('u6')); const [qx, qy] = Q.toAffine();
    const w = Buffer.from(b48(qx.c1.value) + b48(qx.c0.value + CURVE.P) + b48(qy.c1.value) + b48(qy.c0.value), 'hex');
    raw2.push({ hex: hex(w), compressed: null, ...tryG2((b) => PointG2.fromHex(b), w) });
    for (const o of offGroup) { const Qo = PointG2.fromHex(Uint8Array.from(Buffer.from(o.hex, 'hex'))); const u = Buffer.from(b48(Qo.toAffine()[0].c1.value) + b48(Qo.toAffine()[0].c0.value) + b48(Qo.toAffine()[1].c1.value) + b48(Qo.toAffine()[1].c0.value), 'hex'); raw2.push({ hex: hex(u), compressed: null, ...tryG2((b) => PointG2.fromHex(b), u) }); }
  }
  out.g1_raw96 = raw1; out.g2_raw192 = raw2;

  // ---- PointG2.clearCofactor as a public method (index.ts:659-672; test/point.test.ts:388-478 checks it against multiplyUnsafe(h2Eff))
  // [k]Q by double-and-add over the reference's own add / double (multiplyUnsafe refuses scalars above r, and h2Eff is 636 bits)
  const mulBig = (Q, k) => { let R = PointG2.ZERO, D = Q; while (k > 0n) { if (k & 1n) R = R.add(D); D = D.double(); k >>= 1n; } return R; };
  const cc = [];
  for (const o of offGroup) {
    const Q = PointG2.fromHex(Uint8Array.from(Buffer.from(o.hex, 'hex')));
    const R = Q.clearCofactor();
    cc.push({ aff: o.aff, out: R.isZero() ? null : g2aff(R), equals_h2eff: R.equals(mulBig(Q, CURVE.h2Eff)) });
  }
  for (let i = 0; i < 2; i++) { const Q = PointG2.BASE.multiplyUnsafe(rScalar('cc')); const R = Q.clearCofactor(); cc.push({ aff: g2aff(Q), out: g2aff(R), equals_h2eff: R.equals(mulBig(Q, CURVE.h2Eff)) }); }
  out.g2_clear_cofactor = cc;
  // PointG1.clearCofactor (index.ts:401-405) on points outside the subgroup
  const cc1 = [];
  let tries = 0;
  while (cc1.length < 4 && tries++ < 64) {
    const raw = rnd('cc1', 48); raw[0] = (raw[0] & 0x0f) | 0x80;
    const x = new Fp(BigInt('0x' + hex(raw)) % (1n << 381n));
    const y = x.pow(3n).add(new Fp(4n)).sqrt();
    if (!y) continue;
    const P = new PointG1(x, y);
    const R = P.clearCofactor();
    cc1.push({ aff: g1aff(P), out: R.isZero() ? null : g1aff(R) });
  }
  out.g1_clear_cofactor = cc1;
  process.stdout.write(JSON.stringify(out));
}
main().catch((e) => { console.error(e); process.exit(1); });

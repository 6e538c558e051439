// Golden-vector generator, third file (round 3; our tooling): flag bits in the first byte of the UNCOMPRESSED wire forms.
// PointG2.fromHex applies its flag rules to 192-byte input too (index.ts:534-537 'Invalid encoding flag', :563 / :571 the compression
// bit sends 192 bytes to 'Invalid point G2, expected 96/192 bytes'), PointG1.fromHex(96 B) only looks at the infinity bit and lets the
// Fp constructor reduce the rest (index.ts:317-321).  Runs the REAL reference (type-stripped copy under /tmp, tools/strip_ts.py).
// Driver: tools/gen_golden.py -> tests/golden/ref_vectors3.json.gz        node tools/gen_golden3.mjs /tmp/nbls_ref
import { createHash } from 'crypto';
import { pathToFileURL } from 'url';
import path from 'path';

const refDir = process.argv[2] || '/tmp/nbls_ref';

async function main() {
  const bls = await import(pathToFileURL(path.join(refDir, 'index.mjs')).href);
  const { PointG1, PointG2, CURVE } = bls;
  const hex = (u8) => Buffer.from(u8).toString('hex');
  let ctr = 0;
  const rnd = (tag, bytes) => {
    let out = Buffer.alloc(0);
    while (out.length < bytes) {
      const c = Buffer.alloc(4); c.writeUInt32BE(ctr++);
      out = Buffer.concat([out, createHash('sha256').update('nbls-golden-v3').update(tag).update(c).digest()]);
    }
    return out.slice(0, bytes);
  };
  const rScalar = (tag) => (BigInt('0x' + rnd(tag, 40).toString('hex')) % (CURVE.r - 1n)) + 1n;
  const b48 = (v) => v.toString(16).padStart(96, '0');
  const g1aff = (P) => { const [x, y] = P.toAffine(); return b48(x.value) + b48(y.value); };
  const g2aff = (Q) => { const [x, y] = Q.toAffine(); return b48(x.c0.value) + b48(x.c1.value) + b48(y.c0.value) + b48(y.c1.value); };
  const tryG2 = (bytes) => { try { const Q = PointG2.fromHex(Uint8Array.from(bytes)); return Q.isZero() ? { result: 'zero', aff: null } : { result: 'ok', aff: g2aff(Q) }; } catch (e) { return { result: e.message, aff: null }; } };
  const tryG1 = (bytes) => { try { const P = PointG1.fromHex(Uint8Array.from(bytes)); return P.isZero() ? { result: 'zero', aff: null } : { result: 'ok', aff: g1aff(P) }; } catch (e) { return { result: e.message, aff: null }; } };
  const FLAGS = [0x00, 0x20, 0x40, 0x60, 0x80, 0xa0, 0xc0, 0xe0];
  const out = { g1_raw96_flags: [], g2_raw192_flags: [] };
  for (let i = 0; i < 2; i++) {
    const P = PointG1.BASE.multiplyUnsafe(rScalar('f1')), Q = PointG2.BASE.multiplyUnsafe(rScalar('f2'));
    for (const f of FLAGS) {
      const a = Buffer.from(P.toRawBytes(false)); a[0] |= f; out.g1_raw96_flags.push({ hex: hex(a), flag: f, ...tryG1(a) });
      const c = Buffer.from(Q.toRawBytes(false)); c[0] |= f; out.g2_raw192_flags.push({ hex: hex(c), flag: f, ...tryG2(c) });
    }
    // non-canonical coordinates x + p, x + 2p (still below 2^384): bit 381 / 382 / 383 of the first word may be set by the VALUE
    const [px, py] = P.toAffine(); const [qx, qy] = Q.toAffine();
    for (const k of [1n, 2n, 3n, 4n]) {
      if (px.value + k * CURVE.P < (1n << 384n)) { const a = Buffer.from(b48(px.value + k * CURVE.P) + b48(py.value), 'hex'); out.g1_raw96_flags.push({ hex: hex(a), flag: a[0] & 0xe0, ...tryG1(a) }); }
      if (qx.c1.value + k * CURVE.P < (1n << 384n)) { const c = Buffer.from(b48(qx.c1.value + k * CURVE.P) + b48(qx.c0.value) + b48(qy.c1.value) + b48(qy.c0.value), 'hex'); out.g2_raw192_flags.push({ hex: hex(c), flag: c[0] & 0xe0, ...tryG2(c) }); }
    }
  }
  // flags over the zero encodings and over garbage
  for (const f of FLAGS) {
    const z1 = Buffer.alloc(96); z1[0] = f; out.g1_raw96_flags.push({ hex: hex(z1), flag: f, ...tryG1(z1) });
    const z2 = Buffer.alloc(192); z2[0] = f; out.g2_raw192_flags.push({ hex: hex(z2), flag: f, ...tryG2(z2) });
    const a = rnd('f3', 96); a[0] = (a[0] & 0x1f) | f; a[48] &= 0x0f; out.g1_raw96_flags.push({ hex: hex(a), flag: f, ...tryG1(a) });
    const c = rnd('f4', 192); c[0] = (c[0] & 0x1f) | f; for (const k of [48, 96, 144]) c[k] &= 0x0f; out.g2_raw192_flags.push({ hex: hex(c), flag: f, ...tryG2(c) });
  }
  process.stdout.write(JSON.stringify(out));
}
main().catch((e) => { console.error(e); process.exit(1); });

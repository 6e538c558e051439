// Golden-vector generator (our tooling).  Runs the REAL reference (type-stripped copy under
// /tmp, see tools/strip_ts.py) through its public API on deterministic inputs and prints one
// JSON document on stdout.  Driver: tools/gen_golden.py -> tests/golden/ref_vectors.json.gz
//
//   node tools/gen_golden.mjs /tmp/nbls_ref
import { createHash } from 'crypto';
import { pathToFileURL } from 'url';
import path from 'path';

const refDir = process.argv[2] || '/tmp/nbls_ref';

async function main() {
  const bls = await import(pathToFileURL(path.join(refDir, 'index.mjs')).href);
  const math = await import(pathToFileURL(path.join(refDir, 'math.mjs')).href);
  const { PointG1, PointG2, pairing, Fp, Fp2, Fp12, CURVE, utils } = bls;
  const { Fp6, calcPairingPrecomputes, millerLoop, psi, psi2, map_to_curve_simple_swu_9mod16, isogenyMapG2, map_to_curve_simple_swu_3mod4, isogenyMapG1 } = math;
  const hex = (u8) => Buffer.from(u8).toString('hex');
  const sha = (u8) => createHash('sha256').update(u8).digest('hex');

  // deterministic stream: s_k = SHA-256("nbls-golden-v1" || tag || u32be(k))
  let ctr = 0;
  const rnd = (tag, bytes) => {
    let out = Buffer.alloc(0);
    while (out.length < bytes) {
      const c = Buffer.alloc(4); c.writeUInt32BE(ctr++);
      out = Buffer.concat([out, createHash('sha256').update('nbls-golden-v1').update(tag).update(c).digest()]);
    }
    return out.slice(0, bytes);
  };
  const rndBig = (tag, bytes) => BigInt('0x' + rnd(tag, bytes).toString('hex'));
  const rFp = (t) => new Fp(rndBig(t, 64));
  const rFp2 = (t) => new Fp2(rFp(t), rFp(t));
  const rFp6 = (t) => new Fp6(rFp2(t), rFp2(t), rFp2(t));
  const rFp12 = (t) => new Fp12(rFp6(t), rFp6(t));
  const rScalar = (t) => (rndBig(t, 48) % (CURVE.r - 1n)) + 1n;
  const b = (x) => hex(x.toBytes());
  const out = {};

  // ---- A. field tower -------------------------------------------------------------------
  const fp = [];
  const fpEdge = [[0n, 1n], [1n, CURVE.P - 1n], [CURVE.P - 1n, CURVE.P - 1n], [2n, (CURVE.P + 1n) / 2n]];
  for (let i = 0; i < 12; i++) {
    const a = i < fpEdge.length ? new Fp(fpEdge[i][0]) : rFp('fp');
    const c = i < fpEdge.length ? new Fp(fpEdge[i][1]) : rFp('fp');
    const sq = a.square();
    const rt = sq.sqrt();
    fp.push({ a: b(a), b: b(c), add: b(a.add(c)), sub: b(a.subtract(c)), mul: b(a.multiply(c)), sqr: b(sq), neg: b(a.negate()),
      inv: a.isZero() ? null : b(a.invert()), sqrt_of_sqr: rt ? b(rt) : null, sqrt_of_a: a.sqrt() ? b(a.sqrt()) : null });
  }
  out.fp = fp;
  const fp2 = [];
  for (let i = 0; i < 10; i++) {
    const a = i === 0 ? new Fp2(new Fp(0n), new Fp(5n)) : i === 1 ? new Fp2(new Fp(7n), new Fp(0n)) : rFp2('fp2');
    const c = rFp2('fp2');
    const sq = a.square();
    const r1 = sq.sqrt(), r2 = a.sqrt();
    fp2.push({ a: b(a), b: b(c), add: b(a.add(c)), sub: b(a.subtract(c)), mul: b(a.multiply(c)), sqr: b(sq), inv: b(a.invert()),
      frob1: b(a.frobeniusMap(1)), mulnr: b(a.mulByNonresidue()), mulB: b(a.multiplyByB()),
      sqrt_of_sqr: r1 ? b(r1) : null, sqrt_of_a: r2 ? b(r2) : null });
  }
  out.fp2 = fp2;
  const fp6 = [];
  for (let i = 0; i < 6; i++) {
    const a = rFp6('fp6'), c = rFp6('fp6'), b0 = rFp2('fp6'), b1 = rFp2('fp6');
    fp6.push({ a: b(a), b: b(c), b0: b(b0), b1: b(b1), mul: b(a.multiply(c)), sqr: b(a.square()), inv: b(a.invert()),
      mulnr: b(a.mulByNonresidue()), mul1: b(a.multiplyBy1(b1)), mul01: b(a.multiplyBy01(b0, b1)),
      frob: [1, 2, 3, 4, 5].map((k) => b(a.frobeniusMap(k))) });
  }
  out.fp6 = fp6;
  const fp12 = [];
  for (let i = 0; i < 6; i++) {
    const a = rFp12('fp12'), c = rFp12('fp12'), o0 = rFp2('fp12'), o1 = rFp2('fp12'), o4 = rFp2('fp12');
    // a unitary (cyclotomic) element for cyclotomicSquare: easy part of the final exponentiation
    const t0 = a.frobeniusMap(6).div(a);
    const u = t0.frobeniusMap(2).multiply(t0);
    fp12.push({ a: b(a), b: b(c), o0: b(o0), o1: b(o1), o4: b(o4), mul: b(a.multiply(c)), sqr: b(a.square()), inv: b(a.invert()),
      conj: b(a.conjugate()), mul014: b(a.multiplyBy014(o0, o1, o4)),
      frob: [1, 2, 3, 6].map((k) => b(a.frobeniusMap(k))), unitary: b(u), cyclosqr: b(u.cyclotomicSquare()),
      cycloexp: b(u.cyclotomicExp(CURVE.x)), finalexp: b(a.finalExponentiate()) });
  }
  out.fp12 = fp12;

  // ---- B. pairing -----------------------------------------------------------------------
  const g1aff = (P) => { const [x, y] = P.toAffine(); return b(x) + b(y); };
  const g2aff = (Q) => { const [x, y] = Q.toAffine(); return b(x) + b(y); };
  const ellBytes = (ell) => Buffer.concat(ell.map((t) => Buffer.concat(t.map((e) => Buffer.from(e.toBytes())))));
  const pairs = [];
  for (let i = 0; i < 12; i++) {
    const ka = i === 0 ? 1n : i === 1 ? 2n : rScalar('pair');
    const kb = i === 0 ? 1n : i === 1 ? 3n : rScalar('pair');
    const P = PointG1.BASE.multiplyUnsafe(ka), Q = PointG2.BASE.multiplyUnsafe(kb);
    const [qx, qy] = Q.toAffine();
    const ell = calcPairingPrecomputes(qx, qy);
    const eb = ellBytes(ell);
    const ml = pairing(P, Q, false);
    pairs.push({ ka: ka.toString(16), kb: kb.toString(16), g1: g1aff(P), g2: g2aff(Q), ell_len: ell.length, ell_sha256: sha(eb),
      ell_first: hex(eb.slice(0, 288)), ell_last: hex(eb.slice(eb.length - 288)), miller: b(ml), pairing: b(ml.finalExponentiate()) });
  }
  out.pairs = pairs;
  {
    // shared final exponentiation: (ML(2G1,3G2) * ML(5G1,7G2))^fe == e(G1,G2)^41
    const m1 = pairing(PointG1.BASE.multiplyUnsafe(2n), PointG2.BASE.multiplyUnsafe(3n), false);
    const m2 = pairing(PointG1.BASE.multiplyUnsafe(5n), PointG2.BASE.multiplyUnsafe(7n), false);
    out.product = { g1: [g1aff(PointG1.BASE.multiplyUnsafe(2n)), g1aff(PointG1.BASE.multiplyUnsafe(5n))],
      g2: [g2aff(PointG2.BASE.multiplyUnsafe(3n)), g2aff(PointG2.BASE.multiplyUnsafe(7n))],
      miller_product: b(m1.multiply(m2)), result: b(m1.multiply(m2).finalExponentiate()),
      e_pow_41: b(pairing(PointG1.BASE, PointG2.BASE).pow(41n)) };
  }

  // ---- C. points ------------------------------------------------------------------------
  const proj1 = (P) => [b(P.x), b(P.y), b(P.z)];
  const g1pts = [], g2pts = [];
  for (let i = 0; i < 6; i++) {
    const k1 = rScalar('pt'), k2 = rScalar('pt');
    // projective inputs with Z != 1 (scale an affine point by a random z)
    const A = PointG1.BASE.multiplyUnsafe(k1), Bp = PointG1.BASE.multiplyUnsafe(k2);
    g1pts.push({ P: proj1(A), Q: proj1(Bp), dbl: proj1(A.double()), add: proj1(A.add(Bp)), add_self: proj1(A.add(A)),
      add_neg_iszero: A.add(A.negate()).isZero(), mulx: g1aff(A.multiplyUnsafe(CURVE.x)), aff: g1aff(A), affQ: g1aff(Bp), sum_aff: g1aff(A.add(Bp)) });
    const C = PointG2.BASE.multiplyUnsafe(k1), D = PointG2.BASE.multiplyUnsafe(k2);
    const [cx, cy] = C.toAffine();
    g2pts.push({ P: proj1(C), Q: proj1(D), dbl: proj1(C.double()), add: proj1(C.add(D)), add_self: proj1(C.add(C)),
      add_neg_iszero: C.add(C.negate()).isZero(), mulx: g2aff(C.multiplyUnsafe(CURVE.x)), aff: g2aff(C), affQ: g2aff(D), sum_aff: g2aff(C.add(D)),
      psi: psi(cx, cy).map(b).join(''), psi2: psi2(cx, cy).map(b).join('') });
  }
  out.g1pts = g1pts; out.g2pts = g2pts;

  // validity: in-subgroup, on-curve-but-not-in-subgroup, off-curve
  const validity = { g1: [], g2: [] };
  const tryValid = (f) => { try { f(); return 'ok'; } catch (e) { return e.message; } };
  {
    let found = 0;
    for (let xi = 1n; found < 4; xi++) {
      const x = new Fp(xi + (rndBig('v1', 40) % CURVE.P));
      const y = x.pow(3n).add(new Fp(4n)).sqrt();
      if (!y) continue;
      const P = new PointG1(x, y);
      validity.g1.push({ aff: b(x) + b(y), result: tryValid(() => P.assertValidity()) });
      found++;
    }
    for (let i = 0; i < 3; i++) {
      const P = PointG1.BASE.multiplyUnsafe(rScalar('v1'));
      validity.g1.push({ aff: g1aff(P), result: tryValid(() => P.assertValidity()) });
    }
    const [gx, gy] = PointG1.BASE.toAffine();
    validity.g1.push({ aff: b(gx) + b(gy.add(Fp.ONE)), result: tryValid(() => new PointG1(gx, gy.add(Fp.ONE)).assertValidity()) });
    found = 0;
    while (found < 4) {
      const x = rFp2('v2');
      const y = x.pow(3n).add(Fp2.fromBigTuple(CURVE.b2)).sqrt();
      if (!y) continue;
      const Q = new PointG2(x, y);
      validity.g2.push({ aff: b(x) + b(y), result: tryValid(() => Q.assertValidity()) });
      found++;
    }
    for (let i = 0; i < 3; i++) {
      const Q = PointG2.BASE.multiplyUnsafe(rScalar('v2'));
      validity.g2.push({ aff: g2aff(Q), result: tryValid(() => Q.assertValidity()) });
    }
    const [hx, hy] = PointG2.BASE.toAffine();
    validity.g2.push({ aff: b(hx) + b(hy.add(Fp2.ONE)), result: tryValid(() => new PointG2(hx, hy.add(Fp2.ONE)).assertValidity()) });
  }
  out.validity = validity;

  // codecs: compressed G1 (48 B) and G2 signature (96 B)
  const codec = { g1: [], g2: [] };
  for (let i = 0; i < 8; i++) {
    const P = PointG1.BASE.multiplyUnsafe(rScalar('c1'));
    codec.g1.push({ hex: P.toHex(true), aff: g1aff(P), result: 'ok' });
    const Q = PointG2.BASE.multiplyUnsafe(rScalar('c2'));
    codec.g2.push({ hex: hex(Q.toSignature()), aff: g2aff(Q), result: 'ok' });
  }
  codec.g1.push({ hex: PointG1.ZERO.toHex(true), aff: null, result: 'zero' });
  codec.g2.push({ hex: hex(PointG2.ZERO.toSignature()), aff: null, result: 'zero' });
  // malformed / not-on-curve / not-in-subgroup encodings
  for (let i = 0; i < 10; i++) {
    const raw = rnd('bad1', 48); raw[0] = (raw[0] & 0x1f) | 0x80 | ((i & 1) << 5);
    raw[0] &= 0x9f | ((i & 1) << 5); if ((raw[0] & 0x1f) > 0x19) raw[0] &= 0xef;
    let res, aff = null;
    try { const P = PointG1.fromHex(raw); res = P.isZero() ? 'zero' : 'ok'; if (res === 'ok') aff = g1aff(P); } catch (e) { res = e.message; }
    codec.g1.push({ hex: hex(raw), aff, result: res });
    const raw2 = rnd('bad2', 96); raw2[0] = (raw2[0] & 0x1f) | 0x80 | ((i & 1) << 5); if ((raw2[0] & 0x1f) > 0x19) raw2[0] &= 0xef;
    raw2[48] &= 0x0f;
    let res2, aff2 = null;
    try { const Q = PointG2.fromSignature(raw2); res2 = Q.isZero() ? 'zero' : 'ok'; if (res2 === 'ok') aff2 = g2aff(Q); } catch (e) { res2 = e.message; }
    codec.g2.push({ hex: hex(raw2), aff: aff2, result: res2 });
  }
  out.codec = codec;

  // ---- D. hash to G2 --------------------------------------------------------------------
  const h2c = [];
  const msgs = [Buffer.alloc(0), Buffer.from('abc'), Buffer.from('abcdef0123456789'), rnd('m', 32), rnd('m', 32), rnd('m', 64), rnd('m', 131), rnd('m', 200)];
  for (const m of msgs) {
    const u = await utils.hashToField(m, 2);
    const Q = await PointG2.hashToCurve(m);
    const [x0, y0] = map_to_curve_simple_swu_9mod16(Fp2.fromBigTuple(u[0]));
    const [xi, yi] = isogenyMapG2(x0, y0);
    h2c.push({ msg: hex(m), dst: utils.getDSTLabel(), u: u.map((e) => e.map((v) => v.toString(16).padStart(96, '0')).join('')).join(''),
      swu0: b(x0) + b(y0), iso0: b(xi) + b(yi), aff: g2aff(Q) });
  }
  {
    const dst = 'QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_';
    for (const m of [Buffer.alloc(0), Buffer.from('abc'), Buffer.from('abcdef0123456789')]) {
      const u = await utils.hashToField(m, 2, { DST: dst });
      const Q = await PointG2.hashToCurve(m, { DST: dst });
      h2c.push({ msg: hex(m), dst, u: u.map((e) => e.map((v) => v.toString(16).padStart(96, '0')).join('')).join(''), aff: g2aff(Q) });
    }
    const xmd = await utils.expandMessageXMD(Buffer.from('abc'), Buffer.from('QUUX-V01-CS02-with-expander-SHA256-128'), 0x20);
    out.xmd_abc_32 = hex(xmd);
  }
  out.h2c = h2c;
  // hash / encode to G1 and encode to G2 driven through the reference itself (default DST and the RFC G1 DST), with the
  // SWU and isogeny intermediates of the first field element
  {
    const other = [];
    const g1aff = (P) => { const [x, y] = P.toAffine(); return b(x) + b(y); };
    const more = [Buffer.alloc(0), Buffer.from('abc'), rnd('g', 1), rnd('g', 31), rnd('g', 32), rnd('g', 55), rnd('g', 56), rnd('g', 64), rnd('g', 119), rnd('g', 300)];
    for (const dst of [utils.getDSTLabel(), 'QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_']) {
      for (const m of more) {
        const u = await utils.hashToField(m, 2, { m: 1, DST: dst });
        const [x0, y0] = map_to_curve_simple_swu_3mod4(new Fp(u[0][0]));
        const [xi, yi] = isogenyMapG1(x0, y0);
        other.push({ msg: hex(m), dst, u: u.map((e) => e[0].toString(16).padStart(96, '0')).join(''), swu0: b(x0) + b(y0), iso0: b(xi) + b(yi),
          g1_hash: g1aff(await PointG1.hashToCurve(m, { DST: dst })), g1_encode: g1aff(await PointG1.encodeToCurve(m, { DST: dst })),
          g2_encode: g2aff(await PointG2.encodeToCurve(m, { DST: dst })) });
      }
    }
    out.h2c_more = other;
  }
  {
    const [bx, by] = PointG2.BASE.multiplyUnsafe(rScalar('cc')).toAffine();
    // clearCofactor on an E2 point outside the subgroup
    let x, y;
    for (;;) { x = rFp2('cc'); y = x.pow(3n).add(Fp2.fromBigTuple(CURVE.b2)).sqrt(); if (y) break; }
    const Q = new PointG2(x, y);
    out.clear_cofactor = { in: b(x) + b(y), out: g2aff(Q.clearCofactor()) };
  }

  // ---- E. signatures --------------------------------------------------------------------
  const sigs = [];
  const sks = [], pks = [], ms = [], ss = [];
  for (let i = 0; i < 6; i++) {
    const sk = rScalar('sk'); const skHex = sk.toString(16).padStart(64, '0');
    const msg = rnd('sm', 32);
    const pk = bls.getPublicKey(skHex);
    const sig = await bls.sign(msg, skHex);
    sks.push(skHex); pks.push(pk); ms.push(msg); ss.push(sig);
    const bad = Buffer.from(msg); bad[0] ^= 1;
    sigs.push({ sk: skHex, msg: hex(msg), pk: hex(pk), sig: hex(sig), verify: await bls.verify(sig, msg, pk),
      verify_badmsg: await bls.verify(sig, bad, pk), verify_badpk: await bls.verify(sig, msg, bls.getPublicKey(sks[0] === skHex ? '02' .padStart(64, '0') : sks[0])) });
  }
  out.sigs = sigs;
  const aggSig = bls.aggregateSignatures(ss);
  const aggPk = bls.aggregatePublicKeys(pks);
  const vb = { agg_sig: hex(aggSig), agg_pk: hex(aggPk), msgs: ms.map(hex), pks: pks.map(hex) };
  vb.ok = await bls.verifyBatch(aggSig, ms, pks);
  const ms2 = ms.map((m) => Buffer.from(m)); ms2[2][5] ^= 0x40;
  vb.bad_msg = await bls.verifyBatch(aggSig, ms2, pks);
  const pks2 = pks.slice(); pks2[1] = pks[0];
  vb.bad_pk = await bls.verifyBatch(aggSig, ms, pks2);
  // same message signed by everyone (aggregate verify as single)
  const same = rnd('same', 32);
  const ss2 = []; for (const sk of sks) ss2.push(await bls.sign(same, sk));
  const aggSame = bls.aggregateSignatures(ss2);
  vb.same_msg = hex(same); vb.same_sig = hex(aggSame);
  vb.same_ok = await bls.verify(aggSame, same, aggPk);
  out.verify_batch = vb;

  process.stdout.write(JSON.stringify(out));
}
main().catch((e) => { console.error(e); process.exit(1); });

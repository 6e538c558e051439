// GPU test of the noble-compatible JS facade (noble-bls12-381_amd/js/index.js) against the committed golden vectors.
// Mirrors the reference's own tests: test/pairing.test.ts:46-64, test/index.test.ts:308-426 (verify / verifyBatch / aggregate).
'use strict';
const fs = require('fs'), zlib = require('zlib'), path = require('path'), assert = require('assert');
const bls = require(path.join(__dirname, '..', '..', 'noble-bls12-381_amd', 'js', 'index.js'));
const load = (f) => JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, '..', 'golden', f))).toString());
const gold = load('ref_vectors.json.gz'), td = load('ref_testdata.json.gz');
const hex = bls.utils.bytesToHex, un = bls.utils.hexToBytes;

(async () => {
  // pairing(G1, G2) == zkcrypto vector; pairing(..., false) == reference Miller value
  const e = bls.pairing(bls.PointG1.BASE, bls.PointG2.BASE);
  assert.strictEqual(hex(e.toBytes()), td.e_G1_G2);
  assert.strictEqual(hex(bls.pairing(bls.PointG1.BASE, bls.PointG2.BASE, false).toBytes()), gold.pairs[0].miller);
  assert.strictEqual(hex(new bls.Fp12(un(td.finalexp_in)).finalExponentiate().toBytes()), td.finalexp_out);
  assert.throws(() => bls.pairing(bls.PointG1.ZERO, bls.PointG2.BASE), /No pairings at point of Infinity/);
  // codecs
  for (const v of gold.codec.g1) {
    if (v.result === 'ok') { const p = bls.PointG1.fromHex(v.hex); assert.strictEqual(hex(p.aff), v.aff); assert.strictEqual(p.toHex(true), v.hex); }
    else if (v.result === 'zero') assert.ok(bls.PointG1.fromHex(v.hex).isZero());
    else assert.throws(() => bls.PointG1.fromHex(v.hex), new RegExp(v.result.replace(/[()]/g, '.')));
  }
  for (const v of gold.codec.g2) {
    if (v.result === 'ok') { const p = bls.PointG2.fromSignature(v.hex); assert.strictEqual(hex(p.aff), v.aff); assert.strictEqual(hex(p.toSignature()), v.hex); }
    else if (v.result === 'zero') assert.ok(bls.PointG2.fromSignature(v.hex).isZero());
    else assert.throws(() => bls.PointG2.fromSignature(v.hex), new RegExp(v.result));
  }
  // hashToCurve
  for (const v of gold.h2c.filter((x) => x.dst === bls.utils.getDSTLabel())) assert.strictEqual(hex((await bls.PointG2.hashToCurve(un(v.msg))).aff), v.aff);
  // every hashToCurve / encodeToCurve known-answer block of the reference's test/hashToCurve.test.ts (G1 and G2, RO and NU)
  for (const k of td.h2c_kats) {
    const f = k.group === 'g1' ? (k.kind === 'hash' ? bls.PointG1.hashToCurve : bls.PointG1.encodeToCurve) : (k.kind === 'hash' ? bls.PointG2.hashToCurve : bls.PointG2.encodeToCurve);
    for (const v of k.vectors) assert.strictEqual((await f(un(v.msg), { DST: k.dst })).toHex(), v.expected, k.suite);
  }
  // verify on reference-produced signatures
  for (const s of gold.sigs) {
    assert.strictEqual(await bls.verify(s.sig, s.msg, s.pk), true);
    const bad = un(s.msg); bad[0] ^= 1;
    assert.strictEqual(await bls.verify(s.sig, bad, s.pk), false);
  }
  // getPublicKey / sign: the reference's sign vectors (test/index.test.ts:20-29) and the reference-run triples
  for (const [priv, msg, sig] of td.sign_vectors.slice(0, 24)) assert.strictEqual(hex(await bls.sign(msg, priv)), sig);
  const batch = td.sign_vectors.slice(24, 152);
  assert.deepStrictEqual((await bls.signBatch(batch.map((v) => v[1]), batch.map((v) => v[0]))).map(hex), batch.map((v) => v[2]));
  for (const s of gold.sigs) {
    assert.strictEqual(hex(bls.getPublicKey(s.sk)), s.pk);
    assert.strictEqual(hex(await bls.sign(s.msg, s.sk)), s.sig);
    assert.strictEqual(hex(bls.getPublicKey(BigInt('0x' + s.sk))), s.pk);
    const sp = await bls.sign(await bls.PointG2.hashToCurve(un(s.msg)), s.sk);        // point in, point out (index.ts:745, 751)
    assert.strictEqual(hex(sp.toSignature()), s.sig);
  }
  // point arithmetic of the facade objects (reference test/point.test.ts style identities)
  {
    const G = bls.PointG1.BASE, H = bls.PointG2.BASE;
    assert.ok(G.multiply(5n).equals(G.double().double().add(G)));
    assert.ok(G.multiply(7n).subtract(G.multiply(3n)).equals(G.multiply(4n)));
    assert.ok(G.multiply(bls.CURVE.r).isZero() && H.multiply(bls.CURVE.r).isZero());
    assert.ok(H.multiply(6n).equals(H.double().add(H.double()).add(H.double())));
    assert.ok(H.multiply(9n).subtract(H.multiply(2n)).equals(H.multiply(7n)));
    assert.ok(bls.PointG1.fromPrivateKey(gold.sigs[0].sk).toHex(true) === gold.sigs[0].pk);
    assert.throws(() => G.multiply(0n), /invalid scalar/);
    // multi-scalar multiplication: sum_i [k_i]P_i against the scalar algebra on the generators
    const k = [3n, (1n << 255n) + 12345n, bls.CURVE.r - 1n, 0xfff000n], a = [2n, 5n, 7n, 11n];
    const t = k.reduce((acc, ki, i) => (acc + ki * a[i]) % bls.CURVE.r, 0n);
    assert.ok(bls.PointG1.msm(a.map((x) => G.multiply(x)), k).equals(G.multiply(t)));
    assert.ok(bls.PointG2.msm(a.map((x) => H.multiply(x)), k).equals(H.multiply(t)));
    assert.ok(bls.PointG1.msm([G, G.negate()], [77n, 77n]).isZero());
    // bilinearity through the facade: e(aG, bH) == e(G, abH)
    assert.ok(bls.pairing(G.multiply(3n), H.multiply(5n)).equals(bls.pairing(G, H.multiply(15n))));
  }
  assert.throws(() => bls.getPublicKey(0n), /Expected valid private key/);
  assert.throws(() => bls.getPublicKey(bls.CURVE.r), /Private key must be 0 < key < CURVE.r/);
  assert.throws(() => bls.getPublicKey('zz'), /Expected valid private key/);
  // aggregate + verifyBatch
  const vb = gold.verify_batch;
  assert.strictEqual(hex(bls.aggregatePublicKeys(vb.pks)), vb.agg_pk);
  assert.strictEqual(hex(bls.aggregateSignatures(gold.sigs.map((s) => s.sig))), vb.agg_sig);
  assert.strictEqual(await bls.verifyBatch(vb.agg_sig, vb.msgs, vb.pks), true);
  const m2 = vb.msgs.slice(); m2[2] = m2[2].slice(0, 10) + ((parseInt(m2[2][10], 16) ^ 4).toString(16)) + m2[2].slice(11);
  assert.strictEqual(await bls.verifyBatch(vb.agg_sig, m2, vb.pks), false);
  const p2 = vb.pks.slice(); p2[1] = vb.pks[0];
  assert.strictEqual(await bls.verifyBatch(vb.agg_sig, vb.msgs, p2), false);
  // point-object path of verifyBatch with a repeated message object (grouping, reference index.ts:804-809)
  const same = await bls.PointG2.hashToCurve(un(vb.same_msg));
  assert.strictEqual(await bls.verifyBatch(bls.PointG2.fromSignature(vb.same_sig), vb.pks.map(() => same), vb.pks.map((k) => bls.PointG1.fromHex(k))), true);
  assert.strictEqual(await bls.verify(vb.same_sig, vb.same_msg, vb.agg_pk), true);
  await assert.rejects(() => bls.verifyBatch(vb.agg_sig, [], []), /Expected non-empty messages array/);
  console.log('JS facade ok');
})().catch((e) => { console.error(e); process.exit(1); });

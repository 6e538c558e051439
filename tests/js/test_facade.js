// GPU test of the noble-compatible JS facade (noble-bls12-381_amd/js/index.js) against the committed golden vectors.
// Mirrors the reference's own tests: test/pairing.test.ts:46-64, test/index.test.ts:308-426 (verify / verifyBatch / aggregate).
'use strict';
const fs = require('fs'), zlib = require('zlib'), path = require('path'), assert = require('assert');
const bls = require(path.join(__dirname, '..', '..', 'noble-bls12-381_amd', 'js', 'index.js'));
const load = (f) => JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, '..', 'golden', f))).toString());
const gold = load('ref_vectors.json.gz'), td = load('ref_testdata.json.gz'), gold2 = load('ref_vectors2.json.gz'), gold3 = load('ref_vectors3.json.gz');
const hex = bls.utils.bytesToHex, un = bls.utils.hexToBytes;

(async () => {
  // pairing(G1, G2) == zkcrypto vector; pairing(..., false) == reference Miller value
  const e = bls.pairing(bls.PointG1.BASE, bls.PointG2.BASE);
  assert.strictEqual(hex(e.toBytes()), td.e_G1_G2);
  assert.strictEqual(hex(bls.pairing(bls.PointG1.BASE, bls.PointG2.BASE, false).toBytes()), gold.pairs[0].miller);
  assert.strictEqual(hex(bls.Fp12.fromBytes(un(td.finalexp_in)).finalExponentiate().toBytes()), td.finalexp_out);
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
  // field classes re-exported by the reference (Fp, Fr, Fp2): values produced by the reference itself under Node (tools/gen_golden.mjs
  // environment; which square root is returned matters), plus algebraic identities
  {
    const { Fp, Fr, Fp2 } = bls;
    const KAT = [{"a": "6db81c7c3bb8001a70db6825237a68d16afbe3ef96ff20d09f166f6fa64fbc0f74b64f128ab6978006d1c34c1c30898", "b": "1443e64d873318f63ccb594417232f71634dc5dfb781376bd86e4d5f55134f5d035f1309be5fc1357a580b18313fb18d", "c": "2d468a88aca03e5eb9dbf78cc5f44c92b506d77fbba72c1e58edcfb825e39e7e", "fp_sqrt_of_square": "6db81c7c3bb8001a70db6825237a68d16afbe3ef96ff20d09f166f6fa64fbc0f74b64f128ab6978006d1c34c1c30898", "fr_sqrt_of_square": "2d468a88aca03e5eb9dbf78cc5f44c92b506d77fbba72c1e58edcfb825e39e7e", "fr_sqrt": null, "fp2_sqrt": ["dddcf5cb8350cc85f0ff240bb335215cad168a568a0da66bea228a73f657b82256a8a023397f1ed138ab3247372e685", "16c51d24f20c1a9767d7932911baa8f2d7f312b2965e085c399cb50f5a1c44f875a2037200cb7963f0d47a25dbbabb24"], "fp2_sqrt_of_square": ["6db81c7c3bb8001a70db6825237a68d16afbe3ef96ff20d09f166f6fa64fbc0f74b64f128ab6978006d1c34c1c30898", "1443e64d873318f63ccb594417232f71634dc5dfb781376bd86e4d5f55134f5d035f1309be5fc1357a580b18313fb18d"], "fp2_inv": ["d03dac58e0765a784312f19e7ae790c3453d32959c359c5a819b5f15abd6bccc6e9eb48a625cb70d35ea532a62b13f9", "155c4bfaf04f9da958711a7537e4dd7b55e6a279a405e3441c8a08dcbe06c7f7a0b2cf704b065d930443a45c476d01ae"], "fp_div": "153a8c5d13453a4b38f4b4b74b8f9426ede543dcda1946e374f2b623f8f5c4f6e9e3bfabd005a6d4721d2f5e7af8bf4b"}, {"a": "30c200e8a22e36adad8e4586ae7a3f97ad4aaa338ca9f0c35ab60b911c9dd31c02e7b7eb80eb3ea51ab325b69ec9f7", "b": "55599e8869490c15daa5045056940efe235038593a3ad78c82e6e7e7612aa1dc68d7c58fec3cb71e4b224519db7ebff", "c": "35622a117accb8f34bdb6d535a052c95635bcc4a450217a341c24439fd08ba1d", "fp_sqrt_of_square": "19d04fe950ddb8639d6e1970bc9d3297ccca00dabff868cea3d61c956594585102a91846c5d314c114e44cda4960e0b4", "fr_sqrt_of_square": "35622a117accb8f34bdb6d535a052c95635bcc4a450217a341c24439fd08ba1d", "fr_sqrt": null, "fp2_sqrt": ["faee05e96dbe63b7b07225d2e2b49dfdd1e0c14da372b72d44c4d4dd6c70b6d24b5b483436969df7ee3ceeb273bda8d", "d314c32b24f433d5710445438268bfc061fdf134769f527b526d1da58e07b1f3286c66f4a2e1e29a62f5e8f8127aac0"], "fp2_sqrt_of_square": ["19d04fe950ddb8639d6e1970bc9d3297ccca00dabff868cea3d61c956594585102a91846c5d314c114e44cda4960e0b4", "14ab7801b2eb55d8ed7157713de26be7824247ff5fe165469f026422809e4c06581e83a5b290348dd54cdbae6247beac"], "fp2_inv": ["c877ae1592afe694586aa8b7996014290e3f768d239b1a7bbacc488e72edfad120533aa392a6c7da9f70989a2e0cb29", "c5674761fc1f999304857aa872d4ac80ce4dfee4dd92e35abb3c8106a257960f75716d320f55d558d7b47220d32b682"], "fp_div": "50b5b2048185d5ea0d5d46ddb4653bfd2b909fdce00e6bbbe7200ca9956209b1d7888cea244dbc7a09b527be9c8be9d"}, {"a": "19cb36f11763bec08aa94fd5c0bf780fd5e6b9583c45725158e4711ead88b2cdce5aa6ccb35ad31e3c87a37c3f6c7f66", "b": "56e8e51b187c6f232a59abc75f6f44e3412e951be10401fb043c9c3d5c797a99ceeb384c34f7e34867ee0b2f4aa74ea", "c": "6db17ed227d658a2678c54caa51f1aacc340168d165974882f8d8c962970ce90", "fp_sqrt_of_square": "35daf9221c27d9c07257e0828c34c78e90922cb73fa06e0e4c61824928435650515931fdf92ce17d775c83c0932b45", "fr_sqrt_of_square": "6db17ed227d658a2678c54caa51f1aacc340168d165974882f8d8c962970ce90", "fr_sqrt": "6a940374128dab090a51cc5ff7698768f5c9f1fdc37750ec69a9bf57d733c0bd", "fp2_sqrt": ["1471e9edd6bf4c55d75e056a0f964ac1867fd9fa1f0b2f922cea8f4ccd5dd8085b8c9b44a57e10e0f5fa797f7e3a7067", "142a632148692db7d3818ca240ec386724a7e332db6621f46c8e8679f1fbae64b9a4c336ac522047facc4b98d4c9ec37"], "fp2_sqrt_of_square": ["35daf9221c27d9c07257e0828c34c78e90922cb73fa06e0e4c61824928435650515931fdf92ce17d775c83c0932b45", "1492839887f81fa818760cf9cd54b889306462333574d29fb6ed08dd20e95e7a81bd4c79ee0481cb33801f4d0b5535c1"], "fp2_inv": ["907ab06a8064f0b9324c12e34d9b440f86eef354f9149e0f16c6e9c2ac5141545bc20dbbf2bbe906f8781b71f55394d", "11caa470ea4316e37a2d48e753ae826b2a2b428fe892a585109f94a16a1902623ccd01cb051e1dcb85cb72d0489984e7"], "fp_div": "1515bc6dee2926b2f779fb6d87c4ac092d27d049b58968a8a6b13b5165407301620d71d441abe37d54905bda19c4c11b"}, {"a": "4a89f1b2c2e7fc525b0e7682e913b8e7f0da47dcd56e5b6047bdb0121ba2280db77320bd22687ab86aad588cf39ba2e", "b": "3f57552e0bcf8366eb948a9017b65b9dbef4d964d98db859cad15b814654bcd2c5255bf82b984256c68586f17a96d8e", "c": "3e9ea9ed5f0111856cef067c0a13db244de795c310a49798eeb778e21a3fe8ca", "fp_sqrt_of_square": "155872cf0d5166d5256ac04e14ba7148e569a707262e2d0962b4f79fd4f6d3a34334cdf2df2d785433542a7730c5f07d", "fr_sqrt_of_square": "3e9ea9ed5f0111856cef067c0a13db244de795c310a49798eeb778e21a3fe8ca", "fr_sqrt": "63dbaf4e8e09ae7236e6f0b07383a33b3d7cc8743383292bea8f7c1ed4ae776b", "fp2_sqrt": ["109e41de364bf504efbd9a40af951f3914272351aa981d6cfa807971e0f1cd34076aae924b9273ecdb48eba918ffb79", "edef1a2b953e9e32d7d1db2bc1c9b7e2ac244149e6f4777313bdbe7292c6aed932a789bfc8b0b93ff2b245ccd83850a"], "fp2_sqrt_of_square": ["155872cf0d5166d5256ac04e14ba7148e569a707262e2d0962b4f79fd4f6d3a34334cdf2df2d785433542a7730c5f07d", "160b9c9758c2ee63dc625f0d41d0471d8887fdeea5ec3739ca83bce8e24baa56f259aa3f2e9a7bda4d96a790e8563d1d"], "fp2_inv": ["13c54bd48921112a4e0ba83c78866d65381a912bf0fcf618995911b9a49235b7e393dfc1e26f3cc9cc82189085465d7b", "177ffbc26f87289c13c07d2bdb3028498b21965cebbe3a6ad9ca86b2fe65b1b3057810745f89d908ee599c0ba5afffd2"], "fp_div": "11db0a7bf032c48412a85f35d1c0138e4bb4312b8ba964fcc9f77d363311b28e8e430fe2cf0946a7575255f3b6ed900a"}, {"a": "56bd2891a44c6f21cfa73267f577253ac233882ad8fdff7713fb5c59dc14bdf6980c9f1fe8ff60bcce5350ee95cf3ab", "b": "1ec8a011f798c6152095f866c7ecf4a4969f4eea35fa624d8707419dc143c203dce5012d83ddc98efdd25e8d305130d", "c": "65813fba1798b3abe1df8d1662bcbf04e89d63a73b344808c670c13fd6083b6d", "fp_sqrt_of_square": "14953f611f3b1fa82e21348fc3f43a83b854130245f532c7f5f11cdb58efaa44b52b360cb2c409f3ed19caf116a2b700", "fr_sqrt_of_square": "e6c67991204c99c515a4af1a6e519006b20405bc4ca13f6398f3ebf29f7c494", "fr_sqrt": "52c29eaae3ad4c9f7eb6aa8468264ca6962ab66d1f0d935bea19c4ef1f7f524e", "fp2_sqrt": ["15ea2f1973040eef9a744eb099dcfe755a90e8cc73509179dbfadf4b0fa555cd0664c6626eff60c9bbb938fee30d0b1c", "10fa37e8713a458f909c39cf0c0e22053d0db97bfe22631fab3a65b2b455f44cd3e443b67f279a0641ac739aa89b14c9"], "fp2_sqrt_of_square": ["14953f611f3b1fa82e21348fc3f43a83b854130245f532c7f5f11cdb58efaa44b52b360cb2c409f3ed19caf116a2b700", "181487e91a065a38f912482fd6ccdd8d1b0d569650256c9a8ec05e871a9cba03e0ddafebd9162366ca21da172cfa979e"], "fp2_inv": ["26360366bc4ad86772f383d29cd13421584ff41fb22c0783c3121eef38b5cc3685736218356d7bab4c6a6e5d5bc6c5b", "16a3d8ac8d63a1f140b32159f9819245534f6a06c83f848afeab3b8fe3fca0e1bbd2e2c8cd5b6aa41d86e6ba18914323"], "fp_div": "1fed9d55b6e5832fe5574bce6e85db44fdfe85c212a32bcb51c3e19234815dd844c2d69475ac4cf135421c841f20127"}, {"a": "f283201353fe94507405ffe9b31d8df9cb7ce4d948fc1ca47eeb0e83d25bcd58b1a2acf2e64803fc38f31d57c3f75d8", "b": "e0eb5d1f5efd4c188bb5a989f1393ee43088c03e71f8e571982fdcecb122103dc4cf2b4b6369e27ed8d544d44f46e2a", "c": "491132c71c63f191d4ecde98ed4e5575bd6871c500a82c42fe275cf3c488d4e3", "fp_sqrt_of_square": "ad8dfe9043ffd5543db47b7a819d3f7c7bf7d375ef550f51f4221b8b98b394e9391d52f82ef7fbff66fce2a83c034d3", "fr_sqrt_of_square": "491132c71c63f191d4ecde98ed4e5575bd6871c500a82c42fe275cf3c488d4e3", "fr_sqrt": "701f1e6dace483e818f67fb1edc5039e45a3024853617ca0ac38bab76c0a886f", "fp2_sqrt": ["15c617950589f5baaf74565e1e907ed7c397c5c201e48996829cfc33185163c99dae0990905a83b39ebb8fb35200968a", "17952092484638e27fdb0a0ed34c467c6390e34c6bf8c9407dd6ea39250bdd5cef9216ca8b1fe7890135a3468058bab7"], "fp2_sqrt_of_square": ["f283201353fe94507405ffe9b31d8df9cb7ce4d948fc1ca47eeb0e83d25bcd58b1a2acf2e64803fc38f31d57c3f75d8", "e0eb5d1f5efd4c188bb5a989f1393ee43088c03e71f8e571982fdcecb122103dc4cf2b4b6369e27ed8d544d44f46e2a"], "fp2_inv": ["6998decd77b4571ba2e208ed9dd9d03eb40eeebb67a3f0c8b12cdddd3fdf466c614b4216ae83524824a504a85e2184e", "920ad18056b6caa770623c43fb529c3c2d6e7fb6e5cb074865ab3cb223a0f9c665f3d4f88aceff771242d96f1f8ccd2"], "fp_div": "197441b89e269250a4caeb7e81e301be77a12f20a44169c57be39d23c888702332551d1ac3d75b405d62ea7ef1eaef99"}];
    for (const k of KAT) {
      const a = BigInt('0x' + k.a), b = BigInt('0x' + k.b), c = BigInt('0x' + k.c);
      assert.strictEqual(new Fp(a).square().sqrt().value.toString(16), k.fp_sqrt_of_square);
      assert.strictEqual(new Fp(a).div(new Fp(b)).value.toString(16), k.fp_div);
      const fs = new Fr(c).sqrt();
      assert.strictEqual(fs ? fs.value.toString(16) : null, k.fr_sqrt);
      assert.strictEqual(new Fr(c).square().sqrt().value.toString(16), k.fr_sqrt_of_square);
      const X = Fp2.fromBigTuple([a, b]), s = X.sqrt(), ss = X.square().sqrt(), inv = X.invert();
      assert.deepStrictEqual(s ? [s.c0.value.toString(16), s.c1.value.toString(16)] : null, k.fp2_sqrt);
      assert.deepStrictEqual([ss.c0.value.toString(16), ss.c1.value.toString(16)], k.fp2_sqrt_of_square);
      assert.deepStrictEqual([inv.c0.value.toString(16), inv.c1.value.toString(16)], k.fp2_inv);
      assert.ok(X.multiply(inv).equals(Fp2.ONE) && X.pow(Fp2.ORDER).equals(Fp2.ONE) && X.frobeniusMap(1).equals(X.pow(bls.CURVE.P)));
      assert.ok(Fp2.fromBytes(X.toBytes()).equals(X) && Fp.fromBytes(new Fp(a).toBytes()).equals(new Fp(a)));
    }
    assert.strictEqual(Fp.BYTES_LEN, 48); assert.strictEqual(Fp2.BYTES_LEN, 96); assert.strictEqual(Fp.MAX_BITS, 381);
    assert.strictEqual(bls.CURVE.h * bls.CURVE.r, bls.CURVE.P + bls.CURVE.x);             // #E(Fp) = p + 1 - t with t = z + 1, z = -x
    assert.throws(() => new Fp(0n).invert(), /invert: expected positive integers/);
    assert.throws(() => new Fp2(1n, 2n), /c0: Expected Fp/);
  }
  // Fp12 on the host against the GPU: e(P, Q)^2 = e(2P, Q) = e(P, Q) * e(P, Q); inverse = conjugate for pairing values; Frobenius
  {
    const G = bls.PointG1.BASE, H = bls.PointG2.BASE;
    const e1 = bls.pairing(G.multiply(3n), H), e2 = bls.pairing(G.multiply(6n), H);
    assert.ok(e1.multiply(e1).equals(e2) && e1.square().equals(e2) && e1.pow(2n).equals(e2));
    assert.ok(e1.invert().equals(e1.conjugate()) && e1.multiply(e1.invert()).equals(bls.Fp12.ONE));
    assert.ok(e1.frobeniusMap(12).equals(e1) && e1.frobeniusMap(1).equals(e1.pow(bls.CURVE.P)));
    assert.ok(bls.pairing(G, H, false).finalExponentiate().equals(bls.pairing(G, H)));
    assert.ok(bls.Fp12.fromBytes(e1.toBytes()).equals(e1) && !e1.isZero());
  }
  // the reference's coordinate view of points: constructor from field elements (projective), x / y / z, toAffine
  {
    const { Fp, Fp2, PointG1, PointG2, CURVE } = bls;
    const G = new PointG1(new Fp(CURVE.Gx), new Fp(CURVE.Gy));
    assert.ok(G.equals(PointG1.BASE) && G.x.value === CURVE.Gx && G.z.equals(Fp.ONE));
    const lam = new Fp(12345n);
    assert.ok(new PointG1(G.x.multiply(lam), G.y.multiply(lam), lam).equals(G));         // (lx : ly : l) is the same point
    assert.ok(new PointG1(Fp.ONE, Fp.ONE, Fp.ZERO).isZero() && PointG1.ZERO.z.isZero());
    const H = new PointG2(Fp2.fromBigTuple(CURVE.G2x), Fp2.fromBigTuple(CURVE.G2y), Fp2.ONE);
    assert.ok(H.equals(PointG2.BASE) && H.toAffine()[0].equals(Fp2.fromBigTuple(CURVE.G2x)));
    assert.ok(H.fromAffineTuple(H.double().toAffine()).equals(H.multiply(2n)));
    assert.ok(G.millerLoop(H).finalExponentiate().equals(bls.pairing(G, H)));
    assert.ok(G.multiply(5n).clearCofactor().equals(G.multiply(5n * (CURVE.x + 1n) % CURVE.r)));
  }
  // utils: expand_message_xmd / hash_to_field on the host against the device path (hashToCurve of the same message goes through the
  // device SHA-256), key derivation helpers
  {
    const msg = un('abcdef0123456789');
    const xmd = await bls.utils.expandMessageXMD(msg, bls.utils.stringToBytes('QUUX-V01-CS02-with-expander-SHA256-128'), 0x20);
    assert.strictEqual(hex(await bls.utils.expandMessageXMD(bls.utils.stringToBytes('abc'), bls.utils.stringToBytes('QUUX-V01-CS02-with-expander-SHA256-128'), 0x20)),
      'd8ccab23b5985ccea865c6c97b6e5b8350e794e603b4b97902f53a8a0d605615');       // RFC 9380 appendix K.1
    assert.strictEqual(xmd.length, 32);
    const u = await bls.utils.hashToField(msg, 2);
    assert.ok(u.length === 2 && u[0].length === 2 && u.every((e) => e.every((v) => typeof v === 'bigint' && v < bls.CURVE.P)));
    const k = bls.utils.hashToPrivateKey(new Uint8Array(40).fill(7));
    assert.ok(k.length === 32 && hex(bls.getPublicKey(k)).length === 96);
    assert.throws(() => bls.utils.hashToPrivateKey(new Uint8Array(39)), /Expected 40-1024 bytes/);
    assert.strictEqual(bls.utils.randomPrivateKey().length, 32);
    assert.strictEqual(bls.utils.mod(-1n, 5n), 4n);
    assert.strictEqual(hex(await bls.utils.sha256(bls.utils.stringToBytes('abc'))), 'ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad');
  }
  // round 2: every wire form with the reference's own error messages (vectors of tools/gen_golden2.mjs), public clearCofactor, prepared tables
  {
    const { PointG1, PointG2 } = bls;
    const check = (fn, v, affOf) => {
      if (v.result === 'ok') assert.strictEqual(hex(affOf(fn(v.hex))), v.aff);
      else if (v.result === 'zero') assert.ok(fn(v.hex).isZero());
      else assert.throws(() => fn(v.hex), (e) => e.message === v.result, v.result + ' for ' + v.hex.slice(0, 16));
    };
    for (const v of gold2.g2_fromhex96) check((h) => PointG2.fromHex(h), v, (p) => p.aff);
    for (const v of gold2.g2_fromsig192) check((h) => PointG2.fromSignature(h), v, (p) => p.aff);
    for (const v of gold2.g1_raw96) check((h) => PointG1.fromHex(h), v, (p) => p.aff);
    for (const v of gold2.g2_raw192) check((h) => PointG2.fromHex(h), v, (p) => p.aff);
    for (const v of gold2.g1_raw96.filter((x) => x.compressed)) { const p = PointG1.fromHex(v.hex); assert.strictEqual(p.toHex(false), v.hex); assert.strictEqual(p.toHex(true), v.compressed); }
    for (const v of gold2.g2_raw192.filter((x) => x.compressed)) { const p = PointG2.fromHex(v.hex); assert.strictEqual(p.toHex(false), v.hex); assert.strictEqual(p.toHex(true), v.compressed); }
    for (const v of gold2.g2_clear_cofactor) assert.strictEqual(hex(new PointG2(un(v.aff)).clearCofactor().aff), v.out);
    for (const v of gold2.g1_clear_cofactor) assert.strictEqual(hex(new PointG1(un(v.aff)).clearCofactor().aff), v.out);
    // PointG2.pairingPrecomputes (index.ts:703-711): 68 triples, first / last equal to the reference's; millerLoop over the memoised table
    const crypto = require('crypto');
    for (const v of gold.pairs.slice(0, 3)) {
      const Q = new PointG2(un(v.g2)), P = new PointG1(un(v.g1));
      const ell = Q.pairingPrecomputes();
      assert.strictEqual(ell.length, 68);
      const bytes = Buffer.concat(ell.map((t) => Buffer.concat(t.map((c) => Buffer.from(c.toBytes())))));
      assert.strictEqual(bytes.slice(0, 288).toString('hex'), v.ell_first);
      assert.strictEqual(crypto.createHash('sha256').update(bytes).digest('hex'), v.ell_sha256);
      assert.strictEqual(Q.pairingPrecomputes(), ell);                                   // memoised
      assert.strictEqual(hex(P.millerLoop(Q).toBytes()), v.miller);                      // prepared path
      Q.clearPairingPrecomputes();
      assert.strictEqual(hex(P.millerLoop(Q).toBytes()), v.miller);                      // plain path
    }
  }
  // round 3: (i) the reference's whole kilic vector set (test/deterministic.test.ts:34-46: e(i G1, i G2), i = 1..1000) in ONE pairingBatch call, compared in
  // kilic's own byte order through Fp12.toKilicBytes / fromKilicBytes (:9-12, 41; SURVEY 8(f).4); (ii) the doubling known answers of test/point.test.ts:89-146,
  // 270-346 -- projective inputs, a.double() against the reference's projective double, a.multiply(2n) and a.add(a); (iii) flag bits over uncompressed input
  {
    const { Fp, Fp2, Fp12, PointG1, PointG2 } = bls;
    const N = td.pairing_iG1_iG2.length;
    assert.strictEqual(N, 1000);
    const ks = Array.from({ length: N }, (_, i) => BigInt(i + 1));
    const g1s = [], g2s = [];
    // i G1 / i G2 by repeated addition would be 2000 single-point calls; the batched scalar multiplication does it in two
    const sc = ks.map((k) => un(k.toString(16).padStart(64, '0')));
    const pk = bls.getPublicKeys(sc);                                                   // 48-byte compressed i G1
    for (let i = 0; i < N; i++) g1s.push(PointG1.fromHex(pk[i]));
    const h = bls.PointG2.BASE;
    let acc = h;
    for (let i = 0; i < N; i++) { g2s.push(acc); if (i + 1 < N) acc = (i % 50 === 49) ? h.multiply(BigInt(i + 2)) : acc.add(h); }   // additions, re-anchored on multiply every 50
    const { out, status } = bls.pairingBatch(g1s, g2s, true, true);
    assert.ok(status.every((x) => x === 0));
    for (let i = 0; i < N; i++) {
      const e = Fp12.fromBytes(out.subarray(576 * i, 576 * (i + 1)));
      const kilic = td.pairing_iG1_iG2[i].match(/.{96}/g).reverse().join('');           // the file's own order (tools/gen_golden.py stores it reversed)
      assert.strictEqual(hex(e.toKilicBytes()), kilic, 'kilic vector ' + (i + 1));
      if (i < 8) assert.ok(Fp12.fromKilicBytes(kilic).equals(e) && Fp12.fromKilicBytes(un(kilic)).equals(e));
    }
    assert.strictEqual(hex(bls.pairing(PointG1.BASE, PointG2.BASE).toKilicBytes()), td.e_G1_G2.match(/.{96}/g).reverse().join(''));
    assert.throws(() => Fp12.fromKilicBytes(new Uint8Array(575)), /wrong length/);
    const f = (hx_) => new Fp(BigInt('0x' + hx_)), f2 = (a, b) => Fp2.fromBigTuple([BigInt('0x' + a), BigInt('0x' + b)]);
    for (const v of td.point_double_kats.g1) {
      const a = new PointG1(f(v.a[0]), f(v.a[1]), f(v.a[2])), want = new PointG1(f(v.double[0]), f(v.double[1]), f(v.double[2]));
      const d = a.double(); d.assertValidity();
      assert.ok(d.equals(want) && d.equals(a.multiply(2n)) && d.equals(a.add(a)));
    }
    for (const v of td.point_double_kats.g2) {
      const a = new PointG2(f2(v.a[0], v.a[1]), f2(v.a[2], v.a[3]), f2(v.a[4], v.a[5])), want = new PointG2(f2(v.double[0], v.double[1]), f2(v.double[2], v.double[3]), f2(v.double[4], v.double[5]));
      const d = a.double(); d.assertValidity();
      assert.ok(d.equals(want) && d.equals(a.multiply(2n)) && d.equals(a.add(a)));
    }
    for (const k of td.wnaf_scalars) {                                                   // test/point.test.ts:347-372: multiply == multiplyUnsafe on the wNAF scalar list
      const s_ = BigInt('0x' + k);
      assert.ok(s_ > 0n && s_ < bls.CURVE.r);
      assert.ok(PointG1.BASE.multiply(s_).equals(PointG1.BASE.multiplyUnsafe(s_)) && PointG1.BASE.multiply(s_).equals(PointG1.BASE.multiply(s_ - 1n).add(PointG1.BASE)));
      assert.ok(PointG2.BASE.multiply(s_).equals(PointG2.BASE.multiplyUnsafe(s_)) && PointG2.BASE.multiply(s_).equals(PointG2.BASE.multiply(s_ - 1n).add(PointG2.BASE)));
    }
    const check3 = (fn, v) => {
      if (v.result === 'ok') assert.strictEqual(hex(fn(v.hex).aff), v.aff);
      else if (v.result === 'zero') assert.ok(fn(v.hex).isZero());
      else assert.throws(() => fn(v.hex), (e) => e.message === v.result, v.result + ' for ' + v.hex.slice(0, 16));
    };
    for (const v of gold3.g1_raw96_flags) check3((h_) => PointG1.fromHex(h_), v);
    for (const v of gold3.g2_raw192_flags) check3((h_) => PointG2.fromHex(h_), v);
  }
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
  // calls in flight: with NBLS_CONTEXTS > 1 (tests/test_js_facade.py runs this file a second time with 3) concurrent verifyBatch promises run on
  // separate engine contexts and overlap on the GPU; the answers are those of the sequential calls above
  const many = await Promise.all([bls.verifyBatch(vb.agg_sig, vb.msgs, vb.pks), bls.verifyBatch(vb.agg_sig, m2, vb.pks), bls.verifyBatch(vb.agg_sig, vb.msgs, p2),
    bls.verifyBatch(vb.agg_sig, vb.msgs, vb.pks), bls.verifyBatch(vb.agg_sig, vb.msgs, vb.pks), bls.verifyBatch(vb.agg_sig, m2, vb.pks)]);
  assert.deepStrictEqual(many, [true, false, false, true, true, false]);
  // verify / sign of wire-format inputs run on a worker thread (reference: both are async, index.ts:744-767): the event loop keeps turning while the GPU works,
  // exceptions are the reference's, and the wall time of one call from JavaScript is reported (what a user of the facade gets; bench.py records the same figures)
  {
    const s0 = gold.sigs[0];
    let ticks = 0, live = true;
    const spin = () => { if (live) { ticks++; setImmediate(spin); } };
    setImmediate(spin);
    const t0 = process.hrtime.bigint();
    const N = 20;
    for (let i = 0; i < N; i++) assert.strictEqual(await bls.verify(s0.sig, s0.msg, s0.pk), true);
    const t1 = process.hrtime.bigint();
    const ticksVerify = ticks;
    const keys = gold.sigs.slice(0, 4);
    const sigs = [];
    for (const v of keys) sigs.push(hex(await bls.sign(v.msg, v.sk)));
    const t2 = process.hrtime.bigint();
    live = false;
    assert.deepStrictEqual(sigs, keys.map((v) => v.sig));
    assert.ok(ticksVerify >= N, 'the event loop did not run during verify(): ' + ticksVerify + ' turns in ' + N + ' calls');
    assert.ok(ticks > ticksVerify, 'the event loop did not run during sign()');
    console.log('facade timing: verify %s ms per call (event-loop turns meanwhile: %d), sign %s ms per call', (Number(t1 - t0) / 1e6 / N).toFixed(3), ticksVerify, (Number(t2 - t1) / 1e6 / keys.length).toFixed(3));
    // the fast path must raise what the reference raises: an undecodable key, a key off the subgroup, the point at infinity
    const bad = vb.pks[0].slice(0, 2) + 'ff' + vb.pks[0].slice(4);
    await assert.rejects(() => bls.verify(s0.sig, s0.msg, bad), (e) => e instanceof Error);
    const inf1 = 'c0' + '00'.repeat(47), inf2 = 'c0' + '00'.repeat(95);
    await assert.rejects(() => bls.verify(s0.sig, s0.msg, inf1), /No pairings at point of Infinity/);
    await assert.rejects(() => bls.verify(inf2, s0.msg, s0.pk), /No pairings at point of Infinity/);
    assert.strictEqual(await bls.verify(s0.sig, s0.msg, gold.sigs[1].pk), false);
  }
  console.log('JS facade ok');
})().catch((e) => { console.error(e); process.exit(1); });

// oracle/js_bigint_pairing.js -- TEST / BENCHMARK INFRASTRUCTURE ONLY (never loaded by the product: noble-bls12-381_amd/js/index.js has no CPU pairing path).
//
// A single-threaded BigInt pairing in the reference's own language and number representation: the algorithm of paulmillr/noble-bls12-381 v1.4.0
// (calcPairingPrecomputes + millerLoop math.ts:1331-1388, Fp12 multiplyBy014 / square math.ts:768-791, cyclotomicSquare / cyclotomicExp math.ts:811-852,
// finalExponentiate math.ts:856-874) restated over the facade's own field classes (noble-bls12-381_amd/js/fields.js).  The reference itself cannot
// travel to the GPU box; this file is what bench.py times there as `cpu_baseline.js_bigint` (SURVEY 8(d)(ii)), and its speed relative to the real
// reference was measured once in the build container under the same Node (BASELINE.md).  Checked against the reference's e(G1, G2) vector on every run.
//
//   node oracle/js_bigint_pairing.js [seconds]      -> one JSON line {pairings, seconds, pairings_per_s, node, ok}
'use strict';
const path = require('path');
const { Fp, Fp2, Fp6, Fp12 } = require(path.join(__dirname, '..', 'noble-bls12-381_amd', 'js', 'fields.js'));
const X = 0xd201000000010000n;
const XBITS = 64;
const bit = (i) => (X >> BigInt(i)) & 1n;

// the 68 line triples of a G2 point (affine coordinates as Fp2), math.ts:1331-1371
function linesOf(Qx, Qy) {
  let Rx = Qx, Ry = Qy, Rz = Fp2.ONE;
  const ell = [];
  for (let i = XBITS - 2; i >= 0; i--) {
    const t0 = Ry.square(), t1 = Rz.square(), t2 = t1.multiply(3n).multiplyByB(), t3 = t2.multiply(3n);
    const t4 = Ry.add(Rz).square().subtract(t1).subtract(t0);
    ell.push([t2.subtract(t0), Rx.square().multiply(3n), t4.negate()]);
    const nRx = t0.subtract(t3).multiply(Rx).multiply(Ry).div(2n);
    const nRy = t0.add(t3).div(2n).square().subtract(t2.square().multiply(3n));
    Rz = t0.multiply(t4); Rx = nRx; Ry = nRy;
    if (bit(i)) {
      const a0 = Ry.subtract(Qy.multiply(Rz)), a1 = Rx.subtract(Qx.multiply(Rz));
      ell.push([a0.multiply(Qx).subtract(a1.multiply(Qy)), a0.negate(), a1]);
      const a2 = a1.square(), a3 = a2.multiply(a1), a4 = a2.multiply(Rx);
      const a5 = a3.subtract(a4.multiply(2n)).add(a0.square().multiply(Rz));
      const nx = a1.multiply(a5), ny = a4.subtract(a5).multiply(a0).subtract(a3.multiply(Ry));
      Rz = Rz.multiply(a3); Rx = nx; Ry = ny;
    }
  }
  return ell;
}
// Fp6 times (b0 + b1 v) and times (b1 v), math.ts:631-651
const mul6By01 = (a, b0, b1) => new Fp6(a.c2.multiply(b1).mulByNonresidue().add(a.c0.multiply(b0)), a.c0.multiply(b1).add(a.c1.multiply(b0)), a.c1.multiply(b1).add(a.c2.multiply(b0)));
const mul6By1 = (a, b1) => new Fp6(a.c2.multiply(b1).mulByNonresidue(), a.c0.multiply(b1), a.c1.multiply(b1));
// f * (o0 + o1 v + o4 v w), math.ts:768-777
function mulBy014(f, o0, o1, o4) {
  const t0 = mul6By01(f.c0, o0, o1), t1 = mul6By1(f.c1, o4);
  return new Fp12(t1.mulByNonresidue().add(t0), mul6By01(f.c1.add(f.c0), o0, o1.add(o4)).subtract(t0).subtract(t1));
}
// math.ts:783-791
function sqr12(f) {
  const ab = f.c0.multiply(f.c1);
  return new Fp12(f.c1.mulByNonresidue().add(f.c0).multiply(f.c0.add(f.c1)).subtract(ab).subtract(ab.mulByNonresidue()), ab.add(ab));
}
// math.ts:1373-1388
function millerLoop(ell, Px, Py) {
  let f = Fp12.ONE;
  for (let j = 0, i = XBITS - 2; i >= 0; i--, j++) {
    f = mulBy014(f, ell[j][0], ell[j][1].multiply(Px.value), ell[j][2].multiply(Py.value));
    if (bit(i)) { j++; f = mulBy014(f, ell[j][0], ell[j][1].multiply(Px.value), ell[j][2].multiply(Py.value)); }
    if (i !== 0) f = sqr12(f);
  }
  return f.conjugate();
}
// math.ts:811-843
const sq4 = (a, b) => { const a2 = a.square(), b2 = b.square(); return [b2.mulByNonresidue().add(a2), a.add(b).square().subtract(a2).subtract(b2)]; };
function cycSqr(x) {
  const [t3, t4] = sq4(x.c0.c0, x.c1.c1), [t5, t6] = sq4(x.c1.c0, x.c0.c2), [t7, t8] = sq4(x.c0.c1, x.c1.c2), t9 = t8.mulByNonresidue();
  const f = (t, g, s) => (s < 0 ? t.subtract(g) : t.add(g)).multiply(2n).add(t);
  return new Fp12(new Fp6(f(t3, x.c0.c0, -1), f(t5, x.c0.c1, -1), f(t7, x.c0.c2, -1)), new Fp6(f(t9, x.c1.c0, 1), f(t4, x.c1.c1, 1), f(t6, x.c1.c2, 1)));
}
// math.ts:845-852
function cycExp(x) { let z = Fp12.ONE; for (let i = XBITS - 1; i >= 0; i--) { z = cycSqr(z); if (bit(i)) z = z.multiply(x); } return z; }
// math.ts:856-874
function finalExponentiate(f) {
  const t0 = f.frobeniusMap(6).div(f), t1 = t0.frobeniusMap(2).multiply(t0);
  const t2 = cycExp(t1).conjugate(), t3 = cycSqr(t1).conjugate().multiply(t2);
  const t4 = cycExp(t3).conjugate(), t5 = cycExp(t4).conjugate(), t6 = cycExp(t5).conjugate().multiply(cycSqr(t2)), t7 = cycExp(t6).conjugate();
  return t2.multiply(t5).frobeniusMap(2).multiply(t4.multiply(t1).frobeniusMap(3)).multiply(t6.multiply(t1.conjugate()).frobeniusMap(1)).multiply(t7.multiply(t3.conjugate()).multiply(t1));
}
function pairing(Px, Py, Qx, Qy, withFinalExponent = true) {
  const f = millerLoop(linesOf(Qx, Qy), Px, Py);
  return withFinalExponent ? finalExponentiate(f) : f;
}
module.exports = { pairing, millerLoop, linesOf, finalExponentiate };

if (require.main === module) {
  const seconds = Number(process.argv[2] || 5);
  const Gx = new Fp(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bbn), Gy = new Fp(0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1n);
  const Hx = Fp2.fromBigTuple([0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8n, 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7en]);
  const Hy = Fp2.fromBigTuple([0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801n, 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79ben]);
  // e(G1, G2).c0.c0.c0 of the reference's test/pairing.test.ts:50
  const ok = pairing(Gx, Gy, Hx, Hy).c0.c0.c0.value === 0x1250ebd871fc0a92a7b2d83168d0d727272d441befa15c503dd8e90ce98db3e7b6d194f60839c508a84305aaca1789b6n;
  let n = 0; const t0 = process.hrtime.bigint();
  while (Number(process.hrtime.bigint() - t0) / 1e9 < seconds) { pairing(Gx, Gy, Hx, Hy); n++; }
  const dt = Number(process.hrtime.bigint() - t0) / 1e9;
  console.log(JSON.stringify({ pairings: n, seconds: Number(dt.toFixed(3)), pairings_per_s: Number((n / dt).toFixed(2)), node: process.version, ok }));
}

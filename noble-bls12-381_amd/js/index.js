/*
 * index.js -- drop-in facade with the exported names of paulmillr/noble-bls12-381 v1.4.0 (reference index.ts:715-821)
 * over the MI355X engine (libnbls.so via the N-API addon).  Argument handling, error messages and result encodings follow
 * the reference; the arithmetic runs on the GPU.  Additive batched entry points: pairingBatch, millerProduct.
 *
 * getPublicKey / sign run the double-and-add-always ladders of the engine (SURVEY 8(f).1); additive: getPublicKeys, signBatch.
 * The reference's re-exported field classes Fp, Fr, Fp2 (index.ts:22) are single-element bigint helpers outside the batched hot
 * path: fields.js provides them on the host, together with Fp6 and Fp12 (finalExponentiate goes to the GPU).
 */
'use strict';
const path = require('path');
const native = require(path.join(__dirname, 'nbls_napi.node'));
const { Fp, Fr, Fp2, Fp6, Fp12 } = require('./fields.js');

// curve parameters under the reference's key names (math.ts:13-63).  h = (z - 1)^2 / 3 and h2 = (z^8 - 4z^7 + 5z^6 - 4z^4 + 6z^3 - 4z^2 - 4z + 13) / 9
// with z = -x; h2Eff is the effective G2 cofactor of RFC 9380 section 8.8.2; P2 is p^2 - 1 as in the reference.
const P_ = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn;
const CURVE = {
  P: P_,
  r: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n,
  h: 0x396c8c005555e1568c00aaab0000aaabn,
  Gx: 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bbn,
  Gy: 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1n,
  b: 4n,
  P2: P_ ** 2n - 1n,
  h2: 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5n,
  G2x: [0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8n, 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7en],
  G2y: [0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801n, 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79ben],
  b2: [4n, 4n],
  x: 0xd201000000010000n,
  h2Eff: 0xbc69f08f2ee75b3584c6a0ea91b352888e2a8e9145ad7689986ff031508ffe1329c2f178731db956d82bf015d1212b02ec0ec69d7477c1ae954cbc06689f6a359894c0adebbf6b4e8020005aaa95551n,
};
const htfDefaults = { DST: 'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_' };      // reference index.ts:60-81
let inited = false;
// NBLS_DEVICES = "0,1,2,3" (or "all") opens every listed GPU behind one handle: pairing batches, Miller products and verifyBatch then shard
// over them inside the library (nbls_multi_*, include/nbls.h); NBLS_DEVICE = n selects a single GPU (default 0).
function ensureInit() {
  if (inited) return;
  const list = process.env.NBLS_DEVICES;
  if (list) native.initMulti(list === 'all' ? null : Int32Array.from(list.split(',').map(Number)));
  else native.init(Number(process.env.NBLS_DEVICE || 0));
  inited = true;
}

// ---- byte helpers (reference math.ts:158-212)
function hexToBytes(hex) {
  if (typeof hex !== 'string') throw new TypeError('hexToBytes: expected string, got ' + typeof hex);
  if (hex.length % 2) throw new Error('hexToBytes: received invalid unpadded hex');
  const a = new Uint8Array(hex.length / 2);
  for (let i = 0; i < a.length; i++) { const b = Number.parseInt(hex.slice(2 * i, 2 * i + 2), 16); if (Number.isNaN(b)) throw new Error('Invalid byte sequence'); a[i] = b; }
  return a;
}
const hexes = Array.from({ length: 256 }, (v, i) => i.toString(16).padStart(2, '0'));
function bytesToHex(u8) { let s = ''; for (let i = 0; i < u8.length; i++) s += hexes[u8[i]]; return s; }
function ensureBytes(hex) { return hex instanceof Uint8Array ? Uint8Array.from(hex) : hexToBytes(hex); }
function concat(...arrs) { const n = arrs.reduce((a, b) => a + b.length, 0); const r = new Uint8Array(n); let o = 0; for (const a of arrs) { r.set(a, o); o += a.length; } return r; }
function toBig(u8) { return BigInt('0x' + (bytesToHex(u8) || '0')); }
function stringToBytes(str) { const b = new Uint8Array(str.length); for (let i = 0; i < str.length; i++) b[i] = str.charCodeAt(i); return b; }
function isZeroBytes(u8) { for (let i = 0; i < u8.length; i++) if (u8[i]) return false; return true; }
const G1_STATUS = { 2: 'Invalid G1 point: not on curve Fp', 3: 'Invalid G1 point: must be of prime-order subgroup', 4: 'Invalid compressed G1 point' };
const G2_STATUS = { 2: 'Invalid G2 point: not on curve Fp2', 3: 'Invalid G2 point: must be of prime-order subgroup', 4: 'Failed to find a square root' };

// ---- Fp12: the host class of fields.js (coordinates, math.ts:705-885 method names); the final exponentiation runs on the GPU
Fp12.prototype.finalExponentiate = function () { ensureInit(); return Fp12.fromBytes(native.finalExpBatch(this.toBytes())); };

// ---- points: affine wire bytes, or the zero point
// reference math.ts:1035-1043
function validateScalar(n) {
  if (typeof n === 'number') n = BigInt(n);
  if (typeof n !== 'bigint' || n <= 0 || n > CURVE.r) throw new Error(`Point#multiply: invalid scalar, expected positive integer < CURVE.r. Got: ${n}`);
  return n;
}
const scalarBytes = (n) => hexToBytes(validateScalar(n).toString(16).padStart(64, '0'));

// Points are held as affine wire bytes (what the engine consumes) or as the zero point.  The reference's constructor form
// new PointG1(x: Fp, y: Fp, z?: Fp) (index.ts:291) is accepted too: the projective triple is made affine on the host.
class PointG1 {
  constructor(aff, zero = false, z) {
    if (aff instanceof Fp) {
      const x = aff, y = zero; z = z === undefined ? Fp.ONE : z;
      if (!(y instanceof Fp) || !(z instanceof Fp)) throw new Error('Expected Fp coordinates');
      if (z.isZero()) { this.aff = new Uint8Array(96); this.zero = true; return; }
      const zi = z.invert();
      this.aff = concat(x.multiply(zi).toBytes(), y.multiply(zi).toBytes()); this.zero = false; return;
    }
    this.aff = aff; this.zero = zero;
  }
  // coordinates as the reference exposes them (affine representative, z = 1; the zero point is (1, 1, 0) as getZero(), math.ts:910-912)
  get x() { return this.zero ? Fp.ONE : Fp.fromBytes(this.aff.subarray(0, 48)); }
  get y() { return this.zero ? Fp.ONE : Fp.fromBytes(this.aff.subarray(48)); }
  get z() { return this.zero ? Fp.ZERO : Fp.ONE; }
  getZero() { return PointG1.ZERO; }
  fromAffineTuple(xy) { return new PointG1(xy[0], xy[1], Fp.ONE); }
  toAffineBatch(points) { return points.map((p) => p.toAffine()); }
  normalizeZ(points) { return points; }
  toString() { return this.zero ? 'Point<Zero>' : `Point<x=${this.x}, y=${this.y}>`; }
  // index.ts:452-454: millerLoop(P.pairingPrecomputes(), this.toAffine()); a Q whose table is already memoised is paired through the prepared path
  millerLoop(Q) {
    if (Q._lineTable && !this.zero) { ensureInit(); return Fp12.fromBytes(native.pairingPrepared(this.aff, Q._lineTable, false, false)); }
    return pairing(this, Q, false);
  }
  clearCofactor() {                                                                       // [x]P + P with x = |z| (index.ts:401-405), for any point of the curve
    if (this.zero) return this;
    ensureInit();
    const { out, status } = native.clearCofactor(false, this.aff);
    return status[0] === 1 ? PointG1.ZERO : new PointG1(out);
  }
  calcMultiplyPrecomputes() {} clearMultiplyPrecomputes() {}                               // window tables of the reference's host ladder: nothing to cache here
  static get ZERO() { return new PointG1(new Uint8Array(96), true); }
  static get BASE() {
    return new PointG1(hexToBytes('17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb' +
      '08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1'));
  }
  isZero() { return this.zero; }
  // reference index.ts:298-327
  // reference index.ts:331-350
  static async hashToCurve(msg, options) {
    msg = ensureBytes(msg); ensureInit();
    const dst = stringToBytes((options && options.DST) || htfDefaults.DST);
    return new PointG1(native.hashToCurve(0, msg, Uint32Array.from([0, msg.length]), dst));
  }
  static async encodeToCurve(msg, options) {
    msg = ensureBytes(msg); ensureInit();
    const dst = stringToBytes((options && options.DST) || htfDefaults.DST);
    return new PointG1(native.hashToCurve(1, msg, Uint32Array.from([0, msg.length]), dst));
  }
  static fromHex(bytes) {
    bytes = ensureBytes(bytes); ensureInit();
    if (bytes.length === 48) {
      const { out, status } = native.g1Decompress(bytes);
      if (status[0] === 1) return PointG1.ZERO;
      if (status[0]) throw new Error(G1_STATUS[status[0]]);
      return new PointG1(out);
    } else if (bytes.length === 96) {
      const { out, status } = native.decodePoints(0, bytes, 96);      // infinity flag, coordinates reduced by the Fp constructor, assertValidity (index.ts:317-325)
      if (status[0] === 1) return PointG1.ZERO;
      if (status[0]) throw new Error(G1_STATUS[status[0]]);
      return new PointG1(out);
    }
    throw new Error('Invalid point G1, expected 48/96 bytes');
  }
  assertValidity() {
    if (this.zero) return this;
    ensureInit();
    const st = native.g1Validate(this.aff).status[0];
    if (st) throw new Error(G1_STATUS[st]);
    return this;
  }
  negate() {
    if (this.zero) return this;
    const y = toBig(this.aff.subarray(48));
    const ny = y === 0n ? 0n : CURVE.P - y;
    return new PointG1(concat(this.aff.subarray(0, 48), hexToBytes(ny.toString(16).padStart(96, '0'))));
  }
  add(rhs) { return PointG1.sum([this, rhs]); }
  subtract(rhs) { return this.add(rhs.negate()); }
  double() { return this.add(this); }
  // reference math.ts:1048-1167 (multiplyUnsafe / multiply / multiplyPrecomputed: same group element), the engine's ladder
  multiply(scalar) {
    const k = scalarBytes(scalar);
    if (this.zero) return PointG1.ZERO;
    ensureInit();
    const { out, status } = native.g1Mul(this.aff, k);
    return status[0] ? PointG1.ZERO : new PointG1(out);
  }
  multiplyUnsafe(scalar) { return this.multiply(scalar); }
  multiplyPrecomputed(scalar) { return this.multiply(scalar); }
  static fromPrivateKey(privateKey) { return PointG1.BASE.multiply(normalizePrivKey(privateKey)); }
  static sum(points) {
    ensureInit();
    const nz = points.filter((p) => !p.zero);
    if (!nz.length) return PointG1.ZERO;
    const { out, status } = native.g1Sum(concat(...nz.map((p) => p.aff)));
    return status[0] === 1 ? PointG1.ZERO : new PointG1(out);
  }
  // sum_i [k_i]P_i on the GPU (bucket method); scalars: bigint | number, any non-negative value below 2^256.  Not in the reference
  // (its aggregatePublicKeys is the unweighted sum): the building block of random-linear-combination batch verification
  static msm(points, scalars) {
    ensureInit();
    if (points.length !== scalars.length) throw new Error('points / scalars length mismatch');
    const keep = points.map((p, i) => i).filter((i) => !points[i].zero);
    if (!keep.length) return PointG1.ZERO;
    const { out, status } = native.g1Msm(concat(...keep.map((i) => points[i].aff)), concat(...keep.map((i) => hexToBytes(BigInt(scalars[i]).toString(16).padStart(64, '0')))));
    return status[0] === 1 ? PointG1.ZERO : new PointG1(out);
  }
  equals(rhs) { return this.zero === rhs.zero && (this.zero || bytesToHex(this.aff) === bytesToHex(rhs.aff)); }
  // reference index.ts:359-381
  toHex(isCompressed = false) {
    this.assertValidity();
    if (isCompressed) {
      if (this.zero) return 'c' + '0'.repeat(95);
      const x = toBig(this.aff.subarray(0, 48)), y = toBig(this.aff.subarray(48));
      const flag = (y * 2n) / CURVE.P;
      return (x + flag * (1n << 381n) + (1n << 383n)).toString(16).padStart(96, '0');
    }
    if (this.zero) return '4'.padEnd(192, '0');
    return bytesToHex(this.aff);
  }
  toRawBytes(isCompressed = false) { return hexToBytes(this.toHex(isCompressed)); }
  toAffine() { return [this.x, this.y]; }     // [Fp, Fp] (math.ts:949-958); the zero point gives (1, 1) scaled by an undefined inverse in the reference: not meaningful
}

class PointG2 {
  constructor(aff, zero = false, z) {      // aff = x.c0 || x.c1 || y.c0 || y.c1 (192 bytes), or the reference's (x: Fp2, y: Fp2, z?: Fp2) (index.ts:472)
    if (aff instanceof Fp2) {
      const x = aff, y = zero; z = z === undefined ? Fp2.ONE : z;
      if (!(y instanceof Fp2) || !(z instanceof Fp2)) throw new Error('Expected Fp2 coordinates');
      if (z.isZero()) { this.aff = new Uint8Array(192); this.zero = true; return; }
      const zi = z.invert();
      this.aff = concat(x.multiply(zi).toBytes(), y.multiply(zi).toBytes()); this.zero = false; return;
    }
    this.aff = aff; this.zero = zero;
  }
  get x() { return this.zero ? Fp2.ONE : Fp2.fromBytes(this.aff.subarray(0, 96)); }
  get y() { return this.zero ? Fp2.ONE : Fp2.fromBytes(this.aff.subarray(96)); }
  get z() { return this.zero ? Fp2.ZERO : Fp2.ONE; }
  getZero() { return PointG2.ZERO; }
  fromAffineTuple(xy) { return new PointG2(xy[0], xy[1], Fp2.ONE); }
  toAffine() { return [this.x, this.y]; }
  toAffineBatch(points) { return points.map((p) => p.toAffine()); }
  normalizeZ(points) { return points; }
  toString() { return this.zero ? 'Point<Zero>' : `Point<x=${this.x}, y=${this.y}>`; }
  calcMultiplyPrecomputes() {} clearMultiplyPrecomputes() {}      // caches of the reference's host path
  // reference index.ts:659-672 (clear_cofactor_bls12381_g2: [x^2 - x - 1]P + [x - 1]psi(P) + psi^2(2P)), for any point of the curve
  clearCofactor() {
    if (this.zero) return this;
    ensureInit();
    const { out, status } = native.clearCofactor(true, this.aff);
    return status[0] === 1 ? PointG2.ZERO : new PointG2(out);
  }
  // reference index.ts:695-711: the 68 line triples of calcPairingPrecomputes (math.ts:1331-1371), memoised on the point.  Computed on the GPU
  // (nbls_g2_prepare); the wire form is kept beside the Fp2 view so that PointG1.millerLoop can hand the table straight back to the engine.
  clearPairingPrecomputes() { this._PPRECOMPUTES = undefined; this._lineTable = undefined; }
  pairingPrecomputes() {
    if (this._PPRECOMPUTES) return this._PPRECOMPUTES;
    if (this.zero) throw new Error('No pairings at point of Infinity');
    ensureInit();
    const t = native.g2Prepare(this.aff);
    this._lineTable = t;
    const ell = [];
    for (let j = 0; j < 68; j++) ell.push([0, 1, 2].map((k) => Fp2.fromBytes(t.subarray(288 * j + 96 * k, 288 * j + 96 * k + 96))));
    this._PPRECOMPUTES = ell;
    return ell;
  }
  static get ZERO() { return new PointG2(new Uint8Array(192), true); }
  static get BASE() {
    return new PointG2(hexToBytes('024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8' +
      '13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e' +
      '0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801' +
      '0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be'));
  }
  isZero() { return this.zero; }
  // reference index.ts:481-490
  static async hashToCurve(msg, options) {
    msg = ensureBytes(msg); ensureInit();
    const dst = stringToBytes((options && options.DST) || htfDefaults.DST);
    return new PointG2(native.hashToG2(msg, Uint32Array.from([0, msg.length]), dst));
  }
  // reference index.ts:491-497
  static async encodeToCurve(msg, options) {
    msg = ensureBytes(msg); ensureInit();
    const dst = stringToBytes((options && options.DST) || htfDefaults.DST);
    return new PointG2(native.hashToCurve(2, msg, Uint32Array.from([0, msg.length]), dst));
  }
  // reference index.ts:500-530: 96 bytes, or 192 bytes read as two 96-byte integers z1 || z2
  static fromSignature(hex) {
    hex = ensureBytes(hex); ensureInit();
    const half = hex.length / 2;
    if (half !== 48 && half !== 96) throw new Error('Invalid compressed signature length, must be 96 or 192');
    const { out, status } = native.decodePoints(2, hex, hex.length);
    if (status[0] === 1) return PointG2.ZERO;
    if (status[0]) throw new Error(G2_STATUS[status[0]]);
    return new PointG2(out);
  }
  // reference index.ts:532-579: 96 compressed bytes (flag rules, the root named by the S bit, NO subgroup check) or 192 uncompressed bytes
  // x.c1 || x.c0 || y.c1 || y.c0 with the infinity flag 0x40, followed by assertValidity
  static fromHex(bytes) {
    bytes = ensureBytes(bytes);
    const m_byte = bytes[0] & 0xe0;
    if (m_byte === 0x20 || m_byte === 0x60 || m_byte === 0xe0) throw new Error('Invalid encoding flag: ' + m_byte);
    const bitC = m_byte & 0x80;
    if (!((bytes.length === 96 && bitC) || (bytes.length === 192 && !bitC))) throw new Error('Invalid point G2, expected 96/192 bytes');
    ensureInit();
    const { out, status } = native.decodePoints(1, bytes, bytes.length);
    if (status[0] === 1) return PointG2.ZERO;
    if (status[0]) throw new Error(bytes.length === 96 ? 'Invalid compressed G2 point' : G2_STATUS[status[0]]);
    return new PointG2(out);
  }
  assertValidity() {
    if (this.zero) return this;
    ensureInit();
    const st = native.g2Validate(this.aff).status[0];
    if (st) throw new Error(G2_STATUS[st]);
    return this;
  }
  add(rhs) { return PointG2.sum([this, rhs]); }
  negate() {
    if (this.zero) return this;
    const neg = (b) => { const v = toBig(b); return hexToBytes((v === 0n ? 0n : CURVE.P - v).toString(16).padStart(96, '0')); };
    return new PointG2(concat(this.aff.subarray(0, 96), neg(this.aff.subarray(96, 144)), neg(this.aff.subarray(144, 192))));
  }
  subtract(rhs) { return this.add(rhs.negate()); }
  double() { return this.add(this); }
  multiply(scalar) {
    const k = scalarBytes(scalar);
    if (this.zero) return PointG2.ZERO;
    ensureInit();
    const { out, status } = native.g2Mul(this.aff, k);
    return status[0] ? PointG2.ZERO : new PointG2(out);
  }
  multiplyUnsafe(scalar) { return this.multiply(scalar); }
  multiplyPrecomputed(scalar) { return this.multiply(scalar); }
  static fromPrivateKey(privateKey) { return PointG2.BASE.multiply(normalizePrivKey(privateKey)); }
  static sum(points) {
    ensureInit();
    const nz = points.filter((p) => !p.zero);
    if (!nz.length) return PointG2.ZERO;
    const { out, status } = native.g2Sum(concat(...nz.map((p) => p.aff)));
    return status[0] === 1 ? PointG2.ZERO : new PointG2(out);
  }
  static msm(points, scalars) {
    ensureInit();
    if (points.length !== scalars.length) throw new Error('points / scalars length mismatch');
    const keep = points.map((p, i) => i).filter((i) => !points[i].zero);
    if (!keep.length) return PointG2.ZERO;
    const { out, status } = native.g2Msm(concat(...keep.map((i) => points[i].aff)), concat(...keep.map((i) => hexToBytes(BigInt(scalars[i]).toString(16).padStart(64, '0')))));
    return status[0] === 1 ? PointG2.ZERO : new PointG2(out);
  }
  equals(rhs) { return this.zero === rhs.zero && (this.zero || bytesToHex(this.aff) === bytesToHex(rhs.aff)); }
  // reference index.ts:586-598
  toSignature() {
    if (this.zero) return hexToBytes('c' + '0'.repeat(191));
    const a = this.aff, x0 = a.subarray(0, 48), x1 = toBig(a.subarray(48, 96)), y0 = toBig(a.subarray(96, 144)), y1 = toBig(a.subarray(144, 192));
    const tmp = y1 > 0n ? y1 * 2n : y0 * 2n;
    const z1 = x1 + (tmp / CURVE.P) * (1n << 381n) + (1n << 383n);
    return concat(hexToBytes(z1.toString(16).padStart(96, '0')), x0);
  }
  toHex(isCompressed = false) {
    this.assertValidity();
    if (isCompressed) return bytesToHex(this.toSignature());
    if (this.zero) return '4'.padEnd(384, '0');
    const a = this.aff;
    return bytesToHex(concat(a.subarray(48, 96), a.subarray(0, 48), a.subarray(144, 192), a.subarray(96, 144)));
  }
  toRawBytes(isCompressed = false) { return hexToBytes(this.toHex(isCompressed)); }
}

// ---- reference index.ts:715-722
function pairing(P, Q, withFinalExponent = true) {
  if (P.isZero() || Q.isZero()) throw new Error('No pairings at point of Infinity');
  ensureInit();
  const { out, status } = native.pairingBatch(P.aff, Q.aff, withFinalExponent, true);
  if (status[0]) throw new Error(status[0] >= 10 ? G2_STATUS[status[0] - 10] : G1_STATUS[status[0]]);
  return Fp12.fromBytes(out);
}
// additive: n independent pairings in one call; points are PointG1[] / PointG2[] or packed affine byte arrays
function pairingBatch(Ps, Qs, withFinalExponent = true, validate = true) {
  ensureInit();
  const g1 = Ps instanceof Uint8Array ? Ps : concat(...Ps.map((p) => p.aff));
  const g2 = Qs instanceof Uint8Array ? Qs : concat(...Qs.map((q) => q.aff));
  return native.pairingBatch(g1, g2, withFinalExponent, validate);
}
function millerProduct(Ps, Qs, finalExp = true, validate = true) {
  ensureInit();
  const g1 = Ps instanceof Uint8Array ? Ps : concat(...Ps.map((p) => p.aff));
  const g2 = Qs instanceof Uint8Array ? Qs : concat(...Qs.map((q) => q.aff));
  return native.millerProduct(g1, g2, finalExp, validate);
}

function normP1(point) { return point instanceof PointG1 ? point : PointG1.fromHex(point); }
function normP2(point) { return point instanceof PointG2 ? point : PointG2.fromSignature(point); }
async function normP2Hash(point) { return point instanceof PointG2 ? point : PointG2.hashToCurve(point); }

// reference index.ts:269-279
function normalizePrivKey(key) {
  let int;
  if (key instanceof Uint8Array && key.length === 32) int = toBig(key);
  else if (typeof key === 'string' && key.length === 64) int = BigInt('0x' + key);
  else if (typeof key === 'number' && key > 0 && Number.isSafeInteger(key)) int = BigInt(key);
  else if (typeof key === 'bigint' && key > 0n) int = key;
  else throw new TypeError('Expected valid private key');
  int = ((int % CURVE.r) + CURVE.r) % CURVE.r;
  if (!(0n < int && int < CURVE.r)) throw new Error('Private key must be 0 < key < CURVE.r');
  return int;
}
const keyBytes = (k) => hexToBytes(normalizePrivKey(k).toString(16).padStart(64, '0'));
// reference index.ts:738-740: PointG1.fromPrivateKey(privateKey).toRawBytes(true)
function getPublicKeys(privateKeys) {
  ensureInit();
  const { out } = native.g1Mul(null, concat(...privateKeys.map(keyBytes)));
  return privateKeys.map((_, i) => new PointG1(out.slice(96 * i, 96 * i + 96)).toRawBytes(true));
}
function getPublicKey(privateKey) { return getPublicKeys([privateKey])[0]; }
// reference index.ts:744-752
async function sign(message, privateKey) {
  ensureInit();
  if (message instanceof PointG2) {
    message.assertValidity();
    const { out } = native.g2Mul(message.aff, keyBytes(privateKey));
    return new PointG2(out);
  }
  return (await signBatch([message], [privateKey]))[0];
}
async function signBatch(messages, privateKeys) {
  ensureInit();
  if (messages.length !== privateKeys.length) throw new Error('Expected equal number of messages and private keys');
  const msgs = messages.map(ensureBytes);
  const offs = new Uint32Array(msgs.length + 1);
  msgs.forEach((m, i) => { offs[i + 1] = offs[i] + m.length; });
  const { out } = await native.signBatchAsync(concat(...msgs), offs, stringToBytes(htfDefaults.DST), concat(...privateKeys.map(keyBytes)));   // worker thread: the event loop keeps running
  return msgs.map((_, i) => new PointG2(out.slice(192 * i, 192 * i + 192)).toSignature());
}

// reference index.ts:756-767: e(-P, H(m)) * e(G, S) == 1 with one final exponentiation.
// Wire-format inputs (96-byte signature, 48-byte key, message bytes: what a caller of the reference normally passes) take ONE engine call on a worker thread
// (nbls_verify_batch with n = 1: key decode + subgroup check, hash-to-G2 and signature decode overlap on three streams, then two Miller loops and one final
// exponentiation) and leave the event loop free; anything else -- point objects, other encodings, and every input the fast path rejects -- goes through the
// step-by-step form below, which raises the reference's exceptions.
async function verify(signature, message, publicKey) {
  ensureInit();
  if (!(signature instanceof PointG2) && !(message instanceof PointG2) && !(publicKey instanceof PointG1)) {
    let sig, msg, pk;
    try { sig = ensureBytes(signature); msg = ensureBytes(message); pk = ensureBytes(publicKey); } catch (e) { sig = null; }
    // the infinity flag (bit 6 of the first byte) is left to the slow path: the reference throws 'No pairings at point of Infinity' where verifyBatch's engine call answers false
    if (sig && sig.length === 96 && pk.length === 48 && !(sig[0] & 0x40) && !(pk[0] & 0x40)) {
      const r = await native.verifyBatchAsync(sig, msg, Uint32Array.of(0, msg.length), pk, stringToBytes(htfDefaults.DST));
      if (!r.code) return r.ok;
    }
  }
  const P = normP1(publicKey);
  const Hm = await normP2Hash(message);
  const S = normP2(signature);
  if (P.isZero() || Hm.isZero() || S.isZero()) throw new Error('No pairings at point of Infinity');
  const r = native.millerProduct(concat(P.negate().aff, PointG1.BASE.aff), concat(Hm.aff, S.aff), true, true);
  if (r.code) { const st = r.status[0] || r.status[1]; throw new Error(st >= 10 ? G2_STATUS[st - 10] : G1_STATUS[st]); }
  return Fp12.fromBytes(r.out).equals(Fp12.ONE);
}
// reference index.ts:771-788
function aggregatePublicKeys(publicKeys) {
  if (!publicKeys.length) throw new Error('Expected non-empty array');
  const agg = PointG1.sum(publicKeys.map(normP1));
  if (publicKeys[0] instanceof PointG1) return agg.assertValidity();
  return agg.toRawBytes(true);
}
function aggregateSignatures(signatures) {
  if (!signatures.length) throw new Error('Expected non-empty array');
  const agg = PointG2.sum(signatures.map(normP2));
  if (signatures[0] instanceof PointG2) return agg.assertValidity();
  return agg.toSignature();
}
// reference index.ts:792-821
async function verifyBatch(signature, messages, publicKeys) {
  if (!messages.length) throw new Error('Expected non-empty messages array');
  if (publicKeys.length !== messages.length) throw new Error('Pubkey count should equal msg count');
  ensureInit();
  const allWire = !(signature instanceof PointG2) && messages.every((m) => !(m instanceof PointG2)) && publicKeys.every((k) => !(k instanceof PointG1)) &&
    publicKeys.every((k) => ensureBytes(k).length === 48) && ensureBytes(signature).length === 96;
  if (allWire) {
    // fast path: one engine call (hex inputs hash to distinct message objects in the reference, so no grouping applies)
    const msgs = messages.map(ensureBytes);
    const offs = new Uint32Array(msgs.length + 1); msgs.forEach((m, i) => { offs[i + 1] = offs[i] + m.length; });
    const r = await native.verifyBatchAsync(ensureBytes(signature), concat(...msgs), offs, concat(...publicKeys.map(ensureBytes)), stringToBytes(htfDefaults.DST));   // worker thread: the event loop keeps running
    if (r.code) { normP2(signature); publicKeys.forEach(normP1); throw new Error('invalid point'); }   // re-derive the reference's exception
    return r.ok;
  }
  const sig = normP2(signature);
  const nMessages = await Promise.all(messages.map(normP2Hash));
  const nPublicKeys = publicKeys.map(normP1);
  try {
    const g1 = [], g2 = [];
    for (const message of new Set(nMessages)) {            // group keys per identical message OBJECT (reference index.ts:804-809)
      const group = PointG1.sum(nMessages.map((m, i) => (m === message ? nPublicKeys[i] : null)).filter((p) => p));
      if (group.isZero() || message.isZero()) throw new Error('No pairings at point of Infinity');
      g1.push(group); g2.push(message);
    }
    if (sig.isZero()) throw new Error('No pairings at point of Infinity');
    g1.push(PointG1.BASE.negate()); g2.push(sig);
    const r = native.millerProduct(concat(...g1.map((p) => p.aff)), concat(...g2.map((p) => p.aff)), true, true);
    if (r.code) throw new Error('invalid point');
    return Fp12.fromBytes(r.out).equals(Fp12.ONE);
  } catch (e) {
    return false;
  }
}

// ---- utils (reference index.ts:94-149): host-side helpers around the signature API
const nodeCrypto = require('crypto');
const sha256 = async (message) => Uint8Array.from(nodeCrypto.createHash('sha256').update(message).digest());
const modBig = (a, b) => { const r = a % b; return r >= 0n ? r : b + r; };
// expand_message_xmd of RFC 9380 section 5.3.1 (reference index.ts:207-231); H is an async digest with 32-byte output and 64-byte blocks
async function expandMessageXMD(msg, DST, lenInBytes, H = sha256) {
  if (DST.length > 255) DST = await H(concat(stringToBytes('H2C-OVERSIZE-DST-'), DST));
  const bInBytes = 32, rInBytes = 64, ell = Math.ceil(lenInBytes / bInBytes);
  if (ell > 255) throw new Error('Invalid xmd length');
  const dstPrime = concat(DST, Uint8Array.of(DST.length));
  const lib = Uint8Array.of((lenInBytes >> 8) & 0xff, lenInBytes & 0xff);
  const b0 = await H(concat(new Uint8Array(rInBytes), msg, lib, Uint8Array.of(0), dstPrime));
  const blocks = [await H(concat(b0, Uint8Array.of(1), dstPrime))];
  for (let i = 2; i <= ell; i++) blocks.push(await H(concat(b0.map((v, k) => v ^ blocks[i - 2][k]), Uint8Array.of(i), dstPrime)));
  return concat(...blocks).slice(0, lenInBytes);
}
// hash_to_field (reference index.ts:236-267): count elements of F_{p^m}, each coordinate from L = ceil((log2 p + k) / 8) uniform bytes
async function hashToField(msg, count, options = {}) {
  const o = { DST: htfDefaults.DST, p: CURVE.P, m: 2, k: 128, expand: true, hash: sha256, ...options };
  const L = Math.ceil((o.p.toString(2).length + o.k) / 8), len = count * o.m * L;
  const bytes = o.expand ? await expandMessageXMD(msg, stringToBytes(o.DST), len, o.hash) : msg;
  const u = [];
  for (let i = 0; i < count; i++) {
    const e = [];
    for (let j = 0; j < o.m; j++) { const off = L * (j + i * o.m); e.push(modBig(toBig(bytes.subarray(off, off + L)), o.p)); }
    u.push(e);
  }
  return u;
}
const utils = {
  hashToField, expandMessageXMD, sha256, mod: modBig,
  hashToPrivateKey: (hash) => {          // FIPS 186 B.1.1: 40..1024 uniform bytes -> key (reference index.ts:105-113)
    hash = ensureBytes(hash);
    if (hash.length < 40 || hash.length > 1024) throw new Error('Expected 40-1024 bytes of private key as per FIPS 186');
    const num = modBig(toBig(hash), CURVE.r);
    if (num === 0n || num === 1n) throw new Error('Invalid private key');
    return hexToBytes(num.toString(16).padStart(64, '0'));
  },
  randomBytes: (bytesLength = 32) => Uint8Array.from(nodeCrypto.randomBytes(bytesLength)),
  randomPrivateKey: () => utils.hashToPrivateKey(utils.randomBytes(40)),
  bytesToHex, hexToBytes, stringToBytes,
  getDSTLabel() { return htfDefaults.DST; },
  setDSTLabel(newLabel) {
    if (typeof newLabel !== 'string' || newLabel.length > 2048 || newLabel.length === 0) throw new TypeError('Invalid DST');
    htfDefaults.DST = newLabel;
  },
};

module.exports = { CURVE, Fp, Fr, Fp2, Fp6, Fp12, PointG1, PointG2, pairing, pairingBatch, millerProduct, getPublicKey, getPublicKeys, sign, signBatch, verify, verifyBatch,
  aggregatePublicKeys, aggregateSignatures, utils, init: (dev, contexts) => { if (contexts === undefined) native.init(dev || 0); else native.init(dev || 0, contexts); inited = true; } };

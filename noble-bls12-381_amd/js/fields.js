/*
 * fields.js -- the field classes the reference re-exports next to its signature API (index.ts:22: Fp, Fr, Fp2; math.ts:215-550).
 * They are single-element bigint helpers outside the batched hot path, so they stay on the host: plain BigInt arithmetic,
 * written for this facade (same method names, argument conventions and results as the reference's classes).
 */
'use strict';

const P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn;
const R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n;

const modp = (a, m) => { const r = a % m; return r >= 0n ? r : r + m; };
function powm(base, e, m) {
  if (e < 0n) throw new Error('Expected power > 0');
  let r = 1n, b = modp(base, m);
  for (; e > 0n; e >>= 1n) { if (e & 1n) r = (r * b) % m; b = (b * b) % m; }
  return r;
}
// modular inverse by the extended Euclidean algorithm; rejects 0 like the reference's invert (math.ts:134-156)
function invm(a, m) {
  if (a === 0n || m <= 0n) throw new Error(`invert: expected positive integers, got n=${a} mod=${m}`);
  let r0 = modp(a, m), r1 = m, s0 = 1n, s1 = 0n;
  while (r0 !== 0n) { const q = r1 / r0; [r1, r0] = [r0, r1 - q * r0]; [s1, s0] = [s0, s1 - q * s0]; }
  if (r1 !== 1n) throw new Error('invert: does not exist');
  return modp(s1, m);
}
const bitLen = (n) => n.toString(2).length;
function beBytes(v, len) { const h = v.toString(16).padStart(2 * len, '0'); const o = new Uint8Array(len); for (let i = 0; i < len; i++) o[i] = parseInt(h.slice(2 * i, 2 * i + 2), 16); return o; }
function beToBig(b) { let v = 0n; for (const x of b) v = (v << 8n) | BigInt(x); return v; }

// prime field template
function primeField(name, ORDER) {
  const F = class {
    constructor(value) { this.value = modp(BigInt(value), ORDER); }
    isZero() { return this.value === 0n; }
    equals(rhs) { return this.value === rhs.value; }
    negate() { return new F(-this.value); }
    invert() { return new F(invm(this.value, ORDER)); }
    add(rhs) { return new F(this.value + rhs.value); }
    subtract(rhs) { return new F(this.value - rhs.value); }
    square() { return new F(this.value * this.value); }
    multiply(rhs) { return new F(this.value * (rhs instanceof F ? rhs.value : BigInt(rhs))); }
    div(rhs) { return this.multiply(typeof rhs === 'bigint' ? new F(invm(rhs, ORDER)) : rhs.invert()); }
    pow(n) { return new F(powm(this.value, BigInt(n), ORDER)); }
    toString() { const s = this.value.toString(16).padStart(96, '0'); return s.slice(0, 2) + '.' + s.slice(-2); }
  };
  Object.defineProperty(F, 'name', { value: name });
  F.ORDER = ORDER;
  F.ZERO = new F(0n);
  F.ONE = new F(1n);
  return F;
}

const Fp = primeField('Fp', P);
Fp.MAX_BITS = bitLen(P);
Fp.BYTES_LEN = Math.ceil(Fp.MAX_BITS / 8);
// p = 3 mod 4: the candidate root is a^((p+1)/4); undefined when a is not a square (math.ts:260-264)
Fp.prototype.sqrt = function () { const r = this.pow((P + 1n) / 4n); return r.square().equals(this) ? r : undefined; };
Fp.fromBytes = function (b) { if (b.length !== Fp.BYTES_LEN) throw new Error(`fromBytes wrong length=${b.length}`); return new Fp(beToBig(b)); };
Fp.prototype.toBytes = function () { return beBytes(this.value, Fp.BYTES_LEN); };

const Fr = primeField('Fr', R);
Fr.isValid = (b) => b <= R;
Fr.prototype.toString = function () { return '0x' + this.value.toString(16).padStart(64, '0'); };
Fr.prototype.legendre = function () { return this.pow((R - 1n) / 2n); };
// Tonelli-Shanks with the smallest quadratic non-residue as generator of the 2-Sylow subgroup (r - 1 = 2^32 q)
Fr.prototype.sqrt = function () {
  if (!this.legendre().equals(Fr.ONE)) return undefined;
  let q = R - 1n, s = 0;
  while ((q & 1n) === 0n) { q >>= 1n; s++; }
  let z = 2n;
  while (powm(z, (R - 1n) / 2n, R) !== R - 1n) z++;
  let c = powm(z, q, R), x = powm(this.value, (q + 1n) / 2n, R), t = powm(this.value, q, R), m = s;
  while (t !== 1n) {
    let i = 0, tt = t;
    while (tt !== 1n) { tt = (tt * tt) % R; i++; }
    const b = powm(c, 1n << BigInt(m - i - 1), R);
    x = (x * b) % R; c = (b * b) % R; t = (t * c) % R; m = i;
  }
  return new Fr(x);
};

// Fp2 = Fp[u] / (u^2 + 1)
class Fp2 {
  constructor(c0, c1) {
    if (typeof c0 === 'bigint') throw new Error('c0: Expected Fp');
    if (typeof c1 === 'bigint') throw new Error('c1: Expected Fp');
    this.c0 = c0; this.c1 = c1;
  }
  static fromBigTuple(t) { return new Fp2(new Fp(t[0]), new Fp(t[1])); }
  one() { return Fp2.ONE; }
  isZero() { return this.c0.isZero() && this.c1.isZero(); }
  toString() { return `Fp2(${this.c0} + ${this.c1}×i)`; }
  reim() { return { re: this.c0.value, im: this.c1.value }; }
  negate() { return new Fp2(this.c0.negate(), this.c1.negate()); }
  equals(rhs) { return this.c0.equals(rhs.c0) && this.c1.equals(rhs.c1); }
  add(rhs) { return new Fp2(this.c0.add(rhs.c0), this.c1.add(rhs.c1)); }
  subtract(rhs) { return new Fp2(this.c0.subtract(rhs.c0), this.c1.subtract(rhs.c1)); }
  multiply(rhs) {
    if (typeof rhs === 'bigint') return new Fp2(this.c0.multiply(rhs), this.c1.multiply(rhs));
    const a0 = this.c0.value, a1 = this.c1.value, b0 = rhs.c0.value, b1 = rhs.c1.value;
    return new Fp2(new Fp(a0 * b0 - a1 * b1), new Fp(a0 * b1 + a1 * b0));
  }
  square() { const a = this.c0.value, b = this.c1.value; return new Fp2(new Fp((a + b) * (a - b)), new Fp(2n * a * b)); }
  pow(n) {
    n = BigInt(n);
    if (n < 0n) throw new Error('Expected power > 0');
    let r = Fp2.ONE, b = this;
    for (; n > 0n; n >>= 1n) { if (n & 1n) r = r.multiply(b); b = b.square(); }
    return r;
  }
  invert() { const n = new Fp(this.c0.value * this.c0.value + this.c1.value * this.c1.value).invert(); return new Fp2(this.c0.multiply(n), this.c1.negate().multiply(n)); }
  div(rhs) { return this.multiply(typeof rhs === 'bigint' ? new Fp(rhs).invert().value : rhs.invert()); }
  mulByNonresidue() { return new Fp2(this.c0.subtract(this.c1), this.c0.add(this.c1)); }      // * (1 + u)
  multiplyByB() { const t0 = this.c0.multiply(4n), t1 = this.c1.multiply(4n); return new Fp2(t0.subtract(t1), t0.add(t1)); }   // * 4(1 + u)
  frobeniusMap(power) { return (power & 1) ? new Fp2(this.c0, this.c1.negate()) : this; }       // x^(p^k) = conj^k(x)
  // a square root, or undefined.  Norm method: |x|^2 = sqrt(N(a)); of the two roots +-x the reference returns the one with the
  // larger imaginary part (real part on ties), math.ts:486-507 -- the same choice is made here.
  sqrt() {
    if (this.isZero()) throw new Error('invert: expected positive integers, got n=0 mod=' + P);   // the reference divides by the argument
    const a0 = this.c0, a1 = this.c1;
    let x;
    if (a1.isZero()) {
      const r = a0.sqrt();
      if (r) x = new Fp2(r, Fp.ZERO);
      else { const i = a0.negate().sqrt(); if (!i) return undefined; x = new Fp2(Fp.ZERO, i); }      // u^2 = -1
    } else {
      const n = a0.square().add(a1.square()).sqrt();
      if (!n) return undefined;
      const half = new Fp(2n).invert();
      let t = a0.add(n).multiply(half), x0 = t.sqrt();
      if (!x0) { t = a0.subtract(n).multiply(half); x0 = t.sqrt(); }
      if (!x0) return undefined;
      x = new Fp2(x0, a1.multiply(x0.multiply(2n).invert()));
    }
    if (!x.square().equals(this)) return undefined;
    const y = x.negate();
    return (x.c1.value > y.c1.value || (x.c1.value === y.c1.value && x.c0.value > y.c0.value)) ? x : y;
  }
  static fromBytes(b) { if (b.length !== Fp2.BYTES_LEN) throw new Error(`fromBytes wrong length=${b.length}`); return new Fp2(Fp.fromBytes(b.subarray(0, Fp.BYTES_LEN)), Fp.fromBytes(b.subarray(Fp.BYTES_LEN))); }
  toBytes() { const o = new Uint8Array(Fp2.BYTES_LEN); o.set(this.c0.toBytes(), 0); o.set(this.c1.toBytes(), Fp.BYTES_LEN); return o; }
}
Fp2.ORDER = P * P - 1n;          // CURVE.P2 of the reference (math.ts:29-32, 404)
Fp2.MAX_BITS = bitLen(Fp2.ORDER);
Fp2.BYTES_LEN = 2 * Fp.BYTES_LEN;     // two 48-byte field elements (math.ts:540-549)
Fp2.ZERO = new Fp2(Fp.ZERO, Fp.ZERO);
Fp2.ONE = new Fp2(Fp.ONE, Fp.ZERO);

// ---- Fp6 = Fp2[v] / (v^3 - (1 + u)),  Fp12 = Fp6[w] / (w^2 - v): single-element host arithmetic for the results the engine hands back
// (products / inverses of a few pairing values; anything batched goes to the GPU).  Coordinates as in the reference's classes
// (math.ts:554-885): Fp6(c0, c1, c2), Fp12(c0, c1); bytes = coordinates in that order, 48 bytes each.
const XI = new Fp2(Fp.ONE, Fp.ONE);
class Fp6 {
  constructor(c0, c1, c2) { this.c0 = c0; this.c1 = c1; this.c2 = c2; }
  static fromBigSix(t) { return new Fp6(Fp2.fromBigTuple(t.slice(0, 2)), Fp2.fromBigTuple(t.slice(2, 4)), Fp2.fromBigTuple(t.slice(4, 6))); }
  one() { return Fp6.ONE; }
  isZero() { return this.c0.isZero() && this.c1.isZero() && this.c2.isZero(); }
  equals(r) { return this.c0.equals(r.c0) && this.c1.equals(r.c1) && this.c2.equals(r.c2); }
  negate() { return new Fp6(this.c0.negate(), this.c1.negate(), this.c2.negate()); }
  add(r) { return new Fp6(this.c0.add(r.c0), this.c1.add(r.c1), this.c2.add(r.c2)); }
  subtract(r) { return new Fp6(this.c0.subtract(r.c0), this.c1.subtract(r.c1), this.c2.subtract(r.c2)); }
  toString() { return `Fp6(${this.c0} + ${this.c1} * v, ${this.c2} * v^2)`; }
  multiply(r) {
    if (typeof r === 'bigint') return new Fp6(this.c0.multiply(r), this.c1.multiply(r), this.c2.multiply(r));
    const { c0: a0, c1: a1, c2: a2 } = this, { c0: b0, c1: b1, c2: b2 } = r;      // schoolbook, v^3 = xi
    return new Fp6(a0.multiply(b0).add(a1.multiply(b2).add(a2.multiply(b1)).mulByNonresidue()),
      a0.multiply(b1).add(a1.multiply(b0)).add(a2.multiply(b2).mulByNonresidue()),
      a0.multiply(b2).add(a1.multiply(b1)).add(a2.multiply(b0)));
  }
  square() { return this.multiply(this); }
  mulByNonresidue() { return new Fp6(this.c2.mulByNonresidue(), this.c0, this.c1); }     // * v
  pow(n) { return powGeneric(this, Fp6.ONE, n); }
  invert() {
    const { c0, c1, c2 } = this;
    const t0 = c0.square().subtract(c2.multiply(c1).mulByNonresidue()), t1 = c2.square().mulByNonresidue().subtract(c0.multiply(c1)), t2 = c1.square().subtract(c0.multiply(c2));
    const d = c2.multiply(t1).add(c1.multiply(t2)).mulByNonresidue().add(c0.multiply(t0)).invert();
    return new Fp6(d.multiply(t0), d.multiply(t1), d.multiply(t2));
  }
  div(r) { return this.multiply(typeof r === 'bigint' ? new Fp(r).invert().value : r.invert()); }
  frobeniusMap(power) { return new Fp6(this.c0.frobeniusMap(power), this.c1.frobeniusMap(power).multiply(FROB6_1[power % 6]), this.c2.frobeniusMap(power).multiply(FROB6_2[power % 6])); }
  static fromBytes(b) { if (b.length !== Fp6.BYTES_LEN) throw new Error(`fromBytes wrong length=${b.length}`); return new Fp6(Fp2.fromBytes(b.subarray(0, 96)), Fp2.fromBytes(b.subarray(96, 192)), Fp2.fromBytes(b.subarray(192, 288))); }
  toBytes() { const o = new Uint8Array(288); o.set(this.c0.toBytes(), 0); o.set(this.c1.toBytes(), 96); o.set(this.c2.toBytes(), 192); return o; }
}
function powGeneric(base, one, n) {
  n = BigInt(n);
  if (n < 0n) throw new Error('Expected power > 0');
  let r = one, b = base;
  for (; n > 0n; n >>= 1n) { if (n & 1n) r = r.multiply(b); b = b.square(); }
  return r;
}
Fp6.BYTES_LEN = 3 * Fp2.BYTES_LEN;
Fp6.ZERO = new Fp6(Fp2.ZERO, Fp2.ZERO, Fp2.ZERO);
Fp6.ONE = new Fp6(Fp2.ONE, Fp2.ZERO, Fp2.ZERO);
// Frobenius constants: xi^((p^k - 1) / 3), xi^(2 (p^k - 1) / 3), xi^((p^k - 1) / 6)
const FROB6_1 = [], FROB6_2 = [], FROB12 = [];
for (let k = 0; k < 12; k++) { const e = P ** BigInt(k) - 1n; if (k < 6) { FROB6_1.push(XI.pow(e / 3n)); FROB6_2.push(XI.pow(2n * e / 3n)); } FROB12.push(XI.pow(e / 6n)); }

class Fp12 {
  constructor(c0, c1) { this.c0 = c0; this.c1 = c1; }
  static fromBigTwelve(t) { return new Fp12(Fp6.fromBigSix(t.slice(0, 6)), Fp6.fromBigSix(t.slice(6, 12))); }
  one() { return Fp12.ONE; }
  isZero() { return this.c0.isZero() && this.c1.isZero(); }
  equals(r) { return this.c0.equals(r.c0) && this.c1.equals(r.c1); }
  negate() { return new Fp12(this.c0.negate(), this.c1.negate()); }
  add(r) { return new Fp12(this.c0.add(r.c0), this.c1.add(r.c1)); }
  subtract(r) { return new Fp12(this.c0.subtract(r.c0), this.c1.subtract(r.c1)); }
  toString() { return `Fp12(${this.c0} + ${this.c1} * w)`; }
  multiply(r) {
    if (typeof r === 'bigint') return new Fp12(this.c0.multiply(r), this.c1.multiply(r));
    const t0 = this.c0.multiply(r.c0), t1 = this.c1.multiply(r.c1);                // w^2 = v
    return new Fp12(t0.add(t1.mulByNonresidue()), this.c0.add(this.c1).multiply(r.c0.add(r.c1)).subtract(t0.add(t1)));
  }
  square() { return this.multiply(this); }
  pow(n) { return powGeneric(this, Fp12.ONE, n); }
  invert() { const t = this.c0.square().subtract(this.c1.square().mulByNonresidue()).invert(); return new Fp12(this.c0.multiply(t), this.c1.multiply(t).negate()); }
  div(r) { return this.multiply(typeof r === 'bigint' ? new Fp(r).invert().value : r.invert()); }
  conjugate() { return new Fp12(this.c0, this.c1.negate()); }
  frobeniusMap(power) {
    const r0 = this.c0.frobeniusMap(power), { c0, c1, c2 } = this.c1.frobeniusMap(power), k = FROB12[power % 12];
    return new Fp12(r0, new Fp6(c0.multiply(k), c1.multiply(k), c2.multiply(k)));
  }
  static fromBytes(b) { if (b.length !== Fp12.BYTES_LEN) throw new Error(`fromBytes wrong length=${b.length}`); return new Fp12(Fp6.fromBytes(b.subarray(0, 288)), Fp6.fromBytes(b.subarray(288, 576))); }
  toBytes() { const o = new Uint8Array(576); o.set(this.c0.toBytes(), 0); o.set(this.c1.toBytes(), 288); return o; }
  // kilic/bls12-381 (and zkcrypto) write the twelve 48-byte coefficients in the opposite order: highest tower coefficient first, c1 before c0 in
  // every Fp2.  The reference's tests convert by reversing the 48-byte chunks (test/deterministic.test.ts:9-12, 41); this is that adaptor
  // (SURVEY 8(f).4).  Accepts / returns 576 bytes; a 1152-character hex string is accepted too.
  static fromKilicBytes(b) {
    if (typeof b === 'string') { if (b.length !== 1152) throw new Error(`fromKilicBytes wrong length=${b.length / 2}`); const u = new Uint8Array(576); for (let i = 0; i < 576; i++) u[i] = Number.parseInt(b.slice(2 * i, 2 * i + 2), 16); b = u; }
    if (b.length !== Fp12.BYTES_LEN) throw new Error(`fromKilicBytes wrong length=${b.length}`);
    const o = new Uint8Array(576);
    for (let k = 0; k < 12; k++) o.set(b.subarray(48 * (11 - k), 48 * (12 - k)), 48 * k);
    return Fp12.fromBytes(o);
  }
  toKilicBytes() { const b = this.toBytes(), o = new Uint8Array(576); for (let k = 0; k < 12; k++) o.set(b.subarray(48 * (11 - k), 48 * (12 - k)), 48 * k); return o; }
}
Fp12.BYTES_LEN = 2 * Fp6.BYTES_LEN;
Fp12.ZERO = new Fp12(Fp6.ZERO, Fp6.ZERO);
Fp12.ONE = new Fp12(Fp6.ONE, Fp6.ZERO);

module.exports = { Fp, Fr, Fp2, Fp6, Fp12 };

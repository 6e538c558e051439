// Type declarations of the facade (index.js): the exported surface of paulmillr/noble-bls12-381 v1.4.0 (index.ts:22, 715-821)
// plus the additive batched entry points of the MI355X engine.
export type Hex = Uint8Array | string;
export type PrivateKey = Hex | bigint | number;

export declare const CURVE: {
  P: bigint; r: bigint; h: bigint; Gx: bigint; Gy: bigint; b: bigint; P2: bigint; h2: bigint;
  G2x: [bigint, bigint]; G2y: [bigint, bigint]; b2: [bigint, bigint]; x: bigint; h2Eff: bigint;
};

interface Field<T> {
  isZero(): boolean; equals(rhs: T): boolean; negate(): T; add(rhs: T): T; subtract(rhs: T): T; invert(): T;
  multiply(rhs: T | bigint): T; square(): T; pow(n: bigint): T; div(rhs: T | bigint): T;
}
export declare class Fp implements Field<Fp> {
  static readonly ORDER: bigint; static readonly MAX_BITS: number; static readonly BYTES_LEN: number; static readonly ZERO: Fp; static readonly ONE: Fp;
  readonly value: bigint;
  constructor(value: bigint);
  isZero(): boolean; equals(rhs: Fp): boolean; negate(): Fp; invert(): Fp; add(rhs: Fp): Fp; subtract(rhs: Fp): Fp; square(): Fp;
  multiply(rhs: Fp | bigint): Fp; div(rhs: Fp | bigint): Fp; pow(n: bigint): Fp; sqrt(): Fp | undefined; toString(): string;
  static fromBytes(b: Uint8Array): Fp; toBytes(): Uint8Array;
}
export declare class Fr implements Field<Fr> {
  static readonly ORDER: bigint; static readonly ZERO: Fr; static readonly ONE: Fr; static isValid(b: bigint): boolean;
  readonly value: bigint;
  constructor(value: bigint);
  isZero(): boolean; equals(rhs: Fr): boolean; negate(): Fr; invert(): Fr; add(rhs: Fr): Fr; subtract(rhs: Fr): Fr; square(): Fr;
  multiply(rhs: Fr | bigint): Fr; div(rhs: Fr | bigint): Fr; pow(n: bigint): Fr; legendre(): Fr; sqrt(): Fr | undefined; toString(): string;
}
export declare class Fp2 implements Field<Fp2> {
  static readonly ORDER: bigint; static readonly MAX_BITS: number; static readonly BYTES_LEN: number; static readonly ZERO: Fp2; static readonly ONE: Fp2;
  readonly c0: Fp; readonly c1: Fp;
  constructor(c0: Fp, c1: Fp);
  static fromBigTuple(t: [bigint, bigint] | bigint[]): Fp2;
  one(): Fp2; isZero(): boolean; equals(rhs: Fp2): boolean; reim(): { re: bigint; im: bigint }; negate(): Fp2; add(rhs: Fp2): Fp2; subtract(rhs: Fp2): Fp2;
  multiply(rhs: Fp2 | bigint): Fp2; square(): Fp2; pow(n: bigint): Fp2; div(rhs: Fp2 | bigint): Fp2; invert(): Fp2; sqrt(): Fp2 | undefined;
  mulByNonresidue(): Fp2; multiplyByB(): Fp2; frobeniusMap(power: number): Fp2; toString(): string;
  static fromBytes(b: Uint8Array): Fp2; toBytes(): Uint8Array;
}
export declare class Fp6 implements Field<Fp6> {
  static readonly BYTES_LEN: number; static readonly ZERO: Fp6; static readonly ONE: Fp6;
  readonly c0: Fp2; readonly c1: Fp2; readonly c2: Fp2;
  constructor(c0: Fp2, c1: Fp2, c2: Fp2);
  static fromBigSix(t: bigint[]): Fp6;
  one(): Fp6; isZero(): boolean; equals(rhs: Fp6): boolean; negate(): Fp6; add(rhs: Fp6): Fp6; subtract(rhs: Fp6): Fp6; multiply(rhs: Fp6 | bigint): Fp6;
  square(): Fp6; pow(n: bigint): Fp6; div(rhs: Fp6 | bigint): Fp6; invert(): Fp6; mulByNonresidue(): Fp6; frobeniusMap(power: number): Fp6; toString(): string;
  static fromBytes(b: Uint8Array): Fp6; toBytes(): Uint8Array;
}
export declare class Fp12 implements Field<Fp12> {
  static readonly BYTES_LEN: number; static readonly ZERO: Fp12; static readonly ONE: Fp12;
  readonly c0: Fp6; readonly c1: Fp6;
  constructor(c0: Fp6, c1: Fp6);
  static fromBigTwelve(t: bigint[]): Fp12;
  one(): Fp12; isZero(): boolean; equals(rhs: Fp12): boolean; negate(): Fp12; add(rhs: Fp12): Fp12; subtract(rhs: Fp12): Fp12; multiply(rhs: Fp12 | bigint): Fp12;
  square(): Fp12; pow(n: bigint): Fp12; div(rhs: Fp12 | bigint): Fp12; invert(): Fp12; conjugate(): Fp12; frobeniusMap(power: number): Fp12; toString(): string;
  /** runs on the GPU (nbls_final_exp_batch) */
  finalExponentiate(): Fp12;
  static fromBytes(b: Uint8Array): Fp12; toBytes(): Uint8Array;
  /** kilic / zkcrypto coefficient order (the twelve 48-byte words reversed; reference test/deterministic.test.ts:9-12, 41) */
  static fromKilicBytes(b: Uint8Array | string): Fp12; toKilicBytes(): Uint8Array;
}

export declare class PointG1 {
  static readonly BASE: PointG1; static readonly ZERO: PointG1;
  constructor(x: Fp, y: Fp, z?: Fp);
  readonly x: Fp; readonly y: Fp; readonly z: Fp;
  static fromHex(bytes: Hex): PointG1;
  static fromPrivateKey(privateKey: PrivateKey): PointG1;
  static hashToCurve(msg: Hex, options?: { DST?: string }): Promise<PointG1>;
  static encodeToCurve(msg: Hex, options?: { DST?: string }): Promise<PointG1>;
  /** additive: sum of points / multi-scalar multiplication on the GPU */
  static sum(points: PointG1[]): PointG1;
  static msm(points: PointG1[], scalars: (bigint | number)[]): PointG1;
  isZero(): boolean; equals(rhs: PointG1): boolean; negate(): PointG1; add(rhs: PointG1): PointG1; subtract(rhs: PointG1): PointG1; double(): PointG1;
  multiply(scalar: bigint | number): PointG1; multiplyUnsafe(scalar: bigint | number): PointG1; multiplyPrecomputed(scalar: bigint | number): PointG1;
  assertValidity(): this; toAffine(): [Fp, Fp]; toRawBytes(isCompressed?: boolean): Uint8Array; toHex(isCompressed?: boolean): string;
  millerLoop(P: PointG2): Fp12; clearCofactor(): PointG1; toString(): string;
}
export declare class PointG2 {
  static readonly BASE: PointG2; static readonly ZERO: PointG2;
  constructor(x: Fp2, y: Fp2, z?: Fp2);
  readonly x: Fp2; readonly y: Fp2; readonly z: Fp2;
  static fromHex(bytes: Hex): PointG2;
  static fromSignature(hex: Hex): PointG2;
  static fromPrivateKey(privateKey: PrivateKey): PointG2;
  static hashToCurve(msg: Hex, options?: { DST?: string }): Promise<PointG2>;
  static encodeToCurve(msg: Hex, options?: { DST?: string }): Promise<PointG2>;
  static sum(points: PointG2[]): PointG2;
  static msm(points: PointG2[], scalars: (bigint | number)[]): PointG2;
  isZero(): boolean; equals(rhs: PointG2): boolean; negate(): PointG2; add(rhs: PointG2): PointG2; subtract(rhs: PointG2): PointG2; double(): PointG2;
  multiply(scalar: bigint | number): PointG2; multiplyUnsafe(scalar: bigint | number): PointG2; multiplyPrecomputed(scalar: bigint | number): PointG2;
  assertValidity(): this; toAffine(): [Fp2, Fp2]; toSignature(): Uint8Array; toRawBytes(isCompressed?: boolean): Uint8Array; toHex(isCompressed?: boolean): string;
  toString(): string;
}

export declare function pairing(P: PointG1, Q: PointG2, withFinalExponent?: boolean): Fp12;
export declare function getPublicKey(privateKey: PrivateKey): Uint8Array;
export declare function sign(message: Hex, privateKey: PrivateKey): Promise<Uint8Array>;
export declare function sign(message: PointG2, privateKey: PrivateKey): Promise<PointG2>;
export declare function verify(signature: Hex | PointG2, message: Hex | PointG2, publicKey: Hex | PointG1): Promise<boolean>;
export declare function aggregatePublicKeys(publicKeys: Hex[]): Uint8Array;
export declare function aggregatePublicKeys(publicKeys: PointG1[]): PointG1;
export declare function aggregateSignatures(signatures: Hex[]): Uint8Array;
export declare function aggregateSignatures(signatures: PointG2[]): PointG2;
export declare function verifyBatch(signature: Hex | PointG2, messages: (Hex | PointG2)[], publicKeys: (Hex | PointG1)[]): Promise<boolean>;

// additive batched entry points (one engine call each)
export declare function pairingBatch(Ps: PointG1[] | Uint8Array /* n x 96 affine bytes */, Qs: PointG2[] | Uint8Array /* n x 192 */, withFinalExponent?: boolean, validate?: boolean): { out: Uint8Array /* n x 576 */; status: Uint8Array };
export declare function millerProduct(Ps: PointG1[] | Uint8Array, Qs: PointG2[] | Uint8Array, finalExponent?: boolean, validate?: boolean): { out: Uint8Array /* 576 */; status: Uint8Array };
export declare function getPublicKeys(privateKeys: PrivateKey[]): Uint8Array[];
export declare function signBatch(messages: Hex[], privateKeys: PrivateKey[]): Promise<Uint8Array[]>;
/** contexts (1..8, default NBLS_CONTEXTS or 1): engine contexts that asynchronous verifyBatch calls take round-robin, so that concurrent promises overlap on the GPU */
export declare function init(deviceId?: number, contexts?: number): void;

export declare const utils: {
  hashToField(msg: Uint8Array, count: number, options?: { DST?: string; p?: bigint; m?: number; k?: number; expand?: boolean; hash?: (m: Uint8Array) => Promise<Uint8Array> }): Promise<bigint[][]>;
  expandMessageXMD(msg: Uint8Array, DST: Uint8Array, lenInBytes: number, H?: (m: Uint8Array) => Promise<Uint8Array>): Promise<Uint8Array>;
  hashToPrivateKey(hash: Hex): Uint8Array;
  stringToBytes(str: string): Uint8Array; bytesToHex(b: Uint8Array): string; hexToBytes(hex: string): Uint8Array;
  randomBytes(bytesLength?: number): Uint8Array; randomPrivateKey(): Uint8Array;
  sha256(message: Uint8Array): Promise<Uint8Array>; mod(a: bigint, b: bigint): bigint;
  getDSTLabel(): string; setDSTLabel(newLabel: string): void;
};

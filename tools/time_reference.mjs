// Times the REAL reference (type-stripped copy under /tmp, tools/strip_ts.py) on the same single pairing oracle/js_bigint_pairing.js times, under the
// same Node: the ratio of the two is recorded in BASELINE.md so that the figure bench.py measures on the GPU box (where the reference cannot travel)
// can be read as a reference figure.   node tools/time_reference.mjs /tmp/nbls_ref [seconds]
import { pathToFileURL } from 'url';
import path from 'path';
const refDir = process.argv[2] || '/tmp/nbls_ref', seconds = Number(process.argv[3] || 5);
async function main() {
  const bls = await import(pathToFileURL(path.join(refDir, 'index.mjs')).href);
  const P = bls.PointG1.BASE, Q = bls.PointG2.BASE;
  const ok = bls.pairing(P, Q).c0.c0.c0.value === 0x1250ebd871fc0a92a7b2d83168d0d727272d441befa15c503dd8e90ce98db3e7b6d194f60839c508a84305aaca1789b6n;
  let n = 0; const t0 = process.hrtime.bigint();
  while (Number(process.hrtime.bigint() - t0) / 1e9 < seconds) {
    Q.clearPairingPrecomputes();          // a fresh pairing every time (the reference memoises the line table on the point object)
    bls.pairing(P, Q); n++;
  }
  const dt = Number(process.hrtime.bigint() - t0) / 1e9;
  console.log(JSON.stringify({ what: 'reference pairing(G1, G2), fresh precomputes', pairings: n, seconds: Number(dt.toFixed(3)), pairings_per_s: Number((n / dt).toFixed(2)), node: process.version, ok }));
}
main().catch((e) => { console.error(e); process.exit(1); });

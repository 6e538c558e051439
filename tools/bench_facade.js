// Wall time of the facade's verify() and sign() from JavaScript (noble-bls12-381_amd/js/index.js over the N-API addon), one call at a time, wire-format inputs:
// what a user who switches from the reference gets for `await bls.verify(sig, msg, pk)` / `await bls.sign(msg, sk)`.  One JSON line; bench.py embeds it.
'use strict';
const fs = require('fs'), zlib = require('zlib'), path = require('path');
const bls = require(path.join(__dirname, '..', 'noble-bls12-381_amd', 'js', 'index.js'));
const gold = JSON.parse(zlib.gunzipSync(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'ref_vectors.json.gz'))).toString());
(async () => {
  const s = gold.sigs[0], hex = bls.utils.bytesToHex;
  for (let i = 0; i < 3; i++) if (!(await bls.verify(s.sig, s.msg, s.pk))) throw new Error('verify failed');
  let turns = 0, live = true;
  const spin = () => { if (live) { turns++; setImmediate(spin); } };
  setImmediate(spin);
  const N = 50;
  let t0 = process.hrtime.bigint();
  for (let i = 0; i < N; i++) await bls.verify(s.sig, s.msg, s.pk);
  const verifyMs = Number(process.hrtime.bigint() - t0) / 1e6 / N;
  if (hex(await bls.sign(s.msg, s.sk)) !== s.sig) throw new Error('sign differs from the reference signature');
  t0 = process.hrtime.bigint();
  for (let i = 0; i < 20; i++) await bls.sign(s.msg, s.sk);
  const signMs = Number(process.hrtime.bigint() - t0) / 1e6 / 20;
  live = false;
  console.log(JSON.stringify({ verify_ms: Number(verifyMs.toFixed(3)), sign_ms: Number(signMs.toFixed(3)), event_loop_turns_during_calls: turns, node: process.version,
    note: 'await bls.verify(sig, msg, pk) / await bls.sign(msg, sk) on 96 / 48 / 32-byte wire inputs, one call at a time, worker-thread N-API calls (nbls_verify_batch n = 1, nbls_sign_batch n = 1)' }));
})().catch((e) => { console.error(e); process.exit(1); });

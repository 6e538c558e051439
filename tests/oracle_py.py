"""ctypes binding of oracle/libnbls_oracle.so -- the CPU parity oracle (test infrastructure only)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, 'oracle')
DST_DEFAULT = b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_'   # reference index.ts:64


def build():
    subprocess.check_call(['make', '-s', '-C', ODIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def _out(self, n):
        return C.create_string_buffer(n)

    def call(self, name, outlen, *args):
        """fn(args..., out) -> (int status, out bytes)"""
        out = self._out(outlen)
        fn = getattr(self.lib, 'oracle_' + name)
        fn.restype = C.c_int
        r = fn(*args, out)
        return r, out.raw

    def bin(self, name, a, b, n):
        return self.call(name, n, a, b)[1]

    def un(self, name, a, n):
        return self.call(name, n, a)[1]

    def fp12_frob(self, a, k):
        return self.call('fp12_frob', 576, a, C.c_int(k))[1]

    def fp6_frob(self, a, k):
        return self.call('fp6_frob', 288, a, C.c_int(k))[1]

    def miller_loop(self, g1, g2):
        return self.call('miller_loop', 576, g1, g2)[1]

    def pairing(self, g1, g2, final_exp=True, validate=True):
        return self.call('pairing', 576, g1, g2, C.c_int(int(final_exp)), C.c_int(int(validate)))

    def pairing_batch(self, g1s, g2s, final_exp=True, validate=False, threads=1):
        n = len(g1s) // 96
        out = self._out(576 * n)
        st = self._out(n)
        self.lib.oracle_pairing_batch(C.c_size_t(n), g1s, g2s, C.c_int(int(final_exp)), C.c_int(int(validate)), out, st, C.c_int(threads))
        return out.raw, st.raw

    def decompress_batch(self, comp, g2=False, threads=8):
        """PointG1.fromHex (48 B) / PointG2.fromSignature (96 B) of every encoding -> (affine bytes, zeroed where the status is not 0; status bytes)"""
        a = 96 if g2 else 48
        n = len(comp) // a
        out = self._out(2 * a * n)
        st = self._out(n)
        self.lib.oracle_decompress_batch(C.c_size_t(n), C.c_int(int(g2)), comp, out, st, C.c_int(threads))
        return out.raw, st.raw

    def miller_product(self, g1s, g2s, final_exp=True):
        n = len(g1s) // 96
        out = self._out(576)
        self.lib.oracle_miller_product(C.c_size_t(n), g1s, g2s, C.c_int(int(final_exp)), out)
        return out.raw

    def g1_mul(self, aff, k):
        return self.call('g1_mul', 96, aff, k.to_bytes(32, 'big'), C.c_size_t(32))

    def g2_mul(self, aff, k):
        return self.call('g2_mul', 192, aff, k.to_bytes(32, 'big'), C.c_size_t(32))

    def g1_generator(self):
        out = self._out(96)
        self.lib.oracle_g1_generator(out)
        return out.raw

    def g2_generator(self):
        out = self._out(192)
        self.lib.oracle_g2_generator(out)
        return out.raw

    def g1_sum(self, pts):
        return self.call('g1_sum', 96, C.c_size_t(len(pts) // 96), pts)

    def g2_sum(self, pts):
        return self.call('g2_sum', 192, C.c_size_t(len(pts) // 192), pts)

    def hash_to_g2(self, msg, dst=DST_DEFAULT):
        return self.call('hash_to_g2', 192, msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)))

    def encode_to_g2(self, msg, dst=DST_DEFAULT):
        return self.call('encode_to_g2', 192, msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)))

    def hash_to_g1(self, msg, dst=DST_DEFAULT):
        return self.call('hash_to_g1', 96, msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)))

    def encode_to_g1(self, msg, dst=DST_DEFAULT):
        return self.call('encode_to_g1', 96, msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)))

    def hash_to_field(self, msg, dst=DST_DEFAULT):
        return self.call('hash_to_field', 192, msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)))

    def expand_message_xmd(self, msg, dst, n):
        out = self._out(n)
        self.lib.oracle_expand_message_xmd(msg, C.c_size_t(len(msg)), dst, C.c_size_t(len(dst)), out, C.c_size_t(n))
        return out.raw

    def sign(self, msg, sk32, dst=DST_DEFAULT):
        return self.call('sign', 96, msg, C.c_size_t(len(msg)), sk32, dst, C.c_size_t(len(dst)))

    def get_public_key(self, sk32):
        out = self._out(48)
        self.lib.oracle_get_public_key(sk32, out)
        return out.raw

    def verify(self, sig, msg, pk, dst=DST_DEFAULT):
        self.lib.oracle_verify.restype = C.c_int
        return self.lib.oracle_verify(sig, msg, C.c_size_t(len(msg)), pk, dst, C.c_size_t(len(dst)))

    def verify_batch(self, sig, msgs, pks, dst=DST_DEFAULT):
        offs = [0]
        for m in msgs:
            offs.append(offs[-1] + len(m))
        arr = (C.c_uint32 * len(offs))(*offs)
        self.lib.oracle_verify_batch.restype = C.c_int
        return self.lib.oracle_verify_batch(C.c_size_t(len(msgs)), sig, b''.join(msgs), arr, b''.join(pks), dst, C.c_size_t(len(dst)))

    @staticmethod
    def _pack(msgs):
        offs = [0]
        for m in msgs:
            offs.append(offs[-1] + len(m))
        return b''.join(msgs), (C.c_uint32 * len(offs))(*offs)

    def aggregate_sign(self, msgs, sks, dst=DST_DEFAULT, threads=8):
        """-> (list of 48-byte public keys, aggregate signature sum_i sk_i H(m_i))"""
        blob, offs = self._pack(msgs)
        n = len(msgs)
        pks = self._out(48 * n)
        sig = self._out(96)
        self.lib.oracle_aggregate_sign(C.c_size_t(n), blob, offs, b''.join(sks), dst, C.c_size_t(len(dst)), pks, sig, C.c_int(threads))
        return [pks.raw[48 * i:48 * (i + 1)] for i in range(n)], sig.raw

    def verify_batch_mt(self, sig, msgs, pks, dst=DST_DEFAULT, threads=8):
        blob, offs = self._pack(msgs)
        self.lib.oracle_verify_batch_mt.restype = C.c_int
        return self.lib.oracle_verify_batch_mt(C.c_size_t(len(msgs)), sig, blob, offs, b''.join(pks), dst, C.c_size_t(len(dst)), C.c_int(threads))

    def aggregate_public_keys(self, pks):
        return self.call('aggregate_public_keys', 48, C.c_size_t(len(pks)), b''.join(pks))

    def aggregate_signatures(self, sigs):
        return self.call('aggregate_signatures', 96, C.c_size_t(len(sigs)), b''.join(sigs))


def load(rebuild=True):
    so = os.path.join(ODIR, 'libnbls_oracle.so')
    if rebuild or not os.path.exists(so):
        build()
    return Oracle(C.CDLL(so))

"""ctypes binding of libnbls.so (include/nbls.h).  Mirrors the reference's batched entry points:
pairing (index.ts:715), the Miller-product core of verify/verifyBatch (index.ts:763-766, 811-816) and
Fp12.finalExponentiate (math.ts:856).

A process that also uses PyTorch-ROCm must import torch BEFORE the first Engine is created: the torch wheel bundles its own HIP
runtime, libnbls.so binds to whichever runtime is already loaded, and torch loaded second reports "No HIP GPUs are available"."""
import ctypes as C
import os
try:
    import numpy as _np
except ImportError:      # the binding itself needs only ctypes
    _np = None

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4   # include/nbls.h NBLS_ABI_VERSION: checked at load (round 4's advisor: an ABI-1 caller of *_partial read stale bytes from an ABI-2 library with no error)
PROGRAMS = []   # names of the step programs in the library's numbering (filled by load_library from nbls_program_name)
DST_DEFAULT = b'BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_'   # htfDefaults.DST, reference index.ts:64


class NblsError(RuntimeError):
    pass


def lib_path():
    return os.environ.get('NBLS_LIBRARY') or os.path.join(_HERE, 'libnbls.so')   # NBLS_LIBRARY: explicit path of the engine library


def load_library():
    p = lib_path()
    if not os.path.exists(p):
        raise NblsError('libnbls.so is not built (run __graft_entry__.build() or make -C noble-bls12-381_amd/csrc)')
    lib = C.CDLL(p)
    lib.nbls_strerror.restype = C.c_char_p
    lib.nbls_config_describe.restype = C.c_char_p
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.nbls_init.argtypes = [i32, C.POINTER(vp)]
    lib.nbls_destroy.argtypes = [vp]
    lib.nbls_last_hip_error.argtypes = [vp]
    lib.nbls_device_synchronize.argtypes = [vp]
    lib.nbls_pairing_batch.argtypes = [vp, sz, vp, vp, i32, i32, vp, vp]
    lib.nbls_pairing_batch_dev.argtypes = [vp, sz, vp, vp, i32, vp, vp]
    lib.nbls_miller_product.argtypes = [vp, sz, vp, vp, i32, i32, vp, vp]
    lib.nbls_miller_product_dev.argtypes = [vp, sz, vp, vp, i32, vp, vp]
    lib.nbls_final_exp_batch.argtypes = [vp, sz, vp, vp]
    lib.nbls_final_exp_batch_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_fp12_product_final_dev.argtypes = [vp, sz, vp, i32, vp, vp]
    lib.nbls_program_stats.argtypes = [vp, i32, C.POINTER(C.c_uint32)]
    lib.nbls_g1_validate_batch.argtypes = [vp, sz, vp, vp]
    lib.nbls_g2_validate_batch.argtypes = [vp, sz, vp, vp]
    lib.nbls_g1_decompress_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_g2_decompress_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_hash_to_g2_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp]
    lib.nbls_hash_to_g1_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp]
    lib.nbls_encode_to_g1_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp]
    lib.nbls_encode_to_g2_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp]
    lib.nbls_g1_sum.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_g2_sum.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_g1_compress_batch.argtypes = [vp, sz, vp, vp]
    lib.nbls_g2_compress_batch.argtypes = [vp, sz, vp, vp]
    lib.nbls_g1_mul_batch.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.nbls_g2_mul_batch.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.nbls_g1_msm.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.nbls_g2_msm.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.nbls_msm_dev.argtypes = [vp, i32, sz, vp, vp, C.c_uint32, vp, vp, vp]
    lib.nbls_sign_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp]
    lib.nbls_sign_batch_dev.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.nbls_verify_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, C.POINTER(i32)]
    lib.nbls_verify_batch_dev_inputs.argtypes = [vp, sz, vp, vp, vp, C.POINTER(i32), vp, vp]
    lib.nbls_verify_batch_msgs_dev.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, C.POINTER(i32), vp]
    lib.nbls_verify_batch_partial_dev.argtypes = [vp, sz, vp, vp, vp, vp, C.POINTER(i32), vp, vp]
    lib.nbls_g2_prepare.argtypes = [vp, sz, vp, vp]
    lib.nbls_g2_prepare_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_lines_to_wire_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_lines_from_wire_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_pairing_prepared_dev.argtypes = [vp, sz, vp, vp, sz, i32, vp, vp]
    lib.nbls_miller_product_prepared_dev.argtypes = [vp, sz, vp, vp, sz, i32, vp, vp]
    lib.nbls_pairing_prepared.argtypes = [vp, sz, vp, vp, sz, i32, i32, vp]
    lib.nbls_set_tuning.argtypes = [vp, i32, C.c_longlong]
    for nm in ('nbls_g1_from_hex_batch', 'nbls_g2_from_hex_batch', 'nbls_g2_from_signature_batch'):
        getattr(lib, nm).argtypes = [vp, sz, vp, sz, vp, vp]
    lib.nbls_g1_to_hex_batch.argtypes = [vp, sz, vp, vp, i32, vp]
    lib.nbls_g2_to_hex_batch.argtypes = [vp, sz, vp, vp, i32, vp]
    lib.nbls_g1_clear_cofactor_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_g2_clear_cofactor_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.nbls_init_multi.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
    lib.nbls_destroy_multi.argtypes = [vp]
    lib.nbls_multi_device_count.argtypes = [vp]
    lib.nbls_multi_peer_access.argtypes = [vp, i32]
    lib.nbls_multi_pairing_batch.argtypes = [vp, sz, vp, vp, i32, i32, vp, vp]
    lib.nbls_multi_miller_product.argtypes = [vp, sz, vp, vp, i32, i32, vp, vp]
    lib.nbls_multi_verify_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, C.POINTER(i32)]
    lib.nbls_program_name.restype = C.c_char_p
    lib.nbls_program_name.argtypes = [i32]
    lib.nbls_context_device.argtypes = [vp]
    lib.nbls_pool_init.argtypes = [i32, i32, C.POINTER(vp)]
    lib.nbls_pool_destroy.argtypes = [vp]
    lib.nbls_pool_depth.argtypes = [vp]
    lib.nbls_pool_context.restype = vp
    lib.nbls_pool_context.argtypes = [vp, i32]
    lib.nbls_pool_next_slot.argtypes = [vp]
    lib.nbls_pool_pairing_batch_dev.argtypes = [vp, sz, vp, vp, i32, vp, C.POINTER(i32)]
    lib.nbls_pool_synchronize.argtypes = [vp]
    lib.nbls_program_kernel.restype = C.c_char_p
    lib.nbls_program_kernel.argtypes = [vp, i32]
    if lib.nbls_abi_version() != ABI_VERSION:
        raise NblsError('libnbls.so has ABI %d, this binding is written for ABI %d (rebuild: make -C noble-bls12-381_amd/csrc)' % (lib.nbls_abi_version(), ABI_VERSION))
    if not PROGRAMS:
        PROGRAMS.extend(lib.nbls_program_name(k).decode() for k in range(lib.nbls_program_count()))
    lib.nbls_timing_enable.argtypes = [vp, i32]
    lib.nbls_timing_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    return lib


class Engine:
    """One engine context = one GPU."""

    def __init__(self, device_id=0, _handle=None):
        self.lib = load_library()
        self._owned = _handle is None
        if _handle is not None:      # a context owned by a pool / multi handle (nbls_pool_context, nbls_multi_context): not destroyed by this object
            self.h = C.c_void_p(_handle)
            self.device_id = self.lib.nbls_context_device(self.h)
            return
        h = C.c_void_p()
        r = self.lib.nbls_init(device_id, C.byref(h))
        if r != 0:
            raise NblsError('nbls_init failed: %s (code %d)' % (self.lib.nbls_strerror(r).decode(), r))
        self.h = h
        self.device_id = device_id

    def close(self):
        if getattr(self, 'h', None):
            if self._owned:
                self.lib.nbls_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise NblsError('%s (code %d, hip %d)' % (self.lib.nbls_strerror(r).decode(), r, self.lib.nbls_last_hip_error(self.h)))

    # ---- host-buffer entry points (bytes in, bytes out)
    def pairing_batch(self, g1_aff, g2_aff, with_final_exp=True, validate=False):
        n = len(g1_aff) // 96
        assert len(g1_aff) == 96 * n and len(g2_aff) == 192 * n
        out = C.create_string_buffer(576 * n)
        st = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.nbls_pairing_batch(self.h, n, g1_aff, g2_aff, int(with_final_exp), int(validate), out, st))
        return out.raw, st.raw[:n]

    def miller_product(self, g1_aff, g2_aff, final_exp=True, validate=False):
        n = len(g1_aff) // 96
        out = C.create_string_buffer(576)
        st = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.nbls_miller_product(self.h, n, g1_aff, g2_aff, int(final_exp), int(validate), out, st))
        return out.raw, st.raw[:n]

    # ---- prepared G2 points (PointG2.pairingPrecomputes index.ts:703-711, PointG1.millerLoop index.ts:452-454)
    LINE_TABLE_BYTES = 26112
    LINE_WIRE_BYTES = 19584

    def g2_prepare(self, g2_aff):
        """n affine G2 points -> n line tables in wire form (68 x [Fp2, Fp2, Fp2] as Fp2.toBytes, 19,584 B each)"""
        n = len(g2_aff) // 192
        out = C.create_string_buffer(max(self.LINE_WIRE_BYTES * n, 1))
        self._chk(self.lib.nbls_g2_prepare(self.h, n, g2_aff, out))
        return out.raw[:self.LINE_WIRE_BYTES * n]

    def pairing_prepared(self, g1_aff, tables_wire, with_final_exp=True, product=False):
        """n G1 points against n (or 1) prepared tables: n pairings, or with product=True the one product of their Miller values"""
        n = len(g1_aff) // 96
        nt = len(tables_wire) // self.LINE_WIRE_BYTES
        out = C.create_string_buffer(576 * (1 if product else max(n, 1)))
        self._chk(self.lib.nbls_pairing_prepared(self.h, n, g1_aff, tables_wire, nt, int(with_final_exp), int(product), out))
        return out.raw[:576 * (1 if product else n)]

    def g2_prepare_dev(self, n, d_g2, d_tables, stream=None):
        self._chk(self.lib.nbls_g2_prepare_dev(self.h, n, d_g2, d_tables, stream))

    def pairing_prepared_dev(self, n, d_g1, d_tables, d_out, with_final_exp=True, shared_table=False, stream=None):
        self._chk(self.lib.nbls_pairing_prepared_dev(self.h, n, d_g1, d_tables, 0 if shared_table else self.LINE_TABLE_BYTES, int(with_final_exp), d_out, stream))

    def miller_product_prepared_dev(self, n, d_g1, d_tables, d_out, final_exp=True, shared_table=False, stream=None):
        self._chk(self.lib.nbls_miller_product_prepared_dev(self.h, n, d_g1, d_tables, 0 if shared_table else self.LINE_TABLE_BYTES, int(final_exp), d_out, stream))

    def final_exp_batch(self, fp12s):
        n = len(fp12s) // 576
        out = C.create_string_buffer(576 * max(n, 1))
        self._chk(self.lib.nbls_final_exp_batch(self.h, n, fp12s, out))
        return out.raw[:576 * n]

    def validate_batch(self, pts, g2=False):
        sz = 192 if g2 else 96
        n = len(pts) // sz
        st = C.create_string_buffer(max(n, 1))
        self._chk((self.lib.nbls_g2_validate_batch if g2 else self.lib.nbls_g1_validate_batch)(self.h, n, pts, st))
        return list(st.raw[:n])

    def decompress_batch(self, comp, g2=False):
        e = 96 if g2 else 48
        n = len(comp) // e
        out = C.create_string_buffer(max(2 * e * n, 1))
        st = C.create_string_buffer(max(n, 1))
        self._chk((self.lib.nbls_g2_decompress_batch if g2 else self.lib.nbls_g1_decompress_batch)(self.h, n, comp, out, st))
        return out.raw[:2 * e * n], list(st.raw[:n])

    # ---- every wire form of the point codecs (include/nbls.h): kind 'g1' / 'g2' = fromHex, 'sig' = PointG2.fromSignature
    def decode_points(self, kind, blob, length):
        """-> (canonical affine wire bytes, status list); length = bytes per encoded point"""
        n = len(blob) // length
        a = 96 if kind == 'g1' else 192
        out = C.create_string_buffer(max(a * n, 1)); st = C.create_string_buffer(max(n, 1))
        f = {'g1': self.lib.nbls_g1_from_hex_batch, 'g2': self.lib.nbls_g2_from_hex_batch, 'sig': self.lib.nbls_g2_from_signature_batch}[kind]
        self._chk(f(self.h, n, blob, length, out, st))
        return out.raw[:a * n], list(st.raw[:n])

    def encode_points(self, aff, g2=False, compressed=True, zero=None):
        a = 192 if g2 else 96
        n = len(aff) // a
        c = a // 2 if compressed else a
        out = C.create_string_buffer(max(c * n, 1))
        z = bytes(zero) if zero is not None else None
        self._chk((self.lib.nbls_g2_to_hex_batch if g2 else self.lib.nbls_g1_to_hex_batch)(self.h, n, aff, z, int(compressed), out))
        return out.raw[:c * n]

    def clear_cofactor(self, aff, g2=False):
        a = 192 if g2 else 96
        n = len(aff) // a
        out = C.create_string_buffer(max(a * n, 1)); st = C.create_string_buffer(max(n, 1))
        self._chk((self.lib.nbls_g2_clear_cofactor_batch if g2 else self.lib.nbls_g1_clear_cofactor_batch)(self.h, n, aff, out, st))
        return out.raw[:a * n], list(st.raw[:n])

    @staticmethod
    def _pack(msgs):
        """messages -> (their bytes back to back, uint32 offsets with the end appended) as the C ABI takes them"""
        n = len(msgs)
        if _np is not None and n >= 256:      # 65,536 messages: 6 ms instead of 16 ms of Python loop
            offs = _np.zeros(n + 1, dtype=_np.uint32)
            _np.cumsum(_np.fromiter(map(len, msgs), dtype=_np.uint32, count=n), out=offs[1:])
            return b''.join(msgs), (C.c_uint32 * (n + 1)).from_buffer(offs)
        offs = [0]
        for m in msgs:
            offs.append(offs[-1] + len(m))
        return b''.join(msgs), (C.c_uint32 * len(offs))(*offs)

    def hash_to_g2_batch(self, msgs, dst=DST_DEFAULT):
        blob, offs = self._pack(msgs)
        out = C.create_string_buffer(max(192 * len(msgs), 1))
        self._chk(self.lib.nbls_hash_to_g2_batch(self.h, len(msgs), blob, offs, dst, len(dst), out))
        return out.raw[:192 * len(msgs)]

    def hash_to_curve_batch(self, msgs, dst=DST_DEFAULT, g2=False, encode=False):
        """PointG1/PointG2 .hashToCurve (encode=False) or .encodeToCurve (encode=True) -> affine wire bytes"""
        if g2 and not encode:
            return self.hash_to_g2_batch(msgs, dst)
        blob, offs = self._pack(msgs)
        sz = 192 if g2 else 96
        out = C.create_string_buffer(max(sz * len(msgs), 1))
        f = self.lib.nbls_encode_to_g2_batch if g2 else (self.lib.nbls_encode_to_g1_batch if encode else self.lib.nbls_hash_to_g1_batch)
        self._chk(f(self.h, len(msgs), blob, offs, dst, len(dst), out))
        return out.raw[:sz * len(msgs)]

    def point_sum(self, pts, g2=False):
        sz = 192 if g2 else 96
        out = C.create_string_buffer(sz)
        st = C.create_string_buffer(1)
        self._chk((self.lib.nbls_g2_sum if g2 else self.lib.nbls_g1_sum)(self.h, len(pts) // sz, pts, out, st))
        return out.raw, st.raw[0]

    # ---- secret-scalar side (reference index.ts:738-752); scalars / keys are 32-byte big-endian strings
    P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab

    def point_mul_batch(self, scalars, pts=None, g2=False):
        """[k_i]P_i -> (affine wire bytes, status bytes); pts=None multiplies the G1 generator (PointG1.fromPrivateKey)"""
        n = len(scalars)
        sz = 192 if g2 else 96
        out = C.create_string_buffer(max(sz * n, 1)); st = C.create_string_buffer(max(n, 1))
        f = self.lib.nbls_g2_mul_batch if g2 else self.lib.nbls_g1_mul_batch
        self._chk(f(self.h, n, pts, b''.join(scalars), out, st))
        return out.raw[:sz * n], st.raw[:n]

    def msm(self, pts, scalars, g2=False):
        """sum_i [k_i]P_i (bucket method on the GPU) -> (affine wire bytes, status); status 1 = the sum is the zero point.
        pts: concatenated affine wire points, scalars: list of 32-byte big-endian strings"""
        n = len(scalars)
        sz = 192 if g2 else 96
        assert len(pts) == sz * n
        out = C.create_string_buffer(sz); st = C.c_int8(0)
        f = self.lib.nbls_g2_msm if g2 else self.lib.nbls_g1_msm
        self._chk(f(self.h, n, pts, b''.join(scalars), out, C.byref(st)))
        return out.raw, st.value

    def msm_dev(self, g2, n, d_pts, d_scalars, nbits, d_out, d_status, stream=0):
        self._chk(self.lib.nbls_msm_dev(self.h, int(bool(g2)), n, d_pts, d_scalars, nbits, d_out, d_status, stream))

    @classmethod
    def compress_g1(cls, aff96):
        """PointG1.toHex(true) of a non-zero affine point (index.ts:359-371)"""
        x = int.from_bytes(aff96[:48], 'big'); y = int.from_bytes(aff96[48:], 'big')
        return (x + ((y * 2) // cls.P_MOD << 381) + (1 << 383)).to_bytes(48, 'big')

    @classmethod
    def compress_g2(cls, aff192):
        """PointG2.toSignature of a non-zero affine point (index.ts:586-602)"""
        x0, x1, y0, y1 = (int.from_bytes(aff192[48 * i:48 * i + 48], 'big') for i in range(4))
        tmp = y1 * 2 if y1 > 0 else y0 * 2
        z1 = x1 + ((tmp // cls.P_MOD) << 381) + (1 << 383)
        return z1.to_bytes(48, 'big') + x0.to_bytes(48, 'big')

    def compress_batch(self, aff, g2=False):
        """PointG1.toHex(true) / PointG2.toSignature for a batch of non-zero affine points, on the GPU"""
        sz = 192 if g2 else 96
        n = len(aff) // sz
        out = C.create_string_buffer(max(n * sz // 2, 1))
        self._chk((self.lib.nbls_g2_compress_batch if g2 else self.lib.nbls_g1_compress_batch)(self.h, n, aff, out))
        return out.raw[:n * sz // 2]

    def get_public_keys(self, keys):
        """getPublicKey for a batch of private keys -> list of 48-byte compressed keys; raises like the reference on a zero key"""
        aff, st = self.point_mul_batch(keys)
        if any(st):
            raise NblsError('Private key must be 0 < key < CURVE.r')
        c = self.compress_batch(aff)
        return [c[48 * i:48 * i + 48] for i in range(len(keys))]

    def sign_batch_affine(self, msgs, keys, dst=DST_DEFAULT):
        """nbls_sign_batch as is: (n * 192 affine signature bytes, status bytes)"""
        blob, offs = self._pack(msgs)
        n = len(msgs)
        out = C.create_string_buffer(max(192 * n, 1)); st = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.nbls_sign_batch(self.h, n, blob, offs, dst, len(dst), b''.join(keys), out, st))
        return out.raw[:192 * n], st.raw[:n]

    def sign_batch_dev(self, n, d_msgs, d_offsets, d_keys32, d_out192, d_status, dst=DST_DEFAULT, stream=None):
        """nbls_sign_batch_dev: everything resident in device memory (pointers as integers); synchronises"""
        self._chk(self.lib.nbls_sign_batch_dev(self.h, n, C.c_void_p(d_msgs), C.c_void_p(d_offsets), dst, len(dst), C.c_void_p(d_keys32), C.c_void_p(d_out192), C.c_void_p(d_status), stream))

    def sign_packed(self, n, blob, offs, keys_blob, out, st, dst=DST_DEFAULT):
        """nbls_sign_batch on buffers the caller has already packed (message bytes, ctypes uint32 offsets, n * 32 key bytes) into preallocated ctypes outputs: the C-ABI call by itself"""
        self._chk(self.lib.nbls_sign_batch(self.h, n, blob, offs, dst, len(dst), keys_blob, out, st))

    def sign_batch(self, msgs, keys, dst=DST_DEFAULT):
        """sign(msg_i, key_i) -> list of 96-byte compressed signatures"""
        aff, st = self.sign_batch_affine(msgs, keys, dst)
        if any(st):
            raise NblsError('Private key must be 0 < key < CURVE.r')
        c = self.compress_batch(aff, g2=True)
        return [c[96 * i:96 * i + 96] for i in range(len(msgs))]

    def verify_batch(self, sig96, msgs, pks48, dst=DST_DEFAULT):
        """-> True/False; raises NblsError where the reference throws while decoding its arguments"""
        blob, offs = self._pack(msgs)
        ok = C.c_int(0)
        self._chk(self.lib.nbls_verify_batch(self.h, len(msgs), sig96, blob, offs, b''.join(pks48), dst, len(dst), C.byref(ok)))
        return bool(ok.value)

    def verify_batch_dev(self, n, d_sig, d_uniform, d_pk, stream=None):
        ok = C.c_int(0)
        self._chk(self.lib.nbls_verify_batch_dev_inputs(self.h, n, d_sig, d_uniform, d_pk, C.byref(ok), None, stream))
        return bool(ok.value)

    def verify_batch_msgs_dev(self, n, d_sig, d_msgs, d_offsets, d_pk, dst=DST_DEFAULT, stream=None):
        """verifyBatch with signature, message bytes + uint32 offsets and compressed keys resident in HBM: expand_message_xmd runs on the device as part of the call"""
        ok = C.c_int32(0)
        self._chk(self.lib.nbls_verify_batch_msgs_dev(self.h, n, d_sig, d_msgs, d_offsets, d_pk, dst, len(dst), C.byref(ok), stream))
        return bool(ok.value)

    def verify_batch_partial_dev(self, n, d_sig, d_uniform, d_pk, d_out, stream=None):
        """one rank's share of a multi-GPU verifyBatch: Miller product of its n pairs (plus (-G, S) when d_sig is not None/0) without
        the final exponentiation -> 576 wire bytes at d_out; returns True when a zero point was met (the batch verifies false)"""
        z = C.c_int(0)
        self._chk(self.lib.nbls_verify_batch_partial_dev(self.h, n, d_sig or None, d_uniform, d_pk, d_out, C.byref(z), None, stream))
        return bool(z.value)

    # ---- device-pointer entry points (torch uint8 CUDA tensors); enqueue on `stream` (int handle) or the context stream
    def pairing_batch_dev(self, n, d_g1, d_g2, d_out, with_final_exp=True, stream=None):
        self._chk(self.lib.nbls_pairing_batch_dev(self.h, n, d_g1, d_g2, int(with_final_exp), d_out, stream))

    def miller_product_dev(self, n, d_g1, d_g2, d_out, final_exp=True, stream=None):
        self._chk(self.lib.nbls_miller_product_dev(self.h, n, d_g1, d_g2, int(final_exp), d_out, stream))

    def final_exp_batch_dev(self, n, d_in, d_out, stream=None):
        self._chk(self.lib.nbls_final_exp_batch_dev(self.h, n, d_in, d_out, stream))

    def fp12_product_final_dev(self, n, d_in, d_out, final_exp=True, stream=None):
        self._chk(self.lib.nbls_fp12_product_final_dev(self.h, n, d_in, int(final_exp), d_out, stream))

    def set_expc_min(self, n):
        """items from which the final exponentiation uses compressed cyclotomic squarings (NBLS_TUNE_EXPC_MIN; 0 = always)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 3, n))

    def set_split_miller_min(self, n):
        """pairs from which the Miller loop runs as LINES + ACC (0: always, a huge value: never)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 1, n))

    def set_halves_min(self, n):
        """pairs from which pairing_batch_dev runs a batch as two halves on two streams (default 16384; 0: never)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 2, n))

    def set_chain_max(self, n):
        """items below which the middle of the final exponentiation is one chained launch (NBLS_TUNE_CHAIN_MAX; default 8192, 0: one launch per program)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 4, n))

    def set_sac_max(self, n):
        """keys up to which sign's ladder is the sign-aligned one-addition-per-bit form (NBLS_TUNE_SAC_MAX; default 6144, 0: the windowed psi-split ladder at every size)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 8, n))

    def set_wide_max(self, n):
        """items up to which the programs that allow it run on the one-limb-per-lane interpreter (NBLS_TUNE_WIDE_MAX; an experiment, measured slower than the lane-split forms: default 0 = never)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 10, n))

    def set_ls_max(self, ls_max=None, ls2_max=None):
        """items up to which the pairing programs run in their four-lane / two-lane forms (NBLS_TUNE_LS_MAX = 13, NBLS_TUNE_LS2_MAX = 14; defaults 1024 / 2048, pool contexts 0 / 0)"""
        if ls_max is not None: self._chk(self.lib.nbls_set_tuning(self.h, 13, int(ls_max)))
        if ls2_max is not None: self._chk(self.lib.nbls_set_tuning(self.h, 14, int(ls2_max)))

    def set_inv_wide_max(self, n):
        """elements up to which an Fp inversion launch runs with one limb per lane (NBLS_TUNE_INV_WIDE_MAX = 12; default 4096, pool contexts 256, 0: never)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 12, int(n)))

    def set_h2c_norm_min(self, n):
        """messages from which hash-to-G2 takes its SWU square root by the norm method (NBLS_TUNE_H2C_NORM_MIN = 11; default 32768, 0: always)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 11, int(n)))

    def set_pt_ls2_max(self, n):
        """items up to which the G2 point chains of verify / sign run in their two-lane forms (NBLS_TUNE_PT_LS2_MAX; default 4096, 0: never)"""
        self._chk(self.lib.nbls_set_tuning(self.h, 9, n))

    def set_verify_pipeline(self, chunks=None, last_pct=None, pipe_min=None):
        """verifyBatch as a software pipeline (NBLS_TUNE_VERIFY_CHUNKS / _LAST_PCT / _PIPE_MIN): number of chunks (0 / 1: one), size of the last chunk in
        per cent of the batch, signatures from which a call is chunked at all"""
        for key, v in ((5, chunks), (6, last_pct), (7, pipe_min)):
            if v is not None:
                self._chk(self.lib.nbls_set_tuning(self.h, key, v))

    def synchronize(self):
        self._chk(self.lib.nbls_device_synchronize(self.h))

    def program_stats(self, name):
        o = (C.c_uint32 * 8)()
        self._chk(self.lib.nbls_program_stats(self.h, PROGRAMS.index(name), o))
        keys = ['steps', 'dot_steps', 'lin_steps', 'dot_ops', 'products', 'lin_ops', 'slots', 'lds_bytes']
        return dict(zip(keys, list(o)))

    def program_kernel(self, name):
        """the kernel that executes step program `name` in this context: 'nbls_aot_<kernel>' or 'nbls_vm_kernel[_ls4]' (the interpreter)"""
        k = self.lib.nbls_program_kernel(self.h, PROGRAMS.index(name))
        if k is None:
            raise NblsError('nbls_program_kernel(%s) failed' % name)
        return k.decode()

    def kernel_bindings(self):
        """{program: kernel} over every step program"""
        return {p: self.program_kernel(p) for p in PROGRAMS}

    def device_synchronize(self):
        self._chk(self.lib.nbls_device_synchronize(self.h))

    def timing_enable(self, on=True):
        self._chk(self.lib.nbls_timing_enable(self.h, int(on)))

    def tower_op(self, field, op, a, b=None, c=None, d=None, param=0):
        """one tower operation (include/nbls.h NBLS_TOP_*) on len(a) / (48 * field) elements given as wire bytes -> wire bytes"""
        esz = 48 * field
        n = len(a) // esz
        out = C.create_string_buffer(n * esz)
        args = [C.c_char_p(x) if x is not None else None for x in (a, b, c, d)]
        self._chk(self.lib.nbls_tower_op_batch(self.h, C.c_int(field), C.c_int(op), C.c_int(param), C.c_size_t(n), args[0], args[1], args[2], args[3], out))
        return out.raw

    def config_describe(self):
        """the environment switches the library has read so far, with the values in force"""
        return self.lib.nbls_config_describe().decode()

    def timing_read(self):
        """-> {kernel name: (total ms, launches)} since timing_enable(True)"""
        n = len(PROGRAMS) + 1
        ms = (C.c_float * n)()
        cnt = (C.c_uint32 * n)()
        self._chk(self.lib.nbls_timing_read(self.h, ms, cnt))
        names = PROGRAMS + ['fp_inv']
        return {names[i]: (ms[i], cnt[i]) for i in range(n) if cnt[i]}


class MultiEngine:
    """Several GPUs of one node behind one handle (include/nbls.h nbls_init_multi): contiguous shards, one host thread and stream per
    device, 576-byte Fp12 partials gathered on the first device by hipMemcpyPeer.  devices=None takes every visible device; a device id
    may be listed more than once (several contexts on one GPU: the tests exercise the sharded paths on a one-GPU box that way)."""

    def __init__(self, devices=None):
        self.lib = load_library()
        h = C.c_void_p()
        if devices is None:
            r = self.lib.nbls_init_multi(0, None, C.byref(h))
        else:
            ids = (C.c_int * len(devices))(*devices)
            r = self.lib.nbls_init_multi(len(devices), ids, C.byref(h))
        if r != 0:
            raise NblsError('nbls_init_multi failed: %s (code %d)' % (self.lib.nbls_strerror(r).decode(), r))
        self.h = h
        self.n_devices = self.lib.nbls_multi_device_count(h)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.nbls_destroy_multi(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise NblsError('%s (code %d)' % (self.lib.nbls_strerror(r).decode(), r))

    def pairing_batch(self, g1_aff, g2_aff, with_final_exp=True, validate=False):
        n = len(g1_aff) // 96
        out = C.create_string_buffer(max(576 * n, 1)); st = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.nbls_multi_pairing_batch(self.h, n, g1_aff, g2_aff, int(with_final_exp), int(validate), out, st))
        return out.raw[:576 * n], st.raw[:n]

    def miller_product(self, g1_aff, g2_aff, final_exp=True, validate=False):
        n = len(g1_aff) // 96
        out = C.create_string_buffer(576); st = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.nbls_multi_miller_product(self.h, n, g1_aff, g2_aff, int(final_exp), int(validate), out, st))
        return out.raw, st.raw[:n]

    def verify_batch(self, sig96, msgs, pks48, dst=DST_DEFAULT):
        blob, offs = Engine._pack(msgs)
        ok = C.c_int(0)
        self._chk(self.lib.nbls_multi_verify_batch(self.h, len(msgs), sig96, blob, offs, b''.join(pks48), dst, len(dst), C.byref(ok)))
        return bool(ok.value)

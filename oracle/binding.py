"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from heongpu_amd/.  See hegpu_oracle.h
("parity unpinned" notice).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

u64 = ctypes.c_uint64
P64 = ctypes.POINTER(u64)
PI = ctypes.POINTER(ctypes.c_int)
BFV, CKKS = 1, 2


class OMod(ctypes.Structure):
    _fields_ = [("value", u64), ("bit", u64), ("mu", u64)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = ctypes.CDLL(LIB_PATH)
    L.o_mod.restype = OMod
    L.o_mod.argtypes = [u64]
    for nm in ("o_add", "o_sub", "o_mult"):
        getattr(L, nm).restype = u64
        getattr(L, nm).argtypes = [u64, u64, ctypes.POINTER(OMod)]
    L.o_reduce_forced.restype = u64
    L.o_reduce_forced.argtypes = [u64, ctypes.POINTER(OMod)]
    L.o_min_primitive_root.restype = u64
    L.o_min_primitive_root.argtypes = [u64, u64]
    L.o_is_prime.argtypes = [u64]
    L.o_generate_primes.argtypes = [u64, PI, ctypes.c_int, P64]
    L.o_generate_internal_primes.argtypes = [u64, ctypes.c_int, P64]
    L.o_default_modulus_128.argtypes = [u64, P64]
    L.o_default_modulus.argtypes = [u64, ctypes.c_int, P64]
    L.o_max_logq.argtypes = [u64, ctypes.c_int]
    L.o_steps_to_galois_elt.argtypes = [ctypes.c_int] * 3
    L.o_fill_poly.argtypes = [ctypes.c_void_p, u64, ctypes.c_int, u64, u64]
    L.o_ctx_create.restype = ctypes.c_void_p
    L.o_ctx_create.argtypes = [ctypes.c_int, ctypes.c_int, P64, ctypes.c_int, ctypes.c_int, u64]
    L.o_ctx_free.argtypes = [ctypes.c_void_p]
    L.o_ctx_get.restype = ctypes.c_long
    L.o_ctx_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
    vp = ctypes.c_void_p
    ci = ctypes.c_int
    L.o_gpu_ntt.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_gpu_intt.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci]
    L.o_gpu_ntt_modulus_ordered.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.o_gpu_ntt_poly_ordered.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.o_addition.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_substraction.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_negation.argtypes = [vp, vp, vp, ci, ci, ci]
    L.o_cross_multiplication.argtypes = [vp, vp, vp, vp, ci, ci]
    L.o_ckks_multiply.argtypes = [vp, vp, vp, vp, ci]
    L.o_ckks_relinearize.argtypes = [vp, vp, vp, ci]
    L.o_ckks_rescale.argtypes = [vp, vp, ci]
    L.o_ckks_apply_galois.argtypes = [vp, vp, vp, vp, ci, ci]
    L.o_ckks_rotate_hoisted.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ci, ci]
    L.o_bfv_multiply.argtypes = [vp, vp, vp, vp]
    L.o_bfv_relinearize.argtypes = [vp, vp, vp]
    L.o_bfv_apply_galois.argtypes = [vp, vp, vp, vp, ci]
    L.o_ckks_mul_relin_batch.argtypes = [vp, vp, vp, vp, vp, ci, ci]
    L.o_ckks_mul_relin_batch_tiled.argtypes = [vp, vp, vp, ci, ci, vp, vp, ci, ci]
    L.o_ckks_mul_relin_batch_tiled.restype = ci
    L.o_bfv_relinearize_II.argtypes = [vp, vp, vp]
    L.o_bfv_apply_galois_II.argtypes = [vp, vp, vp, vp, ci]
    L.o_ckks_relinearize_II.argtypes = [vp, vp, vp, ci]
    L.o_ckks_apply_galois_II.argtypes = [vp, vp, vp, vp, ci, ci]
    L.o_gen_secret_key.argtypes = [vp, vp, ci, vp]
    L.o_gen_public_key.argtypes = [vp, vp, vp, vp]
    L.o_gen_switch_key.argtypes = [vp, vp, vp, ci, vp]
    L.o_gen_switch_key_new_old.argtypes = [vp, vp, vp, vp, vp]
    L.o_ckks_encrypt.argtypes = [vp, vp, vp, vp, vp]
    L.o_ckks_decrypt.argtypes = [vp, vp, vp, ci, vp]
    L.o_bfv_encrypt.argtypes = [vp, vp, vp, vp, vp]
    L.o_bfv_decrypt.argtypes = [vp, vp, vp, vp]
    L.o_bfv_encode.argtypes = [vp, vp, ci, vp]
    L.o_tfhe_gen_secret.argtypes = [vp, vp, vp]
    L.o_tfhe_gen_bootkey.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.o_tfhe_encrypt.argtypes = [vp, vp, vp, ci, vp, vp]
    L.o_tfhe_phase.argtypes = [vp, vp, vp, ci, vp]
    L.o_ckks_encode.argtypes = [vp, vp, ci, ctypes.c_double, vp]
    L.o_ckks_decode.argtypes = [vp, vp, ci, ctypes.c_double, vp]
    L.o_ckks_encode_ex.argtypes = [vp, ci, vp, ci, ctypes.c_double, vp]
    L.o_bfv_plain_to_ntt.argtypes = [vp, vp, vp]
    L.o_bfv_multiply_plain.argtypes = [vp, vp, vp, vp]
    L.o_negacyclic_shift.argtypes = [vp, vp, vp, ci, ci, ci]
    L.o_ckks_constant_op.argtypes = [vp, ci, vp, ctypes.c_double, vp, ci, ci]
    L.o_ckks_mult_i.argtypes = [vp, vp, vp, ci, ci, ci]
    L.o_ckks_gaussian_integer_op.argtypes = [vp, ci, vp, ctypes.c_double, ctypes.c_double, vp, ci, ci]
    L.o_ckks_decode_ex.argtypes = [vp, ci, vp, ci, ctypes.c_double, vp]
    L.o_bfv_decode.argtypes = [vp, vp, vp]
    L.o_fast_convertion.argtypes = [vp, vp, vp, vp]
    L.o_fast_floor.argtypes = [vp, vp, vp]
    L.o_cipher_broadcast.argtypes = [vp, vp, vp, ci, ci, ci]
    L.o_keyswitch_mac.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_divide_round_lastq.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci]
    L.o_drbg_block.argtypes = [ctypes.POINTER(ctypes.c_uint32), u64, u64, ctypes.POINTER(ctypes.c_uint32)]
    L.o_cipher_broadcast_leveled.argtypes = [vp, vp, vp, ci, ci, ci, ci]
    L.o_keyswitch_mac_leveled.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_divide_round_lastq_permute.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci]
    L.o_base_conversion_DtoQtilde.argtypes = [vp, vp, vp, ci]
    L.o_divide_round_lastq_extended.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    L.o_divide_round_lastq_leveled_stage_one.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci]
    L.o_divide_round_lastq_leveled_stage_two.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci]
    L.o_move_cipher_leveled.argtypes = [vp, vp, ci, ci]
    L.o_divide_round_lastq_rescale.argtypes = [vp, vp, vp, vp, vp, ci, ci]
    i32p = ctypes.c_void_p
    L.o_tfhe_create.restype = ctypes.c_void_p
    L.o_tfhe_free.argtypes = [vp]
    L.o_tfhe_prime.restype = u64
    L.o_tfhe_prime.argtypes = [vp]
    L.o_tfhe_encode_to_torus32.restype = ctypes.c_int32
    L.o_tfhe_encode_to_torus32.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.o_tfhe_gate_pre.argtypes = [i32p, i32p, i32p, i32p, i32p, i32p, ctypes.c_int32, ci, ci, ci, ci, ci]
    L.o_tfhe_bootstrapping.argtypes = [vp, i32p, i32p, vp, i32p, i32p, ctypes.c_int32, ci]
    L.o_tfhe_key_switching.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p, ci]
    L.o_tfhe_to_ntt.argtypes = [vp, i32p, vp]
    L.o_tfhe_polymul.argtypes = [vp, i32p, i32p, i32p]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data if a is not None else None


def splitmix64(x):
    """numpy-vectorised splitmix64 (same as o_splitmix64)."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def fill_poly(seed, limb, n, q):
    """splitmix64(seed + limb*2^32 + idx) mod q  (SURVEY.md 8d synthetic data)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) + (np.uint64(limb) << np.uint64(32))
        return splitmix64(idx) % np.uint64(q)


class ORng(ctypes.Structure):
    """orng_t of o_keygen.c: (256-bit ChaCha20 key, next stream id).  ORng(int): the reproducible
    64-bit test seed in key words 0,1 (as hegpu_rng_create); ORng(bytes of length 32): a full seed."""
    _fields_ = [("seed", ctypes.c_uint32 * 8), ("stream", ctypes.c_uint64)]

    def __init__(self, seed):
        if isinstance(seed, (bytes, bytearray)):
            words = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        else:
            v = int(seed) & (2**64 - 1)
            words = [v & 0xFFFFFFFF, v >> 32, 0, 0, 0, 0, 0, 0]
        super().__init__((ctypes.c_uint32 * 8)(*words), 0)


class OracleContext:
    def __init__(self, scheme, n_power, primes, q_count, p_count, plain_modulus=0):
        L = lib()
        arr = (u64 * len(primes))(*[int(x) for x in primes])
        self.h = L.o_ctx_create(scheme, n_power, arr, q_count, p_count, plain_modulus)
        self.L = L
        self.scheme, self.n_power, self.n = scheme, n_power, 1 << n_power
        self.Q, self.P, self.Qp = q_count, p_count, q_count + p_count
        self.primes = [int(x) for x in primes]
        self._mods = None

    def close(self):
        if self.h:
            self.L.o_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, name, cap=None):
        cap = cap or (self.n * (self.Qp + 70) + 4096)
        out = np.zeros(cap, dtype=np.uint64)
        cnt = self.L.o_ctx_get(self.h, name.encode(), _p(out), cap)
        if cnt < 0:
            raise KeyError(name)
        return out[:cnt].copy()

    def mods(self, values):
        arr = (OMod * len(values))()
        for i, v in enumerate(values):
            arr[i] = self.L.o_mod(int(v))
        return arr

    @property
    def qp_mods(self):
        if self._mods is None:
            self._mods = self.mods(self.primes)
        return self._mods

    # NTT family on the Q' chain tables (offsets as the reference passes them)
    def ntt(self, data, batch, mod_count, mod_offset=0, inverse=False, out=None):
        tab = self.table("intt_table" if inverse else "ntt_table")
        ninv = self.table("n_inverse")
        mods = self.qp_mods
        out = data if out is None else out
        moff = ctypes.addressof(mods) + mod_offset * ctypes.sizeof(OMod)
        tp = tab.ctypes.data + mod_offset * self.n * 8
        if inverse:
            self.L.o_gpu_intt(_p(data), _p(out), tp, moff, ninv.ctypes.data + mod_offset * 8, self.n_power, batch,
                              mod_count)
        else:
            self.L.o_gpu_ntt(_p(data), _p(out), tp, moff, self.n_power, batch, mod_count)
        return out

    def mods_addr(self, offset=0):
        return ctypes.addressof(self.qp_mods) + offset * ctypes.sizeof(OMod)

    def ckks_multiply(self, ct1, ct2, depth=0):
        l = self.Q - depth
        out = np.zeros(3 * l * self.n, dtype=np.uint64)
        self.L.o_ckks_multiply(self.h, _p(ct1), _p(ct2), _p(out), depth)
        return out

    def ckks_relinearize(self, ct3, key, depth=0):
        self.L.o_ckks_relinearize(self.h, _p(ct3), _p(key), depth)
        return ct3

    def ckks_rescale(self, ct, depth=0):
        self.L.o_ckks_rescale(self.h, _p(ct), depth)
        return ct

    def ckks_rotate_hoisted(self, ct, keys, galois_elts, depth=0):
        """keys: list of numpy key arrays (None for a zero element); returns [count][2][l][N]"""
        l = self.Q - depth
        cnt = len(galois_elts)
        out = np.zeros(cnt * 2 * l * self.n, dtype=np.uint64)
        kp = (ctypes.c_void_p * cnt)(*[(k.ctypes.data if k is not None else None) for k in keys])
        ge = (ctypes.c_int * cnt)(*[int(g) for g in galois_elts])
        self.L.o_ckks_rotate_hoisted(self.h, _p(ct), _p(out), kp, ge, cnt, depth)
        return out

    def ckks_apply_galois(self, ct, key, galois_elt, depth=0):
        l = self.Q - depth
        out = np.zeros(2 * l * self.n, dtype=np.uint64)
        self.L.o_ckks_apply_galois(self.h, _p(ct), _p(out), _p(key), galois_elt, depth)
        return out

    # key generation / encryption / decryption (o_keygen.c); rng = ORng(seed)
    def gen_secret_key(self, rng, hamming_weight=None):
        sk = np.zeros(self.Qp * self.n, dtype=np.uint64)
        self.L.o_gen_secret_key(self.h, ctypes.byref(rng), self.n // 2 if hamming_weight is None else hamming_weight,
                                _p(sk))
        return sk

    def gen_public_key(self, rng, sk):
        pk = np.zeros(2 * self.Qp * self.n, dtype=np.uint64)
        self.L.o_gen_public_key(self.h, ctypes.byref(rng), _p(sk), _p(pk))
        return pk

    def switch_key_digits(self):
        """digits of a key-switching key: Q (method I) or the depth-0 partition (method II)"""
        return self.Q if self.P == 1 else -(-self.Q // (2 if self.scheme == BFV else self.P))

    def gen_switch_key(self, rng, sk, galois_elt=0):
        key = np.zeros(self.switch_key_digits() * 2 * self.Qp * self.n, dtype=np.uint64)
        self.L.o_gen_switch_key(self.h, ctypes.byref(rng), _p(sk), galois_elt, _p(key))
        return key

    def gen_switch_key_new_old(self, rng, new_sk, old_sk):
        key = np.zeros(self.switch_key_digits() * 2 * self.Qp * self.n, dtype=np.uint64)
        self.L.o_gen_switch_key_new_old(self.h, ctypes.byref(rng), _p(new_sk), _p(old_sk), _p(key))
        return key

    def ckks_encrypt(self, rng, pk, plain):
        ct = np.zeros(2 * self.Q * self.n, dtype=np.uint64)
        self.L.o_ckks_encrypt(self.h, ctypes.byref(rng), _p(pk), _p(np.ascontiguousarray(plain, dtype=np.uint64)),
                              _p(ct))
        return ct

    def ckks_decrypt(self, ct, sk, depth=0):
        plain = np.zeros((self.Q - depth) * self.n, dtype=np.uint64)
        self.L.o_ckks_decrypt(self.h, _p(ct), _p(sk), depth, _p(plain))
        return plain

    def bfv_encrypt(self, rng, pk, plain):
        ct = np.zeros(2 * self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_encrypt(self.h, ctypes.byref(rng), _p(pk), _p(np.ascontiguousarray(plain, dtype=np.uint64)),
                             _p(ct))
        return ct

    def bfv_decrypt(self, ct, sk):
        plain = np.zeros(self.n, dtype=np.uint64)
        self.L.o_bfv_decrypt(self.h, _p(ct), _p(sk), _p(plain))
        return plain

    def bfv_encode(self, message):
        m = np.ascontiguousarray(message, dtype=np.int64)
        plain = np.zeros(self.n, dtype=np.uint64)
        self.L.o_bfv_encode(self.h, _p(m), len(m), _p(plain))
        return plain

    def bfv_decode(self, plain):
        out = np.zeros(self.n, dtype=np.uint64)
        self.L.o_bfv_decode(self.h, _p(np.ascontiguousarray(plain, dtype=np.uint64)), _p(out))
        return out

    def ckks_encode(self, message, scale):
        m = np.ascontiguousarray(message, dtype=np.float64)
        plain = np.zeros(self.Q * self.n, dtype=np.uint64)
        self.L.o_ckks_encode(self.h, _p(m), len(m), float(scale), _p(plain))
        return plain

    def ckks_decode(self, plain, scale, depth=0):
        out = np.zeros(self.n // 2, dtype=np.float64)
        self.L.o_ckks_decode(self.h, _p(np.ascontiguousarray(plain, dtype=np.uint64)), depth, float(scale), _p(out))
        return out

    def bfv_plain_to_ntt(self, plain):
        out = np.zeros(self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_plain_to_ntt(self.h, _p(np.ascontiguousarray(plain, dtype=np.uint64)), _p(out))
        return out

    def bfv_multiply_plain(self, ct, plain):
        out = np.zeros(2 * self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_multiply_plain(self.h, _p(np.ascontiguousarray(ct, dtype=np.uint64)),
                                    _p(np.ascontiguousarray(plain, dtype=np.uint64)), _p(out))
        return out

    def negacyclic_shift(self, ct, shift, limbs, parts=2):
        out = np.zeros(parts * limbs * self.n, dtype=np.uint64)
        self.L.o_negacyclic_shift(self.h, _p(np.ascontiguousarray(ct, dtype=np.uint64)), _p(out), shift, limbs, parts)
        return out

    def ckks_constant_op(self, op, ct, value, limbs, parts=2):
        out = np.zeros(parts * limbs * self.n, dtype=np.uint64)
        self.L.o_ckks_constant_op(self.h, op, _p(np.ascontiguousarray(ct, dtype=np.uint64)), float(value), _p(out),
                                  limbs, parts)
        return out

    def ckks_gaussian_integer_op(self, op, ct, re, im, limbs, parts=2):
        out = np.zeros(parts * limbs * self.n, dtype=np.uint64)
        self.L.o_ckks_gaussian_integer_op(self.h, op, _p(np.ascontiguousarray(ct, dtype=np.uint64)), float(re), float(im),
                                          _p(out), limbs, parts)
        return out

    def ckks_mult_i(self, ct, limbs, parts=2, divide=False):
        out = np.zeros(parts * limbs * self.n, dtype=np.uint64)
        self.L.o_ckks_mult_i(self.h, _p(np.ascontiguousarray(ct, dtype=np.uint64)), _p(out), limbs, parts, int(divide))
        return out

    # mode: 0 real slots, 1 complex slots, 2 coefficients, 3 one scalar in every slot
    def ckks_encode_ex(self, mode, message, scale):
        m = np.ascontiguousarray(message, dtype=np.complex128 if mode == 1 else np.float64).reshape(-1)
        plain = np.zeros(self.Q * self.n, dtype=np.uint64)
        self.L.o_ckks_encode_ex(self.h, mode, _p(m.view(np.float64)), len(m), float(scale), _p(plain))
        return plain

    def ckks_decode_ex(self, mode, plain, scale, depth=0):
        out = np.zeros(self.n if mode else self.n // 2, dtype=np.float64)
        self.L.o_ckks_decode_ex(self.h, mode, _p(np.ascontiguousarray(plain, dtype=np.uint64)), depth, float(scale),
                                _p(out))
        return out.view(np.complex128) if mode == 1 else out

    # key-switching method II (P_size > 1)
    def ckks_relinearize_II(self, ct3, key, depth=0):
        self.L.o_ckks_relinearize_II(self.h, _p(ct3), _p(key), depth)
        return ct3

    def ckks_apply_galois_II(self, ct, key, galois_elt, depth=0):
        l = self.Q - depth
        out = np.zeros(2 * l * self.n, dtype=np.uint64)
        self.L.o_ckks_apply_galois_II(self.h, _p(ct), _p(out), _p(key), galois_elt, depth)
        return out

    def bfv_relinearize_II(self, ct3, key):
        self.L.o_bfv_relinearize_II(self.h, _p(ct3), _p(key))
        return ct3

    def bfv_apply_galois_II(self, ct, key, galois_elt):
        out = np.zeros(2 * self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_apply_galois_II(self.h, _p(ct), _p(out), _p(key), galois_elt)
        return out

    def bfv_multiply(self, ct1, ct2):
        out = np.zeros(3 * self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_multiply(self.h, _p(ct1), _p(ct2), _p(out))
        return out

    def bfv_relinearize(self, ct3, key):
        self.L.o_bfv_relinearize(self.h, _p(ct3), _p(key))
        return ct3

    def bfv_apply_galois(self, ct, key, galois_elt):
        out = np.zeros(2 * self.Q * self.n, dtype=np.uint64)
        self.L.o_bfv_apply_galois(self.h, _p(ct), _p(out), _p(key), galois_elt)
        return out


# gate id -> (encoded numerator/denominator sign, s1, s2, m)  (tfhe/operator.cu:24-198)
TFHE_GATES = {0: (+1, 8, -1, -1, 1), 1: (-1, 8, 1, 1, 1), 2: (-1, 8, -1, 1, 1), 3: (-1, 8, -1, -1, 1),
              4: (+1, 8, 1, 1, 1), 5: (-1, 4, -1, -1, 2), 6: (+1, 4, 1, 1, 2)}


class OracleTfhe:
    """CPU oracle of the TFHE gate path (o_tfhe.c)."""

    n, N, k, l, ks_length, ks_base = 512, 1024, 1, 2, 8, 4

    def __init__(self):
        self.L = lib()
        self.h = self.L.o_tfhe_create()
        self.prime = int(self.L.o_tfhe_prime(self.h))
        self.mu = int(self.L.o_tfhe_encode_to_torus32(1, 8))

    def __del__(self):
        try:
            self.L.o_tfhe_free(self.h)
        except Exception:
            pass

    def to_ntt(self, poly):
        out = np.zeros(self.N, dtype=np.uint64)
        self.L.o_tfhe_to_ntt(self.h, _p(np.ascontiguousarray(poly, dtype=np.int32)), _p(out))
        return out

    def polymul(self, a, s):
        out = np.zeros(self.N, dtype=np.int32)
        self.L.o_tfhe_polymul(self.h, _p(np.ascontiguousarray(a, dtype=np.int32)),
                              _p(np.ascontiguousarray(s, dtype=np.int32)), _p(out))
        return out

    # front end (o_keygen.c); rng = ORng(seed)
    def gen_secret(self, rng):
        lwe = np.zeros(self.n, dtype=np.int32)
        tlwe = np.zeros(self.N * self.k, dtype=np.int32)
        self.L.o_tfhe_gen_secret(ctypes.byref(rng), _p(lwe), _p(tlwe))
        return lwe, tlwe

    def gen_bootkey(self, rng, lwe, tlwe):
        bk = np.zeros(self.n * 2 * self.l * 2 * self.N, dtype=np.uint64)
        rows = self.N * self.ks_length * (self.ks_base - 1)
        ks_a = np.zeros(rows * self.n, dtype=np.int32)
        ks_b = np.zeros(rows, dtype=np.int32)
        self.L.o_tfhe_gen_bootkey(self.h, ctypes.byref(rng), _p(lwe), _p(tlwe), _p(bk), _p(ks_a), _p(ks_b))
        return bk, ks_a, ks_b

    def encrypt(self, rng, lwe, messages):
        m = np.ascontiguousarray(messages, dtype=np.int32)
        a = np.zeros(len(m) * self.n, dtype=np.int32)
        b = np.zeros(len(m), dtype=np.int32)
        self.L.o_tfhe_encrypt(ctypes.byref(rng), _p(lwe), _p(m), len(m), _p(a), _p(b))
        return a, b

    def phase(self, lwe, a, b):
        out = np.zeros(len(b), dtype=np.int32)
        self.L.o_tfhe_phase(_p(lwe), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), len(b), _p(out))
        return out

    def gate_pre(self, gate, a1, b1, a2, b2):
        sign, den, s1, s2, m = TFHE_GATES[gate]
        enc = sign * int(self.L.o_tfhe_encode_to_torus32(1, den))
        shape = b1.shape[0]
        oa = np.zeros(shape * self.n, dtype=np.int32)
        ob = np.zeros(shape, dtype=np.int32)
        self.L.o_tfhe_gate_pre(_p(oa), _p(ob), _p(a1), _p(b1), _p(a2), _p(b2), enc, s1, s2, m, self.n, shape)
        return oa, ob

    def bootstrapping(self, a, b, boot_key):
        shape = b.shape[0]
        oa = np.zeros(shape * self.k * self.N, dtype=np.int32)
        ob = np.zeros(shape, dtype=np.int32)
        self.L.o_tfhe_bootstrapping(self.h, _p(a), _p(b), _p(boot_key), _p(oa), _p(ob), self.mu, shape)
        return oa, ob

    def key_switching(self, a, b, ks_a, ks_b):
        shape = b.shape[0]
        oa = np.zeros(shape * self.n, dtype=np.int32)
        ob = np.zeros(shape, dtype=np.int32)
        self.L.o_tfhe_key_switching(self.h, _p(a), _p(b), _p(oa), _p(ob), _p(ks_a), _p(ks_b), shape)
        return oa, ob

    def gate(self, gate, a1, b1, a2, b2, boot_key, ks_a, ks_b):
        ta, tb = self.gate_pre(gate, a1, b1, a2, b2)
        ea, eb = self.bootstrapping(ta, tb, boot_key)
        return self.key_switching(ea, eb, ks_a, ks_b)

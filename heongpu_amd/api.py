"""Thin Python handle over the C ABI (include/hegpu.h) for tests and bench.

PyTorch is used only for device memory and streams; all arithmetic happens in
libhegpu.so.  Polynomial data is held in torch.int64 tensors that carry the
uint64 bit patterns (torch has no full uint64 support); `to_device`/`to_host`
convert from/to numpy uint64.
"""
import contextlib
import ctypes

import numpy as np

from . import _lib

BFV, CKKS = 1, 2
SEC_NONE, SEC_128, SEC_192, SEC_256 = 0, 128, 192, 256
TABLES_QP, TABLES_Q_BSK = 0, 1
OP_CKKS_RELIN, OP_CKKS_RESCALE, OP_CKKS_GALOIS, OP_BFV_MULTIPLY, OP_BFV_RELIN, OP_BFV_GALOIS = 1, 2, 3, 4, 5, 6
OP_CKKS_ROTATE_HOISTED = 17
OP_KEYGEN_SECRET, OP_KEYGEN_PUBLIC, OP_KEYGEN_SWITCH, OP_CKKS_ENCRYPT, OP_BFV_ENCRYPT, OP_BFV_DECRYPT, OP_BFV_DECODE = 7, 8, 9, 10, 11, 12, 13
OP_CKKS_ENCODE, OP_CKKS_DECODE = 14, 15
OP_BFV_MULTIPLY_PLAIN = 16
TABLES_PLAIN = 2

E_INVALID, E_LOGIC, E_RUNTIME, E_NODEVICE = 10001, 10002, 10003, 10004


class HEError(RuntimeError):
    """Error returned by the C ABI.  `code` distinguishes the reference's
    exception classes: E_INVALID = std::invalid_argument, E_LOGIC =
    std::logic_error, E_RUNTIME = std::runtime_error; other = hipError_t."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise HEError(rc, _lib.load().hegpu_last_error().decode())


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def to_device(arr, device="cuda"):
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    return t.detach().cpu().numpy().view(np.uint64)


_default_options = {}


@contextlib.contextmanager
def default_options(**options):
    """Options (hegpu_context_set_option) applied to every Context CREATED inside the block -- a convenience of
    this ctypes layer for tests and measurements; nothing is written to the environment."""
    old = dict(_default_options)
    _default_options.update(options)
    try:
        yield
    finally:
        _default_options.clear()
        _default_options.update(old)


BCAST_FLAT, BCAST_TREE, BCAST_STAGED, BCAST_SAME_DEVICE = 1, 2, 0x100, 0x200


def broadcast_path_name(path):
    """words for the value of hegpu_last_broadcast_path / *path_out (include/hegpu.h)"""
    if not path:
        return "none (one buffer)"
    shape = {BCAST_FLAT: "flat fan-out from device 0 (every peer directly reachable: one hop)",
             BCAST_TREE: "binomial tree in 32 MiB chunks (some peer not directly reachable from device 0)"}[path & 0xff]
    if path & BCAST_STAGED:
        shape += ", at least one edge WITHOUT peer access (staged through host memory by the runtime)"
    if path & BCAST_SAME_DEVICE:
        shape += ", all buffers on ONE device (functional run)"
    return shape


def broadcast_key(contexts, keys, elems, streams=None):
    """hegpu_broadcast_key: keys[0] (a tensor on contexts[0]'s device) -> keys[i] on contexts[i]'s device.
    Returns the path taken (BCAST_* flags)."""
    n = len(contexts)
    lib = _lib.load()
    hs = (ctypes.c_void_p * n)(*[c._h for c in contexts])
    ks = (ctypes.c_void_p * n)(*[_ptr(k) for k in keys])
    ss = (ctypes.c_void_p * n)(*[s for s in streams]) if streams is not None else None
    _check(lib.hegpu_broadcast_key(hs, n, ks, int(elems), ss))
    return int(lib.hegpu_last_broadcast_path())


def broadcast_bytes(devices, bufs, nbytes, streams=None):
    """hegpu_broadcast_bytes: bufs[0] on devices[0] -> bufs[i] on devices[i] (tensors or addresses).  Returns the path."""
    n = len(devices)
    lib = _lib.load()
    ds = (ctypes.c_int * n)(*[int(d) for d in devices])
    bs = (ctypes.c_void_p * n)(*[_ptr(b) for b in bufs])
    ss = (ctypes.c_void_p * n)(*[s for s in streams]) if streams is not None else None
    path = ctypes.c_int(0)
    _check(lib.hegpu_broadcast_bytes(ds, n, bs, int(nbytes), ss, ctypes.byref(path)))
    return int(path.value)


class Context:
    """HEContext<BFV|CKKS>: parameter set + device tables."""

    def __init__(self, handle):
        self._lib = _lib.load()
        self._h = handle
        for k, v in _default_options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        _check(self._lib.hegpu_context_set_option(self._h, name.encode(), int(value)))

    def clone(self):
        """hegpu_context_clone: the same parameter set and options, not uploaded (one context per device)."""
        h = ctypes.c_void_p()
        _check(self._lib.hegpu_context_clone(self._h, ctypes.byref(h)))
        saved = dict(_default_options)
        _default_options.clear()  # the clone carries the source's options
        try:
            return Context(h)
        finally:
            _default_options.update(saved)

    def upload_device(self, device):
        _check(self._lib.hegpu_context_upload_device(self._h, int(device)))

    @property
    def device(self):
        return self._lib.hegpu_context_device(self._h)

    def get_option(self, name):
        v = ctypes.c_int()
        _check(self._lib.hegpu_context_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    # -- constructors (mirror set_coeff_modulus_* of the reference) --
    @classmethod
    def from_bit_sizes(cls, scheme, n, log_q, log_p, plain_modulus=0, sec=SEC_128):
        lib = _lib.load()
        q = (ctypes.c_int * len(log_q))(*log_q)
        p = (ctypes.c_int * max(len(log_p), 1))(*log_p)
        h = ctypes.c_void_p()
        _check(lib.hegpu_context_create(scheme, n, q, len(log_q), p, len(log_p), plain_modulus, sec,
                                        ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_default(cls, scheme, n, p_count=1, plain_modulus=0, sec=SEC_128):
        lib = _lib.load()
        h = ctypes.c_void_p()
        _check(lib.hegpu_context_create_default(scheme, n, p_count, plain_modulus, sec, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_primes(cls, scheme, n, primes, q_count, p_count, plain_modulus=0):
        lib = _lib.load()
        arr = (ctypes.c_uint64 * len(primes))(*[int(x) for x in primes])
        h = ctypes.c_void_p()
        _check(lib.hegpu_context_create_from_primes(scheme, n, arr, q_count, p_count, plain_modulus,
                                                    ctypes.byref(h)))
        return cls(h)

    def close(self):
        if self._h:
            self._lib.hegpu_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties --
    def int(self, name):
        return int(self._lib.hegpu_context_int(self._h, name.encode()))

    @property
    def n_power(self):
        return self.int("n_power")

    @property
    def n(self):
        return 1 << self.n_power

    @property
    def Q_size(self):
        return self.int("Q_size")

    @property
    def P_size(self):
        return self.int("P_size")

    @property
    def Q_prime_size(self):
        return self.int("Q_prime_size")

    @property
    def bsk_modulus(self):
        return self.int("bsk_modulus")

    def table(self, name):
        cnt = self._lib.hegpu_context_get(self._h, name.encode(), None, 0)
        if cnt < 0:
            raise KeyError(name)
        out = np.zeros(cnt, dtype=np.uint64)
        got = self._lib.hegpu_context_get(self._h, name.encode(), out.ctypes.data, cnt)
        assert got == cnt
        return out

    def upload(self):
        _check(self._lib.hegpu_context_upload(self._h))

    def device_ptr(self, name):
        return self._lib.hegpu_context_device_ptr(self._h, name.encode())

    def workspace_bytes(self, op, depth, batch):
        return int(self._lib.hegpu_workspace_bytes(self._h, op, depth, batch))

    def workspace(self, op, depth, batch, device="cuda"):
        import torch
        nbytes = self.workspace_bytes(op, depth, batch)
        return torch.empty(max(nbytes // 8, 1), dtype=torch.int64, device=device)

    # -- NTT seam --
    def ntt(self, src, dst, inverse, batch, mod_count, mod_offset=0, mod_order=None, poly_order=None,
            table_set=TABLES_QP, stream=None):
        _check(self._lib.hegpu_ntt(self._h, table_set, _ptr(src), _ptr(dst), int(inverse), batch, mod_count,
                                   mod_offset, _ptr(mod_order), _ptr(poly_order),
                                   stream if stream is not None else _stream()))

    # -- kernels --
    def addition(self, a, b, out, limbs, parts, batch=1, op=0, stream=None):
        _check(self._lib.hegpu_addition(self._h, _ptr(a), _ptr(b), _ptr(out), limbs, parts, batch, op,
                                        stream if stream is not None else _stream()))

    def cross_multiplication(self, in1, s1, in2, s2, out, so, decomp_size, batch=1, table_set=TABLES_QP,
                             stream=None):
        _check(self._lib.hegpu_cross_multiplication(self._h, table_set, _ptr(in1), s1, _ptr(in2), s2, _ptr(out),
                                                    so, decomp_size, batch,
                                                    stream if stream is not None else _stream()))

    def cipher_broadcast(self, src, s_in, out, s_out, digits, nmods, split, level, batch=1, stream=None):
        _check(self._lib.hegpu_cipher_broadcast(self._h, _ptr(src), s_in, _ptr(out), s_out, digits, nmods, split,
                                                level, batch, stream if stream is not None else _stream()))

    def keyswitch_multiply_accumulate(self, src, s_in, key, out, s_out, digits, nmods, key_limbs, split, level,
                                      batch=1, stream=None):
        _check(self._lib.hegpu_keyswitch_multiply_accumulate(self._h, _ptr(src), s_in, _ptr(key), _ptr(out), s_out,
                                                             digits, nmods, key_limbs, split, level, batch,
                                                             stream if stream is not None else _stream()))

    def base_conversion_DtoQtilde(self, src, s_in, out, s_out, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_base_conversion_DtoQtilde(self._h, _ptr(src), s_in, _ptr(out), s_out, depth, batch,
                                                         stream if stream is not None else _stream()))

    def divide_round_lastq(self, src, s_in, ct, s_ct, out, s_out, switchkey=0, batch=1, stream=None):
        _check(self._lib.hegpu_divide_round_lastq(self._h, _ptr(src), s_in, _ptr(ct), s_ct, _ptr(out), s_out,
                                                  switchkey, batch, stream if stream is not None else _stream()))

    def divide_round_lastq_permute(self, src, s_in, in2, s_in2, out, s_out, galois_elt, depth=0, batch=1,
                                   stream=None):
        _check(self._lib.hegpu_divide_round_lastq_permute(self._h, _ptr(src), s_in, _ptr(in2), s_in2, _ptr(out),
                                                          s_out, galois_elt, depth, batch,
                                                          stream if stream is not None else _stream()))

    def divide_round_lastq_leveled_stage_one(self, src, s_in, out, s_out, rescale=0, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_divide_round_lastq_leveled_stage_one(self._h, _ptr(src), s_in, _ptr(out), s_out, rescale,
                                                                    depth, batch,
                                                                    stream if stream is not None else _stream()))

    def divide_round_lastq_leveled_stage_two(self, last, s_last, src, s_in, ct, s_ct, out, s_out, switchkey=0, depth=0,
                                             batch=1, stream=None):
        _check(self._lib.hegpu_divide_round_lastq_leveled_stage_two(self._h, _ptr(last), s_last, _ptr(src), s_in,
                                                                    _ptr(ct), s_ct, _ptr(out), s_out, switchkey, depth,
                                                                    batch, stream if stream is not None else _stream()))

    def move_cipher_leveled(self, src, s_in, out, s_out, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_move_cipher_leveled(self._h, _ptr(src), s_in, _ptr(out), s_out, depth, batch,
                                                   stream if stream is not None else _stream()))

    def divide_round_lastq_rescale(self, last, s_last, src, s_in, out, s_out, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_divide_round_lastq_rescale(self._h, _ptr(last), s_last, _ptr(src), s_in, _ptr(out), s_out,
                                                          depth, batch, stream if stream is not None else _stream()))

    def divide_round_lastq_extended(self, src, s_in, ct, s_ct, out, s_out, mode=0, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_divide_round_lastq_extended(self._h, _ptr(src), s_in, _ptr(ct) if ct is not None else None,
                                                           s_ct, _ptr(out), s_out, mode, depth, batch,
                                                           stream if stream is not None else _stream()))

    def fast_convertion(self, in1, s1, in2, s2, out, so, batch=1, stream=None):
        _check(self._lib.hegpu_fast_convertion(self._h, _ptr(in1), s1, _ptr(in2), s2, _ptr(out), so, batch,
                                               stream if stream is not None else _stream()))

    def fast_floor(self, src, si, out, so, batch=1, stream=None):
        _check(self._lib.hegpu_fast_floor(self._h, _ptr(src), si, _ptr(out), so, batch,
                                          stream if stream is not None else _stream()))

    # -- operators (HEArithmeticOperator) --
    def ckks_multiply(self, ct1, s1, ct2, s2, out, so, depth=0, batch=1, stream=None):
        _check(self._lib.hegpu_ckks_multiply(self._h, _ptr(ct1), s1, _ptr(ct2), s2, _ptr(out), so, depth, batch,
                                             stream if stream is not None else _stream()))

    def ckks_relinearize_inplace(self, ct, cs, key, depth, batch, ws, stream=None):
        _check(self._lib.hegpu_ckks_relinearize_inplace(self._h, _ptr(ct), cs, _ptr(key), depth, batch, _ptr(ws),
                                                        ws.numel() * 8,
                                                        stream if stream is not None else _stream()))

    def probe_ckks_relinearize(self, ct, cs, key, depth, batch, ws, phases, stream=None):
        """measurement seam: only the launches selected by `phases` (include/hegpu.h)"""
        _check(self._lib.hegpu_probe_ckks_relinearize(self._h, _ptr(ct), cs, _ptr(key), depth, batch, _ptr(ws),
                                                      ws.numel() * 8, phases,
                                                      stream if stream is not None else _stream()))

    def ckks_rescale_inplace(self, ct, cs, depth, batch, ws, stream=None):
        _check(self._lib.hegpu_ckks_rescale_inplace(self._h, _ptr(ct), cs, depth, batch, _ptr(ws), ws.numel() * 8,
                                                    stream if stream is not None else _stream()))

    def ckks_apply_galois(self, ct, cs, out, so, key, galois_elt, depth, batch, ws, stream=None):
        _check(self._lib.hegpu_ckks_apply_galois(self._h, _ptr(ct), cs, _ptr(out), so, _ptr(key), galois_elt, depth,
                                                 batch, _ptr(ws), ws.numel() * 8,
                                                 stream if stream is not None else _stream()))

    def ckks_rotate_hoisted(self, ct, cs, out, so, keys, galois_elts, depth, batch, ws, stream=None):
        """fast_single_hoisting_rotation_ckks: keys = list of device tensors (None where galois_elts[i] == 0)"""
        cnt = len(galois_elts)
        kp = (ctypes.c_void_p * cnt)(*[(_ptr(k) if k is not None else None) for k in keys])
        ge = (ctypes.c_int * cnt)(*[int(g) for g in galois_elts])
        _check(self._lib.hegpu_ckks_rotate_hoisted(self._h, _ptr(ct), cs, _ptr(out), so, kp, ge, cnt, depth, batch,
                                                   _ptr(ws), ws.numel() * 8,
                                                   stream if stream is not None else _stream()))

    def bfv_multiply(self, ct1, s1, ct2, s2, out, so, batch, ws, stream=None):
        _check(self._lib.hegpu_bfv_multiply(self._h, _ptr(ct1), s1, _ptr(ct2), s2, _ptr(out), so, batch, _ptr(ws),
                                            ws.numel() * 8, stream if stream is not None else _stream()))

    def bfv_relinearize_inplace(self, ct, cs, key, batch, ws, stream=None):
        _check(self._lib.hegpu_bfv_relinearize_inplace(self._h, _ptr(ct), cs, _ptr(key), batch, _ptr(ws),
                                                       ws.numel() * 8,
                                                       stream if stream is not None else _stream()))

    def bfv_apply_galois(self, ct, cs, out, so, key, galois_elt, batch, ws, stream=None):
        _check(self._lib.hegpu_bfv_apply_galois(self._h, _ptr(ct), cs, _ptr(out), so, _ptr(key), galois_elt, batch,
                                                _ptr(ws), ws.numel() * 8,
                                                stream if stream is not None else _stream()))


    # ---- key generation / encryption / decryption (method I keys)
    def _kg_ws(self, op):
        return self.workspace(op, 0, 1)

    def generate_secret_key(self, rng, hamming_weight=None, stream=None):
        import torch
        n, Qp = self.n, self.Q_prime_size
        sk = torch.empty(Qp * n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_KEYGEN_SECRET)
        hw = n // 2 if hamming_weight is None else hamming_weight  # secretkey.cu:23
        _check(self._lib.hegpu_generate_secret_key(self._h, rng._h, hw, _ptr(sk), _ptr(ws),
                                                   ws.numel() * ws.element_size(),
                                                   stream if stream is not None else _stream()))
        return sk

    def generate_public_key(self, rng, sk, stream=None):
        import torch
        pk = torch.empty(2 * self.Q_prime_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_KEYGEN_PUBLIC)
        _check(self._lib.hegpu_generate_public_key(self._h, rng._h, _ptr(sk), _ptr(pk), _ptr(ws),
                                                   ws.numel() * ws.element_size(),
                                                   stream if stream is not None else _stream()))
        return pk

    def switch_key_digits(self):
        """digits of a key-switching key: Q (method I) or the depth-0 partition of method II
        (digits of 2 primes for BFV, of P_size primes for CKKS; contextpool.cpp:161-438)"""
        if self.P_size == 1:
            return self.Q_size
        m = 2 if self.int("scheme") == BFV else self.P_size
        return -(-self.Q_size // m)

    def generate_relin_key(self, rng, sk, stream=None):
        import torch
        rk = torch.empty(self.switch_key_digits() * 2 * self.Q_prime_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_KEYGEN_SWITCH)
        _check(self._lib.hegpu_generate_relin_key(self._h, rng._h, _ptr(sk), _ptr(rk), _ptr(ws),
                                                  ws.numel() * ws.element_size(),
                                                  stream if stream is not None else _stream()))
        return rk

    def generate_switch_key(self, rng, new_sk, old_sk, stream=None):
        """HEKeyGenerator::generate_switch_key (ckks/keygenerator.cu:996-1095): key under new_sk carrying old_sk."""
        import torch
        swk = torch.empty(self.switch_key_digits() * 2 * self.Q_prime_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_KEYGEN_SWITCH)
        _check(self._lib.hegpu_generate_switch_key(self._h, rng._h, _ptr(new_sk), _ptr(old_sk), _ptr(swk), _ptr(ws),
                                                   ws.numel() * ws.element_size(),
                                                   stream if stream is not None else _stream()))
        return swk

    def generate_galois_key(self, rng, sk, galois_elt, stream=None):
        import torch
        gk = torch.empty(self.switch_key_digits() * 2 * self.Q_prime_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_KEYGEN_SWITCH)
        _check(self._lib.hegpu_generate_galois_key(self._h, rng._h, _ptr(sk), galois_elt, _ptr(gk), _ptr(ws),
                                                   ws.numel() * ws.element_size(),
                                                   stream if stream is not None else _stream()))
        return gk

    def ckks_encrypt(self, rng, pk, plain, stream=None):
        import torch
        ct = torch.empty(2 * self.Q_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_CKKS_ENCRYPT)
        _check(self._lib.hegpu_ckks_encrypt(self._h, rng._h, _ptr(pk), _ptr(plain), _ptr(ct), _ptr(ws),
                                            ws.numel() * ws.element_size(),
                                            stream if stream is not None else _stream()))
        return ct

    def bfv_encrypt(self, rng, pk, plain, stream=None):
        import torch
        ct = torch.empty(2 * self.Q_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_BFV_ENCRYPT)
        _check(self._lib.hegpu_bfv_encrypt(self._h, rng._h, _ptr(pk), _ptr(plain), _ptr(ct), _ptr(ws),
                                           ws.numel() * ws.element_size(),
                                           stream if stream is not None else _stream()))
        return ct

    def bfv_decrypt(self, ct, sk, stream=None):
        import torch
        plain = torch.empty(self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_BFV_DECRYPT)
        _check(self._lib.hegpu_bfv_decrypt(self._h, _ptr(ct), _ptr(sk), _ptr(plain), _ptr(ws),
                                           ws.numel() * ws.element_size(),
                                           stream if stream is not None else _stream()))
        return plain

    def bfv_encode(self, message, stream=None):
        """message: device int64 tensor with at most N entries"""
        import torch
        plain = torch.empty(self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_bfv_encode(self._h, _ptr(message), message.numel(), _ptr(plain),
                                          stream if stream is not None else _stream()))
        return plain

    def bfv_decode(self, plain, stream=None):
        import torch
        out = torch.empty(self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_BFV_DECODE)
        _check(self._lib.hegpu_bfv_decode(self._h, _ptr(plain), _ptr(out), _ptr(ws), ws.numel() * ws.element_size(),
                                          stream if stream is not None else _stream()))
        return out

    def ckks_encode(self, message, scale, stream=None):
        """message: device float64 tensor with at most N/2 entries"""
        import torch
        plain = torch.empty(self.Q_size * self.n, dtype=torch.int64, device="cuda")
        ws = self._kg_ws(OP_CKKS_ENCODE)
        _check(self._lib.hegpu_ckks_encode(self._h, _ptr(message), message.numel(), float(scale), _ptr(plain),
                                           _ptr(ws), ws.numel() * ws.element_size(),
                                           stream if stream is not None else _stream()))
        return plain

    def ckks_decode(self, plain, scale, depth=0, stream=None):
        import torch
        out = torch.empty(self.n // 2, dtype=torch.float64, device="cuda")
        ws = self.workspace(OP_CKKS_DECODE, depth, 1)
        _check(self._lib.hegpu_ckks_decode(self._h, _ptr(plain), depth, float(scale), _ptr(out), _ptr(ws),
                                           ws.numel() * ws.element_size(),
                                           stream if stream is not None else _stream()))
        return out

    def bfv_plain_to_ntt(self, plain, stream=None):
        import torch
        out = torch.empty(self.Q_size * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_bfv_plain_to_ntt(self._h, _ptr(plain), _ptr(out), stream if stream is not None else _stream()))
        return out

    def negacyclic_shift(self, ct, shift, limbs, parts=2, stream=None):
        import torch
        out = torch.empty(parts * limbs * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_negacyclic_shift(self._h, _ptr(ct), _ptr(out), shift, limbs, parts,
                                                stream if stream is not None else _stream()))
        return out

    def ckks_constant_op(self, op, ct, value, limbs, parts=2, out=None, stream=None):
        """op 0/1/2 = add / subtract / multiply by round(value) (value = constant * scale); NTT-domain ciphertext"""
        import torch
        if out is None:
            out = torch.empty(parts * limbs * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_ckks_constant_op(self._h, op, _ptr(ct), float(value), _ptr(out), limbs, parts,
                                                stream if stream is not None else _stream()))
        return out

    def ckks_gaussian_integer_op(self, op, ct, re, im, limbs, parts=2, out=None, stream=None):
        """op 0 / 1 = add / multiply by the constant round(re) + round(im) i in every slot"""
        import torch
        if out is None:
            out = torch.empty(parts * limbs * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_ckks_gaussian_integer_op(self._h, op, _ptr(ct), float(re), float(im), _ptr(out), limbs,
                                                        parts, stream if stream is not None else _stream()))
        return out

    def ckks_mult_i(self, ct, limbs, parts=2, divide=False, out=None, stream=None):
        import torch
        if out is None:
            out = torch.empty(parts * limbs * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_ckks_mult_i(self._h, _ptr(ct), _ptr(out), limbs, parts, int(divide),
                                           stream if stream is not None else _stream()))
        return out

    def ckks_encode_ex(self, mode, message, scale, stream=None):
        """The other encodings of HEEncoder<CKKS> (ckks/encoder.cu:222-446).  mode 1: complex128 device
        tensor (<= N/2) into the slots; mode 2: float64 device tensor (<= N) as polynomial coefficients;
        mode 3: `message` is one Python number placed in every slot."""
        import torch
        st = stream if stream is not None else _stream()
        plain = torch.empty(self.Q_size * self.n, dtype=torch.int64, device="cuda")
        if mode == 3:
            _check(self._lib.hegpu_ckks_encode_scalar(self._h, float(message), float(scale), _ptr(plain), st))
        elif mode == 2:
            _check(self._lib.hegpu_ckks_encode_coeff(self._h, _ptr(message), message.numel(), float(scale), _ptr(plain), st))
        else:
            assert mode == 1 and message.dtype == torch.complex128
            ws = self._kg_ws(OP_CKKS_ENCODE)
            _check(self._lib.hegpu_ckks_encode_complex(self._h, _ptr(message), message.numel(), float(scale),
                                                       _ptr(plain), _ptr(ws), ws.numel() * ws.element_size(), st))
        return plain

    def ckks_decode_ex(self, mode, plain, scale, depth=0, stream=None):
        """mode 1: the N/2 complex slots; mode 2: the N polynomial coefficients"""
        import torch
        st = stream if stream is not None else _stream()
        ws = self.workspace(OP_CKKS_DECODE, depth, 1)
        if mode == 1:
            out = torch.empty(self.n // 2, dtype=torch.complex128, device="cuda")
            fn = self._lib.hegpu_ckks_decode_complex
        else:
            out = torch.empty(self.n, dtype=torch.float64, device="cuda")
            fn = self._lib.hegpu_ckks_decode_coeff
        _check(fn(self._h, _ptr(plain), depth, float(scale), _ptr(out), _ptr(ws), ws.numel() * ws.element_size(), st))
        return out

    def ckks_decrypt(self, ct, sk, depth=0, stream=None):
        import torch
        plain = torch.empty((self.Q_size - depth) * self.n, dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_ckks_decrypt(self._h, _ptr(ct), _ptr(sk), depth, _ptr(plain),
                                            stream if stream is not None else _stream()))
        return plain



GATE_NAND, GATE_AND, GATE_AND_FIRST_NOT, GATE_NOR, GATE_OR, GATE_XNOR, GATE_XOR, GATE_NOT = range(8)


class Rng:
    """The backend's DRBG handle (ChaCha20 counter-mode streams, csrc/drbg.hpp).  Rng(int): the
    reproducible 64-bit test seed; Rng(bytes of length 32): a full seed; Rng(None): OS entropy."""

    def __init__(self, seed=None):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        if seed is None:
            _check(self._lib.hegpu_rng_create_from_entropy(ctypes.byref(h)))
        elif isinstance(seed, (bytes, bytearray)):
            assert len(seed) == 32
            _check(self._lib.hegpu_rng_create_seeded(bytes(seed), ctypes.byref(h)))
        else:
            _check(self._lib.hegpu_rng_create(int(seed) & (2**64 - 1), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.hegpu_rng_destroy(self._h)
                self._h = None
        except Exception:
            pass


class TfheContext:
    """HEContext<TFHE> + HELogicOperator<TFHE> over the C ABI (fixed STD128 set)."""

    def __init__(self):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _check(self._lib.hegpu_tfhe_context_create(ctypes.byref(h)))
        self._h = h

    def set_option(self, name, value):
        _check(self._lib.hegpu_tfhe_context_set_option(self._h, name.encode(), int(value)))

    def close(self):
        if self._h:
            self._lib.hegpu_tfhe_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def int(self, name):
        return int(self._lib.hegpu_tfhe_context_int(self._h, name.encode()))

    @property
    def prime(self):
        return int(self._lib.hegpu_tfhe_prime(self._h))

    def prepare_bootkey(self, boot_key, stream=None):
        import torch
        out = torch.empty(self.int("prepared_bootkey_elems"), dtype=torch.int64, device=boot_key.device)
        _check(self._lib.hegpu_tfhe_prepare_bootkey(self._h, _ptr(boot_key), _ptr(out),
                                                    stream if stream is not None else _stream()))
        return out

    @staticmethod
    def prepared_is_fp64(prepared):
        """True when the prepared key uses the FP64 blind-rotate layout (real torus32 key)."""
        return int(prepared[0].item()) == 1

    def prepared_format(self, prepared, refresh=False):
        """1 = FP64 layout, 0 = integer layout: the buffer's header word, read after the device has drained
        (hegpu_tfhe_prepared_format; a query -- the blind rotate reads the word itself, on the device)"""
        f = self._lib.hegpu_tfhe_prepared_format(self._h, _ptr(prepared), 1 if refresh else 0)
        if f not in (0, 1):
            raise HEError(f, _lib.load().hegpu_last_error().decode())
        return f

    def gate_precompute(self, gate, out_a, out_b, a1, b1, a2, b2, shape, stream=None):
        _check(self._lib.hegpu_tfhe_gate_precompute(self._h, gate, _ptr(out_a), _ptr(out_b), _ptr(a1), _ptr(b1),
                                                    _ptr(a2), _ptr(b2), shape,
                                                    stream if stream is not None else _stream()))

    def bootstrapping(self, in_a, in_b, prepared_bk, out_a, out_b, shape, stream=None):
        _check(self._lib.hegpu_tfhe_bootstrapping(self._h, _ptr(in_a), _ptr(in_b), _ptr(prepared_bk), _ptr(out_a),
                                                  _ptr(out_b), shape, stream if stream is not None else _stream()))

    def status(self, stream=None):
        """drains the stream; raises HEError(E_INVALID) if a bootstrapping call was handed a buffer that is no prepared key"""
        _check(self._lib.hegpu_tfhe_status(self._h, stream if stream is not None else _stream()))

    def key_switching(self, in_a, in_b, out_a, out_b, ks_a, ks_b, shape, stream=None):
        _check(self._lib.hegpu_tfhe_key_switching(self._h, _ptr(in_a), _ptr(in_b), _ptr(out_a), _ptr(out_b),
                                                  _ptr(ks_a), _ptr(ks_b), shape,
                                                  stream if stream is not None else _stream()))

    # ---- front end: keys, bit encryption, decryption, MUX
    def generate_secret_key(self, rng, stream=None):
        import torch
        lwe = torch.empty(self.int("n"), dtype=torch.int32, device="cuda")
        tlwe = torch.empty(self.int("N") * self.int("k"), dtype=torch.int32, device="cuda")
        _check(self._lib.hegpu_tfhe_generate_secret_key(self._h, rng._h, _ptr(lwe), _ptr(tlwe),
                                                        stream if stream is not None else _stream()))
        return lwe, tlwe

    def generate_bootstrapping_key(self, rng, lwe, tlwe, stream=None):
        """returns (boot key in the reference layout, ks_a, ks_b)"""
        import torch
        bk = torch.empty(self.int("bootkey_elems"), dtype=torch.int64, device="cuda")
        ks_a = torch.empty(self.int("kskey_a_elems"), dtype=torch.int32, device="cuda")
        ks_b = torch.empty(self.int("kskey_b_elems"), dtype=torch.int32, device="cuda")
        ws = torch.empty(self.int("N"), dtype=torch.int64, device="cuda")
        _check(self._lib.hegpu_tfhe_generate_bootstrapping_key(self._h, rng._h, _ptr(lwe), _ptr(tlwe), _ptr(bk),
                                                               _ptr(ks_a), _ptr(ks_b), _ptr(ws), ws.numel() * 8,
                                                               stream if stream is not None else _stream()))
        return bk, ks_a, ks_b

    def encrypt(self, rng, lwe, messages, stream=None):
        import torch
        shape = messages.numel()
        a = torch.empty(shape * self.int("n"), dtype=torch.int32, device="cuda")
        b = torch.empty(shape, dtype=torch.int32, device="cuda")
        _check(self._lib.hegpu_tfhe_encrypt(self._h, rng._h, _ptr(lwe), _ptr(messages), shape, _ptr(a), _ptr(b),
                                            stream if stream is not None else _stream()))
        return a, b

    def decrypt_phase(self, lwe, a, b, stream=None):
        import torch
        out = torch.empty(b.numel(), dtype=torch.int32, device="cuda")
        _check(self._lib.hegpu_tfhe_decrypt_phase(self._h, _ptr(lwe), _ptr(a), _ptr(b), b.numel(), _ptr(out),
                                                  stream if stream is not None else _stream()))
        return out

    def mux(self, a1, b1, a2, b2, ca, cb, out_a, out_b, prepared_bk, ks_a, ks_b, shape, ws, stream=None):
        _check(self._lib.hegpu_tfhe_mux(self._h, _ptr(a1), _ptr(b1), _ptr(a2), _ptr(b2), _ptr(ca), _ptr(cb),
                                        _ptr(out_a), _ptr(out_b), _ptr(prepared_bk), _ptr(ks_a), _ptr(ks_b), shape,
                                        _ptr(ws), ws.numel() * ws.element_size(),
                                        stream if stream is not None else _stream()))

    def gate(self, gate, a1, b1, a2, b2, out_a, out_b, prepared_bk, ks_a, ks_b, shape, ws, stream=None):
        _check(self._lib.hegpu_tfhe_gate(self._h, gate, _ptr(a1), _ptr(b1), _ptr(a2), _ptr(b2), _ptr(out_a),
                                         _ptr(out_b), _ptr(prepared_bk), _ptr(ks_a), _ptr(ks_b), shape, _ptr(ws),
                                         ws.numel() * ws.element_size(),
                                         stream if stream is not None else _stream()))


def steps_to_galois_elt(steps, n, group_order):
    return int(_lib.load().hegpu_steps_to_galois_elt(steps, n, group_order))

"""wire.py -- an independent reader of the reference's serialization format (TEST INFRASTRUCTURE, like everything
under oracle/: only tests/ may import it).

Written from the reference's sources alone -- field order, widths, tags and framing as the `save` functions write
them -- so that blobs produced by the class layer of this repository (include/heongpu/heongpu.hpp) can be checked
against a second, unrelated implementation of the format instead of only against their own `load`:

  framing      src/include/heongpu/util/serializer.h:62-112 (serialize = zlib compress() of the save() stream;
               save_to_file = u64 size + that buffer), src/lib/util/serializer.cpp:22-50
  context      src/lib/host/ckks/context.cu:576-630, src/lib/host/bfv/context.cu:805-858 (+ plain_modulus_)
  secret key   src/lib/host/{ckks,bfv}/secretkey.cu save();  members include/heongpu/host/ckks/secretkey.cuh:316-324
  public key   .../publickey.cu save();  publickey.cuh:277-283
  plaintext    .../plaintext.cu save();  plaintext.cuh:233-240
  ciphertext   src/lib/host/ckks/ciphertext.cu:171-230, bfv/ciphertext.cu save();  ciphertext.cuh:333-344
  relin key    src/lib/host/ckks/evaluationkey.cu:103-147;  evaluationkey.cuh:400-411
  galois key   evaluationkey.cu:716-811;  evaluationkey.cuh:833-859
  switch key   evaluationkey.cu:1013-1050;  evaluationkey.cuh:982-989
  enums        include/heongpu/util/schemes.h:70-133 (all std::uint8_t), util/storagemanager.cuh:23-27

Everything is little-endian, written member by member with os.write(&member, sizeof(member)): enums 1 byte, bool 1
byte, int 4 bytes, double 8 bytes, Data64 8 bytes.  Modulus64 is GPU-NTT's Modulus<Data64> (unvendored; recalled as
{value, bit, mu} = 24 bytes): its size on the wire CANNOT be verified in this pipeline; the reader takes 24 bytes and
checks the recalled relation between the three words, which is all that can be said.
"""
import struct
import zlib

SCHEMES = {0: "none", 1: "bfv", 2: "ckks", 3: "bgv"}
SEC_LEVELS = {0: "none", 1: "sec128", 2: "sec192", 3: "sec256"}
KEYSWITCH = {0: "NONE", 1: "KEYSWITCHING_METHOD_I", 2: "KEYSWITCHING_METHOD_II"}
STORAGE = {1: "HOST", 2: "DEVICE"}
ENCODING = {0: "SLOT", 1: "COEFFICIENT"}


class Reader:
    def __init__(self, blob):
        self.b, self.o = memoryview(blob), 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def u8(self): return self.take("B")
    def boolean(self):
        v = self.take("B")
        if v not in (0, 1):
            raise ValueError("bool byte %d at offset %d" % (v, self.o - 1))
        return bool(v)
    def i32(self): return self.take("i")
    def u32(self): return self.take("I")
    def u64(self): return self.take("Q")
    def f64(self): return self.take("d")

    def u64s(self, count):
        v = self.b[self.o:self.o + 8 * count]
        if len(v) != 8 * count:
            raise ValueError("payload truncated: %d of %d bytes" % (len(v), 8 * count))
        self.o += 8 * count
        return bytes(v)

    def array(self, fmt, count):
        return list(struct.unpack_from("<%d%s" % (count, fmt), self.b, self._adv(struct.calcsize(fmt) * count)))

    def _adv(self, nbytes):
        o = self.o
        self.o += nbytes
        return o

    def modulus64(self):
        value, bit, mu = self.take("QQQ")
        return {"value": value, "bit": bit, "mu": mu}

    def done(self):
        if self.o != len(self.b):
            raise ValueError("%d trailing bytes" % (len(self.b) - self.o))


def modulus64_is_consistent(m):
    """GPU-NTT's Barrett record as recalled: bit = floor(log2 q) + 1, mu = floor(2^(2 bit + 1) / q)"""
    q = m["value"]
    return q > 1 and m["bit"] == q.bit_length() and m["mu"] == (1 << (2 * q.bit_length() + 1)) // q


def _enum(table, v, what):
    if v not in table:
        raise ValueError("invalid %s tag %d" % (what, v))
    return table[v]


def parse_context(blob):
    r = Reader(blob)
    d = {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "sec_level": _enum(SEC_LEVELS, r.u8(), "sec_level"),
         "keyswitching_type": _enum(KEYSWITCH, r.u8(), "keyswitching_type")}
    for name in ("n", "n_power", "coeff_modulus", "total_coeff_bit_count", "Q_prime_size", "Q_size", "P_size"):
        d[name] = r.i32()
    d["prime_vector"] = [r.modulus64() for _ in range(r.u32())]
    d["base_q"] = r.array("Q", r.u32())
    d["Qprime_mod_bit_sizes"] = r.array("i", r.u32())
    d["Q_mod_bit_sizes"] = r.array("i", r.u32())
    d["P_mod_bit_sizes"] = r.array("i", r.u32())
    if d["scheme"] == "bfv":
        d["plain_modulus"] = r.modulus64()
    r.done()
    return d


def parse_secretkey(blob):
    r = Reader(blob)
    d = {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "ring_size": r.i32(), "coeff_modulus_count": r.i32(),
         "n_power": r.i32(), "hamming_weight": r.i32(), "in_ntt_domain": r.boolean(), "generated": r.boolean(),
         "storage_type": _enum(STORAGE, r.u8(), "storage")}
    d["size"] = r.u32()
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def parse_publickey(blob):
    r = Reader(blob)
    d = {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "ring_size": r.i32(), "coeff_modulus_count": r.i32(),
         "in_ntt_domain": r.boolean(), "generated": r.boolean(), "storage_type": _enum(STORAGE, r.u8(), "storage")}
    d["size"] = r.u32()
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def parse_plaintext(blob):
    r = Reader(blob)
    d = {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "plain_size": r.i32()}
    if d["scheme"] == "ckks":
        d["depth"], d["scale"] = r.i32(), r.f64()
        d["in_ntt_domain"] = r.boolean()
        d["encoding"] = _enum(ENCODING, r.u8(), "encoding")
    else:
        d["in_ntt_domain"] = r.boolean()
    d["generated"] = r.boolean()
    d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
    d["size"] = r.i32()  # plain_size_ once more, as the length of the payload
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def parse_ciphertext(blob):
    r = Reader(blob)
    d = {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "ring_size": r.i32(), "coeff_modulus_count": r.i32(),
         "cipher_size": r.i32()}
    if d["scheme"] == "ckks":
        d["depth"] = r.i32()
        d["in_ntt_domain"] = r.boolean()
        d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
        d["scale"] = r.f64()
        d["encoding"] = _enum(ENCODING, r.u8(), "encoding")
        d["rescale_required"] = r.boolean()
    else:
        d["in_ntt_domain"] = r.boolean()
        d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
    d["relinearization_required"] = r.boolean()
    d["generated"] = r.boolean()
    d["size"] = r.u32()
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def _key_header(r):
    return {"scheme": _enum(SCHEMES, r.u8(), "scheme"), "key_type": _enum(KEYSWITCH, r.u8(), "key_type"),
            "ring_size": r.i32(), "Q_prime_size": r.i32(), "Q_size": r.i32(), "d": r.i32()}


def parse_relinkey(blob):
    r = Reader(blob)
    d = _key_header(r)
    d["d_tilda"], d["r_prime"] = r.i32(), r.i32()
    d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
    d["generated"] = r.boolean()
    d["size"] = r.u64()
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def parse_switchkey(blob):
    r = Reader(blob)
    d = _key_header(r)
    d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
    d["generated"] = r.boolean()
    d["size"] = r.u64()
    d["payload"] = r.u64s(d["size"])
    r.done()
    return d


def parse_galoiskey(blob):
    r = Reader(blob)
    d = _key_header(r)
    d["customized"] = r.boolean()
    d["group_order"] = r.i32()
    d["storage_type"] = _enum(STORAGE, r.u8(), "storage")
    d["generated"] = r.boolean()
    if d["customized"]:
        d["custom_galois_elt"] = r.array("I", r.u32())
    else:
        cnt = r.u32()
        pairs = r.array("i", 2 * cnt)
        d["galois_elt"] = {pairs[2 * i]: pairs[2 * i + 1] for i in range(cnt)}  # shift -> element
    d["galois_elt_zero"] = r.i32()
    d["size"] = r.u64()
    d["keys"] = {}
    for _ in range(r.u32()):
        elt = r.i32()
        d["keys"][elt] = r.u64s(d["size"])
    d["zero_key"] = r.u64s(d["size"])
    r.done()
    return d


PARSERS = {"context": parse_context, "secretkey": parse_secretkey, "publickey": parse_publickey,
           "plaintext": parse_plaintext, "ciphertext": parse_ciphertext, "relinkey": parse_relinkey,
           "galoiskey": parse_galoiskey, "switchkey": parse_switchkey}


def unframe_buffer(buf):
    """serializer::serialize: one zlib stream (compress(), default level) holding the save() bytes"""
    d = zlib.decompressobj()
    out = d.decompress(bytes(buf))
    if not d.eof or d.unused_data:
        raise ValueError("not exactly one complete zlib stream")
    return out


def unframe_file(data):
    """serializer::save_to_file: u64 size of the compressed buffer, then the buffer"""
    (size,) = struct.unpack_from("<Q", data, 0)
    if size != len(data) - 8:
        raise ValueError("size prefix %d, %d bytes follow" % (size, len(data) - 8))
    return unframe_buffer(data[8:])

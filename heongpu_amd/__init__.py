"""heongpu_amd -- MI355X-native RNS-polynomial backend behind HEonGPU's operator API.

The product is heongpu_amd/lib/libhegpu.so (hand-written HIP for gfx950 plus
a C ABI, see include/hegpu.h).  This package is the ctypes plumbing used by
tests/ and bench.py; importing it without the built library fails loudly.
"""
from . import _lib
from .api import (BFV, CKKS, SEC_NONE, SEC_128, SEC_192, SEC_256, TABLES_QP, TABLES_Q_BSK, OP_CKKS_RELIN, OP_CKKS_RESCALE,
                  OP_CKKS_GALOIS, OP_CKKS_ROTATE_HOISTED, OP_BFV_MULTIPLY, OP_BFV_RELIN, OP_BFV_GALOIS, E_INVALID, E_LOGIC, E_RUNTIME,
                  E_NODEVICE, Context, HEError, Rng, TfheContext, OP_KEYGEN_SECRET, OP_KEYGEN_PUBLIC,
                  OP_KEYGEN_SWITCH, OP_CKKS_ENCRYPT, OP_BFV_ENCRYPT, OP_BFV_DECRYPT, OP_BFV_DECODE, OP_CKKS_ENCODE,
                  OP_CKKS_DECODE, OP_BFV_MULTIPLY_PLAIN, GATE_NAND, GATE_AND, GATE_AND_FIRST_NOT,
                  GATE_NOR, GATE_OR, GATE_XNOR, GATE_XOR, GATE_NOT, steps_to_galois_elt, to_device, to_host, default_options, broadcast_key, broadcast_bytes, broadcast_path_name,
                  BCAST_FLAT, BCAST_TREE, BCAST_STAGED, BCAST_SAME_DEVICE)

_lib.load()

__all__ = ["BFV", "CKKS", "SEC_NONE", "SEC_128", "SEC_192", "SEC_256", "TABLES_QP", "TABLES_Q_BSK", "Context", "HEError", "Rng",
           "steps_to_galois_elt", "to_device", "to_host", "default_options", "broadcast_key", "broadcast_bytes", "broadcast_path_name",
           "BCAST_FLAT", "BCAST_TREE", "BCAST_STAGED", "BCAST_SAME_DEVICE"]

"""ctypes binding of libhegpu.so (the C ABI declared in include/hegpu.h).

This is plumbing only: every function here forwards to a C entry point; there
is no Python/CPU fallback.  If the HIP library is missing the import fails
loudly, and any device call without a GPU returns HEGPU_E_NODEVICE.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhegpu.so")
# TESTS ONLY: tests/audit/run_audit.py binds the INSTRUMENTED build of the same sources (tests/audit/Makefile: every FP64
# operation checked against exact integer arithmetic).  Only that file name is accepted; nothing in the product sets it.
if os.environ.get("HEGPU_AUDIT_LIB"):
    if os.path.basename(os.environ["HEGPU_AUDIT_LIB"]) != "libhegpu_audit.so":
        raise ImportError("HEGPU_AUDIT_LIB must name tests/audit/lib/libhegpu_audit.so")
    LIB_PATH = os.environ["HEGPU_AUDIT_LIB"]

u64 = ctypes.c_uint64
u64p = ctypes.c_void_p  # device or host address passed as integer
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
voidp = ctypes.c_void_p

# (name, restype, argtypes) -- must list EVERY symbol of include/hegpu.h
SIGNATURES = [
    ("hegpu_last_error", ctypes.c_char_p, []),
    ("hegpu_version", ctypes.c_char_p, []),
    ("hegpu_context_create", c_int,
     [c_int, c_int, ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int), c_int, u64, c_int,
      ctypes.POINTER(voidp)]),
    ("hegpu_context_create_default", c_int, [c_int, c_int, c_int, u64, c_int, ctypes.POINTER(voidp)]),
    ("hegpu_context_create_from_primes", c_int,
     [c_int, c_int, ctypes.POINTER(u64), c_int, c_int, u64, ctypes.POINTER(voidp)]),
    ("hegpu_validate_coeff_modulus_values", c_int, [c_int, ctypes.POINTER(u64), c_int, c_int, c_int]),
    ("hegpu_context_destroy", None, [voidp]),
    ("hegpu_context_upload", c_int, [voidp]),
    ("hegpu_context_clone", c_int, [voidp, ctypes.POINTER(voidp)]),
    ("hegpu_context_upload_device", c_int, [voidp, c_int]),
    ("hegpu_context_device", c_int, [voidp]),
    ("hegpu_broadcast_key", c_int, [ctypes.POINTER(voidp), c_int, ctypes.POINTER(voidp), c_size_t, ctypes.POINTER(voidp)]),
    ("hegpu_broadcast_bytes", c_int, [ctypes.POINTER(c_int), c_int, ctypes.POINTER(voidp), c_size_t, ctypes.POINTER(voidp),
                                      ctypes.POINTER(c_int)]),
    ("hegpu_last_broadcast_path", c_int, []),
    ("hegpu_context_set_option", c_int, [voidp, ctypes.c_char_p, c_int]),
    ("hegpu_context_get_option", c_int, [voidp, ctypes.c_char_p, ctypes.POINTER(c_int)]),
    ("hegpu_tfhe_context_set_option", c_int, [voidp, ctypes.c_char_p, c_int]),
    ("hegpu_context_int", ctypes.c_long, [voidp, ctypes.c_char_p]),
    ("hegpu_context_get", ctypes.c_long, [voidp, ctypes.c_char_p, voidp, ctypes.c_long]),
    ("hegpu_context_device_ptr", voidp, [voidp, ctypes.c_char_p]),
    ("hegpu_steps_to_galois_elt", c_int, [c_int, c_int, c_int]),
    ("hegpu_ntt", c_int, [voidp, c_int, u64p, u64p, c_int, c_int, c_int, c_int, voidp, voidp, voidp]),
    ("hegpu_GPU_NTT", c_int, [voidp, c_int, u64p, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_GPU_NTT_Inplace", c_int, [voidp, c_int, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_GPU_INTT", c_int, [voidp, c_int, u64p, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_GPU_INTT_Inplace", c_int, [voidp, c_int, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_GPU_NTT_Modulus_Ordered_Inplace", c_int, [voidp, c_int, u64p, c_int, c_int, c_int, c_int, voidp, voidp]),
    ("hegpu_GPU_NTT_Poly_Ordered_Inplace", c_int, [voidp, c_int, u64p, c_int, c_int, c_int, c_int, voidp, voidp]),
    ("hegpu_addition", c_int, [voidp, u64p, u64p, u64p, c_int, c_int, c_int, c_int, voidp]),
    ("hegpu_cross_multiplication", c_int,
     [voidp, c_int, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_cipher_broadcast", c_int, [voidp, u64p, u64, u64p, u64, c_int, c_int, c_int, c_int, c_int, voidp]),
    ("hegpu_keyswitch_multiply_accumulate", c_int,
     [voidp, u64p, u64, u64p, u64p, u64, c_int, c_int, c_int, c_int, c_int, c_int, voidp]),
    ("hegpu_base_conversion_DtoQtilde", c_int, [voidp, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq", c_int, [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq_permute", c_int,
     [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq_leveled_stage_one", c_int, [voidp, u64p, u64, u64p, u64, c_int, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq_leveled_stage_two", c_int,
     [voidp, u64p, u64, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, c_int, voidp]),
    ("hegpu_move_cipher_leveled", c_int, [voidp, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq_rescale", c_int, [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_divide_round_lastq_extended", c_int,
     [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, c_int, voidp]),
    ("hegpu_fast_convertion", c_int, [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, voidp]),
    ("hegpu_fast_floor", c_int, [voidp, u64p, u64, u64p, u64, c_int, voidp]),
    ("hegpu_workspace_bytes", c_size_t, [voidp, c_int, c_int, c_int]),
    ("hegpu_ckks_multiply", c_int, [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, c_int, voidp]),
    ("hegpu_ckks_relinearize_inplace", c_int, [voidp, u64p, u64, u64p, c_int, c_int, voidp, c_size_t, voidp]),
    ("hegpu_probe_ckks_relinearize", c_int,
     [voidp, u64p, u64, u64p, c_int, c_int, voidp, c_size_t, ctypes.c_uint, voidp]),
    ("hegpu_ckks_rescale_inplace", c_int, [voidp, u64p, u64, c_int, c_int, voidp, c_size_t, voidp]),
    ("hegpu_ckks_apply_galois", c_int,
     [voidp, u64p, u64, u64p, u64, u64p, c_int, c_int, c_int, voidp, c_size_t, voidp]),
    ("hegpu_ckks_rotate_hoisted", c_int,
     [voidp, u64p, u64, u64p, u64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_int), c_int, c_int, c_int, voidp,
      c_size_t, voidp]),
    ("hegpu_bfv_multiply", c_int, [voidp, u64p, u64, u64p, u64, u64p, u64, c_int, voidp, c_size_t, voidp]),
    ("hegpu_bfv_relinearize_inplace", c_int, [voidp, u64p, u64, u64p, c_int, voidp, c_size_t, voidp]),
    ("hegpu_bfv_apply_galois", c_int,
     [voidp, u64p, u64, u64p, u64, u64p, c_int, c_int, voidp, c_size_t, voidp]),
    ("hegpu_rng_create", c_int, [u64, ctypes.POINTER(voidp)]),
    ("hegpu_rng_create_seeded", c_int, [ctypes.c_char_p, ctypes.POINTER(voidp)]),
    ("hegpu_rng_create_from_entropy", c_int, [ctypes.POINTER(voidp)]),
    ("hegpu_drbg_block", c_int, [ctypes.c_char_p, u64, u64, ctypes.POINTER(ctypes.c_uint32)]),
    ("hegpu_rng_destroy", None, [voidp]),
    ("hegpu_generate_secret_key", c_int, [voidp, voidp, c_int, u64p, voidp, c_size_t, voidp]),
    ("hegpu_generate_public_key", c_int, [voidp, voidp, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_generate_relin_key", c_int, [voidp, voidp, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_generate_switch_key", c_int, [voidp, voidp, u64p, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_generate_galois_key", c_int, [voidp, voidp, u64p, c_int, u64p, voidp, c_size_t, voidp]),
    ("hegpu_ckks_encrypt", c_int, [voidp, voidp, u64p, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_ckks_decrypt", c_int, [voidp, u64p, u64p, c_int, u64p, voidp]),
    ("hegpu_bfv_encrypt", c_int, [voidp, voidp, u64p, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_bfv_decrypt", c_int, [voidp, u64p, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_bfv_noise_rns", c_int, [voidp, u64p, u64p, u64p, voidp]),
    ("hegpu_bfv_plain_to_ntt", c_int, [voidp, u64p, u64p, voidp]),
    ("hegpu_negacyclic_shift", c_int, [voidp, u64p, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_ckks_constant_op", c_int, [voidp, c_int, u64p, ctypes.c_double, u64p, c_int, c_int, voidp]),
    ("hegpu_ckks_gaussian_integer_op", c_int,
     [voidp, c_int, u64p, ctypes.c_double, ctypes.c_double, u64p, c_int, c_int, voidp]),
    ("hegpu_ckks_mult_i", c_int, [voidp, u64p, u64p, c_int, c_int, c_int, voidp]),
    ("hegpu_cipherplain_multiplication", c_int, [voidp, u64p, u64p, u64p, c_int, voidp]),
    ("hegpu_bfv_plain_addsub", c_int, [voidp, u64p, u64p, u64p, c_int, voidp]),
    ("hegpu_bfv_multiply_plain", c_int, [voidp, u64p, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_bfv_encode", c_int, [voidp, voidp, c_int, u64p, voidp]),
    ("hegpu_ckks_encode", c_int, [voidp, voidp, c_int, ctypes.c_double, u64p, voidp, c_size_t, voidp]),
    ("hegpu_ckks_decode", c_int, [voidp, u64p, c_int, ctypes.c_double, voidp, voidp, c_size_t, voidp]),
    ("hegpu_ckks_encode_complex", c_int, [voidp, voidp, c_int, ctypes.c_double, u64p, voidp, c_size_t, voidp]),
    ("hegpu_ckks_decode_complex", c_int, [voidp, u64p, c_int, ctypes.c_double, voidp, voidp, c_size_t, voidp]),
    ("hegpu_ckks_encode_coeff", c_int, [voidp, voidp, c_int, ctypes.c_double, u64p, voidp]),
    ("hegpu_ckks_decode_coeff", c_int, [voidp, u64p, c_int, ctypes.c_double, voidp, voidp, c_size_t, voidp]),
    ("hegpu_ckks_encode_scalar", c_int, [voidp, ctypes.c_double, ctypes.c_double, u64p, voidp]),
    ("hegpu_bfv_decode", c_int, [voidp, u64p, u64p, voidp, c_size_t, voidp]),
    ("hegpu_tfhe_context_create", c_int, [ctypes.POINTER(voidp)]),
    ("hegpu_tfhe_context_destroy", None, [voidp]),
    ("hegpu_tfhe_context_int", ctypes.c_long, [voidp, ctypes.c_char_p]),
    ("hegpu_tfhe_prime", u64, [voidp]),
    ("hegpu_tfhe_prepare_bootkey", c_int, [voidp, u64p, u64p, voidp]),
    ("hegpu_tfhe_prepared_format", c_int, [voidp, u64p, c_int]),
    ("hegpu_tfhe_gate_precompute", c_int, [voidp, c_int, voidp, voidp, voidp, voidp, voidp, voidp, c_int, voidp]),
    ("hegpu_tfhe_bootstrapping", c_int, [voidp, voidp, voidp, u64p, voidp, voidp, c_int, voidp]),
    ("hegpu_tfhe_status", c_int, [voidp, voidp]),
    ("hegpu_tfhe_key_switching", c_int, [voidp, voidp, voidp, voidp, voidp, voidp, voidp, c_int, voidp]),
    ("hegpu_tfhe_generate_secret_key", c_int, [voidp, voidp, voidp, voidp, voidp]),
    ("hegpu_tfhe_generate_bootstrapping_key", c_int,
     [voidp, voidp, voidp, voidp, u64p, voidp, voidp, voidp, c_size_t, voidp]),
    ("hegpu_tfhe_encrypt", c_int, [voidp, voidp, voidp, voidp, c_int, voidp, voidp, voidp]),
    ("hegpu_tfhe_decrypt_phase", c_int, [voidp, voidp, voidp, voidp, c_int, voidp, voidp]),
    ("hegpu_tfhe_mux", c_int,
     [voidp, voidp, voidp, voidp, voidp, voidp, voidp, voidp, voidp, u64p, voidp, voidp, c_int, voidp, c_size_t,
      voidp]),
    ("hegpu_tfhe_gate", c_int,
     [voidp, c_int, voidp, voidp, voidp, voidp, voidp, voidp, u64p, voidp, voidp, c_int, voidp, c_size_t, voidp]),
]

_lib = None


def load():
    """Load libhegpu.so and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C heongpu_amd/csrc`.  There is no CPU fallback.")
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must
    # be the copy already in the process when libhegpu.so is loaded, otherwise
    # two HIP runtimes coexist and neither sees the other's device state.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib

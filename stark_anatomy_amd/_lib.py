"""
_lib.py — loads stark_anatomy_amd/libstark_hip.so (built in-tree by `make -C stark_anatomy_amd/csrc`)
and declares the ctypes prototypes of every entry point in include/stark_hip.h.

There is no fallback: if the shared library is missing the import fails, and if no gfx950 device is
usable every compute entry point returns SH_EHIP (raised as StarkHipError).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libstark_hip.so")

SH_OK, SH_EINVAL, SH_EDIVZERO, SH_EHIP, SH_ENOMEM, SH_ECALLBACK, SH_EBOUNDS, SH_EPARSE = 0, -1, -2, -3, -4, -5, -6, -7


class StarkHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libstark_hip error {code}: {msg}")
        self.code = code


if not os.path.exists(SO_PATH):
    raise ImportError(
        f"{SO_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C stark_anatomy_amd/csrc). stark_anatomy_amd has no CPU fallback.")

try:
    # torch bundles its own HIP runtime.  If libstark_hip.so (linked against /opt/rocm) is loaded first the process ends
    # up with two runtimes and torch then reports no device; loading torch first makes both share torch's copy.
    import torch  # noqa: F401
except ImportError:
    pass

lib = C.CDLL(SO_PATH)

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
szp = C.POINTER(C.c_size_t)
ctx_t = C.c_void_p
ps_t = C.c_void_p


class FriParams(C.Structure):
    _fields_ = [("expansion", C.c_uint32), ("num_colinearity_tests", C.c_uint32), ("offset", C.c_uint64 * 2)]


TRANSCRIPT_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint32, u8p, C.c_int32, u64p)
dist_t = C.c_void_p


class P2POp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("is_send", C.c_int32), ("d_buf", C.c_void_p), ("bytes", C.c_size_t)]


ALL_TO_ALL_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SENDRECV_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(P2POp), C.c_size_t, C.c_void_p)


class DistTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all", ALL_TO_ALL_CB), ("all_gather", ALL_GATHER_CB), ("sendrecv", SENDRECV_CB)]


_PROTOS = {
    "sh_last_error": (C.c_char_p, []),
    "sh_version": (C.c_char_p, []),
    "sh_ctx_create": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(ctx_t)]),
    "sh_ctx_destroy": (C.c_int32, [ctx_t]),
    "sh_ctx_chain_retries": (C.c_uint64, [ctx_t]),
    "sh_ctx_sync": (C.c_int32, [ctx_t]),
    "sh_ctx_device": (C.c_int32, [ctx_t, C.POINTER(C.c_int32)]),
    "sh_dev_alloc": (C.c_int32, [ctx_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "sh_dev_free": (C.c_int32, [ctx_t, C.c_void_p]),
    "sh_dev_upload": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_dev_download": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_fp_new": (C.c_int32, [ctx_t, u64p, u64p, C.c_size_t]),
    "sh_fp_add": (C.c_int32, [ctx_t, u64p, u64p, u64p, C.c_size_t]),
    "sh_fp_sub": (C.c_int32, [ctx_t, u64p, u64p, u64p, C.c_size_t]),
    "sh_fp_mul": (C.c_int32, [ctx_t, u64p, u64p, u64p, C.c_size_t]),
    "sh_fp_div": (C.c_int32, [ctx_t, u64p, u64p, u64p, C.c_size_t]),
    "sh_fp_inv": (C.c_int32, [ctx_t, u64p, u64p, C.c_size_t]),
    "sh_dev_fp_add": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_dev_fp_sub": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_dev_fp_mul": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_dev_fp_div": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_dev_fp_inv": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sh_ntt": (C.c_int32, [ctx_t, u64p, C.c_uint32, C.c_int32]),
    "sh_dev_ntt": (C.c_int32, [ctx_t, C.c_void_p, C.c_uint32, C.c_int32]),
    "sh_ntt_batch": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.c_uint32, C.c_int32]),
    "sh_dev_ntt_batch": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32]),
    "sh_twiddle_scale": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32]),
    "sh_dev_twiddle_scale": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32]),
    "sh_coset_lde": (C.c_int32, [ctx_t, u64p, C.c_uint32, C.c_uint32, u64p, u64p]),
    "sh_dev_zero": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t]),
    "sh_dev_transpose": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "sh_dev_interleave_blocks": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "sh_dev_merkle_root_elems": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sh_dev_merkle_commit_pairs": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u8p]),
    "sh_poly_mul": (C.c_int32, [ctx_t, u64p, C.c_size_t, u64p, C.c_size_t, u64p]),
    "sh_dev_poly_mul": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sh_coset_zerofier_div": (C.c_int32, [ctx_t, u64p, C.c_uint32, C.c_uint32, u64p, u64p, u64p]),
    "sh_dev_coset_zerofier_div": (C.c_int32, [ctx_t, C.c_void_p, C.c_uint32, C.c_uint32, u64p, u64p, C.c_void_p]),
    "sh_poly_divmod_zerofier": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.c_size_t, u64p, u64p, u64p]),
    "sh_dev_poly_divmod_zerofier": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_size_t, u64p, C.c_void_p, C.c_void_p]),
    "sh_dev_coset_lde": (C.c_int32, [ctx_t, C.c_void_p, C.c_uint32, C.c_uint32, u64p, C.c_void_p]),
    "sh_dev_coset_lde_batch": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, u64p, C.c_void_p]),
    "sh_merkle_commit": (C.c_int32, [ctx_t, u8p, C.c_size_t, u8p]),
    "sh_dev_merkle_commit": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u8p]),
    "sh_hash_leaves": (C.c_int32, [ctx_t, u64p, C.c_size_t, u8p]),
    "sh_dev_hash_leaves": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sh_merkle_commit_elems": (C.c_int32, [ctx_t, u64p, C.c_size_t, u8p]),
    "sh_dev_merkle_commit_elems": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u8p]),
    "sh_dev_merkle_commit_elems_noalloc": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u8p]),
    "sh_dev_merkle_subtrees": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "sh_merkle_open": (C.c_int32, [ctx_t, u8p, C.c_size_t, C.c_size_t, u8p, szp]),
    "sh_fri_fold": (C.c_int32, [ctx_t, u64p, C.c_size_t, u64p, u64p, u64p, u64p]),
    "sh_dev_fri_fold": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u64p, u64p, u64p, C.c_void_p]),
    "sh_dev_fri_fold_range": (C.c_int32, [ctx_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, u64p, u64p, u64p, C.c_void_p]),
    "sh_dev_scale_powers": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_uint64, u64p]),
    "sh_fri_num_rounds": (C.c_uint32, [C.c_size_t, C.c_uint32, C.c_uint32]),
    "sh_fri_commit": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.POINTER(FriParams), TRANSCRIPT_CB, C.c_void_p, u8p, u64p, szp]),
    "sh_dev_fri_commit": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.POINTER(FriParams), TRANSCRIPT_CB, C.c_void_p, u8p, u64p, szp]),
    "sh_proofstream_new": (C.c_int32, [C.POINTER(ps_t)]),
    "sh_proofstream_free": (C.c_int32, [ps_t]),
    "sh_proofstream_push_json": (C.c_int32, [ps_t, C.c_char_p, C.c_size_t]),
    "sh_proofstream_push_merkle_root": (C.c_int32, [ps_t, u8p]),
    "sh_proofstream_push_codeword": (C.c_int32, [ps_t, u64p, C.c_size_t]),
    "sh_proofstream_len": (C.c_int32, [ps_t, szp, szp]),
    "sh_proofstream_pull": (C.c_int32, [ps_t, C.POINTER(C.c_void_p), szp]),
    "sh_proofstream_serialize": (C.c_int32, [ps_t, u8p, C.c_size_t, szp]),
    "sh_proofstream_deserialize": (C.c_int32, [C.c_char_p, C.c_size_t, C.POINTER(ps_t)]),
    "sh_proofstream_prover_fiat_shamir": (C.c_int32, [ps_t, u8p]),
    "sh_proofstream_verifier_fiat_shamir": (C.c_int32, [ps_t, u8p]),
    "sh_fri_commit_proofstream": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.POINTER(FriParams), ps_t, u8p, u64p, szp]),
    "sh_dev_fri_commit_proofstream": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.POINTER(FriParams), ps_t, u8p, u64p, szp]),
    "sh_fri_commit_batch": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.c_size_t, C.POINTER(FriParams), C.POINTER(ps_t), u8p, u64p, szp]),
    "sh_dev_fri_commit_batch": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(FriParams), C.POINTER(ps_t), u8p, u64p, szp]),
    "sh_proofstream_push_leaves": (C.c_int32, [ps_t, u64p, C.c_size_t]),
    "sh_proofstream_push_path": (C.c_int32, [ps_t, u8p, C.c_size_t]),
    "sh_merkle_open_batch": (C.c_int32, [ctx_t, u8p, C.c_size_t, u64p, C.c_size_t, u8p]),
    "sh_dev_merkle_open_elems_batch": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, u64p, C.c_size_t, u8p]),
    "sh_fri_sample_indices": (C.c_int32, [u8p, C.c_uint64, C.c_uint64, C.c_uint32, u64p]),
    "sh_fri_prove": (C.c_int32, [ctx_t, u64p, C.c_size_t, C.POINTER(FriParams), ps_t, u64p, szp]),
    "sh_dev_fri_prove": (C.c_int32, [ctx_t, C.c_void_p, C.c_size_t, C.POINTER(FriParams), ps_t, u64p, szp]),
    "sh_fri_proof_format": (C.c_char_p, []),
    "sh_fri_verify": (C.c_int32, [ps_t, C.c_size_t, C.POINTER(FriParams), C.POINTER(C.c_int32), u64p, szp]),
    "sh_dist_unique_id": (C.c_int32, [u8p]),
    "sh_dist_create": (C.c_int32, [ctx_t, C.c_int32, C.c_int32, u8p, C.POINTER(dist_t)]),
    "sh_dist_create_with_transport": (C.c_int32, [ctx_t, C.c_int32, C.c_int32, C.POINTER(DistTransport), C.POINTER(dist_t)]),
    "sh_dist_destroy": (C.c_int32, [dist_t]),
    "sh_dist_rank": (C.c_int32, [dist_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sh_dist_selftest": (C.c_int32, [dist_t]),
    "sh_dist_ntt": (C.c_int32, [dist_t, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sh_dist_coset_lde": (C.c_int32, [dist_t, C.c_void_p, C.c_uint32, C.c_uint32, u64p, C.c_void_p]),
    "sh_dist_all_gather": (C.c_int32, [dist_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sh_dist_fri_commit": (C.c_int32, [dist_t, C.c_void_p, C.c_size_t, C.POINTER(FriParams), ps_t, C.c_size_t, u8p, u64p, szp]),
    "sh_sha256": (C.c_int32, [C.c_char_p, C.c_size_t, u8p]),
    "sh_blake2b512": (C.c_int32, [C.c_char_p, C.c_size_t, u8p]),
    "sh_merkle_verify": (C.c_int32, [u8p, u8p, u8p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32)]),
    "sh_sample_field": (C.c_int32, [u8p, u64p]),
    "sh_root_of_unity": (C.c_int32, [C.c_uint32, u64p]),
    "sh_generator": (C.c_int32, [u64p]),
    "sh_fp_scalar": (C.c_int32, [C.c_int32, u64p, u64p, u64p]),
}
EXPORTS = sorted(_PROTOS)

for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)          # AttributeError here == the .so does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args


def check(rc):
    if rc != SH_OK:
        raise StarkHipError(rc, lib.sh_last_error().decode(errors="replace"))

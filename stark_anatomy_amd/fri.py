"""
fri.py — the FRI commit phase (SURVEY.md §3e, §8c G3–G8) over the GPU kernels, with the interface of
the upstream tutorial's `Fri` class that the reference crate is being written towards:
    Fri(offset, omega, domain_length, expansion_factor, num_colinearity_tests).commit(codeword, proof_stream)
Per round: leaf-hash + Merkle root of the codeword (GPU) -> proof_stream.push(root) ->
alpha = sample(proof_stream.prover_fiat_shamir()) (host, byte-exact transcript) -> split-and-fold
fused with the next round's leaf hashing (GPU).  The last codeword is pushed in the clear.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FriParams, check, lib
from .context import _u8p, _u64p, array_to_ints, default_context, device_ptr, elems_to_array, scalar
from .field import GENERATOR, PRIME, FieldElement, root_of_unity
from .merkle import MerkleHash
from .proof_stream import ProofItem, ProofStream


def sample(digest):
    """G3: field element from a 32-byte Fiat-Shamir digest (big-endian integer mod p)."""
    d = np.frombuffer(bytes(digest), dtype=np.uint8).copy()
    out = np.zeros((1, 2), dtype=np.uint64)
    check(lib.sh_sample_field(_u8p(d), _u64p(out)))
    return FieldElement(array_to_ints(out)[0])


def fold(codeword, alpha, offset, omega, ctx=None):
    """one split-and-fold step (G4) on the GPU; host arrays in/out."""
    ctx = ctx or default_context()
    cw = elems_to_array(codeword)
    out = np.empty((cw.shape[0] // 2, 2), dtype=np.uint64)
    v = lambda x: x.value() if isinstance(x, FieldElement) else int(x)
    rc = lib.sh_fri_fold(ctx.handle, _u64p(cw), cw.shape[0], _u64p(scalar(v(alpha))), _u64p(scalar(v(offset))),
                         _u64p(scalar(v(omega))), _u64p(out))
    if rc == _lib.SH_EDIVZERO:
        raise ZeroDivisionError("Cannot divide by zero")
    check(rc)
    return out


def sample_indices(seed, size, reduced_size, number):
    """G10: `number` indices in [0,size) with distinct residues mod reduced_size from a 32-byte seed (host)."""
    d = np.frombuffer(bytes(seed), dtype=np.uint8).copy()
    out = np.zeros(max(number, 1), dtype=np.uint64)
    check(lib.sh_fri_sample_indices(_u8p(d), size, reduced_size, number, _u64p(out)))
    return [int(v) for v in out[:number]]


class Fri:
    def __init__(self, offset=GENERATOR, omega=None, domain_length=None, expansion_factor=4, num_colinearity_tests=17, ctx=None):
        self.offset = offset.value() if isinstance(offset, FieldElement) else int(offset)
        self.domain_length = int(domain_length)
        self.expansion_factor = int(expansion_factor)
        self.num_colinearity_tests = int(num_colinearity_tests)
        log_n = self.domain_length.bit_length() - 1
        assert 1 << log_n == self.domain_length, "domain length must be a power of two"
        w = root_of_unity(log_n).value()
        if omega is not None:
            o = omega.value() if isinstance(omega, FieldElement) else int(omega)
            assert o == w, "omega must be the canonical primitive root g^(2^119/N) (SURVEY.md §8c G1)"
        self.omega = w
        self.ctx = ctx or default_context()

    def num_rounds(self):                                     # G5
        return int(lib.sh_fri_num_rounds(self.domain_length, self.expansion_factor, self.num_colinearity_tests))

    def _params(self):
        p = FriParams()
        p.expansion, p.num_colinearity_tests = self.expansion_factor, self.num_colinearity_tests
        p.offset[0], p.offset[1] = self.offset & 0xFFFFFFFFFFFFFFFF, self.offset >> 64
        return p

    def commit(self, codeword, proof_stream):
        """codeword: host values (list of ints / (N,2) uint64) or a device buffer / CUDA tensor of N elements.
        Pushes R Merkle roots and the last codeword to `proof_stream`; returns (roots, last_codeword)."""
        N, R = self.domain_length, self.num_rounds()
        roots = np.zeros((max(R, 1), 64), dtype=np.uint8)
        last = np.empty((N >> max(R - 1, 0), 2), dtype=np.uint64)
        last_len = C.c_size_t(0)
        p = self._params()
        on_device = not isinstance(codeword, (list, tuple, np.ndarray))
        if on_device:
            rc = lib.sh_dev_fri_commit_proofstream(self.ctx.handle, device_ptr(codeword), N, C.byref(p), proof_stream.handle,
                                                   _u8p(roots), _u64p(last), C.byref(last_len))
        else:
            cw = elems_to_array(codeword)
            assert cw.shape[0] == N, "initial codeword length does not match length of initial codeword"
            rc = lib.sh_fri_commit_proofstream(self.ctx.handle, _u64p(cw), N, C.byref(p), proof_stream.handle,
                                               _u8p(roots), _u64p(last), C.byref(last_len))
        check(rc)
        return [MerkleHash(roots[i].tobytes()) for i in range(R)], last[:last_len.value]

    def prove(self, codeword, proof_stream):
        """commit + query phase (sh_fri_prove).  Returns the top-level test indices."""
        N = self.domain_length
        top = np.zeros(max(self.num_colinearity_tests, 1), dtype=np.uint64)
        n_top = C.c_size_t(0)
        p = self._params()
        if isinstance(codeword, (list, tuple, np.ndarray)):
            cw = elems_to_array(codeword)
            assert cw.shape[0] == N, "initial codeword length does not match length of initial codeword"
            rc = lib.sh_fri_prove(self.ctx.handle, _u64p(cw), N, C.byref(p), proof_stream.handle, _u64p(top), C.byref(n_top))
        else:
            rc = lib.sh_dev_fri_prove(self.ctx.handle, device_ptr(codeword), N, C.byref(p), proof_stream.handle, _u64p(top), C.byref(n_top))
        check(rc)
        return [int(v) for v in top[:n_top.value]]

    def verify(self, proof_stream):
        """host-side verifier (sh_fri_verify; G12) reading `proof_stream` from its read index.
        Returns (accepted, reason): reason names the first failed check."""
        ok = C.c_int32(0)
        top = np.zeros(max(self.num_colinearity_tests, 1), dtype=np.uint64)
        n_top = C.c_size_t(0)
        p = self._params()
        check(lib.sh_fri_verify(proof_stream.handle, self.domain_length, C.byref(p), C.byref(ok), _u64p(top), C.byref(n_top)))
        return bool(ok.value), ("" if ok.value else lib.sh_last_error().decode(errors="replace"))

    def commit_batch(self, codewords, proof_streams):
        """B independent commits in lock step (sh_fri_commit_batch).  `codewords`: host array of shape (B, N, 2) /
        list of B codewords, or a device buffer holding B*N elements back to back; `proof_streams`: B ProofStreams.
        Returns (roots[b][r], last_codewords (B, last_len, 2))."""
        N, R, B = self.domain_length, self.num_rounds(), len(proof_streams)
        roots = np.zeros((B, max(R, 1), 64), dtype=np.uint8)
        L_last = N >> max(R - 1, 0)
        last = np.empty((B, L_last, 2), dtype=np.uint64)
        last_len = C.c_size_t(0)
        p = self._params()
        handles = (_lib.ps_t * B)(*[ps.handle for ps in proof_streams])
        if isinstance(codewords, (list, tuple, np.ndarray)):
            cw = np.ascontiguousarray(np.stack([elems_to_array(x) for x in codewords]) if not isinstance(codewords, np.ndarray) else codewords,
                                      dtype=np.uint64).reshape(B * N, 2)
            rc = lib.sh_fri_commit_batch(self.ctx.handle, _u64p(cw), B, N, C.byref(p), handles, _u8p(roots), _u64p(last), C.byref(last_len))
        else:
            rc = lib.sh_dev_fri_commit_batch(self.ctx.handle, device_ptr(codewords), B, N, C.byref(p), handles, _u8p(roots), _u64p(last), C.byref(last_len))
        check(rc)
        assert last_len.value == L_last
        return [[MerkleHash(roots[b, r].tobytes()) for r in range(R)] for b in range(B)], last

    def commit_with_callback(self, codeword, callback):
        """Low-level form: `callback(round, root_bytes, want_alpha) -> int|None` owns the transcript
        (sh_fri_commit / sh_transcript_cb)."""
        N, R = self.domain_length, self.num_rounds()
        roots = np.zeros((max(R, 1), 64), dtype=np.uint8)
        last = np.empty((N >> max(R - 1, 0), 2), dtype=np.uint64)
        last_len = C.c_size_t(0)
        p = self._params()
        err = []

        def cb(_user, rnd, root_p, want, alpha_p):
            try:
                a = callback(int(rnd), bytes(root_p[:64]), bool(want))
                if want:
                    a = int(a) % PRIME
                    alpha_p[0], alpha_p[1] = a & 0xFFFFFFFFFFFFFFFF, a >> 64
                return 0
            except Exception as e:  # never unwind through the C frame
                err.append(e)
                return 1
        c_cb = _lib.TRANSCRIPT_CB(cb)
        if isinstance(codeword, (list, tuple, np.ndarray)):
            cw = elems_to_array(codeword)
            rc = lib.sh_fri_commit(self.ctx.handle, _u64p(cw), N, C.byref(p), c_cb, None, _u8p(roots), _u64p(last), C.byref(last_len))
        else:
            rc = lib.sh_dev_fri_commit(self.ctx.handle, device_ptr(codeword), N, C.byref(p), c_cb, None, _u8p(roots), _u64p(last), C.byref(last_len))
        if err:
            raise err[0]
        check(rc)
        return [MerkleHash(roots[i].tobytes()) for i in range(R)], last[:last_len.value]

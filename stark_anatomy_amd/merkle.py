"""
merkle.py — mirror of src/iop/types/merkle.rs: MerkleHash, MerkleTree::{commit, open, verify}.
`commit` and `open` run on the GPU (Blake2b-512 kernels); `verify` is log2(L) host compressions
exactly as merkle.rs:114-154.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .context import _u8p, _u64p, default_context, elems_to_array


class MerkleHash:
    """64-byte digest newtype (merkle.rs:9-25); serde shape {"inner":[b0..b63]}."""
    __slots__ = ("inner",)

    def __init__(self, inner):
        inner = bytes(inner)
        if len(inner) != 64:
            raise ValueError("MerkleHash is 64 bytes")
        self.inner = inner

    def as_bytes(self): return self.inner                     # merkle.rs:22-24
    def __eq__(self, o): return isinstance(o, MerkleHash) and o.inner == self.inner
    def __hash__(self): return hash(self.inner)
    def __repr__(self): return f"MerkleHash({self.inner.hex()[:16]}…)"
    def to_json(self): return '{"inner":[' + ",".join(str(b) for b in self.inner) + "]}"


def _leaves_array(leaves):
    if isinstance(leaves, np.ndarray):
        return np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 64)
    return np.frombuffer(b"".join(h.as_bytes() if isinstance(h, MerkleHash) else bytes(h) for h in leaves),
                         dtype=np.uint8).reshape(-1, 64).copy()


class MerkleTree:
    """owns a vector of (already hashed) leaves and the root derived from them (merkle.rs:28-31)."""

    def __init__(self, leaves_arr, root, ctx):
        self._leaves, self.root, self._ctx = leaves_arr, root, ctx

    @classmethod
    def commit(cls, leaves, ctx=None):                        # merkle.rs:35-68
        ctx = ctx or default_context()
        arr = _leaves_array(leaves)
        L = arr.shape[0]
        assert L > 0 and (L & (L - 1)) == 0, "assertion failed: leaves.len().is_power_of_two()"   # merkle.rs:36
        root = np.zeros(64, dtype=np.uint8)
        check(lib.sh_merkle_commit(ctx.handle, _u8p(arr), L, _u8p(root)))
        return cls(arr, MerkleHash(root.tobytes()), ctx)

    def open(self, index):                                    # merkle.rs:70-108
        L = self._leaves.shape[0]
        assert index < L, "assertion failed: index < self.leaves.len()"                          # merkle.rs:75
        path = np.zeros((64, 64), dtype=np.uint8)
        n = C.c_size_t(0)
        rc = lib.sh_merkle_open(self._ctx.handle, _u8p(self._leaves), L, index, _u8p(path), C.byref(n))
        if rc == _lib.SH_EBOUNDS:
            raise IndexError(lib.sh_last_error().decode())
        check(rc)
        return [MerkleHash(path[i].tobytes()) for i in range(n.value)]

    def open_batch(self, indices):
        """many openings of this tree in one GPU pass (levels built once)"""
        L = self._leaves.shape[0]
        idx = np.asarray(list(indices), dtype=np.uint64)
        assert all(int(i) < L for i in idx), "assertion failed: index < self.leaves.len()"      # merkle.rs:75
        depth = L.bit_length() - 1
        out = np.zeros((max(len(idx), 1), max(depth, 1), 64), dtype=np.uint8)
        rc = lib.sh_merkle_open_batch(self._ctx.handle, _u8p(self._leaves), L, _u64p(idx), len(idx), _u8p(out))
        if rc == _lib.SH_EBOUNDS:
            raise IndexError(lib.sh_last_error().decode())
        check(rc)
        return [[MerkleHash(out[q, l].tobytes()) for l in range(depth)] for q in range(len(idx))]

    def verify(self, leaf, path, index):                      # merkle.rs:110-112
        return self.is_leaf_in_tree(self.root, leaf, path, index)

    @staticmethod
    def is_leaf_in_tree(root, leaf, path, index):             # merkle.rs:114-154
        assert len(path) > 0, "assertion failed: !path.is_empty()"                               # merkle.rs:126
        assert index < (1 << len(path)), "assertion failed: index < (1 << path.len())"           # merkle.rs:122
        p = np.frombuffer(b"".join(h.as_bytes() for h in path), dtype=np.uint8).copy()
        r = np.frombuffer(root.as_bytes(), dtype=np.uint8).copy()
        l = np.frombuffer(leaf.as_bytes(), dtype=np.uint8).copy()
        ok = C.c_int32(0)
        check(lib.sh_merkle_verify(_u8p(r), _u8p(l), _u8p(p), len(path), index, C.byref(ok)))
        return bool(ok.value)


def hash_leaves(elems, ctx=None):
    """leaf hashes of field elements: Blake2b512(value.to_le_bytes()), on the GPU."""
    ctx = ctx or default_context()
    e = elems_to_array(elems)
    out = np.empty((e.shape[0], 64), dtype=np.uint8)
    check(lib.sh_hash_leaves(ctx.handle, _u64p(e), e.shape[0], _u8p(out)))
    return out


def commit_elems(elems, ctx=None):
    """root of MerkleTree::commit(hash_leaves(elems)) in one fused GPU pass."""
    ctx = ctx or default_context()
    e = elems_to_array(elems)
    root = np.zeros(64, dtype=np.uint8)
    check(lib.sh_merkle_commit_elems(ctx.handle, _u64p(e), e.shape[0], _u8p(root)))
    return MerkleHash(root.tobytes())

"""
proof_stream.py — mirror of src/iop/types/proof_stream.rs: ProofStream<S> with push / pull /
serialize / deserialize / prover_fiat_shamir / verifier_fiat_shamir, backed by the library's
host-side transcript (sh_proofstream_*), byte-identical to serde_json's compact output.

Items: anything with `to_json() -> str`, or plain JSON-able Python values (dict order == struct
field declaration order, as serde writes it).  The FRI item type is ProofItem.
"""
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import check, lib
from .context import _u8p, _u64p, elems_to_array
from .merkle import MerkleHash


class ProofStreamError(Exception):
    """proof_stream.rs:9-15"""


class OutOfBoundsReadIndexError(ProofStreamError):
    def __init__(self): super().__init__("Read index is out of bounds")


class SerializationError(ProofStreamError):
    def __init__(self): super().__init__("Serialization error")


class ProofItem:
    """enum ProofItem { MerkleRoot(MerkleHash), Codeword(Vec<u128>), Leaves(Vec<u128>), Path(Vec<MerkleHash>) } — serde
    externally tagged (G6; Leaves and Path are the query-phase items, G9)."""

    def __init__(self, kind, payload):
        self.kind, self.payload = kind, payload

    @classmethod
    def MerkleRoot(cls, h): return cls("MerkleRoot", h if isinstance(h, MerkleHash) else MerkleHash(h))
    @classmethod
    def Codeword(cls, values): return cls("Codeword", elems_to_array(values))
    @classmethod
    def Leaves(cls, values): return cls("Leaves", elems_to_array(values))
    @classmethod
    def Path(cls, hashes): return cls("Path", [h if isinstance(h, MerkleHash) else MerkleHash(h) for h in hashes])

    def __eq__(self, o):
        if not isinstance(o, ProofItem) or o.kind != self.kind:
            return False
        if self.kind in ("MerkleRoot", "Path"):
            return self.payload == o.payload
        return self.payload.shape == o.payload.shape and bool((self.payload == o.payload).all())

    @classmethod
    def from_json(cls, text):
        obj = json.loads(text)
        if "MerkleRoot" in obj:
            return cls.MerkleRoot(bytes(obj["MerkleRoot"]["inner"]))
        if "Leaves" in obj:
            return cls.Leaves(obj["Leaves"])
        if "Path" in obj:
            return cls.Path([bytes(h["inner"]) for h in obj["Path"]])
        return cls.Codeword(obj["Codeword"])


def _to_json(item):
    if hasattr(item, "to_json"):
        return item.to_json()
    return json.dumps(item, separators=(",", ":"))


class ProofStream:
    def __init__(self, item_from_json=None):
        h = _lib.ps_t()
        check(lib.sh_proofstream_new(C.byref(h)))
        self.handle = h
        self._from_json = item_from_json or json.loads

    def __del__(self):
        if getattr(self, "handle", None):
            lib.sh_proofstream_free(self.handle)
            self.handle = None

    def push(self, item):                                     # proof_stream.rs:25-27
        if isinstance(item, ProofItem):
            if item.kind == "MerkleRoot":
                r = np.frombuffer(item.payload.as_bytes(), dtype=np.uint8).copy()
                check(lib.sh_proofstream_push_merkle_root(self.handle, _u8p(r)))
            elif item.kind == "Codeword":
                check(lib.sh_proofstream_push_codeword(self.handle, _u64p(item.payload), item.payload.shape[0]))
            elif item.kind == "Leaves":
                check(lib.sh_proofstream_push_leaves(self.handle, _u64p(item.payload), item.payload.shape[0]))
            else:
                d = np.frombuffer(b"".join(h.as_bytes() for h in item.payload), dtype=np.uint8).copy() if item.payload else np.zeros(1, np.uint8)
                check(lib.sh_proofstream_push_path(self.handle, _u8p(d), len(item.payload)))
            return
        js = _to_json(item).encode()
        check(lib.sh_proofstream_push_json(self.handle, js, len(js)))

    def pull(self):                                           # proof_stream.rs:29-36
        p, n = C.c_void_p(), C.c_size_t()
        rc = lib.sh_proofstream_pull(self.handle, C.byref(p), C.byref(n))
        if rc == _lib.SH_EBOUNDS:
            raise OutOfBoundsReadIndexError()
        check(rc)
        return self._from_json(C.string_at(p.value, n.value).decode())

    def __len__(self):
        n, r = C.c_size_t(), C.c_size_t()
        check(lib.sh_proofstream_len(self.handle, C.byref(n), C.byref(r)))
        return n.value

    @property
    def read_index(self):
        n, r = C.c_size_t(), C.c_size_t()
        check(lib.sh_proofstream_len(self.handle, C.byref(n), C.byref(r)))
        return r.value

    def serialize(self):                                      # proof_stream.rs:38-40
        need = C.c_size_t()
        check(lib.sh_proofstream_serialize(self.handle, None, 0, C.byref(need)))
        buf = np.empty(need.value, dtype=np.uint8)
        check(lib.sh_proofstream_serialize(self.handle, _u8p(buf), buf.size, C.byref(need)))
        return buf.tobytes()

    @classmethod
    def deserialize(cls, data, item_from_json=None):          # proof_stream.rs:42-49
        data = bytes(data)
        h = _lib.ps_t()
        rc = lib.sh_proofstream_deserialize(data, len(data), C.byref(h))
        if rc == _lib.SH_EPARSE:
            raise SerializationError()
        check(rc)
        ps = cls.__new__(cls)
        ps.handle, ps._from_json = h, item_from_json or json.loads
        return ps

    def __eq__(self, o):                                      # derive(PartialEq), proof_stream.rs:17
        return isinstance(o, ProofStream) and self.serialize() == o.serialize() and self.read_index == o.read_index

    def prover_fiat_shamir(self):                             # proof_stream.rs:51-56
        out = np.zeros(32, dtype=np.uint8)
        check(lib.sh_proofstream_prover_fiat_shamir(self.handle, _u8p(out)))
        return out.tobytes()

    def verifier_fiat_shamir(self):                           # proof_stream.rs:58-64
        out = np.zeros(32, dtype=np.uint8)
        check(lib.sh_proofstream_verifier_fiat_shamir(self.handle, _u8p(out)))
        return out.tobytes()

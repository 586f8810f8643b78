"""
stark_anatomy_amd — MI355X (gfx950) implementation of the FRI-commit hot path of the `stark-anat`
crate (greged93/stark-anatomy), behind the crate's own API surface:

    field        FieldElement (+ - * / pow is_zero value), FieldVector (batched, on the GPU)
    merkle       MerkleHash, MerkleTree.commit / open / verify
    proof_stream ProofStream.push / pull / serialize / deserialize / *_fiat_shamir
    ntt, fri     NTT / INTT / coset LDE, split-and-fold, Fri.commit  (the layer the reference lacks)
    poly         products through the NTT, zerofier quotients, batched inversion

Everything computes in hand-written HIP kernels through libstark_hip.so (C ABI: include/stark_hip.h).
There is no CPU fallback: importing this package without the built library raises ImportError.
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP extension is missing)
from . import dist  # noqa: F401  (sharding helpers; imports torch lazily)
from . import dist_ntt  # noqa: F401  (one transform / one commit over several GPUs; imports torch lazily)
from ._lib import StarkHipError
from .context import Context, DeviceBuffer, array_to_ints, default_context, elems_to_array
from .field import GENERATOR, ONE, PRIME, ZERO, FieldElement, FieldVector, generator, root_of_unity
from .fri import Fri, fold, sample, sample_indices
from .merkle import MerkleHash, MerkleTree, commit_elems, hash_leaves
from .ntt import coset_lde, dev_coset_lde, dev_coset_lde_batch, dev_ntt, intt, ntt
from .poly import coset_zerofier_div, poly_divmod_zerofier, poly_mul
from .proof_stream import OutOfBoundsReadIndexError, ProofItem, ProofStream, ProofStreamError, SerializationError

__all__ = [
    "StarkHipError", "Context", "DeviceBuffer", "default_context", "elems_to_array", "array_to_ints",
    "PRIME", "GENERATOR", "ZERO", "ONE", "FieldElement", "FieldVector", "generator", "root_of_unity",
    "Fri", "fold", "sample", "MerkleHash", "MerkleTree", "commit_elems", "hash_leaves",
    "ntt", "intt", "coset_lde", "dev_ntt", "dev_coset_lde", "dev_coset_lde_batch", "poly_mul", "coset_zerofier_div", "poly_divmod_zerofier",
    "ProofStream", "ProofItem", "ProofStreamError", "OutOfBoundsReadIndexError", "SerializationError",
]

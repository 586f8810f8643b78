"""
Host-side pieces of the query phase, no GPU needed: index sampling, the Leaves / Path proof
items, and the verifier (sh_fri_verify) — against the pure-Python oracle, the committed golden proof of config C1
(FRI prove+verify, 2^10 domain, expansion 4) and tampered proofs.
"""
import hashlib
import json

import numpy as np
import pytest

import stark_anatomy_amd as sa
from oracle import pyoracle as po
from stark_anatomy_amd.fri import sample_indices
from stark_anatomy_amd.proof_stream import ProofItem, ProofStream

P, G = po.P, po.G


def host_fri(N, expansion=4, t=17, offset=G):
    f = sa.Fri.__new__(sa.Fri)                    # no GPU context: only verify() is used
    f.offset, f.domain_length, f.expansion_factor, f.num_colinearity_tests = offset, N, expansion, t
    return f


def test_sample_indices_golden_and_oracle(golden):
    for v in golden("sample_indices.json"):
        seed = bytes.fromhex(v["seed"])
        assert sample_indices(seed, v["size"], v["reduced_size"], v["number"]) == v["indices"]
        assert po.sample_indices(seed, v["size"], v["reduced_size"], v["number"]) == v["indices"]
    idx = sample_indices(hashlib.sha256(b"q").digest(), 1 << 19, 128, 17)
    assert len({i % 128 for i in idx}) == 17 and all(0 <= i < (1 << 19) for i in idx)
    with pytest.raises(sa.StarkHipError):         # more indices than residues: the tutorial's assert
        sample_indices(bytes(32), 64, 8, 9)


def test_query_items_json_shape():
    ps, ops = ProofStream(), po.ProofStream()
    vals = [0, 1, P - 1]
    path = [hashlib.blake2b(bytes([i]), digest_size=64).digest() for i in range(3)]
    ps.push(ProofItem.Leaves(vals)); ops.push(po.item_leaves(*vals))
    ps.push(ProofItem.Path(path)); ops.push(po.item_path(path))
    ps.push(ProofItem.Path([])); ops.push(po.item_path([]))
    assert ps.serialize() == ops.serialize()
    back = ProofStream.deserialize(ps.serialize(), ProofItem.from_json)
    assert back.pull() == ProofItem.Leaves(vals) and back.pull() == ProofItem.Path(path)


def test_verify_golden_c1_proof(golden):
    g = golden("fri_prove_c1.json")
    ser = g["stream"].encode()
    assert hashlib.sha256(ser).hexdigest() == g["stream_sha256"]
    ok, why = host_fri(g["N"]).verify(ProofStream.deserialize(ser))
    assert ok, why
    assert po.fri_verify(po.ProofStream.deserialize(ser), g["N"], 4, 17, G) == (True, "")


@pytest.fixture(scope="module")
def c1_items(golden):
    return json.loads(golden("fri_prove_c1.json")["stream"])


def _verify_items(items, N=1 << 10):
    ser = json.dumps(items, separators=(",", ":")).encode()
    got = host_fri(N).verify(ProofStream.deserialize(ser))
    want = po.fri_verify(po.ProofStream.deserialize(ser), N, 4, 17, G)
    assert got[0] == want[0], (got, want)
    return got


def test_verify_rejects_tampering(c1_items):
    import copy
    R, t = 4, 17
    first_leaves = R + 1
    first_path = first_leaves + t

    def mutated(fn):
        items = copy.deepcopy(c1_items)
        fn(items)
        return items
    cases = {
        "root": lambda it: it[1]["MerkleRoot"]["inner"].__setitem__(5, it[1]["MerkleRoot"]["inner"][5] ^ 1),
        "last codeword value": lambda it: it[R]["Codeword"].__setitem__(3, (it[R]["Codeword"][3] + 1) % P),
        "last codeword length": lambda it: it[R]["Codeword"].pop(),
        "leaf a": lambda it: it[first_leaves]["Leaves"].__setitem__(0, (it[first_leaves]["Leaves"][0] + 1) % P),
        "leaf c": lambda it: it[first_leaves + 3]["Leaves"].__setitem__(2, (it[first_leaves + 3]["Leaves"][2] + 5) % P),
        "path byte": lambda it: it[first_path + 4]["Path"][2]["inner"].__setitem__(0, it[first_path + 4]["Path"][2]["inner"][0] ^ 0x80),
        "path too short": lambda it: it[first_path]["Path"].pop(),
        "swapped paths": lambda it: it.__setitem__(slice(first_path, first_path + 2), [it[first_path + 1], it[first_path]]),
        "truncated": lambda it: it.__delitem__(slice(len(it) - 3, None)),
        "wrong kind": lambda it: it.__setitem__(first_leaves, it[0]),
    }
    for name, fn in cases.items():
        ok, why = _verify_items(mutated(fn))
        assert not ok and why, name
    ok, _ = _verify_items(c1_items)
    assert ok


def test_verify_rejects_high_degree_and_wrong_parameters(c1_items):
    # a random codeword is not low degree: the honest prover's proof of it must be rejected at the degree test
    cw = po.splitmix_elems(2, 1 << 10)
    ops = po.ProofStream()
    po.fri_prove(cw, 4, 17, G, ops)
    ok, why = host_fri(1 << 10).verify(ProofStream.deserialize(ops.serialize()))
    assert not ok and "degree" in why
    # same proof, verifier told other parameters
    ser = json.dumps(c1_items, separators=(",", ":")).encode()
    assert not host_fri(1 << 11).verify(ProofStream.deserialize(ser))[0]
    assert not host_fri(1 << 10, offset=pow(G, 3, P)).verify(ProofStream.deserialize(ser))[0]
    assert not host_fri(1 << 10, t=16).verify(ProofStream.deserialize(ser))[0]


def test_verify_tiny_domains():
    # R = 0 (no rounds) and R = 1 (one root, no queries) follow the commit definition G5
    for N, t in ((4, 17), (8, 1)):
        c = po.splitmix_elems(3, N // 4)
        cw = po.coset_lde(c, 2, G)
        ops = po.ProofStream()
        po.fri_prove(cw, 4, t, G, ops)
        ser = ops.serialize()
        assert po.fri_verify(po.ProofStream.deserialize(ser), N, 4, t, G)[0]
        assert host_fri(N, t=t).verify(ProofStream.deserialize(ser))[0]

"""
GPU parity for the query phase: batched Merkle openings and the full prover, byte-for-byte against the
pure-Python oracle / committed golden proofs, then accepted by the host verifier; larger domains through the
prove -> serialize -> deserialize -> verify round trip and through tampering.
"""
import hashlib
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stark_anatomy_amd as sa  # noqa: E402
from oracle import coracle as co  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from stark_anatomy_amd import _lib  # noqa: E402
from stark_anatomy_amd.context import _u8p, _u64p  # noqa: E402
from stark_anatomy_amd.proof_stream import ProofStream  # noqa: E402

P, G = po.P, po.G


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_l", [1, 2, 5, 10, 13])
def test_open_batch_vs_reference_open(ctx, log_l):
    L = 1 << log_l
    leaves = [hashlib.blake2b(i.to_bytes(4, "little"), digest_size=64).digest() for i in range(L)]
    tree = sa.MerkleTree.commit([sa.MerkleHash(h) for h in leaves], ctx)
    idx = sorted({0, L - 1, L // 2, (L // 3) | 1 if L > 2 else 0, 7 % L})
    got = tree.open_batch(idx)
    for q, i in enumerate(idx):
        want = po.merkle_open(leaves, i) if L <= 1024 else co.merkle_open(np.frombuffer(b"".join(leaves), dtype=np.uint8).reshape(L, 64), i)
        assert [h.as_bytes() for h in got[q]] == [bytes(w) for w in want]
        assert [h.as_bytes() for h in tree.open(i)] == [h.as_bytes() for h in got[q]]      # single-open entry point agrees
        assert tree.verify(sa.MerkleHash(leaves[i]), got[q], i)
    with pytest.raises(AssertionError):                                              # merkle.rs:75
        tree.open_batch([0, L])
    assert tree.open_batch([]) == []


def test_open_elems_batch_device(ctx):
    L = 1 << 12
    x = co.splitmix_elems(8, L)
    d = ctx.alloc(L * 16).upload(x)
    idx = np.array([0, 1, 2047, 2048, 4095, 1234], dtype=np.uint64)
    out = np.zeros((len(idx), 12, 64), dtype=np.uint8)
    _lib.check(_lib.lib.sh_dev_merkle_open_elems_batch(ctx.handle, d.ptr, L, _u64p(idx), len(idx), _u8p(out)))
    leaves = co.hash_leaves(x)
    root = co.merkle_commit_elems(x)
    for q, i in enumerate(idx):
        assert co.merkle_verify(root, leaves[int(i)].tobytes(), [out[q, l].tobytes() for l in range(12)], int(i))
        assert po.merkle_verify(root, leaves[int(i)].tobytes(), [out[q, l].tobytes() for l in range(12)], int(i))
    assert _lib.lib.sh_dev_merkle_open_elems_batch(ctx.handle, d.ptr, L, _u64p(np.array([L], dtype=np.uint64)), 1, _u8p(out)) == _lib.SH_EBOUNDS
    assert _lib.lib.sh_dev_merkle_open_elems_batch(ctx.handle, d.ptr, 3, _u64p(idx), 1, _u8p(out)) == _lib.SH_EINVAL


@pytest.mark.parametrize("name", ["fri_prove_c1.json", "fri_prove_n4096.json"])
def test_prove_golden(ctx, golden, name):
    g = golden(name)
    c = co.splitmix_elems(g["seed"], 1 << g["log_n"])
    cw = sa.coset_lde(c, 2, G, ctx)
    fri = sa.Fri(G, None, g["N"], 4, 17, ctx=ctx)
    ps = ProofStream()
    top = fri.prove(cw, ps)
    ser = ps.serialize()
    assert top == g["top_level_indices"]
    assert len(ps) == g["num_items"] and len(ser) == g["stream_len"]
    assert hashlib.sha256(ser).hexdigest() == g["stream_sha256"]
    if "stream" in g:
        assert ser == g["stream"].encode()
    ok, why = fri.verify(ProofStream.deserialize(ser))
    assert ok, why


@pytest.mark.parametrize("log_n,t,seed", [(5, 4, 1), (6, 17, 2), (9, 17, 3), (11, 8, 4)])
def test_prove_vs_oracle_stream_bytes(ctx, log_n, t, seed):
    c = po.splitmix_elems(seed, 1 << log_n)
    cw_o = po.coset_lde(c, 2, G)
    ops = po.ProofStream()
    top_o = po.fri_prove(cw_o, 4, t, G, ops)
    N = len(cw_o)
    fri = sa.Fri(G, None, N, 4, t, ctx=ctx)
    ps = ProofStream()
    d = ctx.upload_elems(cw_o)                                   # device-resident entry point
    top = fri.prove(d, ps)
    assert top == top_o
    assert ps.serialize() == ops.serialize()
    assert fri.verify(ProofStream.deserialize(ps.serialize()))[0]
    assert po.fri_verify(po.ProofStream.deserialize(ps.serialize()), N, 4, t, G) == (True, "")


@pytest.mark.parametrize("log_n", [16, 20])
def test_prove_verify_round_trip_large(ctx, log_n):
    c = co.splitmix_elems(50 + log_n, 1 << log_n)
    N = 4 << log_n
    d_c = ctx.alloc((1 << log_n) * 16).upload(c)
    d_cw = ctx.alloc(N * 16)
    sa.dev_coset_lde(ctx, d_c, log_n, 2, G, d_cw)
    fri = sa.Fri(G, None, N, 4, 17, ctx=ctx)
    ps = ProofStream()
    top = fri.prove(d_cw, ps)
    R = fri.num_rounds()
    assert len(top) == 17 and len(ps) == R + 1 + (R - 1) * 17 * 4
    # the commit part of the proof is what Fri.commit produces on its own
    ps_c = ProofStream()
    fri.commit(d_cw, ps_c)
    assert ps.serialize()[: len(ps_c.serialize()) - 1] == ps_c.serialize()[:-1]
    ser = ps.serialize()
    ok, why = fri.verify(ProofStream.deserialize(ser))
    assert ok, why
    items = json.loads(ser)
    k = R + 1 + 17 * 4 * (R - 2) + 5                              # a Leaves item of the last tested round
    assert "Leaves" in items[k]
    items[k]["Leaves"][1] = (items[k]["Leaves"][1] + 1) % P
    assert not fri.verify(ProofStream.deserialize(json.dumps(items, separators=(",", ":")).encode()))[0]


def test_prove_of_random_codeword_is_rejected(ctx):
    cw = co.splitmix_elems(77, 1 << 12)
    fri = sa.Fri(G, None, 1 << 12, 4, 17, ctx=ctx)
    ps = ProofStream()
    fri.prove(cw, ps)
    ok, why = fri.verify(ProofStream.deserialize(ps.serialize()))
    assert not ok and "degree" in why


def test_prove_degenerate_round_counts(ctx):
    for N, t in ((4, 17), (8, 1), (64, 17)):
        c = po.splitmix_elems(5, max(N // 4, 1))
        cw = po.coset_lde(c, 2, G)
        ops = po.ProofStream()
        top_o = po.fri_prove(cw, 4, t, G, ops)
        fri = sa.Fri(G, None, N, 4, t, ctx=ctx)
        ps = ProofStream()
        assert fri.prove(cw, ps) == top_o
        assert ps.serialize() == ops.serialize()
        assert fri.verify(ProofStream.deserialize(ps.serialize()))[0]

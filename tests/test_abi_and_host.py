"""
CPU-side checks of the product library (no GPU needed, no compute kernels run):
  - libstark_hip.so loads and exports every symbol include/stark_hip.h declares;
  - compute entry points fail loudly (SH_EHIP) without a device — there is no CPU fallback;
  - the host-side pieces of the path (ProofStream bytes, SHA-256, Blake2b, Merkle verify, sampling,
    roots of unity, scalar field ops) match the reference's KATs, the golden vectors and the oracle.
"""
import ctypes as C
import hashlib
import json
import os
import random
import re

import numpy as np
import pytest

import stark_anatomy_amd as sa
from oracle import coracle as co
from oracle import pyoracle as po
from stark_anatomy_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = po.P


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "stark_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(sh_[a-z0-9_]+)\s*\(", hdr)) - {"sh_transcript_cb"})
    assert len(declared) > 50
    raw = C.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in stark_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes prototypes and header disagree"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sa.StarkHipError) as e:
        sa.Context(0)
    assert e.value.code == _lib.SH_EHIP and "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "stark_anatomy_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".cuh", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "stark_oracle.h" not in text and "liboracle" not in text, f


def test_scalar_field_ops_reference_kats(golden):
    k = golden("ref_kat.json")["field"]
    F = sa.FieldElement
    assert F(2).pow(F(160)).value() == int(k["pow"]["expected"])                       # field.rs:140-151
    assert (F(P - 10) + F(12)).value() == 2                                            # field.rs:153-166
    assert (sa.ZERO - F(12)).value() == P - 12                                         # field.rs:168-180
    a, b = F(int(k["mul"]["a"])), F(int(k["mul"]["b"]))
    assert (a * b).value() == int(k["mul"]["expected"])                                # field.rs:182-194
    assert (a / b).value() == int(k["div"]["expected"])                                # field.rs:196-208
    for m in k["inv_minus"]["k"]:                                                      # field.rs:210-220
        x = F(P - m)
        assert ((F(1) / x) * x).value() == 1
    with pytest.raises(ZeroDivisionError):                                             # field.rs:101-103
        F(1) / F(0)
    assert F(P + 5).value() == 5 and F(0).is_zero() and not F(1).is_zero()


def test_scalar_field_ops_vs_golden_table_and_oracle(golden):
    t = golden("field_ops.json")
    F = sa.FieldElement
    for r in t["rows"]:
        a, b = F(int(r["a"])), F(int(r["b"]))
        assert (a + b).value() == int(r["add"]) and (a - b).value() == int(r["sub"]) and (a * b).value() == int(r["mul"])
        if r["div"] is not None:
            assert (a / b).value() == int(r["div"])
    rnd = random.Random(5)
    for _ in range(500):
        x, y = rnd.randrange(P), rnd.randrange(1, P)
        assert (F(x) * F(y)).value() == co.fp_mul(x, y) and (F(x) / F(y)).value() == co.fp_div(x, y)
        assert F(x).pow(y % 1000).value() == pow(x, y % 1000, P)


def test_roots_of_unity_and_sampling(golden):
    t = golden("ntt_vectors.json")
    assert sa.generator().value() == int(t["generator"])
    for k, v in t["roots_of_unity"].items():
        assert sa.root_of_unity(int(k)).value() == int(v)
    for v in golden("sample_vectors.json"):
        assert sa.sample(bytes.fromhex(v["digest"])).value() == int(v["alpha"])
    rnd = random.Random(2)
    for _ in range(200):
        d = bytes(rnd.randrange(256) for _ in range(32))
        assert sa.sample(d).value() == co.sample(d) == int.from_bytes(d, "big") % P


def _h(fn, data, n):
    out = np.zeros(n, dtype=np.uint8)
    assert fn(bytes(data), len(data), out.ctypes.data_as(_lib.u8p)) == 0
    return out.tobytes()


def test_host_hashes(golden):
    for v in golden("blake2b_vectors.json"):
        assert _h(_lib.lib.sh_blake2b512, bytes.fromhex(v["msg"]), 64).hex() == v["digest"]
    for v in golden("sha256_vectors.json"):
        assert _h(_lib.lib.sh_sha256, bytes.fromhex(v["msg"]), 32).hex() == v["digest"]
    rnd = random.Random(9)
    for n in list(range(0, 140)) + [255, 256, 257, 1000]:
        m = bytes(rnd.randrange(256) for _ in range(n))
        assert _h(_lib.lib.sh_blake2b512, m, 64) == co.blake2b512(m)
        assert _h(_lib.lib.sh_sha256, m, 32) == co.sha256(m)


def test_proof_stream_reference_kats(golden):
    k = golden("ref_kat.json")["proof_stream"]
    ps = sa.ProofStream()
    for js in k["items_json"]:
        ps.push(json.loads(js))                                                        # TestStruct {a, b}
    assert list(ps.serialize()) == k["serialized"]                                     # proof_stream.rs:90-110
    assert list(ps.prover_fiat_shamir()) == k["prover_fiat_shamir"]                    # proof_stream.rs:166-193
    ps2 = sa.ProofStream.deserialize(ps.serialize())                                   # proof_stream.rs:112-129
    assert ps2 == ps
    for _ in range(k["verifier_pulls"]):
        ps.pull()
    assert list(ps.verifier_fiat_shamir()) == k["verifier_fiat_shamir"]                # proof_stream.rs:195-225
    assert ps.pull() == {"a": 4, "b": 5}
    with pytest.raises(sa.OutOfBoundsReadIndexError):                                  # proof_stream.rs:30-32
        ps.pull()
    psc = sa.ProofStream()
    for js in k["complex_items_json"]:
        psc.push(json.loads(js))
    back = sa.ProofStream.deserialize(psc.serialize())                                 # proof_stream.rs:131-164
    assert back == psc and back.pull() == json.loads(k["complex_items_json"][0])
    assert sa.ProofStream.deserialize(b" [ {\"a\" : 1, \"b\":2} ,\n{\"a\":2,\"b\":3}]").serialize() == b'[{"a":1,"b":2},{"a":2,"b":3}]'
    for bad in (b"", b"[", b"[1,]", b"{\"a\":1}", b"[1] x", b"[{\"a\":}]", b"[\"abc]"):
        with pytest.raises(sa.SerializationError):                                     # proof_stream.rs:43
            sa.ProofStream.deserialize(bad)
    assert sa.ProofStream().serialize() == b"[]"


def test_proof_items_match_oracle_bytes(golden):
    e = golden("fri_c1.json")
    ps, ops, pps = sa.ProofStream(item_from_json=None), co.ProofStream(), po.ProofStream()
    for r in e["roots"]:
        ps.push(sa.ProofItem.MerkleRoot(bytes.fromhex(r))); ops.push_root(bytes.fromhex(r)); pps.push(po.item_root(bytes.fromhex(r)))
    last = [int(v) for v in e["last_codeword"]]
    ps.push(sa.ProofItem.Codeword(last)); ops.push_codeword(co.to_arr(last)); pps.push(po.item_codeword(last))
    assert ps.serialize() == ops.serialize() == pps.serialize() == e["stream"].encode()
    assert ps.prover_fiat_shamir().hex() == e["final_fiat_shamir"]
    back = sa.ProofStream.deserialize(ps.serialize(), item_from_json=sa.ProofItem.from_json)
    assert back.pull() == sa.ProofItem.MerkleRoot(bytes.fromhex(e["roots"][0]))
    big = [0, 1, P - 1, 2**64, 2**127 + 12345]                                          # u128 values print as bare decimals
    ps3 = sa.ProofStream(); ps3.push(sa.ProofItem.Codeword(big))
    assert ps3.serialize() == ('[{"Codeword":[' + ",".join(str(v) for v in big) + "]}]").encode()


def test_deserialize_bulk_items_raw_or_verbatim():
    """sh_proofstream_deserialize keeps Path / Leaves / Codeword items that are in canonical text in raw form (no second
    parse for the verifier); the bytes it serialises again, the items it hands out and both Fiat-Shamir digests must be what
    the text path gives — and anything not canonical stays text, verbatim (only white space between tokens goes)."""
    rnd = random.Random(11)
    hashes = [bytes(rnd.randrange(256) for _ in range(64)) for _ in range(5)] + [bytes(64), bytes([255]) * 64]
    vals = [0, 1, 255, 256, P - 1, 2**64, 2**127 + 12345, 2**128 - 1]
    items = [sa.ProofItem.MerkleRoot(hashes[0]), sa.ProofItem.Codeword(vals), sa.ProofItem.Leaves(vals[:3]), sa.ProofItem.Path(hashes),
             sa.ProofItem.Path([]), sa.ProofItem.Codeword([]), sa.ProofItem.Path(hashes[:1])]
    ps = sa.ProofStream(item_from_json=sa.ProofItem.from_json)
    pps = po.ProofStream()
    for it in items:
        ps.push(it)
    pps.push(po.item_root(hashes[0])); pps.push(po.item_codeword(vals)); pps.push(po.item_leaves(*vals[:3])); pps.push(po.item_path(hashes))
    pps.push(po.item_path([])); pps.push(po.item_codeword([])); pps.push(po.item_path(hashes[:1]))
    text = ps.serialize()
    assert text == pps.serialize()
    back = sa.ProofStream.deserialize(text, item_from_json=sa.ProofItem.from_json)
    assert back.serialize() == text and back.prover_fiat_shamir() == ps.prover_fiat_shamir()
    for k, it in enumerate(items):
        assert back.pull() == it
        ps.pull()
        assert back.verifier_fiat_shamir() == ps.verifier_fiat_shamir(), k
    # the same stream with white space inside the items, and variants the raw form could not reproduce: text is kept
    spaced = text.replace(b",", b" ,\n").replace(b"[", b"[ ")
    assert sa.ProofStream.deserialize(spaced).serialize() == text
    for odd in (b'[{"Path":[{"inner":[007]}]}]', b'[{"Leaves":[1,2,340282366920938463463374607431768211456]}]', b'[{"Codeword":[1.0,2]}]',
                b'[{"Path":[{"inner":[256]}]}]', b'[{"Leaves":[1,2,3],"x":1}]', b'[{"Codeword":[1,2]}{}]'[:0] or b'[{"Codeword":[-1]}]'):
        try:
            again = sa.ProofStream.deserialize(odd).serialize()
        except sa.SerializationError:
            assert odd == b'[{"Path":[{"inner":[007]}]}]'              # a leading zero is not JSON: rejected by the general scanner
            continue
        assert again == odd


def test_deserialize_random_json_round_trips():
    """Random JSON streams, tokens separated by random white space: deserialize -> serialize must give exactly the tokens
    joined without white space (proof_stream.rs:42-49 then :38-40 on canonical text) — through the general scanner, its
    integer-run shortcut and the raw take-over of Path / Leaves / Codeword items alike."""
    rnd = random.Random(2024)

    def num():
        k = rnd.randrange(8)
        if k == 0: return "0"
        if k == 1: return str(rnd.randrange(256))
        if k == 2: return str(rnd.randrange(1 << 128))
        if k == 3: return "-" + str(rnd.randrange(1, 10**6))
        if k == 4: return str(rnd.randrange(1, 10**4)) + "." + str(rnd.randrange(10)) + str(rnd.randrange(1, 10))
        if k == 5: return str(rnd.randrange(1, 99)) + "e" + rnd.choice(["", "+", "-"]) + str(rnd.randrange(1, 30))
        if k == 6: return str(rnd.randrange(1 << 140))
        return str(rnd.randrange(10**rnd.randrange(1, 12)))

    def value(depth):
        k = rnd.randrange(10 if depth < 4 else 6)
        if k < 3: return [num()]
        if k == 3: return [rnd.choice(["true", "false", "null"])]
        if k < 6: return ['"' + "".join(rnd.choice(["a", "b", " ", "x", "_", ":", ",", "[", "]", "{", "}", "\\n", '\\"', "\\\\", "\\u0041"]) for _ in range(rnd.randrange(6))) + '"']
        if k < 8:
            out = ["["]
            for i in range(rnd.randrange(0, 12)):
                if i: out.append(",")
                out += [num()] if rnd.random() < 0.7 else value(depth + 1)
            return out + ["]"]
        out = ["{"]
        for i in range(rnd.randrange(0, 4)):
            if i: out.append(",")
            out += ['"k%d"' % i, ":"] + value(depth + 1)
        return out + ["}"]

    def typed():
        k = rnd.randrange(4)
        if k == 0:
            hs = [[str(rnd.randrange(256 if rnd.random() < 0.97 else 300)) for _ in range(64 if rnd.random() < 0.95 else 63)] for _ in range(rnd.randrange(0, 4))]
            out = ["{", '"Path"', ":", "["]
            for i, h in enumerate(hs):
                if i: out.append(",")
                out += ["{", '"inner"', ":", "["] + [t for j, v in enumerate(h) for t in (([","] if j else []) + [v])] + ["]", "}"]
            return out + ["]", "}"]
        tag = '"Leaves"' if k == 1 else '"Codeword"' if k == 2 else '"MerkleRoot"'
        if k == 3:
            return ["{", tag, ":", "{", '"inner"', ":", "["] + [t for j in range(64) for t in (([","] if j else []) + [str(rnd.randrange(256))])] + ["]", "}", "}"]
        vals = [str(rnd.randrange(1 << rnd.choice([7, 64, 127, 128, 130]))) for _ in range(rnd.randrange(0, 9))]
        return ["{", tag, ":", "["] + [t for j, v in enumerate(vals) for t in (([","] if j else []) + [v])] + ["]", "}"]

    for trial in range(300):
        toks = ["["]
        for i in range(rnd.randrange(0, 6)):
            if i: toks.append(",")
            toks += typed() if rnd.random() < 0.5 else value(0)
        toks.append("]")
        compact = "".join(toks).encode()
        ws = rnd.choice([0.0, 0.0, 0.05, 0.5])
        spaced = "".join(t + ("".join(rnd.choice(" \n\t\r") for _ in range(rnd.randrange(1, 3))) if rnd.random() < ws else "") for t in toks).encode()
        ps = sa.ProofStream.deserialize(spaced)
        assert ps.serialize() == compact, (trial, spaced[:200])
        assert sa.ProofStream.deserialize(compact).prover_fiat_shamir() == hashlib.sha256(compact).digest()


def test_merkle_verify_host_matches_oracle():
    rnd = random.Random(4)
    leaves = [bytes(rnd.randrange(256) for _ in range(64)) for _ in range(64)]
    root = co.merkle_commit(leaves)
    for i in (0, 1, 31, 63):
        path = co.merkle_open(leaves, i)
        mh = [sa.MerkleHash(h) for h in path]
        assert sa.MerkleTree.is_leaf_in_tree(sa.MerkleHash(root), sa.MerkleHash(leaves[i]), mh, i)
        assert not sa.MerkleTree.is_leaf_in_tree(sa.MerkleHash(root), sa.MerkleHash(leaves[i]), mh, i ^ 1)
        assert not sa.MerkleTree.is_leaf_in_tree(sa.MerkleHash(root), sa.MerkleHash(leaves[i ^ 5]), mh, i)
    with pytest.raises(AssertionError):                                                # merkle.rs:126
        sa.MerkleTree.is_leaf_in_tree(sa.MerkleHash(root), sa.MerkleHash(leaves[0]), [], 0)
    with pytest.raises(AssertionError):                                                # merkle.rs:122
        sa.MerkleTree.is_leaf_in_tree(sa.MerkleHash(root), sa.MerkleHash(leaves[0]), [sa.MerkleHash(leaves[1])], 2)


def test_num_rounds():
    for N in (1, 2, 64, 128, 1024, 1 << 14, 1 << 20, 1 << 24):
        for exp, t in ((4, 17), (4, 1), (1, 0), (8, 40)):
            assert _lib.lib.sh_fri_num_rounds(N, exp, t) == co.fri_num_rounds(N, exp, t) == po.fri_num_rounds(N, exp, t)
    assert _lib.lib.sh_fri_num_rounds(1024, 4, 17) == 4 and _lib.lib.sh_fri_num_rounds(1 << 20, 4, 17) == 14

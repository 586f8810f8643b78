"""
Pins the CPU oracle (oracle/stark_oracle.c, oracle/pyoracle.py) to the reference crate's OWN
known-answer tests, transcribed in tests/golden/ref_kat.json (field.rs / base.rs / utils.rs /
proof_stream.rs unit tests).  CPU only.
"""
import random

import pytest

from oracle import coracle as co
from oracle import pyoracle as po

P = po.P
I = co.I320


def val(s):
    s = str(s)
    if s.startswith("p"):
        return P + int(s[1:] or 0)
    if s.startswith("-p"):
        return -P
    return int(s)


def limbs(x):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(5)]


@pytest.fixture(scope="module")
def kat(golden):
    return golden("ref_kat.json")


def test_prime_literal(kat):                                # field.rs:15
    assert int(kat["prime"]["value"]) == P == 1 + 407 * 2**119
    import numpy as np
    out = np.zeros((1, 2), dtype=np.uint64); co.lib.or_fp_prime(co._p(out))
    assert co.to_ints(out)[0] == P


@pytest.mark.parametrize("ref", [False, True])
def test_field_kats(kat, ref):                              # field.rs:140-220
    f = kat["field"]
    assert co.fp_pow(val(f["pow"]["base"]), val(f["pow"]["exp"]), ref=ref) == val(f["pow"]["expected"])
    assert co.fp_add(val(f["add"]["a"]), val(f["add"]["b"]), ref=ref) == val(f["add"]["expected"])
    assert co.fp_sub(val(f["sub"]["a"]), val(f["sub"]["b"]), ref=ref) == val(f["sub"]["expected"])
    assert co.fp_mul(val(f["mul"]["a"]), val(f["mul"]["b"]), ref=ref) == val(f["mul"]["expected"])
    assert co.fp_div(val(f["div"]["a"]), val(f["div"]["b"]), ref=ref) == val(f["div"]["expected"])
    for k in f["inv_minus"]["k"]:
        b = P - k
        assert co.fp_mul(co.fp_div(1, b, ref=ref), b, ref=ref) == 1


def test_field_kats_pyoracle(kat):
    f = kat["field"]
    assert po.fp_pow(2, 160) == val(f["pow"]["expected"])
    assert po.fp_mul(val(f["mul"]["a"]), val(f["mul"]["b"])) == val(f["mul"]["expected"])
    assert po.fp_div(val(f["div"]["a"]), val(f["div"]["b"])) == val(f["div"]["expected"])
    assert po.fp_add(P - 10, 12) == 2 and po.fp_sub(0, 12) == P - 12
    for k in f["inv_minus"]["k"]:
        assert po.fp_mul(po.fp_div(1, P - k), P - k) == 1


def test_div_by_zero_is_an_error():                         # field.rs:101-103
    with pytest.raises(ZeroDivisionError):
        co.fp_div(5, 0)
    with pytest.raises(ZeroDivisionError):
        co.fp_div(5, 0, ref=True)
    with pytest.raises(ZeroDivisionError):
        po.fp_div(5, 0)


def test_i320_kats(kat):                                    # base.rs:224-641
    k = kat["i320"]
    mk = lambda l: I.from_int(sum(v << (64 * i) for i, v in enumerate(l)))
    assert co.lib.or_i320_pow(mk(k["pow"]["a"]), mk(k["pow"]["e"])).limbs() == k["pow"]["expected"]
    assert co.lib.or_i320_neg(mk(k["neg"]["a"])).limbs() == k["neg"]["expected"]
    assert co.lib.or_i320_add(mk(k["add"]["a"]), mk(k["add"]["b"])).limbs() == k["add"]["expected"]
    assert co.lib.or_i320_sub(mk(k["sub"]["a"]), mk(k["sub"]["b"])).limbs() == k["sub"]["expected"]
    assert co.lib.or_i320_mul(mk(k["mul"]["a"]), mk(k["mul"]["b"])).limbs() == k["mul"]["expected"]
    m = k["mul_negative"]
    assert co.lib.or_i320_mul(I.from_int(val(m["a"])), I.from_int(val(m["b"]))).to_int() == val(m["expected"])
    m = k["mul_high"]
    assert co.lib.or_i320_mul(I.from_int(val(m["a"])), I.from_int(val(m["b"]))).limbs() == m["expected"]
    prod = co.lib.or_i320_mul(I.from_int(P - 1), I.from_int(P - 1))
    r = I(); assert co.lib.or_i320_rem(prod, I.from_int(P), r) == 0
    assert r.limbs() == k["mul_near_boundary"]["expected"]
    q = I(); assert co.lib.or_i320_div(mk(k["division"]["a"]), mk(k["division"]["b"]), q) == 0
    assert q.limbs() == k["division"]["expected"]
    assert co.lib.or_i320_rem(mk(k["rem_simple"]["a"]), mk(k["rem_simple"]["b"]), r) == 0
    assert r.limbs() == k["rem_simple"]["expected"]
    a = mk(k["rem_complex"]["a"])
    assert co.lib.or_i320_rem(a, I.from_int(P), r) == 0 and r.limbs() == k["rem_complex"]["expected"]
    assert co.lib.or_i320_rem(a, I.from_int(-P), r) == 0 and r.limbs() == k["rem_complex_negative_modulus"]["expected"]
    assert co.lib.or_i320_rem(co.lib.or_i320_neg(a), I.from_int(P), r) == 0
    assert co.lib.or_i320_neg(r).limbs() == k["rem_complex_negative_dividend"]["expected_neg"]
    a, b = mk(k["compare"]["a"]), mk(k["compare"]["b"])
    na, nb = co.lib.or_i320_neg(a), co.lib.or_i320_neg(b)
    assert co.lib.or_i320_cmp(a, b) < 0 and co.lib.or_i320_cmp(a, nb) > 0
    assert co.lib.or_i320_cmp(na, b) < 0 and co.lib.or_i320_cmp(na, nb) > 0 and co.lib.or_i320_cmp(b, b) == 0
    assert I.from_int(val(k["from_u128"]["x"])).limbs() == k["from_u128"]["expected"]
    assert co.lib.or_i320_neg(I.from_int(val(k["from_i128"]["x"]))).limbs() == k["from_i128"]["expected_neg"]
    assert co.lib.or_i320_div(a, I.from_int(0), q) == -2   # base.rs:129-131


def test_i320_random_against_python_ints():
    rnd = random.Random(7)
    for _ in range(300):
        a = rnd.randrange(-(1 << 250), 1 << 250); b = rnd.randrange(-(1 << 60), 1 << 60) or 3
        ia, ib = I.from_int(a), I.from_int(b)
        assert co.lib.or_i320_add(ia, ib).to_int() == a + b
        assert co.lib.or_i320_sub(ia, ib).to_int() == a - b
        assert co.lib.or_i320_mul(ia, ib).to_int() == a * b
        r = I(); co.lib.or_i320_rem(ia, ib, r)
        expect = abs(a) % abs(b)                            # truncated, sign follows dividend (base.rs:139-151)
        assert r.to_int() == (-expect if a < 0 else expect)
    for _ in range(300):                                    # unsigned Knuth division, multi-limb divisors
        a = rnd.randrange(1 << rnd.randrange(1, 319)); b = rnd.randrange(1, 1 << rnd.randrange(1, 300))
        q = I(); co.lib.or_i320_div(I.from_int(a), I.from_int(b), q)
        assert q.to_int() == a // b


def test_ext_euclid_kats(kat):                              # utils.rs:24-67
    for e in kat["ext_euclid"]:
        g, s, t = I(), I(), I()
        assert co.lib.or_ext_euclid(I.from_int(val(e["a"])), I.from_int(val(e["b"])), g, s, t) == 0
        assert (g.to_int(), s.to_int(), t.to_int()) == (val(e["g"]), val(e["s"]), val(e["t"]))
        assert po.ext_euclid(val(e["a"]), val(e["b"])) == (val(e["g"]), val(e["s"]), val(e["t"]))


def test_proof_stream_kats(kat):                            # proof_stream.rs:90-225
    k = kat["proof_stream"]
    ps = co.ProofStream(); pps = po.ProofStream()
    import json
    for js in k["items_json"]:
        ps.push_raw(js.encode()); pps.push(json.loads(js))
    assert list(ps.serialize()) == k["serialized"] == list(pps.serialize())
    assert list(ps.prover_fiat_shamir()) == k["prover_fiat_shamir"] == list(pps.prover_fiat_shamir())
    for _ in range(k["verifier_pulls"]):
        ps.pull(); pps.pull()
    assert list(ps.verifier_fiat_shamir()) == k["verifier_fiat_shamir"] == list(pps.verifier_fiat_shamir())
    ps.pull()
    with pytest.raises(IndexError):                         # proof_stream.rs:30-32
        ps.pull()
    # round trip of the nested struct (proof_stream.rs:131-164)
    ps2 = po.ProofStream()
    for js in k["complex_items_json"]:
        ps2.push(json.loads(js))
    assert po.ProofStream.deserialize(ps2.serialize()).items == ps2.items
    assert ps2.serialize().decode() == "[" + ",".join(k["complex_items_json"]) + "]"


def test_merkle_reference_properties():                     # merkle.rs:178-282 (N = 64, random leaves)
    rnd = random.Random(3)
    N = 64
    leaves = [bytes(rnd.randrange(256) for _ in range(64)) for _ in range(N)]
    root = co.merkle_commit(leaves)
    assert root == po.merkle_commit(leaves)
    other = co.merkle_commit([bytes(rnd.randrange(256) for _ in range(64))] * N)
    for i, leaf in enumerate(leaves):
        path = co.merkle_open(leaves, i)
        assert path == po.merkle_open(leaves, i)
        assert co.merkle_verify(root, leaf, path, i) and po.merkle_verify(root, leaf, path, i)
        rl = bytes(rnd.randrange(256) for _ in range(64))
        assert not co.merkle_verify(root, rl, path, i)                       # random leaf
        j = (i + 1 + rnd.randrange(N - 1)) % N
        assert not co.merkle_verify(root, leaves[j], path, i)                # wrong leaf
        assert not co.merkle_verify(root, leaf, path, j)                     # wrong index
        assert not co.merkle_verify(rl, leaf, path, i)                       # changed root
        assert not co.merkle_verify(other, leaf, path, i)                    # different tree
        for t in range(len(path)):                                           # tampered path
            fp = list(path); fp[t] = rl
            assert not co.merkle_verify(root, leaf, fp, i)
    assert co.merkle_commit(leaves[:1]) == leaves[0]                         # Q5: one leaf => root == leaf
    with pytest.raises(ValueError):
        co.merkle_commit(leaves[:3])                                         # merkle.rs:36

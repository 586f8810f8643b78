#!/usr/bin/env python3
"""
gen_golden.py — regenerates the committed golden vectors under tests/golden/.

Source of truth = Python stdlib only (big ints, hashlib.blake2b / hashlib.sha256, json), through
oracle/pyoracle.py — an implementation independent of the reference's Rust crates, of the C
oracle and of the HIP kernels.  The reference crate itself cannot be run (no cargo/rustc in this
environment); its own known-answer tests are transcribed by hand in ref_kat.json (not produced
by this script).

Run from the repo root:  python tests/golden/gen_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402

P, G = po.P, po.G


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)


def elems_digest(vals):
    """sha256 over the canonical 16-byte little-endian encoding of each element"""
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(16, "little"))
    return h.hexdigest()


def main():
    # ---- Blake2b-512 (RFC 7693) ----
    msgs = [b"", b"abc", bytes(16), bytes(128), bytes(range(128)), bytes(range(16)), bytes(129), bytes(range(256)),
            hashlib.sha256(b"seed0").digest() * 4, hashlib.sha256(b"seed1").digest() * 4]
    dump("blake2b_vectors.json", [{"msg": m.hex(), "digest": po.blake2b512(m).hex()} for m in msgs])

    # ---- SHA-256 ----
    smsgs = [b"", b"abc", bytes(55), bytes(56), bytes(64), bytes(range(200))]
    dump("sha256_vectors.json", [{"msg": m.hex(), "digest": po.sha256(m).hex()} for m in smsgs])

    # ---- Merkle roots over leaf_i = blake2b(le16(i)) (G2 leaf hash of the integers 0..L-1) ----
    mk = []
    for L in (1, 2, 4, 64, 1024, 4096):
        leaves = po.hash_leaves(range(L))
        e = {"L": L, "root": po.merkle_commit(leaves).hex()}
        if L == 64:
            e["open"] = {str(i): [h.hex() for h in po.merkle_open(leaves, i)] for i in (0, 1, 37, 63)}
        mk.append(e)
    dump("merkle_roots.json", mk)

    # ---- field-op table over the edge set ----
    R = (1 << 128) % P
    edge = [0, 1, 2, P - 1, P - 2, 2**64 - 1, 2**64, 2**127, R, 2**64 - 3, 2**64 - 2, 2**32, 2**96 + 5, (P + 1) // 2, G]
    rows = []
    for a in edge:
        for b in edge:
            rows.append({"a": str(a), "b": str(b), "add": str(po.fp_add(a, b)), "sub": str(po.fp_sub(a, b)),
                         "mul": str(po.fp_mul(a, b)), "div": (str(po.fp_div(a, b)) if b else None)})
    dump("field_ops.json", {"p": str(P), "rows": rows})

    # ---- roots of unity + NTT of x_j = j+1 ----
    nt = {"generator": str(G), "roots_of_unity": {str(k): str(po.root_of_unity(k)) for k in (0, 1, 2, 3, 10, 16, 20, 24, 119)}, "ntt": []}
    for n in (2, 4, 8, 1024):
        x = [j + 1 for j in range(n)]
        y = po.ntt(x)
        assert po.ntt(y, inverse=True) == x
        if n <= 8:
            assert y == po.ntt_naive(x)
        nt["ntt"].append({"n": n, "out": [str(v) for v in y]} if n <= 8 else
                         {"n": n, "out_head": [str(v) for v in y[:8]], "out_sha256": elems_digest(y)})
    # bigger sizes: checksum only (inputs = SplitMix64 elements)
    for log_n, seed in ((12, 12), (13, 13), (16, 2)):
        x = po.splitmix_elems(seed, 1 << log_n)
        y = po.ntt(x)
        nt["ntt"].append({"n": 1 << log_n, "seed": seed, "in_sha256": elems_digest(x), "out_head": [str(v) for v in y[:4]],
                          "out_sha256": elems_digest(y)})
    dump("ntt_vectors.json", nt)

    # ---- coset LDE ----
    lde = []
    for log_n, seed in ((4, 9), (8, 1), (14, 7)):
        c = po.splitmix_elems(seed, 1 << log_n)
        y = po.coset_lde(c, 2, G)
        if log_n == 4:
            assert y == po.coset_eval_naive(c, 2, G)
        lde.append({"log_n": log_n, "log_expansion": 2, "seed": seed, "offset": str(G), "out_head": [str(v) for v in y[:4]],
                    "out_sha256": elems_digest(y)})
    dump("lde_vectors.json", lde)

    # ---- fold ----
    cw = po.splitmix_elems(11, 64)
    alpha, off, om = po.splitmix_elems(12, 1)[0], G, po.root_of_unity(6)
    dump("fold_vectors.json", {"seed": 11, "L": 64, "alpha": str(alpha), "offset": str(off), "omega": str(om),
                               "out": [str(v) for v in po.fri_fold(cw, alpha, off, om)]})

    # ---- sampling (G3) ----
    dump("sample_vectors.json", [{"digest": d.hex(), "alpha": str(po.sample(d))} for d in
                                 (bytes(32), bytes([255] * 32), po.sha256(b"x"), po.sha256(b"y"), bytes(range(32)))])

    # ---- full C1 transcript: N = 2^10, expansion 4, 17 colinearity tests, offset g, seed 1 ----
    for name, N, seed in (("fri_c1.json", 1 << 10, 1), ("fri_n4096.json", 1 << 12, 4), ("fri_n16384.json", 1 << 14, 1000)):
        cw = po.splitmix_elems(seed, N)
        ps = po.ProofStream()
        roots, alphas, last = po.fri_commit(cw, 4, 17, G, ps)
        ser = ps.serialize()
        obj = {"N": N, "seed": seed, "expansion": 4, "num_colinearity_tests": 17, "offset": str(G),
               "num_rounds": len(roots), "roots": [r.hex() for r in roots], "alphas": [str(a) for a in alphas],
               "last_codeword_sha256": elems_digest(last), "last_len": len(last),
               "stream_sha256": hashlib.sha256(ser).hexdigest(), "stream_len": len(ser),
               "final_fiat_shamir": ps.prover_fiat_shamir().hex()}
        if N == 1 << 10:
            obj["last_codeword"] = [str(v) for v in last]
            obj["stream"] = ser.decode()
        dump(name, obj)

    # ---- C1 prove + verify (G9-G12): n = 256 coefficients (seed 1) -> LDE -> N = 2^10 codeword -> full proof ----
    for name, log_n, seed in (("fri_prove_c1.json", 8, 1), ("fri_prove_n4096.json", 10, 7)):
        c = po.splitmix_elems(seed, 1 << log_n)
        cw = po.coset_lde(c, 2, G)
        ps = po.ProofStream()
        top = po.fri_prove(cw, 4, 17, G, ps)
        ser = ps.serialize()
        ok, why = po.fri_verify(po.ProofStream.deserialize(ser), len(cw), 4, 17, G)
        assert ok, why
        obj = {"log_n": log_n, "seed": seed, "N": len(cw), "expansion": 4, "num_colinearity_tests": 17, "offset": str(G),
               "top_level_indices": top, "num_items": len(ps.items), "stream_sha256": hashlib.sha256(ser).hexdigest(),
               "stream_len": len(ser), "accepted": ok}
        if log_n == 8:
            obj["stream"] = ser.decode()
        dump(name, obj)
    seed32 = bytes(range(32))
    dump("sample_indices.json", [{"seed": seed32.hex(), "size": sz, "reduced_size": rs, "number": k, "indices": po.sample_indices(seed32, sz, rs, k)}
                                 for sz, rs, k in ((512, 128, 17), (1 << 23, 64, 40), (8, 8, 8), (1 << 40, 1 << 20, 3))])

    # ---- C2: n = 2^16 coefficients -> LDE N = 2^18 -> Merkle root (seed 2) ----
    c = po.splitmix_elems(2, 1 << 16)
    y = po.coset_lde(c, 2, G)
    dump("c2_lde_merkle.json", {"log_n": 16, "log_expansion": 2, "seed": 2, "offset": str(G), "lde_sha256": elems_digest(y),
                                "ntt_sha256": elems_digest(po.ntt(c)), "merkle_root": po.merkle_commit_elems(y).hex()})

    poly_goldens()


def poly_goldens():
    """---- polynomial layer: schoolbook product, long division, pointwise zerofier quotient ----
    (also callable alone: python tests/golden/gen_golden.py poly)"""
    EDGE = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (1 << 64) - 1, 1 << 64, 1 << 119, (1 << 127) - 1]
    mul = []
    for seed, na, nb in ((40, 1, 1), (41, 3, 5), (42, 64, 64), (43, 100, 29), (44, 257, 255)):
        a, b = po.splitmix_elems(seed, na), po.splitmix_elems(seed + 100, nb)
        prod = po.poly_mul(a, b)
        e = {"seed_a": seed, "na": na, "seed_b": seed + 100, "nb": nb, "sha256": elems_digest(prod)}
        if len(prod) <= 8:
            e["a"], e["b"], e["product"] = [str(v) for v in a], [str(v) for v in b], [str(v) for v in prod]
        mul.append(e)
    a, b = EDGE[:7], EDGE[3:]
    mul.append({"a": [str(v) for v in a], "b": [str(v) for v in b], "product": [str(v) for v in po.poly_mul(a, b)]})
    dm = []
    for seed, nf, n, c in ((50, 40, 8, 1), (51, 33, 5, 7), (52, 8, 8, 3), (53, 5, 8, 1), (54, 9, 1, P - 1), (55, 64, 16, 0), (56, 300, 7, 1 << 100)):
        f = po.splitmix_elems(seed, nf)
        q, r = po.poly_divmod_zerofier(f, n, c)
        # f == q·(X^n − c) + r
        z = [(-c) % P] + [0] * (n - 1) + [1]
        back = po.poly_mul(q, z) if q else []
        back = [((back[i] if i < len(back) else 0) + (r[i] if i < n else 0)) % P for i in range(max(len(back), n))]
        assert back[:nf] == f and not any(back[nf:])
        dm.append({"seed": seed, "nf": nf, "n": n, "c": str(c), "q": [str(v) for v in q], "r": [str(v) for v in r]})
    zd = []
    for seed, log_n, log_e, c, off in ((60, 3, 2, 1, G), (61, 4, 1, 5, pow(G, 3, P)), (62, 5, 0, 1, G), (63, 0, 4, 9, 7)):
        # numerator that vanishes on the zerofier's roots times a quotient: the result is then a polynomial's evaluations
        n = 1 << log_n
        qpoly = po.splitmix_elems(seed, n)
        f = po.poly_mul(qpoly, [(-c) % P] + [0] * (n - 1) + [1])
        N = n << log_e
        while N < len(f):
            N <<= 1
        cw = po.coset_lde((f + [0] * N)[:N], 0, off)
        quo = po.coset_zerofier_div(cw, log_n, c, off)
        assert quo == po.coset_lde((qpoly + [0] * N)[:N], 0, off)
        zd.append({"seed": seed, "log_n": log_n, "log_N": N.bit_length() - 1, "c": str(c), "offset": str(off), "cw": [str(v) for v in cw],
                   "quotient": [str(v) for v in quo]})
    dump("poly_vectors.json", {"mul": mul, "divmod_zerofier": dm, "coset_zerofier_div": zd})


if __name__ == "__main__":
    if sys.argv[1:] == ["poly"]:
        poly_goldens()
    else:
        main()

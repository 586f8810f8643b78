"""
pyoracle.py — SECOND, independent CPU restatement of the FRI-commit hot path in pure Python
(big ints + hashlib + json).  TEST INFRASTRUCTURE ONLY: imported by tests/ and by
tests/golden/gen_golden.py, never by the product.

It exists to cross-check oracle/stark_oracle.c (two restatements written against the same
spec, sharing no code) and to generate the committed golden vectors.  Everything here is
"obvious math" on Python ints; hashing is hashlib (an implementation independent of both the
reference's Rust crates and of this repo's C/HIP code).

Citations are file:line under /root/reference/src; G1..G8 are the spec-gap definitions of
SURVEY.md §8(c).
"""
import hashlib
import json

P = 1 + 407 * 2**119                     # field.rs:15
assert P == 270497897142230380135924736767050121217
G = 85408008396924667383611388730472331217   # G1: order 2^119
MASK64 = (1 << 64) - 1


# ---- field (field.rs:30-114; results are canonical residues) ----
def fp_new(x): return x % P
def fp_add(a, b): return (a + b) % P
def fp_sub(a, b): return ((a - b) + P) % P
def fp_mul(a, b): return (a * b) % P


def ext_euclid(a, b):                      # utils.rs:3-18 (truncating division == floor for a,b >= 0)
    old_r, r, old_s, s, old_t, t = a, b, 1, 0, 0, 1
    while True:
        q = old_r // r
        old_r, r = r, old_r - q * r
        old_s, s = s, old_s - q * s
        old_t, t = t, old_t - q * t
        if r == 0:
            return old_r, old_s, old_t


def fp_div(a, b):                          # field.rs:97-114
    if b == 0:
        raise ZeroDivisionError("Cannot divide by zero")
    _, inv, _ = ext_euclid(b, P)
    inv = (inv + P) % P
    return (a * inv) % P


def fp_inv(a): return fp_div(1, a)
def fp_pow(a, e): return pow(a, e, P)      # true modpow


# ---- hashes ----
def blake2b512(data): return hashlib.blake2b(bytes(data), digest_size=64).digest()   # merkle.rs:47-54
def sha256(data): return hashlib.sha256(bytes(data)).digest()                        # proof_stream.rs:52-55


# ---- Merkle (merkle.rs) ----
def merkle_commit(leaves):                 # merkle.rs:35-68
    assert len(leaves) > 0 and (len(leaves) & (len(leaves) - 1)) == 0
    hashes = list(leaves)
    while len(hashes) > 1:
        hashes = [blake2b512(hashes[i] + hashes[i + 1]) for i in range(0, len(hashes), 2)]
    return hashes[0]


def merkle_open(leaves, index):            # merkle.rs:70-108
    assert (len(leaves) & (len(leaves) - 1)) == 0 and index < len(leaves)
    results, cur, idx = [], leaves, index
    while len(cur) > 2:
        mid = len(cur) // 2
        if idx < mid:
            results.append(merkle_commit(cur[mid:])); cur = cur[:mid]
        else:
            results.append(merkle_commit(cur[:mid])); cur = cur[mid:]; idx -= mid
    results.append(cur[1 - idx])
    results.reverse()
    return results


def merkle_verify(root, leaf, path, index):   # merkle.rs:114-154
    assert index < (1 << len(path)) and len(path) > 0
    for ph in path:
        leaf = blake2b512(leaf + ph) if index % 2 == 0 else blake2b512(ph + leaf)
        index >>= 1
    return leaf == root


def le16(x): return int(x).to_bytes(16, "little")
def hash_leaf(x): return blake2b512(le16(x))                     # G2
def hash_leaves(elems): return [hash_leaf(x) for x in elems]
def merkle_commit_elems(elems): return merkle_commit(hash_leaves(elems))


# ---- G1 / G7 / G4 / G3 / G5 ----
def root_of_unity(log_n):
    assert log_n <= 119
    return pow(G, 1 << (119 - log_n), P)


def ntt_naive(x, inverse=False):
    n = len(x); log_n = n.bit_length() - 1
    w = root_of_unity(log_n)
    if inverse:
        w = fp_inv(w)
    out = [sum(x[j] * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]
    if inverse:
        ninv = fp_inv(n % P)
        out = [(v * ninv) % P for v in out]
    return out


def ntt(x, inverse=False):
    """recursive radix-2 (a different formulation from the C oracle's iterative DIT)"""
    n = len(x); log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = root_of_unity(log_n)
    if inverse:
        w = fp_inv(w)

    def rec(v, w):
        if len(v) == 1:
            return v
        ev, od = rec(v[0::2], w * w % P), rec(v[1::2], w * w % P)
        h = len(v) // 2
        out = [0] * len(v); wk = 1
        for k in range(h):
            t = od[k] * wk % P
            out[k] = (ev[k] + t) % P
            out[k + h] = (ev[k] - t) % P
            wk = wk * w % P
        return out
    out = rec(list(x), w)
    if inverse:
        ninv = fp_inv(n % P)
        out = [(v * ninv) % P for v in out]
    return out


def coset_lde(coeffs, log_expansion, offset):                    # G7
    n = len(coeffs); N = n << log_expansion
    scaled = [(c * pow(offset, j, P)) % P for j, c in enumerate(coeffs)] + [0] * (N - n)
    return ntt(scaled)


def coset_eval_naive(coeffs, log_expansion, offset):
    n = len(coeffs); N = n << log_expansion
    w = root_of_unity(N.bit_length() - 1)
    out = []
    for i in range(N):
        x = offset * pow(w, i, P) % P
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * x + c) % P
        out.append(acc)
    return out


def fri_fold(cw, alpha, offset, omega):                          # G4, literally
    L = len(cw); two_inv = fp_div(1, 2)
    out = []
    for i in range(L // 2):
        t = fp_div(alpha, fp_mul(offset, pow(omega, i, P)))
        out.append(fp_mul(two_inv, fp_add(fp_mul(fp_add(1, t), cw[i]), fp_mul(fp_sub(1, t), cw[L // 2 + i]))))
    return out


def sample(digest):                                              # G3
    acc = 0
    for b in digest:
        acc = (acc << 8) ^ int(b)
    return acc % P


def fri_num_rounds(N, expansion, num_colinearity_tests):         # G5
    L, r = N, 0
    while L > expansion and 4 * num_colinearity_tests < L:
        L //= 2; r += 1
    return r


# ---- ProofStream (proof_stream.rs) with the G6 item type ----
def item_root(root): return {"MerkleRoot": {"inner": list(root)}}        # merkle.rs:9-13 + G6
def item_codeword(cw): return {"Codeword": [int(v) for v in cw]}         # G6


class ProofStream:
    def __init__(self):
        self.items = []
        self.read_index = 0

    def push(self, item): self.items.append(item)                # proof_stream.rs:25-27

    def pull(self):                                              # proof_stream.rs:29-36
        if self.read_index >= len(self.items):
            raise IndexError("Read index is out of bounds")
        v = self.items[self.read_index]; self.read_index += 1
        return v

    @staticmethod
    def _ser(items): return json.dumps(items, separators=(",", ":")).encode()   # proof_stream.rs:67-69
    def serialize(self): return self._ser(self.items)           # proof_stream.rs:38-40

    @classmethod
    def deserialize(cls, data):                                  # proof_stream.rs:42-49
        ps = cls(); ps.items = json.loads(bytes(data)); return ps

    def prover_fiat_shamir(self): return sha256(self.serialize())                       # proof_stream.rs:51-56
    def verifier_fiat_shamir(self): return sha256(self._ser(self.items[:self.read_index]))  # proof_stream.rs:58-64


def fri_commit(codeword, expansion, num_colinearity_tests, offset, ps):
    N = len(codeword)
    R = fri_num_rounds(N, expansion, num_colinearity_tests)
    omega = root_of_unity(N.bit_length() - 1)
    cw = list(codeword); roots, alphas = [], []
    for r in range(R):
        root = merkle_commit_elems(cw)
        roots.append(root); ps.push(item_root(root))
        if r == R - 1:
            break
        alpha = sample(ps.prover_fiat_shamir()); alphas.append(alpha)
        cw = fri_fold(cw, alpha, offset, omega)
        omega = omega * omega % P; offset = offset * offset % P
    ps.push(item_codeword(cw))
    return roots, alphas, cw


# ---- query phase, prover, verifier (G9-G12, after the upstream tutorial's fri.py) ----
def item_leaves(a, b, c): return {"Leaves": [int(a), int(b), int(c)]}                       # G9
def item_path(path): return {"Path": [{"inner": list(h)} for h in path]}                   # G9: Vec<MerkleHash>


def sample_index(byte_array, size):                              # G10 (tutorial: sample_index)
    acc = 0
    for b in byte_array:
        acc = (acc << 8) ^ int(b)
    return acc % size


def sample_indices(seed, size, reduced_size, number):            # G10 (tutorial: sample_indices; counter as u64 LE)
    assert number <= reduced_size, "cannot sample more indices than available in last codeword"
    indices, reduced_indices, counter = [], [], 0
    while len(indices) < number:
        index = sample_index(blake2b512(bytes(seed) + counter.to_bytes(8, "little")), size)
        reduced_index = index % reduced_size
        counter += 1
        if reduced_index not in reduced_indices:
            indices.append(index); reduced_indices.append(reduced_index)
    return indices


def fri_commit_codewords(codeword, expansion, num_colinearity_tests, offset, ps):
    """fri_commit, keeping every round's codeword (tutorial: commit returns `codewords`)"""
    N = len(codeword)
    R = fri_num_rounds(N, expansion, num_colinearity_tests)
    omega = root_of_unity(N.bit_length() - 1)
    cw = list(codeword); codewords = []
    for r in range(R):
        ps.push(item_root(merkle_commit_elems(cw)))
        codewords.append(cw)
        if r == R - 1:
            break
        alpha = sample(ps.prover_fiat_shamir())
        cw = fri_fold(cw, alpha, offset, omega)
        omega = omega * omega % P; offset = offset * offset % P
    ps.push(item_codeword(cw))
    if R == 0:
        codewords.append(cw)
    return codewords


def fri_query(current, nxt, c_indices, ps):                      # G11 (tutorial: query)
    a_indices = list(c_indices)
    b_indices = [i + len(current) // 2 for i in c_indices]
    for s in range(len(c_indices)):
        ps.push(item_leaves(current[a_indices[s]], current[b_indices[s]], nxt[c_indices[s]]))
    cur_leaves, nxt_leaves = hash_leaves(current), hash_leaves(nxt)
    for s in range(len(c_indices)):
        ps.push(item_path(merkle_open(cur_leaves, a_indices[s])))
        ps.push(item_path(merkle_open(cur_leaves, b_indices[s])))
        ps.push(item_path(merkle_open(nxt_leaves, c_indices[s])))
    return a_indices + b_indices


def fri_prove(codeword, expansion, num_colinearity_tests, offset, ps):     # G11 (tutorial: prove)
    codewords = fri_commit_codewords(codeword, expansion, num_colinearity_tests, offset, ps)
    if len(codewords) < 2:
        return []
    top = sample_indices(ps.prover_fiat_shamir(), len(codewords[1]), len(codewords[-1]), num_colinearity_tests)
    indices = list(top)
    for i in range(len(codewords) - 1):
        indices = [index % (len(codewords[i]) // 2) for index in indices]
        fri_query(codewords[i], codewords[i + 1], indices, ps)
    return top


def fri_verify(ps, N, expansion, num_colinearity_tests, offset):           # G12 (tutorial: verify); returns (ok, reason)
    R = fri_num_rounds(N, expansion, num_colinearity_tests)
    omega = root_of_unity(N.bit_length() - 1)
    try:
        roots, alphas = [], []
        for r in range(R):
            roots.append(bytes(ps.pull()["MerkleRoot"]["inner"]))
            if r < R - 1:
                alphas.append(sample(ps.verifier_fiat_shamir()))
        last = [int(v) for v in ps.pull()["Codeword"]]
    except (KeyError, TypeError, IndexError):
        return False, "malformed commit items"
    L_last = N >> max(R - 1, 0)
    if len(last) != L_last or any(not (0 <= v < P) for v in last):
        return False, "last codeword malformed"
    if R == 0:
        return True, ""
    if merkle_commit_elems(last) != roots[-1]:
        return False, "last codeword is not well formed"
    # degree of the interpolant through (offset_last * omega_last^i, last[i]): coefficients by naive inverse DFT, unscaled
    off_last = pow(offset, 1 << (R - 1), P)
    coeffs = [fp_div(c, pow(off_last, j, P)) for j, c in enumerate(ntt_naive(last, inverse=True))]
    degree = max((j for j, c in enumerate(coeffs) if c), default=-1)
    if degree > max(L_last // expansion, 1) - 1:
        return False, "last codeword does not correspond to polynomial of low enough degree"
    if R < 2:
        return True, ""
    top = sample_indices(ps.verifier_fiat_shamir(), N // 2, L_last, num_colinearity_tests)
    t = num_colinearity_tests
    for r in range(R - 1):
        L = N >> r
        c_indices = [i % (L // 2) for i in top]
        try:
            triples = [ps.pull()["Leaves"] for _ in range(t)]
            for s in range(t):
                ay, by, cy = triples[s]
                ax = fp_mul(offset, pow(omega, c_indices[s], P)); bx = fp_mul(offset, pow(omega, c_indices[s] + L // 2, P)); cx = alphas[r]
                # colinearity of (ax,ay), (bx,by), (cx,cy): cross product of differences
                if fp_mul(fp_sub(by, ay), fp_sub(cx, ax)) != fp_mul(fp_sub(cy, ay), fp_sub(bx, ax)):
                    return False, "colinearity check failure"
            for s in range(t):
                pa = [bytes(h["inner"]) for h in ps.pull()["Path"]]
                pb = [bytes(h["inner"]) for h in ps.pull()["Path"]]
                pc = [bytes(h["inner"]) for h in ps.pull()["Path"]]
                ay, by, cy = triples[s]
                depth = L.bit_length() - 1
                if len(pa) != depth or len(pb) != depth or len(pc) != depth - 1:
                    return False, "path length"
                if not merkle_verify(roots[r], hash_leaf(ay), pa, c_indices[s]):
                    return False, "merkle authentication path verification fails for aa"
                if not merkle_verify(roots[r], hash_leaf(by), pb, c_indices[s] + L // 2):
                    return False, "merkle authentication path verification fails for bb"
                if not merkle_verify(roots[r + 1], hash_leaf(cy), pc, c_indices[s]):
                    return False, "merkle authentication path verification fails for cc"
        except (KeyError, TypeError, IndexError, ValueError):
            return False, "malformed query items"
        omega = omega * omega % P; offset = offset * offset % P
    return True, ""


# ------------------------------------------------------------------ polynomial layer
# The reference crate has no polynomial code; these are the textbook definitions (schoolbook product, long division,
# pointwise quotient), deliberately NOT the NTT / Montgomery's-trick algorithms the GPU uses.  Coefficients lowest degree first.
def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
    return out


def poly_eval(f, x):
    acc = 0
    for c in reversed(f):
        acc = (acc * x + c) % P
    return acc


def poly_divmod(f, d):
    """long division by a polynomial with non-zero leading coefficient: (q, r), len(q) = max(len(f)-deg d, 0), len(r) = deg d"""
    f, dd = list(f), len(d) - 1
    lead_inv = fp_inv(d[-1])
    q = [0] * max(len(f) - dd, 0)
    for k in range(len(f) - 1, dd - 1, -1):
        coef = fp_mul(f[k], lead_inv)
        q[k - dd] = coef
        for j, y in enumerate(d):
            f[k - dd + j] = fp_sub(f[k - dd + j], fp_mul(coef, y))
    r = (f[:dd] + [0] * dd)[:dd]
    return q, r


def poly_divmod_zerofier(f, n, c):
    """f = q·(X^n − c) + r by generic long division"""
    return poly_divmod(f, [fp_sub(0, c)] + [0] * (n - 1) + [1])


def coset_zerofier_div(cw, log_n, c, offset):
    """cw[i] / ((offset·w_N^i)^n − c), n = 2^log_n; ZeroDivisionError where the coset meets a root (field.rs:101-103)"""
    N = len(cw)
    w = root_of_unity(N.bit_length() - 1)
    out, x = [], offset % P
    for v in cw:
        out.append(fp_div(v, fp_sub(pow(x, 1 << log_n, P), c)))
        x = x * w % P
    return out


def splitmix_elems(seed, n):
    s = seed & MASK64; out = []
    for _ in range(n):
        w = []
        for _k in range(2):
            s = (s + 0x9E3779B97F4A7C15) & MASK64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
            w.append(z ^ (z >> 31))
        out.append(((w[1] << 64) | w[0]) % P)
    return out

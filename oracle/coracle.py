"""
coracle.py — ctypes binding of oracle/liboracle.so (the C oracle).  TEST INFRASTRUCTURE ONLY.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Field elements are numpy uint64 arrays of shape (n, 2) = canonical little-endian u128 {lo, hi}.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


lib = _load()
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class I320(C.Structure):
    _fields_ = [("v", C.c_uint64 * 5)]

    @classmethod
    def from_int(cls, x):
        x &= (1 << 320) - 1
        r = cls()
        for i in range(5):
            r.v[i] = (x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF
        return r

    def limbs(self): return [int(self.v[i]) for i in range(5)]

    def to_int(self):
        x = sum(int(self.v[i]) << (64 * i) for i in range(5))
        return x - (1 << 320) if x >> 319 else x


def _sig(name, res, *args):
    f = getattr(lib, name); f.restype = res; f.argtypes = list(args); return f


for _n in ("add", "sub", "mul"):
    _sig(f"or_fp_{_n}", None, u64p, u64p, u64p)
    _sig(f"or_ref_fp_{_n}", None, u64p, u64p, u64p)
_sig("or_fp_div", C.c_int, u64p, u64p, u64p)
_sig("or_ref_fp_div", C.c_int, u64p, u64p, u64p)
_sig("or_fp_inv", C.c_int, u64p, u64p)
_sig("or_fp_pow", None, u64p, u64p, u64p)
_sig("or_ref_fp_pow", None, u64p, u64p, u64p)
_sig("or_fp_new", None, u64p, u64p)
_sig("or_fp_prime", None, u64p)
_sig("or_fp_batch", C.c_int, C.c_int, u64p, u64p, u64p, C.c_size_t)
for _n in ("add", "sub", "mul", "pow"):
    _sig(f"or_i320_{_n}", I320, I320, I320)
_sig("or_i320_neg", I320, I320)
_sig("or_i320_div", C.c_int, I320, I320, C.POINTER(I320))
_sig("or_i320_rem", C.c_int, I320, I320, C.POINTER(I320))
_sig("or_i320_cmp", C.c_int, I320, I320)
_sig("or_ext_euclid", C.c_int, I320, I320, C.POINTER(I320), C.POINTER(I320), C.POINTER(I320))
_sig("or_blake2b512", None, C.c_char_p, C.c_size_t, u8p)
_sig("or_sha256", None, C.c_char_p, C.c_size_t, u8p)
_sig("or_merkle_commit", C.c_int, u8p, C.c_size_t, u8p)
_sig("or_merkle_open", C.c_int, u8p, C.c_size_t, C.c_size_t, u8p, C.POINTER(C.c_size_t))
_sig("or_merkle_verify", C.c_int, u8p, u8p, u8p, C.c_size_t, C.c_size_t)
_sig("or_hash_leaves", None, u64p, C.c_size_t, u8p)
_sig("or_merkle_commit_elems", C.c_int, u64p, C.c_size_t, u8p)
_sig("or_root_of_unity", C.c_int, C.c_uint32, u64p)
_sig("or_generator", None, u64p)
_sig("or_ntt", C.c_int, u64p, C.c_uint32, C.c_int)
_sig("or_ntt_naive", C.c_int, u64p, C.c_uint32, C.c_int, u64p)
_sig("or_coset_lde", C.c_int, u64p, C.c_uint32, C.c_uint32, u64p, u64p)
_sig("or_fri_fold", C.c_int, u64p, C.c_size_t, u64p, u64p, u64p, u64p)
_sig("or_fri_fold_faithful", C.c_int, u64p, C.c_size_t, u64p, u64p, u64p, u64p, C.c_int)
_sig("or_sample", None, u8p, u64p)
_sig("or_fri_num_rounds", C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32)
_sig("or_ps_new", C.c_void_p)
_sig("or_ps_free", None, C.c_void_p)
_sig("or_ps_push_raw", C.c_int, C.c_void_p, C.c_char_p, C.c_size_t)
_sig("or_ps_push_root", C.c_int, C.c_void_p, u8p)
_sig("or_ps_push_codeword", C.c_int, C.c_void_p, u64p, C.c_size_t)
_sig("or_ps_len", C.c_size_t, C.c_void_p)
_sig("or_ps_pull", C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_sig("or_ps_serialize", C.c_size_t, C.c_void_p, u8p, C.c_size_t)
_sig("or_ps_serialize_read", C.c_size_t, C.c_void_p, u8p, C.c_size_t)
_sig("or_ps_prover_fiat_shamir", C.c_int, C.c_void_p, u8p)
_sig("or_ps_verifier_fiat_shamir", C.c_int, C.c_void_p, u8p)
_sig("or_fri_commit", C.c_int, u64p, C.c_size_t, C.c_uint32, C.c_uint32, u64p, C.c_void_p, C.c_int,
     u8p, C.POINTER(C.c_uint32), u64p, C.POINTER(C.c_size_t), u64p)
_sig("or_splitmix_elems", None, C.c_uint64, u64p, C.c_size_t)
_sig("or_set_threads", None, C.c_int)
_sig("or_get_threads", C.c_int)


# ---------------- helpers ----------------
def to_arr(vals):
    """list of python ints -> (n,2) uint64"""
    a = np.empty((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        a[i, 1] = v >> 64
    return a


def to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in arr]


def _p(a): return a.ctypes.data_as(u64p)
def _b(a): return a.ctypes.data_as(u8p)
def _one(v): return to_arr([v])


def _scalar(fn, *vals):
    out = np.zeros((1, 2), dtype=np.uint64)
    ins = [_one(v) for v in vals]
    rc = fn(*[_p(x) for x in ins], _p(out))
    if rc not in (None, 0):
        raise ZeroDivisionError("Cannot divide by zero")
    return to_ints(out)[0]


def fp_add(a, b, ref=False): return _scalar(lib.or_ref_fp_add if ref else lib.or_fp_add, a, b)
def fp_sub(a, b, ref=False): return _scalar(lib.or_ref_fp_sub if ref else lib.or_fp_sub, a, b)
def fp_mul(a, b, ref=False): return _scalar(lib.or_ref_fp_mul if ref else lib.or_fp_mul, a, b)
def fp_div(a, b, ref=False): return _scalar(lib.or_ref_fp_div if ref else lib.or_fp_div, a, b)
def fp_pow(a, e, ref=False): return _scalar(lib.or_ref_fp_pow if ref else lib.or_fp_pow, a, e)
def fp_inv(a): return _scalar(lib.or_fp_inv, a)
def fp_new(a): return _scalar(lib.or_fp_new, a)


def fp_batch(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    bp = _p(np.ascontiguousarray(b, dtype=np.uint64)) if b is not None else None
    rc = lib.or_fp_batch({"add": 0, "sub": 1, "mul": 2, "div": 3, "inv": 3}[op], _p(a), bp, _p(out), a.shape[0])
    if rc == -2:
        raise ZeroDivisionError("Cannot divide by zero")
    assert rc == 0
    return out


def blake2b512(data):
    out = (C.c_uint8 * 64)(); lib.or_blake2b512(bytes(data), len(data), out); return bytes(out)


def sha256(data):
    out = (C.c_uint8 * 32)(); lib.or_sha256(bytes(data), len(data), out); return bytes(out)


def _leaves_arr(leaves):
    if isinstance(leaves, np.ndarray):
        return np.ascontiguousarray(leaves, dtype=np.uint8).reshape(-1, 64)
    return np.frombuffer(b"".join(leaves), dtype=np.uint8).reshape(-1, 64).copy()


def merkle_commit(leaves):
    a = _leaves_arr(leaves); root = np.zeros(64, dtype=np.uint8)
    rc = lib.or_merkle_commit(_b(a), a.shape[0], _b(root))
    if rc: raise ValueError(f"or_merkle_commit rc={rc}")
    return root.tobytes()


def merkle_open(leaves, index):
    a = _leaves_arr(leaves); path = np.zeros((64, 64), dtype=np.uint8); n = C.c_size_t(0)
    rc = lib.or_merkle_open(_b(a), a.shape[0], index, _b(path), C.byref(n))
    if rc: raise IndexError(f"or_merkle_open rc={rc}")
    return [path[i].tobytes() for i in range(n.value)]


def merkle_verify(root, leaf, path, index):
    p = np.frombuffer(b"".join(path), dtype=np.uint8).copy() if path else np.zeros(1, dtype=np.uint8)
    r = np.frombuffer(root, dtype=np.uint8).copy(); l = np.frombuffer(leaf, dtype=np.uint8).copy()
    rc = lib.or_merkle_verify(_b(r), _b(l), _b(p), len(path), index)
    if rc < 0: raise AssertionError(f"or_merkle_verify rc={rc}")
    return bool(rc)


def hash_leaves(elems):
    e = np.ascontiguousarray(elems, dtype=np.uint64); out = np.empty((e.shape[0], 64), dtype=np.uint8)
    lib.or_hash_leaves(_p(e), e.shape[0], _b(out)); return out


def merkle_commit_elems(elems):
    e = np.ascontiguousarray(elems, dtype=np.uint64); root = np.zeros(64, dtype=np.uint8)
    rc = lib.or_merkle_commit_elems(_p(e), e.shape[0], _b(root))
    if rc: raise ValueError(f"rc={rc}")
    return root.tobytes()


def root_of_unity(log_n):
    out = np.zeros((1, 2), dtype=np.uint64)
    if lib.or_root_of_unity(log_n, _p(out)): raise ValueError("log_n > 119")
    return to_ints(out)[0]


def generator():
    out = np.zeros((1, 2), dtype=np.uint64); lib.or_generator(_p(out)); return to_ints(out)[0]


def ntt(data, inverse=False):
    d = np.array(data, dtype=np.uint64, copy=True, order="C")
    log_n = int(d.shape[0]).bit_length() - 1
    assert 1 << log_n == d.shape[0]
    rc = lib.or_ntt(_p(d), log_n, int(inverse)); assert rc == 0
    return d


def ntt_naive(data, inverse=False):
    d = np.ascontiguousarray(data, dtype=np.uint64); out = np.empty_like(d)
    log_n = int(d.shape[0]).bit_length() - 1
    rc = lib.or_ntt_naive(_p(d), log_n, int(inverse), _p(out)); assert rc == 0
    return out


def coset_lde(coeffs, log_expansion, offset):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64)
    log_n = int(c.shape[0]).bit_length() - 1
    out = np.empty((c.shape[0] << log_expansion, 2), dtype=np.uint64)
    rc = lib.or_coset_lde(_p(c), log_n, log_expansion, _p(_one(offset)), _p(out)); assert rc == 0
    return out


def fri_fold(cw, alpha, offset, omega, faithful=None):
    c = np.ascontiguousarray(cw, dtype=np.uint64); out = np.empty((c.shape[0] // 2, 2), dtype=np.uint64)
    if faithful is None:
        rc = lib.or_fri_fold(_p(c), c.shape[0], _p(_one(alpha)), _p(_one(offset)), _p(_one(omega)), _p(out))
    else:
        rc = lib.or_fri_fold_faithful(_p(c), c.shape[0], _p(_one(alpha)), _p(_one(offset)), _p(_one(omega)), _p(out),
                                      1 if faithful == "i320" else 0)
    assert rc == 0, rc
    return out


def sample(digest):
    d = np.frombuffer(bytes(digest), dtype=np.uint8).copy(); out = np.zeros((1, 2), dtype=np.uint64)
    lib.or_sample(_b(d), _p(out)); return to_ints(out)[0]


def fri_num_rounds(N, expansion, t): return int(lib.or_fri_num_rounds(N, expansion, t))


def set_threads(t): lib.or_set_threads(int(t))


def splitmix_elems(seed, n):
    out = np.empty((n, 2), dtype=np.uint64); lib.or_splitmix_elems(seed, _p(out), n); return out


class ProofStream:
    def __init__(self): self.h = lib.or_ps_new()
    def __del__(self):
        if getattr(self, "h", None): lib.or_ps_free(self.h); self.h = None
    def push_raw(self, js): lib.or_ps_push_raw(self.h, bytes(js), len(js))
    def push_root(self, root):
        r = np.frombuffer(bytes(root), dtype=np.uint8).copy(); lib.or_ps_push_root(self.h, _b(r))
    def push_codeword(self, elems):
        e = np.ascontiguousarray(elems, dtype=np.uint64); lib.or_ps_push_codeword(self.h, _p(e), e.shape[0])
    def __len__(self): return lib.or_ps_len(self.h)
    def pull(self):
        p = C.c_void_p(); n = C.c_size_t()
        if lib.or_ps_pull(self.h, C.byref(p), C.byref(n)): raise IndexError("Read index is out of bounds")
        return C.string_at(p.value, n.value)
    def serialize(self, read_only=False):
        fn = lib.or_ps_serialize_read if read_only else lib.or_ps_serialize
        need = fn(self.h, None, 0); buf = np.empty(need, dtype=np.uint8); fn(self.h, _b(buf), need); return buf.tobytes()
    def prover_fiat_shamir(self):
        out = np.zeros(32, dtype=np.uint8); lib.or_ps_prover_fiat_shamir(self.h, _b(out)); return out.tobytes()
    def verifier_fiat_shamir(self):
        out = np.zeros(32, dtype=np.uint8); lib.or_ps_verifier_fiat_shamir(self.h, _b(out)); return out.tobytes()


def fri_commit(codeword, expansion, num_colinearity_tests, offset, ps=None, mode=0):
    """returns (roots[list of 64B], alphas[list int], last_codeword (n,2) uint64, ps)"""
    cw = np.ascontiguousarray(codeword, dtype=np.uint64); N = cw.shape[0]
    ps = ps or ProofStream()
    roots = np.zeros((64, 64), dtype=np.uint8); R = C.c_uint32(0)
    last = np.empty((N, 2), dtype=np.uint64); last_len = C.c_size_t(0)
    alphas = np.zeros((64, 2), dtype=np.uint64)
    rc = lib.or_fri_commit(_p(cw), N, expansion, num_colinearity_tests, _p(_one(offset)), ps.h, mode,
                           _b(roots), C.byref(R), _p(last), C.byref(last_len), _p(alphas))
    assert rc == 0, rc
    return ([roots[i].tobytes() for i in range(R.value)], to_ints(alphas[:max(R.value - 1, 0)]),
            last[:last_len.value].copy(), ps)

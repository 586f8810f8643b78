"""
GPU parity for the polynomial layer (include/stark_hip.h "polynomial layer") and for the batched
inversion behind sh_fp_inv / sh_fp_div: bit-exact against the oracle's per-element inverse (field.rs:97-114 restated),
its schoolbook product and long division, the committed goldens (tests/golden/poly_vectors.json), and — at sizes the
schoolbook oracle cannot reach — evaluation identities at random points.  The reference crate holds no polynomial code
(SURVEY.md §0): parity for this layer is pinned by the two restatements only ("parity unpinned by the reference").
"""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stark_anatomy_amd as sa  # noqa: E402
from oracle import coracle as co  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from stark_anatomy_amd import _lib  # noqa: E402
from stark_anatomy_amd._lib import lib  # noqa: E402
from stark_anatomy_amd.context import _u64p  # noqa: E402

P, G = po.P, po.G
EDGE = [1, 2, P - 1, P - 2, 2**64 - 1, 2**64, 2**127, (1 << 128) % P, 2**64 - 3, 2**32, 2**96 + 5, (P + 1) // 2, (P - 1) // 2, G, 407, 1 << 119]


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


def ints(arr): return sa.array_to_ints(arr)


# ------------------------------------------------------------------ batched inversion
@pytest.mark.parametrize("n", [1, 2, 7, 255, 256, 257, 2047, 2048, 2049, 5000, 1 << 16, (1 << 20) + 3])
def test_batch_inverse_and_divide_vs_oracle(ctx, n):
    """one Fermat inversion per 2048 elements (Montgomery's trick) must give the per-element results of field.rs:97-114"""
    b = co.splitmix_elems(900 + n % 97, n)
    a = co.splitmix_elems(1900 + n % 89, n)
    for i, e in enumerate(EDGE):                                        # edge values sprinkled over lanes and chunks
        b[(i * 131) % n] = (e & (2**64 - 1), e >> 64)
    zero = (b[:, 0] == 0) & (b[:, 1] == 0)
    b[zero] = (1, 0)
    assert np.array_equal(sa.FieldVector(b, ctx).inverse().data, co.fp_batch("inv", b))
    assert np.array_equal((sa.FieldVector(a, ctx) / sa.FieldVector(b, ctx)).data, co.fp_batch("div", a, b))


def test_batch_inverse_scattered_zeros(ctx):
    """field.rs:101-103 per element: SH_EDIVZERO for the call, 0 at the zero divisors, every other quotient intact —
    a zero must not poison the product of its workgroup."""
    n = 1 << 20
    b = co.splitmix_elems(31, n)
    a = co.splitmix_elems(32, n)
    rnd = random.Random(5)
    zpos = sorted({0, 1, 255, 256, 2047, 2048, n - 1} | {rnd.randrange(n) for _ in range(200)})
    b[zpos] = (0, 0)
    fixed = b.copy()
    fixed[zpos] = (1, 0)
    exp_inv, exp_div = co.fp_batch("inv", fixed), co.fp_batch("div", a, fixed)
    exp_inv[zpos] = (0, 0)
    exp_div[zpos] = (0, 0)
    out = np.empty_like(b)
    assert lib.sh_fp_inv(ctx.handle, _u64p(b), _u64p(out), n) == _lib.SH_EDIVZERO
    assert b"Cannot divide by zero" in lib.sh_last_error()
    assert np.array_equal(out, exp_inv)
    assert lib.sh_fp_div(ctx.handle, _u64p(a), _u64p(b), _u64p(out), n) == _lib.SH_EDIVZERO
    assert np.array_equal(out, exp_div)
    with pytest.raises(ZeroDivisionError):
        sa.FieldVector(a, ctx) / sa.FieldVector(b, ctx)
    # all zeros, and a single zero
    z = np.zeros((300, 2), dtype=np.uint64)
    assert lib.sh_fp_inv(ctx.handle, _u64p(z), _u64p(out), 300) == _lib.SH_EDIVZERO and not out[:300].any()
    # and the next call on the same context is clean again
    assert np.array_equal(sa.FieldVector(fixed[:4096], ctx).inverse().data, co.fp_batch("inv", fixed[:4096]))


def test_batch_inverse_split_form_large(ctx):
    """From 2^22 elements up the inversion runs as chunk products -> one small inversion -> finish (api.hip: dev_fp_op).
    n = 2^23 + 5 with zeros in it: x · inv(x) == 1 and (a / x) · x == a at every non-zero position (GPU multiply, itself
    checked against the oracle above), 0 and SH_EDIVZERO at the zeros, and 2^16 scattered positions bit-exact vs the oracle."""
    n = (1 << 23) + 5
    b = co.splitmix_elems(77, n)
    a = co.splitmix_elems(78, n)
    for i, e in enumerate(EDGE):
        b[(i * 1000003) % n] = (e & (2**64 - 1), e >> 64)
    rnd = random.Random(9)
    zpos = np.array(sorted({0, 2047, 2048, n - 1, n - 6} | {rnd.randrange(n) for _ in range(50)}))
    b[(b[:, 0] == 0) & (b[:, 1] == 0)] = (1, 0)
    b[zpos] = (0, 0)
    inv = np.empty_like(b); quo = np.empty_like(b)
    assert lib.sh_fp_inv(ctx.handle, _u64p(b), _u64p(inv), n) == _lib.SH_EDIVZERO
    assert lib.sh_fp_div(ctx.handle, _u64p(a), _u64p(b), _u64p(quo), n) == _lib.SH_EDIVZERO
    assert not inv[zpos].any() and not quo[zpos].any()
    one = np.zeros_like(b); one[:, 0] = 1
    one[zpos] = (0, 0)
    assert np.array_equal((sa.FieldVector(b, ctx) * sa.FieldVector(inv, ctx)).data, one)
    exp = a.copy(); exp[zpos] = (0, 0)
    assert np.array_equal((sa.FieldVector(quo, ctx) * sa.FieldVector(b, ctx)).data, exp)
    pick = np.array(sorted({rnd.randrange(n) for _ in range(1 << 16)} - set(zpos.tolist())))
    assert np.array_equal(inv[pick], co.fp_batch("inv", np.ascontiguousarray(b[pick])))
    assert np.array_equal(quo[pick], co.fp_batch("div", np.ascontiguousarray(a[pick]), np.ascontiguousarray(b[pick])))
    clean = b.copy(); clean[zpos] = (3, 0)
    assert lib.sh_fp_inv(ctx.handle, _u64p(clean), _u64p(inv), n) == 0        # no zero: clean status on the same context


# ------------------------------------------------------------------ product through the NTT
def test_poly_mul_goldens_and_schoolbook(ctx, golden):
    for v in golden("poly_vectors.json")["mul"]:
        if "seed_a" in v:
            a, b = po.splitmix_elems(v["seed_a"], v["na"]), po.splitmix_elems(v["seed_b"], v["nb"])
        else:
            a, b = [int(x) for x in v["a"]], [int(x) for x in v["b"]]
        assert ints(sa.poly_mul(a, b, ctx)) == po.poly_mul(a, b)
    rnd = random.Random(9)
    for na, nb in ((1, 1), (1, 17), (2, 2), (3, 6), (33, 32), (129, 1), (500, 300), (1024, 1025)):
        a = [rnd.randrange(P) for _ in range(na)]; b = [rnd.randrange(P) for _ in range(nb)]
        assert ints(sa.poly_mul(a, b, ctx)) == po.poly_mul(a, b)
    assert ints(sa.poly_mul([0, 0, 0], [5, 6], ctx)) == [0, 0, 0, 0]      # no trimming, zero polynomial included


def test_poly_mul_large_by_evaluation(ctx):
    """2^19 x (2^19 + 1) coefficients (product length 2^20: the padded transform is exactly full) and a ragged pair:
    (a·b)(z) == a(z)·b(z) at random z; commutes; the device-resident form gives the same bytes."""
    for na, nb in (((1 << 19), (1 << 19) + 1), (300001, 12345)):
        a, b = co.splitmix_elems(71, na), co.splitmix_elems(72, nb)
        prod = sa.poly_mul(a, b, ctx)
        assert prod.shape[0] == na + nb - 1
        ai, bi, pi = ints(a), ints(b), ints(prod)
        for z in (2, G, 0x123456789abcdef123456789 % P):
            assert po.poly_eval(ai, z) * po.poly_eval(bi, z) % P == po.poly_eval(pi, z)
        assert np.array_equal(sa.poly_mul(b, a, ctx), prod)
        da, db = ctx.alloc(na * 16).upload(a), ctx.alloc(nb * 16).upload(b)
        dout = ctx.alloc((na + nb - 1) * 16)
        sa.poly.dev_poly_mul(ctx, da, na, db, nb, dout)
        assert np.array_equal(dout.download(np.uint64, (na + nb - 1, 2)), prod)
        for d in (da, db, dout):
            d.free()


def test_poly_mul_argument_errors(ctx):
    out = np.zeros((4, 2), dtype=np.uint64)
    a = np.ones((2, 2), dtype=np.uint64)
    assert lib.sh_poly_mul(ctx.handle, _u64p(a), 0, _u64p(a), 2, _u64p(out)) == _lib.SH_EINVAL
    assert lib.sh_poly_mul(ctx.handle, None, 2, _u64p(a), 2, _u64p(out)) == _lib.SH_EINVAL
    assert lib.sh_poly_mul(ctx.handle, _u64p(a), (1 << 28) + 1, _u64p(a), 2, _u64p(out)) == _lib.SH_EINVAL


# ------------------------------------------------------------------ zerofier quotients
def test_divmod_zerofier_goldens_and_long_division(ctx, golden):
    for v in golden("poly_vectors.json")["divmod_zerofier"]:
        f = po.splitmix_elems(v["seed"], v["nf"])
        q, r = sa.poly_divmod_zerofier(f, v["n"], int(v["c"]), ctx)
        assert ints(q) == [int(x) for x in v["q"]] and ints(r) == [int(x) for x in v["r"]]
    rnd = random.Random(3)
    for nf, n in ((1, 1), (2, 1), (100, 1), (100, 3), (257, 256), (256, 256), (255, 256), (1000, 64), (4097, 1024)):
        f = [rnd.randrange(P) for _ in range(nf)]
        c = rnd.randrange(P)
        q, r = sa.poly_divmod_zerofier(f, n, c, ctx)
        eq, er = po.poly_divmod_zerofier(f, n, c)
        assert ints(q) == eq and ints(r) == er


def test_divmod_zerofier_large_by_evaluation(ctx):
    """nf = 2^22 + 5 coefficients by X^(2^20) − c: f(z) == q(z)·(z^n − c) + r(z), and an exact multiple leaves r = 0."""
    nf, n, c = (1 << 22) + 5, 1 << 20, 0xabcdef
    f = co.splitmix_elems(81, nf)
    q, r = sa.poly_divmod_zerofier(f, n, c, ctx)
    assert q.shape[0] == nf - n and r.shape[0] == n
    fi, qi, ri = ints(f), ints(q), ints(r)
    for z in (3, G):
        assert po.poly_eval(fi, z) == (po.poly_eval(qi, z) * (pow(z, n, P) - c) + po.poly_eval(ri, z)) % P
    n2 = 1 << 10
    qq = co.splitmix_elems(82, 5000)
    zer = np.zeros((n2 + 1, 2), dtype=np.uint64)
    zer[0] = ((P - 7) & (2**64 - 1), (P - 7) >> 64)
    zer[n2] = (1, 0)
    prod = sa.poly_mul(qq, zer, ctx)                                      # qq · (X^1024 − 7)
    q2, r2 = sa.poly_divmod_zerofier(prod, n2, 7, ctx)
    assert np.array_equal(q2, qq) and not r2.any()


def test_coset_zerofier_div_goldens(ctx, golden):
    for v in golden("poly_vectors.json")["coset_zerofier_div"]:
        cw = [int(x) for x in v["cw"]]
        out = sa.coset_zerofier_div(cw, v["log_n"], int(v["c"]), int(v["offset"]), ctx)
        assert ints(out) == [int(x) for x in v["quotient"]]
    rnd = random.Random(4)
    for log_N, log_n in ((0, 0), (1, 0), (1, 1), (6, 2), (9, 9), (12, 5)):
        cw = [rnd.randrange(P) for _ in range(1 << log_N)]
        c, off = rnd.randrange(P), rnd.randrange(1, P)
        assert ints(sa.coset_zerofier_div(cw, log_n, c, off, ctx)) == po.coset_zerofier_div(cw, log_n, c, off)


def test_coset_zerofier_div_meets_a_root(ctx):
    """offset 1: the evaluation domain contains the roots of X^n − 1 -> 'Cannot divide by zero' (field.rs:101-103)"""
    with pytest.raises(ZeroDivisionError):
        sa.coset_zerofier_div(list(range(1, 65)), 4, 1, 1, ctx)
    out = np.zeros((4, 2), dtype=np.uint64)
    one = np.array([[1, 0]], dtype=np.uint64)
    assert lib.sh_coset_zerofier_div(ctx.handle, _u64p(out), 2, 3, _u64p(one), _u64p(one), _u64p(out)) == _lib.SH_EINVAL   # n > N


def test_quotient_pipeline_2_22(ctx):
    """The use the layer is for, at size: f = q·(X^n − 1) with deg q < 3n, n = 2^20 -> LDE of f on the coset g·<w_N>,
    N = 2^22 -> pointwise zerofier quotient -> equals the LDE of q.  All on the device; compared as whole arrays."""
    log_n, log_N = 20, 22
    n, N = 1 << log_n, 1 << log_N
    q = co.splitmix_elems(91, 3 * n)
    zer = np.zeros((n + 1, 2), dtype=np.uint64)
    zer[0] = ((P - 1) & (2**64 - 1), (P - 1) >> 64)
    zer[n] = (1, 0)
    f = sa.poly_mul(q, zer, ctx)                                          # 4n coefficients exactly
    assert f.shape[0] == N
    cw = sa.coset_lde(f, 0, G, ctx)
    quo = sa.coset_zerofier_div(cw, log_n, 1, G, ctx)
    qpad = np.zeros((N, 2), dtype=np.uint64)
    qpad[: 3 * n] = q
    assert np.array_equal(quo, sa.coset_lde(qpad, 0, G, ctx))

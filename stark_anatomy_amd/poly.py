"""
poly.py — the polynomial layer next to the FRI commit (include/stark_hip.h "polynomial layer").
The reference crate has no polynomial type; a polynomial here is its coefficient array, lowest degree
first, never trimmed.  Everything runs on the GPU: products through the NTT, zerofier quotients through one batched
inversion (Montgomery's trick) or a strided Horner walk.
"""
import numpy as np

from . import _lib
from ._lib import check, lib
from .context import _u64p, default_context, device_ptr, elems_to_array, scalar
from .field import GENERATOR


def _chk(rc):
    if rc == _lib.SH_EDIVZERO:
        raise ZeroDivisionError("Cannot divide by zero")          # field.rs:102 panic!
    check(rc)


def poly_mul(a, b, ctx=None):
    """coefficients of a·b (len(a) + len(b) - 1 of them)"""
    ctx = ctx or default_context()
    a, b = elems_to_array(a), elems_to_array(b)
    out = np.empty((a.shape[0] + b.shape[0] - 1, 2), dtype=np.uint64)
    check(lib.sh_poly_mul(ctx.handle, _u64p(a), a.shape[0], _u64p(b), b.shape[0], _u64p(out)))
    return out


def coset_zerofier_div(codeword, log_n, c=1, offset=GENERATOR, ctx=None):
    """codeword[i] / ((offset·w_N^i)^(2^log_n) - c) for the evaluations of a polynomial on the coset offset·<w_N>"""
    ctx = ctx or default_context()
    cw = elems_to_array(codeword)
    N = cw.shape[0]
    log_N = N.bit_length() - 1
    if N <= 0 or (1 << log_N) != N:
        raise ValueError("length must be a power of two")
    out = np.empty_like(cw)
    _chk(lib.sh_coset_zerofier_div(ctx.handle, _u64p(cw), log_N, log_n, _u64p(scalar(offset)), _u64p(scalar(c)), _u64p(out)))
    return out


def poly_divmod_zerofier(f, n, c=1, ctx=None):
    """(q, r) with f = q·(X^n - c) + r, deg r < n; q has max(len(f) - n, 0) coefficients, r has n"""
    ctx = ctx or default_context()
    f = elems_to_array(f)
    nf = f.shape[0]
    q = np.empty((max(nf - n, 0), 2), dtype=np.uint64)
    r = np.empty((n, 2), dtype=np.uint64)
    check(lib.sh_poly_divmod_zerofier(ctx.handle, _u64p(f), nf, n, _u64p(scalar(c)), _u64p(q) if q.shape[0] else None, _u64p(r)))
    return q, r


def dev_poly_mul(ctx, d_a, na, d_b, nb, d_out):
    check(lib.sh_dev_poly_mul(ctx.handle, device_ptr(d_a), na, device_ptr(d_b), nb, device_ptr(d_out)))


def dev_coset_zerofier_div(ctx, d_cw, log_N, log_n, c, offset, d_out):
    _chk(lib.sh_dev_coset_zerofier_div(ctx.handle, device_ptr(d_cw), log_N, log_n, _u64p(scalar(offset)), _u64p(scalar(c)), device_ptr(d_out)))


def dev_poly_divmod_zerofier(ctx, d_f, nf, n, c, d_q, d_r):
    check(lib.sh_dev_poly_divmod_zerofier(ctx.handle, device_ptr(d_f), nf, n, _u64p(scalar(c)), device_ptr(d_q) if d_q is not None else None, device_ptr(d_r)))

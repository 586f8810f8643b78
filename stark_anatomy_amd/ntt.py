"""
ntt.py — NTT / INTT / coset low-degree extension on the GPU: natural order in
and out, X[k] = sum_j x[j]·w^(jk), w = g^(2^119/n).  Host arrays in, host arrays out; the
device-resident forms take DeviceBuffer / torch tensors (see Context, sh_dev_ntt, sh_dev_coset_lde).
"""
import numpy as np

from ._lib import check, lib
from .context import _u64p, default_context, device_ptr, elems_to_array, scalar
from .field import GENERATOR


def _log2(n):
    l = int(n).bit_length() - 1
    if n <= 0 or (1 << l) != n:
        raise ValueError("length must be a power of two")
    return l


def ntt(values, inverse=False, ctx=None):
    ctx = ctx or default_context()
    a = np.array(elems_to_array(values), copy=True)
    check(lib.sh_ntt(ctx.handle, _u64p(a), _log2(a.shape[0]), int(bool(inverse))))
    return a


def intt(values, ctx=None): return ntt(values, inverse=True, ctx=ctx)


def coset_lde(coeffs, log_expansion=2, offset=GENERATOR, ctx=None):
    ctx = ctx or default_context()
    c = elems_to_array(coeffs)
    out = np.empty((c.shape[0] << log_expansion, 2), dtype=np.uint64)
    check(lib.sh_coset_lde(ctx.handle, _u64p(c), _log2(c.shape[0]), log_expansion, _u64p(scalar(offset)), _u64p(out)))
    return out


def dev_ntt(ctx, d_data, log_n, inverse=False):
    check(lib.sh_dev_ntt(ctx.handle, device_ptr(d_data), log_n, int(bool(inverse))))


def dev_coset_lde(ctx, d_coeffs, log_n, log_expansion, offset, d_out):
    check(lib.sh_dev_coset_lde(ctx.handle, device_ptr(d_coeffs), log_n, log_expansion, _u64p(scalar(offset)), device_ptr(d_out)))


def dev_coset_lde_batch(ctx, d_coeffs, batch, log_n, log_expansion, offset, d_out):
    """`batch` polynomials of 2^log_n coefficients back to back -> their coset extensions back to back (sh_dev_coset_lde_batch)"""
    check(lib.sh_dev_coset_lde_batch(ctx.handle, device_ptr(d_coeffs), batch, log_n, log_expansion, _u64p(scalar(offset)), device_ptr(d_out)))

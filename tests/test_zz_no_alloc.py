"""
The allocation-free entry point sh_dev_merkle_commit_elems_noalloc (include/stark_hip.h; no counterpart in the
reference) against the entry point it is a variant of, sh_dev_merkle_commit_elems: bit-identical roots, through the
C ABI, on two calls in a row, the second with a smaller input (the fixed __device__ scratch keeps the first call's
digests).  Collected last (file name) so a failure here cannot hide the other GPU tests behind `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _roots(ctx, lib, check, u8p, d_buf, L):
    a = np.zeros(64, dtype=np.uint8)
    b = np.zeros(64, dtype=np.uint8)
    check(lib.sh_dev_merkle_commit_elems_noalloc(ctx.handle, d_buf.ptr, L, u8p(a)))
    check(lib.sh_dev_merkle_commit_elems(ctx.handle, d_buf.ptr, L, u8p(b)))
    return a.tobytes(), b.tobytes()


def test_noalloc_merkle_commit_matches_allocating_entry_point():
    import stark_anatomy_amd as sa
    from stark_anatomy_amd import _lib
    from stark_anatomy_amd.context import _u8p, _u64p
    from oracle import coracle as co            # the checker, for one size the oracle finishes quickly

    lib, check = _lib.lib, _lib.check
    ctx = sa.Context(0)
    try:
        big, small = 1 << 20, 1 << 13
        cw = co.splitmix_elems(77, big)
        d_big = ctx.alloc(big * 16).upload(cw)
        # second input: different values, smaller — must not see anything the first call left in the scratch
        cw2 = co.splitmix_elems(78, small)
        d_small = ctx.alloc(small * 16).upload(cw2)
        r_na, r_ref = _roots(ctx, lib, check, _u8p, d_big, big)
        assert r_na == r_ref, "2^20: no-alloc root differs from sh_dev_merkle_commit_elems"
        r_na2, r_ref2 = _roots(ctx, lib, check, _u8p, d_small, small)
        assert r_na2 == r_ref2, "2^13 after 2^20: no-alloc root differs"
        assert r_na2 == bytes(co.merkle_commit_elems(cw2)), "2^13: root differs from the oracle"
        # every size class: direct quad (<= 2^9 digests), two quad launches, K = 1..3 subtrees; descending sizes
        for lg in (20, 19, 18, 17, 10, 9, 3, 1, 0):
            a, b = _roots(ctx, lib, check, _u8p, d_big, 1 << lg)
            assert a == b, f"2^{lg}: roots differ"
        # limits and contract violations
        out = np.zeros(64, dtype=np.uint8)
        d_over = ctx.alloc((1 << 21) * 16)
        assert lib.sh_dev_merkle_commit_elems_noalloc(ctx.handle, d_over.ptr, 1 << 21, _u8p(out)) == _lib.SH_EINVAL
        assert b"exceeds the fixed scratch" in lib.sh_last_error()
        assert lib.sh_dev_merkle_commit_elems_noalloc(ctx.handle, d_big.ptr, 3000, _u8p(out)) == _lib.SH_EINVAL
        assert lib.sh_dev_merkle_commit_elems_noalloc(ctx.handle, None, 16, _u8p(out)) == _lib.SH_EINVAL
        for d in (d_big, d_small, d_over):
            d.free()
    finally:
        ctx.close()

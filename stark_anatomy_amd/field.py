"""
field.py — mirror of the reference's `field` module (src/field/types/field.rs).

  FieldElement  : the reference's scalar type — new / pow / is_zero / value and + - * /
                  (field.rs:29-114).  Single values are evaluated by the library's host scalar
                  routine (sh_fp_scalar: the same fp128 Montgomery code the kernels use, compiled
                  for the host); a kernel launch per scalar would be all latency.
  FieldVector   : the batched form the GPU path is for — elementwise + - * / and inverse on the
                  device (sh_fp_add/sub/mul/div/inv), canonical u128 in, canonical u128 out.
"""
import numpy as np

from . import _lib
from ._lib import check, lib
from .context import _u64p, array_to_ints, default_context, elems_to_array, scalar

PRIME = 270497897142230380135924736767050121217          # field.rs:15  (1 + 407·2^119)
GENERATOR = 85408008396924667383611388730472331217       # SURVEY.md §8c G1, order 2^119


def _scalar_op(op, a, b):
    out = np.zeros((1, 2), dtype=np.uint64)
    rc = lib.sh_fp_scalar(op, _u64p(scalar(a)), _u64p(scalar(b)), _u64p(out))
    if rc == _lib.SH_EDIVZERO:
        raise ZeroDivisionError("Cannot divide by zero")          # field.rs:102 panic!
    check(rc)
    return array_to_ints(out)[0]


class FieldElement:
    __slots__ = ("_value",)
    prime = PRIME

    def __init__(self, num):                 # FieldElement::new, field.rs:30-35
        self._value = int(num) % PRIME

    @classmethod
    def new(cls, num): return cls(num)
    def value(self): return self._value      # field.rs:52-54
    def is_zero(self): return self._value == 0   # field.rs:48-50

    def pow(self, exponent):
        """True modular exponentiation.  The reference's pow (field.rs:37-46) is a plain integer power
        reduced once, correct only while base^exp < 2^319; on that range the two agree."""
        e = exponent.value() if isinstance(exponent, FieldElement) else int(exponent)
        return FieldElement(_scalar_op(4, self._value, e))

    def __add__(self, o): return FieldElement(_scalar_op(0, self._value, o._value))       # field.rs:57-68
    def __sub__(self, o): return FieldElement(_scalar_op(1, self._value, o._value))       # field.rs:70-82
    def __mul__(self, o): return FieldElement(_scalar_op(2, self._value, o._value))       # field.rs:84-95
    def __truediv__(self, o): return FieldElement(_scalar_op(3, self._value, o._value))   # field.rs:97-114
    def __eq__(self, o): return isinstance(o, FieldElement) and o._value == self._value
    def __hash__(self): return hash(self._value)
    def __repr__(self): return f"FieldElement {{ value: {self._value}, prime: {PRIME} }}"


ZERO, ONE = FieldElement(0), FieldElement(1)


def root_of_unity(log_n):
    out = np.zeros((1, 2), dtype=np.uint64)
    check(lib.sh_root_of_unity(log_n, _u64p(out)))
    return FieldElement(array_to_ints(out)[0])


def generator():
    out = np.zeros((1, 2), dtype=np.uint64)
    check(lib.sh_generator(_u64p(out)))
    return FieldElement(array_to_ints(out)[0])


class FieldVector:
    """A host-resident vector of canonical field elements whose arithmetic runs on the GPU."""

    def __init__(self, values, ctx=None):
        self.ctx = ctx or default_context()
        self.data = elems_to_array(values)

    def __len__(self): return self.data.shape[0]
    def to_ints(self): return array_to_ints(self.data)

    def _bin(self, fn, other):
        assert len(other) == len(self)
        out = np.empty_like(self.data)
        rc = fn(self.ctx.handle, _u64p(self.data), _u64p(other.data), _u64p(out), len(self))
        if rc == _lib.SH_EDIVZERO:
            raise ZeroDivisionError("Cannot divide by zero")
        check(rc)
        return FieldVector(out, self.ctx)

    def __add__(self, o): return self._bin(lib.sh_fp_add, o)
    def __sub__(self, o): return self._bin(lib.sh_fp_sub, o)
    def __mul__(self, o): return self._bin(lib.sh_fp_mul, o)
    def __truediv__(self, o): return self._bin(lib.sh_fp_div, o)

    def inverse(self):
        out = np.empty_like(self.data)
        rc = lib.sh_fp_inv(self.ctx.handle, _u64p(self.data), _u64p(out), len(self))
        if rc == _lib.SH_EDIVZERO:
            raise ZeroDivisionError("Cannot divide by zero")
        check(rc)
        return FieldVector(out, self.ctx)

    @classmethod
    def new(cls, raw_u128_values, ctx=None):
        """FieldElement::new over a batch: reduce arbitrary u128 values mod p on the device."""
        ctx = ctx or default_context()
        a = elems_to_array(raw_u128_values)
        out = np.empty_like(a)
        check(lib.sh_fp_new(ctx.handle, _u64p(a), _u64p(out), a.shape[0]))
        return cls(out, ctx)

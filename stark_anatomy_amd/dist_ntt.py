"""
dist_ntt.py — ONE transform / ONE commit spread over several GPUs (BASELINE.json C5): the
four-step NTT with all-to-all transposes and the FRI commit of a codeword that is sharded over ranks.
One process per GPU under torch.distributed ("nccl" == RCCL over xGMI; "gloo" for CPU rehearsals and for
several ranks on one card, where tensors are staged through host memory).

Data model.  A vector of n = 2^log_n field elements is spread in natural order: rank r of P holds the contiguous
slice [r·n/P, (r+1)·n/P) as an int64 tensor of shape (n/P, 2) ({lo, hi} words of the canonical value, the bytes the
C ABI uses; torch has no uint64 arithmetic and none is needed — torch only moves and permutes).

Four-step NTT (n = n1·n2, j = j1·n2 + j2, k = k1 + n1·k2):
  A  exchange-transpose  rows j1 -> columns j2      [all-to-all]      (skipped for input_layout="columns")
  B  n2/P local NTTs of n1 points over j1, then  ·w_n^(j2·k1)         [sh_dev_ntt_batch, sh_dev_twiddle_scale]
  C  exchange-transpose  columns j2 -> rows k1      [all-to-all]
  D  n1/P local NTTs of n2 points over j2                             [sh_dev_ntt_batch]
  E  exchange-transpose  rows k1 -> natural order   [all-to-all]      (skipped for output_layout="rows")
Every all-to-all moves the rank's whole slice once (16·n/P bytes out, 16·n/P in); xGMI is point-to-point, so an
all-to-all is the collective it is best at (every link busy, one hop).

Sharded FRI commit.  Round r works on a codeword of length L.  Rank g holds half-slices g and P+g (h = L/(2P)
elements each), so cw[i] and cw[L/2+i] are on the same card: it hashes both into Merkle sub-roots (GPU), the 2P
sub-roots are all-gathered (128 B per rank) and reduced to the round's root on every rank, every rank keeps an
identical transcript replica and derives the same alpha, and the fold of the rank's h outputs is local.  The h
outputs are slice g of the next codeword; before the next round a point-to-point regroup moves its two halves to
ranks 2g mod P and (2g+1) mod P.  When half-slices drop under `min_half` elements (default 2^16: from there a round is
latency, not work, and one GPU walks the remaining rounds in under a millisecond) the codeword is all-gathered and every
rank finishes the remaining rounds locally (the round count depends on the current length only).

`engine` is the local compute.  `GpuEngine` is the product engine (HIP kernels through the C ABI, no fallback);
tests inject a CPU engine with the same methods to rehearse the exchange logic on gloo ranks.
"""
import contextlib
import ctypes as C
import os

import numpy as np

_MASK = 0xFFFFFFFFFFFFFFFF


def _dist():
    import torch.distributed as dist
    return dist


def _world_rank(group):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _needs_staging(t, group):
    return t.is_cuda and _dist().get_backend(group) != "nccl"


# ------------------------------------------------------------------------------------------------ GPU engine
class GpuEngine:
    """local steps on this rank's MI355X.  All kernels and all torch work run on one non-default HIP stream."""

    def __init__(self, device=None):
        import torch
        from ._lib import lib  # noqa: F401  (ImportError if the HIP library is not built: there is no CPU path)
        from .context import Context
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)
        if not torch.cuda.is_available():
            raise RuntimeError("GpuEngine needs a HIP device; stark_anatomy_amd has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", dev)
        self.stream = torch.cuda.Stream(self.device)
        self.ctx = Context(dev, stream=self.stream.cuda_stream)

    def scope(self):
        return self.torch.cuda.stream(self.stream)

    def from_host(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 2)
        with self.scope():
            return self.torch.from_numpy(a.view(np.int64)).to(self.device)

    def to_host(self, t):
        with self.scope():
            h = t.contiguous().cpu()
        return h.numpy().view(np.uint64).reshape(-1, 2)

    def _chk(self, rc):
        from ._lib import check
        check(rc)

    @staticmethod
    def _sc(v):
        from .context import _u64p, scalar
        s = scalar(int(v))
        return s, _u64p(s)

    def ntt_rows(self, t, log_len, inverse=False):
        """in place: every contiguous run of 2^log_len elements of `t` is transformed."""
        from ._lib import lib
        assert t.is_contiguous() and (t.numel() // 2) % (1 << log_len) == 0
        self._chk(lib.sh_dev_ntt_batch(self.ctx.handle, t.data_ptr(), (t.numel() // 2) >> log_len, log_len, int(bool(inverse))))
        return t

    def twiddle(self, t, log_cols, row0, col0, log_n, inverse=False):
        from ._lib import lib
        assert t.is_contiguous()
        self._chk(lib.sh_dev_twiddle_scale(self.ctx.handle, t.data_ptr(), (t.numel() // 2) >> log_cols, log_cols, row0, col0, log_n, int(bool(inverse))))
        return t

    def scale_powers(self, t, j0, base):
        from ._lib import lib
        assert t.is_contiguous()
        keep, p = self._sc(base)
        self._chk(lib.sh_dev_scale_powers(self.ctx.handle, t.data_ptr(), t.numel() // 2, j0, p))
        return t

    def zeros(self, m):
        """(m, 2) elements of zeros (hipMemsetAsync, no fill kernel)"""
        from ._lib import lib
        with self.scope():
            t = self.torch.empty((m, 2), dtype=self.torch.int64, device=self.device)
        self._chk(lib.sh_dev_zero(self.ctx.handle, t.data_ptr(), m * 16))
        return t

    def transpose(self, x):
        """x: (rows, cols, 2) contiguous -> (cols, rows, 2): the pack of an exchange (k_transpose16, LDS tiles)"""
        from ._lib import lib
        assert x.is_contiguous() and x.dim() == 3
        rows, cols = x.shape[0], x.shape[1]
        with self.scope():
            out = self.torch.empty((cols, rows, 2), dtype=x.dtype, device=x.device)
        self._chk(lib.sh_dev_transpose(self.ctx.handle, x.data_ptr(), rows, cols, out.data_ptr()))
        return out

    def interleave(self, recv, P, b, a):
        """recv: [src s][col c][row r] -> (b, P·a, 2) = [c][s][r]: the unpack after an exchange (k_interleave_blocks)"""
        from ._lib import lib
        assert recv.is_contiguous() and recv.numel() == 2 * P * b * a
        with self.scope():
            out = self.torch.empty((b, P * a, 2), dtype=recv.dtype, device=recv.device)
        self._chk(lib.sh_dev_interleave_blocks(self.ctx.handle, recv.data_ptr(), P, b, a, out.data_ptr()))
        return out

    def subroot_pair(self, lo, hi):
        """Merkle roots over the leaf hashes of `lo` and of `hi` -> 128 bytes that stay on the device (no synchronise)."""
        from ._lib import lib
        assert lo.is_contiguous() and hi.is_contiguous()
        with self.scope():
            pair = self.torch.empty(128, dtype=self.torch.uint8, device=self.device)
        self._chk(lib.sh_dev_merkle_root_elems(self.ctx.handle, lo.data_ptr(), lo.numel() // 2, pair.data_ptr()))
        self._chk(lib.sh_dev_merkle_root_elems(self.ctx.handle, hi.data_ptr(), hi.numel() // 2, pair.data_ptr() + 64))
        return pair

    def top_root_pairs(self, pairs, P):
        """pairs: the all-gathered (P, 2, 64) sub-roots on the device -> the round's root (the round's one host touch)"""
        from ._lib import lib
        from .context import _u8p
        assert pairs.is_contiguous() and pairs.numel() == 128 * P
        root = np.zeros(64, dtype=np.uint8)
        self._chk(lib.sh_dev_merkle_commit_pairs(self.ctx.handle, pairs.data_ptr(), P, _u8p(root)))
        return root.tobytes()

    def fold_range(self, lo, hi, i0, alpha, offset, omega):
        from . import _lib
        assert lo.is_contiguous() and hi.is_contiguous() and lo.shape == hi.shape
        with self.scope():
            out = self.torch.empty_like(lo)
        k1, a = self._sc(alpha)
        k2, o = self._sc(offset)
        k3, w = self._sc(omega)
        rc = _lib.lib.sh_dev_fri_fold_range(self.ctx.handle, lo.data_ptr(), hi.data_ptr(), lo.numel() // 2, i0, a, o, w, out.data_ptr())
        if rc == _lib.SH_EDIVZERO:
            raise ZeroDivisionError("Cannot divide by zero")
        self._chk(rc)
        return out

    def commit_tail(self, cw, L, offset, expansion, t, proof_stream):
        """the remaining rounds of a commit whose current codeword (L elements, whole) is on this rank."""
        from .fri import Fri
        roots, last = Fri(offset, None, L, expansion, t, ctx=self.ctx).commit(cw.contiguous(), proof_stream)
        return [h.as_bytes() for h in roots], last

    def new_proof_stream(self):
        from .proof_stream import ProofStream
        return ProofStream()

    def root_of_unity(self, log_n):
        from .field import root_of_unity
        return root_of_unity(log_n).value()

    def push_root_and_sample(self, ps, root):
        """push(MerkleRoot) then alpha = sample(prover_fiat_shamir())"""
        from .fri import sample
        from .proof_stream import ProofItem
        ps.push(ProofItem.MerkleRoot(root))
        return sample(ps.prover_fiat_shamir()).value()


# ------------------------------------------------------------------------------------------------ native path (sh_dist_*)
class NativeDist:
    """The C++ orchestration of csrc/api_dist.hip for one process group: sh_dist_ntt / sh_dist_coset_lde / sh_dist_fri_commit.
    Over "nccl" the library talks to RCCL itself (rank 0's unique id travels through torch.distributed once); over any other
    backend (gloo rehearsals, several ranks on one card) the three transport operations are Python callbacks that stage the
    device buffers through host memory and the torch collective — the orchestration in C++ is the same code either way."""

    def __init__(self, engine, group=None, force_callbacks=False):
        import torch
        from . import _lib
        self.engine, self.group, self._lib = engine, group, _lib
        self.P, self.g = _world_rank(group)
        self.handle = _lib.dist_t()
        ctx = engine.ctx
        dist = _dist()
        use_rccl = (self.P == 1 or dist.get_backend(group) == "nccl") and not force_callbacks
        if use_rccl:
            uid = np.zeros(128, dtype=np.uint8)
            if self.g == 0:
                _lib.check(_lib.lib.sh_dist_unique_id(uid.ctypes.data_as(_lib.u8p)))
            if self.P > 1:
                t = torch.from_numpy(uid).to(engine.device)
                dist.broadcast(t, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
                uid = t.cpu().numpy()
            _lib.check(_lib.lib.sh_dist_create(ctx.handle, self.g, self.P, uid.ctypes.data_as(_lib.u8p), C.byref(self.handle)))
            self.transport = "rccl"
        else:
            self._cbs = (_lib.ALL_TO_ALL_CB(self._cb_all_to_all), _lib.ALL_GATHER_CB(self._cb_all_gather), _lib.SENDRECV_CB(self._cb_sendrecv))
            self._tr = _lib.DistTransport(None, *self._cbs)
            _lib.check(_lib.lib.sh_dist_create_with_transport(ctx.handle, self.g, self.P, C.byref(self._tr), C.byref(self.handle)))
            self.transport = "callbacks"
        self.errors = []

    # -- host-staged transport (the library calls these from its enqueue thread or from the calling thread)
    def _down(self, ptr, nbytes):
        a = np.empty(nbytes, dtype=np.uint8)
        self._lib.check(self._lib.lib.sh_dev_download(self.engine.ctx.handle, a.ctypes.data, ptr, nbytes))
        return a

    def _up(self, ptr, a):
        a = np.ascontiguousarray(a)
        self._lib.check(self._lib.lib.sh_dev_upload(self.engine.ctx.handle, ptr, a.ctypes.data, a.nbytes))

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # never unwind through the C frame
            self.errors.append(e)
            return 1

    def _cb_all_to_all(self, _user, d_send, d_recv, nbytes, _stream):
        def run():
            import torch
            s = torch.from_numpy(self._down(d_send, nbytes * self.P))
            r = torch.empty_like(s)
            _dist().all_to_all_single(r, s, group=self.group)
            self._up(d_recv, r.numpy())
        return self._guard(run)

    def _cb_all_gather(self, _user, d_send, d_recv, nbytes, _stream):
        def run():
            import torch
            s = torch.from_numpy(self._down(d_send, nbytes))
            r = torch.empty(nbytes * self.P, dtype=torch.uint8)
            _dist().all_gather_into_tensor(r, s, group=self.group)
            self._up(d_recv, r.numpy())
        return self._guard(run)

    def _cb_sendrecv(self, _user, ops, n_ops, _stream):
        def run():
            import torch
            dist = _dist()
            gr = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
            p2p, landed = [], []
            for i in range(n_ops):
                op = ops[i]
                if op.is_send:
                    buf = torch.from_numpy(self._down(op.d_buf, op.bytes))
                    landed.append((None, buf))
                    p2p.append(dist.P2POp(dist.isend, buf, gr(op.peer), group=self.group))
                else:
                    buf = torch.empty(op.bytes, dtype=torch.uint8)
                    landed.append((op.d_buf, buf))
                    p2p.append(dist.P2POp(dist.irecv, buf, gr(op.peer), group=self.group))
            if p2p:
                for w in dist.batch_isend_irecv(p2p):
                    w.wait()
            for ptr, buf in landed:
                if ptr is not None:
                    self._up(ptr, buf.numpy())
        return self._guard(run)

    def _check(self, rc):
        if self.errors:
            e = self.errors[:]
            self.errors.clear()
            raise e[0]
        self._lib.check(rc)

    def selftest(self):
        self._check(self._lib.lib.sh_dist_selftest(self.handle))

    def ntt(self, local, log_n, inverse=False, input_layout="natural", output_layout="natural"):
        with self.engine.scope():
            out = self.engine.torch.empty_like(local.reshape(-1, 2))
        self._check(self._lib.lib.sh_dist_ntt(self.handle, local.data_ptr(), log_n, int(bool(inverse)), int(input_layout == "columns"),
                                              int(output_layout == "rows"), out.data_ptr()))
        return out

    def coset_lde(self, local_coeffs, log_n, log_expansion, offset):
        from .context import _u64p, scalar
        m_out = (local_coeffs.numel() // 2) << log_expansion
        with self.engine.scope():
            out = self.engine.torch.empty((m_out, 2), dtype=local_coeffs.dtype, device=local_coeffs.device)
        off = scalar(int(offset.value() if hasattr(offset, "value") else offset))
        self._check(self._lib.lib.sh_dist_coset_lde(self.handle, local_coeffs.data_ptr(), log_n, log_expansion, _u64p(off), out.data_ptr()))
        return out

    def fri_commit(self, local, N, offset, expansion_factor=4, num_colinearity_tests=17, proof_stream=None, min_half=1 << 16):
        from .context import _u8p, _u64p
        _lib = self._lib
        ps = proof_stream if proof_stream is not None else self.engine.new_proof_stream()
        R = int(_lib.lib.sh_fri_num_rounds(N, expansion_factor, num_colinearity_tests))
        roots = np.zeros((max(R, 1), 64), dtype=np.uint8)
        last = np.empty((N >> max(R - 1, 0), 2), dtype=np.uint64)
        last_len = C.c_size_t(0)
        p = _lib.FriParams()
        off = int(offset.value() if hasattr(offset, "value") else offset)
        p.expansion, p.num_colinearity_tests = expansion_factor, num_colinearity_tests
        p.offset[0], p.offset[1] = off & _MASK, off >> 64
        self._check(_lib.lib.sh_dist_fri_commit(self.handle, local.data_ptr(), N, C.byref(p), ps.handle, min_half, _u8p(roots), _u64p(last),
                                                C.byref(last_len)))
        return [roots[i].tobytes() for i in range(R)], last[:last_len.value], ps

    def close(self):
        if self.handle:
            self._lib.lib.sh_dist_destroy(self.handle)
            self.handle = None


def _native(engine, group):
    """the engine's NativeDist for this group, or None (CPU rehearsal engines, STARK_DIST_PYTHON=1: the Python orchestration below)"""
    if not hasattr(engine, "ctx") or os.environ.get("STARK_DIST_PYTHON") == "1":
        return None
    cache = engine.__dict__.setdefault("_native_dist", {})
    key = id(group)
    if key not in cache:
        cache[key] = NativeDist(engine, group, force_callbacks=os.environ.get("STARK_DIST_CALLBACKS") == "1")
    return cache[key]


# ------------------------------------------------------------------------------------------------ exchanges
def _all_to_all(send, group):
    """send: (P, ...) contiguous; block s goes to rank s; returns (P, ...) with block s received from rank s."""
    import torch
    dist = _dist()
    if _needs_staging(send, group):
        s = send.cpu()
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=group)
        return r.to(send.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv


def exchange_transpose(x, engine, group=None):
    """The global matrix is (P·a) × (P·b) elements, row-major, rank r holding rows [r·a, (r+1)·a) as x of shape
    (a, P·b, 2).  Returns this rank's columns [r·b, (r+1)·b) as rows: shape (b, P·a, 2).
    Pack = the local transpose (its rows [s·b, (s+1)·b) are what rank s gets), unpack = swap of the two outer indices
    of what arrives; both are the engine's kernels, torch only carries the buffers through the collective."""
    P, _ = _world_rank(group)
    a, cols = x.shape[0], x.shape[1]
    assert cols % P == 0, "column count must be divisible by the number of ranks"
    b = cols // P
    send = engine.transpose(x.contiguous())                           # (P·b, a, 2) == [dest s][col c][row]
    if P == 1:
        return send
    recv = _all_to_all(send.view(P, b, a, 2), group)                  # [src s][col c][row of s]
    return engine.interleave(recv, P, b, a)


def _all_gather_pairs(pair, group):
    """pair: 128 bytes on the engine's device -> (P·128,) bytes, rank order, on the same device (one tiny collective)."""
    import torch
    dist = _dist()
    P, _ = _world_rank(group)
    if P == 1:
        return pair
    if _needs_staging(pair, group):
        s = pair.cpu()
        out = torch.empty(P * 128, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, s, group=group)
        return out.to(pair.device)
    out = torch.empty(P * 128, dtype=torch.uint8, device=pair.device)
    dist.all_gather_into_tensor(out, pair, group=group)
    return out


def all_gather_slices(x, group=None):
    """concatenate the ranks' (m, 2) slices in rank order -> (P·m, 2) on every rank."""
    import torch
    dist = _dist()
    P, _ = _world_rank(group)
    if P == 1:
        return x
    src = x.contiguous()
    if _needs_staging(src, group):
        s = src.cpu()
        out = torch.empty((P * s.shape[0], 2), dtype=s.dtype)
        dist.all_gather_into_tensor(out, s, group=group)
        return out.to(x.device)
    out = torch.empty((P * src.shape[0], 2), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out


def _regroup(chunk, group):
    """chunk = slice g of P of the codeword (halves 2g, 2g+1 of 2P).  Returns (half g, half P+g)."""
    import torch
    dist = _dist()
    P, g = _world_rank(group)
    h = chunk.shape[0] // 2
    parts = (chunk[:h].contiguous(), chunk[h:].contiguous())
    if P == 1:
        return parts
    stage = _needs_staging(chunk, group)
    got = {}
    ops, keep = [], []
    for q in (0, 1):                                  # half 2g+q goes to rank (2g+q) mod P, as lo when 2g+q < P
        idx = 2 * g + q
        dst = idx % P
        if dst == g:
            got[idx] = parts[q]
        else:
            buf = parts[q].cpu() if stage else parts[q]
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, dst if group is None else dist.get_global_rank(group, dst), group=group))
    for idx in (g, P + g):                            # half idx lives on rank idx // 2
        src = idx // 2
        if src != g:
            buf = torch.empty((h, 2), dtype=chunk.dtype, device="cpu" if stage else chunk.device)
            got[idx] = buf
            ops.append(dist.P2POp(dist.irecv, buf, src if group is None else dist.get_global_rank(group, src), group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    lo, hi = got[g], got[P + g]
    if stage:
        lo, hi = lo.to(chunk.device), hi.to(chunk.device)
    return lo, hi


# ------------------------------------------------------------------------------------------------ four-step NTT
def split_log(log_n):
    """(log n1, log n2): rows x columns of the four-step matrix, n1 <= n2."""
    return log_n // 2, log_n - log_n // 2


def dist_ntt(local, log_n, engine, group=None, inverse=False, input_layout="natural", output_layout="natural"):
    """NTT of the 2^log_n-point vector whose slice this rank holds (see module docstring for the layouts).
    input_layout  "natural": local = x[g·n/P : (g+1)·n/P];  "columns": local[c][j1] = x[j1·n2 + g·n2/P + c].
    output_layout "natural": returns X[g·n/P : (g+1)·n/P];  "rows":    returns [k1l][k2] = X[g·n1/P + k1l + n1·k2]."""
    P, g = _world_rank(group)
    nat = _native(engine, group)
    if nat is not None:
        if local.numel() != 2 * ((1 << log_n) // P):
            raise ValueError("local slice has the wrong length")
        return nat.ntt(local.contiguous(), log_n, inverse, input_layout, output_layout)
    l1, l2 = split_log(log_n)
    n1, n2 = 1 << l1, 1 << l2
    if n1 % P or n2 % P:
        raise ValueError(f"2^{l1} x 2^{l2} matrix does not split over {P} ranks")
    if local.numel() != 2 * (n1 * n2 // P):
        raise ValueError("local slice has the wrong length")
    with engine.scope():
        if input_layout == "natural":
            y = exchange_transpose(local.reshape(n1 // P, n2, 2), engine, group)  # (n2/P, n1, 2): [j2l][j1]
        elif input_layout == "columns":
            y = local.reshape(n2 // P, n1, 2).contiguous()
        else:
            raise ValueError(input_layout)
        engine.ntt_rows(y, l1, inverse)                                          # [j2l][k1]
        engine.twiddle(y, l1, g * (n2 // P), 0, log_n, inverse)
        z = exchange_transpose(y, engine, group)                                 # (n1/P, n2, 2): [k1l][j2]
        engine.ntt_rows(z, l2, inverse)                                          # [k1l][k2] = X[k1 + n1·k2]
        if output_layout == "rows":
            return z
        if output_layout != "natural":
            raise ValueError(output_layout)
        return exchange_transpose(z, engine, group).reshape(-1, 2)               # [k2l][k1] = natural slice


def dist_coset_lde(local_coeffs, log_n, log_expansion, offset, engine, group=None):
    """coset LDE of a polynomial with 2^log_n coefficients spread in natural order: returns this
    rank's natural-order slice of the 2^(log_n+log_expansion) evaluations f(offset·w^i).  Each rank scales its
    coefficients by offset^j, the zero padding is laid out by slice arithmetic (no traffic), then dist_ntt."""
    import torch
    P, g = _world_rank(group)
    n, N = 1 << log_n, 1 << (log_n + log_expansion)
    m_in, m_out = n // P, N // P
    if local_coeffs.numel() != 2 * m_in:
        raise ValueError("local coefficient slice has the wrong length")
    nat = _native(engine, group)
    if nat is not None:
        return nat.coset_lde(local_coeffs.contiguous(), log_n, log_expansion, offset)
    with engine.scope():
        scaled = engine.scale_powers(local_coeffs.reshape(-1, 2).clone(), g * m_in, offset)
        # padded vector = [coeffs | zeros]; rank g owns padded[g·m_out, (g+1)·m_out), m_out = 2^e·m_in: the coefficient slice
        # of rank s lands on rank s >> e at offset (s mod 2^e)·m_in.  Only those slices move, point to point; ranks
        # >= P/2^e own zeros only.
        mine = engine.zeros(m_out)
        E = 1 << log_expansion
        dst = g // E
        if P == 1:
            mine[:m_in] = scaled
        else:
            dist = _dist()
            stage = _needs_staging(scaled, group)
            gr = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
            ops, landed = [], []
            if dst == g:
                mine[(g % E) * m_in:(g % E + 1) * m_in] = scaled
            else:
                buf = scaled.cpu() if stage else scaled
                landed.append((None, buf))
                ops.append(dist.P2POp(dist.isend, buf, gr(dst), group=group))
            for s_ in range(g * E, min((g + 1) * E, P)):
                if s_ == g:
                    continue
                view = mine[(s_ % E) * m_in:(s_ % E + 1) * m_in]
                buf = torch.empty((m_in, 2), dtype=scaled.dtype) if stage else view
                landed.append((view if stage else None, buf))
                ops.append(dist.P2POp(dist.irecv, buf, gr(s_), group=group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for view, buf in landed:
                if view is not None:
                    view.copy_(buf)
    return dist_ntt(mine, log_n + log_expansion, engine, group)


# ------------------------------------------------------------------------------------------------ sharded FRI commit
def _num_rounds(L, expansion, t):
    r = 0
    while L > expansion and 4 * t < L:
        L //= 2
        r += 1
    return r


def _mul_mod(a, b):
    from .field import PRIME
    return (a * b) % PRIME


def dist_fri_commit(local, N, engine, offset, expansion_factor=4, num_colinearity_tests=17, group=None,
                    proof_stream=None, min_half=1 << 16):
    """FRI commit of ONE codeword of N elements spread in natural order over the ranks.
    Every rank returns (roots [R × 64 B], last_codeword (L,2) uint64, proof_stream) — identical on all ranks."""
    P, g = _world_rank(group)
    if N % P or (N // P) < 2:
        raise ValueError("codeword does not split over the ranks")
    if local.numel() != 2 * (N // P):
        raise ValueError("local slice has the wrong length")
    nat = _native(engine, group)
    if nat is not None:
        return nat.fri_commit(local.contiguous(), N, offset, expansion_factor, num_colinearity_tests, proof_stream, max(1, min_half))
    ps = proof_stream if proof_stream is not None else engine.new_proof_stream()
    offset = int(offset.value() if hasattr(offset, "value") else offset)
    L, chunk, roots = N, local.reshape(-1, 2), []
    with engine.scope():
        while L // (2 * P) >= max(1, min_half) and _num_rounds(L, expansion_factor, num_colinearity_tests) > 1:
            lo, hi = _regroup(chunk, group)
            pairs = _all_gather_pairs(engine.subroot_pair(lo, hi), group)     # sub-roots never leave the device
            root = engine.top_root_pairs(pairs, P)                            # the round's one host touch
            roots.append(root)
            alpha = engine.push_root_and_sample(ps, root)
            log_L = L.bit_length() - 1
            chunk = engine.fold_range(lo, hi, g * (L // (2 * P)), alpha, offset, engine.root_of_unity(log_L))
            offset = _mul_mod(offset, offset)
            L //= 2
        full = all_gather_slices(chunk, group)
        tail_roots, last = engine.commit_tail(full, L, offset, expansion_factor, num_colinearity_tests, ps)
    return roots + list(tail_roots), last, ps

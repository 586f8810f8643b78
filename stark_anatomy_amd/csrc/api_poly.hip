// api_poly.hip — polynomial layer behind include/stark_hip.h: product through the NTT, quotient by a
// zerofier X^n − c on an evaluation coset (one batched inversion over a single period of the zerofier) and in coefficient form.
#include "api_internal.h"

static int32_t check_flag_divzero(sh_ctx* c) {
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (*(volatile uint32_t*)c->h_pinned) return sh_fail(SH_EDIVZERO, "Cannot divide by zero");   // field.rs:102
    return SH_OK;
}

extern "C" {

int32_t sh_dev_poly_mul(sh_ctx* c, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb, uint64_t* d_out) {
    CTXCHK(c);
    if (!na || !nb) return sh_fail(SH_EINVAL, "sh_poly_mul: a polynomial needs at least one coefficient");
    if (!d_a || !d_b || !d_out) return sh_fail(SH_EINVAL, "sh_poly_mul: null pointer");
    const size_t m = na + nb - 1;
    if (m < na || m > ((size_t)1 << SH_MAX_LOG_N)) return sh_failf(SH_EINVAL, "sh_poly_mul: product has more than 2^%d coefficients", SH_MAX_LOG_N);
    const uint32_t lg = ilog2(m);
    const size_t n = (size_t)1 << lg;
    void *A, *B;                                              // both zero-padded operands back to back: their two forward transforms are one batch (one launch per pass)
    SHCHK(ensure_scratch(c, SCR_POLY_A, 2 * n * sizeof(fp), &A));
    B = (fp*)A + n;
    HIPCHK(hipMemcpyAsync(A, d_a, na * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    if (n > na) HIPCHK(hipMemsetAsync((fp*)A + na, 0, (n - na) * sizeof(fp), c->stream));
    HIPCHK(hipMemcpyAsync(B, d_b, nb * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    if (n > nb) HIPCHK(hipMemsetAsync((fp*)B + nb, 0, (n - nb) * sizeof(fp), c->stream));
    SHCHK(dev_ntt(c, (fp*)A, lg, false, 2));
    HIPCHK(launch_fp_binop(FPOP_MUL, (const fp*)A, (const fp*)B, (fp*)A, n, c->d_flag, c->stream));
    SHCHK(dev_ntt(c, (fp*)A, lg, true));
    HIPCHK(hipMemcpyAsync(d_out, A, m * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    return SH_OK;
}
int32_t sh_poly_mul(sh_ctx* c, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) {
    CTXCHK(c);
    if (!na || !nb) return sh_fail(SH_EINVAL, "sh_poly_mul: a polynomial needs at least one coefficient");
    if (!a || !b || !out) return sh_fail(SH_EINVAL, "sh_poly_mul: null pointer");
    const size_t m = na + nb - 1;
    if (m < na || m > ((size_t)1 << SH_MAX_LOG_N)) return sh_failf(SH_EINVAL, "sh_poly_mul: product has more than 2^%d coefficients", SH_MAX_LOG_N);
    void *da, *db, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, na * 16, &da));
    SHCHK(ensure_scratch(c, SCR_IO_B, nb * 16, &db));
    SHCHK(ensure_scratch(c, SCR_IO_C, m * 16, &dout));
    HIPCHK(hipMemcpyAsync(da, a, na * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(db, b, nb * 16, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_poly_mul(c, (const uint64_t*)da, na, (const uint64_t*)db, nb, (uint64_t*)dout));
    HIPCHK(hipMemcpyAsync(out, dout, m * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

int32_t sh_dev_coset_zerofier_div(sh_ctx* c, const uint64_t* d_cw, uint32_t log_N, uint32_t log_n, const uint64_t offset[2], const uint64_t cc[2], uint64_t* d_out) {
    CTXCHK(c);
    if (!d_cw || !offset || !cc || !d_out) return sh_fail(SH_EINVAL, "sh_coset_zerofier_div: null pointer");
    if (log_N > SH_MAX_LOG_N || log_n > log_N) return sh_failf(SH_EINVAL, "sh_coset_zerofier_div: need log_n <= log_N <= %d", SH_MAX_LOG_N);
    const fp off = fp_load_host(offset), cv = fp_load_host(cc);
    if (!fp_is_canonical(off) || !fp_is_canonical(cv)) return sh_fail(SH_EINVAL, "sh_coset_zerofier_div: non-canonical scalar");
    // (offset·w_N^i)^n = offset^n · (w_N^n)^i has period E = N/n in i: E zerofier values, one batched inversion, one multiply per element
    const size_t N = (size_t)1 << log_N, E = (size_t)1 << (log_N - log_n);
    fp offn = fp_to_mont(off);
    for (uint32_t i = 0; i < log_n; ++i) offn = mont_mul(offn, offn);
    offn = fp_from_mont(offn);
    void *z, *zinv;
    SHCHK(ensure_scratch(c, SCR_POLY_A, E * sizeof(fp), &z));
    SHCHK(ensure_scratch(c, SCR_POLY_B, E * sizeof(fp), &zinv));
    HIPCHK(hipMemsetAsync(c->d_flag, 0, 4, c->stream));
    HIPCHK(launch_zerofier_values((fp*)z, E, offn, fp_to_mont(host_root_of_unity(log_N - log_n)), cv, c->stream));
    HIPCHK(launch_fp_binop(FPOP_INV_MONT, (const fp*)z, nullptr, (fp*)zinv, E, c->d_flag, c->stream));
    SHCHK(check_flag_divzero(c));                        // the coset meets the zerofier's roots: no quotient there
    HIPCHK(launch_mul_periodic((const fp*)d_cw, (const fp*)zinv, E - 1, (fp*)d_out, N, c->stream));
    return SH_OK;
}
int32_t sh_coset_zerofier_div(sh_ctx* c, const uint64_t* cw, uint32_t log_N, uint32_t log_n, const uint64_t offset[2], const uint64_t cc[2], uint64_t* out) {
    CTXCHK(c);
    if (!cw || !out) return sh_fail(SH_EINVAL, "sh_coset_zerofier_div: null pointer");
    if (log_N > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_coset_zerofier_div: log_N > %d", SH_MAX_LOG_N);
    const size_t bytes = ((size_t)1 << log_N) * 16;
    void *din, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, bytes, &din));
    SHCHK(ensure_scratch(c, SCR_IO_C, bytes, &dout));
    HIPCHK(hipMemcpyAsync(din, cw, bytes, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_coset_zerofier_div(c, (const uint64_t*)din, log_N, log_n, offset, cc, (uint64_t*)dout));
    HIPCHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

int32_t sh_dev_poly_divmod_zerofier(sh_ctx* c, const uint64_t* d_f, size_t nf, size_t n, const uint64_t cc[2], uint64_t* d_q, uint64_t* d_r) {
    CTXCHK(c);
    if (!n) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: the zerofier X^n - c needs n >= 1");
    if (!cc || !d_r || (nf && !d_f) || (nf > n && !d_q)) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: null pointer");
    if (n > ((size_t)1 << (SH_MAX_LOG_N + 2)) || nf > ((size_t)1 << (SH_MAX_LOG_N + 2))) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: too many coefficients");
    const fp cv = fp_load_host(cc);
    if (!fp_is_canonical(cv)) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: non-canonical c");
    HIPCHK(launch_divmod_zerofier((const fp*)d_f, nf, n, fp_to_mont(cv), (fp*)d_q, (fp*)d_r, c->stream));
    return SH_OK;
}
int32_t sh_poly_divmod_zerofier(sh_ctx* c, const uint64_t* f, size_t nf, size_t n, const uint64_t cc[2], uint64_t* q, uint64_t* r) {
    CTXCHK(c);
    if (!n) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: the zerofier X^n - c needs n >= 1");
    if (!cc || !r || (nf && !f) || (nf > n && !q)) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: null pointer");
    if (n > ((size_t)1 << (SH_MAX_LOG_N + 2)) || nf > ((size_t)1 << (SH_MAX_LOG_N + 2))) return sh_fail(SH_EINVAL, "sh_poly_divmod_zerofier: too many coefficients");
    const size_t nq = nf > n ? nf - n : 0;
    void *df, *dq, *dr;
    SHCHK(ensure_scratch(c, SCR_IO_A, (nf ? nf : 1) * 16, &df));
    SHCHK(ensure_scratch(c, SCR_IO_B, (nq ? nq : 1) * 16, &dq));
    SHCHK(ensure_scratch(c, SCR_IO_C, n * 16, &dr));
    if (nf) HIPCHK(hipMemcpyAsync(df, f, nf * 16, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_poly_divmod_zerofier(c, (const uint64_t*)df, nf, n, cc, (uint64_t*)dq, (uint64_t*)dr));
    if (nq) HIPCHK(hipMemcpyAsync(q, dq, nq * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(r, dr, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

}  // extern "C"

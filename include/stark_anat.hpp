// stark_anat.hpp — C++17 host-side mirror of the `stark-anat` crate's public API over the C ABI of libstark_hip.so
// (include/stark_hip.h).  The reference is Rust and no Rust toolchain exists where this is built, so this header is the
// compiled-language host side: same type and method names, argument meaning and failure behaviour as the crate, so that
// the crate's own unit tests can be restated line by line (tests/cpp/reference_suite.cpp).  Header only; link
// -lstark_hip.  Reference citations are file:line under /root/reference/src.
//
//   crate item                                        here
//   field::types::field::{FieldElement, PRIME, ...}   stark_anat::FieldElement, PRIME, ZERO, ONE        (field.rs:7-114)
//   iop::types::merkle::{MerkleHash, MerkleTree}      stark_anat::MerkleHash, MerkleTree                (merkle.rs:9-154)
//   iop::types::proof_stream::ProofStream<S>          stark_anat::ProofStream<S>                        (proof_stream.rs:17-69)
//   (the FRI layer the crate is growing towards)      stark_anat::ProofItem, Fri, ntt, coset_lde        
//
// Rust panics (`assert!`, `panic!`) become C++ exceptions of type stark_anat::Panic carrying the same message;
// `Result::Err(ProofStreamError::…)` becomes stark_anat::ProofStreamError.  Scalar FieldElement operators run on the host
// (one element is not GPU work; the crate's operators are scalar too); everything batched — field vectors, Merkle commit /
// open, NTT, LDE, the FRI prover — runs on the GPU through a Gpu context, and fails loudly if there is none.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "stark_hip.h"

namespace stark_anat {

using u128 = unsigned __int128;

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
struct ProofStreamError : std::runtime_error {                       // proof_stream.rs:9-15
    enum Kind { OutOfBoundsReadIndexError, ErrorSerializingProofItems } kind;
    ProofStreamError(Kind k, const char* m) : std::runtime_error(m), kind(k) {}
};

inline void check(int32_t rc, const char* what) {
    if (rc != SH_OK) throw Panic(std::string(what) + ": " + sh_last_error());
}

// ------------------------------------------------------------------ GPU context (one per device / thread)
class Gpu {
  public:
    explicit Gpu(int device = 0) { check(sh_ctx_create(device, nullptr, &ctx_), "sh_ctx_create"); }
    ~Gpu() { if (ctx_) sh_ctx_destroy(ctx_); }
    Gpu(const Gpu&) = delete;
    Gpu& operator=(const Gpu&) = delete;
    sh_ctx* raw() const { return ctx_; }
    static Gpu& default_gpu() { static Gpu g(0); return g; }
  private:
    sh_ctx* ctx_ = nullptr;
};

// ------------------------------------------------------------------ field (field.rs)
constexpr u128 PRIME = ((u128)0xcb80000000000000ULL << 64) | 1u;     // 1 + 407·2^119, field.rs:15

class FieldElement {
  public:
    constexpr FieldElement() : value_(0) {}
    static FieldElement new_(u128 num) { return FieldElement(num % PRIME); }             // field.rs:30-35
    u128 value() const { return value_; }                                                  // field.rs:52-54
    bool is_zero() const { return value_ == 0; }                                           // field.rs:48-50
    FieldElement pow(FieldElement e) const { return scalar(4, *this, e); }                 // true modpow
    friend FieldElement operator+(FieldElement a, FieldElement b) { return scalar(0, a, b); }   // field.rs:57-68
    friend FieldElement operator-(FieldElement a, FieldElement b) { return scalar(1, a, b); }   // field.rs:70-82
    friend FieldElement operator*(FieldElement a, FieldElement b) { return scalar(2, a, b); }   // field.rs:84-95
    friend FieldElement operator/(FieldElement a, FieldElement b) {                              // field.rs:97-114
        if (b.is_zero()) throw Panic("Cannot divide by zero");                                   // field.rs:101-103
        return scalar(3, a, b);
    }
    friend bool operator==(FieldElement a, FieldElement b) { return a.value_ == b.value_; }
    friend bool operator!=(FieldElement a, FieldElement b) { return a.value_ != b.value_; }
  private:
    constexpr explicit FieldElement(u128 v) : value_(v) {}
    static FieldElement scalar(int op, FieldElement a, FieldElement b) {
        uint64_t x[2] = {(uint64_t)a.value_, (uint64_t)(a.value_ >> 64)}, y[2] = {(uint64_t)b.value_, (uint64_t)(b.value_ >> 64)}, o[2];
        check(sh_fp_scalar(op, x, y, o), "sh_fp_scalar");
        return FieldElement(((u128)o[1] << 64) | o[0]);
    }
    u128 value_;
};
inline const FieldElement ZERO = FieldElement::new_(0);              // field.rs:17-20
inline const FieldElement ONE = FieldElement::new_(1);               // field.rs:22-25

inline std::vector<uint64_t> to_words(const std::vector<FieldElement>& v) {
    std::vector<uint64_t> w(2 * v.size());
    for (size_t i = 0; i < v.size(); ++i) { w[2 * i] = (uint64_t)v[i].value(); w[2 * i + 1] = (uint64_t)(v[i].value() >> 64); }
    return w;
}
inline std::vector<FieldElement> from_words(const uint64_t* w, size_t n) {
    std::vector<FieldElement> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = FieldElement::new_(((u128)w[2 * i + 1] << 64) | w[2 * i]);
    return v;
}

// elementwise field ops on the GPU, with the operators' semantics (division by zero panics like field.rs:101-103)
enum class VecOp { Add, Sub, Mul, Div };
inline std::vector<FieldElement> vec_op(VecOp op, const std::vector<FieldElement>& a, const std::vector<FieldElement>& b, Gpu& gpu = Gpu::default_gpu()) {
    if (a.size() != b.size()) throw Panic("vec_op: length mismatch");
    auto wa = to_words(a), wb = to_words(b);
    std::vector<uint64_t> out(wa.size());
    int32_t rc = SH_EINVAL;
    switch (op) {
        case VecOp::Add: rc = sh_fp_add(gpu.raw(), wa.data(), wb.data(), out.data(), a.size()); break;
        case VecOp::Sub: rc = sh_fp_sub(gpu.raw(), wa.data(), wb.data(), out.data(), a.size()); break;
        case VecOp::Mul: rc = sh_fp_mul(gpu.raw(), wa.data(), wb.data(), out.data(), a.size()); break;
        case VecOp::Div: rc = sh_fp_div(gpu.raw(), wa.data(), wb.data(), out.data(), a.size()); break;
    }
    if (rc == SH_EDIVZERO) throw Panic("Cannot divide by zero");
    check(rc, "field batch op");
    return from_words(out.data(), a.size());
}

// ------------------------------------------------------------------ Merkle (merkle.rs)
struct MerkleHash {                                                  // merkle.rs:9-25
    std::array<uint8_t, 64> inner{};
    static MerkleHash from(const std::array<uint8_t, 64>& h) { MerkleHash m; m.inner = h; return m; }   // merkle.rs:15-19
    static MerkleHash from_bytes(const uint8_t* p) { MerkleHash m; std::memcpy(m.inner.data(), p, 64); return m; }
    const std::array<uint8_t, 64>& as_bytes() const { return inner; }                                   // merkle.rs:22-24
    friend bool operator==(const MerkleHash& a, const MerkleHash& b) { return a.inner == b.inner; }
    friend bool operator!=(const MerkleHash& a, const MerkleHash& b) { return !(a == b); }
    std::string to_json() const {                                    // serde shape {"inner":[b0,...,b63]}
        std::string s = "{\"inner\":[";
        for (int i = 0; i < 64; ++i) { if (i) s += ','; s += std::to_string((unsigned)inner[i]); }
        return s + "]}";
    }
};

class MerkleTree {                                                   // merkle.rs:28-31
  public:
    MerkleHash root;
    static MerkleTree commit(const std::vector<MerkleHash>& leaves, Gpu& gpu = Gpu::default_gpu()) {   // merkle.rs:35-68
        if (leaves.empty() || (leaves.size() & (leaves.size() - 1))) throw Panic("assertion failed: leaves.len().is_power_of_two()");   // merkle.rs:36
        MerkleTree t(gpu);
        t.leaves_ = leaves;
        check(sh_merkle_commit(gpu.raw(), t.bytes().data(), leaves.size(), t.root.inner.data()), "sh_merkle_commit");
        return t;
    }
    std::vector<MerkleHash> open(size_t index) const {               // merkle.rs:70-108
        if (index >= leaves_.size()) throw Panic("assertion failed: index < self.leaves.len()");        // merkle.rs:75
        return std::move(open_batch({(uint64_t)index})[0]);
    }
    // all of `indices` in one GPU pass; each path in `open`'s order (merkle.rs:104-107)
    std::vector<std::vector<MerkleHash>> open_batch(const std::vector<uint64_t>& indices) const {
        for (uint64_t i : indices) if (i >= leaves_.size()) throw Panic("assertion failed: index < self.leaves.len()");
        size_t depth = 0;
        while (((size_t)1 << depth) < leaves_.size()) ++depth;
        std::vector<uint8_t> out(indices.size() * depth * 64 + 1);
        const int32_t rc = sh_merkle_open_batch(gpu_->raw(), bytes().data(), leaves_.size(), indices.data(), indices.size(), out.data());
        if (rc == SH_EBOUNDS) throw Panic(sh_last_error());          // 1-leaf tree: merkle.rs:104 indexes out of bounds
        check(rc, "sh_merkle_open_batch");
        std::vector<std::vector<MerkleHash>> paths(indices.size());
        for (size_t q = 0; q < indices.size(); ++q)
            for (size_t l = 0; l < depth; ++l) paths[q].push_back(MerkleHash::from_bytes(out.data() + (q * depth + l) * 64));
        return paths;
    }
    bool verify(const MerkleHash& leaf, const std::vector<MerkleHash>& path, size_t index) const {    // merkle.rs:110-112
        return is_leaf_in_tree(root, leaf, path, index);
    }
    static bool is_leaf_in_tree(const MerkleHash& root, const MerkleHash& leaf, const std::vector<MerkleHash>& path, size_t index) {   // merkle.rs:114-154
        if (path.empty()) throw Panic("assertion failed: !path.is_empty()");                           // merkle.rs:126
        if (path.size() < 64 && index >= ((size_t)1 << path.size())) throw Panic("assertion failed: index < (1 << path.len())");   // merkle.rs:122
        std::vector<uint8_t> p(path.size() * 64);
        for (size_t i = 0; i < path.size(); ++i) std::memcpy(p.data() + 64 * i, path[i].inner.data(), 64);
        int32_t ok = 0;
        check(sh_merkle_verify(root.inner.data(), leaf.inner.data(), p.data(), path.size(), index, &ok), "sh_merkle_verify");
        return ok != 0;
    }
  private:
    explicit MerkleTree(Gpu& g) : gpu_(&g) {}
    std::vector<uint8_t> bytes() const {
        std::vector<uint8_t> b(leaves_.size() * 64);
        for (size_t i = 0; i < leaves_.size(); ++i) std::memcpy(b.data() + 64 * i, leaves_[i].inner.data(), 64);
        return b;
    }
    std::vector<MerkleHash> leaves_;
    Gpu* gpu_;
};

// leaf hashes of field elements, on the GPU
inline std::vector<MerkleHash> hash_leaves(const std::vector<FieldElement>& elems, Gpu& gpu = Gpu::default_gpu()) {
    auto w = to_words(elems);
    std::vector<uint8_t> out(elems.size() * 64 + 1);
    check(sh_hash_leaves(gpu.raw(), w.data(), elems.size(), out.data()), "sh_hash_leaves");
    std::vector<MerkleHash> h(elems.size());
    for (size_t i = 0; i < elems.size(); ++i) h[i] = MerkleHash::from_bytes(out.data() + 64 * i);
    return h;
}

// ------------------------------------------------------------------ ProofStream<S> (proof_stream.rs:17-69)
// S: `std::string to_json() const` (compact JSON, what serde_json::to_vec writes for S) and
// `static S from_json(std::string_view)`; operator== for the derive(PartialEq) of the stream.
template <class S>
class ProofStream {
  public:
    ProofStream() { check(sh_proofstream_new(&ps_), "sh_proofstream_new"); }
    ~ProofStream() { if (ps_) sh_proofstream_free(ps_); }
    ProofStream(ProofStream&& o) noexcept : ps_(o.ps_) { o.ps_ = nullptr; }
    ProofStream(const ProofStream&) = delete;
    ProofStream& operator=(const ProofStream&) = delete;
    static ProofStream default_() { return ProofStream(); }

    void push(const S& item) {                                       // proof_stream.rs:25-27
        const std::string js = item.to_json();
        check(sh_proofstream_push_json(ps_, (const uint8_t*)js.data(), js.size()), "sh_proofstream_push_json");
    }
    S pull() {                                                       // proof_stream.rs:29-36
        const uint8_t* js; size_t len;
        const int32_t rc = sh_proofstream_pull(ps_, &js, &len);
        if (rc == SH_EBOUNDS) throw ProofStreamError(ProofStreamError::OutOfBoundsReadIndexError, "Read index is out of bounds");
        check(rc, "sh_proofstream_pull");
        return S::from_json(std::string_view((const char*)js, len));
    }
    std::vector<uint8_t> serialize() const {                         // proof_stream.rs:38-40
        size_t need = 0;
        check(sh_proofstream_serialize(ps_, nullptr, 0, &need), "sh_proofstream_serialize");
        std::vector<uint8_t> out(need);
        check(sh_proofstream_serialize(ps_, out.data(), out.size(), &need), "sh_proofstream_serialize");
        return out;
    }
    static ProofStream deserialize(const std::vector<uint8_t>& bytes) {   // proof_stream.rs:42-49
        sh_proofstream* p = nullptr;
        const int32_t rc = sh_proofstream_deserialize(bytes.data(), bytes.size(), &p);
        if (rc == SH_EPARSE) throw ProofStreamError(ProofStreamError::ErrorSerializingProofItems, "Serialization error");
        check(rc, "sh_proofstream_deserialize");
        ProofStream s(p);
        return s;
    }
    std::array<uint8_t, 32> prover_fiat_shamir() const {             // proof_stream.rs:51-56
        std::array<uint8_t, 32> d{};
        check(sh_proofstream_prover_fiat_shamir(ps_, d.data()), "sh_proofstream_prover_fiat_shamir");
        return d;
    }
    std::array<uint8_t, 32> verifier_fiat_shamir() const {           // proof_stream.rs:58-64
        std::array<uint8_t, 32> d{};
        check(sh_proofstream_verifier_fiat_shamir(ps_, d.data()), "sh_proofstream_verifier_fiat_shamir");
        return d;
    }
    size_t len() const { size_t n = 0, r = 0; check(sh_proofstream_len(ps_, &n, &r), "sh_proofstream_len"); return n; }
    size_t read_index() const { size_t n = 0, r = 0; check(sh_proofstream_len(ps_, &n, &r), "sh_proofstream_len"); return r; }
    friend bool operator==(const ProofStream& a, const ProofStream& b) {   // derive(PartialEq), proof_stream.rs:17
        return a.read_index() == b.read_index() && a.serialize() == b.serialize();
    }
    sh_proofstream* raw() const { return ps_; }
  private:
    explicit ProofStream(sh_proofstream* p) : ps_(p) {}
    sh_proofstream* ps_ = nullptr;
};

// ------------------------------------------------------------------ the FRI layer
inline FieldElement generator() { uint64_t o[2]; check(sh_generator(o), "sh_generator"); return FieldElement::new_(((u128)o[1] << 64) | o[0]); }
inline FieldElement root_of_unity(uint32_t log_n) { uint64_t o[2]; check(sh_root_of_unity(log_n, o), "sh_root_of_unity"); return FieldElement::new_(((u128)o[1] << 64) | o[0]); }
inline uint32_t log2_exact(size_t n, const char* what) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) ++l;
    if (n == 0 || ((size_t)1 << l) != n) throw Panic(std::string(what) + ": length must be a power of two");
    return l;
}
inline std::vector<FieldElement> ntt(const std::vector<FieldElement>& x, bool inverse = false, Gpu& gpu = Gpu::default_gpu()) {   // G1/G7
    auto w = to_words(x);
    check(sh_ntt(gpu.raw(), w.data(), log2_exact(x.size(), "ntt"), inverse ? 1 : 0), "sh_ntt");
    return from_words(w.data(), x.size());
}
inline std::vector<FieldElement> coset_lde(const std::vector<FieldElement>& coeffs, uint32_t log_expansion, FieldElement offset, Gpu& gpu = Gpu::default_gpu()) {   // G7
    auto w = to_words(coeffs);
    std::vector<uint64_t> out(w.size() << log_expansion);
    uint64_t off[2] = {(uint64_t)offset.value(), (uint64_t)(offset.value() >> 64)};
    check(sh_coset_lde(gpu.raw(), w.data(), log2_exact(coeffs.size(), "coset_lde"), log_expansion, off, out.data()), "sh_coset_lde");
    return from_words(out.data(), coeffs.size() << log_expansion);
}

// ------------------------------------------------------------------ polynomial layer
// coefficients lowest degree first, never trimmed; no counterpart in the crate
inline std::vector<FieldElement> poly_mul(const std::vector<FieldElement>& a, const std::vector<FieldElement>& b, Gpu& gpu = Gpu::default_gpu()) {
    if (a.empty() || b.empty()) throw Panic("poly_mul: a polynomial needs at least one coefficient");
    auto wa = to_words(a), wb = to_words(b);
    std::vector<uint64_t> out(2 * (a.size() + b.size() - 1));
    check(sh_poly_mul(gpu.raw(), wa.data(), a.size(), wb.data(), b.size(), out.data()), "sh_poly_mul");
    return from_words(out.data(), a.size() + b.size() - 1);
}
// codeword[i] / ((offset·w_N^i)^(2^log_n) - c); panics like field.rs:101-103 when the coset meets a root of the zerofier
inline std::vector<FieldElement> coset_zerofier_div(const std::vector<FieldElement>& codeword, uint32_t log_n, FieldElement c, FieldElement offset, Gpu& gpu = Gpu::default_gpu()) {
    auto w = to_words(codeword);
    std::vector<uint64_t> out(w.size());
    uint64_t off[2] = {(uint64_t)offset.value(), (uint64_t)(offset.value() >> 64)}, cc[2] = {(uint64_t)c.value(), (uint64_t)(c.value() >> 64)};
    const int32_t rc = sh_coset_zerofier_div(gpu.raw(), w.data(), log2_exact(codeword.size(), "coset_zerofier_div"), log_n, off, cc, out.data());
    if (rc == SH_EDIVZERO) throw Panic("Cannot divide by zero");
    check(rc, "sh_coset_zerofier_div");
    return from_words(out.data(), codeword.size());
}
// (q, r) with f = q·(X^n - c) + r, deg r < n
inline std::pair<std::vector<FieldElement>, std::vector<FieldElement>> poly_divmod_zerofier(const std::vector<FieldElement>& f, size_t n, FieldElement c, Gpu& gpu = Gpu::default_gpu()) {
    auto w = to_words(f);
    const size_t nq = f.size() > n ? f.size() - n : 0;
    std::vector<uint64_t> q(2 * (nq ? nq : 1)), r(2 * n);
    uint64_t cc[2] = {(uint64_t)c.value(), (uint64_t)(c.value() >> 64)};
    check(sh_poly_divmod_zerofier(gpu.raw(), w.data(), f.size(), n, cc, nq ? q.data() : nullptr, r.data()), "sh_poly_divmod_zerofier");
    return {from_words(q.data(), nq), from_words(r.data(), n)};
}

// enum ProofItem { MerkleRoot(MerkleHash), Codeword(Vec<u128>), Leaves(Vec<u128>), Path(Vec<MerkleHash>) } (G6, G9), kept
// as the compact JSON serde's externally tagged representation gives it.
inline std::string u128_dec(u128 v) {
    if (v == 0) return "0";
    char buf[40]; int n = 0;
    while (v) { buf[n++] = (char)('0' + (int)(v % 10)); v /= 10; }
    return std::string(std::string_view(buf, (size_t)n).rbegin(), std::string_view(buf, (size_t)n).rend());
}
struct ProofItem {
    std::string json;
    static ProofItem MerkleRoot(const MerkleHash& h) { return {"{\"MerkleRoot\":" + h.to_json() + "}"}; }
    static ProofItem Codeword(const std::vector<FieldElement>& v) { return {"{\"Codeword\":" + list(v) + "}"}; }
    static ProofItem Leaves(const std::vector<FieldElement>& v) { return {"{\"Leaves\":" + list(v) + "}"}; }
    static ProofItem Path(const std::vector<MerkleHash>& p) {
        std::string s = "{\"Path\":[";
        for (size_t i = 0; i < p.size(); ++i) { if (i) s += ','; s += p[i].to_json(); }
        return {s + "]}"};
    }
    bool is(const char* tag) const { return json.rfind(std::string("{\"") + tag + "\":", 0) == 0; }
    std::string to_json() const { return json; }
    static ProofItem from_json(std::string_view sv) { return {std::string(sv)}; }
    friend bool operator==(const ProofItem& a, const ProofItem& b) { return a.json == b.json; }
  private:
    static std::string list(const std::vector<FieldElement>& v) {
        std::string s = "[";
        for (size_t i = 0; i < v.size(); ++i) { if (i) s += ','; s += u128_dec(v[i].value()); }
        return s + "]";
    }
};

class Fri {
  public:
    Fri(FieldElement offset, FieldElement omega, size_t domain_length, uint32_t expansion_factor, uint32_t num_colinearity_tests, Gpu& gpu = Gpu::default_gpu())
        : gpu_(&gpu), N_(domain_length) {
        if (omega != root_of_unity(log2_exact(domain_length, "Fri"))) throw Panic("omega must be the canonical primitive root g^(2^119/N) (G1)");
        prm_.expansion = expansion_factor; prm_.num_colinearity_tests = num_colinearity_tests;
        prm_.offset[0] = (uint64_t)offset.value(); prm_.offset[1] = (uint64_t)(offset.value() >> 64);
    }
    uint32_t num_rounds() const { return sh_fri_num_rounds(N_, prm_.expansion, prm_.num_colinearity_tests); }   // G5
    // pushes R Merkle roots and the last codeword into `stream` (byte-identical to a ProofStream<ProofItem>); returns the roots
    template <class S>
    std::vector<MerkleHash> commit(const std::vector<FieldElement>& codeword, ProofStream<S>& stream) const {
        if (codeword.size() != N_) throw Panic("initial codeword length does not match length of initial codeword");
        const uint32_t R = num_rounds();
        auto w = to_words(codeword);
        std::vector<uint8_t> roots((size_t)(R ? R : 1) * 64);
        std::vector<uint64_t> last(2 * (N_ >> (R ? R - 1 : 0)));
        size_t last_len = 0;
        check(sh_fri_commit_proofstream(gpu_->raw(), w.data(), N_, &prm_, stream.raw(), roots.data(), last.data(), &last_len), "sh_fri_commit_proofstream");
        std::vector<MerkleHash> out(R);
        for (uint32_t r = 0; r < R; ++r) out[r] = MerkleHash::from_bytes(roots.data() + 64 * (size_t)r);
        return out;
    }
    template <class S>
    std::vector<uint64_t> prove(const std::vector<FieldElement>& codeword, ProofStream<S>& stream) const {   // G11
        if (codeword.size() != N_) throw Panic("initial codeword length does not match length of initial codeword");
        auto w = to_words(codeword);
        std::vector<uint64_t> top(prm_.num_colinearity_tests + 1);
        size_t n = 0;
        check(sh_fri_prove(gpu_->raw(), w.data(), N_, &prm_, stream.raw(), top.data(), &n), "sh_fri_prove");
        top.resize(n);
        return top;
    }
    template <class S>
    bool verify(ProofStream<S>& stream, std::string* why = nullptr) const {                                    // G12 (host)
        int32_t ok = 0;
        check(sh_fri_verify(stream.raw(), N_, &prm_, &ok, nullptr, nullptr), "sh_fri_verify");
        if (!ok && why) *why = sh_last_error();
        return ok != 0;
    }
  private:
    Gpu* gpu_;
    size_t N_;
    sh_fri_params prm_{};
};

}  // namespace stark_anat

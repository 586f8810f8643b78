// reference_suite.cpp — the reference crate's own unit tests (cargo test: field.rs:116-220, merkle.rs:156-282,
// proof_stream.rs:70-226), restated against include/stark_anat.hpp, i.e. against the GPU library through its C ABI,
// plus the FRI layer's config-1 run (prove + verify at N = 2^10) pinned to the committed golden digests.
//   reference_suite host                       tests that need no GPU (scalar field, ProofStream)
//   reference_suite gpu <commit_sha256> <prove_sha256>   everything; the two digests come from tests/golden/*.json
// Exit status 0 iff every test passed; one line per test.
#include <cstdio>
#include <functional>
#include <random>
#include <string>

#include "../../include/stark_anat.hpp"

using namespace stark_anat;

static int g_failed = 0, g_run = 0;
#define EXPECT(cond)                                                                                   \
    do { if (!(cond)) throw std::runtime_error(std::string("expectation failed: ") + #cond + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); } while (0)
static void run(const char* name, const std::function<void()>& f) {
    ++g_run;
    try { f(); std::printf("test %s ... ok\n", name); }
    catch (const std::exception& e) { ++g_failed; std::printf("test %s ... FAILED: %s\n", name, e.what()); }
    std::fflush(stdout);
}
template <class E, class F>
static bool panics(F&& f) { try { f(); } catch (const E&) { return true; } return false; }

static u128 dec(const char* s) { u128 v = 0; for (; *s; ++s) v = v * 10 + (u128)(*s - '0'); return v; }
static std::string hex(const uint8_t* p, size_t n) { static const char* d = "0123456789abcdef"; std::string s; for (size_t i = 0; i < n; ++i) { s += d[p[i] >> 4]; s += d[p[i] & 15]; } return s; }
static std::string sha256_hex(const std::vector<uint8_t>& b) { uint8_t o[32]; sh_sha256(b.data(), b.size(), o); return hex(o, 32); }

// ---------------------------------------------------------------- field::types::field::tests
static void field_tests(bool gpu) {
    run("field::test_pow", [] {                                   // field.rs:139-151
        EXPECT(FieldElement::new_(2).pow(FieldElement::new_(160)).value() == dec("242584109230747146804944788495759879579"));
    });
    run("field::test_add", [] { EXPECT((FieldElement::new_(PRIME - 10) + FieldElement::new_(12)).value() == 2); });          // field.rs:153-166
    run("field::test_sub", [] { EXPECT((ZERO - FieldElement::new_(12)).value() == PRIME - 12); });                           // field.rs:168-180
    const FieldElement a = FieldElement::new_((u128)UINT64_MAX - 2), b = FieldElement::new_((u128)UINT64_MAX - 1);
    run("field::test_mul", [=] { EXPECT((a * b).value() == dec("69784469778708083235216150296170332165")); });               // field.rs:182-194
    run("field::test_div", [=] { EXPECT((a / b).value() == dec("263166645724356846472197722797662682189")); });              // field.rs:196-208
    for (unsigned k = 1; k <= 11; ++k)                                                                                      // field.rs:118-137, 210-220
        run(("field::test_inv_minus_" + std::to_string(k)).c_str(), [k] {
            const FieldElement x = FieldElement::new_(1), y = FieldElement::new_(PRIME - k);
            EXPECT(((x / y) * y).value() == 1);
        });
    run("field::div_by_zero_panics", [] { EXPECT(panics<Panic>([] { (void)(ONE / ZERO); })); });                             // field.rs:101-103
    run("field::new_reduces", [] { EXPECT(FieldElement::new_(PRIME + 5).value() == 5 && FieldElement::new_(~(u128)0).value() == (~(u128)0) % PRIME); });   // field.rs:30-35
    if (!gpu) return;
    run("field::batched_ops_match_operators (GPU)", [=] {
        std::mt19937_64 rng(7);
        std::vector<FieldElement> x, y;
        for (int i = 0; i < 1000; ++i) { x.push_back(FieldElement::new_(((u128)rng() << 64) | rng())); y.push_back(FieldElement::new_(((u128)rng() << 64) | rng() | 1)); }
        x[0] = a; y[0] = b; x[1] = FieldElement::new_(PRIME - 10); y[1] = FieldElement::new_(12);
        auto s = vec_op(VecOp::Add, x, y), d = vec_op(VecOp::Sub, x, y), m = vec_op(VecOp::Mul, x, y), q = vec_op(VecOp::Div, x, y);
        for (size_t i = 0; i < x.size(); ++i) EXPECT(s[i] == x[i] + y[i] && d[i] == x[i] - y[i] && m[i] == x[i] * y[i] && q[i] == x[i] / y[i]);
        EXPECT(m[0].value() == dec("69784469778708083235216150296170332165") && q[0].value() == dec("263166645724356846472197722797662682189"));
        y[17] = ZERO;
        EXPECT(panics<Panic>([&] { vec_op(VecOp::Div, x, y); }));
    });
}

// ---------------------------------------------------------------- iop::types::merkle::tests (N = 64)
static std::mt19937_64 g_rng(12345);
static MerkleHash random_leaf() { MerkleHash h; for (int i = 0; i < 64; i += 8) { uint64_t v = g_rng(); std::memcpy(h.inner.data() + i, &v, 8); } return h; }
static const size_t MN = 64;
static size_t get_different_index(size_t i) { size_t j = i; while (j == i) j = (i + 1 + (size_t)(g_rng() % MN)) % MN; return j; }
static std::vector<MerkleHash> random_leaves() { std::vector<MerkleHash> v; for (size_t i = 0; i < MN; ++i) v.push_back(random_leaf()); return v; }

static void merkle_tests() {
    run("merkle::verify_path_for_each_leaf", [] {                 // merkle.rs:176-185
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        for (size_t i = 0; i < MN; ++i) EXPECT(tree.verify(leaves[i], tree.open(i), i));
    });
    run("merkle::random_leaf_with_original_path", [] {            // merkle.rs:187-193, 238-243
        auto tree = MerkleTree::commit(random_leaves());
        for (size_t i = 0; i < MN; ++i) EXPECT(!tree.verify(random_leaf(), tree.open(i), i));
    });
    run("merkle::different_leaves_with_original_path", [] {       // merkle.rs:195-201, 245-251
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        for (size_t i = 0; i < MN; ++i) EXPECT(!tree.verify(leaves[get_different_index(i)], tree.open(i), i));
    });
    run("merkle::original_leaf_with_different_path_index", [] {   // merkle.rs:203-209, 253-259
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        for (size_t i = 0; i < MN; ++i) EXPECT(!tree.verify(leaves[i], tree.open(i), get_different_index(i)));
    });
    run("merkle::with_changed_root", [] {                         // merkle.rs:211-217, 261-269
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        const MerkleHash original = tree.root;
        for (size_t i = 0; i < MN; ++i) { auto path = tree.open(i); tree.root = random_leaf(); EXPECT(!tree.verify(leaves[i], path, i)); }
        tree.root = original;
        EXPECT(tree.verify(leaves[3], tree.open(3), 3));
    });
    run("merkle::with_tampered_path", [] {                        // merkle.rs:219-225, 271-281
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        for (size_t i = 0; i < MN; ++i) {
            auto path = tree.open(i);
            for (size_t j = 0; j < path.size(); ++j) { auto fake = path; fake[j] = random_leaf(); EXPECT(!tree.verify(leaves[i], fake, i)); }
        }
    });
    run("merkle::different_tree_original_leaves_and_paths", [] {  // merkle.rs:227-236
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        auto fake_tree = MerkleTree::commit(std::vector<MerkleHash>(MN, random_leaf()));
        for (size_t i = 0; i < MN; ++i) EXPECT(!fake_tree.verify(leaves[i], tree.open(i), i));
    });
    run("merkle::contract_violations_panic", [] {                 // merkle.rs:36, 75, 104, 122, 126
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        EXPECT(panics<Panic>([&] { MerkleTree::commit(std::vector<MerkleHash>(leaves.begin(), leaves.begin() + 3)); }));
        EXPECT(panics<Panic>([&] { MerkleTree::commit({}); }));
        EXPECT(panics<Panic>([&] { tree.open(MN); }));
        EXPECT(panics<Panic>([&] { MerkleTree::commit({leaves[0]}).open(0); }));
        EXPECT(MerkleTree::commit({leaves[0]}).root == leaves[0]);                       // merkle.rs:48,62
        EXPECT(panics<Panic>([&] { tree.verify(leaves[0], {}, 0); }));
        EXPECT(panics<Panic>([&] { tree.verify(leaves[0], tree.open(0), MN); }));
    });
    run("merkle::open_batch_equals_open", [] {
        auto leaves = random_leaves(); auto tree = MerkleTree::commit(leaves);
        std::vector<uint64_t> idx; for (size_t i = 0; i < MN; ++i) idx.push_back(i);
        auto all = tree.open_batch(idx);
        for (size_t i = 0; i < MN; ++i) EXPECT(all[i] == tree.open(i));
    });
}

// ---------------------------------------------------------------- iop::types::proof_stream::tests
struct TestStruct {                                                // proof_stream.rs:76-80
    uint8_t a = 0, b = 0;
    std::string to_json() const { return "{\"a\":" + std::to_string(a) + ",\"b\":" + std::to_string(b) + "}"; }
    static TestStruct from_json(std::string_view s) { unsigned x = 0, y = 0; if (std::sscanf(std::string(s).c_str(), "{\"a\":%u,\"b\":%u}", &x, &y) != 2) throw std::runtime_error("bad TestStruct"); return {(uint8_t)x, (uint8_t)y}; }
    bool operator==(const TestStruct& o) const { return a == o.a && b == o.b; }
};
struct TestStructComplex {                                         // proof_stream.rs:82-87
    std::vector<uint8_t> a; uint8_t b = 0; TestStruct c;
    std::string to_json() const {
        std::string s = "{\"a\":[";
        for (size_t i = 0; i < a.size(); ++i) { if (i) s += ','; s += std::to_string(a[i]); }
        return s + "],\"b\":" + std::to_string(b) + ",\"c\":" + c.to_json() + "}";
    }
    static TestStructComplex from_json(std::string_view) { return {}; }
};
static ProofStream<TestStruct> four_items() {
    ProofStream<TestStruct> ps;
    ps.push({1, 2}); ps.push({2, 3}); ps.push({3, 4}); ps.push({4, 5});
    return ps;
}
static void proof_stream_tests() {
    run("proof_stream::test_serialize_inner", [] {                // proof_stream.rs:89-110
        const std::vector<uint8_t> expected = {91, 123, 34, 97, 34, 58, 49, 44, 34, 98, 34, 58, 50, 125, 44, 123, 34, 97, 34, 58, 50, 44, 34, 98, 34, 58, 51, 125, 44, 123,
                                               34, 97, 34, 58, 51, 44, 34, 98, 34, 58, 52, 125, 44, 123, 34, 97, 34, 58, 52, 44, 34, 98, 34, 58, 53, 125, 93};
        EXPECT(four_items().serialize() == expected);
    });
    run("proof_stream::test_deserialize", [] {                    // proof_stream.rs:112-131
        auto ps = four_items();
        auto de = ProofStream<TestStruct>::deserialize(ps.serialize());
        EXPECT(ps == de);
        EXPECT(de.pull() == (TestStruct{1, 2}) && de.pull() == (TestStruct{2, 3}));
    });
    run("proof_stream::test_deserialize_complex", [] {            // proof_stream.rs:133-168
        ProofStream<TestStructComplex> ps;
        ps.push({{1, 2, 3, 4}, 5, {6, 7}}); ps.push({{8, 9, 10, 11}, 12, {13, 14}}); ps.push({{15, 16, 17, 18}, 19, {20, 21}}); ps.push({{22, 23, 24, 25}, 26, {27, 28}});
        auto de = ProofStream<TestStructComplex>::deserialize(ps.serialize());
        EXPECT(ps == de && de.len() == 4);
    });
    run("proof_stream::test_prover_fiat_shamir", [] {             // proof_stream.rs:170-195
        const std::array<uint8_t, 32> expected = {201, 176, 198, 41, 77, 42, 190, 176, 93, 90, 51, 57, 129, 77, 162, 158, 96, 4, 167, 126, 67, 85, 94, 5, 241, 172, 158, 164, 239, 74, 93, 192};
        EXPECT(four_items().prover_fiat_shamir() == expected);
    });
    run("proof_stream::test_verifier_fiat_shamir", [] {           // proof_stream.rs:197-225
        auto ps = four_items();
        ps.pull(); ps.pull(); ps.pull();
        const std::array<uint8_t, 32> expected = {188, 90, 181, 158, 133, 61, 76, 0, 246, 85, 241, 132, 91, 7, 84, 157, 111, 193, 104, 105, 236, 15, 89, 163, 86, 118, 20, 24, 98, 195, 116, 174};
        EXPECT(ps.verifier_fiat_shamir() == expected);
    });
    run("proof_stream::pull_past_end_and_bad_json", [] {          // proof_stream.rs:30-32, 43
        auto ps = four_items();
        for (int i = 0; i < 4; ++i) ps.pull();
        EXPECT(panics<ProofStreamError>([&] { ps.pull(); }));
        const std::string bad = "[{\"a\":1,";
        EXPECT(panics<ProofStreamError>([&] { ProofStream<TestStruct>::deserialize(std::vector<uint8_t>(bad.begin(), bad.end())); }));
    });
}

// ---------------------------------------------------------------- FRI layer, config 1 (seed 1, N = 2^10, n = 256, t = 17)
static std::vector<FieldElement> splitmix_elems(uint64_t seed, size_t n) {          // SURVEY.md §8d generator
    std::vector<FieldElement> out;
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        uint64_t w[2];
        for (int k = 0; k < 2; ++k) {
            s += 0x9E3779B97F4A7C15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            w[k] = z ^ (z >> 31);
        }
        out.push_back(FieldElement::new_(((u128)w[1] << 64) | w[0]));
    }
    return out;
}
static void fri_tests(const std::string& commit_sha, const std::string& prove_sha) {
    const FieldElement g = generator();
    run("fri::c1_commit_transcript_matches_golden", [&] {
        auto cw = splitmix_elems(1, 1 << 10);
        Fri fri(g, root_of_unity(10), 1 << 10, 4, 17);
        ProofStream<ProofItem> ps;
        auto roots = fri.commit(cw, ps);
        EXPECT(roots.size() == 4 && ps.len() == 5);
        EXPECT(ps.pull() == ProofItem::MerkleRoot(roots[0]));
        EXPECT(sha256_hex(ps.serialize()) == commit_sha);
    });
    run("fri::c1_prove_verify_matches_golden", [&] {
        auto cw = coset_lde(splitmix_elems(1, 256), 2, g);
        Fri fri(g, root_of_unity(10), 1 << 10, 4, 17);
        ProofStream<ProofItem> ps;
        auto top = fri.prove(cw, ps);
        EXPECT(top.size() == 17 && ps.len() == 209);
        const auto ser = ps.serialize();
        EXPECT(sha256_hex(ser) == prove_sha);
        auto vs = ProofStream<ProofItem>::deserialize(ser);
        std::string why;
        EXPECT(fri.verify(vs, &why));
        auto bad = ser;                                            // flip one decimal digit of the last codeword
        const std::string needle = "{\"Codeword\":[";
        auto it = std::search(bad.begin(), bad.end(), needle.begin(), needle.end());
        EXPECT(it != bad.end());
        uint8_t& digit = *(it + needle.size() + 3);
        digit = digit == '5' ? '6' : '5';
        auto vb = ProofStream<ProofItem>::deserialize(bad);
        EXPECT(!fri.verify(vb, &why) && !why.empty());
    });
    run("fri::ntt_round_trip_and_lde_consistency", [&] {
        auto c = splitmix_elems(9, 512);
        EXPECT(ntt(ntt(c), true) == c);
        auto lde = coset_lde(c, 2, g);                             // f(g·w^i): check three points by Horner on the host
        const FieldElement w = root_of_unity(11);
        for (size_t i : {(size_t)0, (size_t)1, (size_t)1234}) {
            const FieldElement x = g * w.pow(FieldElement::new_(i));
            FieldElement acc = ZERO;
            for (size_t j = c.size(); j-- > 0;) acc = acc * x + c[j];
            EXPECT(acc == lde[i]);
        }
    });
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    const bool gpu = mode == "gpu";
    if (gpu && argc < 4) { std::fprintf(stderr, "usage: %s gpu <commit_sha256> <prove_sha256>\n", argv[0]); return 2; }
    field_tests(gpu);
    proof_stream_tests();
    if (gpu) { merkle_tests(); fri_tests(argv[2], argv[3]); }
    std::printf("\ntest result: %s. %d passed; %d failed\n", g_failed ? "FAILED" : "ok", g_run - g_failed, g_failed);
    return g_failed ? 1 : 0;
}

// host_transcript.cpp — host side of the hot path: SHA-256, general-length BLAKE2b-512, Merkle path
// verification and the ProofStream transcript (proof_stream.rs:17-69).  Per Fiat-Shamir round this is
// ~300 bytes of JSON and a few SHA-256 blocks, so it stays on the host exactly where the reference
// has it; what matters is that the BYTES are identical to serde_json's compact output.
//
// This is product code (part of libstark_hip.so).  It shares nothing with oracle/.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include "../../include/stark_hip.h"
#include "host_internal.h"

// ------------------------------------------------------------------ SHA-256 (FIPS 180-4)
namespace {
const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
    0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
    0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
    0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
    0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
    0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// One block with the x86 SHA extensions (same FIPS 180-4 function; selected at run time when the CPU has them).
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_block_shani(uint32_t st[8], const uint8_t* p) {
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);            // DCBA
    __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);             // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                               // CDAB
    s1 = _mm_shuffle_epi32(s1, 0x1B);                                 // EFGH
    __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);                         // ABEF
    s1 = _mm_blend_epi16(s1, tmp, 0xF0);                              // CDGH
    const __m128i save0 = s0, save1 = s1;
    __m128i m[4];
    for (int i = 0; i < 16; ++i) {                                    // 16 groups of 4 rounds
        if (i < 4) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), MASK);
        else {
            __m128i t = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);
            t = _mm_add_epi32(t, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));
            m[i & 3] = _mm_sha256msg2_epu32(t, m[(i + 3) & 3]);
        }
        __m128i msg = _mm_add_epi32(m[i & 3], _mm_loadu_si128((const __m128i*)&K256[4 * i]));
        s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
        msg = _mm_shuffle_epi32(msg, 0x0E);
        s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    tmp = _mm_shuffle_epi32(s0, 0x1B);                                // FEBA
    s1 = _mm_shuffle_epi32(s1, 0xB1);                                 // DCHG
    s0 = _mm_blend_epi16(tmp, s1, 0xF0);                              // DCBA
    s1 = _mm_alignr_epi8(s1, tmp, 8);                                 // HGFE
    _mm_storeu_si128((__m128i*)&st[0], s0);
    _mm_storeu_si128((__m128i*)&st[4], s1);
}
const bool HAVE_SHANI = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");

struct Sha256 {
    uint32_t st[8];
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    Sha256() {
        static const uint32_t init[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(st, init, sizeof st);
    }
    void block(const uint8_t* p) {
        if (HAVE_SHANI) { sha256_block_shani(st, p); return; }
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t a = w[i - 15], b = w[i - 2];
            w[i] = w[i - 16] + (ror(a, 7) ^ ror(a, 18) ^ (a >> 3)) + w[i - 7] + (ror(b, 17) ^ ror(b, 19) ^ (b >> 10));
        }
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t t1 = h + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
            uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (fill) {
            size_t take = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, p, take); fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
        while (n >= 64) { block(p); p += 64; n -= 64; }
        if (n) { memcpy(buf, p, n); fill = n; }
    }
    void finish(uint8_t out[32]) {
        uint64_t bits = total * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (fill < 56) ? 56 - fill : 120 - fill;
        uint8_t len[8];
        for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(pad, padlen);
        update(len, 8);
        for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
    }
};

// ------------------------------------------------------------------ BLAKE2b-512 (RFC 7693), any length, unkeyed
const uint64_t BIV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                         0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
const uint8_t BSIG[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
inline uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
void b2_compress(uint64_t h[8], const uint8_t blk[128], uint64_t t, bool last) {
    uint64_t v[16], m[16];
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = BIV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    for (int i = 0; i < 16; ++i) { uint64_t w; memcpy(&w, blk + 8 * i, 8); m[i] = w; }   // little-endian host
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] += v[b] + x; v[d] = ror64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 24);
        v[a] += v[b] + y; v[d] = ror64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror64(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = BSIG[r % 10];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
}  // namespace

void host_sha256(const uint8_t* in, size_t len, uint8_t out[32]) {
    Sha256 s;
    s.update(in, len);
    s.finish(out);
}
void host_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]) {
    uint64_t h[8];
    memcpy(h, BIV, sizeof h);
    h[0] ^= 0x01010040ULL;
    uint64_t t = 0;
    uint8_t blk[128];
    while (len > 128) { t += 128; b2_compress(h, in, t, false); in += 128; len -= 128; }
    memset(blk, 0, sizeof blk);
    if (len) memcpy(blk, in, len);
    t += len;
    b2_compress(h, blk, t, true);
    memcpy(out, h, 64);
}

// ------------------------------------------------------------------ ProofStream (proof_stream.rs:17-69)
struct Sha256Ctx;
struct sh_proofstream {
    std::vector<std::string> items;   // compact-JSON fragment per item
    size_t read_index = 0;
    // Running SHA-256 over "[" item0 "," item1 ... for the first `absorbed` items: the stream is append-only, so
    // prover_fiat_shamir (proof_stream.rs:51-56, re-serialises and re-hashes everything each call) only has to
    // absorb the new items and a closing "]" on a copy.  Same digest, O(new bytes) instead of O(stream).
    Sha256Ctx* running = nullptr;
    size_t absorbed = 0;
    ~sh_proofstream();
};

struct Sha256Ctx { Sha256 h; };
sh_proofstream::~sh_proofstream() { delete running; }

namespace {
// decimal formatting into a caller-provided buffer (no per-character std::string growth); returns chars written
const char DIGIT_PAIRS[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960"
    "616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline size_t put_u64_dec(char* out, uint64_t v) {
    char tmp[20]; int n = 0;
    while (v >= 100) { const unsigned r = (unsigned)(v % 100); v /= 100; tmp[n++] = DIGIT_PAIRS[2 * r + 1]; tmp[n++] = DIGIT_PAIRS[2 * r]; }
    if (v >= 10) { tmp[n++] = DIGIT_PAIRS[2 * v + 1]; tmp[n++] = DIGIT_PAIRS[2 * v]; }
    else tmp[n++] = (char)('0' + v);
    for (int i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return (size_t)n;
}
inline size_t put_u64_dec_padded19(char* out, uint64_t v) {                 // exactly 19 digits, zero padded
    for (int i = 18; i >= 0; --i) { out[i] = (char)('0' + v % 10); v /= 10; }
    return 19;
}
inline size_t put_u128_dec(char* out, unsigned __int128 v) {               // serde_json writes u128 as a bare decimal integer
    const uint64_t E19 = 10000000000000000000ULL;                          // base-10^19 limbs: two 128-bit divisions at most
    if ((uint64_t)(v >> 64) == 0) return put_u64_dec(out, (uint64_t)v);
    const uint64_t l0 = (uint64_t)(v % E19); v /= E19;
    size_t n;
    if ((uint64_t)(v >> 64) == 0 && (uint64_t)v < E19) n = put_u64_dec(out, (uint64_t)v);
    else { const uint64_t l1 = (uint64_t)(v % E19); v /= E19; n = put_u64_dec(out, (uint64_t)v); n += put_u64_dec_padded19(out + n, l1); }
    return n + put_u64_dec_padded19(out + n, l0);
}
// An item whose first byte is LAZY_PATH (no JSON text starts with it) still holds raw digests: write it out in place.
const char LAZY_PATH = 0x01, LAZY_CODEWORD = 0x02, LAZY_LEAVES = 0x03;
struct ByteDec { char txt[256][4]; uint8_t len[256]; ByteDec() { for (int v = 0; v < 256; ++v) { len[v] = (uint8_t)put_u64_dec(txt[v], (uint64_t)v); txt[v][len[v]] = ','; } } };
const ByteDec BYTE_DEC;                                                   // "0," .. "255,": one 4-byte copy per value
void materialize_elems(std::string& it, const char* head, size_t head_len) {      // {"Codeword":[dec,...]} / {"Leaves":[dec,...]}: u128 as bare decimals
    const size_t n = (it.size() - 1) / 16;
    std::string s(head_len + n * 40 + 2, '\0');
    char* buf = &s[0];
    size_t o = head_len;
    memcpy(buf, head, head_len);
    for (size_t i = 0; i < n; ++i) {
        uint64_t w[2];
        memcpy(w, it.data() + 1 + 16 * i, 16);
        if (i) buf[o++] = ',';
        o += put_u128_dec(buf + o, ((unsigned __int128)w[1] << 64) | w[0]);
    }
    buf[o++] = ']'; buf[o++] = '}';
    s.resize(o);
    it = std::move(s);
}
void materialize(std::string& it) {
    if (it.empty()) return;
    if (it[0] == LAZY_CODEWORD) { materialize_elems(it, "{\"Codeword\":[", 13); return; }
    if (it[0] == LAZY_LEAVES) { materialize_elems(it, "{\"Leaves\":[", 11); return; }
    if (it[0] != LAZY_PATH) return;
    const size_t n = (it.size() - 1) / 64;
    const uint8_t* d = (const uint8_t*)it.data() + 1;
    std::string s(9 + n * (10 + 64 * 4 + 3) + 2 + 4, '\0');
    char* buf = &s[0];
    size_t o = 9;
    memcpy(buf, "{\"Path\":[", 9);
    for (size_t k = 0; k < n; ++k) {
        if (k) buf[o++] = ',';
        memcpy(buf + o, "{\"inner\":[", 10); o += 10;
        for (int i = 0; i < 64; ++i) { const uint8_t v = d[64 * k + i]; memcpy(buf + o, BYTE_DEC.txt[v], 4); o += BYTE_DEC.len[v] + 1; }
        buf[o - 1] = ']'; buf[o++] = '}';                                 // the last comma becomes the bracket
    }
    buf[o++] = ']'; buf[o++] = '}';
    s.resize(o);
    it = std::move(s);
}
void materialize_all(std::vector<std::string>& items, size_t count) { for (size_t i = 0; i < count; ++i) materialize(items[i]); }
std::string serialize_items(const std::vector<std::string>& items, size_t count) {   // serialize_inner, proof_stream.rs:67-69
    size_t need = 2;
    for (size_t i = 0; i < count; ++i) need += items[i].size() + 1;
    std::string out;
    out.reserve(need);
    out.push_back('[');
    for (size_t i = 0; i < count; ++i) { if (i) out.push_back(','); out += items[i]; }
    out.push_back(']');
    return out;
}

// Minimal JSON value scanner: validates one value starting at p and appends its compact form to `out`.
struct JsonScan {
    const uint8_t* p; const uint8_t* end; int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool lit(const char* s, std::string& out) { size_t n = strlen(s); if ((size_t)(end - p) < n || memcmp(p, s, n)) return false; out.append(s, n); p += n; return true; }
    bool str(std::string& out) {
        if (p >= end || *p != '"') return false;
        out.push_back('"'); ++p;
        while (p < end) {
            uint8_t c = *p++;
            if (c < 0x20) return false;
            out.push_back((char)c);
            if (c == '\\') { if (p >= end) return false; out.push_back((char)*p++); }
            else if (c == '"') return true;
        }
        return false;
    }
    bool num(std::string& out) {
        const uint8_t* s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') return false;
        if (*p == '0') ++p; else while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') { ++p; if (p >= end || *p < '0' || *p > '9') return false; while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (p < end && (*p == 'e' || *p == 'E')) { ++p; if (p < end && (*p == '+' || *p == '-')) ++p; if (p >= end || *p < '0' || *p > '9') return false; while (p < end && *p >= '0' && *p <= '9') ++p; }
        out.append((const char*)s, (size_t)(p - s));
        return true;
    }
    bool value(std::string& out) {
        if (++depth > 128) return false;                      // serde_json's recursion limit
        ws();
        if (p >= end) return false;
        bool ok;
        if (*p == '{') {
            out.push_back('{'); ++p; ws();
            if (p < end && *p == '}') { ++p; out.push_back('}'); ok = true; }
            else {
                ok = false;
                for (;;) {
                    ws(); if (!str(out)) break;
                    ws(); if (p >= end || *p != ':') break; ++p; out.push_back(':');
                    if (!value(out)) break;
                    ws(); if (p >= end) break;
                    if (*p == ',') { ++p; out.push_back(','); continue; }
                    if (*p == '}') { ++p; out.push_back('}'); ok = true; }
                    break;
                }
            }
        } else if (*p == '[') {
            out.push_back('['); ++p; ws();
            if (p < end && *p == ']') { ++p; out.push_back(']'); ok = true; }
            else {
                ok = false;
                for (;;) {
                    // a run of plain unsigned integers (digests, field elements: most of a proof's bytes) is copied as it
                    // stands; anything else (sign, fraction, exponent, leading zero, nesting) takes the general path
                    while (p < end && *p >= '1' && *p <= '9') {
                        const uint8_t* s0 = p;
                        while (p < end && *p >= '0' && *p <= '9') ++p;
                        if (p < end && *p == ',' && p + 1 < end && p[1] >= '1' && p[1] <= '9') { ++p; out.append((const char*)s0, (size_t)(p - s0)); continue; }
                        p = s0;
                        break;
                    }
                    if (!value(out)) break;
                    ws(); if (p >= end) break;
                    if (*p == ',') { ++p; out.push_back(','); continue; }
                    if (*p == ']') { ++p; out.push_back(']'); ok = true; }
                    break;
                }
            }
        } else if (*p == '"') ok = str(out);
        else if (*p == 't') ok = lit("true", out);
        else if (*p == 'f') ok = lit("false", out);
        else if (*p == 'n') ok = lit("null", out);
        else ok = num(out);
        --depth;
        return ok;
    }
};
}  // namespace

// deserialize's fast path (defined below, next to the typed item parsers): a Path / Leaves / Codeword item in exactly the
// text that materialize() writes is taken over in raw form; false = not such an item, nothing consumed.
static bool read_typed_item_raw(const uint8_t*& p, const uint8_t* end, std::string* raw);

// Nothing unwinds through the C ABI: an allocation failure inside an entry point comes back as SH_ENOMEM.
#define SH_GUARD_BEGIN try {
#define SH_GUARD_END(name)                                                                                  \
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, name ": out of host memory");               \
    } catch (...) { return sh_fail(SH_ENOMEM, name ": unexpected exception on the host side"); }

extern "C" {

int32_t sh_sha256(const uint8_t* in, size_t len, uint8_t out[32]) {
    if ((!in && len) || !out) return sh_fail(SH_EINVAL, "sh_sha256: null pointer");
    host_sha256(in, len, out);
    return SH_OK;
}
int32_t sh_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]) {
    if ((!in && len) || !out) return sh_fail(SH_EINVAL, "sh_blake2b512: null pointer");
    host_blake2b512(in, len, out);
    return SH_OK;
}
// MerkleTree::verify / is_leaf_in_tree, merkle.rs:110-154
int32_t sh_merkle_verify(const uint8_t root[64], const uint8_t leaf[64], const uint8_t* path, size_t path_len, size_t index, int32_t* ok) {
    if (!root || !leaf || !ok || (!path && path_len)) return sh_fail(SH_EINVAL, "sh_merkle_verify: null pointer");
    if (path_len == 0) return sh_fail(SH_EINVAL, "sh_merkle_verify: empty path (merkle.rs:126)");
    if (path_len < 8 * sizeof(size_t) && index >= ((size_t)1 << path_len)) return sh_fail(SH_EBOUNDS, "sh_merkle_verify: index >= 2^path_len (merkle.rs:122)");
    uint8_t cur[64], buf[128];
    memcpy(cur, leaf, 64);
    for (size_t i = 0; i < path_len; ++i) {
        if ((index & 1) == 0) { memcpy(buf, cur, 64); memcpy(buf + 64, path + 64 * i, 64); }   // merkle.rs:133-134
        else { memcpy(buf, path + 64 * i, 64); memcpy(buf + 64, cur, 64); }                   // merkle.rs:135-136
        host_blake2b512(buf, 128, cur);
        index >>= 1;
    }
    *ok = memcmp(cur, root, 64) == 0;
    return SH_OK;
}

int32_t sh_proofstream_new(sh_proofstream** out) {
    if (!out) return sh_fail(SH_EINVAL, "sh_proofstream_new: null out");
    *out = new (std::nothrow) sh_proofstream();
    return *out ? SH_OK : sh_fail(SH_ENOMEM, "sh_proofstream_new: out of memory");
}
int32_t sh_proofstream_free(sh_proofstream* ps) { delete ps; return SH_OK; }
int32_t sh_proofstream_push_json(sh_proofstream* ps, const uint8_t* json, size_t len) {       // proof_stream.rs:25-27
    SH_GUARD_BEGIN
    if (!ps || (!json && len)) return sh_fail(SH_EINVAL, "sh_proofstream_push_json: null pointer");
    if (len && json[0] <= 0x03) return sh_fail(SH_EINVAL, "sh_proofstream_push_json: not JSON text");   // the marker of a path kept in raw form
    ps->items.emplace_back((const char*)json, len);
    return SH_OK;
    SH_GUARD_END("sh_proofstream_push_json")
}
int32_t sh_proofstream_push_merkle_root(sh_proofstream* ps, const uint8_t root[64]) {         // merkle.rs:9-13 shape
    SH_GUARD_BEGIN
    if (!ps || !root) return sh_fail(SH_EINVAL, "sh_proofstream_push_merkle_root: null pointer");
    char buf[24 + 64 * 4 + 4];
    size_t o = 24;
    memcpy(buf, "{\"MerkleRoot\":{\"inner\":[", 24);
    for (int i = 0; i < 64; ++i) { memcpy(buf + o, BYTE_DEC.txt[root[i]], 4); o += BYTE_DEC.len[root[i]] + 1; }   // "v," per byte: one 4-byte copy
    memcpy(buf + o - 1, "]}}", 3);                                                                                 // the last comma becomes the bracket
    ps->items.emplace_back(buf, o + 2);
    return SH_OK;
    SH_GUARD_END("sh_proofstream_push_merkle_root")
}
int32_t sh_proofstream_push_codeword(sh_proofstream* ps, const uint64_t* elems, size_t n) {
    SH_GUARD_BEGIN
    if (!ps || (!elems && n)) return sh_fail(SH_EINVAL, "sh_proofstream_push_codeword: null pointer");
    std::string s(1 + n * 16, '\0');                   // raw until read, like a path (see sh_proofstream_push_path)
    s[0] = LAZY_CODEWORD;
    if (n) memcpy(&s[1], elems, n * 16);
    ps->items.push_back(std::move(s));
    return SH_OK;
    SH_GUARD_END("sh_proofstream_push_codeword")
}
// Query-phase items: {"Leaves":[a,b,c]} — the three codeword values of one colinearity test — and
// {"Path":[{"inner":[..]},..]} — a Merkle authentication path, Vec<MerkleHash> as serde writes it (merkle.rs:9-13).
int32_t sh_proofstream_push_leaves(sh_proofstream* ps, const uint64_t* elems, size_t n) {
    SH_GUARD_BEGIN
    if (!ps || (!elems && n)) return sh_fail(SH_EINVAL, "sh_proofstream_push_leaves: null pointer");
    std::string s(1 + n * 16, '\0');
    s[0] = LAZY_LEAVES;
    if (n) memcpy(&s[1], elems, n * 16);
    ps->items.push_back(std::move(s));
    return SH_OK;
    SH_GUARD_END("sh_proofstream_push_leaves")
}
int32_t sh_proofstream_push_path(sh_proofstream* ps, const uint8_t* digests, size_t n) {
    SH_GUARD_BEGIN
    if (!ps || (!digests && n)) return sh_fail(SH_EINVAL, "sh_proofstream_push_path: null pointer");
    // Kept as the raw digests behind a marker byte and written out as JSON when something reads the item (pull, serialize,
    // a Fiat-Shamir digest) — as the reference's push stores the object and serialises on demand (proof_stream.rs:25-27,
    // 38-40).  A proof's paths are most of its bytes; the prover never reads them back.
    std::string s(1 + n * 64, '\0');
    s[0] = LAZY_PATH;
    if (n) memcpy(&s[1], digests, n * 64);
    ps->items.push_back(std::move(s));
    return SH_OK;
    SH_GUARD_END("sh_proofstream_push_path")
}
int32_t sh_proofstream_len(sh_proofstream* ps, size_t* n_items, size_t* read_index) {
    if (!ps) return sh_fail(SH_EINVAL, "sh_proofstream_len: null pointer");
    if (n_items) *n_items = ps->items.size();
    if (read_index) *read_index = ps->read_index;
    return SH_OK;
}
int32_t sh_proofstream_pull(sh_proofstream* ps, const uint8_t** json, size_t* len) {          // proof_stream.rs:29-36
    if (!ps || !json || !len) return sh_fail(SH_EINVAL, "sh_proofstream_pull: null pointer");
    if (ps->read_index >= ps->items.size()) return sh_fail(SH_EBOUNDS, "Read index is out of bounds");
    materialize(ps->items[ps->read_index]);
    const std::string& s = ps->items[ps->read_index++];
    *json = (const uint8_t*)s.data();
    *len = s.size();
    return SH_OK;
}
int32_t sh_proofstream_serialize(sh_proofstream* ps, uint8_t* out, size_t cap, size_t* needed) {   // proof_stream.rs:38-40
    SH_GUARD_BEGIN
    if (!ps || !needed) return sh_fail(SH_EINVAL, "sh_proofstream_serialize: null pointer");
    materialize_all(ps->items, ps->items.size());
    std::string s = serialize_items(ps->items, ps->items.size());
    *needed = s.size();
    if (out && cap >= s.size()) memcpy(out, s.data(), s.size());
    else if (out) return sh_fail(SH_EINVAL, "sh_proofstream_serialize: buffer too small");
    return SH_OK;
    SH_GUARD_END("sh_proofstream_serialize")
}
int32_t sh_proofstream_deserialize(const uint8_t* bytes, size_t len, sh_proofstream** out) {       // proof_stream.rs:42-49
    SH_GUARD_BEGIN
    if (!bytes || !out) return sh_fail(SH_EINVAL, "sh_proofstream_deserialize: null pointer");
    JsonScan sc{bytes, bytes + len};
    sc.ws();
    if (sc.p >= sc.end || *sc.p != '[') return sh_fail(SH_EPARSE, "Serialization error: expected a JSON array");
    ++sc.p;
    sh_proofstream* ps = new (std::nothrow) sh_proofstream();
    if (!ps) return sh_fail(SH_ENOMEM, "sh_proofstream_deserialize: out of memory");
    sc.ws();
    bool ok = false;
    if (sc.p < sc.end && *sc.p == ']') { ++sc.p; ok = true; }
    else {
        for (;;) {
            std::string item;
            sc.ws();
            // The bulk of a proof (authentication paths, leaf values, the last codeword) in canonical text is kept raw, as the
            // prover's own pushes keep it: serialising it again gives the same bytes, and the verifier reads it without parsing
            // the text a second time.  Everything else — other items, the same items with white space or non-canonical numbers —
            // is validated and compacted by the general scanner and kept as text.
            if (!read_typed_item_raw(sc.p, sc.end, &item) && !sc.value(item)) break;
            ps->items.push_back(std::move(item));
            sc.ws();
            if (sc.p >= sc.end) break;
            if (*sc.p == ',') { ++sc.p; continue; }
            if (*sc.p == ']') { ++sc.p; ok = true; }
            break;
        }
    }
    sc.ws();
    if (!ok || sc.p != sc.end) { delete ps; return sh_fail(SH_EPARSE, "Serialization error: malformed JSON"); }
    *out = ps;
    return SH_OK;
    SH_GUARD_END("sh_proofstream_deserialize")
}
int32_t sh_proofstream_prover_fiat_shamir(sh_proofstream* ps, uint8_t out[32]) {                   // proof_stream.rs:51-56
    if (!ps || !out) return sh_fail(SH_EINVAL, "sh_proofstream_prover_fiat_shamir: null pointer");
    if (!ps->running) {
        ps->running = new (std::nothrow) Sha256Ctx();
        if (!ps->running) return sh_fail(SH_ENOMEM, "sh_proofstream_prover_fiat_shamir: out of memory");
        ps->running->h.update((const uint8_t*)"[", 1);
        ps->absorbed = 0;
    }
    for (; ps->absorbed < ps->items.size(); ++ps->absorbed) {
        if (ps->absorbed) ps->running->h.update((const uint8_t*)",", 1);
        materialize(ps->items[ps->absorbed]);
        const std::string& it = ps->items[ps->absorbed];
        ps->running->h.update((const uint8_t*)it.data(), it.size());
    }
    Sha256 fin = ps->running->h;                 // copy the midstate, close the array on the copy
    fin.update((const uint8_t*)"]", 1);
    fin.finish(out);
    return SH_OK;
}
int32_t sh_proofstream_verifier_fiat_shamir(sh_proofstream* ps, uint8_t out[32]) {                 // proof_stream.rs:58-64
    SH_GUARD_BEGIN
    if (!ps || !out) return sh_fail(SH_EINVAL, "sh_proofstream_verifier_fiat_shamir: null pointer");
    materialize_all(ps->items, ps->read_index);
    std::string s = serialize_items(ps->items, ps->read_index);
    host_sha256((const uint8_t*)s.data(), s.size(), out);
    return SH_OK;
    SH_GUARD_END("sh_proofstream_verifier_fiat_shamir")
}

}  // extern "C"

// ------------------------------------------------------------------ typed item parsers for the verifier (internal)
namespace {
struct ItemParse {
    const uint8_t* p; const uint8_t* end;
    bool lit(const char* t) { size_t n = strlen(t); if ((size_t)(end - p) < n || memcmp(p, t, n)) return false; p += n; return true; }
    bool u128(unsigned __int128* v) {                         // bare decimal, no sign, no leading zeros except "0", < 2^128
        if (p >= end || *p < '0' || *p > '9') return false;
        if (*p == '0' && p + 1 < end && p[1] >= '0' && p[1] <= '9') return false;
        // 38 digits cannot overflow (10^38 < 2^128); only a 39th needs the check — no 128-bit division per digit
        unsigned __int128 acc = 0;
        int digits = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            const unsigned d = (unsigned)(*p - '0');
            if (++digits > 39) return false;
            if (digits == 39) {
                const unsigned __int128 lim = ~(unsigned __int128)0;
                if (acc > lim / 10 || (acc == lim / 10 && d > (unsigned)(lim % 10))) return false;
            }
            acc = acc * 10 + d; ++p;
        }
        *v = acc;
        return true;
    }
    bool byte(uint8_t* v) {                                   // the same grammar for a value <= 255: at most three digits
        if (p >= end || *p < '0' || *p > '9') return false;
        unsigned acc = (unsigned)(*p++ - '0');
        if (acc == 0) { if (p < end && *p >= '0' && *p <= '9') return false; *v = 0; return true; }
        if (p < end && *p >= '0' && *p <= '9') { acc = acc * 10 + (unsigned)(*p++ - '0');
            if (p < end && *p >= '0' && *p <= '9') { acc = acc * 10 + (unsigned)(*p++ - '0');
                if (p < end && *p >= '0' && *p <= '9') return false; } }
        if (acc > 255) return false;
        *v = (uint8_t)acc;
        return true;
    }
    bool hash(uint8_t out[64]) {                              // {"inner":[b0,...,b63]}
        if (!lit("{\"inner\":[")) return false;
        for (int i = 0; i < 64; ++i) {
            if (p < end && *p == ',') { if (!i) return false; ++p; }
            else if (i) return false;
            if (!byte(out + i)) return false;
        }
        return lit("]}");
    }
};
}  // namespace

bool host_parse_merkle_root(const uint8_t* json, size_t len, uint8_t out[64]) {
    ItemParse ip{json, json + len};
    return ip.lit("{\"MerkleRoot\":") && ip.hash(out) && ip.lit("}") && ip.p == ip.end;
}
bool host_parse_elems(const uint8_t* json, size_t len, const char* tag, std::vector<uint64_t>* out) {   // {"<tag>":[dec,...]}
    ItemParse ip{json, json + len};
    out->clear();
    if (!ip.lit("{\"") || !ip.lit(tag) || !ip.lit("\":[")) return false;
    if (ip.lit("]")) return ip.lit("}") && ip.p == ip.end;
    for (;;) {
        unsigned __int128 v;
        if (!ip.u128(&v)) return false;
        out->push_back((uint64_t)v); out->push_back((uint64_t)(v >> 64));
        if (ip.lit(",")) continue;
        return ip.lit("]}") && ip.p == ip.end;
    }
}
bool host_parse_path(const uint8_t* json, size_t len, std::vector<uint8_t>* out) {                      // {"Path":[{"inner":[..]},...]}
    ItemParse ip{json, json + len};
    out->clear();
    if (!ip.lit("{\"Path\":[")) return false;
    if (ip.lit("]")) return ip.lit("}") && ip.p == ip.end;
    for (;;) {
        uint8_t h[64];
        if (!ip.hash(h)) return false;
        out->insert(out->end(), h, h + 64);
        if (ip.lit(",")) continue;
        return ip.lit("]}") && ip.p == ip.end;
    }
}

// Typed pulls for the verifier (internal): an item that is still raw is handed over as it is, one that is text is parsed.
// false: no item left, or not an item of that type.
bool host_ps_pull_path(sh_proofstream* ps, std::vector<uint8_t>* out) {
    if (ps->read_index >= ps->items.size()) return false;
    const std::string& it = ps->items[ps->read_index++];
    if (!it.empty() && it[0] == LAZY_PATH) { out->assign((const uint8_t*)it.data() + 1, (const uint8_t*)it.data() + it.size()); return true; }
    if (!it.empty() && (it[0] == LAZY_CODEWORD || it[0] == LAZY_LEAVES)) return false;
    return host_parse_path((const uint8_t*)it.data(), it.size(), out);
}
bool host_ps_pull_elems(sh_proofstream* ps, const char* tag, std::vector<uint64_t>* out) {
    if (ps->read_index >= ps->items.size()) return false;
    const std::string& it = ps->items[ps->read_index++];
    const char want = strcmp(tag, "Leaves") == 0 ? LAZY_LEAVES : strcmp(tag, "Codeword") == 0 ? LAZY_CODEWORD : 0;
    if (!it.empty() && (it[0] == LAZY_PATH || it[0] == LAZY_CODEWORD || it[0] == LAZY_LEAVES)) {
        if (it[0] != want) return false;
        out->resize((it.size() - 1) / 8);
        if (!out->empty()) memcpy(out->data(), it.data() + 1, out->size() * 8);
        return true;
    }
    return host_parse_elems((const uint8_t*)it.data(), it.size(), tag, out);
}

static bool read_typed_item_raw(const uint8_t*& p, const uint8_t* end, std::string* out) {
    ItemParse ip{p, end};
    std::string buf, *raw = &buf;                      // `out` is touched only on success
    if (ip.lit("{\"Path\":[")) {
        raw->assign(1, LAZY_PATH);
        if (!ip.lit("]")) {
            for (;;) {
                uint8_t h[64];
                if (!ip.hash(h)) return false;
                raw->append((const char*)h, 64);
                if (ip.lit(",")) continue;
                if (!ip.lit("]")) return false;
                break;
            }
        }
    } else {
        const bool leaves = ip.lit("{\"Leaves\":[");
        if (!leaves && !ip.lit("{\"Codeword\":[")) return false;
        raw->assign(1, leaves ? LAZY_LEAVES : LAZY_CODEWORD);
        if (!ip.lit("]")) {
            for (;;) {
                unsigned __int128 v;
                if (!ip.u128(&v)) return false;
                const uint64_t w[2] = {(uint64_t)v, (uint64_t)(v >> 64)};
                raw->append((const char*)w, 16);
                if (ip.lit(",")) continue;
                if (!ip.lit("]")) return false;
                break;
            }
        }
    }
    if (!ip.lit("}")) return false;
    if (ip.p < end && *ip.p != ',' && *ip.p != ']' && *ip.p != ' ' && *ip.p != '\n' && *ip.p != '\t' && *ip.p != '\r') return false;
    p = ip.p;
    *out = std::move(buf);
    return true;
}

// host_internal.h — shared by the host translation units of libstark_hip.so (internal).
#pragma once
#include <stddef.h>
#include <stdint.h>

// records `msg` as the thread-local last error and returns `code`
int32_t sh_fail(int32_t code, const char* msg);
int32_t sh_failf(int32_t code, const char* fmt, ...);

void host_sha256(const uint8_t* in, size_t len, uint8_t out[32]);
void host_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]);

#ifdef __cplusplus
#include <vector>
// typed parsers of the FRI proof items (compact JSON exactly as the prover writes it); false on any deviation
bool host_parse_merkle_root(const uint8_t* json, size_t len, uint8_t out[64]);
bool host_parse_elems(const uint8_t* json, size_t len, const char* tag, std::vector<uint64_t>* out);   // {lo,hi} pairs
bool host_parse_path(const uint8_t* json, size_t len, std::vector<uint8_t>* out);                      // n x 64 bytes
struct sh_proofstream;
bool host_ps_pull_path(sh_proofstream* ps, std::vector<uint8_t>* out);                                 // next item as a path: raw if still raw, else parsed
bool host_ps_pull_elems(sh_proofstream* ps, const char* tag, std::vector<uint64_t>* out);              // next item as {"<tag>":[...]} ("Leaves" / "Codeword")
#endif

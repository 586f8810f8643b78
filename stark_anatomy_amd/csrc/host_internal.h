// host_internal.h — shared by the host translation units of libstark_hip.so (internal).
#pragma once
#include <stddef.h>
#include <stdint.h>

// records `msg` as the thread-local last error and returns `code`
int32_t sh_fail(int32_t code, const char* msg);
int32_t sh_failf(int32_t code, const char* fmt, ...);

void host_sha256(const uint8_t* in, size_t len, uint8_t out[32]);
void host_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]);

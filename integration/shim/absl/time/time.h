// Syntax-check shim (integration/README.md): declaration-level stand-in, never linked.
#pragma once
#include <cstdint>
namespace absl {
class Duration { public: int64_t ns = 0; };
class Time { public: int64_t ns = 0; };
inline Time InfiniteFuture() { return Time{INT64_MAX}; }
inline Duration Milliseconds(int64_t n) { return Duration{n * 1000000}; }
inline Duration Seconds(int64_t n) { return Duration{n * 1000000000}; }
inline Time Now() { return Time{}; }
inline Duration operator-(Time a, Time b) { return Duration{a.ns - b.ns}; }
inline bool operator<(Duration a, Duration b) { return a.ns < b.ns; }
inline int64_t ToInt64Microseconds(Duration d) { return d.ns / 1000; }
}  // namespace absl

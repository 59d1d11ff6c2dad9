// Syntax-check shim (integration/README.md): declaration-level stand-in, never linked.
#pragma once
#include <cstdio>
#include <string>
namespace absl {
template <typename... A>
inline std::string StrFormat(const char* fmt, const A&... a) {
  char buf[512];
  snprintf(buf, sizeof buf, fmt, a...);
  return buf;
}
}  // namespace absl

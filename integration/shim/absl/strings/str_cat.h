// Syntax-check shim (integration/README.md): declaration-level stand-in, never linked.
#pragma once
#include <sstream>
#include <string>
namespace absl {
template <typename... A>
inline std::string StrCat(const A&... a) {
  std::ostringstream os;
  (void)std::initializer_list<int>{((os << a), 0)...};
  return os.str();
}
}  // namespace absl

#pragma once
#include <cstring>
#include <string>
#include <vector>
#include "absl/strings/string_view.h"
#include "absl/types/optional.h"
namespace absl {
enum class StatusCode : int { kOk = 0, kCancelled = 1, kUnknown = 2, kInvalidArgument = 3, kDeadlineExceeded = 4,
  kNotFound = 5, kAlreadyExists = 6, kPermissionDenied = 7, kResourceExhausted = 8, kFailedPrecondition = 9,
  kAborted = 10, kOutOfRange = 11, kUnimplemented = 12, kInternal = 13, kUnavailable = 14, kDataLoss = 15,
  kUnauthenticated = 16 };
class Status {
 public:
  Status() = default;
  Status(StatusCode c, string_view m) : code_(c), msg_(m) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  string_view message() const { return msg_; }
  std::string ToString() const { return msg_; }
 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
}  // namespace absl

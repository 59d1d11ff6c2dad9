#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>
#include "absl/base/thread_annotations.h"
#include "absl/time/time.h"
namespace absl {
class Mutex {
 public:
  void Lock() { m_.lock(); }
  void Unlock() { m_.unlock(); }
  bool TryLock() { return m_.try_lock(); }
  std::mutex m_;
};
class MutexLock {
 public:
  explicit MutexLock(Mutex* m) : m_(m) { m_->Lock(); }
  ~MutexLock() { m_->Unlock(); }
 private:
  Mutex* m_;
};
class ReleasableMutexLock {
 public:
  explicit ReleasableMutexLock(Mutex* m) : m_(m) { m_->Lock(); }
  ~ReleasableMutexLock() { if (m_) m_->Unlock(); }
  void Release() { m_->Unlock(); m_ = nullptr; }
 private:
  Mutex* m_;
};
class CondVar {
 public:
  void Signal() { cv_.notify_one(); }
  void SignalAll() { cv_.notify_all(); }
  void Wait(Mutex*) {}
  bool WaitWithTimeout(Mutex*, Duration) { return false; }
  bool WaitWithDeadline(Mutex*, Time) { return false; }
 private:
  std::condition_variable cv_;
};
}  // namespace absl

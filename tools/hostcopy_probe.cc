// hostcopy_probe: what ONE host core gives the copy in front of an endpoint write -- the 130 slices of a 1 MiB message
// (65 x [9 B frame header][16 KiB payload]) into a 4 MiB pinned send buffer -- with the C library's memcpy, with
// non-temporal 32-byte stores (no read-for-ownership of the destination lines), and with `rep movsb`; next to it the
// byte sum the reading thread of tools/endpoint_stream runs over what was delivered.  The vtable leg of bench.py cannot be
// faster than the slower of the two.
// build: g++ -O2 -std=c++17 tools/hostcopy_probe.cc -o tools/hostcopy_probe -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
#include <emmintrin.h>
#include <hip/hip_runtime_api.h>
#include <immintrin.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__attribute__((target("avx2"))) static void copy_nt(uint8_t* dst, const uint8_t* src, size_t n) {
  // head up to a 32-byte boundary of the destination
  size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
  if (head > n) head = n;
  memcpy(dst, src, head);
  dst += head; src += head; n -= head;
  size_t k = 0;
  for (; k + 128 <= n; k += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + k)), b = _mm256_loadu_si256((const __m256i*)(src + k + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + k + 64)), d = _mm256_loadu_si256((const __m256i*)(src + k + 96));
    _mm256_stream_si256((__m256i*)(dst + k), a);
    _mm256_stream_si256((__m256i*)(dst + k + 32), b);
    _mm256_stream_si256((__m256i*)(dst + k + 64), c);
    _mm256_stream_si256((__m256i*)(dst + k + 96), d);
  }
  memcpy(dst + k, src + k, n - k);
}
static void copy_movsb(uint8_t* dst, const uint8_t* src, size_t n) {
  asm volatile("rep movsb" : "+D"(dst), "+S"(src), "+c"(n) : : "memory");
}
static uint64_t sum_bytes(const uint8_t* b, size_t n) {
  __m128i a0 = _mm_setzero_si128(), a1 = a0, a2 = a0, a3 = a0;
  const __m128i z = _mm_setzero_si128();
  size_t k = 0;
  for (; k + 64 <= n; k += 64) {
    a0 = _mm_add_epi64(a0, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(b + k)), z));
    a1 = _mm_add_epi64(a1, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(b + k + 16)), z));
    a2 = _mm_add_epi64(a2, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(b + k + 32)), z));
    a3 = _mm_add_epi64(a3, _mm_sad_epu8(_mm_loadu_si128((const __m128i*)(b + k + 48)), z));
  }
  a0 = _mm_add_epi64(_mm_add_epi64(a0, a1), _mm_add_epi64(a2, a3));
  uint64_t t = (uint64_t)_mm_cvtsi128_si64(a0) + (uint64_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(a0, a0));
  for (; k < n; k++) t += b[k];
  return t;
}

__attribute__((target("avx2"))) static uint64_t sum_bytes_avx2(const uint8_t* b, size_t n) {
  __m256i a0 = _mm256_setzero_si256(), a1 = a0, a2 = a0, a3 = a0;
  const __m256i z = _mm256_setzero_si256();
  size_t k = 0;
  for (; k + 128 <= n; k += 128) {
    _mm_prefetch((const char*)(b + k + 1024), _MM_HINT_T0);
    _mm_prefetch((const char*)(b + k + 1088), _MM_HINT_T0);
    a0 = _mm256_add_epi64(a0, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i*)(b + k)), z));
    a1 = _mm256_add_epi64(a1, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i*)(b + k + 32)), z));
    a2 = _mm256_add_epi64(a2, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i*)(b + k + 64)), z));
    a3 = _mm256_add_epi64(a3, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i*)(b + k + 96)), z));
  }
  a0 = _mm256_add_epi64(_mm256_add_epi64(a0, a1), _mm256_add_epi64(a2, a3));
  uint64_t l[4];
  _mm256_storeu_si256((__m256i*)l, a0);
  uint64_t t = l[0] + l[1] + l[2] + l[3];
  for (; k < n; k++) t += b[k];
  return t;
}

int main(int argc, char** argv) {
  const int msgs = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t cap = 4u << 20;
  uint8_t* pinned[2];
  for (auto& p : pinned)
    if (hipHostMalloc((void**)&p, cap, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { printf("no pinned memory\n"); return 1; }
  // big pinned area: what the reader walks (receive windows are fresh memory every time)
  const size_t win = 256u << 20;
  uint8_t* window;
  if (hipHostMalloc((void**)&window, win, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { printf("no window\n"); return 1; }
  memset(window, 1, win);
  std::vector<std::vector<uint8_t>> slices;
  for (int f = 0; f < 65; f++) {
    slices.emplace_back(9, (uint8_t)f);
    slices.emplace_back(f == 64 ? 600 : 16384, (uint8_t)(f + 1));
  }
  size_t per_msg = 0;
  for (auto& s : slices) per_msg += s.size();
  auto run = [&](const char* name, void (*cp)(uint8_t*, const uint8_t*, size_t)) {
    for (int rep = 0; rep < 2; rep++) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int m = 0; m < msgs; m++) {
        uint8_t* d = pinned[m & 1];
        size_t off = 0;
        for (auto& s : slices) { cp(d + off, s.data(), s.size()); off += s.size(); }
      }
      _mm_sfence();
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("copy %-10s %6.2f GiB/s  (%.1f us per 1 MiB message)\n", name, per_msg * (double)msgs / sec / (1 << 30), sec / msgs * 1e6);
    }
  };
  run("memcpy", [](uint8_t* d, const uint8_t* s, size_t n) { memcpy(d, s, n); });
  run("nt-avx2", [](uint8_t* d, const uint8_t* s, size_t n) { if (n >= 512) copy_nt(d, s, n); else memcpy(d, s, n); });
  run("rep-movsb", [](uint8_t* d, const uint8_t* s, size_t n) { copy_movsb(d, s, n); });
  // the reader's byte sum over fresh pinned memory
  for (int rep = 0; rep < 2; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t t = 0;
    for (size_t o = 0; o + 16384 <= win; o += 16384) t += sum_bytes(window + o, 16384);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rep) printf("byte sum over 256 MiB of pinned memory: %6.2f GiB/s (sum %llu)\n", win / sec / (1 << 30), (unsigned long long)t);
  }
  for (int rep = 0; rep < 2; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t t = 0;
    for (size_t o = 0; o + 16384 <= win; o += 16384) t += sum_bytes_avx2(window + o, 16384);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rep) printf("the same with 32-byte psadbw + prefetch:  %6.2f GiB/s (sum %llu)\n", win / sec / (1 << 30), (unsigned long long)t);
  }
  return 0;
}

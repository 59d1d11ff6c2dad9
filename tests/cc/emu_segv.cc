// TEST INFRASTRUCTURE: with EMU_SEGV_TRACE=1 a fault inside libgrdma_emu.so prints the faulting address and a
// backtrace (an out-of-bounds access that the GPU's coarse page mapping would let through shows up here).
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
namespace {
void on_segv(int, siginfo_t* si, void*) {
  char msg[96];
  int n = snprintf(msg, sizeof(msg), "emu: SIGSEGV at address %p\n", si->si_addr);
  if (write(2, msg, (size_t)n) < 0) {}
  void* bt[48];
  int k = backtrace(bt, 48);
  backtrace_symbols_fd(bt, k, 2);
  _exit(139);
}
struct installer {
  installer() {
    const char* e = getenv("EMU_SEGV_TRACE");
    if (!e || e[0] != '1') return;
    static char alt[1 << 16];
    stack_t ss;
    ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    sa.sa_sigaction = on_segv;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
  }
} g_installer;
}  // namespace

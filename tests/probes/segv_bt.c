/* LD_PRELOAD helper for the GPU box: prints the native backtrace of the faulting thread on SIGSEGV / SIGBUS / SIGABRT, then
 * hands the signal on (python -X faulthandler chains to the handler installed before it).  Build: gcc -shared -fPIC -O1 -o
 * tests/probes/libsegv_bt.so tests/probes/segv_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* info, void* ctx) {
  void* frames[96];
  const char msg[] = "\n==== native backtrace (segv_bt) ====\n";
  (void)!write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}

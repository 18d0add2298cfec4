/* Test infrastructure: print the NATIVE call stack when the process aborts (glibc heap checks, HIP runtime asserts) or
 * takes a fatal signal.  Python's faulthandler only shows Python frames; round 1 and round 3 each lost a whole GPU test run
 * to an abort whose C frames nobody saw.  Loaded by tests/conftest.py with ctypes; never part of the product. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction previous;

static void handler(int sig) {
  void* frames[96];
  const char head[] = "\n==== native backtrace (tests/emu/abort_trace.c) ====\n";
  if (write(2, head, sizeof(head) - 1) < 0) {}
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  const char tail[] = "==== end of native backtrace ====\n";
  if (write(2, tail, sizeof(tail) - 1) < 0) {}
  sigaction(sig, &previous, 0); /* Python's faulthandler (its frames), then the default action */
  raise(sig);
}

int seld_install_abort_trace(void) {
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now: not async-signal-safe later */
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = handler;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGABRT, &sa, &previous);
  return 0;
}

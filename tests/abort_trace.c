/* Test infrastructure only: a SIGABRT handler that prints the native backtrace of the thread that called abort() before
 * the default action runs.  An abort raised inside the HIP / HSA runtime (queue error, memory access fault) or by glibc
 * otherwise leaves only Python frames in the log.  Built and loaded by tests/conftest.py. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
  static const char msg[] = "\n[abort_trace] SIGABRT: native backtrace of the aborting thread:\n";
  void* frames[64];
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

int abort_trace_install(void) {
  struct sigaction sa;
  void* warm[4];
  (void)backtrace(warm, 4);          /* loads libgcc now, not inside the handler */
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_abort;
  sigemptyset(&sa.sa_mask);
  return sigaction(SIGABRT, &sa, 0);
}

/* SIGABRT / SIGSEGV handler that prints the NATIVE backtrace (glibc backtrace_symbols_fd) to stderr before the process
 * dies: names the faulting frame of a runtime abort without a debugger.  Diagnostic only (scripts/repro_nested_capture_fork.py):
 *   gcc -shared -fPIC -O1 -o /tmp/libabort_bt.so scripts/abort_bt.c ;  ctypes.CDLL(...).abort_bt_install() */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
  void* frames[64];
  const char msg[] = "\n[abort_bt] native backtrace at fatal signal:\n";
  (void)!write(2, msg, sizeof msg - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void abort_bt_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
}

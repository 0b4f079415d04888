/* r06, the CPU-suite stall: a signal handler that prints the NATIVE backtrace of the thread it runs in -- sent with
 * tgkill to every task of the process by tests/stall_probe.py's watchdog, it shows what the OpenMP / MKL worker threads execute, which
 * faulthandler (Python frames only) cannot.  Diagnostic only:  gcc -shared -fPIC -O1 -o /tmp/libstall_bt.so scripts/stall_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int out_fd = 2;
static volatile int lock_ = 0;

static void handler(int sig) {
  (void)sig;
  void* frames[48];
  char head[96];
  while (__sync_lock_test_and_set(&lock_, 1)) {}
  int n = snprintf(head, sizeof head, "\n[stall_bt] tid %ld\n", (long)syscall(SYS_gettid));
  (void)!write(out_fd, head, n);
  n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, out_fd);
  __sync_lock_release(&lock_);
}

void stall_bt_install(int sig, int fd) {
  struct sigaction sa;
  void* warm[4];
  backtrace(warm, 4);                 /* loads libgcc now, not inside the handler */
  out_fd = fd;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sa.sa_flags = SA_RESTART;
  sigaction(sig, &sa, 0);
}

int stall_bt_kick(int sig, long tid) { return (int)syscall(SYS_tgkill, (long)getpid(), tid, sig); }

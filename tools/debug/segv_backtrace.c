/* Debugging aid, not part of the library: on SIGSEGV / SIGBUS / SIGABRT write the native call stack of the
 * faulting thread to stderr (glibc backtrace), then let the signal take its default course.
 * build: gcc -shared -fPIC -O1 -o tools/debug/libsegvbt.so tools/debug/segv_backtrace.c
 * use:   python tools/debug/with_backtrace.py bench.py --no-cpu-baseline ...                                  */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig) {
  void* frames[96];
  const char msg[] = "\n---- native stack of the faulting thread ----\n";
  if (write(2, msg, sizeof msg - 1) < 0) {}
  const int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void segvbt_install(void) {
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fault;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
}

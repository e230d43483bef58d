/* segv_backtrace.c -- diagnostic shim for tools/rccl_capture_probe.py: a SIGSEGV handler that prints the C backtrace
 * (module + offset per frame, glibc backtrace_symbols_fd) to stderr and exits with 139.  No debugger in this image.
 *   gcc -O0 -g -shared -fPIC -o tools/libsegv_bt.so tools/segv_backtrace.c                                            */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_segv(int sig) {
  void *bt[96];
  const char m[] = "\n=== SIGSEGV backtrace (module(+offset)) ===\n";
  (void)sig;
  (void)!write(2, m, sizeof m - 1);
  backtrace_symbols_fd(bt, backtrace(bt, 96), 2);
  _exit(139);
}
void segv_backtrace_install(void) { signal(SIGSEGV, on_segv); }

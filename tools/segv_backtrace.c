/* segv_backtrace.c -- diagnostic shim for tools/rccl_capture_probe.py: a SIGSEGV handler ON ITS OWN STACK (sigaltstack: a stack
 * overflow leaves no room for a handler on the faulting one) that prints the faulting address, the stack pointer and the C
 * backtrace (module(+offset) per frame, glibc backtrace_symbols_fd) to stderr and exits with 139.  No debugger in this image.
 *   gcc -O0 -g -shared -fPIC -o tools/libsegv_bt.so tools/segv_backtrace.c                                            */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>
static void on_segv(int sig, siginfo_t *si, void *uc_) {
  void *bt[128];
  char line[256];
  ucontext_t *uc = (ucontext_t *)uc_;
  (void)sig;
  int n = snprintf(line, sizeof line, "\n=== SIGSEGV backtrace (module(+offset)) === fault address %p, rsp %p, rip %p\n", si->si_addr,
                   (void *)uc->uc_mcontext.gregs[REG_RSP], (void *)uc->uc_mcontext.gregs[REG_RIP]);
  (void)!write(2, line, (size_t)n);
  backtrace_symbols_fd(bt, backtrace(bt, 128), 2);
  _exit(139);
}
void segv_backtrace_install(void) {
  static char *alt;
  stack_t ss;
  struct sigaction sa;
  if (!alt) alt = malloc(1 << 18);
  ss.ss_sp = alt; ss.ss_size = 1 << 18; ss.ss_flags = 0;
  sigaltstack(&ss, NULL);
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, NULL);
  sigaction(SIGBUS, &sa, NULL);
}

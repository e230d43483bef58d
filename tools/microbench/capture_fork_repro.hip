// capture_fork_repro.hip -- diagnostic: which shape of forked streams inside a stream capture makes hipStreamEndCapture
// recurse without end?  tools/rccl_capture_probe.py's backtrace (profiles/r04_rccl_capture_probe.txt) shows the crash of the
// "RCCL on a stream of its own inside a capture" form to be a stack overflow in a self-recursive function of libamdhip64.so that walks a
// stream's list of forked ("parallel") capture streams and resets their capture state -- hip::Stream::EndCapture -- i.e. it happens at
// hipStreamEndCapture, not in RCCL's kernels.  RCCL forks an internal stream from the stream it is called on and joins it back; called on
// a stream that is itself a fork of the capturing stream that makes a two-level fork.  Each variant runs in a CHILD process (fork
// before any HIP call in the parent).
//   hipcc --offload-arch=gfx950 -O2 capture_fork_repro.hip -o capture_fork_repro && ./capture_fork_repro
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
__global__ void k(double *p) { p[threadIdx.x] += 1.0; }
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)
static int variant(int v) {
  hipStream_t A, C, D;
  hipEvent_t e1, e2, e3, e4, e5;
  double *p;
  CHK(hipMalloc(&p, 1024));
  CHK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&D, hipStreamNonBlocking));
  for (hipEvent_t *e : {&e1, &e2, &e3, &e4, &e5}) CHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  hipGraph_t g;
  CHK(hipStreamBeginCapture(A, hipStreamCaptureModeRelaxed));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, A, p);
  CHK(hipEventRecord(e1, A)); CHK(hipStreamWaitEvent(C, e1, 0));            // C forks from A
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, C, p);
  if (v >= 1) {                                                             // D forks from C (two levels), joins C
    CHK(hipEventRecord(e2, C)); CHK(hipStreamWaitEvent(D, e2, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, D, p);
    CHK(hipEventRecord(e3, D)); CHK(hipStreamWaitEvent(C, e3, 0));
  }
  if (v >= 2) {                                                             // ... and D is used a second time from C (a second group of the library)
    CHK(hipEventRecord(e2, C)); CHK(hipStreamWaitEvent(D, e2, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, D, p);
    CHK(hipEventRecord(e3, D)); CHK(hipStreamWaitEvent(C, e3, 0));
  }
  if (v >= 3) {                                                             // D also waits for the ORIGIN's event (joined from two parents)
    CHK(hipEventRecord(e5, A)); CHK(hipStreamWaitEvent(D, e5, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, D, p);
    CHK(hipEventRecord(e3, D)); CHK(hipStreamWaitEvent(C, e3, 0));
  }
  CHK(hipEventRecord(e4, C)); CHK(hipStreamWaitEvent(A, e4, 0));            // C joins A
  CHK(hipStreamEndCapture(A, &g));
  hipGraphExec_t ex;
  CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CHK(hipGraphLaunch(ex, A)); CHK(hipStreamSynchronize(A));
  return 0;
}
int main() {
  const char *names[] = {"one-level fork (C from A)", "two-level fork (D from C from A)", "two-level fork, inner stream used twice", "inner stream joined from two parents"};
  for (int v = 0; v < 4; v++) {
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) { _exit(variant(v)); }
    int st = 0; waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("%-44s: killed by signal %d\n", names[v], WTERMSIG(st));
    else printf("%-44s: exit %d (%s)\n", names[v], WEXITSTATUS(st), WEXITSTATUS(st) == 0 ? "captured, instantiated, replayed" : "error");
  }
  return 0;
}

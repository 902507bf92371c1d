/* Diagnostic preload (not part of the product): on SIGABRT / SIGSEGV / SIGBUS write the NATIVE call chain of the faulting
 * thread to the file named by FDN_ABORT_TRACE, then let the signal take its course.  pytest's fd-level capture swallows what
 * the runtime prints before it aborts; this does not go through fd 2.
 *   gcc -O1 -g -shared -fPIC -o /tmp/abort_trace.so tools/abort_trace.c
 *   FDN_ABORT_TRACE=gpurun_out/abort.txt LD_PRELOAD=/tmp/abort_trace.so python -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_fd = -1;

static void on_signal(int sig)
{
    void* frames[96];
    const char* what = sig == SIGABRT ? "SIGABRT\n" : sig == SIGSEGV ? "SIGSEGV\n" : "SIGBUS\n";
    if (g_fd >= 0) {
        (void)!write(g_fd, what, strlen(what));
        const int n = backtrace(frames, 96);
        backtrace_symbols_fd(frames, n, g_fd);
        fsync(g_fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    const char* path = getenv("FDN_ABORT_TRACE");
    if (!path || !*path) return;
    g_fd = open(path, O_CREAT | O_WRONLY | O_APPEND, 0644);
    void* warm[4];
    (void)backtrace(warm, 4);      /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
}

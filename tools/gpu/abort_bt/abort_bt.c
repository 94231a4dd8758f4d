// LD_PRELOAD shim for the hunt of the serial suite's abort (profiles/r06_serial_suite_abort.txt): on SIGABRT, the C backtrace of the raising thread
// (module + offset per frame) is appended to $ABORT_BT_FILE before the signal takes its course.  Python's faulthandler, installed later, dumps the
// Python stacks and then re-raises into the handler that was there before it: this one.
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_abort(int sig) {
    void *bt[96];
    const int n = backtrace(bt, 96);
    const char *path = getenv("ABORT_BT_FILE");
    const int fd = open(path ? path : "/tmp/abort_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        const char *hdr = "---- SIGABRT, backtrace of the raising thread ----\n";
        (void)!write(fd, hdr, strlen(hdr));
        backtrace_symbols_fd(bt, n, fd);
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    void *warm[4];
    (void)backtrace(warm, 4);  // (loads libgcc now, not inside the handler)
    signal(SIGABRT, on_abort);
}

// SIGSEGV / SIGABRT handler on an alternate stack: native backtrace to stderr (debugging aid for crashes inside the HIP runtime)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static char altstack[1 << 16];
static void handler(int sig) {
    void* frames[48];
    int n = backtrace(frames, 48);
    char msg[64];
    int k = snprintf(msg, sizeof msg, "\n==== native backtrace (signal %d) ====\n", sig);
    if (write(2, msg, k)) {}
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
void segv_bt_install(void) {
    stack_t ss; ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler; sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGABRT, &sa, 0); sigaction(SIGBUS, &sa, 0);
}

// Diagnostic only: a SIGSEGV handler that says WHERE a fault inside a profiled run happened — module + offset of every
// frame (backtrace_symbols_fd), the faulting address, and the lines of /proc/self/maps around that address — and then ends the process.
//   gcc -O1 -g -shared -fPIC -o segv_probe.so segv_probe.c        loaded with ctypes by segv_probe.py, which then runs bench.py's main()
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

static int g_fd = 2;

static void put(const char* s) { (void)!write(g_fd, s, strlen(s)); }

static void on_segv(int sig, siginfo_t* si, void* uc_) {
    char line[256];
    ucontext_t* uc = (ucontext_t*)uc_;
    snprintf(line, sizeof line, "\n== segv_probe: signal %d, fault address %p, pc %p\n", sig, si->si_addr, (void*)uc->uc_mcontext.gregs[REG_RIP]);
    put(line);
    snprintf(line, sizeof line, "   rdi %p rsi %p rdx %p rcx %p (memcpy-family: rdi = dst, rsi = src, rdx = n)\n", (void*)uc->uc_mcontext.gregs[REG_RDI],
             (void*)uc->uc_mcontext.gregs[REG_RSI], (void*)uc->uc_mcontext.gregs[REG_RDX], (void*)uc->uc_mcontext.gregs[REG_RCX]);
    put(line);
    void* frames[64];
    const int n = backtrace(frames, 64);
    put("== frames (module(+offset)):\n");
    backtrace_symbols_fd(frames, n, g_fd);
    put("== /proc/self/maps (whole file):\n");
    const int m = open("/proc/self/maps", O_RDONLY);
    if (m >= 0) {
        char buf[4096];
        ssize_t k;
        while ((k = read(m, buf, sizeof buf)) > 0) (void)!write(g_fd, buf, (size_t)k);
        close(m);
    }
    _exit(139);
}

int segv_probe_install(const char* path) {
    if (path) {
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd >= 0) g_fd = fd;
    }
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, NULL);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    void* warm[4];
    backtrace(warm, 4);  // (loads libgcc now, not inside the handler)
    return sigaction(SIGSEGV, &sa, NULL);
}

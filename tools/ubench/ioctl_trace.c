// LD_PRELOAD interposer: logs slow ioctl()s (> 1 ms) with their request number, and every munmap / madvise / brk-sized
// change above 1 MB, with a monotonic time stamp.  Used to find which driver call stalls for 20-30 ms
// (tools/exp_stall.py).   gcc -O2 -shared -fPIC -o ioctl_trace.so ioctl_trace.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <stddef.h>
#include <time.h>
#include <sys/mman.h>

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static long n_ioctl = 0;
static double t_ioctl = 0;
static long n_by_nr[256];
static double t_by_nr[256];

// count and total time of the ioctls since the previous report, by request number
void ioctl_trace_report(const char * label) {
    fprintf(stderr, "[ioctl_trace] %s: %ld ioctls, %.3f ms:", label, n_ioctl, t_ioctl);
    for (int i = 0; i < 256; ++i) {
        if (n_by_nr[i]) fprintf(stderr, "  0x%02x x%ld %.3f ms", i, n_by_nr[i], t_by_nr[i]);
        n_by_nr[i] = 0;
        t_by_nr[i] = 0;
    }
    fprintf(stderr, "\n");
    n_ioctl = 0;
    t_ioctl = 0;
}

int ioctl(int fd, unsigned long req, ...) {
    static int (*real)(int, unsigned long, void *) = 0;
    if (!real) real = (int (*)(int, unsigned long, void *))dlsym(RTLD_NEXT, "ioctl");
    va_list ap;
    va_start(ap, req);
    void * arg = va_arg(ap, void *);
    va_end(ap);
    const double t0 = now_ms();
    int rc = real(fd, req, arg);
    const double dt = now_ms() - t0;
    n_ioctl += 1;
    t_ioctl += dt;
    n_by_nr[req & 0xff] += 1;
    t_by_nr[req & 0xff] += dt;
    if (dt > 1.0) {
        fprintf(stderr, "[ioctl_trace] %12.2f ms  ioctl type '%c' nr 0x%02lx size %lu  took %8.2f ms\n", t0,
                (char)((req >> 8) & 0xff), req & 0xff, (req >> 16) & 0x3fff, dt);
    }
    return rc;
}

int munmap(void * addr, size_t len) {
    static int (*real)(void *, size_t) = 0;
    if (!real) real = (int (*)(void *, size_t))dlsym(RTLD_NEXT, "munmap");
    const double t0 = now_ms();
    int rc = real(addr, len);
    if (len >= (1u << 20)) {
        fprintf(stderr, "[ioctl_trace] %12.2f ms  munmap %p %zu KB took %.2f ms\n", t0, addr, len >> 10, now_ms() - t0);
    }
    return rc;
}

int madvise(void * addr, size_t len, int advice) {
    static int (*real)(void *, size_t, int) = 0;
    if (!real) real = (int (*)(void *, size_t, int))dlsym(RTLD_NEXT, "madvise");
    const double t0 = now_ms();
    int rc = real(addr, len, advice);
    if (len >= (1u << 20)) {
        fprintf(stderr, "[ioctl_trace] %12.2f ms  madvise %p %zu KB advice %d took %.2f ms\n", t0, addr, len >> 10,
                advice, now_ms() - t0);
    }
    return rc;
}

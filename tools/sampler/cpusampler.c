/* cpusampler.c -- development tool: a CPU-time sampler for the HOST side of a call (there is no perf on the GPU boxes).
 * Every thread of the process gets its own POSIX timer on its own CPU clock (SIGEV_THREAD_ID), so a sample is taken per
 * `period` of CPU time of that thread, on that thread; a watcher thread picks up threads as they appear.  The handler stores the
 * backtrace.  cpusampler_start(hz), cpusampler_stop(path) write raw frames + /proc/self/maps; tools/sampler/resolve.py names them.
 * build: gcc -O2 -g -shared -fPIC -o libcpusampler.so cpusampler.c -lpthread */
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define MAX_FRAMES 14
#define MAX_SAMPLES (1 << 20)
#define MAX_THREADS 1024
static void *(*g_frames)[MAX_FRAMES];
static int *g_depth, *g_tid;
static volatile long g_n;
static volatile int g_on;
static int g_hz = 250;
static pid_t g_seen[MAX_THREADS];
static timer_t g_timer[MAX_THREADS];
static int g_n_threads;
static pthread_t g_watcher;

static void on_prof(int sig, siginfo_t *si, void *uc) {
    (void) sig; (void) si; (void) uc;
    if (!g_on) return;
    const long i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    if (i >= MAX_SAMPLES) return;
    g_tid[i] = (int) syscall(SYS_gettid);
    g_depth[i] = backtrace(g_frames[i], MAX_FRAMES);
}

static void add_thread(pid_t tid) {
    for (int i = 0; i < g_n_threads; i++) if (g_seen[i] == tid) return;
    if (g_n_threads >= MAX_THREADS) return;
    struct sigevent sev;
    memset(&sev, 0, sizeof sev);
    sev.sigev_notify = SIGEV_THREAD_ID;
    sev.sigev_signo = SIGPROF;
    sev._sigev_un._tid = tid;
    const clockid_t clk = (clockid_t) ((~(unsigned) tid << 3) | 6); /* the thread's CPU clock (CPUCLOCK_SCHED | per-thread) */
    timer_t t;
    if (timer_create(clk, &sev, &t) != 0) return;
    struct itimerspec its;
    its.it_interval.tv_sec = 0; its.it_interval.tv_nsec = 1000000000L / g_hz;
    its.it_value = its.it_interval;
    timer_settime(t, 0, &its, NULL);
    g_seen[g_n_threads] = tid; g_timer[g_n_threads] = t; g_n_threads++;
}

static void *watch(void *arg) {
    (void) arg;
    const pid_t self = (pid_t) syscall(SYS_gettid);
    while (g_on) {
        DIR *d = opendir("/proc/self/task");
        if (d) {
            struct dirent *e;
            while ((e = readdir(d)) != NULL) { const pid_t tid = (pid_t) atoi(e->d_name); if (tid > 0 && tid != self) add_thread(tid); }
            closedir(d);
        }
        usleep(3000);
    }
    return NULL;
}

int cpusampler_start(int hz) {
    if (!g_frames) {
        g_frames = calloc(MAX_SAMPLES, sizeof(*g_frames));
        g_depth = calloc(MAX_SAMPLES, sizeof(*g_depth));
        g_tid = calloc(MAX_SAMPLES, sizeof(*g_tid));
        void *warm[4];
        backtrace(warm, 4); /* loads libgcc outside the handler */
    }
    if (!g_frames || !g_depth || !g_tid) return -1;
    g_n = 0; g_hz = hz > 0 ? hz : 250; g_n_threads = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGPROF, &sa, NULL) != 0) return -1;
    g_on = 1;
    return pthread_create(&g_watcher, NULL, watch, NULL);
}

int cpusampler_stop(const char *path) {
    g_on = 0;
    pthread_join(g_watcher, NULL);
    for (int i = 0; i < g_n_threads; i++) timer_delete(g_timer[i]);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    long n = g_n < MAX_SAMPLES ? g_n : MAX_SAMPLES;
    fprintf(f, "samples %ld\n", n);
    for (long i = 0; i < n; i++) {
        fprintf(f, "%d", g_tid[i]);
        for (int k = 2; k < g_depth[i]; k++) fprintf(f, " %lx", (unsigned long) (uintptr_t) g_frames[i][k]); /* 0, 1: handler, trampoline */
        fputc('\n', f);
    }
    fprintf(f, "maps\n");
    FILE *m = fopen("/proc/self/maps", "r");
    if (m) { char line[1024]; while (fgets(line, sizeof line, m)) fputs(line, f); fclose(m); }
    fclose(f);
    return 0;
}

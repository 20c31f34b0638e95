// exit_threads_probe.hip -- MEASUREMENT TOOL (not part of the product): does what a GPU process costs at exit depend on how many of its threads are
// alive when it calls _exit?  (The command's exit is bimodal, 3 ms or ~0.28 s)
//   exit_threads_probe N_THREADS JOIN(0/1) GB REPS     a parent times fork -> child's last word -> waitpid
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static std::atomic<int> g_stop{0};
static void *idle(void *) { while(!g_stop.load()) usleep(2000); return nullptr; }
static int child(int n, int join, size_t gb) {
    hipStream_t st = nullptr; if(hipFree(nullptr) != hipSuccess) return 1; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for(size_t i = 0; i < gb; i++) { void *p; if(hipMalloc(&p, 1ull << 30) != hipSuccess) return 1; hipMemsetAsync(p, 1, 1ull << 30, st); }
    void *h = nullptr; hipHostMalloc(&h, 256u << 20, hipHostMallocDefault); memset(h, 1, 256u << 20);
    hipStreamSynchronize(st);
    pthread_t *th = (pthread_t *)calloc((size_t)n + 1, sizeof(pthread_t));
    for(int i = 0; i < n; i++) pthread_create(&th[i], nullptr, idle, nullptr);
    usleep(20000);
    if(join) { g_stop.store(1); for(int i = 0; i < n; i++) pthread_join(th[i], nullptr); }
    return 0;
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 0, join = argc > 2 ? atoi(argv[2]) : 0, reps = argc > 4 ? atoi(argv[4]) : 8; const size_t gb = argc > 3 ? (size_t)atol(argv[3]) : 4;
    printf("[threads %d, %s, %zu GiB of device memory] exit:", n, join ? "joined before _exit" : "alive at _exit", gb);
    for(int r = 0; r < reps; r++) {
        int pfd[2]; if(pipe(pfd)) return 1;
        pid_t c = fork();
        if(!c) { int rc = child(n, join, gb); double t = now(); if(write(pfd[1], &t, sizeof t) < 0) _exit(2); _exit(rc); }
        int st; waitpid(c, &st, 0); const double t1 = now(); double t = 0; if(read(pfd[0], &t, sizeof t) < 0) return 1;
        close(pfd[0]); close(pfd[1]);
        printf(" %.3f", t1 - t); fflush(stdout); usleep(300000);
    }
    printf("\n");
    return 0;
}

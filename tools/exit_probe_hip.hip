// exit_probe_hip.hip -- MEASUREMENT TOOL (not part of the product): what a process that used the GPU costs the kernel at exit, by what it holds.
//   exit_probe_hip MODE [GB=3]      run by a parent that times fork -> child's _exit -> waitpid
// modes: init | vram | reg | reg_unreg | reg_unreg_drop | reg_unreg_sleep_drop | reg_unreg_unmap | hostmalloc | plain | plain_drop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <thread>
#include <vector>
#include <atomic>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double rss_mb() { long vm = 0, rss = 0; FILE *f = fopen("/proc/self/statm", "r"); if(f) { if(fscanf(f, "%ld %ld", &vm, &rss) != 2) rss = 0; fclose(f); } return rss * 4096e-6; }
static void *thp_block(size_t len) {
    char *raw = (char *)mmap(nullptr, len + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if(raw == MAP_FAILED) return nullptr;
    char *base = (char *)(((uintptr_t)raw + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    if(base > raw) munmap(raw, (size_t)(base - raw));
    munmap(base + len, (size_t)(raw + len + (2u << 20) - (base + len)));
    madvise(base, len, MADV_HUGEPAGE);
    return base;
}
static int child(const char *mode, size_t gb) {
    const size_t blk = 50u << 20, nblk = gb * (1ull << 30) / blk; std::vector<void *> b;
    const bool hip = strncmp(mode, "plain", 5) != 0;
    hipStream_t st = nullptr; char *d = nullptr;
    if(hip) { if(hipFree(nullptr) != hipSuccess) return 1; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); }
    if(!strcmp(mode, "vram")) { for(size_t i = 0; i < gb; i++) { void *p; if(hipMalloc(&p, 1ull << 30) != hipSuccess) return 1; hipMemsetAsync(p, 1, 1ull << 30, st); } hipStreamSynchronize(st); }
    if(!strncmp(mode, "reg", 3) || !strncmp(mode, "plain", 5)) {
        std::atomic<size_t> nx{0}; b.resize(nblk);
        auto fill = [&]() { for(;;) { size_t i = nx.fetch_add(1); if(i >= nblk) break; b[i] = thp_block(blk); memset(b[i], 1, blk); } };
        std::vector<std::thread> th; for(int i = 0; i < 16; i++) th.emplace_back(fill); for(auto &t : th) t.join();
    }
    if(!strncmp(mode, "reg", 3)) {
        hipMalloc((void **)&d, blk);
        double t0 = now(); for(void *p : b) hipHostRegister(p, blk, hipHostRegisterDefault); double t1 = now();
        for(void *p : b) hipMemcpyAsync(d, p, blk, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        fprintf(stderr, "  [%s] registered %zu blocks in %.3fs, H2D %.1f GB/s, rss %.0f MB\n", mode, nblk, t1 - t0, nblk * (double)blk / (now() - t1) / 1e9, rss_mb());
    }
    if(!strcmp(mode, "hostmalloc")) { double t0 = now(); for(size_t i = 0; i < nblk; i++) { void *p; hipHostMalloc(&p, blk, hipHostMallocDefault); memset(p, 1, blk); } fprintf(stderr, "  [%s] %zu x 50 MB hipHostMalloc + touch in %.3fs, rss %.0f MB\n", mode, nblk, now() - t0, rss_mb()); }
    if(strstr(mode, "unreg")) { hipDeviceSynchronize(); double t0 = now(); for(void *p : b) hipHostUnregister(p); fprintf(stderr, "  [%s] unregistered in %.3fs\n", mode, now() - t0); }
    if(strstr(mode, "sleep")) usleep(100000);
    if(strstr(mode, "drop") || strstr(mode, "unmap")) {
        const bool unmap = strstr(mode, "unmap") != nullptr; double t0 = now(); std::atomic<size_t> nx{0};
        auto drop = [&]() { for(;;) { size_t i = nx.fetch_add(1); if(i >= b.size()) break; if(unmap) munmap(b[i], blk); else madvise(b[i], blk, MADV_DONTNEED); } };
        std::vector<std::thread> th; for(int i = 0; i < 16; i++) th.emplace_back(drop); for(auto &t : th) t.join();
        const double t1 = now(); usleep(20000);
        fprintf(stderr, "  [%s] %s by 16 threads in %.3fs, rss 20 ms later %.0f MB\n", mode, unmap ? "unmapped" : "dropped", t1 - t0, rss_mb());
    }
    return 0;
}
int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "init"; const size_t gb = argc > 2 ? (size_t)atol(argv[2]) : 3;
    int pfd[2]; if(pipe(pfd)) return 1;
    pid_t c = fork();
    if(!c) { int rc = child(mode, gb); double t = now(); if(write(pfd[1], &t, sizeof t) < 0) _exit(2); _exit(rc); }
    int st; waitpid(c, &st, 0); const double t1 = now(); double t = 0; if(read(pfd[0], &t, sizeof t) < 0) return 1;
    printf("%-22s %zu GB: exit -> reaped %.3f s (rc %d)\n", mode, gb, t1 - t, WEXITSTATUS(st));
    return 0;
}

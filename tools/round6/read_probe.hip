// round 6: how fast T threads get a large file's bytes out of the page cache: pread into their own buffers (ordinary / pinned) against memcpy out of a fresh mapping
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <atomic>
#include <thread>
#include <vector>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *fn = argv[1]; const int T = argc > 2 ? atoi(argv[2]) : 8; const size_t P = 96u << 20;
    int fd = open(fn, O_RDONLY); struct stat st; fstat(fd, &st); const size_t len = (size_t)st.st_size;
    hipFree(nullptr);
    for(int mode = 0; mode < 5; mode++) {      // 0 pread -> malloc, 1 pread -> hipHostMalloc, 2 memcpy from a fresh mapping -> malloc, 3 the same -> hipHostMalloc, 4: 26-byte preads every 19 KB (a header walk), one thread
        std::vector<void *> buf((size_t)T);
        for(int t = 0; t < T; t++) { if(mode == 1 || mode == 3) hipHostMalloc(&buf[(size_t)t], P, hipHostMallocDefault); else { buf[(size_t)t] = malloc(P); } memset(buf[(size_t)t], 1, P); }
        const uint8_t *map = nullptr; if(mode == 2 || mode == 3) map = (const uint8_t *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        std::atomic<size_t> next{0}; const double t0 = now();
        if(mode == 4) { uint8_t h[32]; size_t n = 0; for(size_t off = 0; off + 32 < len; off += 19000) { if(pread(fd, h, 26, (off_t)off) != 26) break; n++; } printf("header walk by pread: %zu reads in %.3f s (%.2f us each)\n", n, now() - t0, (now() - t0) / (double)n * 1e6); }
        else {
            std::vector<std::thread> th;
            for(int t = 0; t < T; t++) th.emplace_back([&, t]() { for(;;) { const size_t off = next.fetch_add(P); if(off >= len) break; const size_t n = len - off < P ? len - off : P;
                if(mode < 2) { size_t g = 0; while(g < n) { ssize_t r = pread(fd, (char *)buf[(size_t)t] + g, n - g, (off_t)(off + g)); if(r <= 0) break; g += (size_t)r; } } else memcpy(buf[(size_t)t], map + off, n); } });
            for(auto &x : th) x.join();
            const double dt = now() - t0;
            printf("%s, %d threads: %.3f s, %.1f GB/s\n", mode == 0 ? "pread -> malloc" : mode == 1 ? "pread -> hipHostMalloc" : mode == 2 ? "memcpy from a fresh mapping -> malloc" : "memcpy from a fresh mapping -> hipHostMalloc", T, dt, len / dt / 1e9);
        }
        if(map) { const double tu = now(); munmap((void *)map, len); printf("   munmap %.3f s\n", now() - tu); }
        for(int t = 0; t < T; t++) { if(mode == 1 || mode == 3) hipHostFree(buf[(size_t)t]); else free(buf[(size_t)t]); }
    }
    { const uint8_t *map = (const uint8_t *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0); const double t0 = now(); size_t n = 0; volatile uint8_t s = 0; for(size_t off = 0; off + 32 < len; off += 19000) { s ^= map[off]; s ^= map[off + 18]; n++; } printf("header walk in a fresh mapping: %zu members in %.3f s (%.2f us each)\n", n, now() - t0, (now() - t0) / (double)n * 1e6); munmap((void *)map, len); }
    return 0;
}

// pin_probe.hip -- MEASUREMENT TOOL (not part of the product): what the host->device feed costs on this box.
//   pin_probe [MB=256] [file]
// hipHostMalloc / hipHostRegister (huge-page backed) / pageable: allocation or registration time, time the submitting thread
// spends inside hipMemcpyAsync for a 64 MB piece, and the achieved H2D rate; optionally registering a read-only file mapping.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

static int h2d(const char *name, const uint8_t *src, size_t bytes, uint8_t *dst, hipStream_t st) {
    const size_t piece = 64u << 20; double t_sub = 0; int n = 0;
    CK(hipStreamSynchronize(st));
    double t0 = now();
    for(size_t o = 0; o + piece <= bytes; o += piece, n++) { double a = now(); CK(hipMemcpyAsync(dst + o, src + o, piece, hipMemcpyHostToDevice, st)); t_sub += now() - a; }
    CK(hipStreamSynchronize(st));
    double dt = now() - t0;
    printf("\"%s\": {\"GBps\": %.1f, \"submit_ms_per_64MB\": %.3f, \"total_ms\": %.2f}, ", name, n * (double)piece / dt / 1e9, t_sub / n * 1e3, dt * 1e3);
    return 0;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 256) << 20;
    printf("{\"MB\": %zu, ", bytes >> 20);
    double t0 = now(); CK(hipFree(nullptr)); printf("\"hip_init_s\": %.3f, ", now() - t0);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint8_t *d; CK(hipMalloc((void **)&d, bytes));
    // (a) hipHostMalloc
    uint8_t *p; t0 = now(); CK(hipHostMalloc((void **)&p, bytes, hipHostMallocDefault)); double ta = now() - t0;
    t0 = now(); memset(p, 1, bytes); double tt = now() - t0;
    printf("\"hipHostMalloc_s\": %.4f, \"first_touch_s\": %.4f, ", ta, tt);
    if(h2d("h2d_hostmalloc", p, bytes, d, st)) return 1; if(h2d("h2d_hostmalloc_again", p, bytes, d, st)) return 1;
    t0 = now(); CK(hipHostFree(p)); printf("\"hipHostFree_s\": %.4f, ", now() - t0);
    // (b) huge-page backed malloc, then register
    void *q = nullptr; if(posix_memalign(&q, 2u << 20, bytes)) return 1; madvise(q, bytes, MADV_HUGEPAGE);
    t0 = now(); memset(q, 2, bytes); printf("\"thp_first_touch_s\": %.4f, ", now() - t0);
    if(h2d("h2d_pageable_thp", (uint8_t *)q, bytes, d, st)) return 1;
    t0 = now(); hipError_t e = hipHostRegister(q, bytes, hipHostRegisterDefault); double tr = now() - t0;
    printf("\"hipHostRegister_s\": %.4f, \"hipHostRegister_ok\": %d, ", tr, e == hipSuccess);
    if(e == hipSuccess) { if(h2d("h2d_registered", (uint8_t *)q, bytes, d, st)) return 1; t0 = now(); CK(hipHostUnregister(q)); printf("\"hipHostUnregister_s\": %.4f, ", now() - t0); }
    // registering in 64 MB slices (as a pool of slabs would)
    t0 = now(); int okn = 0; for(size_t o = 0; o + (64u << 20) <= bytes; o += 64u << 20) okn += hipHostRegister((uint8_t *)q + o, 64u << 20, hipHostRegisterDefault) == hipSuccess;
    printf("\"register_64MB_slices_s\": %.4f, \"slices_ok\": %d, ", now() - t0, okn);
    for(size_t o = 0; o + (64u << 20) <= bytes; o += 64u << 20) (void)hipHostUnregister((uint8_t *)q + o);
    // (c) small-page pageable
    void *r = malloc(bytes); memset(r, 3, bytes); madvise(r, bytes, MADV_NOHUGEPAGE);
    if(h2d("h2d_pageable_4k", (uint8_t *)r, bytes, d, st)) return 1;
    // (d) a read-only file mapping
    if(argc > 2) {
        int fd = open(argv[2], O_RDONLY); struct stat sb; if(fd >= 0 && fstat(fd, &sb) == 0) {
            size_t fl = (size_t)sb.st_size; if(fl > bytes) fl = bytes; fl &= ~(size_t)((64u << 20) - 1);
            void *m = mmap(nullptr, fl, PROT_READ, MAP_PRIVATE, fd, 0);
            if(m != MAP_FAILED && fl) {
                volatile uint8_t acc = 0; for(size_t o = 0; o < fl; o += 4096) acc += ((uint8_t *)m)[o];
                if(h2d("h2d_file_mapping_pageable", (uint8_t *)m, fl, d, st)) return 1;
                t0 = now(); e = hipHostRegister(m, fl, hipHostRegisterReadOnly); printf("\"register_file_mapping_readonly_s\": %.4f, \"register_file_mapping_ok\": %d, ", now() - t0, e == hipSuccess);
                if(e == hipSuccess) { if(h2d("h2d_file_mapping_registered", (uint8_t *)m, fl, d, st)) return 1; (void)hipHostUnregister(m); } else (void)hipGetLastError();
                // pread into pinned memory: what a staging thread would do
                uint8_t *pp; CK(hipHostMalloc((void **)&pp, 64u << 20, hipHostMallocDefault));
                t0 = now(); size_t got = 0; for(size_t o = 0; o + (64u << 20) <= fl; o += 64u << 20) got += (size_t)pread(fd, pp, 64u << 20, (off_t)o);
                double tp = now() - t0; printf("\"pread_to_pinned_GBps_1thread\": %.2f, ", got / tp / 1e9);
                t0 = now(); for(size_t o = 0; o + (64u << 20) <= fl; o += 64u << 20) memcpy(pp, (uint8_t *)m + o, 64u << 20);
                tp = now() - t0; printf("\"memcpy_mapping_to_pinned_GBps_1thread\": %.2f, ", fl / tp / 1e9);
            }
        }
    }
    printf("\"done\": 1}\n");
    return 0;
}

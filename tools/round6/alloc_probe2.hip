// round 6: is fresh device memory cleared when it is allocated (in the background) or when it is first used?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double touch(void *p, size_t n) { double t = now(); hipMemset(p, 1, n); hipDeviceSynchronize(); return (now() - t) * 1e3; }
int main() {
    hipFree(nullptr);
    { void *w; hipMalloc(&w, 1 << 20); touch(w, 1 << 20); }
    void *a, *b, *c; const size_t G = 1ull << 30;
    hipMalloc(&a, G); printf("alloc, touch at once: %.2f ms\n", touch(a, G));
    hipMalloc(&b, G); std::this_thread::sleep_for(std::chrono::milliseconds(100)); printf("alloc, 100 ms later: %.2f ms\n", touch(b, G));
    hipMalloc(&c, G); printf("first quarter %.2f ms", touch(c, G / 4)); printf(", second quarter %.2f ms", touch((char *)c + G / 4, G / 4)); printf(", all again %.2f ms\n", touch(c, G));
    hipFree(a); void *d; hipMalloc(&d, G); printf("freed and allocated again (%s): %.2f ms\n", d == a ? "same address" : "other address", touch(d, G));
    // eight at once, touched one after the other
    void *p[8]; double t = now(); for(int i = 0; i < 8; i++) hipMalloc(&p[i], G); printf("8 x 1 GiB allocated in %.2f ms; touched:", (now() - t) * 1e3); for(int i = 0; i < 8; i++) printf(" %.1f", touch(p[i], G)); printf(" ms\n");
    // a kernel-side first touch vs a copy-engine one
    void *h; hipHostMalloc(&h, 256u << 20, hipHostMallocDefault); void *e; hipMalloc(&e, G); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    t = now(); hipMemcpyAsync(e, h, 256u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("H2D 256 MiB into fresh memory %.2f ms", (now() - t) * 1e3);
    t = now(); hipMemcpyAsync(e, h, 256u << 20, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf(", again %.2f ms\n", (now() - t) * 1e3);
    size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot); printf("free %.1f GiB of %.1f\n", fr / 1073741824.0, tot / 1073741824.0);
    return 0;
}

// round 6: what device / pinned allocations cost the calling thread on this box, alone and with a second thread allocating at the same time
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(nullptr);
    size_t sizes[] = {64ull << 20, 256ull << 20, 1ull << 30, 4ull << 30, 1ull << 30, 1ull << 30};
    std::vector<void *> keep;
    for(size_t sz : sizes) { void *p = nullptr; double t = now(); hipError_t e = hipMalloc(&p, sz); double d = now() - t; printf("hipMalloc %5zu MiB: %.2f ms (%s)\n", sz >> 20, d * 1e3, hipGetErrorName(e)); keep.push_back(p); }
    { void *p = nullptr; double t = now(); hipMalloc(&p, 1ull << 30); double d = now() - t; t = now(); hipMemset(p, 0, 1ull << 30); hipDeviceSynchronize(); printf("hipMalloc 1 GiB %.2f ms, first touch (memset) %.2f ms", d * 1e3, (now() - t) * 1e3); t = now(); hipMemset(p, 0, 1ull << 30); hipDeviceSynchronize(); printf(", second memset %.2f ms\n", (now() - t) * 1e3); t = now(); hipFree(p); printf("hipFree 1 GiB %.2f ms\n", (now() - t) * 1e3); }
    for(int i = 0; i < 3; i++) { void *p = nullptr; double t = now(); hipHostMalloc(&p, 32u << 20, hipHostMallocDefault); printf("hipHostMalloc 32 MiB: %.2f ms\n", (now() - t) * 1e3); }
    // two threads at once
    auto worker = [](int id, size_t sz, int n) { hipSetDevice(0); for(int i = 0; i < n; i++) { void *p = nullptr; double t = now(); hipMalloc(&p, sz); printf("  thread %d hipMalloc %zu MiB: %.2f ms\n", id, sz >> 20, (now() - t) * 1e3); } };
    std::thread a(worker, 0, 1ull << 30, 3), b(worker, 1, 400ull << 20, 3), c(worker, 2, 400ull << 20, 3);
    // and a thread that issues async copies meanwhile: how long does a call block?
    void *h = nullptr, *d = nullptr; hipHostMalloc(&h, 64u << 20, hipHostMallocDefault); hipMalloc(&d, 64u << 20); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double worst = 0; for(int i = 0; i < 200; i++) { double t = now(); hipMemcpyAsync(d, h, 1u << 20, hipMemcpyHostToDevice, s); double dd = now() - t; if(dd > worst) worst = dd; std::this_thread::sleep_for(std::chrono::microseconds(200)); }
    a.join(); b.join(); c.join();
    printf("worst hipMemcpyAsync call while three threads allocate: %.2f ms\n", worst * 1e3);
    // virtual memory management: reserve once, map as needed?
    { size_t gran = 0; hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended); printf("VMM granularity %zu (%s)\n", gran, hipGetErrorName(e));
      void *va = nullptr; double t = now(); e = hipMemAddressReserve(&va, 8ull << 30, 0, nullptr, 0); printf("reserve 8 GiB VA: %.2f ms (%s)\n", (now() - t) * 1e3, hipGetErrorName(e));
      if(e == hipSuccess) for(int i = 0; i < 3; i++) { hipMemGenericAllocationHandle_t hd; t = now(); e = hipMemCreate(&hd, 1ull << 30, &prop, 0); double t1 = now(); hipError_t e2 = hipMemMap((char *)va + ((size_t)i << 30), 1ull << 30, 0, hd, 0); hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite; hipError_t e3 = hipMemSetAccess((char *)va + ((size_t)i << 30), 1ull << 30, &ad, 1); printf("VMM 1 GiB: create %.2f ms map+access %.2f ms (%s %s %s)\n", (t1 - t) * 1e3, (now() - t1) * 1e3, hipGetErrorName(e), hipGetErrorName(e2), hipGetErrorName(e3)); } }
    return 0;
}

// mdk_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of k_pileup.
// /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports exactly half of the bytes of a wide coalesced
// stream (16 B per lane); "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern".  Each kernel below touches a KNOWN set of 64-byte lines of a 1 GiB buffer (four times the Infinity
// Cache) exactly once; the tool prints, per kernel, the bytes of the distinct 64-byte lines (and 128-byte line pairs) it
// touched.  Run under `rocprofv3 --pmc FETCH_SIZE` (and once more with WRITE_SIZE); tools/summarize_prof.py divides.
//   calib_stream16     16 B per lane, coalesced (the guide's reference pattern; expected factor 2)
//   calib_byte_line    one byte from every 64-byte line, lanes on consecutive lines
//   calib_byte_sparse  one byte from every second 64-byte line (does the fabric fetch 128-byte pairs?)
//   calib_gather_pair  k_pileup's pattern: per lane a sequence byte and a quality byte ~80 B apart inside 228-byte read
//                      payloads, reads visited in order, one random base per read
//   calib_write16      16 B per lane coalesced stores (site records are 16-byte stores)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

__global__ void calib_stream16(const uint4 *p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if(acc == 0x12345678u) *sink = acc;
}
__global__ void calib_byte_line(const uint8_t *p, size_t nlines, int stride, uint32_t *sink) {
    uint32_t acc = 0;
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * (size_t)stride + (i & 63)];
    if(acc == 0x12345678u) *sink = acc;
}
__global__ void calib_byte_sparse(const uint8_t *p, size_t nlines, int stride, uint32_t *sink) {
    uint32_t acc = 0;
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * (size_t)stride + (i & 63)];
    if(acc == 0x12345678u) *sink = acc;
}
__global__ void calib_gather_pair(const uint8_t *p, const uint8_t *q, size_t nreads, uint32_t *sink) {
    uint32_t acc = 0;
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nreads; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t *r = p + i * 228; const int b = q[i];           // base index 0..149
        acc += r[b >> 1]; acc += r[76 + b];
    }
    if(acc == 0x12345678u) *sink = acc;
}
__global__ void calib_write16(uint4 *p, size_t n16) {
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const size_t N = (size_t)1 << 30;
    uint8_t *d = nullptr, *dq = nullptr; uint32_t *sink = nullptr;
    CK(hipMalloc((void **)&d, N)); CK(hipMemset(d, 1, N)); CK(hipMalloc((void **)&sink, 4));
    const int grid = 256 * 16, block = 256;
    // gather pattern: which base of each read
    const size_t nreads = N / 228;
    std::vector<uint8_t> hq(nreads); uint64_t x = 88172645463325252ull;
    std::vector<uint8_t> line(N / 64, 0);
    for(size_t i = 0; i < nreads; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hq[i] = (uint8_t)(x % 150); size_t a = i * 228 + (hq[i] >> 1), b = i * 228 + 76 + hq[i]; line[a / 64] = 1; line[b / 64] = 1; }
    size_t l64 = 0, l128 = 0;
    for(size_t i = 0; i < line.size(); i++) l64 += line[i];
    for(size_t i = 0; i + 1 < line.size(); i += 2) l128 += (line[i] | line[i + 1]);
    CK(hipMalloc((void **)&dq, nreads)); CK(hipMemcpy(dq, hq.data(), nreads, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    for(int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_stream16, dim3(grid), dim3(block), 0, 0, (const uint4 *)d, N / 16, sink);
        hipLaunchKernelGGL(calib_byte_line, dim3(grid), dim3(block), 0, 0, d, N / 64, 64, sink);
        hipLaunchKernelGGL(calib_byte_sparse, dim3(grid), dim3(block), 0, 0, d, N / 128, 128, sink);
        hipLaunchKernelGGL(calib_gather_pair, dim3(grid), dim3(block), 0, 0, d, dq, nreads, sink);
        hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(block), 0, 0, (uint4 *)d, N / 16);
        CK(hipDeviceSynchronize());
    }
    printf("{\"buffer_bytes\": %zu, \"expected\": {\"calib_stream16\": {\"bytes64\": %zu, \"bytes128\": %zu}, \"calib_byte_line\": {\"bytes64\": %zu, \"bytes128\": %zu}, "
           "\"calib_byte_sparse\": {\"bytes64\": %zu, \"bytes128\": %zu}, \"calib_gather_pair\": {\"bytes64\": %zu, \"bytes128\": %zu, \"aux_bytes\": %zu}, \"calib_write16\": {\"bytes64\": %zu, \"bytes128\": %zu}}}\n",
           N, N, N, N, N, N / 2, N, l64 * 64, l128 * 128, nreads, N, N);
    return 0;
}

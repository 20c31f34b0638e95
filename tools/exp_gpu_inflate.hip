// exp_gpu_inflate.hip -- EXPERIMENT (not part of the product): DEFLATE decoding of BGZF members on the GPU, one lane per member.
// build: hipcc --offload-arch=gfx950 -O3 -o exp_gpu_inflate tools/exp_gpu_inflate.hip -lz ; run: exp_gpu_inflate file.bam 64
// Round 2 result on MI355X (32 Mb synthetic BAM: 27,774 members, 519 MB compressed, 1.7 GB inflated; output compared byte for
// byte with zlib: identical): global decode tables + byte refills 108 ms (5.0 GB/s compressed); fast tables in LDS, 32-bit
// refills, 8-byte match copies 69 ms (7.9 GB/s); refill prefetch + literal batching: no change; a uniform micro-step loop (every
// iteration a lane either copies one chunk of a pending match or decodes one symbol; not kept in this file): 114 ms -- slower.  The whole file is only 434
// waves (1.7 per CU) and a wave's 64 members diverge: every step costs the longest path among its lanes, and the match copies
// are loops of dependent global loads whose s_waitcnt also waits for the lane's preceding byte stores (a match source has to come
// back through L2 after the stores that produced it; the 32 KiB window per member does not fit on chip for 64 members a wave).  About the speed of
// 64-128 host threads with libdeflate -- not enough to move the inflate to the device as it is.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>
#include <zlib.h>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

struct Member { uint64_t in_off; uint32_t in_len; uint32_t out_len; uint64_t out_off; };

// per-lane decode tables in global scratch: primary lookup tables (litlen 10 bits, dist 8 bits) with entries
// (symbol << 4 | code length); longer codes fall back to a canonical bit-by-bit walk
#define LBITS 9
#define DBITS 7
struct Tables { uint16_t lcount[16], lsym[288], dcount[16], dsym[32]; };      // slow path (codes longer than the fast tables), global scratch
struct Fast { uint16_t lit[1 << LBITS]; uint16_t dist[1 << DBITS]; };          // per lane, in LDS

struct BitReader {
    const uint8_t *p, *end; uint64_t buf; int cnt; uint32_t nextw;      // nextw: the 4 bytes at p, requested one refill ahead of their use
    __device__ void init(const uint8_t *s, uint32_t n) { p = s; end = s + n; buf = 0; cnt = 0; __builtin_memcpy(&nextw, p, 4); }
    __device__ void fill() { if(cnt <= 32) { buf |= (uint64_t)nextw << cnt; cnt += 32; p += 4; __builtin_memcpy(&nextw, p, 4); } }      // the stream has slack behind it
    __device__ uint32_t peek(int n) { return (uint32_t)(buf & ((1ull << n) - 1)); }
    __device__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ uint32_t get(int n) { fill(); uint32_t v = peek(n); drop(n); return v; }
};

__device__ int build(const uint8_t *len, int n, uint16_t *count, uint16_t *sym, uint16_t *fast, int fbits) {
    uint16_t offs[16];
    for(int i = 0; i < 16; i++) count[i] = 0;
    for(int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    int left = 1;
    for(int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if(left < 0) return -1; }
    offs[1] = 0;
    for(int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for(int i = 0; i < n; i++) if(len[i]) sym[offs[len[i]]++] = (uint16_t)i;
    // fast table: canonical codes, bit-reversed (DEFLATE packs codes MSB first into an LSB-first stream)
    for(int i = 0; i < (1 << fbits); i++) fast[i] = 0;
    int code = 0, idx = 0;
    for(int l = 1; l <= 15; l++) {
        for(int k = 0; k < count[l]; k++, idx++, code++) {
            if(l <= fbits) {
                int r = 0; for(int b = 0; b < l; b++) if(code & (1 << b)) r |= 1 << (l - 1 - b);
                for(int f = r; f < (1 << fbits); f += 1 << l) fast[f] = (uint16_t)((sym[idx] << 4) | l);
            }
        }
        code <<= 1;
    }
    return 0;
}
__device__ int decode_slow(BitReader &br, const uint16_t *count, const uint16_t *sym) {
    int code = 0, first = 0, index = 0;
    for(int l = 1; l <= 15; l++) {
        code |= (int)br.peek(1); br.drop(1);
        int c = count[l];
        if(code - c < first) return sym[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}
__device__ __forceinline__ int decode(BitReader &br, const uint16_t *fast, int fbits, const uint16_t *count, const uint16_t *sym) {
    br.fill();
    uint32_t e = fast[br.peek(fbits)];
    if(e) { br.drop(e & 15); return e >> 4; }
    return decode_slow(br, count, sym);
}

__constant__ uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t LEXT[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__constant__ uint8_t DEXT[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__constant__ uint8_t CLORD[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

__global__ __launch_bounds__(64) void k_inflate(const uint8_t *comp, const Member *tab, int n, uint8_t *out, Tables *scratch, int *status) {
    __shared__ Fast fast[64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const Member m = tab[i];
    Tables &T = scratch[i]; Fast &F = fast[threadIdx.x];
    BitReader br; br.init(comp + m.in_off, m.in_len);
    uint8_t *o = out + m.out_off; uint32_t pos = 0; const uint32_t cap = m.out_len;
    int err = 0, last;
    uint8_t lens[344];
    do {
        last = (int)br.get(1);
        const int type = (int)br.get(2);
        if(type == 0) {
            br.drop(br.cnt & 7);
            br.fill();
            uint32_t len = br.get(16), nlen = br.get(16);
            if((len ^ 0xffff) != nlen) { err = 1; break; }
            // bytes still in the bit buffer first, then straight from the stream
            while(len && br.cnt >= 8) { if(pos >= cap) { err = 2; break; } o[pos++] = (uint8_t)br.peek(8); br.drop(8); len--; }
            if(err) break;
            br.p -= br.cnt >> 3;             // whole bytes still buffered go back to the stream
            br.buf = 0; br.cnt = 0;
            while(len--) { if(pos >= cap || br.p >= br.end) { err = 2; break; } o[pos++] = *br.p++; }
            if(err) break;
            __builtin_memcpy(&br.nextw, br.p, 4);
            continue;
        }
        if(type == 3) { err = 3; break; }
        if(type == 1) {
            for(int k = 0; k < 144; k++) lens[k] = 8;
            for(int k = 144; k < 256; k++) lens[k] = 9;
            for(int k = 256; k < 280; k++) lens[k] = 7;
            for(int k = 280; k < 288; k++) lens[k] = 8;
            build(lens, 288, T.lcount, T.lsym, F.lit, LBITS);
            for(int k = 0; k < 30; k++) lens[k] = 5;
            build(lens, 30, T.dcount, T.dsym, F.dist, DBITS);
        } else {
            const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
            if(nlen > 286 || ndist > 30) { err = 4; break; }
            for(int k = 0; k < 19; k++) lens[k] = 0;
            for(int k = 0; k < ncode; k++) lens[CLORD[k]] = (uint8_t)br.get(3);
            if(build(lens, 19, T.lcount, T.lsym, F.lit, 7) < 0) { err = 5; break; }     // code-length code in the litlen slots (7-bit fast table)
            int idx = 0;
            while(idx < nlen + ndist) {
                int sym = decode(br, F.lit, 7, T.lcount, T.lsym);
                if(sym < 0) { err = 6; break; }
                if(sym < 16) lens[19 + idx++] = (uint8_t)sym;       // (kept after the 19 code-length lengths)
                else {
                    int prev = 0, rep;
                    if(sym == 16) { if(idx == 0) { err = 7; break; } prev = lens[19 + idx - 1]; rep = 3 + (int)br.get(2); }
                    else if(sym == 17) rep = 3 + (int)br.get(3);
                    else rep = 11 + (int)br.get(7);
                    if(idx + rep > nlen + ndist) { err = 8; break; }
                    while(rep--) lens[19 + idx++] = (uint8_t)prev;
                }
            }
            if(err) break;
            if(lens[19 + 256] == 0) { err = 9; break; }
            // dist lengths first (they sit after the litlen ones), then litlen (overwrites the code-length tables)
            if(build(lens + 19 + nlen, ndist, T.dcount, T.dsym, F.dist, DBITS) < 0) { /* incomplete distance codes are legal with one code */ }
            if(build(lens + 19, nlen, T.lcount, T.lsym, F.lit, LBITS) < 0) { err = 10; break; }
        }
        uint64_t acc = 0; int nacc = 0;          // literals waiting to be stored together (o[pos .. pos+nacc))
#define FLUSH() do { if(nacc == 8) { __builtin_memcpy(o + pos, &acc, 8); } else { for(int k_ = 0; k_ < nacc; k_++) o[pos + k_] = (uint8_t)(acc >> (8 * k_)); } pos += nacc; nacc = 0; acc = 0; } while(0)
        for(;;) {
            int sym = decode(br, F.lit, LBITS, T.lcount, T.lsym);
            if(sym < 0) { err = 11; break; }
            if(sym < 256) { if(pos + nacc >= cap) { err = 12; break; } acc |= (uint64_t)sym << (8 * nacc); if(++nacc == 8) FLUSH(); }
            else if(sym == 256) { FLUSH(); break; }
            else {
                FLUSH();
                sym -= 257; if(sym >= 29) { err = 13; break; }
                int len = LBASE[sym] + (int)br.get(LEXT[sym]);
                int ds = decode(br, F.dist, DBITS, T.dcount, T.dsym);
                if(ds < 0 || ds >= 30) { err = 14; break; }
                uint32_t dist = DBASE[ds] + br.get(DEXT[ds]);
                if(dist > pos || pos + len > cap) { err = 15; break; }
                if(dist >= 8) {
                    while(len >= 8) { uint64_t v; __builtin_memcpy(&v, o + pos - dist, 8); __builtin_memcpy(o + pos, &v, 8); pos += 8; len -= 8; }      // may write up to 7 bytes past the match: inside the member, or the slack
                }
                for(int k = 0; k < len; k++, pos++) o[pos] = o[pos - dist];
            }
        }
        if(err) break;
    } while(!last);
    if(!err && pos != cap) err = 16;
    if(err) atomicExch(status, err | (i << 8));
}

int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw(n); if(fread(raw.data(), 1, n, f) != n) return 1; fclose(f);
    std::vector<Member> tab; size_t o = 0, tot = 0;
    while(o + 18 <= n) { uint16_t xlen = raw[o + 10] | raw[o + 11] << 8; uint32_t bs = (raw[o + 16] | raw[o + 17] << 8) + 1; uint32_t isz; memcpy(&isz, &raw[o + bs - 4], 4);
        Member m = {o + 12 + xlen, bs - 12 - xlen - 8, isz, tot}; tab.push_back(m); tot += isz; o += bs; }
    printf("%zu members, %zu MB compressed, %zu MB inflated\n", tab.size(), n >> 20, tot >> 20);
    // reference
    std::vector<uint8_t> ref(tot + 1);
    auto t0 = std::chrono::steady_clock::now();
    for(auto &m : tab) { if(!m.out_len) continue; z_stream zs; memset(&zs, 0, sizeof zs); zs.next_in = raw.data() + m.in_off; zs.avail_in = m.in_len; zs.next_out = ref.data() + m.out_off; zs.avail_out = m.out_len; inflateInit2(&zs, -15); inflate(&zs, Z_FINISH); inflateEnd(&zs); }
    printf("zlib 1 thread: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    uint8_t *dc, *dout; Member *dt; Tables *ds; int *dst;
    CK(hipMalloc((void **)&dc, n + 64)); CK(hipMalloc((void **)&dout, tot + 64)); CK(hipMalloc((void **)&dt, tab.size() * sizeof(Member))); CK(hipMalloc((void **)&ds, tab.size() * sizeof(Tables))); CK(hipMalloc((void **)&dst, 4));
    CK(hipMemcpy(dc, raw.data(), n, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tab.data(), tab.size() * sizeof(Member), hipMemcpyHostToDevice)); CK(hipMemset(dst, 0, 4));
    int block = argc > 2 ? atoi(argv[2]) : 64;
    for(int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_inflate, dim3((tab.size() + block - 1) / block), dim3(block), 0, 0, dc, dt, (int)tab.size(), dout, ds, dst);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("GPU inflate: %.2f ms  (%.2f GB/s compressed, %.2f GB/s inflated)\n", ms, n / ms / 1e6, tot / ms / 1e6);
    }
    int st; CK(hipMemcpy(&st, dst, 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> got(tot + 1); CK(hipMemcpy(got.data(), dout, tot, hipMemcpyDeviceToHost));
    size_t bad = 0; for(size_t i = 0; i < tot; i++) if(got[i] != ref[i]) { if(!bad) printf("first mismatch at %zu\n", i); bad++; }
    printf("status %d (err %d member %d), mismatching bytes %zu\n", st, st & 255, st >> 8, bad);
    return 0;
}

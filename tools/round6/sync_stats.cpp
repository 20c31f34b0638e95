// sync_stats.cpp -- EXPERIMENT (round 6): how far a DEFLATE decoder that starts at a wrong bit of a Huffman block runs before it stands on a bit
// the true decoder also stands on (from there on the two agree).  Decides the geometry of k_inflate's chains (csrc/mdk_inflate_core.h).
//   g++ -O2 -o /tmp/sync_stats tools/round6/sync_stats.cpp -Imethyldackel_amd/csrc -lz ; sync_stats file.bam [members]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mdk_inflate_core.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw(n + 64, 0); if(fread(raw.data(), 1, n, f) != n) return 2; fclose(f);
    const long maxm = argc > 2 ? atol(argv[2]) : 100;
    static InfShared S; size_t o = 0; long m = 0;
    std::vector<uint64_t> hist(4097, 0); uint64_t tries = 0, nosync = 0, symbits = 0, nsym = 0, blocks = 0;
    uint64_t rs = 88172645463325252ull; auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    while(o + 18 <= n && m < maxm) {
        const uint16_t xlen = (uint16_t)(raw[o + 10] | raw[o + 11] << 8); const uint32_t bs = (uint32_t)(raw[o + 16] | raw[o + 17] << 8) + 1;
        const uint64_t in_off = o + 12 + xlen; const uint32_t in_len = bs - 12 - xlen - 8;
        // the whole member's stream into a big "ring" (INF_IN_WORDS must cover it: compile with -DINF_SUB=8192)
        if((in_len + 8) / 4 + 4 < INF_IN_WORDS && in_len > 100) {
            memset(S.in, 0, sizeof S.in); memcpy(S.in, raw.data() + in_off, in_len);
            uint32_t bitpos = 0;
            for(;;) {
                inf_header_open(S, bitpos); if(S.err || S.h.type == 0) break;
                if(S.h.type == 1) for(uint32_t lane = 0; lane < 64; lane++) inf_header_fixed_lens(S, lane);
                else { if(inf_build_serial<inf_dist_t>(S.h.cl, 19, INF_CL_TB, S.dist, nullptr, nullptr, 2, 1)) break; inf_header_lens(S); if(S.err) break; }
                { const int nlit = (int)S.h.nlit, ndist = (int)S.h.ndist; if(inf_build_serial<inf_dist_t>(S.h.lens + nlit, ndist, INF_DIST_TB, S.dist, S.dsym, &S.dl, 1, S.h.type == 2) || inf_build_serial<inf_lit_t>(S.h.lens, nlit, INF_LIT_TB, S.lit, S.lsym, &S.ll, 0, 1)) break; }
                blocks++;
                std::vector<uint8_t> onchain(8u * in_len + 64, 0); uint32_t p = S.bitpos; bool ok = true;
                for(;;) { onchain[p] = 1; const InfSym s = inf_decode_at(S, p); if(s.kind == 2) { p += s.nbits; break; } if(s.kind > 2 || p > 8u * in_len) { ok = false; break; } p += s.nbits; symbits += s.nbits; nsym++; }
                if(!ok) break;
                const uint32_t b0 = S.bitpos, b1 = p;
                for(int t = 0; t < 400 && b1 > b0 + 5000; t++) {
                    uint32_t q = b0 + (uint32_t)(rnd() % (b1 - b0 - 4500)); if(onchain[q]) continue;
                    const uint32_t q0 = q; tries++; bool synced = false;
                    while(q < q0 + 4096) { if(onchain[q]) { synced = true; break; } const InfSym s = inf_decode_at(S, q); if(s.kind >= 2) break; q += s.nbits; }
                    if(synced) hist[q - q0]++; else nosync++;
                }
                if(S.last) break;
                bitpos = b1;
            }
        }
        o += bs; m++;
    }
    printf("blocks %llu, bits per symbol %.2f, tries %llu, never within 4096 bits (or ran into a bad code) %llu\n", (unsigned long long)blocks, (double)symbits / (double)nsym, (unsigned long long)tries, (unsigned long long)nosync);
    uint64_t cum = 0; int marks[] = {32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096};
    int mi = 0; for(int d = 0; d <= 4096; d++) { cum += hist[d]; if(d == marks[mi]) { printf("  synced within %4d bits: %.4f\n", d, (double)cum / (double)tries); mi++; } }
    return 0;
}

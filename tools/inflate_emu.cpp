// inflate_emu.cpp -- TEST INFRASTRUCTURE: the device inflate (csrc/mdk_inflate_core.h + the wavefront phases of k_inflate in
// csrc/mdk_inflate.hip) executed on the host, lane by lane, over every member of a BGZF file and compared with zlib byte for
// byte.  The decoding lane's code is the very code the kernel runs (the header compiles for both); the 64-lane phases are
// re-stated here with the same per-lane bodies and the kernel's barriers turned into loop boundaries.
//   build: g++ -O2 -o tools/_build/inflate_emu tools/inflate_emu.cpp -Imethyldackel_amd/csrc -lz
//   run:   inflate_emu file.bam [max_members]      (exit 0 = every member identical to zlib)
// Also: inflate_emu --fuzz N runs N damaged streams (see fuzz()); inflate_emu --selftest  runs deflate streams made with zlib at every level/strategy (stored, fixed, dynamic blocks,
// long matches, distance-1 runs, maximum-distance matches, empty input).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <zlib.h>
#include "mdk_inflate_core.h"
#include "mdk_crc32_core.h"

// statistics of the decode rounds (the file mode prints them): rounds, real symbols, literals / matches, near-match rounds, and -- what a round
// over a 128-bit window would have resolved (the walk carried on over the next 64 bits where nothing but the window's end stopped it)
static uint64_t g_rounds, g_syms, g_lits, g_matches, g_syms128, g_near_rounds, g_hdr_batches, g_round_cut;
// one member, as k_inflate does it; returns 0 or the error code
static int emu_member(const uint8_t *comp, uint64_t in_off, uint32_t in_len, uint8_t *out, uint32_t out_len, uint64_t *n_far, uint64_t *n_near, uint64_t *n_batches) {
    static InfShared S;
    if(out_len == 0) return 0;
    const uint64_t a0 = in_off & ~3ull; const uint32_t skip = (uint32_t)(in_off & 3ull);
    const uint32_t n_words = (uint32_t)((in_off + in_len + 3 - a0) >> 2);
    auto word = [&](uint32_t w) -> uint32_t { uint32_t v = 0; if(w < n_words) memcpy(&v, comp + a0 + 4ull * w, 4); return v; };   // (the caller pads the buffer)
    uint32_t filled = 0;
    uint32_t bitpos = 8u * skip, pos = 0, in_block = 0, last = 0, stored_left = 0;
    for(uint64_t turns = 0;; turns++) {
        if(turns > 200000) return 103;                                   // a batch must consume input or produce output: 64 KiB cannot take this long
        while(filled + 64 <= (bitpos >> 5) + INF_IN_WORDS) { for(uint32_t lane = 0; lane < 64; lane++) { const uint32_t w = filled + lane; S.in[w & (INF_IN_WORDS - 1)] = word(w); } filled += 64; }
        (*n_batches)++;
        const uint32_t beg = pos; uint32_t n_tok = 0, err = 0, fin = 0;
        if(in_block == 0) {
            inf_header_batch(S, bitpos); g_hdr_batches++;
            bitpos = S.bitpos; in_block = S.in_block; last = S.last; stored_left = S.stored_left; err = S.err;
        } else if(in_block == 2) {
            const uint32_t n = stored_left < INF_STORED_BATCH ? stored_left : INF_STORED_BATCH;
            for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t i = lane; i < n; i += 64) S.win[(pos + i) & (INF_WIN - 1)] = inf_ring_byte(S, bitpos, i);
            pos += n; bitpos += 8u * n; stored_left -= n;
            if(stored_left == 0) { in_block = 0; fin = last; }
        } else {
            // the rounds of k_inflate: the per-lane decode is the kernel's code, the cross-lane steps (walk by readlane, scan, ballots) run over arrays
            const uint32_t lim = beg + (INF_BATCH_BYTES - 258), blim = bitpos + 32u * INF_BATCH_WORDS;
            for(;;) {
                InfSym sy[64];
                for(uint32_t lane = 0; lane < 64; lane++) sy[lane] = inf_decode_at(S, bitpos + lane);
                uint32_t adv[64]; for(uint32_t lane = 0; lane < 64; lane++) adv[lane] = sy[lane].kind >= 3 ? 0x200u : sy[lane].kind == 2 ? (sy[lane].nbits | 0x100u) : sy[lane].nbits;
                uint32_t off = 0, a = 0, lastl = 0; uint64_t V = 0;
                do { lastl = off; V |= 1ull << off; a = adv[off]; off += a; } while(off < 64);
                off = lastl + (a & 0xffu);
                uint32_t stop = a >= 0x200u ? sy[lastl].kind : a >= 0x100u ? 2u : 0u;
                if(stop >= 3) V &= ~(1ull << lastl);
                uint32_t olen[64], dst[64], mpre[64]; uint32_t run = 0, nm = 0; uint64_t mball = 0, cm = 0;
                for(uint32_t lane = 0; lane < 64; lane++) {
                    const bool valid = (V >> lane) & 1ull;
                    olen[lane] = !valid ? 0u : sy[lane].kind == 0 ? 1u : sy[lane].kind == 1 ? (sy[lane].val & 0xffffu) : 0u;
                    dst[lane] = pos + run; run += olen[lane];
                    mpre[lane] = nm; if(valid && sy[lane].kind == 1) { mball |= 1ull << lane; nm++; }
                }
                for(uint32_t lane = 0; lane < 64; lane++) { const bool valid = (V >> lane) & 1ull, ism = (mball >> lane) & 1ull; if(valid && ((ism && n_tok + mpre[lane] >= INF_MAX_TOK) || dst[lane] + olen[lane] > beg + INF_BATCH_BYTES)) cm |= 1ull << lane; }
                if(cm) { const int c = __builtin_ctzll(cm); V &= (1ull << c) - 1ull; off = (uint32_t)c; stop = 1; mball &= V; }
                bool baddist = false;
                for(uint32_t lane = 0; lane < 64; lane++) if(((mball >> lane) & 1ull) && (sy[lane].val >> 16) > dst[lane]) baddist = true;
                if(baddist) { err = INF_E_DIST; break; }
                for(uint32_t lane = 0; lane < 64; lane++) {
                    const bool valid = (V >> lane) & 1ull;
                    if(valid && sy[lane].kind == 0) S.win[dst[lane] & (INF_WIN - 1)] = (uint8_t)sy[lane].val;
                    if((mball >> lane) & 1ull) { InfToken t; t.dst = dst[lane]; t.len_dist = sy[lane].val; if(n_tok + mpre[lane] >= INF_MAX_TOK) return 104; S.tok[n_tok + mpre[lane]] = t; }
                }
                n_tok += (uint32_t)__builtin_popcountll(mball);
                {   // statistics
                    const int nv = __builtin_popcountll(V); g_rounds++; g_syms += (uint64_t)nv; g_matches += (uint64_t)__builtin_popcountll(mball); g_lits += (uint64_t)(nv - __builtin_popcountll(mball)); g_syms128 += (uint64_t)nv;
                    if(stop == 0) { uint32_t o2 = off; while(o2 < 128) { const InfSym y = inf_decode_at(S, bitpos + o2); if(y.kind >= 2) break; g_syms128++; o2 += y.nbits; } } else if(stop == 1) g_round_cut++;
                }
                if(V) { const int hi = 63 - __builtin_clzll(V); pos = dst[hi] + olen[hi]; }
                bitpos += off;
                if(stop >= 3) { err = stop == 3 ? INF_E_SYMBOL : INF_E_DIST; break; }
                if(stop == 2) { in_block = 0; fin = last; break; }
                if(stop == 1 || n_tok >= INF_MAX_TOK || pos > lim || bitpos > blim) break;
                if(!V) return 105;                                        // a round without progress (cannot happen: the first symbol of a round always fits)
            }
        }
        if(!err && pos > out_len) err = INF_E_OVERRUN;
        if(!err && fin && pos != out_len) err = INF_E_SHORT;
        if(err) return (int)err;
        if((bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, in_len))) return INF_E_INPUT;
        const uint32_t end = pos;
        if(end - beg > INF_BATCH_BYTES || n_tok > INF_MAX_TOK) return 100;
        bool far[INF_MAX_TOK];
        for(uint32_t k = 0; k < INF_MAX_TOK; k++) {                      // the far phase: every token, 64 at a time on the device
            far[k] = false;
            if(k < n_tok) {
                const InfToken t = S.tok[k]; far[k] = inf_tok_far(t, beg);
                if(far[k]) {
                    const uint32_t len = t.len_dist & 0xffffu, src = t.dst - (t.len_dist >> 16);
                    if(src + len > beg) return 101;                       // a far source must lie wholly in what earlier batches wrote
                    for(uint32_t i = 0; i < len; i++) S.win[(t.dst + i) & (INF_WIN - 1)] = out[src + i];
                    (*n_far)++;
                }
            }
        }
        for(uint32_t tb = 0; tb < n_tok; tb += 64) {   // the near phase of k_inflate: the tokens in stream order, 64 at a time; rounds over those not yet copied
            uint64_t pending = 0; for(uint32_t k = 0; k < 64 && tb + k < n_tok; k++) if(!far[tb + k]) pending |= 1ull << k;
            int guard = 0;
            while(pending) {
                g_near_rounds++;
                if(++guard > 200) return 106;
                const int f = __builtin_ctzll(pending); const InfToken q = S.tok[tb + f];
                const uint32_t W = q.dst, flen = q.len_dist & 0xffffu;
                if(q.dst - (q.len_dist >> 16) + INF_WIN < end) return 102;          // a near source must still be in the window at the end of the batch
                if(flen > INF_NEAR_LANE_MAX) {
                    const uint32_t fdist = q.len_dist >> 16; uint32_t done = 0, span = fdist;
                    while(done < flen) {
                        const uint32_t n = span < flen - done ? span : flen - done;
                        uint8_t snap[INF_WIN]; memcpy(snap, S.win, INF_WIN);      // all lanes read before any lane's write is seen: the round must not depend on lane order
                        for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t i = lane; i < n; i += 64) S.win[(W + done + i) & (INF_WIN - 1)] = snap[(W - fdist + i) & (INF_WIN - 1)];
                        done += n; span <<= 1;
                    }
                    pending &= ~(1ull << f); (*n_near)++;
                    continue;
                }
                uint64_t ready = 0;
                for(uint32_t lane = 0; lane < 64 && tb + lane < n_tok; lane++) { const InfToken t = S.tok[tb + lane]; const uint32_t len = t.len_dist & 0xffffu, src = t.dst - (t.len_dist >> 16); if(((pending >> lane) & 1ull) && len <= INF_NEAR_LANE_MAX && ((int)lane == f || src + len <= W)) ready |= 1ull << lane; }
                // the ready lanes copy at the same time, byte i of every token in step i: run from the LAST lane to the first to show that the order between lanes does not matter
                for(int lane = 63; lane >= 0; lane--) if((ready >> lane) & 1ull) { const InfToken t = S.tok[tb + lane]; const uint32_t len = t.len_dist & 0xffffu, src = t.dst - (t.len_dist >> 16); for(uint32_t i = 0; i < len; i++) S.win[(t.dst + i) & (INF_WIN - 1)] = S.win[(src + i) & (INF_WIN - 1)]; (*n_near)++; }
                pending &= ~ready;
            }
        }
        for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t p = beg + lane; p < end; p += 64) out[p] = S.win[p & (INF_WIN - 1)];
        if(fin) return 0;
    }
}

static int check_stream(const std::vector<uint8_t> &raw, int level, int strategy, const char *what) {
    std::vector<uint8_t> comp(compressBound(raw.size()) + 64 + 8); z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    const int lead = 1 + (int)(raw.size() % 3);                                // an unaligned start, as in a file
    zs.next_in = (Bytef *)raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = comp.data() + lead; zs.avail_out = (uInt)(comp.size() - lead - 8);
    deflate(&zs, Z_FINISH); const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
    std::vector<uint8_t> got(raw.size() + 8, 0xEE); uint64_t a = 0, b = 0, c = 0;
    const int rc = emu_member(comp.data(), (uint64_t)lead, clen, got.data(), (uint32_t)raw.size(), &a, &b, &c);
    if(rc || (raw.size() && memcmp(got.data(), raw.data(), raw.size()))) { fprintf(stderr, "selftest FAILED: %s level %d strategy %d size %zu: rc %d\n", what, level, strategy, raw.size(), rc); return 1; }
    return 0;
}
static int selftest(void) {
    int bad = 0; uint64_t s = 88172645463325252ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for(int kind = 0; kind < 8; kind++) for(size_t size : {(size_t)0, (size_t)1, (size_t)7, (size_t)300, (size_t)5000, (size_t)65280, (size_t)65536}) {
        std::vector<uint8_t> raw(size);
        for(size_t i = 0; i < size; i++) {
            switch(kind) {
            case 0: raw[i] = (uint8_t)rnd(); break;                                         // incompressible: stored blocks
            case 1: raw[i] = 'A'; break;                                                    // distance-1 runs, length 258
            case 2: raw[i] = "ACGT"[rnd() & 3]; break;                                      // 4 symbols: short codes
            case 3: raw[i] = (uint8_t)(i < 40000 ? rnd() : raw[i - 32768 > 0 && i >= 32768 ? i - 32768 : 0]); break;     // maximum-distance matches
            case 4: raw[i] = (uint8_t)((i % 600) < 300 ? (rnd() % 200) : raw[i >= 300 ? i - 300 : 0]); break;            // many symbols: long codes + mid-range matches
            case 5: raw[i] = (uint8_t)((i / 3) % 251); break;
            case 6: raw[i] = (uint8_t)(rnd() % 3 ? 'x' : (rnd() & 255)); break;
            default: raw[i] = (uint8_t)((i % 7 == 0) ? rnd() : (i & 255)); break;
            }
        }
        for(int level : {0, 1, 4, 6, 9}) for(int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED}) bad += check_stream(raw, level, strat, "synthetic");
    }
    // a skewed alphabet of 286 used symbols with lengths up to 15: sub-tables of every depth
    { std::vector<uint8_t> raw; double p = 0.5; for(int sym = 0; sym < 256; sym++) { size_t n = (size_t)(60000 * p) + 1; for(size_t k = 0; k < n; k++) raw.push_back((uint8_t)sym); if(sym < 14) p /= 2; }
      for(size_t i = raw.size() - 1; i > 0; i--) { size_t j = rnd() % (i + 1); uint8_t t = raw[i]; raw[i] = raw[j]; raw[j] = t; }
      if(raw.size() > 65536) raw.resize(65536);
      for(int level : {1, 6, 9}) bad += check_stream(raw, level, Z_HUFFMAN_ONLY, "skewed") + check_stream(raw, level, Z_DEFAULT_STRATEGY, "skewed"); }
    printf("selftest: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}

// Damaged input: valid deflate streams with bits flipped, bytes overwritten, the stream cut short or the announced output size wrong.  The
// decoder must come back (it is bounded by its input and by the announced size), must not write a byte beyond that size, and, when it
// reports success, must have produced what zlib produces from the same damaged stream (a flip in a literal's bits is still a valid
// stream).  On the GPU the same code runs with nobody to catch a wild pointer, so this is where it is tried.
static int fuzz(long iters) {
    uint64_t s = 0x9e3779b97f4a7c15ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    long ok = 0, rejected = 0, bad = 0;
    for(long it = 0; it < iters; it++) {
        const size_t size = 1 + rnd() % (it % 7 == 0 ? 65536 : 6000); std::vector<uint8_t> raw(size);
        const int kind = (int)(rnd() % 4);
        for(size_t i = 0; i < size; i++) raw[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? "ACGT"[rnd() & 3] : kind == 2 ? (uint8_t)((i % 97) < 60 ? 'q' : rnd() % 11) : (uint8_t)(i >= 700 && (rnd() & 3) ? raw[i - 700] : rnd());
        std::vector<uint8_t> comp(compressBound(size) + 64 + 16, 0); z_stream zs; memset(&zs, 0, sizeof zs);
        const int levels[4] = {0, 1, 6, 9}, strats[3] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY};
        deflateInit2(&zs, levels[rnd() % 4], Z_DEFLATED, -15, 8, strats[rnd() % 3]);
        const int lead = (int)(rnd() % 4);
        zs.next_in = raw.data(); zs.avail_in = (uInt)size; zs.next_out = comp.data() + lead; zs.avail_out = (uInt)(comp.size() - lead - 16);
        deflate(&zs, Z_FINISH); uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
        uint32_t out_len = (uint32_t)size;
        switch(rnd() % 5) {                                                      // the damage
        case 0: for(int k = 1 + (int)(rnd() % 4); k > 0; k--) comp[lead + rnd() % clen] ^= (uint8_t)(1u << (rnd() & 7)); break;
        case 1: for(int k = 1 + (int)(rnd() % 6); k > 0; k--) comp[lead + rnd() % clen] = (uint8_t)rnd(); break;
        case 2: clen = (uint32_t)(rnd() % clen); break;                          // cut short
        case 3: out_len = (uint32_t)(rnd() & 1 ? rnd() % (size + 1) : size + 1 + rnd() % 300); if(out_len > 65536) out_len = 65536; break;      // wrong ISIZE
        default: comp[lead + rnd() % (clen < 12 ? clen : 12)] ^= (uint8_t)rnd(); break;      // the block header
        }
        std::vector<uint8_t> got((size_t)out_len + 64, 0xEE); uint64_t a = 0, b = 0, c = 0;
        const int rc = emu_member(comp.data(), (uint64_t)lead, clen, got.data(), out_len, &a, &b, &c);
        for(size_t i = out_len; i < got.size(); i++) if(got[i] != 0xEE) { fprintf(stderr, "fuzz %ld: a byte was written beyond the announced size (rc %d)\n", it, rc); bad++; break; }
        if(rc == 0 && out_len) {                                                 // accepted: then zlib accepts it too, with the same bytes
            std::vector<uint8_t> ref((size_t)out_len + 8); z_stream zi; memset(&zi, 0, sizeof zi); inflateInit2(&zi, -15);
            zi.next_in = comp.data() + lead; zi.avail_in = clen; zi.next_out = ref.data(); zi.avail_out = out_len;
            const int zr = inflate(&zi, Z_FINISH); const size_t zout = zi.total_out; inflateEnd(&zi);
            if(zr != Z_STREAM_END || zout != out_len || memcmp(ref.data(), got.data(), out_len)) { fprintf(stderr, "fuzz %ld: accepted a stream zlib does not inflate to the same %u bytes (zlib rc %d, %zu bytes)\n", it, out_len, zr, zout); bad++; }
            ok++;
        } else rejected++;
    }
    printf("fuzz: %ld damaged streams, %ld rejected, %ld accepted and equal to zlib, %ld FAILURES\n", iters, rejected, ok, bad);
    return bad ? 1 : 0;
}

// k_crc32 on the host: the per-lane bodies are the kernel's (mdk_crc32_core.h), the six shuffle levels run over an array
static uint32_t emu_crc32(const CrcConst &K, const uint8_t *d, uint32_t L) {
    if(L == 0) return 0;
    uint32_t c[64];
    for(int lane = 0; lane < 64; lane++) c[lane] = crc_lane(K.T, K.Z, d, L, lane);
    for(int l = 0; l < 6; l++) { uint32_t nx[64]; for(int lane = 0; lane < 64; lane++) { const int src = lane - (1 << l); nx[lane] = c[lane] ^ crc_mul(src >= 0 ? c[src] : 0xdeadbeefu, K.lvl[l]); } memcpy(c, nx, sizeof c); }
    return crc_finish(c[63], L, K.p8);
}
static int crctest(long n) {
    static CrcConst K; crc_make_const(K);
    std::vector<uint8_t> buf(65536 + 64 + 16); uint64_t rs = 0x9E3779B97F4A7C15ull; long bad = 0;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    for(long it = 0; it < n; it++) {
        // every length up to 2100 (first blocks, block edges), then random lengths up to 65536; the member starts at any alignment
        const uint32_t L = it <= 2100 ? (uint32_t)it : (it & 1) ? (uint32_t)(rnd() % 65537) : 65536u - (uint32_t)(rnd() % 40);
        const size_t off = 16 + (size_t)(rnd() % 16);
        const int kind = (int)(rnd() % 4);
        for(size_t i = 0; i < buf.size(); i++) buf[i] = kind == 0 ? 0 : kind == 1 ? 0xff : (uint8_t)rnd();
        const uint32_t want = (uint32_t)crc32(0L, buf.data() + off, L), got = emu_crc32(K, buf.data() + off, L);
        if(want != got) { if(bad < 5) fprintf(stderr, "crc32 of %u bytes (kind %d): zlib %08x, lanes %08x\n", L, kind, want, got); bad++; }
    }
    printf("{\"crc_cases\": %ld, \"mismatches\": %ld}\n", n, bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if(argc > 1 && !strcmp(argv[1], "--selftest")) return selftest();
    if(argc > 2 && !strcmp(argv[1], "--crc")) return crctest(atol(argv[2]));
    if(argc > 2 && !strcmp(argv[1], "--fuzz")) return fuzz(atol(argv[2]));
    if(argc < 2) { fprintf(stderr, "usage: inflate_emu file.bam [max_members] | --selftest\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"); if(!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw(n + 64, 0); if(fread(raw.data(), 1, n, f) != n) return 2; fclose(f);
    const long maxm = argc > 2 ? atol(argv[2]) : -1;
    size_t o = 0; long m = 0, bad = 0; uint64_t tot = 0, n_far = 0, n_near = 0, n_batches = 0;
    std::vector<uint8_t> ref(65536 + 64), got(65536 + 64);
    while(o + 18 <= n && (maxm < 0 || m < maxm)) {
        const uint16_t xlen = (uint16_t)(raw[o + 10] | raw[o + 11] << 8); const uint32_t bs = (uint32_t)(raw[o + 16] | raw[o + 17] << 8) + 1; uint32_t isz; memcpy(&isz, &raw[o + bs - 4], 4);
        const uint64_t in_off = o + 12 + xlen; const uint32_t in_len = bs - 12 - xlen - 8;
        if(isz) {
            z_stream zs; memset(&zs, 0, sizeof zs); zs.next_in = raw.data() + in_off; zs.avail_in = in_len; zs.next_out = ref.data(); zs.avail_out = isz;
            inflateInit2(&zs, -15); const int zr = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if(zr != Z_STREAM_END) { fprintf(stderr, "zlib failed on member %ld\n", m); return 2; }
            memset(got.data(), 0xEE, isz);
            const int rc = emu_member(raw.data(), in_off, in_len, got.data(), isz, &n_far, &n_near, &n_batches);
            if(rc || memcmp(got.data(), ref.data(), isz)) { if(bad < 5) { size_t k = 0; while(k < isz && got[k] == ref[k]) k++; fprintf(stderr, "member %ld (file offset %zu): rc %d, first difference at byte %zu of %u\n", m, o, rc, k, isz); } bad++; }
            tot += isz;
        }
        o += bs; m++;
    }
    printf("{\"members\": %ld, \"inflated_bytes\": %llu, \"mismatching_members\": %ld, \"batches\": %llu, \"far_matches\": %llu, \"near_matches\": %llu, "
           "\"header_batches\": %llu, \"decode_rounds\": %llu, \"symbols\": %llu, \"literals\": %llu, \"matches\": %llu, \"symbols_per_round\": %.2f, \"rounds_cut_by_batch_limits\": %llu, "
           "\"symbols_per_round_if_window_were_128_bits\": %.2f, \"near_rounds\": %llu, \"bytes_per_symbol\": %.2f}\n", m, (unsigned long long)tot, bad,
           (unsigned long long)n_batches, (unsigned long long)n_far, (unsigned long long)n_near, (unsigned long long)g_hdr_batches, (unsigned long long)g_rounds, (unsigned long long)g_syms, (unsigned long long)g_lits,
           (unsigned long long)g_matches, g_rounds ? (double)g_syms / (double)g_rounds : 0.0, (unsigned long long)g_round_cut, g_rounds ? (double)g_syms128 / (double)g_rounds : 0.0, (unsigned long long)g_near_rounds,
           g_syms ? (double)tot / (double)g_syms : 0.0);
    return bad ? 1 : 0;
}

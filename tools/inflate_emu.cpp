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

// statistics of the batches (the file mode prints them)
static uint64_t g_hbatches, g_passes, g_wave_steps, g_lane_steps, g_syms, g_hdr_batches, g_lanes_taken, g_lanes_active, g_obatches, g_jump_rounds, g_far_bytes, g_match_bytes, g_cut_passes, g_tok_groups, g_pass_hist[INF_MAX_PASSES + 1];
// the words of the stream from word `wbase` on into S.in (64 lanes, coalesced, on the device); words behind the stream read as zero
static void emu_stage(InfShared &S, const uint8_t *comp, uint64_t a0, uint32_t n_words, uint32_t wbase, uint32_t count) {
    for(uint32_t i = 0; i < count; i++) { const uint32_t w = wbase + i; uint32_t v = 0; if(w < n_words) memcpy(&v, comp + a0 + 4ull * w, 4); S.in[i] = v; }      // (the caller pads the buffer)
}
// one member, as k_inflate does it; returns 0 or the error code
static int emu_member(const uint8_t *comp, uint64_t in_off, uint32_t in_len, uint8_t *out, uint32_t out_len, uint64_t *n_batches) {
    static InfShared S; static uint32_t tok[INF_TOK_WORDS];
    if(out_len == 0) return 0;
    const uint64_t a0 = in_off & ~3ull; const uint32_t skip = (uint32_t)(in_off & 3ull);
    const uint32_t n_words = (uint32_t)((in_off + in_len + 3 - a0) >> 2);
    uint32_t bitpos = 8u * skip, pos = 0, in_block = 0, last = 0, stored_left = 0;
    auto flush = [&](uint32_t beg, uint32_t end) { for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t p = beg + lane; p < end; p += 64) out[p] = S.win[inf_win_at(p)]; };
    for(uint64_t turns = 0;; turns++) {
        if(turns > 200000) return 103;                                   // a batch must consume input or produce output: 64 KiB cannot take this long
        (*n_batches)++;
        const uint32_t wbase = bitpos >> 5, rel = bitpos & 31u;
        uint32_t err = 0, fin = 0;
        if(in_block == 0) {
            emu_stage(S, comp, a0, n_words, wbase, INF_HDR_WORDS);
            inf_header_open(S, rel); g_hdr_batches++;
            err = S.err; const uint32_t type = S.h.type;
            if(!err && type == 1) for(uint32_t lane = 0; lane < 64; lane++) inf_header_fixed_lens(S, lane);
            if(!err && type == 2) {
                if(inf_build_serial<inf_dist_t>(S.h.cl, 19, INF_CL_TB, S.dist, nullptr, nullptr, 2, 1)) err = INF_E_CODELEN;
                if(!err) { inf_header_lens(S); err = S.err; }
            }
            if(!err && type != 0) {
                const int nlit = (int)S.h.nlit, ndist = (int)S.h.ndist;
                if(inf_build_serial<inf_dist_t>(S.h.lens + nlit, ndist, INF_DIST_TB, S.dist, S.dsym, &S.dl, 1, type == 2)) err = INF_E_DISTTABLE;
                else if(inf_build_serial<inf_lit_t>(S.h.lens, nlit, INF_LIT_TB, S.lit, S.lsym, &S.ll, 0, 1)) err = INF_E_LITTABLE;
                in_block = 1;
            } else in_block = S.in_block;
            bitpos = 32u * wbase + S.bitpos; last = S.last; stored_left = S.stored_left;
            if(err) return (int)err;
            if((bitpos >> 5) > n_words) return INF_E_INPUT;
            continue;
        }
        if(in_block == 2) {
            emu_stage(S, comp, a0, n_words, wbase, INF_STORED_WORDS);
            const uint32_t n = stored_left < INF_STORED_BATCH ? stored_left : INF_STORED_BATCH, beg = pos;
            if(pos + n > out_len) return INF_E_OVERRUN;
            for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t i = lane; i < n; i += 64) S.win[inf_win_at(pos + i)] = inf_ring_byte(S, rel, i);
            pos += n; bitpos += 8u * n; stored_left -= n;
            if(stored_left == 0) { in_block = 0; fin = last; }
            if(fin && pos != out_len) return INF_E_SHORT;
            if((bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, in_len))) return INF_E_INPUT;
            flush(beg, pos);
            if(fin) return 0;
            continue;
        }
        // A Huffman batch: chains from guessed starts, then from the neighbours' ends, until a prefix of the lanes agrees (k_inflate; the per-lane
        // bodies are the kernel's code, the steps across lanes -- shifts, ballots, scans -- run over arrays).  sw words per lane: what is left of the
        // stream spread over the lanes, within [INF_SW_MIN, INF_SW_MAX], odd.
        const uint32_t left_words = n_words > wbase ? n_words - wbase : 1u;
        uint32_t sw = (left_words + 63u) / 64u; sw = sw < INF_SW_MIN ? INF_SW_MIN : sw > INF_SW_MAX ? INF_SW_MAX : sw; sw |= 1u;
        const uint32_t sub = 32u * sw;
        emu_stage(S, comp, a0, n_words, wbase, 64u * sw + 8u);
        g_hbatches++;
        const uint32_t stream_end = 32u * (n_words - wbase) + 64u;        // (relative) no true chain gets this far without an error
        uint32_t start[64]; InfChain c[64]; uint64_t need = 0, written = 0; uint32_t nconf = 0, passes = 0, nact = 0;
        for(uint32_t lane = 0; lane < 64; lane++) { start[lane] = rel + sub * lane; c[lane].status = INF_C_BADLIT; c[lane].n = 0; c[lane].end = start[lane]; if(lane == 0 || start[lane] < stream_end) { need |= 1ull << lane; nact++; } }
        g_lanes_active += nact;
        for(;;) {
            uint32_t mx = 0;
            for(uint32_t lane = 0; lane < 64; lane++) if((need >> lane) & 1ull) {
                c[lane] = inf_chain(S, start[lane], rel + sub * (lane + 1), passes != 0, tok + lane);
                const uint32_t steps = c[lane].n + (c[lane].status >= 2 ? 1u : 0u); if(steps > mx) mx = steps; g_lane_steps += steps;
            }
            g_wave_steps += mx; if(passes) written |= need; passes++;
            // who goes again: a lane whose left neighbour's chain ended elsewhere than where it started, or whose tokens are not written down yet (the first pass writes nothing)
            need = (written & 1ull) ? 0ull : 1ull;
            for(uint32_t lane = 63; lane >= 1; lane--) if(lane < nact && c[lane - 1].status == INF_C_OK && (start[lane] != c[lane - 1].end || !((written >> lane) & 1ull))) { start[lane] = c[lane - 1].end; need |= 1ull << lane; }
            if(passes == 1) continue;
            nconf = 1; while(nconf < nact && c[nconf - 1].status == INF_C_OK && !((need >> nconf) & 1ull)) nconf++;
            if(nconf == nact || c[nconf - 1].status != INF_C_OK || passes >= INF_MAX_PASSES) break;
        }
        g_passes += passes; g_pass_hist[passes]++;
        if(nconf < nact && c[nconf - 1].status == INF_C_OK) g_cut_passes++;
        g_lanes_taken += nconf;
        const InfChain &L = c[nconf - 1];
        if(L.status == INF_C_BADLIT) return INF_E_SYMBOL;
        if(L.status == INF_C_BADDIST) return INF_E_DIST;
        uint32_t tpre[66], total = 0;
        for(uint32_t lane = 0; lane < 66; lane++) { tpre[lane] = total; if(lane < nconf) total += c[lane].n; }
        g_syms += total;
        if(total == 0 && L.status == INF_C_OK && L.end == start[0]) return 108;           // no progress (cannot happen: a stretch is longer than a symbol)
        bitpos = 32u * wbase + L.end;
        if(L.status == INF_C_EOB) { in_block = 0; fin = last; }
        if((bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, in_len))) return INF_E_INPUT;
        // the tokens into bytes, a batch of output at a time
        uint32_t g = 0;
        while(g < total) {
            g_obatches++;
            const uint32_t beg = pos, base0 = beg & ~31u;
            for(uint32_t lane = 0; lane < 66; lane++) S.o.tpre[lane] = tpre[lane];
            for(uint32_t w = 0; w < INF_BATCH_BYTES / 32; w++) S.o.starts[w] = 0;
            bool full = false;
            while(g < total && !full) {
                g_tok_groups++;
                uint32_t t[64], len[64], at[64], run = 0, take = 0;
                for(uint32_t lane = 0; lane < 64; lane++) {
                    t[lane] = 0; len[lane] = 0;
                    if(g + lane < total) { const uint32_t col = inf_tok_column(S, g + lane); t[lane] = tok[64u * (g + lane - S.o.tpre[col]) + col]; len[lane] = inf_tok_len(t[lane]); }
                    at[lane] = pos + run; run += len[lane];
                    if(g + lane < total && at[lane] + len[lane] <= base0 + INF_BATCH_BYTES && take == lane) take = lane + 1;
                }
                if(take == 0) { if(pos == beg) return 105; full = true; break; }           // (a token is at most 258 bytes: the first of a batch always fits)
                for(uint32_t lane = 0; lane < take; lane++) if(!inf_tok_place(S, t[lane], at[lane], base0)) return INF_E_DIST;
                pos = at[take - 1] + len[take - 1]; g += take;
                if(take < 64 && g < total) full = true;
                if(pos > out_len) return INF_E_OVERRUN;
            }
            const uint32_t end = pos;
            if(end - base0 > INF_BATCH_BYTES) return 100;
            // matches: sources, pointer jumping, gather
            uint32_t carry[64]; int32_t run = -1;
            for(uint32_t lane = 0; lane < 64; lane++) { carry[lane] = run >= 0 ? S.o.aux[inf_aux_at((uint32_t)run)] : 0u; const int32_t l = inf_lz_last_start(S, lane); if(l > run) run = l; }
            uint32_t q[64][32], inr[64];
            for(uint32_t lane = 0; lane < 64; lane++) { inr[lane] = inf_lz_inrange(lane, base0, beg, end); inf_lz_sources(S, lane, base0, inr[lane], carry[lane], q[lane]); }
            for(uint32_t lane = 0; lane < 64; lane++) inf_lz_publish(S, lane, q[lane]);
            for(int guard = 0;; guard++) {
                if(guard > 64) return 106;
                bool moved = false;
                for(uint32_t lane = 0; lane < 64; lane++) moved |= inf_lz_jump(S, lane, base0, beg, inr[lane], q[lane]);
                for(uint32_t lane = 0; lane < 64; lane++) inf_lz_publish(S, lane, q[lane]);
                g_jump_rounds++;
                if(!moved) break;
            }
            bool any_far = false;
            for(uint32_t lane = 0; lane < 64; lane++) for(uint32_t j = 0; j < 32; j++) { const uint32_t p = base0 + 32 * lane + j, src = q[lane][j];
                if(((inr[lane] >> j) & 1u) != (p >= beg && p < end ? 1u : 0u)) return 110;
                if(p < beg || p >= end) { if(src != (p & 0xffffu)) return 109; continue; }
                if(src != p) { g_match_bytes++; if(src + INF_WIN < end) { g_far_bytes++; any_far = true; if(src >= beg) return 101; } if(src >= p) return 102; } }
            uint8_t snap[INF_WIN]; memcpy(snap, S.win, INF_WIN);      // all lanes read before any lane's write is seen: the gather must not depend on lane order ...
            for(int lane = 63; lane >= 0; lane--) { InfShared &W = S; uint8_t keep[INF_WIN]; memcpy(keep, W.win, INF_WIN); memcpy(W.win, snap, INF_WIN);      // ... so every lane reads the window as it stood, and its 32 bytes are merged in
                inf_lz_gather(W, (uint32_t)lane, base0, end, inr[lane], q[lane], any_far, [&](uint32_t at, bool wanted) -> uint32_t { return wanted ? out[at] : 0u; });
                const uint32_t slot = (base0 + 32u * (uint32_t)lane) & (INF_WIN - 1); memcpy(keep + slot, W.win + slot, 32); memcpy(W.win, keep, INF_WIN); }
            flush(beg, end);
        }
        if(fin && pos != out_len) return INF_E_SHORT;
        if(fin) return 0;
    }
}

static int check_stream(const std::vector<uint8_t> &raw, int level, int strategy, const char *what) {
    std::vector<uint8_t> comp(compressBound(raw.size()) + 64 + 8); z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    const int lead = 1 + (int)(raw.size() % 3);                                // an unaligned start, as in a file
    zs.next_in = (Bytef *)raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = comp.data() + lead; zs.avail_out = (uInt)(comp.size() - lead - 8);
    deflate(&zs, Z_FINISH); const uint32_t clen = (uint32_t)zs.total_out; deflateEnd(&zs);
    std::vector<uint8_t> got(raw.size() + 8, 0xEE); uint64_t c = 0;
    const int rc = emu_member(comp.data(), (uint64_t)lead, clen, got.data(), (uint32_t)raw.size(), &c);
    if(rc || (raw.size() && memcmp(got.data(), raw.data(), raw.size()))) { fprintf(stderr, "selftest FAILED: %s level %d strategy %d size %zu: rc %d\n", what, level, strategy, raw.size(), rc); return 1; }
    return 0;
}
static int selftest(void) {
    int bad = 0;
    {   // the arithmetic entries against RFC 1951 3.2.5's tables
        const uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258}; const uint8_t LEXT[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
        const uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577}; const uint8_t DEXT[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
        for(uint32_t k = 0; k < 29; k++) if(inf_lit_entry(257 + k) != (INF_L_LEN | ((uint32_t)LEXT[k] << 12) | ((uint32_t)(LBASE[k] - 3) << 4))) { fprintf(stderr, "length symbol %u\n", 257 + k); bad++; }
        for(uint32_t k = 0; k < 30; k++) if(inf_dist_entry(k) != (((uint32_t)DBASE[k] << 16) | ((uint32_t)DEXT[k] << 8))) { fprintf(stderr, "distance symbol %u\n", k); bad++; }
        if(inf_lit_entry(65) != (65u << 4) || inf_lit_entry(256) != INF_L_EOB || inf_lit_entry(286) != INF_L_BAD || inf_dist_entry(30) != INF_D_BAD) bad++;
    } uint64_t s = 88172645463325252ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
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
        std::vector<uint8_t> got((size_t)out_len + 64, 0xEE); uint64_t c = 0;
        const int rc = emu_member(comp.data(), (uint64_t)lead, clen, got.data(), out_len, &c);
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
    size_t o = 0; long m = 0, bad = 0; uint64_t tot = 0, n_batches = 0;
    std::vector<uint8_t> ref(65536 + 64), got(65536 + 64);
    while(o + 18 <= n && (maxm < 0 || m < maxm)) {
        const uint16_t xlen = (uint16_t)(raw[o + 10] | raw[o + 11] << 8); const uint32_t bs = (uint32_t)(raw[o + 16] | raw[o + 17] << 8) + 1; uint32_t isz; memcpy(&isz, &raw[o + bs - 4], 4);
        const uint64_t in_off = o + 12 + xlen; const uint32_t in_len = bs - 12 - xlen - 8;
        if(isz) {
            z_stream zs; memset(&zs, 0, sizeof zs); zs.next_in = raw.data() + in_off; zs.avail_in = in_len; zs.next_out = ref.data(); zs.avail_out = isz;
            inflateInit2(&zs, -15); const int zr = inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if(zr != Z_STREAM_END) { fprintf(stderr, "zlib failed on member %ld\n", m); return 2; }
            memset(got.data(), 0xEE, isz);
            const int rc = emu_member(raw.data(), in_off, in_len, got.data(), isz, &n_batches);
            if(rc || memcmp(got.data(), ref.data(), isz)) { if(bad < 5) { size_t k = 0; while(k < isz && got[k] == ref[k]) k++; fprintf(stderr, "member %ld (file offset %zu): rc %d, first difference at byte %zu of %u\n", m, o, rc, k, isz); } bad++; }
            tot += isz;
        }
        o += bs; m++;
    }
    printf("{\"members\": %ld, \"inflated_bytes\": %llu, \"mismatching_members\": %ld, \"batches\": %llu, \"header_batches\": %llu, \"huffman_batches\": %llu, \"stretch_words\": [%d, %d], \"batch_bytes\": %d, "
           "\"passes_per_huffman_batch\": %.2f, \"wave_steps_per_member\": %.1f, \"symbols\": %llu, \"symbols_per_wave_step\": %.2f, \"lane_steps_per_symbol\": %.2f, \"lanes_active_per_batch\": %.1f, \"lanes_taken_per_batch\": %.1f, "
           "\"output_batches\": %llu, \"token_groups\": %llu, \"batches_cut_by_passes\": %llu, \"jump_rounds_per_output_batch\": %.2f, \"match_bytes\": %llu, \"far_bytes\": %llu, \"chain_passes_hist\": [",
           m, (unsigned long long)tot, bad, (unsigned long long)n_batches, (unsigned long long)g_hdr_batches, (unsigned long long)g_hbatches, INF_SW_MIN, INF_SW_MAX, INF_BATCH_BYTES,
           g_hbatches ? (double)g_passes / (double)g_hbatches : 0.0, m ? (double)g_wave_steps / (double)m : 0.0, (unsigned long long)g_syms,
           g_wave_steps ? (double)g_syms / (double)g_wave_steps : 0.0, g_syms ? (double)g_lane_steps / (double)g_syms : 0.0, g_hbatches ? (double)g_lanes_active / (double)g_hbatches : 0.0, g_hbatches ? (double)g_lanes_taken / (double)g_hbatches : 0.0,
           (unsigned long long)g_obatches, (unsigned long long)g_tok_groups, (unsigned long long)g_cut_passes, g_obatches ? (double)g_jump_rounds / (double)g_obatches : 0.0,
           (unsigned long long)g_match_bytes, (unsigned long long)g_far_bytes);
    for(int i = 1; i <= INF_MAX_PASSES; i++) printf("%s%llu", i > 1 ? ", " : "", (unsigned long long)g_pass_hist[i]);
    printf("]}\n");
    return bad ? 1 : 0;
}

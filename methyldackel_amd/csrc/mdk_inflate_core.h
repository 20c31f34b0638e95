// mdk_inflate_core.h -- DEFLATE (RFC 1951) decoding of one BGZF member by ONE WAVEFRONT: the part that is sequential.
//
// Where this sits: the reference pays for BGZF inflate inside htslib's sam_itr_next (common.c:413); SURVEY.md 8(f) rank 1 moves it
// to the device.  A member (<= 64 KiB of BAM, bgzf.c of htslib: BGZF_BLOCK_SIZE 0xff00) is one raw deflate stream.  Where a symbol
// starts depends on every symbol before it -- but WHAT a symbol is depends only on the bits at its start and on the block's tables.  So
// inside a Huffman block the 64 lanes of the wavefront each decode the symbol that WOULD start at "their" bit (the next 64 bit
// positions of the stream: inf_decode_at, this file), and a short walk over the 64 results (lane 0's symbol is real; the next real one
// starts where it ends; ...) picks the 5-8 of them that are: one round of table lookups per ~7 symbols instead of per symbol.  Block
// headers go through the sequential bit reader below, by one lane.  Everything else is done by all 64 lanes in mdk_inflate.hip:
// staging the compressed words into an LDS ring, copying the LZ77 matches (sources further back than the LDS window come from global
// memory, nearer ones from the window) and writing the finished bytes out coalesced.
//
// The code here is plain C++ with no wave intrinsics: it compiles for the device (hipcc) and for the host (tests/emulation:
// tools/inflate_emu.cpp runs the same functions over real BGZF files and compares with zlib), which is how it is tested
// without a GPU.  Nothing in it is taken from zlib or libdeflate; the two-level table layout follows the published
// technique (root table + sub-tables for codes longer than the root, as in zlib's inflate_table) restated from RFC 1951.
#ifndef MDK_INFLATE_CORE_H
#define MDK_INFLATE_CORE_H
#include <stdint.h>

#if defined(__HIPCC__)
#define MDK_HD __device__ __forceinline__
#define MDK_HDN __device__ __noinline__
#define MDK_HDM __device__ __forceinline__
#else
#define MDK_HD static inline
#define MDK_HDN static
#define MDK_HDM inline
#endif

// ---- geometry of one wavefront's LDS state ----
#define INF_LIT_TB    9                    // root bits of the literal/length table
#define INF_DIST_TB   8                    // root bits of the distance table
#define INF_LIT_CAP   864                  // entries: 512 root + sub-tables (a complete code of 286 symbols with root 9 needs at most 852)
#define INF_DIST_CAP  416                  // entries: 256 root + sub-tables (at most 402 with root 8)
#define INF_IN_WORDS  256                  // compressed-input ring, 32-bit words (power of two)
#ifndef INF_WIN
#define INF_WIN       2048                 // output window ring, bytes (power of two)
#endif
#define INF_BATCH_BYTES (INF_WIN / 2)      // a batch never produces more than this
#define INF_BATCH_WORDS 150                // ... nor takes more than this many words from the input ring (the ring is topped up to >= 193 ahead)
#define INF_MAX_TOK   128                  // ... nor holds more match tokens than this (two places per lane: with one, three quarters of the batches of a BAM member ended at ~470 bytes because the places were full -- 81 % of its symbols are matches)
#define INF_NEAR_LANE_MAX 32u              // a match up to this long whose source is final is copied by its own lane; longer ones by the whole wavefront

// Literal/length table entry, 16 bits (the tables of a wavefront are 3.3 KiB of LDS):
//   length symbol   1 eee bbbbbbbb nnnn   e = extra bits (0..5), b = base length - 3 (0..255), n = code bits to consume
//   everything else 0 kk vvvvvvvvv nnnn   k = 0 literal (v = byte), 1 sub-table pointer (v = first entry - 512, n = its index bits),
//                                         2 end of block, 3 unassigned code space
// so that the decoder's commonest case -- a length -- is one bit test and three field extractions.
typedef uint16_t inf_lit_t;
#define INF_L_LEN   0x8000u
#define INF_L_KIND(e) ((e) & 0x6000u)
#define INF_L_LIT   0x0000u
#define INF_L_SUB   0x2000u
#define INF_L_EOB   0x4000u
#define INF_L_BAD   0x6000u
// Distance table entry, 32 bits: bits 0-3 code bits to consume (pointer: index bits of the sub-table), bits 4-5 kind (0 distance
// symbol, 1 sub-table pointer, 3 unassigned), bits 8-11 extra bits, bits 16-31 base distance (pointer: first entry of the sub-table).
// The code-length alphabet of a dynamic block header is decoded through the same memory: kind 0, "base" = symbol.
typedef uint32_t inf_dist_t;
#define INF_D_KIND(e) ((e) & 0x30u)
#define INF_D_SYM   0x00u
#define INF_D_SUB   0x10u
#define INF_D_BAD   0x30u

// error codes (0 = fine)
enum { INF_OK = 0, INF_E_BTYPE = 1, INF_E_STORED = 2, INF_E_HEADER = 3, INF_E_CODELEN = 4, INF_E_LITTABLE = 5, INF_E_DISTTABLE = 6, INF_E_SYMBOL = 7,
       INF_E_DIST = 8, INF_E_OVERRUN = 9, INF_E_INPUT = 10, INF_E_SHORT = 11, INF_E_CRC = 12 };

struct InfToken { uint32_t dst; uint32_t len_dist; };       // dst: position in the member's output; len | dist << 16

// Everything one wavefront keeps in LDS for the member it inflates.
struct InfShared {
    inf_dist_t dist[INF_DIST_CAP];
    inf_lit_t lit[INF_LIT_CAP];
    uint32_t in[INF_IN_WORDS];             // ring of compressed words: word w of the stream lives in in[w & (INF_IN_WORDS-1)]
    uint8_t  win[INF_WIN];                 // ring of output bytes: byte p of the member lives in win[p & (INF_WIN-1)]
    InfToken tok[INF_MAX_TOK];
    // written by the lane that parsed a block header, read by all lanes after the barrier
    uint32_t bitpos, in_block, last, stored_left, err;
};

// The decoder's registers between batches.  `bb` holds the next `cnt` bits of the stream, least significant first; 32 <= cnt <= 63
// between any two steps, so a step can look at 32 bits without asking.
struct InfDec {
    uint64_t bb; uint32_t cnt;
    uint32_t nx;                           // the stream word after the ones in bb, fetched one refill ahead of its use
    uint32_t widx;                         // next stream word to fetch from the ring
    uint32_t pos;                          // bytes produced so far
    uint32_t out_len;                      // the member's ISIZE
    uint32_t in_block;                     // 0: a block header comes next; 1: inside a Huffman block; 2: inside a stored block
    uint32_t last;                         // the current block is the final one
    uint32_t stored_left;                  // bytes of the stored block still to copy
};

#define INF_LD(x) ((uint32_t)(x))
MDK_HD uint32_t inf_peek(const InfDec &d) { return (uint32_t)d.bb; }
template <bool UNI>
MDK_HD void inf_consume(InfDec &d, const uint32_t *in, uint32_t n) {     // n <= 32
    d.bb >>= n; d.cnt -= n;
    if(d.cnt < 32) { d.bb |= (uint64_t)d.nx << d.cnt; d.cnt += 32; d.nx = INF_LD(in[d.widx & (INF_IN_WORDS - 1)]); d.widx++; }
}
template <bool UNI>
MDK_HD uint32_t inf_get(InfDec &d, const uint32_t *in, uint32_t n) {     // n <= 16
    const uint32_t v = inf_peek(d) & ((1u << n) - 1u);
    inf_consume<UNI>(d, in, n);
    return v;
}
// The stream's bits are numbered from bit 0 of ring word 0 (the member's first byte sits 8 * skip_bytes bits in).  The bit reader at a
// position, and the position of a bit reader:
MDK_HD void inf_dec_seek(InfDec &d, const uint32_t *in, uint32_t bitpos) {
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    d.bb = ((uint64_t)in[w & (INF_IN_WORDS - 1)] | ((uint64_t)in[(w + 1) & (INF_IN_WORDS - 1)] << 32)) >> sh; d.cnt = 64 - sh;
    d.nx = in[(w + 2) & (INF_IN_WORDS - 1)]; d.widx = w + 3;
}
MDK_HD uint32_t inf_dec_tell(const InfDec &d) { return 32u * (d.widx - 1u) - d.cnt; }

MDK_HD uint32_t inf_rev(uint32_t code, int len) {          // the low `len` bits of code, reversed
    uint32_t r = 0;
    for(int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// entries of the three alphabets (without the code-bit count, which the builder adds)
MDK_HD uint32_t inf_lit_entry(int sym) {
    const uint16_t LBASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    const uint8_t LEXT[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    if(sym < 256) return INF_L_LIT | ((uint32_t)sym << 4);
    if(sym == 256) return INF_L_EOB;
    if(sym < 286) return INF_L_LEN | ((uint32_t)LEXT[sym - 257] << 12) | ((uint32_t)(LBASE[sym - 257] - 3) << 4);
    return INF_L_BAD;
}
MDK_HD uint32_t inf_dist_entry(int sym) {
    const uint16_t DBASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    const uint8_t DEXT[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    if(sym < 30) return ((uint32_t)DBASE[sym] << 16) | ((uint32_t)DEXT[sym] << 8) | INF_D_SYM;
    return INF_D_BAD;
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused, <= 15) -> two-level decode table.
// kind 0: literal/length alphabet (T = inf_lit_t), 1: distance alphabet, 2: code-length alphabet (both T = inf_dist_t).
// An over-subscribed set of lengths is an error.  An incomplete one is an error too, as in zlib (inftrees.c: "incomplete set"), unless
// it has no code at all or a single code of length 1 (RFC 1951 3.2.7: one distance code) -- a stream zlib rejects is rejected here
// (tools/inflate_emu.cpp --fuzz); `strict` = 0 is for the fixed block's distance code, whose 30 five-bit codes leave two unassigned.
template <typename T>
MDK_HD int inf_build_t(const uint8_t *lens, int n, int tb, T *tab, int cap, int kind, uint16_t *sorted /* [n] scratch */, int strict) {
    uint16_t count[16], offs[16];
    const uint32_t bad = kind == 0 ? INF_L_BAD : INF_D_BAD;
    for(int l = 0; l < 16; l++) count[l] = 0;
    for(int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    int maxl_used = 0;
    for(int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if(left < 0) return -1; if(count[l]) maxl_used = l; }
    if(strict && left > 0 && maxl_used != 0 && (kind == 2 || maxl_used != 1)) return -3;
    offs[1] = 0;
    for(int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    int total = 0;
    for(int i = 0; i < n; i++) if(lens[i]) { sorted[offs[lens[i]]++] = (uint16_t)i; total++; }
    const int root = 1 << tb;
    for(int i = 0; i < root; i++) tab[i] = (T)bad;
    int next_free = root;
    uint32_t code = 0; int idx = 0;
    int sub_prefix = -1, sub_start = 0, sub_bits = 0;
    for(int l = 1; l <= 15; l++) {
        for(int k = 0; k < count[l]; k++, idx++, code++) {
            const int sym = sorted[idx];
            uint32_t e = kind == 0 ? inf_lit_entry(sym) : kind == 1 ? inf_dist_entry(sym) : (((uint32_t)sym << 16) | INF_D_SYM);
            const uint32_t rev = inf_rev(code, l);
            if(l <= tb) {
                e |= (uint32_t)l;
                for(uint32_t j = rev; j < (uint32_t)root; j += 1u << l) tab[j] = (T)e;
            } else {
                const int prefix = (int)(rev & (uint32_t)(root - 1));
                if(prefix != sub_prefix) {
                    // the codes sharing these first tb bits are the next ones in canonical order, lengths ascending: the last of them sizes the sub-table
                    int maxl = l; uint32_t c2 = code; int l2 = l, k2 = k, i2 = idx;
                    for(;;) {
                        i2++; k2++; c2++;                                     // (c2, l2) -> the next code in canonical order
                        while(l2 <= 15 && k2 >= count[l2]) { l2++; k2 = 0; c2 <<= 1; }
                        if(l2 > 15 || i2 >= total) break;
                        if((int)(inf_rev(c2, l2) & (uint32_t)(root - 1)) != prefix) break;
                        maxl = l2;
                    }
                    sub_prefix = prefix; sub_bits = maxl - tb; sub_start = next_free;
                    if(sub_start + (1 << sub_bits) > cap) return -2;
                    next_free += 1 << sub_bits;
                    for(int j = 0; j < (1 << sub_bits); j++) tab[sub_start + j] = (T)bad;
                    tab[prefix] = kind == 0 ? (T)(INF_L_SUB | ((uint32_t)(sub_start - root) << 4) | (uint32_t)sub_bits)
                                            : (T)(((uint32_t)sub_start << 16) | INF_D_SUB | (uint32_t)sub_bits);
                }
                e |= (uint32_t)(l - tb);
                for(uint32_t j = rev >> tb; j < (1u << sub_bits); j += 1u << (l - tb)) tab[sub_start + j] = (T)e;
            }
        }
        code <<= 1;
    }
    return 0;
}
MDK_HDN int inf_build_lit(const uint8_t *lens, int n, inf_lit_t *tab, uint16_t *sorted) { return inf_build_t<inf_lit_t>(lens, n, INF_LIT_TB, tab, INF_LIT_CAP, 0, sorted, 1); }
MDK_HDN int inf_build_dist(const uint8_t *lens, int n, int tb, inf_dist_t *tab, int kind, uint16_t *sorted, int strict = 1) { return inf_build_t<inf_dist_t>(lens, n, tb, tab, INF_DIST_CAP, kind, sorted, strict); }

// Block header (RFC 1951 3.2.3-3.2.7); for a dynamic block also the code lengths and both tables.  The ring must hold the
// whole header (a dynamic header is at most 14 + 19*3 + 316*14 bits < 160 words).
MDK_HD int inf_block_header_body(InfDec &d, InfShared &S) {
    const uint8_t CLORD[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    uint8_t lens[320]; uint16_t sorted[288];
    d.last = inf_get<false>(d, S.in, 1);
    const uint32_t type = inf_get<false>(d, S.in, 2);
    if(type == 3) return INF_E_BTYPE;
    if(type == 0) {
        inf_consume<false>(d, S.in, d.cnt & 7);                  // to the next byte boundary: every word put into bb was whole bytes, so the bits still in bb tell
        const uint32_t len = inf_get<false>(d, S.in, 16), nlen = inf_get<false>(d, S.in, 16);
        if((len ^ 0xffffu) != nlen) return INF_E_STORED;
        d.stored_left = len; d.in_block = 2;
        return INF_OK;
    }
    if(type == 1) {
        for(int k = 0; k < 144; k++) lens[k] = 8;
        for(int k = 144; k < 256; k++) lens[k] = 9;
        for(int k = 256; k < 280; k++) lens[k] = 7;
        for(int k = 280; k < 288; k++) lens[k] = 8;
        if(inf_build_lit(lens, 288, S.lit, sorted)) return INF_E_LITTABLE;
        for(int k = 0; k < 30; k++) lens[k] = 5;
        if(inf_build_dist(lens, 30, INF_DIST_TB, S.dist, 1, sorted, 0)) return INF_E_DISTTABLE;
        d.in_block = 1;
        return INF_OK;
    }
    const int nlit = (int)inf_get<false>(d, S.in, 5) + 257, ndist = (int)inf_get<false>(d, S.in, 5) + 1, ncode = (int)inf_get<false>(d, S.in, 4) + 4;
    if(nlit > 286 || ndist > 30) return INF_E_HEADER;
    for(int k = 0; k < 19; k++) lens[k] = 0;
    for(int k = 0; k < ncode; k++) lens[CLORD[k]] = (uint8_t)inf_get<false>(d, S.in, 3);
    // the code-length code (<= 7 bits) is decoded through the distance table's memory, which is rebuilt afterwards
    if(inf_build_dist(lens, 19, 7, S.dist, 2, sorted)) return INF_E_CODELEN;
    int idx = 0; const int want = nlit + ndist;
    while(idx < want) {
        const uint32_t e = S.dist[inf_peek(d) & 127u];
        if(INF_D_KIND(e) != INF_D_SYM) return INF_E_CODELEN;
        inf_consume<false>(d, S.in, e & 15u);
        const int sym = (int)(e >> 16);
        if(sym < 16) lens[idx++] = (uint8_t)sym;
        else {
            int prev = 0, rep;
            if(sym == 16) { if(idx == 0) return INF_E_CODELEN; prev = lens[idx - 1]; rep = 3 + (int)inf_get<false>(d, S.in, 2); }
            else if(sym == 17) rep = 3 + (int)inf_get<false>(d, S.in, 3);
            else rep = 11 + (int)inf_get<false>(d, S.in, 7);
            if(idx + rep > want) return INF_E_CODELEN;
            while(rep--) lens[idx++] = (uint8_t)prev;
        }
    }
    if(lens[256] == 0) return INF_E_LITTABLE;                  // a block without an end-of-block code never ends
    if(inf_build_dist(lens + nlit, ndist, INF_DIST_TB, S.dist, 1, sorted)) return INF_E_DISTTABLE;
    if(inf_build_lit(lens, nlit, S.lit, sorted)) return INF_E_LITTABLE;
    d.in_block = 1;
    return INF_OK;
}
// ... out of line, the state in and out BY VALUE: a decoder state whose address an out-of-line call had taken would live in
// private memory, and the compiler treats whatever is loaded from there as divergent (no scalar registers, no scalar branches).
struct InfHdr { InfDec d; int err; };
MDK_HDN InfHdr inf_block_header(const InfDec d_in, InfShared &S) {
    InfHdr H; H.d = d_in; H.err = inf_block_header_body(H.d, S);
    return H;
}

// A block header by ONE lane, through the bit reader; what it leaves (position behind the header, kind of block, tables) goes to S for all lanes.
MDK_HD void inf_header_batch(InfShared &S, uint32_t bitpos) {
    InfDec d; inf_dec_seek(d, S.in, bitpos);
    d.pos = 0; d.out_len = 0; d.in_block = 0; d.last = 0; d.stored_left = 0;
    const InfHdr H = inf_block_header(d, S);
    S.bitpos = inf_dec_tell(H.d); S.in_block = H.d.in_block; S.last = H.d.last; S.stored_left = H.d.stored_left; S.err = (uint32_t)H.err;
}

// The symbol that starts at bit `bitpos` of the stream, whatever stands in front of it: kind 0 a literal (val = the byte), 1 a match (val =
// length | distance << 16), 2 end of block, 3 / 4 not a code of the literal/length / of the distance alphabet.  nbits = its bits, extra
// bits and distance code included (at most 15 + 5 + 15 + 13 = 48, of the 64 fetched).
struct InfSym { uint32_t nbits, kind, val; };
MDK_HD InfSym inf_decode_at(const InfShared &S, const uint32_t bitpos) {
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    const uint32_t w0 = S.in[w & (INF_IN_WORDS - 1)], w1 = S.in[(w + 1) & (INF_IN_WORDS - 1)], w2 = S.in[(w + 2) & (INF_IN_WORDS - 1)];
    // (the three words are asked for together; w2's share without a branch: shifted out whole when sh = 0)
    const uint64_t b = (((uint64_t)w0 | ((uint64_t)w1 << 32)) >> sh) | ((((uint64_t)w2) << 1) << (63u - sh));
    uint32_t x = (uint32_t)b, used = 0;
    uint32_t e = S.lit[x & ((1u << INF_LIT_TB) - 1u)];
    if((e & (INF_L_LEN | 0x6000u)) == INF_L_SUB) {        // a code longer than the root table
        x >>= INF_LIT_TB; used = INF_LIT_TB;
        e = S.lit[(1u << INF_LIT_TB) + ((e >> 4) & 511u) + (x & ((1u << (e & 15u)) - 1u))];
    }
    const uint32_t nb = e & 15u;
    InfSym s;
    if(e & INF_L_LEN) {
        const uint32_t eb = (e >> 12) & 7u, len = ((e >> 4) & 255u) + 3u + ((x >> nb) & ((1u << eb) - 1u));
        const uint32_t t1 = used + nb + eb;
        uint32_t y = (uint32_t)(b >> t1), used2 = 0;
        uint32_t f = S.dist[y & ((1u << INF_DIST_TB) - 1u)];
        if(INF_D_KIND(f) == INF_D_SUB) { y >>= INF_DIST_TB; used2 = INF_DIST_TB; f = S.dist[(f >> 16) + (y & ((1u << (f & 15u)) - 1u))]; }
        const uint32_t nb2 = f & 15u, eb2 = (f >> 8) & 15u, dist = (f >> 16) + ((y >> nb2) & ((1u << eb2) - 1u));
        s.nbits = t1 + used2 + nb2 + eb2; s.kind = INF_D_KIND(f) == INF_D_SYM ? 1u : 4u; s.val = len | (dist << 16);
    } else {
        const uint32_t k = INF_L_KIND(e);
        s.nbits = used + nb; s.kind = k == INF_L_LIT ? 0u : k == INF_L_EOB ? 2u : 3u; s.val = (e >> 4) & 255u;
    }
    return s;
}
// byte `i` of the stream behind the byte-aligned position `bitpos` (a stored block's bytes), out of the ring
MDK_HD uint8_t inf_ring_byte(const InfShared &S, uint32_t bitpos, uint32_t i) {
    const uint32_t a = (bitpos >> 3) + i;
    return (uint8_t)(S.in[(a >> 2) & (INF_IN_WORDS - 1)] >> (8u * (a & 3u)));
}
#define INF_STORED_BATCH 512u                 // bytes of a stored block copied per batch (the ring is kept >= 193 words ahead)

// Has a finished member consumed more bits than its stream holds?  (Words past the stream read as zero, and zeros can decode: seven of
// them are the end-of-block code of a fixed block.  zlib calls that stream truncated; so do we.)
MDK_HD bool inf_overran_input(uint32_t bitpos, uint32_t skip_bytes, uint32_t in_len) {
    return (uint64_t)bitpos > 8ull * ((uint64_t)skip_bytes + in_len);
}

// ---- the parts every lane runs (bodies only; the barriers between them are the caller's) ----
// A token is NEAR when its whole source lies in the window ring as it will be at the end of this batch, FAR when the source
// starts before that: then it is older than INF_WIN - INF_BATCH_BYTES >= 258 bytes before the batch, so it does not overlap its
// own output and every byte of it was written to global memory by an earlier batch.
MDK_HD bool inf_tok_far(const InfToken &t, uint32_t batch_beg) {
    const uint32_t dist = t.len_dist >> 16;
    return t.dst - dist + (INF_WIN - INF_BATCH_BYTES) < batch_beg;        // src_start < batch_beg - (WIN - BATCH), without going negative
}
// one round of a near match (see mdk_inflate.hip): bytes [done, done+n) of the match come from the first n bytes of the pattern
MDK_HD void inf_near_round(uint8_t *win, uint32_t dst, uint32_t dist, uint32_t done, uint32_t n, uint32_t lane) {
    for(uint32_t i = lane; i < n; i += 64) win[(dst + done + i) & (INF_WIN - 1)] = win[(dst - dist + i) & (INF_WIN - 1)];
}
#endif

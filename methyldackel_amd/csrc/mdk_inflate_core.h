// mdk_inflate_core.h -- DEFLATE (RFC 1951) decoding of one BGZF member by ONE WAVEFRONT: the part that is sequential.
//
// Where this sits: the reference pays for BGZF inflate inside htslib's sam_itr_next (common.c:413); SURVEY.md 8(f) rank 1 moves it
// to the device.  A member (<= 64 KiB of BAM, bgzf.c of htslib: BGZF_BLOCK_SIZE 0xff00) is one raw deflate stream.  Where a symbol
// starts depends on every symbol before it -- but WHAT a symbol is depends only on the bits at its start and on the block's tables, and
// inside a Huffman block nothing else is carried from one symbol to the next.  So two decoders that start at different bits of a block and
// ever stand on the same bit stay together from there on, and they do meet: Huffman codes synchronise themselves (on BAM data half of all
// wrong starts have met the true decoder after 100 bits, 97 % after 512, 99.9 % after 1024: tools/round6/sync_stats.cpp).  A Huffman
// batch therefore cuts the next 64 * 32 * sw bits of the stream (sw <= INF_SW_MAX words) into 64 stretches, one per lane.  Every lane
// decodes a CHAIN of symbols (inf_chain, this file) from a guessed start -- the first bit of its stretch -- until the chain leaves the
// stretch: where it leaves is, nearly always, where the true decoder leaves it too.  Then every lane starts again where its left neighbour's
// chain ended, this time writing its symbols down as TOKENS (a literal's byte, a match's length and distance) into a scratch area in global
// memory, token j of all lanes side by side; a lane whose neighbour ended somewhere else this time goes once more.  Lane 0's start is the true
// one, so the lanes up to the first whose start is not its neighbour's end hold the stream (all 64 after two passes in 19 of 20 batches).
// The tokens are then turned into bytes in batches of <= INF_BATCH_BYTES through an output window in LDS (mdk_inflate.hip): a prefix sum
// gives every token its place, literals go into the window, a match notes its distance at its first byte; then ALL bytes of the batch are
// resolved at once -- every byte points at its source, pointers into the batch's own matches are followed by pointer jumping (log of the
// longest dependence chain), one gather fills the window.  Block headers go through the sequential bit reader below, by one lane.
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
#define INF_LIT_TB    10                   // index bits of the literal/length table: a code up to this long is one lookup
#define INF_DIST_TB   8                    // ... of the distance table
#define INF_CL_TB     7                    // ... of the code-length alphabet's table (its codes are never longer)
#ifndef INF_SW_MAX
#define INF_SW_MAX    19                   // words of the stream per lane and Huffman batch, at most (odd: the lanes' stretches start in different LDS banks).  19: the staged words then fit next to the
                                           // tables and the window in 12.9 KB, twelve wavefronts per CU instead of ten (31 words: 15.1 ms for the bench's file, 19: 14.3; 608 bits still synchronise 98 % of the chains)
#endif
#ifndef INF_SW_MIN
#define INF_SW_MIN    19                   // ... and at least (a stretch must be long enough for a chain to meet the true decoder inside it)
#endif
#define INF_IN_CAP    (64 * INF_SW_MAX + 8)    // staged compressed words of a batch: S.in[0] is word `wbase` of the stream, and every bit position the functions below take counts from ITS bit 0
#define INF_HDR_WORDS 168                  // what a header batch stages (a dynamic header is at most 14 + 19*3 + 316*14 bits < 160 words)
#ifndef INF_WIN
#define INF_WIN       4096                 // output window ring, bytes (power of two)
#endif
#define INF_BATCH_BYTES (INF_WIN / 2)      // a batch of output never holds more than this
#define INF_PER       (INF_BATCH_BYTES / 64)   // bytes of a batch each lane resolves (a multiple of 32)
#define INF_TOK_STEPS 256                  // tokens one lane writes per Huffman batch at most (its column of the scratch area); 64 KiB of scratch per wavefront
#define INF_TOK_WORDS (64 * INF_TOK_STEPS)
#ifndef INF_MAX_PASSES
#define INF_MAX_PASSES 6                   // passes of chains per Huffman batch before the lanes that agree so far are taken
#endif

// Literal/length table entry, 16 bits:
//   length symbol   1 eee bbbbbbbb nnnn   e = extra bits (0..5), b = base length - 3 (0..255), n = code bits to consume
//   everything else 0 kk vvvvvvvvv nnnn   k = 0 literal (v = byte), 1 the first bits of a code LONGER than the table's index (found by its
//                                         canonical value instead: inf_long_code), 2 end of block, 3 unassigned code space
// so that the decoder's commonest case -- a length -- is one bit test and three field extractions.
typedef uint16_t inf_lit_t;
#define INF_L_LEN   0x8000u
#define INF_L_KIND(e) ((e) & 0x6000u)
#define INF_L_LIT   0x0000u
#define INF_L_LONG  0x2000u
#define INF_L_EOB   0x4000u
#define INF_L_BAD   0x6000u
// Distance table entry, 32 bits: bits 0-3 code bits to consume, bits 4-5 kind (0 distance symbol, 1 the first bits of a longer code,
// 3 unassigned), bits 8-11 extra bits, bits 16-31 base distance.
// The code-length alphabet of a dynamic block header is decoded through the same memory: kind 0, "base" = symbol.
typedef uint32_t inf_dist_t;
#define INF_D_KIND(e) ((e) & 0x30u)
#define INF_D_SYM   0x00u
#define INF_D_LONG  0x10u
#define INF_D_BAD   0x30u
// The codes longer than a table's index, by canonical value (RFC 1951 3.2.2): the codes of length l are count[l] consecutive values from
// first[l], and the symbols they stand for are the next count[l] of the alphabet's symbols in canonical order (sym[off[l] ..]).
struct InfLong { uint32_t fc[16]; uint16_t off[16]; };      // fc[l] = first[l] | count[l] << 16

// error codes (0 = fine)
enum { INF_OK = 0, INF_E_BTYPE = 1, INF_E_STORED = 2, INF_E_HEADER = 3, INF_E_CODELEN = 4, INF_E_LITTABLE = 5, INF_E_DISTTABLE = 6, INF_E_SYMBOL = 7,
       INF_E_DIST = 8, INF_E_OVERRUN = 9, INF_E_INPUT = 10, INF_E_SHORT = 11, INF_E_CRC = 12 };

// Everything one wavefront keeps in LDS for the member it inflates.
struct InfShared {
    inf_dist_t dist[1 << INF_DIST_TB];
    inf_lit_t lit[1 << INF_LIT_TB];
    uint16_t lsym[288], dsym[32];          // the alphabets' symbols in canonical order (what a long code's value indexes)
    InfLong ll, dl;
    alignas(16) uint8_t win[INF_WIN];      // ring of output bytes: byte p of the member lives in win[p & (INF_WIN-1)]
    union alignas(16) {
        uint32_t in[INF_IN_CAP];           // a header / stored / Huffman batch: the staged compressed words
        struct {                           // a header batch, behind its staged words: the code lengths on their way into tables
            uint32_t in_[INF_HDR_WORDS + 4];
            uint8_t lens[320 + 32];        //   literal/length code lengths, then the distance code lengths (a fixed block: 288 + 32)
            uint8_t cl[20];                //   the code-length alphabet's code lengths
            uint32_t type, nlit, ndist, rank_base[16];
        } h;
        struct {                           // a batch of output:
            alignas(16) uint16_t aux[INF_BATCH_BYTES]; //   per byte: first the distance noted at a match's first byte (0 at a literal), then the byte's source (a position in the member: < 65536)
            uint32_t starts[INF_BATCH_BYTES / 32];   // bit r: a token's output starts at byte r of the batch
            uint32_t tpre[66];             //   tpre[i] = tokens of the lanes before lane i (tpre[k] = all of them)
        } o;
    };
    // written by the lane that parsed a block header, read by all lanes after the barrier
    uint32_t bitpos, in_block, last, stored_left, err;
};
// Where byte p of the member lives in the window ring, and entry r of a batch's per-byte state in S.o.aux.  In the batch steps lane l owns 32
// consecutive bytes, so the 64 lanes' accesses of one instruction lie 64 bytes (16 dwords) apart in aux: straight addressing would put a
// group of 32 lanes on two of the LDS's 32 banks -- every entry of a pointer-jumping round a 16-way bank conflict.  So aux's dword index is
// XORed with the bits above it that count the lanes (dword bits 0-3 with bits 5-8: conflict-free for a lane's own entries and for sources at
// a common distance; the two entries of a dword stay together): k_inflate 14.3 -> 12.8 ms for the bench's file.  The same for the window
// (bytes 32 apart: 8-way) measured nothing (12.86 against 12.81 ms, gpurun_out/r06i4_*: its gather reads scattered sources anyway) and costs
// instructions per byte: the window is addressed straight.  Every access of either array goes through these two.
typedef uint32_t __attribute__((may_alias)) inf_u32a;      // (dword views of the byte / halfword arrays: may_alias keeps the compilers' type-based reordering off them)
MDK_HD uint32_t inf_win_at(uint32_t p) { return p & (INF_WIN - 1); }
#ifndef INF_AUX_STRAIGHT
MDK_HD uint32_t inf_aux_at(uint32_t r) { return r ^ (((r >> 6) & 15u) << 1); }
#else
MDK_HD uint32_t inf_aux_at(uint32_t r) { return r; }      /* (counter comparisons: profiles/r06i5_inflate_lds_counters.txt) */
#endif

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
    if(d.cnt < 32) { d.bb |= (uint64_t)d.nx << d.cnt; d.cnt += 32; d.nx = INF_LD(in[d.widx]); d.widx++; }
}
template <bool UNI>
MDK_HD uint32_t inf_get(InfDec &d, const uint32_t *in, uint32_t n) {     // n <= 16
    const uint32_t v = inf_peek(d) & ((1u << n) - 1u);
    inf_consume<UNI>(d, in, n);
    return v;
}
// Bits are numbered from bit 0 of the first staged word.  The bit reader at a position, and the position of a bit reader:
MDK_HD void inf_dec_seek(InfDec &d, const uint32_t *in, uint32_t bitpos) {
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    d.bb = ((uint64_t)in[w] | ((uint64_t)in[w + 1] << 32)) >> sh; d.cnt = 64 - sh;
    d.nx = in[w + 2]; d.widx = w + 3;
}
MDK_HD uint32_t inf_dec_tell(const InfDec &d) { return 32u * (d.widx - 1u) - d.cnt; }

MDK_HD uint32_t inf_rev32(uint32_t v) {                    // all 32 bits reversed
#if defined(__HIPCC__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1); v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4); v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// entries of the alphabets (without the code-bit count, which the builder adds), by arithmetic on the symbol: RFC 1951 3.2.5's tables of
// base lengths / distances and extra bits are regular (four symbols per number of extra bits, two for distances)
MDK_HD uint32_t inf_lit_entry(uint32_t sym) {
    if(sym < 256) return INF_L_LIT | (sym << 4);
    if(sym == 256) return INF_L_EOB;
    if(sym > 285) return INF_L_BAD;
    const uint32_t k = sym - 257;
    uint32_t eb = 0, base = 3 + k;
    if(k == 28) base = 258; else if(k >= 8) { eb = (k - 4) >> 2; base = 3 + ((4 + (k & 3u)) << eb); }
    return INF_L_LEN | (eb << 12) | ((base - 3) << 4);
}
MDK_HD uint32_t inf_dist_entry(uint32_t sym) {
    if(sym > 29) return INF_D_BAD;
    uint32_t eb = 0, base = 1 + sym;
    if(sym >= 4) { eb = (sym - 2) >> 1; base = 1 + ((2 + (sym & 1u)) << eb); }
    return (base << 16) | (eb << 8) | INF_D_SYM;
}

// ---- canonical Huffman code of n symbols with lengths lens[] (0 = unused, <= 15) -> decode table.  kind 0: literal/length alphabet (T =
// inf_lit_t), 1: distance alphabet, 2: code-length alphabet (both T = inf_dist_t).  The work is cut so that 64 lanes can share it
// (mdk_inflate.hip build_table) and the host runs the same pieces in a loop (inf_build_serial, below): count the lengths; inf_code_space
// says whether they make a usable code; inf_code_layout gives every length its first code and its first place in canonical order; then
// every symbol, knowing its rank among the symbols of its length, fills its own entries (inf_place_symbol). ----
// An over-subscribed set of lengths is an error.  An incomplete one is an error too, as in zlib (inftrees.c: "incomplete set"), unless
// it has no code at all or a single code of length 1 (RFC 1951 3.2.7: one distance code) -- a stream zlib rejects is rejected here
// (tools/inflate_emu.cpp --fuzz); `strict` = 0 is for the fixed block's distance code, whose 30 five-bit codes leave two unassigned.
MDK_HD int inf_code_space(const uint32_t *count /* [16], count[0] ignored */, int kind, int strict) {
    int left = 1, maxl_used = 0;
    for(int l = 1; l < 16; l++) { left <<= 1; left -= (int)count[l]; if(left < 0) return -1; if(count[l]) maxl_used = l; }
    if(strict && left > 0 && maxl_used != 0 && (kind == 2 || maxl_used != 1)) return -3;
    return 0;
}
MDK_HD void inf_code_layout(const uint32_t *count, uint32_t *first /* [16] */, uint32_t *off /* [16] */) {
    uint32_t code = 0, o = 0;
    first[0] = 0; off[0] = 0;
    for(int l = 1; l < 16; l++) { first[l] = code; off[l] = o; code = (code + count[l]) << 1; o += count[l]; }
}
template <typename T>
MDK_HD void inf_table_clear(T *tab, int tb, int kind, uint32_t lane) {     // every entry "unassigned" (what an incomplete code leaves)
    for(uint32_t i = lane; i < (1u << tb); i += 64) tab[i] = (T)(kind == 0 ? INF_L_BAD : INF_D_BAD);
}
// symbol `sym` of length l (> 0) is the `rank`-th symbol in canonical order; its code is first[l] + (rank - off[l])
template <typename T>
MDK_HD void inf_place_symbol(T *tab, uint16_t *symtab, int tb, int kind, uint32_t sym, uint32_t l, uint32_t code, uint32_t rank) {
    uint32_t e = kind == 0 ? inf_lit_entry(sym) : kind == 1 ? inf_dist_entry(sym) : ((sym << 16) | INF_D_SYM);
    const uint32_t rev = inf_rev32(code) >> (32u - l);
    if(symtab) symtab[rank] = (uint16_t)sym;
    if(l <= (uint32_t)tb) { e |= l; for(uint32_t j = rev; j < (1u << tb); j += 1u << l) tab[j] = (T)e; }
    else tab[rev & ((1u << tb) - 1u)] = (T)(kind == 0 ? INF_L_LONG : INF_D_LONG);      // (every code that starts with these bits is a long one: the code is prefix-free)
}
MDK_HD void inf_long_store(InfLong &L, const uint32_t *count, const uint32_t *first, const uint32_t *off, uint32_t l) { L.fc[l] = first[l] | (count[l] << 16); L.off[l] = (uint16_t)off[l]; }
// the same on one thread (host tests; the code-length alphabet's 19 symbols)
template <typename T>
MDK_HD int inf_build_serial(const uint8_t *lens, int n, int tb, T *tab, uint16_t *symtab, InfLong *L, int kind, int strict) {
    uint32_t count[16], first[16], off[16], next[16];
    for(int l = 0; l < 16; l++) count[l] = 0;
    for(int i = 0; i < n; i++) count[lens[i] & 15]++;
    count[0] = 0;
    const int rc = inf_code_space(count, kind, strict);
    if(rc) return rc;
    inf_code_layout(count, first, off);
    for(uint32_t lane = 0; lane < 64; lane++) inf_table_clear(tab, tb, kind, lane);
    for(int l = 0; l < 16; l++) { next[l] = off[l]; if(L && l) inf_long_store(*L, count, first, off, (uint32_t)l); }
    for(int i = 0; i < n; i++) { const uint32_t l = lens[i] & 15; if(l) { const uint32_t rank = next[l]++; inf_place_symbol(tab, symtab, tb, kind, (uint32_t)i, l, first[l] + (rank - off[l]), rank); } }
    return 0;
}
// A code longer than the table's index: its canonical value is the stream's next bits read most significant first.  x = the stream's next
// >= 15 bits (least significant first); returns the symbol and its length, or 0xffff.
MDK_HD uint32_t inf_long_code(const InfLong &L, const uint16_t *symtab, uint32_t symcap, int tb, uint32_t x, uint32_t &nbits) {
    const uint32_t c15 = inf_rev32(x) >> 17;
    for(uint32_t l = (uint32_t)tb + 1u; l < 16u; l++) {
        const uint32_t fc = L.fc[l], d = (c15 >> (15u - l)) - (fc & 0xffffu);
        if(d < (fc >> 16)) { nbits = l; const uint32_t at = L.off[l] + d; return symtab[at < symcap ? at : symcap - 1u]; }
    }
    nbits = 0;
    return 0xffffu;
}

// ---- block header (RFC 1951 3.2.3-3.2.7), in three steps with the tables built in between: the staged words must hold the whole header
// (INF_HDR_WORDS).  Results go through S: bitpos (where the next step / the block's data starts), last, in_block, stored_left, err. ----
// (1) one lane: BFINAL, BTYPE; a stored block's length; a dynamic block's counts and the code-length alphabet's lengths (S.h.cl)
MDK_HD void inf_header_open(InfShared &S, uint32_t bitpos) {
    const uint8_t CLORD[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
    InfDec d; inf_dec_seek(d, S.in, bitpos);
    d.pos = 0; d.out_len = 0; d.in_block = 0; d.last = 0; d.stored_left = 0;
    uint32_t err = INF_OK;
    d.last = inf_get<false>(d, S.in, 1);
    const uint32_t type = inf_get<false>(d, S.in, 2);
    S.h.type = type; S.stored_left = 0; S.in_block = 0;
    if(type == 3) err = INF_E_BTYPE;
    else if(type == 0) {
        inf_consume<false>(d, S.in, d.cnt & 7);                  // to the next byte boundary: every word put into bb was whole bytes, so the bits still in bb tell
        const uint32_t len = inf_get<false>(d, S.in, 16), nlen = inf_get<false>(d, S.in, 16);
        if((len ^ 0xffffu) != nlen) err = INF_E_STORED;
        S.stored_left = len; S.in_block = 2;
    } else if(type == 2) {
        const uint32_t nlit = inf_get<false>(d, S.in, 5) + 257, ndist = inf_get<false>(d, S.in, 5) + 1, ncode = inf_get<false>(d, S.in, 4) + 4;
        if(nlit > 286 || ndist > 30) err = INF_E_HEADER;
        S.h.nlit = nlit; S.h.ndist = ndist;
        for(int k = 0; k < 19; k++) S.h.cl[k] = 0;
        for(uint32_t k = 0; k < ncode; k++) S.h.cl[CLORD[k]] = (uint8_t)inf_get<false>(d, S.in, 3);
    } else { S.h.nlit = 288; S.h.ndist = 30; }
    S.bitpos = inf_dec_tell(d); S.last = d.last; S.err = err;
}
// (a fixed block's code lengths: every lane its share)
MDK_HD void inf_header_fixed_lens(InfShared &S, uint32_t lane) {
    for(uint32_t k = lane; k < 288 + 30; k += 64) S.h.lens[k] = (uint8_t)(k < 144 ? 8 : k < 256 ? 9 : k < 280 ? 7 : k < 288 ? 8 : 5);
}
// (2) one lane, a dynamic block, with the code-length alphabet's table in S.dist: the nlit + ndist code lengths into S.h.lens
MDK_HD void inf_header_lens(InfShared &S) {
    InfDec d; inf_dec_seek(d, S.in, S.bitpos);
    d.pos = 0; d.out_len = 0; d.in_block = 0; d.last = 0; d.stored_left = 0;
    int idx = 0; const int want = (int)(S.h.nlit + S.h.ndist); uint32_t err = INF_OK;
    while(idx < want) {
        const uint32_t e = S.dist[inf_peek(d) & ((1u << INF_CL_TB) - 1u)];
        if(INF_D_KIND(e) != INF_D_SYM) { err = INF_E_CODELEN; break; }
        inf_consume<false>(d, S.in, e & 15u);
        const int sym = (int)(e >> 16);
        if(sym < 16) S.h.lens[idx++] = (uint8_t)sym;
        else {
            int prev = 0, rep;
            if(sym == 16) { if(idx == 0) { err = INF_E_CODELEN; break; } prev = S.h.lens[idx - 1]; rep = 3 + (int)inf_get<false>(d, S.in, 2); }
            else if(sym == 17) rep = 3 + (int)inf_get<false>(d, S.in, 3);
            else rep = 11 + (int)inf_get<false>(d, S.in, 7);
            if(idx + rep > want) { err = INF_E_CODELEN; break; }
            while(rep--) S.h.lens[idx++] = (uint8_t)prev;
        }
    }
    if(!err && S.h.lens[256] == 0) err = INF_E_LITTABLE;          // a block without an end-of-block code never ends
    S.bitpos = inf_dec_tell(d); S.err = err;
}

// The symbol that starts at bit `bitpos` of the stream, whatever stands in front of it: kind 0 a literal (val = the byte), 1 a match (val =
// length | distance << 16), 2 end of block, 3 / 4 not a code of the literal/length / of the distance alphabet.  nbits = its bits, extra
// bits and distance code included (at most 15 + 5 + 15 + 13 = 48, of the 64 fetched).
struct InfSym { uint32_t nbits, kind, val; };
MDK_HD InfSym inf_decode_at(const InfShared &S, const uint32_t bitpos) {
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    const uint32_t w0 = S.in[w], w1 = S.in[w + 1], w2 = S.in[w + 2];
    // (the three words are asked for together; w2's share without a branch: shifted out whole when sh = 0)
    const uint64_t b = (((uint64_t)w0 | ((uint64_t)w1 << 32)) >> sh) | ((((uint64_t)w2) << 1) << (63u - sh));
    const uint32_t x = (uint32_t)b;
    uint32_t e = S.lit[x & ((1u << INF_LIT_TB) - 1u)];
    if((e & (INF_L_LEN | 0x6000u)) == INF_L_LONG) {        // a code longer than the table's index
        uint32_t nb; const uint32_t sym = inf_long_code(S.ll, S.lsym, 288u, INF_LIT_TB, x, nb);
        e = sym == 0xffffu ? INF_L_BAD : (inf_lit_entry(sym) | nb);
    }
    const uint32_t nb = e & 15u;
    InfSym s;
    if(e & INF_L_LEN) {
        const uint32_t eb = (e >> 12) & 7u, len = ((e >> 4) & 255u) + 3u + ((x >> nb) & ((1u << eb) - 1u));
        const uint32_t t1 = nb + eb;
        const uint32_t y = (uint32_t)(b >> t1);
        uint32_t f = S.dist[y & ((1u << INF_DIST_TB) - 1u)];
        if(INF_D_KIND(f) == INF_D_LONG) {
            uint32_t nb2; const uint32_t sym = inf_long_code(S.dl, S.dsym, 32u, INF_DIST_TB, y, nb2);
            f = sym == 0xffffu ? INF_D_BAD : (inf_dist_entry(sym) | nb2);
        }
        const uint32_t nb2 = f & 15u, eb2 = (f >> 8) & 15u, dist = (f >> 16) + ((y >> nb2) & ((1u << eb2) - 1u));
        s.nbits = t1 + nb2 + eb2; s.kind = INF_D_KIND(f) == INF_D_SYM ? 1u : 4u; s.val = len | (dist << 16);
    } else {
        const uint32_t k = INF_L_KIND(e);
        s.nbits = nb; s.kind = k == INF_L_LIT ? 0u : k == INF_L_EOB ? 2u : 3u; s.val = (e >> 4) & 255u;
    }
    return s;
}
// byte `i` of the stream behind the byte-aligned position `bitpos` (a stored block's bytes), out of the ring
MDK_HD uint8_t inf_ring_byte(const InfShared &S, uint32_t bitpos, uint32_t i) {
    const uint32_t a = (bitpos >> 3) + i;
    return (uint8_t)(S.in[a >> 2] >> (8u * (a & 3u)));
}
#define INF_STORED_BATCH 512u                 // bytes of a stored block copied per batch
#define INF_STORED_WORDS (INF_STORED_BATCH / 4 + 4)

// Has a finished member consumed more bits than its stream holds?  (Words past the stream read as zero, and zeros can decode: seven of
// them are the end-of-block code of a fixed block.  zlib calls that stream truncated; so do we.)
MDK_HD bool inf_overran_input(uint32_t bitpos, uint32_t skip_bytes, uint32_t in_len) {
    return (uint64_t)bitpos > 8ull * ((uint64_t)skip_bytes + in_len);
}

// ---- the parts every lane runs (bodies only; the barriers and the steps across lanes between them are the caller's) ----
#if defined(__HIPCC__)
#define INF_OR_BIT(word, bit) atomicOr(&(word), (bit))
#else
#define INF_OR_BIT(word, bit) ((word) |= (bit))
#endif
// The chain of symbols from bit `p` to the first symbol that starts at or behind `sub_end`.  end = where the chain stands when it stops, n =
// its symbols, status = why it stopped: 0 it left the stretch; INF_C_CAP it has made INF_TOK_STEPS symbols; 2 it met the end of the block
// (end = behind that code); 3 / 4 the bits at `end` are no code of the literal/length / distance alphabet.  With `write` the lane also writes
// its tokens down, token j at tok[64 * j] (`tok` = the scratch area + the lane's number): a literal's byte, or length | distance << 16.
enum { INF_C_OK = 0, INF_C_CAP = 1, INF_C_EOB = 2, INF_C_BADLIT = 3, INF_C_BADDIST = 4 };
struct InfChain { uint32_t end, n, status; };
MDK_HD InfChain inf_chain(const InfShared &S, uint32_t p, const uint32_t sub_end, const bool write, uint32_t *tok) {
    InfChain c; c.n = 0; c.status = INF_C_OK;
    while(p < sub_end) {
        if(c.n >= INF_TOK_STEPS) { c.status = INF_C_CAP; break; }
        const InfSym s = inf_decode_at(S, p);
        if(s.kind >= 2) { c.status = s.kind; if(s.kind == 2) p += s.nbits; break; }
        if(write) tok[64u * c.n] = s.val;
        c.n++; p += s.nbits;
    }
    c.end = p;
    return c;
}
MDK_HD uint32_t inf_tok_len(uint32_t t) { return (t >> 16) ? (t & 0xffffu) : 1u; }

// ---- a batch of output: bytes [beg, end) of the member.  Its per-byte state (S.o.aux, S.o.starts) is indexed from base0 = beg rounded down to
// 32, and end <= base0 + INF_BATCH_BYTES: lane l then owns the 32 bytes base0 + 32 l + j, which lie side by side in the window ring and in
// S.o.aux at 16-byte aligned places (16-byte LDS accesses), and the positions of its 32 outside [beg, end) point at themselves throughout ----
struct __attribute__((may_alias, aligned(16))) InfV4 { uint32_t w[4]; };      // (16-byte views of the byte / halfword arrays: may_alias keeps the compilers' type-based reordering off them)
// Which lane's column token g of the Huffman batch stands in (S.o.tpre = the lanes' token counts, summed), and where in it.
MDK_HD uint32_t inf_tok_column(const InfShared &S, uint32_t g) {
    uint32_t lo = 0;
#pragma unroll
    for(uint32_t step = 32; step; step >>= 1) if(S.o.tpre[lo + step] <= g) lo += step;
    return lo;
}
// One token at its place `at` (position in the member): a literal goes into the window, a match notes its distance at its first byte; both
// mark their start.  false: the match reaches in front of the member's first byte.
MDK_HD bool inf_tok_place(InfShared &S, uint32_t t, uint32_t at, uint32_t base0) {
    const uint32_t r = at - base0, dist = t >> 16;
    if(dist == 0) { S.win[inf_win_at(at)] = (uint8_t)t; S.o.aux[inf_aux_at(r)] = 0; }
    else { if(dist > at) return false; S.o.aux[inf_aux_at(r)] = (uint16_t)dist; }      // (dist <= 32768)
    INF_OR_BIT(S.o.starts[r >> 5], 1u << (r & 31u));
    return true;
}
// Matches, all bytes of the batch at once.
// (1) inf_lz_sources: q[j] = the position (in the member) byte base0 + 32 lane + j is a copy of -- itself for a literal (and outside the batch),
// its own position minus the distance for a byte of a match: the distance stands at the first byte of the token the byte belongs to, i.e. at
// the last start mark at or before it; `carry_dist` is the distance at the last start mark before the lane's bytes (the caller's scan over the
// lanes).  inf_lz_last_start: what that scan combines -- the lane's last start mark, or -1.
MDK_HD int32_t inf_lz_last_start(const InfShared &S, uint32_t lane) {
    const uint32_t w = S.o.starts[lane];
    return w ? (int32_t)(32u * lane + 31u - (uint32_t)__builtin_clz(w)) : -1;
}
// which of the lane's 32 bytes lie in the batch (bit j: byte base0 + 32 lane + j)
MDK_HD uint32_t inf_lz_inrange(uint32_t lane, uint32_t base0, uint32_t beg, uint32_t end) {
    const uint32_t p0 = base0 + 32u * lane;
    const uint32_t lo = beg > p0 ? (beg - p0 < 32u ? beg - p0 : 32u) : 0u, hi = end > p0 ? (end - p0 < 32u ? end - p0 : 32u) : 0u;
    const uint32_t below_hi = hi >= 32u ? 0xffffffffu : (1u << hi) - 1u, below_lo = lo >= 32u ? 0xffffffffu : (1u << lo) - 1u;
    return below_hi & ~below_lo;
}
MDK_HD void inf_lz_sources(const InfShared &S, uint32_t lane, uint32_t base0, uint32_t inr, uint32_t carry_dist, uint32_t *q /* [32] */) {
    uint32_t a[16];                        // the lane's 32 entries, a dword (two entries) at a time
#pragma unroll
    for(uint32_t k = 0; k < 16; k++) a[k] = *(const inf_u32a *)(const void *)&S.o.aux[inf_aux_at(32u * lane + 2u * k)];
    const uint32_t w = S.o.starts[lane], p0 = base0 + 32u * lane;
    uint32_t d = carry_dist;
#pragma unroll
    for(uint32_t j = 0; j < 32; j++) {
        const uint32_t av = (a[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
        d = ((w >> j) & 1u) ? av : d;
        const uint32_t p = p0 + j;
        q[j] = ((inr >> j) & 1u) ? p - d : (p & 0xffffu);       // (a position outside the batch may lie behind the member's 65536th byte: what S.o.aux can hold of it)
    }
}
// (2) inf_lz_publish: the sources into S.o.aux (after every lane has read the distances it needs).
MDK_HD void inf_lz_publish(InfShared &S, uint32_t lane, const uint32_t *q) {
#pragma unroll
    for(uint32_t k = 0; k < 16; k++) *(inf_u32a *)(void *)&S.o.aux[inf_aux_at(32u * lane + 2u * k)] = (q[2 * k] & 0xffffu) | (q[2 * k + 1] << 16);
}
// (3) inf_lz_jump: a source that is itself a byte of one of the batch's matches is replaced by THAT byte's source (read from S.o.aux; the
// caller publishes the new sources and repeats until nothing moves: every source is then a literal of the batch or a byte in front of it).
// A source in front of the batch reads the byte's own entry instead, which holds that source: nothing moves.
MDK_HD bool inf_lz_jump(const InfShared &S, uint32_t lane, uint32_t base0, uint32_t beg, uint32_t inr, uint32_t *q) {
    uint32_t t[32];
#pragma unroll
    for(uint32_t j = 0; j < 32; j++) t[j] = S.o.aux[inf_aux_at((((inr >> j) & 1u) && q[j] >= beg) ? q[j] - base0 : 32u * lane + j)];
    uint32_t moved = 0;
#pragma unroll
    for(uint32_t j = 0; j < 32; j++) { moved |= t[j] ^ q[j]; q[j] = t[j]; }
    return moved != 0;
}
// (4) inf_lz_gather: the bytes.  A source still in the window ring as it stands at the end of the batch (the last INF_WIN bytes up to `end`)
// is read there; an older one was written to global memory by an earlier batch (`far(position, wanted)` fetches it; every lane asks for all its
// 32 bytes at once -- `wanted` false where it needs none: no memory access then -- so that the long round trips overlap).  Every lane then writes its 32 bytes of the
// window: a literal, a byte in front of the batch and a slot behind its end get back what they held (the slot of byte p is the slot of
// p - INF_WIN, which may be somebody's source), so no order between the lanes is needed.
template <typename Far>
MDK_HD void inf_lz_gather(InfShared &S, uint32_t lane, uint32_t base0, uint32_t end, uint32_t inr, const uint32_t *q, const bool any_far, Far far) {
    uint32_t b[32];
    if(any_far) {
#pragma unroll
        for(uint32_t j = 0; j < 32; j++) b[j] = far(q[j], ((inr >> j) & 1u) && q[j] + INF_WIN < end);
    }
    uint32_t v[8];
#pragma unroll
    for(uint32_t j = 0; j < 32; j++) {
        const uint32_t nb = S.win[inf_win_at(q[j])];
        const uint32_t x = (any_far && ((inr >> j) & 1u) && q[j] + INF_WIN < end) ? b[j] : nb;
        if((j & 3u) == 0) v[j >> 2] = x; else v[j >> 2] |= x << (8u * (j & 3u));
    }
#pragma unroll
    for(uint32_t k = 0; k < 8; k++) *(inf_u32a *)(void *)&S.win[inf_win_at(base0 + 32u * lane + 4u * k)] = v[k];       // (every read above has been made: in-order issue of one wavefront)
}
#endif

// mdk_inflate.hip -- BGZF inflate and BAM record framing on the device (the step the reference pays for
// inside htslib's sam_itr_next, common.c:413).
//
//   k_inflate     one WAVEFRONT per BGZF member.  Inside a Huffman block every lane decodes the symbol that would start at its bit of the
//                 stream's next 64 (mdk_inflate_core.h inf_decode_at) and a walk over the results picks the real ones -- ~5 symbols per round
//                 of table lookups on the bench's BAM (tools/inflate_emu prints the statistics); their output positions are a prefix sum in DPP.  Batches of <= 128 match tokens / 1 KiB of output: literals
//                 go straight into a 2 KiB output window (INF_WIN) in LDS, matches become tokens.  Between batches all 64 lanes work: (1) top
//                 up the LDS ring of compressed words with one coalesced load, (2) FAR matches -- source older than the LDS window -- one lane
//                 per token, bytes from global memory (written by an earlier batch of this wavefront), (3) NEAR matches: every token whose
//                 source is final is copied at once, a short one by its own lane, a long one by the whole wavefront (a self-overlapping
//                 match doubles the copied span per round), (4) the batch's bytes leave the window for global memory, coalesced.  A 64 KiB
//                 member is ~22 k symbols; the file's ~10^4..10^5 members are what fills the machine (one lane per member, round 2's
//                 experiment, left 434 wavefronts of diverging lanes).
//   k_crc32       one wavefront per member: the CRC32 of the inflated bytes against the member's trailer -- what htslib's bgzf_read_block
//                 checks for every block the reference reads.  Coalesced 16-byte loads; every lane keeps the CRC of its own column of the
//                 member (slice-by-4 tables and a "1008 zero bytes" operator in LDS), the 64 columns are merged with GF(2) multiplications.
//   k_walk<false> one lane per member: chases the block_size words from the member's first byte (htslib never lets a record
//                 straddle two members), counts the records, notes whether the walk ends exactly at the member's end.
//   k_walk_scan   exclusive scan of the counts.
//   k_walk<true>  one lane per member walks again and writes each record's offset, and the member's digest (first/last record, extent
//                 of the read ends, coordinate order inside) -- what the host's inflate threads leave per member (csrc/host/mdk_io.c
//                 note_records).
#include "mdk_hip_internal.hpp"
#include "mdk_inflate_core.h"
#include "mdk_crc32_core.h"

// coherent byte / word loads of what this wavefront stored earlier (plain loads could hit a stale line in the CU's L1)
__device__ __forceinline__ uint32_t ld_u32_l2(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct InfParams {
    const uint8_t *comp;              // the piece's compressed bytes (device), 4-byte aligned base
    const md_inf_member *mem; int n_mem;
    uint8_t *out;                     // inflated bytes
    uint32_t *status;                 // [0] = first error: code | member << 8 (0 = none)
};

// Inclusive prefix sum over the wavefront's 64 lanes in the VALU's own data paths (DPP: shifts inside the rows of 16, then lane 15 / lane 31
// broadcast to the rows behind) -- six adds and no trip through LDS; __shfl_up compiles to ds_bpermute_b32, six dependent LDS round trips
// in the middle of every decode round.
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1 (a lane without a source adds 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}

// One member by one wavefront.
__device__ __forceinline__ void inflate_member(const InfParams &P, InfShared &S, const int m, const int lane) {
    const md_inf_member M = P.mem[m];
    if(M.out_len == 0) return;
    const uint64_t a0 = M.in_off & ~3ull;                               // aligned start of the stream's words
    const uint32_t skip = (uint32_t)(M.in_off & 3ull);
    const uint32_t n_words = (uint32_t)((M.in_off + M.in_len + 3 - a0) >> 2);
    const uint32_t *words = (const uint32_t *)(P.comp + a0);
    uint8_t *out = P.out + M.out_off;
    uint32_t filled = 0;                                               // stream words put into the ring so far
    // everything below is wave-uniform: position in the stream (bits), bytes produced, the block the decoder stands in
    uint32_t bitpos = 8u * skip, pos = 0, in_block = 0, last = 0, stored_left = 0;
    auto fail = [&](uint32_t code) { if(lane == 0) atomicCAS(P.status, 0u, code | ((uint32_t)m << 8)); };
    const unsigned long long lt = (1ull << lane) - 1ull;
    for(;;) {
        // (1) top up the ring: word w may replace word w-256 once the decoder stands behind that one
        while(filled + 64 <= (bitpos >> 5) + INF_IN_WORDS) { const uint32_t w = filled + lane; S.in[w & (INF_IN_WORDS - 1)] = w < n_words ? words[w] : 0u; filled += 64; }
        __syncthreads();
        const uint32_t beg = pos; uint32_t n_tok = 0, err = 0, fin = 0;
        if(in_block == 0) {                                            // a block header: one lane, through the bit reader; a batch of its own
            if(lane == 0) inf_header_batch(S, bitpos);
            __syncthreads();
            bitpos = S.bitpos; in_block = S.in_block; last = S.last; stored_left = S.stored_left; err = S.err;
        } else if(in_block == 2) {                                     // a stored block: bytes out of the ring, all lanes
            const uint32_t n = stored_left < INF_STORED_BATCH ? stored_left : INF_STORED_BATCH;
            for(uint32_t i = lane; i < n; i += 64) S.win[(pos + i) & (INF_WIN - 1)] = inf_ring_byte(S, bitpos, i);
            pos += n; bitpos += 8u * n; stored_left -= n;
            if(stored_left == 0) { in_block = 0; fin = last; }
        } else {
            // a Huffman block, in rounds: every lane decodes the symbol that would start at its bit of the next 64; the walk keeps the real ones
            const uint32_t lim = beg + (INF_BATCH_BYTES - 258), blim = bitpos + 32u * INF_BATCH_WORDS;
            for(;;) {
                const InfSym sy = inf_decode_at(S, bitpos + (uint32_t)lane);
                // the walk: lane 0's symbol is real, the next real one starts where it ends, ... -- a scalar loop over readlane; adv = the symbol's
                // bits, with bit 8 set where the walk ends behind this symbol (end of block) and bit 9 where it ends AT it (not a code)
                const uint32_t adv = sy.kind >= 3 ? 0x200u : sy.kind == 2 ? (sy.nbits | 0x100u) : sy.nbits;
                // (either stop bit carries the walk past lane 63, so the loop tests one thing; where it really stands is put right behind it)
                uint32_t off = 0, a = 0, lastl = 0; unsigned long long V = 0;
                do { lastl = off; V |= 1ull << off; a = (uint32_t)__builtin_amdgcn_readlane((int)adv, (int)off); off += a; } while(off < 64);
                off = lastl + (a & 0xffu);
                uint32_t stop = a >= 0x200u ? (uint32_t)__builtin_amdgcn_readlane((int)sy.kind, (int)lastl) : a >= 0x100u ? 2u : 0u;
                if(stop >= 3) V &= ~(1ull << lastl);
                bool valid = (V >> lane) & 1ull;
                const uint32_t olen = !valid ? 0u : sy.kind == 0 ? 1u : sy.kind == 1 ? (sy.val & 0xffffu) : 0u;
                const uint32_t incl = wave_incl_sum(olen);
                const uint32_t dst = pos + incl - olen;
                bool ism = valid && sy.kind == 1;
                unsigned long long mball = __ballot(ism);
                // the batch's limits: 64 match tokens, INF_BATCH_BYTES of output -- the first symbol that does not fit ends the round in front of it
                // (rarely the case: asked of the round as a whole first -- its last real symbol's end, all its matches)
                const uint32_t vend = V ? (uint32_t)__builtin_amdgcn_readlane((int)(dst + olen), 63 - __clzll((long long)V)) : pos;
                unsigned long long cm = 0;
                if(vend > beg + INF_BATCH_BYTES || n_tok + (uint32_t)__popcll(mball) > INF_MAX_TOK)
                    cm = __ballot(valid && ((ism && n_tok + (uint32_t)__popcll(mball & lt) >= INF_MAX_TOK) || dst + olen > beg + INF_BATCH_BYTES));
                if(cm) { const int c = __ffsll((long long)cm) - 1; V &= (1ull << c) - 1ull; off = (uint32_t)c; stop = 1; valid = (V >> lane) & 1ull; ism = ism && valid; mball = __ballot(ism); }
                if(__ballot(ism && (sy.val >> 16) > dst)) { err = INF_E_DIST; break; }
                if(valid && sy.kind == 0) S.win[dst & (INF_WIN - 1)] = (uint8_t)sy.val;
                if(ism) { InfToken t; t.dst = dst; t.len_dist = sy.val; S.tok[n_tok + (uint32_t)__popcll(mball & lt)] = t; }
                n_tok += (uint32_t)__popcll(mball);
                pos = cm ? (V ? (uint32_t)__builtin_amdgcn_readlane((int)(dst + olen), 63 - __clzll((long long)V)) : pos) : vend;
                bitpos += off;
                if(stop >= 3) { err = stop == 3 ? INF_E_SYMBOL : INF_E_DIST; break; }
                if(stop == 2) { in_block = 0; fin = last; break; }
                if(stop == 1 || n_tok >= INF_MAX_TOK || pos > lim || bitpos > blim) break;
            }
            __syncthreads();                                          // the window and the tokens are in LDS for every lane
        }
        if(!err && pos > M.out_len) err = INF_E_OVERRUN;
        if(!err && fin && pos != M.out_len) err = INF_E_SHORT;
        if(err || (bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, M.in_len))) { fail(err ? err : (uint32_t)INF_E_INPUT); return; }
        const uint32_t end = pos;
        // (2) far matches: one lane per token (of each 64 of them); every byte comes from global memory
        {
            bool waited = false;
            for(uint32_t tb = 0; tb < n_tok; tb += 64) {
                bool far = false;
                InfToken t; t.dst = 0; t.len_dist = 0;
                if(tb + (uint32_t)lane < n_tok) { t = S.tok[tb + (uint32_t)lane]; far = inf_tok_far(t, beg); }
                if(__ballot(far)) {
                    if(!waited) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); waited = true; }      // the earlier batches' stores have reached L2
                    if(far) {
                        // 16 bytes per round trip: the five aligned words that hold them are requested together, then cut to the source's
                        // byte offset (v_alignbyte) and stored into the window byte by byte (its position there is unaligned too)
                        const uint32_t len = t.len_dist & 0xffffu; const uint8_t *sp = out + (t.dst - (t.len_dist >> 16));
                        for(uint32_t i = 0; i < len; i += 16) {
                            const uintptr_t a = (uintptr_t)(sp + i); const uint32_t *wp = (const uint32_t *)(a & ~(uintptr_t)3); const uint32_t k = (uint32_t)(a & 3u);
                            const uint32_t n = len - i < 16u ? len - i : 16u;
                            const uint32_t w0 = ld_u32_l2(wp), w1 = ld_u32_l2(wp + 1), w2 = ld_u32_l2(wp + 2), w3 = ld_u32_l2(wp + 3), w4 = ld_u32_l2(wp + 4);
                            uint32_t b[4] = {__builtin_amdgcn_alignbyte(w1, w0, k), __builtin_amdgcn_alignbyte(w2, w1, k), __builtin_amdgcn_alignbyte(w3, w2, k), __builtin_amdgcn_alignbyte(w4, w3, k)};
#pragma unroll
                            for(uint32_t q = 0; q < 16; q++) if(q < n) S.win[(t.dst + i + q) & (INF_WIN - 1)] = (uint8_t)(b[q >> 2] >> (8 * (q & 3)));
                        }
                    }
                }
            }
            if(waited) __syncthreads();
        }
        // (3) near matches, the tokens in stream order, 64 at a time.  Everything below the first token not yet copied is final (literals were
        // written while decoding, far matches above, the 64 tokens before these), so every token whose source ends below that mark can be
        // copied at once -- short ones (most: a BAM field repeated from the record before) each by its own lane, byte by byte, which also gets a
        // match that overlaps its own output right; a long one, when it is the first, by the whole wavefront (span-doubling rounds).  A handful
        // of rounds per 64 tokens instead of one per token.
        for(uint32_t tb = 0; tb < n_tok; tb += 64) {
            InfToken t; t.dst = 0; t.len_dist = 0; bool mine = false;
            if(tb + (uint32_t)lane < n_tok) { t = S.tok[tb + (uint32_t)lane]; mine = !inf_tok_far(t, beg); }
            const uint32_t len = t.len_dist & 0xffffu, dist = t.len_dist >> 16, src = t.dst - dist;
            unsigned long long pending = __ballot(mine);
            while(pending) {
                const int f = __ffsll((long long)pending) - 1;
                const uint32_t W = (uint32_t)__builtin_amdgcn_readlane((int)t.dst, f), flen = (uint32_t)__builtin_amdgcn_readlane((int)len, f);
                if(flen > INF_NEAR_LANE_MAX) {
                    const uint32_t fdist = (uint32_t)__builtin_amdgcn_readlane((int)dist, f);
                    uint32_t done = 0, span = fdist;
                    while(done < flen) {
                        const uint32_t n = span < flen - done ? span : flen - done;
                        inf_near_round(S.win, W, fdist, done, n, (uint32_t)lane);
                        __syncthreads();
                        done += n; span <<= 1;
                    }
                    pending &= ~(1ull << f);
                    continue;
                }
                const bool ready = ((pending >> lane) & 1ull) && len <= INF_NEAR_LANE_MAX && (lane == f || src + len <= W);
                if(ready) for(uint32_t i = 0; i < len; i++) S.win[(t.dst + i) & (INF_WIN - 1)] = S.win[(src + i) & (INF_WIN - 1)];
                pending &= ~__ballot(ready);
                __syncthreads();
            }
        }
        // (4) the batch leaves the window
        {   // whole aligned words of the window as dwords (the global address is whatever the member's offset makes it), the edges as bytes
            const uint32_t p0 = (beg + 3u) & ~3u, p1 = end & ~3u;
            if(p0 <= p1) {
                for(uint32_t p = p0 + 4u * lane; p < p1; p += 256) { const uint32_t v = *(const uint32_t *)&S.win[p & (INF_WIN - 1)]; __builtin_memcpy(out + p, &v, 4); }
                if(beg + lane < p0) out[beg + lane] = S.win[(beg + lane) & (INF_WIN - 1)];
                if(p1 + lane < end) out[p1 + lane] = S.win[(p1 + lane) & (INF_WIN - 1)];
            } else for(uint32_t p = beg + lane; p < end; p += 64) out[p] = S.win[p & (INF_WIN - 1)];
        }
        if(fin) return;
    }
}
#ifndef INF_WAVES
#define INF_WAVES 6
#endif
__global__ __launch_bounds__(64, INF_WAVES) void k_inflate(const InfParams P) {
    __shared__ InfShared S;
    const int m = blockIdx.x, lane = threadIdx.x;
    if(m >= P.n_mem) return;
    inflate_member(P, S, m, lane);
}

// ---- CRC-32 of the inflated members (mdk_crc32_core.h) ----
struct CrcParams { const uint8_t *out; const md_inf_member *mem; int n_mem; const CrcConst *K; uint32_t *status; };
#define CRC_WAVES 4
__global__ __launch_bounds__(64 * CRC_WAVES) void k_crc32(const CrcParams P) {
    __shared__ uint32_t T[4][256], Z[4][256]; __shared__ uint32_t lvl[6], p8[17];
    for(int i = threadIdx.x; i < 1024; i += 64 * CRC_WAVES) { (&T[0][0])[i] = (&P.K->T[0][0])[i]; (&Z[0][0])[i] = (&P.K->Z[0][0])[i]; }
    if(threadIdx.x < 6) lvl[threadIdx.x] = P.K->lvl[threadIdx.x];
    if(threadIdx.x < 17) p8[threadIdx.x] = P.K->p8[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int m = blockIdx.x * CRC_WAVES + wave; m < P.n_mem; m += gridDim.x * CRC_WAVES) {
        const md_inf_member M = P.mem[m];
        const uint32_t L = M.out_len;
        if(L == 0) { if(lane == 0 && M.crc32 != 0u) atomicCAS(P.status, 0u, (uint32_t)INF_E_CRC | ((uint32_t)m << 8)); continue; }
        uint32_t c = crc_lane(T, Z, P.out + M.out_off, L, lane);
#pragma unroll
        for(int l = 0; l < 6; l++) { const uint32_t left = (uint32_t)__shfl((int)c, lane - (1 << l)); c ^= crc_mul(left, lvl[l]); }      // only the last lane of each group of 2^(l+1) matters
        if(lane == 63) {
            const uint32_t crc = crc_finish(c, L, p8);
            if(crc != M.crc32) atomicCAS(P.status, 0u, (uint32_t)INF_E_CRC | ((uint32_t)m << 8));
        }
    }
}

// ---- record framing ----
struct WalkParams {
    const uint8_t *out; const md_inf_member *mem; int n_mem;
    uint32_t *count;                  // [n_mem] records per member; 0xffffffff = the walk did not end at the member's end
    uint32_t *rec_off;                // record table of the piece: offset of each record's block_size word in `out`
    const uint32_t *first;            // [n_mem] exclusive scan of the counts
    md_inf_digest *dig;               // [n_mem]
};
__device__ __forceinline__ uint32_t ldu32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

template <bool WRITE>
__global__ __launch_bounds__(64) void k_walk(const WalkParams P) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if(m >= P.n_mem) return;
    const md_inf_member M = P.mem[m];
    const uint8_t *d = P.out + M.out_off; const uint32_t L = M.out_len;
    uint32_t o = 0, n = 0; bool ok = true;
    md_inf_digest g; g.n_rec = 0; g.ok = 0; g.sorted = 1; g.tid0 = g.tidN = -1; g.pos0 = g.posN = -1; g.min_endp = 0x7fffffff; g.max_endp = (int32_t)0x80000000; g.first_rec = 0;
    const uint32_t base = WRITE ? P.first[m] : 0u;
    if(WRITE && P.count[m] == 0xffffffffu) { g.first_rec = base; P.dig[m] = g; return; }
    while(o + 4 <= L) {
        const uint8_t *r = d + o + 4;
        const uint32_t bs = ldu32(d + o);
        if(bs < 32 || (uint64_t)o + 4 + bs > L) { ok = false; break; }
        const uint32_t lq = r[8], nc = (uint32_t)r[12] | ((uint32_t)r[13] << 8);
        if(32u + lq + 4u * nc > bs) { ok = false; break; }
        if(WRITE) {
            const int32_t tid = (int32_t)ldu32(r), pos = (int32_t)ldu32(r + 4);
            const uint8_t *c = r + 32 + lq; int32_t rl = 0;
            for(uint32_t k = 0; k < nc; k++) { const uint32_t v = ldu32(c + 4 * k), op = v & 15u; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(v >> 4); }
            const int32_t endp = pos + (rl > 0 ? rl : 1);
            P.rec_off[base + n] = (uint32_t)(M.out_off + o);
            if(n == 0) { g.tid0 = tid; g.pos0 = pos; }
            else if(tid < 0 || tid < g.tidN || (tid == g.tidN && pos < g.posN)) g.sorted = 0;
            if(tid < 0) g.sorted = 0;
            g.tidN = tid; g.posN = pos;
            if(endp < g.min_endp) g.min_endp = endp;
            if(endp > g.max_endp) g.max_endp = endp;
        }
        n++; o += 4 + bs;
    }
    ok = ok && o == L;
    if(!WRITE) { P.count[m] = ok ? n : 0xffffffffu; return; }
    g.n_rec = n; g.ok = ok ? 1 : 0; g.first_rec = base;
    P.dig[m] = g;
}

// exclusive scan of the per-member record counts (one workgroup; a piece has a few thousand members); a member whose walk failed
// counts as 0.  total -> status[1]
__global__ __launch_bounds__(1024) void k_walk_scan(const uint32_t *cnt, uint32_t *first, int n, uint32_t *status) {
    __shared__ uint32_t wsum[16]; __shared__ uint32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if(tid == 0) carry = 0;
    __syncthreads();
    for(int b = 0; b < n; b += 1024) {
        const int i = b + tid; uint32_t v = i < n ? cnt[i] : 0u; if(v == 0xffffffffu) v = 0;
        uint32_t incl = v;
#pragma unroll
        for(int dd = 1; dd < 64; dd <<= 1) { const uint32_t t = __shfl_up(incl, dd); if(lane >= dd) incl += t; }
        if(lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = carry; for(int w = 0; w < wave; w++) pre += wsum[w];
        if(i < n) first[i] = pre + incl - v;
        __syncthreads();
        if(tid == 1023) carry = pre + incl;
        __syncthreads();
    }
    if(tid == 0) status[1] = carry;
}

// ------------------------------------------------------------------------------------------------
// host side: pieces
// ------------------------------------------------------------------------------------------------
static void launch_inflate(int n_mem, hipStream_t st, const InfParams &IP) { hipLaunchKernelGGL(k_inflate, dim3(n_mem), dim3(64), 0, st, IP); }
struct md_piece {
    md_dev *h = nullptr; hipStream_t stream = nullptr; hipEvent_t done = nullptr;
    DBuf<uint8_t> d_comp, d_out; DBuf<md_inf_member> d_mem; DBuf<uint32_t> d_cnt, d_first, d_recoff, d_status; DBuf<md_inf_digest> d_dig;
    HBuf<md_inf_digest> h_dig; HBuf<uint32_t> h_status;
    int n_mem = 0; uint64_t out_bytes = 0, comp_bytes = 0; uint32_t n_rec_cap = 0; bool busy = false;
    bool own_stream = false, recorded = false;      // recorded: `done` has been recorded at least once (what waiting for the piece's own work means)
    bool check_crc = true;              // MDK_NO_CRC=1 leaves the check out (timing comparisons)
};
// the constants of k_crc32, once per device handle
static const CrcConst *crc_const_of(md_dev *h) {
    std::lock_guard<std::mutex> lk(h->crc_mu);
    if(!h->d_crc) {
        CrcConst *K = new CrcConst(); crc_make_const(*K);
        void *d = nullptr;
        if(hipMalloc(&d, sizeof(CrcConst)) == hipSuccess && hipMemcpy(d, K, sizeof(CrcConst), hipMemcpyHostToDevice) == hipSuccess) h->d_crc = d; else { if(d) (void)hipFree(d); (void)hipGetLastError(); }
        delete K;
    }
    return (const CrcConst *)h->d_crc;
}

static void launch_crc(md_dev *h, md_piece *p, hipStream_t st) {
    CrcParams C; C.out = p->d_out.p; C.mem = p->d_mem.p; C.n_mem = p->n_mem; C.K = (const CrcConst *)h->d_crc; C.status = p->d_status.p;
    int grid = (p->n_mem + CRC_WAVES - 1) / CRC_WAVES; if(grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_crc32, dim3(grid), dim3(64 * CRC_WAVES), 0, st, C);
}
void inflate_kernels_warm() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void *)k_inflate); (void)hipFuncGetAttributes(&fa, (const void *)k_crc32);
    (void)hipFuncGetAttributes(&fa, (const void *)k_walk<false>); (void)hipFuncGetAttributes(&fa, (const void *)k_walk<true>); (void)hipFuncGetAttributes(&fa, (const void *)k_walk_scan);
    (void)hipGetLastError();
}
// A piece's work is queued on one of a FEW streams the pieces of a handle share (creating a stream costs the runtime 5-9 ms, one after the other:
// eight teams that each made their own at the same moment kept the first device piece -- which the reader needs in file order -- 40 ms late,
// gpurun_out r04q).  Four: each piece is ~3,400 wavefronts and the device holds ~6,000, so pieces on four streams fill it, and copies of one
// overlap kernels of another.  MDK_PIECE_STREAMS=0: a stream per piece.
static hipStream_t piece_stream_of(md_dev *h, bool *own) {
    static const int want = getenv("MDK_PIECE_STREAMS") ? atoi(getenv("MDK_PIECE_STREAMS")) : 4;
    *own = false;
    if(want < 1) { hipStream_t s = nullptr; if(hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; *own = true; return s; }
    std::lock_guard<std::mutex> lk(h->piece_mu);
    if((int)h->piece_streams.size() < want) { hipStream_t s = mdk_stream_take(h->device); if(!s) return nullptr; h->piece_streams.push_back(s); return s; }
    return h->piece_streams[(size_t)(h->piece_rr++ % want)];
}
static hipError_t piece_sync(md_piece *p) { return p->recorded ? hipEventSynchronize(p->done) : hipSuccess; }      // the piece's own work, not its stream's
extern "C" int md_piece_create(md_dev *h, md_piece **out) {
    if(!h || !out) return fail(MDK_ERR_ARG, "md_piece_create", hipSuccess);
    *out = nullptr;
    HIPCHK(hipSetDevice(h->device));
    md_piece *p = new md_piece(); p->h = h;
    if(!(p->stream = piece_stream_of(h, &p->own_stream)) || hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess) { if(p->own_stream && p->stream) (void)hipStreamDestroy(p->stream); delete p; return fail(MDK_ERR_HIP, "md_piece_create: stream", hipGetLastError()); }
    if(p->d_status.need(4) || p->h_status.need(4)) { delete p; return MDK_ERR_NOMEM; }
    p->check_crc = !getenv("MDK_NO_CRC");
    if(p->check_crc && !crc_const_of(h)) { delete p; return fail(MDK_ERR_NOMEM, "md_piece_create: CRC tables", hipSuccess); }
    *out = p;
    return 0;
}
extern "C" void md_piece_destroy(md_piece *p) {
    if(!p) return;
    (void)hipSetDevice(p->h->device);
    (void)piece_sync(p);
    if(p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    if(p->done) (void)hipEventDestroy(p->done);
    p->d_comp.release(); p->d_out.release(); p->d_mem.release(); p->d_cnt.release(); p->d_first.release(); p->d_recoff.release(); p->d_status.release(); p->d_dig.release();
    p->h_dig.release(); p->h_status.release();
    delete p;
}

// H2D of the compressed bytes and the member table, inflate, record framing, D2H of the digests: all queued on the piece's
// stream; md_piece_wait returns when it is all done.  comp should be pinned memory (md_host_alloc) for the copy to be a DMA.
extern "C" int md_piece_submit(md_piece *p, const uint8_t *comp, uint64_t comp_bytes, const md_inf_member *mem, int n_mem) {
    if(!p || !comp || !mem || n_mem < 1) return fail(MDK_ERR_ARG, "md_piece_submit", hipSuccess);
    ProfScope pf(PF_PIECE_SUBMIT);
    md_dev *h = p->h;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(piece_sync(p));
    uint64_t out_bytes = 0;
    for(int i = 0; i < n_mem; i++) {
        if(mem[i].out_off != out_bytes || mem[i].out_len > 65536u || mem[i].in_off + mem[i].in_len > comp_bytes) return fail(MDK_ERR_ARG, "md_piece_submit: member table", hipSuccess);
        out_bytes += mem[i].out_len;
    }
    if(out_bytes >= (1ull << 32) - 65536) return fail(MDK_ERR_ARG, "md_piece_submit: more than 4 GiB inflated in one piece", hipSuccess);
    const uint32_t rec_cap = (uint32_t)(out_bytes / 36 + 16);          // a BAM record is at least 36 bytes with its block_size word
    if(p->d_comp.need((size_t)comp_bytes + 1024) || p->d_out.need((size_t)out_bytes + 1024) || p->d_mem.need((size_t)n_mem) || p->d_cnt.need((size_t)n_mem) || p->d_first.need((size_t)n_mem) ||
       p->d_dig.need((size_t)n_mem) || p->h_dig.need((size_t)n_mem) || p->d_recoff.need((size_t)rec_cap)) return MDK_ERR_NOMEM;
    p->n_mem = n_mem; p->out_bytes = out_bytes; p->comp_bytes = comp_bytes; p->n_rec_cap = rec_cap;
    hipStream_t st = p->stream;
    host_block_ensure_registered(comp);
    HIPCHK(hipMemcpyAsync(p->d_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->d_mem.p, mem, sizeof(md_inf_member) * (size_t)n_mem, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(p->d_status.p, 0, 16, st));
    InfParams IP; IP.comp = p->d_comp.p; IP.mem = p->d_mem.p; IP.n_mem = n_mem; IP.out = p->d_out.p; IP.status = p->d_status.p;
    launch_inflate(n_mem, st, IP);
    if(p->check_crc) launch_crc(h, p, st);
    WalkParams W; W.out = p->d_out.p; W.mem = p->d_mem.p; W.n_mem = n_mem; W.count = p->d_cnt.p; W.rec_off = p->d_recoff.p; W.first = p->d_first.p; W.dig = p->d_dig.p;
    hipLaunchKernelGGL(k_walk<false>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
    hipLaunchKernelGGL(k_walk_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *)p->d_cnt.p, p->d_first.p, n_mem, p->d_status.p);
    hipLaunchKernelGGL(k_walk<true>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(p->h_dig.p, p->d_dig.p, sizeof(md_inf_digest) * (size_t)n_mem, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(p->h_status.p, p->d_status.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->done, st));
    p->busy = true; p->recorded = true;
    return 0;
}

extern "C" int md_piece_wait(md_piece *p, md_piece_info *info) {
    if(!p || !info || !p->busy) return fail(MDK_ERR_ARG, "md_piece_wait: nothing submitted", hipSuccess);
    ProfScope pf(PF_PIECE_WAIT);
    HIPCHK(hipSetDevice(p->h->device));
    HIPCHK(hipEventSynchronize(p->done));
    p->busy = false;
    const uint32_t st = p->h_status.p[0];
    if((st & 255u) == (uint32_t)INF_E_CRC) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "a BGZF member fails its CRC32 check (corrupt file): member %u of the piece", st >> 8); return MDK_ERR_ARG; }
    if(st) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "BGZF inflate failed on the device (corrupt file?): error %u in member %u of the piece", st & 255u, st >> 8); return MDK_ERR_ARG; }
    info->n_mem = p->n_mem; info->digest = p->h_dig.p; info->n_records = p->h_status.p[1]; info->out_bytes = p->out_bytes;
    info->d_out = p->d_out.p; info->d_rec_off = p->d_recoff.p;
    return 0;
}

// the inflated bytes (or a part of them) back on the host: tests, and files whose records straddle members
extern "C" int md_piece_read(md_piece *p, uint64_t off, uint64_t bytes, uint8_t *dst) {
    if(!p || !dst || off + bytes > p->out_bytes) return fail(MDK_ERR_ARG, "md_piece_read", hipSuccess);
    HIPCHK(hipSetDevice(p->h->device));
    HIPCHK(piece_sync(p));
    if(bytes) HIPCHK(hipMemcpy(dst, p->d_out.p + off, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int md_piece_read_records(md_piece *p, uint32_t first, uint32_t n, uint32_t *dst) {
    if(!p || !dst || (uint64_t)first + n > p->n_rec_cap) return fail(MDK_ERR_ARG, "md_piece_read_records", hipSuccess);
    HIPCHK(hipSetDevice(p->h->device));
    HIPCHK(piece_sync(p));
    if(n) HIPCHK(hipMemcpy(dst, p->d_recoff.p + first, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// the kernels alone on resident input, timed with HIP events on the piece's stream (bench.py / tools)
extern "C" int md_piece_bench_crc(md_piece *p, int iters, float *ms_crc) {
    if(!p || iters < 1 || !p->n_mem || !ms_crc) return fail(MDK_ERR_ARG, "md_piece_bench_crc", hipSuccess);
    HIPCHK(hipSetDevice(p->h->device));
    if(!crc_const_of(p->h)) return fail(MDK_ERR_NOMEM, "md_piece_bench_crc: CRC tables", hipSuccess);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipEventRecord(e0, p->stream));
    for(int i = 0; i < iters; i++) launch_crc(p->h, p, p->stream);
    HIPCHK(hipEventRecord(e1, p->stream));
    HIPCHK(hipEventSynchronize(e1));
    float a = 0; HIPCHK(hipEventElapsedTime(&a, e0, e1));
    *ms_crc = a / (float)iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint32_t st = 0; HIPCHK(hipMemcpy(&st, p->d_status.p, 4, hipMemcpyDeviceToHost));
    if(st) { snprintf(mdk_err_buf(), MDK_ERR_BYTES, "CRC32 / inflate status %u in member %u", st & 255u, st >> 8); return MDK_ERR_ARG; }
    return 0;
}
extern "C" int md_piece_bench(md_piece *p, int iters, float *ms_inflate, float *ms_walk) {
    if(!p || iters < 1 || !p->n_mem) return fail(MDK_ERR_ARG, "md_piece_bench", hipSuccess);
    HIPCHK(hipSetDevice(p->h->device));
    hipEvent_t e0, e1, e2; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
    hipStream_t st = p->stream; const int n_mem = p->n_mem;
    InfParams IP; IP.comp = p->d_comp.p; IP.mem = p->d_mem.p; IP.n_mem = n_mem; IP.out = p->d_out.p; IP.status = p->d_status.p;
    WalkParams W; W.out = p->d_out.p; W.mem = p->d_mem.p; W.n_mem = n_mem; W.count = p->d_cnt.p; W.rec_off = p->d_recoff.p; W.first = p->d_first.p; W.dig = p->d_dig.p;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventRecord(e0, st));
    for(int i = 0; i < iters; i++) launch_inflate(n_mem, st, IP);
    HIPCHK(hipEventRecord(e1, st));
    for(int i = 0; i < iters; i++) {
        hipLaunchKernelGGL(k_walk<false>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
        hipLaunchKernelGGL(k_walk_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *)p->d_cnt.p, p->d_first.p, n_mem, p->d_status.p);
        hipLaunchKernelGGL(k_walk<true>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
    }
    HIPCHK(hipEventRecord(e2, st));
    HIPCHK(hipEventSynchronize(e2));
    float a = 0, b = 0; HIPCHK(hipEventElapsedTime(&a, e0, e1)); HIPCHK(hipEventElapsedTime(&b, e1, e2));
    if(ms_inflate) *ms_inflate = a / (float)iters;
    if(ms_walk) *ms_walk = b / (float)iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return 0;
}

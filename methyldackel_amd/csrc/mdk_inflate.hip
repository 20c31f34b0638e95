// mdk_inflate.hip -- BGZF inflate and BAM record framing on the device (the step the reference pays for
// inside htslib's sam_itr_next, common.c:413).
//
//   k_inflate     one WAVEFRONT per BGZF member at a time; the wavefronts of a launch draw members from a counter until none is left.  A Huffman
//                 block is decoded 64 stretches of the stream at once (mdk_inflate_core.h: every lane runs a chain of symbols through its
//                 stretch, first from a guessed start, then from where its left neighbour's chain ended; two passes settle 19 of 20 batches) and
//                 leaves TOKENS in a scratch area of the wavefront in global memory, token j of the 64 lanes side by side (one coalesced store per
//                 step).  The tokens become bytes in batches of <= 2 KiB through a 4 KiB output window in LDS: a prefix sum (DPP) places 64 tokens
//                 at a time, literals go into the window, and all bytes of the batch's matches are resolved together -- source pointers per byte,
//                 pointer jumping through the batch's own matches, one gather (a source older than the window comes from global memory, written
//                 there by an earlier batch of the same wavefront) --, then the batch leaves the window as coalesced dword stores.  A 64 KiB member
//                 of a BAM file is ~11 k symbols in ~530 chain steps of the wavefront (20 symbols per step) and ~33 batches of output.
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

#ifndef INF_EXP
#define INF_EXP 0
#endif
struct InfParams {
    const uint8_t *comp;              // the piece's compressed bytes (device), 4-byte aligned base
    const md_inf_member *mem; int n_mem;
    uint8_t *out;                     // inflated bytes
    uint32_t *status;                 // [0] = first error: code | member << 8 (0 = none); [2] = the counter the wavefronts draw members from
    uint32_t *tok;                    // token scratch: INF_TOK_WORDS words per wavefront of the launch
#ifdef INF_PROFILE
    unsigned long long *prof;         // [16] cycles per phase, summed over the wavefronts (experiment builds only)
#endif
};
#ifdef INF_PROFILE
#define PROF_T(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_acc[k] += now_ - prof_t; prof_t = now_; } while(0)
enum { PR_STAGE = 0, PR_HEADER, PR_STORED, PR_CHAIN1, PR_CHAIN2, PR_CHAINX, PR_CONFIRM, PR_TOKFETCH, PR_TOKPLACE, PR_SOURCES, PR_JUMP, PR_GATHER, PR_FLUSH, PR_OTHER, PR_N };
#else
#define PROF_T(k) do { } while(0)
#endif

// Inclusive prefix sum over the wavefront's 64 lanes in the VALU's own data paths (DPP: shifts inside the rows of 16, then lane 15 / lane 31
// broadcast to the rows behind) -- six adds and no trip through LDS.
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1 (a lane without a source adds 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}
// the same with max over signed values (a lane without a source keeps its own)
__device__ __forceinline__ int wave_incl_max(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x143, 0xc, 0xf, false));
    return x;
}
__device__ __forceinline__ uint32_t leading_ones(unsigned long long m) { return ~m ? (uint32_t)(__ffsll((long long)~m) - 1) : 64u; }      // lanes 0 .. n-1 all set

// A decode table by the whole wavefront (mdk_inflate_core.h: the pieces): lane i looks after symbols i, i + 64, ...; how many symbols have
// each length and which rank a symbol has among those of its length are ballots; every symbol then fills its own entries.  0, or
// inf_code_space's verdict.  Ends with a barrier: the table is complete for every lane.
template <typename T, int CHUNKS>
__device__ __forceinline__ int build_table(const uint8_t *lens, const int n, const int tb, T *tab, uint16_t *symtab, InfLong *L, const int kind, const int strict, const int lane) {
    uint32_t len_c[CHUNKS], count[16], first[16], off[16];
#pragma unroll
    for(int c = 0; c < CHUNKS; c++) { const int i = 64 * c + lane; len_c[c] = i < n ? (uint32_t)(lens[i] & 15) : 0u; }
    count[0] = 0;
#pragma unroll
    for(int l = 1; l < 16; l++) {
        uint32_t k = 0;
#pragma unroll
        for(int c = 0; c < CHUNKS; c++) k += (uint32_t)__popcll(__ballot(len_c[c] == (uint32_t)l));
        count[l] = k;
    }
    const int rc = inf_code_space(count, kind, strict);
    if(rc) return rc;
    inf_code_layout(count, first, off);
    inf_table_clear(tab, tb, kind, (uint32_t)lane);
    if(L && lane == 0) {
#pragma unroll
        for(int l = 1; l < 16; l++) inf_long_store(*L, count, first, off, (uint32_t)l);
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for(int c = 0; c < CHUNKS; c++) {
        uint32_t rank = 0, code = 0;
#pragma unroll
        for(int l = 1; l < 16; l++) {
            const unsigned long long mk = __ballot(len_c[c] == (uint32_t)l);
            if(len_c[c] == (uint32_t)l) { const uint32_t before = (uint32_t)__popcll(mk & lt); rank = off[l] + before; code = first[l] + before; }
            // (the next chunk's symbols of this length go on from where this chunk's end, in rank and in code)
            const uint32_t seen = (uint32_t)__popcll(mk);
            off[l] += seen; first[l] += seen;
        }
        if(len_c[c]) inf_place_symbol(tab, symtab, tb, kind, (uint32_t)(64 * c + lane), len_c[c], code, rank);
    }
    __syncthreads();
    return 0;
}

// One member by one wavefront.
__device__ __forceinline__ void inflate_member(const InfParams &P, InfShared &S, uint32_t *tok, const int m, const int lane) {
    const md_inf_member M = P.mem[m];
    if(M.out_len == 0) return;
    const uint64_t a0 = M.in_off & ~3ull;                               // aligned start of the stream's words
    const uint32_t skip = (uint32_t)(M.in_off & 3ull);
    const uint32_t n_words = (uint32_t)((M.in_off + M.in_len + 3 - a0) >> 2);
    const uint32_t *words = (const uint32_t *)(P.comp + a0);
    uint8_t *out = P.out + M.out_off;
    // the member's output as a buffer: a lane that wants no byte of it asks behind its end, which costs no memory access
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)M.out_len, 0x00020000);
    // everything below is wave-uniform: position in the stream (bits), bytes produced, the block the decoder stands in
    uint32_t bitpos = 8u * skip, pos = 0, in_block = 0, last = 0, stored_left = 0;
    auto fail = [&](uint32_t code) { if(lane == 0) atomicCAS(P.status, 0u, code | ((uint32_t)m << 8)); };
#ifdef INF_PROFILE
    unsigned long long prof_acc[PR_N] = {0}, prof_t = __builtin_amdgcn_s_memtime();
    struct ProfOut { const InfParams &P; unsigned long long *acc; int lane; __device__ ~ProfOut() { if(lane == 0) for(int k = 0; k < PR_N; k++) atomicAdd(P.prof + k, acc[k]); } } prof_out{P, prof_acc, lane};
#endif
    auto stage = [&](uint32_t wbase, uint32_t count) {                  // words wbase .. of the stream into S.in, coalesced; words behind the stream read as zero
        __syncthreads();
        for(uint32_t i = (uint32_t)lane; i < count; i += 64) { const uint32_t w = wbase + i; S.in[i] = w < n_words ? __builtin_nontemporal_load(words + w) : 0u; }
        __syncthreads();
    };
    auto flush = [&](uint32_t beg, uint32_t end) {                      // whole aligned words of the window as dwords (the global address is whatever the member's offset makes it), the edges as bytes
        const uint32_t p0 = (beg + 3u) & ~3u, p1 = end & ~3u;
        if(p0 <= p1) {
            for(uint32_t p = p0 + 4u * lane; p < p1; p += 256) { const uint32_t v = *(const inf_u32a *)&S.win[inf_win_at(p)]; __builtin_memcpy(out + p, &v, 4); }
            if(beg + lane < p0) out[beg + lane] = S.win[inf_win_at(beg + lane)];
            if(p1 + lane < end) out[p1 + lane] = S.win[inf_win_at(p1 + lane)];
        } else for(uint32_t p = beg + lane; p < end; p += 64) out[p] = S.win[inf_win_at(p)];
    };
    for(;;) {
        const uint32_t wbase = bitpos >> 5, rel = bitpos & 31u;
        uint32_t fin = 0;
        if(in_block == 0) {                                            // a block header: one lane reads it, all lanes build its tables
            PROF_T(PR_OTHER); stage(wbase, INF_HDR_WORDS); PROF_T(PR_STAGE);
            if(lane == 0) inf_header_open(S, rel);
            __syncthreads();
            uint32_t err = S.err; const uint32_t type = S.h.type;
            if(!err && type == 1) { inf_header_fixed_lens(S, (uint32_t)lane); __syncthreads(); }
            if(!err && type == 2) {
                if(build_table<inf_dist_t, 1>(S.h.cl, 19, INF_CL_TB, S.dist, nullptr, nullptr, 2, 1, lane)) err = INF_E_CODELEN;
                if(!err) { if(lane == 0) inf_header_lens(S); __syncthreads(); err = S.err; }
            }
            if(!err && type != 0) {
                const int nlit = (int)S.h.nlit, ndist = (int)S.h.ndist;
                if(build_table<inf_dist_t, 1>(S.h.lens + nlit, ndist, INF_DIST_TB, S.dist, S.dsym, &S.dl, 1, type == 2, lane)) err = INF_E_DISTTABLE;
                else if(build_table<inf_lit_t, 5>(S.h.lens, nlit, INF_LIT_TB, S.lit, S.lsym, &S.ll, 0, 1, lane)) err = INF_E_LITTABLE;
                in_block = 1;
            } else in_block = S.in_block;
            bitpos = 32u * wbase + S.bitpos; last = S.last; stored_left = S.stored_left;
            PROF_T(PR_HEADER);
            if(err || (bitpos >> 5) > n_words) { fail(err ? err : (uint32_t)INF_E_INPUT); return; }
            continue;
        }
        if(in_block == 2) {                                            // a stored block: bytes out of the staged words, all lanes
            stage(wbase, INF_STORED_WORDS);
            const uint32_t n = stored_left < INF_STORED_BATCH ? stored_left : INF_STORED_BATCH, beg = pos;
            if(pos + n > M.out_len) { fail(INF_E_OVERRUN); return; }
            for(uint32_t i = lane; i < n; i += 64) S.win[inf_win_at(pos + i)] = inf_ring_byte(S, rel, i);
            pos += n; bitpos += 8u * n; stored_left -= n;
            if(stored_left == 0) { in_block = 0; fin = last; }
            if(fin && pos != M.out_len) { fail(INF_E_SHORT); return; }
            if((bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, M.in_len))) { fail(INF_E_INPUT); return; }
            __syncthreads();
            flush(beg, pos); PROF_T(PR_STORED);
            if(fin) return;
            continue;
        }
        // ---- a Huffman batch: 64 stretches of sw words, chains until a prefix of the lanes agrees ----
        const uint32_t left_words = n_words > wbase ? n_words - wbase : 1u;
        uint32_t sw = (left_words + 63u) / 64u; sw = sw < INF_SW_MIN ? INF_SW_MIN : sw > INF_SW_MAX ? INF_SW_MAX : sw; sw |= 1u;
        const uint32_t sub = 32u * sw;
        PROF_T(PR_OTHER); stage(wbase, 64u * sw + 8u); PROF_T(PR_STAGE);
        const uint32_t stream_end = 32u * (n_words - wbase) + 64u;      // no true chain gets this far without an error
        uint32_t start = rel + sub * (uint32_t)lane; const uint32_t sub_end = start + sub;
        const bool active = lane == 0 || start < stream_end;
        InfChain c; c.end = start; c.n = 0; c.status = INF_C_BADLIT;
        bool need = active, written = false; uint32_t passes = 0, nconf = 1, lstat = 0;
        for(;;) {
            if(need) c = inf_chain(S, start, sub_end, passes != 0, tok + lane);
            PROF_T(passes == 0 ? PR_CHAIN1 : passes == 1 ? PR_CHAIN2 : PR_CHAINX);
            if(passes) written = written || need;
            passes++;
            // who goes again: a lane whose left neighbour's chain ended elsewhere than where it started, or whose tokens are not written down yet (the first pass writes nothing)
            const uint32_t pend = (uint32_t)__shfl_up((int)c.end, 1), pst = (uint32_t)__shfl_up((int)c.status, 1);
            need = lane == 0 ? !written : (active && pst == INF_C_OK && (start != pend || !written));
            if(need && lane) start = pend;
            if(passes == 1) continue;
            nconf = leading_ones(__ballot(lane == 0 || (active && pst == INF_C_OK && !need)));
            lstat = (uint32_t)__builtin_amdgcn_readlane((int)c.status, (int)nconf - 1);
            if(lstat != INF_C_OK || passes >= INF_MAX_PASSES || !__ballot(need)) break;        // (nobody needs to go again = all active lanes agree)
        }
        if(lstat == INF_C_BADLIT || lstat == INF_C_BADDIST) { fail(lstat == INF_C_BADLIT ? (uint32_t)INF_E_SYMBOL : (uint32_t)INF_E_DIST); return; }
        const uint32_t cn = (uint32_t)lane < nconf ? c.n : 0u, tincl = wave_incl_sum(cn);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)tincl, 63), lend = (uint32_t)__builtin_amdgcn_readlane((int)c.end, (int)nconf - 1);
        bitpos = 32u * wbase + lend;
        if(lstat == INF_C_EOB) { in_block = 0; fin = last; }
        if((bitpos >> 5) > n_words || (fin && inf_overran_input(bitpos, skip, M.in_len))) { fail(INF_E_INPUT); return; }
        __syncthreads();                                               // the staged words are done with: their memory now holds the output batches' state
        S.o.tpre[lane] = tincl - cn; if(lane < 2) S.o.tpre[64 + lane] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // the tokens are read back by other lanes than wrote them
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads(); PROF_T(PR_CONFIRM);
        // ---- the tokens into bytes, a batch of output at a time ----
        uint32_t g = 0;
        while(g < total) {
            const uint32_t beg = pos, base0 = beg & ~31u;
            S.o.starts[lane] = 0;                                        // (INF_BATCH_BYTES / 32 = 64 words)
            __syncthreads();
            bool full = false;
            while(g < total && !full) {
                const uint32_t gi = g + (uint32_t)lane; const bool has = gi < total;
                uint32_t t = 0, len = 0;
                if(has) { const uint32_t col = inf_tok_column(S, gi); t = tok[64u * (gi - S.o.tpre[col]) + col]; len = inf_tok_len(t); }
                const uint32_t incl = wave_incl_sum(len), at = pos + incl - len; PROF_T(PR_TOKFETCH);
                const uint32_t take = leading_ones(__ballot(has && at + len <= base0 + INF_BATCH_BYTES));
                if(take == 0) { full = true; break; }                    // (a token is at most 258 bytes: the first of a batch always fits)
                bool ok = true;
                if((uint32_t)lane < take) ok = inf_tok_place(S, t, at, base0);
                if(__ballot(!ok)) { fail(INF_E_DIST); return; }
                pos = (uint32_t)__builtin_amdgcn_readlane((int)(at + len), (int)take - 1); g += take;
                if(take < 64 && g < total) full = true;
                if(pos > M.out_len) { fail(INF_E_OVERRUN); return; }
                PROF_T(PR_TOKPLACE);
            }
            __syncthreads();
            const uint32_t end = pos;
            // matches: every byte's source, pointer jumping through the batch's own matches, one gather
            {
                const int lst = inf_lz_last_start(S, (uint32_t)lane);
                const int run = __shfl_up(wave_incl_max(lst), 1);        // the last start mark in front of the lane's bytes
                const uint32_t carry = (lane > 0 && run >= 0) ? S.o.aux[inf_aux_at((uint32_t)run)] : 0u;
                const uint32_t inr = inf_lz_inrange((uint32_t)lane, base0, beg, end);
                uint32_t q[32];
                inf_lz_sources(S, (uint32_t)lane, base0, inr, carry, q);
                __syncthreads();
                inf_lz_publish(S, (uint32_t)lane, q);
                __syncthreads(); PROF_T(PR_SOURCES);
                for(;;) {
                    const bool moved = inf_lz_jump(S, (uint32_t)lane, base0, beg, inr, q);
                    if(!__ballot(moved)) break;
                    __syncthreads();
                    inf_lz_publish(S, (uint32_t)lane, q);
                    __syncthreads();
                }
                PROF_T(PR_JUMP);
                bool far = false;
#pragma unroll
                for(uint32_t j = 0; j < 32; j++) far = far || (((inr >> j) & 1u) && q[j] + INF_WIN < end);
#if INF_EXP == 1
                const bool any_far = false;
#else
                const bool any_far = __ballot(far) != 0;
#endif
#if INF_EXP != 3
                if(any_far) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the earlier batches' stores have reached L2
#endif
                inf_lz_gather(S, (uint32_t)lane, base0, end, inr, q, any_far, [&](uint32_t p, bool wanted) -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(out_rsrc, wanted ? p : 0xffffffffu, 0, INF_EXP == 2 ? 0 : 16 /* sc1: past the CU's L1 */); });
                __syncthreads(); PROF_T(PR_GATHER);
            }
            flush(beg, end); PROF_T(PR_FLUSH);
        }
        if(fin && pos != M.out_len) { fail(INF_E_SHORT); return; }
        if(fin) return;
    }
}
#ifndef INF_WAVES
#define INF_WAVES 3
#endif
__global__ __launch_bounds__(64, INF_WAVES) void k_inflate(const InfParams P) {
    __shared__ InfShared S;
    const int lane = threadIdx.x;
    uint32_t *tok = P.tok + (size_t)blockIdx.x * INF_TOK_WORDS;
    for(;;) {
        int m = 0;
        if(lane == 0) m = (int)atomicAdd(P.status + 2, 1u);
        m = __builtin_amdgcn_readfirstlane(m);
        if(m >= P.n_mem) return;
        inflate_member(P, S, tok, m, lane);
        __syncthreads();
    }
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
// The wavefronts of a launch: as many as the device holds at once (what the kernel's LDS and registers allow per CU, times the CUs), or the
// members if they are fewer; each draws members from the counter in status[2] until none is left, so a launch has no second, partly filled
// generation of wavefronts and the last ones end within one member's time of each other.  MDK_INF_GRID overrides.
// MDK_PIECE_CU_RESERVE=n (experiment): the pieces' streams leave n of the device's CUs alone (hipExtStreamCreateWithCUMask), so that the kernels of the
// chunks -- microseconds of work that otherwise waits for a slot until a whole launch of k_inflate has ended -- always find CUs
static int piece_cu_reserve() { static const int r = getenv("MDK_PIECE_CU_RESERVE") ? atoi(getenv("MDK_PIECE_CU_RESERVE")) : 0; return r > 0 && r < 200 ? r : 0; }
static hipStream_t piece_stream_make(int device) {
    hipStream_t s = nullptr; const int r = piece_cu_reserve();
    if(r > 0) {
        int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        uint32_t mask[16]; const int words = (cus + 31) / 32; for(int w = 0; w < 16; w++) mask[w] = 0;
        for(int i = 0; i < cus; i++) mask[i >> 5] |= 1u << (i & 31);
        for(int k = 0; k < r; k++) { const int i = (int)(((long long)k * cus) / r) + (cus / r) - 1; if(i >= 0 && i < cus) mask[i >> 5] &= ~(1u << (i & 31)); }      // evenly spread
        if(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) == hipSuccess) return s;
        (void)hipGetLastError();
    }
    return mdk_piece_stream_take(device);
}
static int inflate_grid_max(int device) {
    static int cached[64] = {0};
    const int slot = device >= 0 && device < 64 ? device : 0;
    if(!cached[slot]) {
        int per_cu = 0, cus = 0;
        if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_inflate, 64, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        if(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
        (void)hipGetLastError();
        int g = per_cu * (cus - piece_cu_reserve());
        if(getenv("MDK_INF_GRID") && atoi(getenv("MDK_INF_GRID")) > 0) g = atoi(getenv("MDK_INF_GRID"));
        cached[slot] = g;
    }
    return cached[slot];
}
static inline int inflate_grid(int device, int n_mem) { const int g = inflate_grid_max(device); return n_mem < g ? n_mem : g; }
static hipError_t launch_inflate(int device, int n_mem, hipStream_t st, const InfParams &IP, bool counter_is_zero = false) {
    if(!counter_is_zero) { hipError_t e = hipMemsetAsync(IP.status + 2, 0, 4, st); if(e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_inflate, dim3(inflate_grid(device, n_mem)), dim3(64), 0, st, IP);
    return hipGetLastError();
}
struct md_piece {
    md_dev *h = nullptr; hipStream_t stream = nullptr; hipEvent_t done = nullptr, ev_in = nullptr, ev_inf = nullptr;
    DBuf<uint8_t> d_comp, d_out; DBuf<md_inf_member> d_mem; DBuf<uint32_t> d_cnt, d_first, d_recoff, d_status, d_tok; DBuf<md_inf_digest> d_dig;
    HBuf<md_inf_digest> h_dig; HBuf<uint32_t> h_status; HBuf<md_inf_member> h_mem;      // h_mem: the caller's member table, pinned (copied from ordinary memory the call would wait for every copy queued before it)
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
    if((int)h->piece_streams.size() < want) { hipStream_t s = piece_stream_make(h->device); if(!s) return nullptr; h->piece_streams.push_back(s); return s; }
    return h->piece_streams[(size_t)(h->piece_rr++ % want)];
}
// The pieces' LANES.  A piece's work used to sit on one of four streams from its copy to its digests, so up to four k_inflate ran at a time: each
// took 4.3-4.8 ms instead of the 2.9 it takes alone, and -- its wavefronts draw members until none is left -- gave its LDS back in 12.9 KB
// slots that the next launch's wavefronts took at once: a workgroup of k_prep_scan (76 KB) or k_crc32 found room only when all queued pieces
// had run dry (k_prep_scan 0.9 ms instead of 0.18, k_crc32 1.4 instead of 0.13, the device idle for 10-16 ms at a time while three groups of
// chunks waited for it: profiles/r06pc_*).  Now every handle has ONE stream the compressed bytes of all pieces cross the link on, in the order
// submitted and at the link's full rate each, and ONE stream all k_inflate run on, one after the other: a launch has the device to itself,
// and where its last wavefronts end everybody else's small kernels find the CUs empty.  CRC, record walk and the digests' way back stay on the
// piece's own (shared) stream, ordered by events.  MDK_PIECE_LANES=0: everything on the piece's stream as before.
static bool piece_lanes_wanted() { static const bool on = !(getenv("MDK_PIECE_LANES") && atoi(getenv("MDK_PIECE_LANES")) == 0); return on; }
static bool piece_lanes_of(md_dev *h, hipStream_t *in, hipStream_t *inf) {
    if(!piece_lanes_wanted()) return false;
    std::lock_guard<std::mutex> lk(h->piece_mu);
    if(!h->piece_in) h->piece_in = piece_stream_make(h->device);
    if(!h->piece_inf) h->piece_inf = piece_stream_make(h->device);
    if(!h->piece_in || !h->piece_inf) return false;
    *in = h->piece_in; *inf = h->piece_inf;
    return true;
}
static hipError_t piece_sync(md_piece *p) { return p->recorded ? hipEventSynchronize(p->done) : hipSuccess; }      // the piece's own work, not its stream's
extern "C" int md_piece_members_per_round(md_dev *h) { if(!h) return 0; if(hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return 0; } return inflate_grid_max(h->device); }
extern "C" int md_piece_create(md_dev *h, md_piece **out) {
    if(!h || !out) return fail(MDK_ERR_ARG, "md_piece_create", hipSuccess);
    *out = nullptr;
    HIPCHK(hipSetDevice(h->device));
    md_piece *p = new md_piece(); p->h = h;
    if(!(p->stream = piece_stream_of(h, &p->own_stream)) || hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->ev_inf, hipEventDisableTiming) != hipSuccess) { if(p->own_stream && p->stream) (void)hipStreamDestroy(p->stream); delete p; return fail(MDK_ERR_HIP, "md_piece_create: stream", hipGetLastError()); }
    if(p->d_status.need(4) || p->h_status.need(8)) { delete p; return MDK_ERR_NOMEM; }      // (h_status: four words back from the device, four zero words on their way to it)
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
    if(p->ev_in) (void)hipEventDestroy(p->ev_in);
    if(p->ev_inf) (void)hipEventDestroy(p->ev_inf);
    p->d_comp.release(); p->d_out.release(); p->d_mem.release(); p->d_cnt.release(); p->d_first.release(); p->d_recoff.release(); p->d_status.release(); p->d_tok.release(); p->d_dig.release();
    p->h_dig.release(); p->h_status.release(); p->h_mem.release();
    delete p;
}

// H2D of the compressed bytes and the member table, inflate, record framing, D2H of the digests: all queued on the piece's
// stream; md_piece_wait returns when it is all done.  comp should be pinned memory (md_host_alloc) for the copy to be a DMA.
extern "C" int md_piece_submit(md_piece *p, const uint8_t *comp, uint64_t comp_bytes, const md_inf_member *mem, int n_mem) {
    if(!p || !comp || !mem || n_mem < 1) return fail(MDK_ERR_ARG, "md_piece_submit", hipSuccess);
    ProfScope pf(PF_PIECE_SUBMIT); MarkScope mk_sub("md_piece_submit");
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
       p->d_dig.need((size_t)n_mem) || p->h_dig.need((size_t)n_mem) || p->h_mem.need((size_t)n_mem) || p->d_recoff.need((size_t)rec_cap) || p->d_tok.need((size_t)inflate_grid(h->device, n_mem) * INF_TOK_WORDS)) return MDK_ERR_NOMEM;
    p->n_mem = n_mem; p->out_bytes = out_bytes; p->comp_bytes = comp_bytes; p->n_rec_cap = rec_cap;
    const bool prof = mdk_prof_on(); double tq[8]; int nq = 0; auto tick = [&]() { if(prof && nq < 8) tq[nq++] = mdk_now(); };
    tick();
    hipStream_t st = p->stream, s_in = st, s_inf = st;
    const bool lanes = piece_lanes_of(h, &s_in, &s_inf);
    host_block_ensure_registered(comp);
    tick();
    // error word, record count, the launch's member counter: zeroed by a 16-byte copy from pinned memory, ahead of the compressed bytes.  (A hipMemsetAsync is a
    // kernel: behind the piece's 96 MB on the copy stream it waited 0.44 ms on average for a CU the running k_inflate had -- with the inflate stream waiting for it.)
    p->h_status.p[4] = p->h_status.p[5] = p->h_status.p[6] = p->h_status.p[7] = 0;
    HIPCHK(hipMemcpyAsync(p->d_status.p, p->h_status.p + 4, 16, hipMemcpyHostToDevice, s_in));
    memcpy(p->h_mem.p, mem, sizeof(md_inf_member) * (size_t)n_mem);
    HIPCHK(hipMemcpyAsync(p->d_mem.p, p->h_mem.p, sizeof(md_inf_member) * (size_t)n_mem, hipMemcpyHostToDevice, s_in));
    HIPCHK(hipMemcpyAsync(p->d_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, s_in));
    tick();
    if(lanes) { HIPCHK(hipEventRecord(p->ev_in, s_in)); HIPCHK(hipStreamWaitEvent(s_inf, p->ev_in, 0)); }
    InfParams IP; IP.comp = p->d_comp.p; IP.mem = p->d_mem.p; IP.n_mem = n_mem; IP.out = p->d_out.p; IP.status = p->d_status.p; IP.tok = p->d_tok.p;
#ifdef INF_PROFILE
    { static unsigned long long *dp = nullptr; if(!dp) { (void)hipMalloc((void **)&dp, 16 * 8); (void)hipMemset(dp, 0, 16 * 8); } IP.prof = dp; }
#endif
    HIPCHK(launch_inflate(h->device, n_mem, s_inf, IP, true));
    if(lanes) { HIPCHK(hipEventRecord(p->ev_inf, s_inf)); HIPCHK(hipStreamWaitEvent(st, p->ev_inf, 0)); }
    tick();
    if(p->check_crc) launch_crc(h, p, st);
    WalkParams W; W.out = p->d_out.p; W.mem = p->d_mem.p; W.n_mem = n_mem; W.count = p->d_cnt.p; W.rec_off = p->d_recoff.p; W.first = p->d_first.p; W.dig = p->d_dig.p;
    hipLaunchKernelGGL(k_walk<false>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
    hipLaunchKernelGGL(k_walk_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *)p->d_cnt.p, p->d_first.p, n_mem, p->d_status.p);
    hipLaunchKernelGGL(k_walk<true>, dim3((n_mem + 63) / 64), dim3(64), 0, st, W);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(p->h_dig.p, p->d_dig.p, sizeof(md_inf_digest) * (size_t)n_mem, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(p->h_status.p, p->d_status.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->done, st));
    tick();
    if(prof && nq == 5 && tq[4] - tq[0] > 0.003) { char w[200]; snprintf(w, sizeof(w), "slow md_piece_submit: %.1f ms (registration %.1f, copies in %.1f, inflate launch %.1f, the rest %.1f)", (tq[4] - tq[0]) * 1e3, (tq[1] - tq[0]) * 1e3, (tq[2] - tq[1]) * 1e3, (tq[3] - tq[2]) * 1e3, (tq[4] - tq[3]) * 1e3); mdk_marks_dump(w); }
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
    InfParams IP; IP.comp = p->d_comp.p; IP.mem = p->d_mem.p; IP.n_mem = n_mem; IP.out = p->d_out.p; IP.status = p->d_status.p; IP.tok = p->d_tok.p;
#ifdef INF_PROFILE
    unsigned long long *dprof = nullptr; HIPCHK(hipMalloc((void **)&dprof, 16 * 8)); HIPCHK(hipMemset(dprof, 0, 16 * 8)); IP.prof = dprof;
#endif
    WalkParams W; W.out = p->d_out.p; W.mem = p->d_mem.p; W.n_mem = n_mem; W.count = p->d_cnt.p; W.rec_off = p->d_recoff.p; W.first = p->d_first.p; W.dig = p->d_dig.p;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventRecord(e0, st));
    for(int i = 0; i < iters; i++) HIPCHK(launch_inflate(p->h->device, n_mem, st, IP));
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
#ifdef INF_PROFILE
    {   unsigned long long hp[16]; HIPCHK(hipMemcpy(hp, dprof, sizeof hp, hipMemcpyDeviceToHost)); unsigned long long tot = 0; for(int k = 0; k < PR_N; k++) tot += hp[k];
        static const char *nm[PR_N] = {"stage", "header", "stored", "chain_pass1", "chain_pass2", "chain_more", "confirm", "tok_fetch", "tok_place", "lz_sources", "lz_jump", "lz_gather", "flush", "other"};
        fprintf(stderr, "[inf profile] %d members x %d launches; share of the wavefronts' time per phase:", n_mem, iters);
        for(int k = 0; k < PR_N; k++) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)hp[k] / (double)(tot ? tot : 1));
        fprintf(stderr, "; memtime ticks per member %.0f\n", (double)tot / ((double)n_mem * iters)); (void)hipFree(dprof); }
#endif
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return 0;
}

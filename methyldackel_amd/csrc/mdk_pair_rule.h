/* mdk_pair_rule.h -- which read a read is overlap-resolved against: the pending/pairing state machine of the overlap callbacks
 * (overlaps.c:121-147: custom_overlap_constructor / custom_overlap_destructor over a khash keyed by read name) together with htslib's
 * pileup buffer, whose eviction decides when the destructor runs.  For the admitted reads of ONE name in file order:
 *   - a read enters the buffer iff its end lies beyond the column about to be emitted, i.e. beyond the start of the read admitted just
 *     before it (the very first admitted read of the chunk: iff it ends after position 0, or the contig is not the first);
 *   - entering, it first drops the name's reads that have been swept out (end < that column); ANY such drop erases the name's pending
 *     entry (the destructor deletes the key whichever read it is called for);
 *   - then, if it is a paired read with a mapped mate (flag & 1, !(flag & 12)): no pending entry -> it becomes pending; otherwise it is
 *     paired with the pending read and the entry is erased.
 * Host-compilable on purpose: tests/test_pair_rule.py builds it with g++ and checks the closed form for two reads (what k_prep_segs uses
 * for nearly every name) against the step-by-step machine over every combination of a small domain. */
#ifndef MDK_PAIR_RULE_H
#define MDK_PAIR_RULE_H
#include <stdint.h>
#ifdef __HIPCC__
#define MDK_PR __host__ __device__ __forceinline__
#else
#define MDK_PR static inline
#endif
#define MDK_MAXLIVE 8                /* reads of one name alive in the pileup buffer at once; more: the host prepares the chunk */

MDK_PR bool mdk_pairs(uint32_t flag) { return (flag & 0x1) && !(flag & 12); }

struct MdkPairState { int32_t pending, mate; int32_t live[MDK_MAXLIVE]; int nlive; bool second, overflow; };
MDK_PR void mdk_pair_init(MdkPairState &S) { S.pending = -1; S.mate = -1; S.nlive = 0; S.second = false; S.overflow = false; }
/* one read of the name: x its index in the chunk (any numbering that follows file order), (flag, rend) its own, first: no read was admitted
 * before it, else prev_pos is the start of the read admitted just before it; a: the read whose mate is asked for; contig: the chunk's
 * contig index */
MDK_PR void mdk_pair_step(MdkPairState &S, int32_t contig, uint32_t a, int32_t x, uint32_t flag, int32_t rend, bool first, int32_t prev_pos) {
    const bool inserted = first ? (contig > 0 || rend > 0) : (rend > prev_pos);
    if(!inserted) return;
    bool evicted = false; int w = 0;
    for(int q = 0; q < S.nlive; q++) { if(!first && S.live[q] < prev_pos) evicted = true; else S.live[w++] = S.live[q]; }
    S.nlive = w;
    if(evicted) S.pending = -1;
    if(mdk_pairs(flag)) {
        if(S.pending < 0) S.pending = x;
        else {
            if((uint32_t)S.pending == a) { S.mate = x; S.second = false; }
            else if((uint32_t)x == a) { S.mate = S.pending; S.second = true; }
            S.pending = -1;
        }
    }
    if(S.nlive == MDK_MAXLIVE) { S.overflow = true; return; }
    S.live[S.nlive++] = rend;
}
/* the same for a name with exactly two admitted reads f < s (file order), one of which is `a` (and mdk_pairs(a's flag) holds): f enters
 * and becomes pending; s enters, sweeps f out if f ends before the start of the read admitted before s -- which erases the pending
 * entry --, and otherwise is paired with it.  Returns the index of the read `a` is resolved against, or -1. */
MDK_PR int32_t mdk_pair_two(int32_t contig, uint32_t a, int32_t f, int32_t s, uint32_t flag_f, int32_t rend_f, bool first_f, int32_t prev_f,
                            uint32_t flag_s, int32_t rend_s, int32_t prev_s, bool &second) {
    const bool in_f = first_f ? (contig > 0 || rend_f > 0) : rend_f > prev_f;
    const bool in_s = rend_s > prev_s;
    second = false;
    if(!in_f || !in_s || !mdk_pairs(flag_f) || !mdk_pairs(flag_s) || rend_f < prev_s) return -1;
    second = (uint32_t)s == a;
    return (uint32_t)f == a ? s : f;
}
#endif

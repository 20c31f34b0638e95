/* mdk_overlap_rule.h -- the mate-overlap quality rule (overlaps.c:90-109) for ONE matched pair of bases, as the kernels
 * evaluate it.  Host-compilable on purpose: tests/test_overlap_rule.py builds it with g++ and compares the two forms over
 * every input (2 x 16 x 16 x 256 x 256 cases). */
#ifndef MDK_OVERLAP_RULE_H
#define MDK_OVERLAP_RULE_H
#ifdef __HIPCC__
#define MDK_HD __host__ __device__ __forceinline__
#else
#define MDK_HD static inline
#endif

/* (uint8_t)(q + 0.2*q) as evaluated by the reference on x86-64 (overlaps.c:103,106): floor(6q/5) mod 256.
 * md_dev_open checks this identity against the C expression for all 256 values. */
/* (for 0 <= q <= 255: x/5 == (x * 52429) >> 18 for every x < 2^16, and both products fit 24-bit multiplies, which the GPU issues at
 * full rate where a 32-bit multiply and a multiply-high take four cycles each) */
MDK_HD int boost(int q) { return (int)(((((unsigned)q & 255u) * 6u) * 52429u) >> 18) & 255; }

/* cust_tweak_overlap_quality, literally: a = the read earlier in the file, b = the later one; returns what the OWN base's
 * quality becomes (b, ql: own base and quality; mb, mq: the partner's) */
MDK_HD int resolve_overlap(bool ownIsSecond, int b, int ql, int mb, int mq) {
    int ba = ownIsSecond ? mb : b, qa = ownIsSecond ? mq : ql, bb = ownIsSecond ? b : mb, qb = ownIsSecond ? ql : mq;
    if(ba != bb) {
        if(qa > qb && ba != 15) { qa -= qb; qb = 0; }
        else if(qb > qa && bb != 15) { qb -= qa; qa = 0; }
        else { qa = 0; qb = 0; }
    } else {
        if(qa > qb) { qa = boost(qa); qb = 0; } else { qb = boost(qb); qa = 0; }
    }
    return ownIsSecond ? qb : qa;
}

/* the same function as three selects (x, bo: own quality and base; y, bm: the partner's): bases differ -> x-y if x > y and
 * the own base is not N, else 0; bases agree -> the better quality is boosted and the other zeroed, a tie going to the
 * later read of the pair */
MDK_HD int resolve_own(bool ownIsSecond, int bo, int x, int bm, int y) {
    const int diff = (x > y && bo != 15) ? x - y : 0;
    const int same = (x + (ownIsSecond ? 1 : 0) > y) ? boost(x) : 0;
    return bo == bm ? same : diff;
}
#endif

"""TEST INFRASTRUCTURE: a slow pure-Python evaluation of one device batch (include/mdk_hip.h md_read_batch).

It restates, for SMALL inputs only, what the HIP kernel computes from a packed batch of segments, so that the host
side of the product (admission, strand, pairing, CIGAR expansion, packing: csrc/host/mdk_extract.c) can be checked
against the oracle on a box without a GPU.  It is never imported by the product."""
import ctypes as C


def _window(cfg, strand, read2, lq):
    if strand < 1:
        return 0, lq
    b = 4 * (strand - 1) + (2 if read2 else 0)
    lb, rb = cfg.bounds[b], cfg.bounds[b + 1]
    alb, arb = cfg.absoluteBounds[b], cfg.absoluteBounds[b + 1]
    lb, alb, arb = min(lb, lq), min(alb, lq), min(arb, lq)
    lo = max(lb, alb)
    hi = rb if (rb and rb < lq) else lq
    hi = min(hi, lq - arb)
    return lo, hi


class Payload:
    """seq/qual of one read (blob + 4*off4) with its trimming window"""
    _cache = {}

    def __init__(self, batch, off4, lq, strand, read2, cfg):
        base = C.addressof(batch.blob.contents) + 4 * off4
        seqb = (lq + 1) // 2
        seq = bytes((C.c_uint8 * seqb).from_address(base)) if seqb else b""
        self.qual = bytes((C.c_uint8 * lq).from_address(base + ((seqb + 3) & ~3))) if lq else b""
        self.bases = [(seq[q >> 1] >> (0 if q & 1 else 4)) & 15 for q in range(lq)]
        self.lo, self.hi = _window(cfg, strand, read2, lq)

    def bq(self, q):
        if q < self.lo or q >= self.hi:
            return 15, 0
        return self.bases[q], self.qual[q]


def boost(q):
    return ((q * 6) // 5) & 255


def resolve(second, b, ql, mb, mq):
    ba, qa, bb, qb = (mb, mq, b, ql) if second else (b, ql, mb, mq)
    if ba != bb:
        if qa > qb and ba != 15:
            qa, qb = qa - qb, 0
        elif qb > qa and bb != 15:
            qa, qb = 0, qb - qa
        else:
            qa = qb = 0
    else:
        if qa > qb:
            qa, qb = boost(qa), 0
        else:
            qa, qb = 0, boost(qb)
    return qb if second else qa


def context(ref, p, keep):
    L = len(ref)
    c = ref[p] & 0x5F
    if c == 0x43:
        isg = 0
        t = 0 if (p + 1 < L and ref[p + 1] & 0x5F == 0x47) else 1 if (p + 2 < L and ref[p + 2] & 0x5F == 0x47) else 2
    elif c == 0x47:
        isg = 1
        t = 0 if (p > 0 and ref[p - 1] & 0x5F == 0x43) else 1 if (p > 1 and ref[p - 2] & 0x5F == 0x43) else 2
    else:
        return None
    if not keep[t]:
        return None
    return t, isg


def region_of(runs, p):
    """strand code of the run holding p, or None (runs: sorted disjoint (start, end, strand))"""
    import bisect
    i = bisect.bisect_right(runs, (p, 1 << 40, 9)) - 1
    if i >= 0 and runs[i][0] <= p < runs[i][1]:
        return runs[i][2]
    return None


def eval_batch(batch, ref: bytes, cfg, runs=None):
    """-> {pos: (type, isG, nmeth, nunmeth, noff, nvar)} for positions with any evidence.
    runs: the -l restriction of this contig (Plan.regions), or None"""
    keep = (cfg.keepCpG, cfg.keepCHG, cfg.keepCHH)
    out = {}
    pay = {}

    def payload(off4, lq, strand, read2):
        k = (off4, lq, strand, read2)
        if k not in pay:
            pay[k] = Payload(batch, off4, lq, strand, read2, cfg)
        return pay[k]

    prev = None
    for i in range(batch.n_segs):
        g = batch.seg[i]
        strand, read2, second, partner = g.sf & 7, bool(g.sf & 8), bool(g.sf & 16), bool(g.sf & 32)
        assert g.len >= 1 and g.q0 + g.len <= g.l_qseq
        o = payload(g.off4, g.l_qseq, strand, read2)
        m = payload(g.m_off4, g.m_l_qseq, g.msf & 7, bool(g.msf & 8)) if partner else None
        if partner:
            assert g.m_q0 + g.len <= g.m_l_qseq
        odd = strand & 1
        for j in range(g.len):
            p = g.rpos + j
            if p < batch.beg or p >= batch.end or p >= len(ref):
                continue
            ctx = context(ref, p, keep)
            if ctx is None:
                continue
            t, isg = ctx
            if runs is not None:
                rs = region_of(runs, p)
                if rs is None or (rs == 1 and strand not in (1, 3)) or (rs == 2 and strand not in (2, 4)):
                    continue
            b, ql = o.bq(g.q0 + j)
            if m is not None:
                mb, mq = m.bq(g.m_q0 + j)
                ql = resolve(second, b, ql, mb, mq)
            e = out.setdefault(p, [t, isg, 0, 0, 0, 0])
            if bool(odd) != bool(isg):
                assert strand != 0, "strand 0 read reached a call"
                if ql >= cfg.minPhred:
                    if odd:
                        e[2] += b == 2
                        e[3] += b == 8
                    else:
                        e[2] += b == 4
                        e[3] += b == 1
            elif cfg.minOppositeDepth > 0:
                if ql >= cfg.minPhred:
                    e[4] += 1
                    e[5] += (b not in (4, 15)) if odd else (b not in (2, 15))
    return {p: tuple(v) for p, v in out.items() if v[2] + v[3] > 0 or v[4] > 0}


def mbias_context(ref, p, beg, end, keep):
    """context as mbias sees it: classified inside the chunk's own window [beg, min(end, len-1)] (MBias.c:147,172-180)"""
    lo, hi = beg, min(end, len(ref) - 1)          # inclusive window
    c = ref[p] & 0x5F
    if c == 0x43:
        isg = 0
        t = 0 if (p + 1 <= hi and ref[p + 1] & 0x5F == 0x47) else 1 if (p + 2 <= hi and ref[p + 2] & 0x5F == 0x47) else 2
    elif c == 0x47:
        isg = 1
        t = 0 if (p - 1 >= lo and ref[p - 1] & 0x5F == 0x43) else 1 if (p - 2 >= lo and ref[p - 2] & 0x5F == 0x43) else 2
    else:
        return None
    return (t, isg) if keep[t] else None


def eval_mbias(batch, ref: bytes, cfg, runs=None, hist=None):
    """adds the batch's calls to hist {(strand, read#, q): [meth, unmeth]} (no mate-overlap handling: MBias.c:158-161)"""
    keep = (cfg.keepCpG, cfg.keepCHG, cfg.keepCHH)
    hist = {} if hist is None else hist
    pay = {}
    for i in range(batch.n_segs):
        g = batch.seg[i]
        strand, read2, partner = g.sf & 7, bool(g.sf & 8), bool(g.sf & 32)
        assert not partner, "mbias batches must be built without pairing"
        k = (g.off4, g.l_qseq, strand, read2)
        if k not in pay:
            pay[k] = Payload(batch, g.off4, g.l_qseq, strand, read2, cfg)
        o = pay[k]
        odd = strand & 1
        for j in range(g.len):
            p = g.rpos + j
            if p < batch.beg or p >= batch.end or p >= len(ref):
                continue
            ctx = mbias_context(ref, p, batch.beg, batch.end, keep)
            if ctx is None or bool(odd) == bool(ctx[1]):
                continue
            if runs is not None:
                rs = region_of(runs, p)
                if rs is None or (rs == 1 and strand not in (1, 3)) or (rs == 2 and strand not in (2, 4)):
                    continue
            assert strand != 0
            q = g.q0 + j
            b, ql = o.bq(q)
            if ql < cfg.minPhred:
                continue
            un = {2: 0, 8: 1}.get(b) if odd else {4: 0, 1: 1}.get(b)
            if un is None:
                continue
            hist.setdefault((strand, 2 if read2 else 1, q), [0, 0])[un] += 1
    return hist


CIGAR_TYPE = [3, 1, 2, 2, 1, 0, 0, 3, 3, 0, 0, 0, 0, 0, 0, 0]      # htslib bam_cigar_type for MIDNSHP=XB


def eval_perread(pr, ref: bytes, min_phred):
    """[(nmeth, nunmeth)] for the reads of a perRead batch (md_pr_batch): processRead of the reference (perRead.c:38-94),
    with the chunk's reference window [max(beg-2,0), min(end+10000, len-1)] (perRead.c:176)"""
    wend = min(pr.end + 10000, len(ref) - 1)
    out = []
    for i in range(pr.n_reads):
        r = pr.read[i]
        base = C.addressof(pr.blob.contents) + 4 * r.off4
        lq = r.l_qseq
        seqb = (lq + 1) // 2
        seq = bytes((C.c_uint8 * max(seqb, 1)).from_address(base))
        qual = bytes((C.c_uint8 * max(lq, 1)).from_address(base + ((seqb + 3) & ~3)))
        cig = [pr.cigar[r.cig_off + k] for k in range(r.n_cigar)]
        odd = r.strand & 1
        rp, mp, k, off, nm, nu = 0, r.pos, 0, 0, 0, 0
        while rp < lq and k < len(cig):
            if off >= cig[k] >> 4:
                off = 0
                k += 1
            if k >= len(cig):
                break
            t = CIGAR_TYPE[cig[k] & 15]
            if t & 2:
                if t & 1:
                    if qual[rp] < min_phred:
                        mp += 1; rp += 1; off += 1
                    d = 0
                    if mp <= wend:
                        c = ref[mp] & 0x5F
                        if c == 0x43 and mp + 1 <= wend and ref[mp + 1] & 0x5F == 0x47:
                            d = 1
                        elif c == 0x47 and mp >= 1 and ref[mp - 1] & 0x5F == 0x43:
                            d = -1
                    if d:
                        if rp < lq:
                            b = (seq[rp >> 1] >> (0 if rp & 1 else 4)) & 15
                        elif lq & 1:
                            b = seq[rp >> 1] & 15
                        else:
                            b = (qual[0] >> 4) & 15
                        if d == 1 and odd:
                            nm += b == 2; nu += b == 8
                        elif d == -1 and not odd:
                            nm += b == 4; nu += b == 1
                    mp += 1; rp += 1; off += 1
                else:
                    mp += cig[k] >> 4; k += 1; off = 0
            elif t & 1:
                rp += cig[k] >> 4; k += 1; off = 0
            else:
                off = 0; k += 1
        out.append((nm, nu))
    return out

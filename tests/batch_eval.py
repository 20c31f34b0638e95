"""TEST INFRASTRUCTURE: a slow pure-Python evaluation of one device batch (include/mdk_hip.h md_read_batch).

It restates, for SMALL inputs only, what the HIP kernel computes from a packed batch, so that the host side of the
product (admission, strand, pairing, packing: csrc/host/mdk_extract.c) can be checked against the oracle on a box
without a GPU.  It is never imported by the product."""
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


class Read:
    def __init__(self, batch, i, cfg):
        h = batch.hdr[i]
        self.pos, self.lq, self.ncig, self.strand, self.flags = h.pos, h.l_qseq, h.n_cigar, h.strand, h.flags
        base = C.addressof(batch.blob.contents) + 4 * h.off4
        self.cig = list((C.c_uint32 * self.ncig).from_address(base)) if self.ncig else []
        seqb = (self.lq + 1) // 2
        seq = bytes((C.c_uint8 * seqb).from_address(base + 4 * self.ncig)) if seqb else b""
        qoff = base + 4 * self.ncig + ((seqb + 3) & ~3)
        self.qual = bytes((C.c_uint8 * self.lq).from_address(qoff)) if self.lq else b""
        self.bases = [(seq[q >> 1] >> (0 if q & 1 else 4)) & 15 for q in range(self.lq)]
        self.lo, self.hi = _window(cfg, self.strand, self.flags & 1, self.lq)
        # reference position -> query index for M/=/X bases
        self.at = {}
        x, y = self.pos, 0
        for c in self.cig:
            op, ln = c & 15, c >> 4
            if op in (0, 7, 8):
                for j in range(ln):
                    if y + j < self.lq:
                        self.at[x + j] = y + j
                x += ln
                y += ln
            elif op in (1, 4):
                y += ln
            elif op in (2, 3):
                x += ln

    def bq(self, q):
        if q < self.lo or q >= self.hi:
            return 15, 0
        return self.bases[q], self.qual[q]


def boost(q):
    return ((q * 6) // 5) & 255


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


def eval_batch(batch, ref: bytes, cfg):
    """-> {pos: (type, isG, nmeth, nunmeth, noff, nvar)} for positions with any evidence"""
    keep = (cfg.keepCpG, cfg.keepCHG, cfg.keepCHH)
    reads = [Read(batch, i, cfg) for i in range(batch.n_reads)]
    out = {}
    for i, o in enumerate(reads):
        mi = batch.mate[i]
        m = reads[mi] if mi >= 0 and ((o.strand - reads[mi].strand) & 1) == 0 else None
        for p, q in o.at.items():
            if p < batch.beg or p >= batch.end or p >= len(ref):
                continue
            ctx = context(ref, p, keep)
            if ctx is None:
                continue
            t, isg = ctx
            odd = o.strand & 1
            b, ql = o.bq(q)
            if m is not None and p in m.at:
                mb, mq = m.bq(m.at[p])
                second = bool(o.flags & 2)
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
                ql = qb if second else qa
            e = out.setdefault(p, [t, isg, 0, 0, 0, 0])
            if bool(odd) != bool(isg):
                assert o.strand != 0, "strand 0 read reached a call"
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

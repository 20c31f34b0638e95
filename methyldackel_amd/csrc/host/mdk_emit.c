/* mdk_emit.c -- text post-pass of `extract` (extract.c:39-99,182-222,443-510) and the ordered emitter (see mdk_plan.h). */
#include "mdk_plan.h"

/* ------------------------------------------------------------------------------------------------ */
/* text post-pass (extract.c:443-510 driving writeCall/processLast, extract.c:39-99,207-222)          */
/* ------------------------------------------------------------------------------------------------ */
/* decimal digits of v at q, as printf's %u / %i would write them; returns the end */
static inline char *put_u32(char *q, uint32_t v) { char t[10]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while(v); while(n) *q++ = t[--n]; return q; }
static inline char *put_i32(char *q, int32_t v) { if(v < 0) { *q++ = '-'; return put_u32(q, (uint32_t)(-(int64_t)v)); } return put_u32(q, (uint32_t)v); }

static void put_site(mdk_plan *p, sbuf *dst, const char *chrom, int32_t pos, int width, uint32_t m, uint32_t u, int ref_is_c, const char *cctx, const char *tri) {
    const opts_t *o = &p->o; char line[10000]; int n; uint32_t cov = m + u;      /* the size of writeCall's buffer (extract.c:40): lines longer than that are cut the same way */
    if(cov < (uint32_t)o->min_depth && !o->cytosine_report) return;
    /* the all-integer formats are written digit by digit straight into the chunk's text (a dense-context run prints some
     * 10^7 lines per 32 Mb); anything with a %f, or a line that could reach writeCall's 10000 bytes, goes through snprintf */
    if(!o->fraction && !o->logit && !o->methylkit) {
        const size_t cl = strlen(chrom);
        if(cl < 9000) {
            char *q0 = sb_room(dst, cl + 128), *q = q0;
            memcpy(q, chrom, cl); q += cl; *q++ = '\t';
            if(o->counts) { q = put_i32(q, pos); *q++ = '\t'; q = put_i32(q, pos + width); *q++ = '\t'; q = put_i32(q, (int32_t)cov); }
            else if(o->cytosine_report) {
                q = put_i32(q, pos + 1); *q++ = '\t'; *q++ = ref_is_c ? '+' : '-'; *q++ = '\t'; q = put_u32(q, m); *q++ = '\t'; q = put_u32(q, u); *q++ = '\t'; *q++ = 'C';
                { size_t k = strlen(cctx); memcpy(q, cctx, k); q += k; } *q++ = '\t'; { size_t k = strlen(tri); memcpy(q, tri, k); q += k; }
            } else {
                q = put_i32(q, pos); *q++ = '\t'; q = put_i32(q, pos + width); *q++ = '\t'; q = put_i32(q, (int)(100.0 * ((double)m) / cov)); *q++ = '\t';
                q = put_u32(q, m); *q++ = '\t'; q = put_u32(q, u);
            }
            *q++ = '\n'; *q = 0; dst->l += (size_t)(q - q0);
            return;
        }
    }
    if(o->fraction) n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, ((double)m) / cov);
    else if(o->counts) n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%i\n", chrom, pos, pos + width, cov);
    else if(o->logit) { double f = ((double)m) / cov; n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%f\n", chrom, pos, pos + width, log(f) - log(1 - f)); }
    else if(o->methylkit) n = snprintf(line, sizeof(line), "%s.%i\t%s\t%i\t%c\t%i\t%6.2f\t%6.2f\n", chrom, pos + 1, chrom, pos + 1, ref_is_c ? 'F' : 'R', cov, 100.0 * ((double)m) / cov, 100.0 * ((double)u) / cov);
    else if(o->cytosine_report) n = snprintf(line, sizeof(line), "%s\t%i\t%c\t%" PRIu32 "\t%" PRIu32 "\tC%s\t%s\n", chrom, pos + 1, ref_is_c ? '+' : '-', m, u, cctx, tri);
    else n = snprintf(line, sizeof(line), "%s\t%i\t%i\t%i\t%" PRIu32 "\t%" PRIu32 "\n", chrom, pos, pos + width, (int)(100.0 * ((double)m) / cov), m, u);
    if(n > 0) sb_put(dst, line, (size_t)n < sizeof(line) ? (size_t)n : sizeof(line) - 1);
}

/* trinucleotide context string of a C (direction +1) or G (direction -1) at contig index i (extract.c:120-180) */
static const char *trinuc(const char *seq, int64_t len, int64_t i, int dir, char out[4]) {
    static const char comp[256] = {['A'] = 'T', ['a'] = 'T', ['C'] = 'G', ['c'] = 'G', ['G'] = 'C', ['g'] = 'C', ['T'] = 'A', ['t'] = 'A'};
    int k;
    out[0] = 'C'; out[3] = 0;
    for(k = 1; k <= 2; k++) {
        int64_t j = i + (int64_t)k * dir; char ch = 'N';
        if(j >= 0 && j < len) { ch = seq[j]; if(dir < 0) ch = comp[(uint8_t)ch] ? comp[(uint8_t)ch] : 'N'; else { ch &= 0x5f; if(ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T') ch = 'N'; } }
        out[k] = ch;
    }
    return out;
}
static const char *cctx_name(int type) { return type == 0 ? "G" : type == 1 ? "HG" : "HH"; }

/* zero-coverage rows of --cytosine_report between *from and upto (extract.c:182-205) */
static void put_blanks(mdk_plan *p, sbuf *dst, const char *chrom, const char *seq, int64_t len, int64_t *from, int64_t upto) {
    char tri[4];
    for(; *from < upto; (*from)++) {
        int code; int dir, type;
        if(*from >= len) continue;
        code = ctx_code(seq, len, *from);
        if(!code) continue;
        type = code - 1;
        if(!p->o.ctx_on[type]) continue;
        dir = ((seq[*from] & 0x5f) == 'C') ? 1 : -1;
        put_site(p, dst, chrom, (int32_t)*from, 1, 0, 0, dir > 0, cctx_name(type), trinuc(seq, len, *from, dir, tri));
    }
}

/* text of one chunk (variant filter, --mergeContext, formats; extract.c:443-510) into e->ob[]; touches nothing shared */
static void emit_format(mdk_plan *p, const mdk_chunk *c, const md_sites *s, emit_ctx *e) {
    const opts_t *o = &p->o; const char *chrom; int64_t i; int fi; const char *seq = NULL; int64_t slen = 0, blank_from;
    char tri[4];
    e->ob[0].l = e->ob[1].l = e->ob[2].l = 0; e->n_variant = 0; e->lastcpg_tid = e->lastchg_tid = -1;
    if(c->skipped & (MDK_CHUNK_NOREF | MDK_CHUNK_BED)) return;
    chrom = p->bam->target_name[c->tid];
    fi = p->fa_of_tid[c->tid]; if(fi >= 0) { seq = p->fa.seq[fi]; slen = p->fa.len[fi]; }
    blank_from = c->beg;
    for(i = 0; i < s->n_sites; i++) {
        int32_t pos = (int32_t)s->site[i].pos; uint32_t m = s->site[i].nmeth, u = s->site[i].nunmeth; int type = (s->site[i].meta >> 1) & 3, is_g = s->site[i].meta & 1;
        if(o->min_opp_depth > 0 && s->var) {
            uint32_t noff = s->var[i].noff, nvar = s->var[i].nvar;
            if(noff >= (uint32_t)o->min_opp_depth && ((double)nvar) / ((double)noff) >= o->max_variant_frac) {
                e->n_variant++;
                if(o->merge && is_g) {
                    if(type == 0 && e->lastcpg_tid == c->tid && e->lastcpg_pos == pos - 1) { e->lastcpg_m = 0; e->lastcpg_u = 0; }
                    else if(type == 1 && e->lastchg_tid == c->tid && e->lastchg_pos == pos - 2) { e->lastchg_m = 0; e->lastchg_u = 0; }
                }
                continue;
            }
        }
        if(m + u == 0 && !o->cytosine_report) continue;
        if(!o->merge || type == 2) {
            if(o->cytosine_report) {
                put_blanks(p, &e->ob[0], chrom, seq, slen, &blank_from, pos);
                put_site(p, &e->ob[0], chrom, pos, 1, m, u, !is_g, cctx_name(type), trinuc(seq, slen, pos, is_g ? -1 : 1, tri));
                blank_from = (int64_t)pos + 1;
            } else put_site(p, &e->ob[type], chrom, pos, 1, m, u, !is_g, NULL, NULL);
        } else if(type == 0) {
            int32_t key = is_g ? pos - 1 : pos;
            if(e->lastcpg_tid == c->tid && e->lastcpg_pos == key) { put_site(p, &e->ob[0], chrom, key, 2, m + e->lastcpg_m, u + e->lastcpg_u, !is_g, NULL, NULL); e->lastcpg_tid = -1; }
            else {
                if(e->lastcpg_tid != -1) put_site(p, &e->ob[0], p->bam->target_name[e->lastcpg_tid], e->lastcpg_pos, 2, e->lastcpg_m, e->lastcpg_u, !is_g, NULL, NULL);
                e->lastcpg_tid = c->tid; e->lastcpg_pos = key; e->lastcpg_m = m; e->lastcpg_u = u;
            }
        } else {
            int32_t key = is_g ? pos - 2 : pos;
            if(e->lastchg_tid == c->tid && e->lastchg_pos == key) { put_site(p, &e->ob[1], chrom, key, 3, m + e->lastchg_m, u + e->lastchg_u, !is_g, NULL, NULL); e->lastchg_tid = -1; }
            else {
                if(e->lastchg_tid != -1) put_site(p, &e->ob[1], p->bam->target_name[e->lastchg_tid], e->lastchg_pos, 3, e->lastchg_m, e->lastchg_u, !is_g, NULL, NULL);
                e->lastchg_tid = c->tid; e->lastchg_pos = key; e->lastchg_m = m; e->lastchg_u = u;
            }
        }
    }
    if(o->merge) {      /* pending sites never cross a chunk boundary (extract.c:496-507) */
        if(o->ctx_on[0] && e->lastcpg_tid != -1) { put_site(p, &e->ob[0], p->bam->target_name[e->lastcpg_tid], e->lastcpg_pos, 2, e->lastcpg_m, e->lastcpg_u, 1, NULL, NULL); e->lastcpg_tid = -1; }
        if(o->ctx_on[1] && e->lastchg_tid != -1) { put_site(p, &e->ob[1], p->bam->target_name[e->lastchg_tid], e->lastchg_pos, 3, e->lastchg_m, e->lastchg_u, 1, NULL, NULL); e->lastchg_tid = -1; }
    } else if(o->cytosine_report) put_blanks(p, &e->ob[0], chrom, seq, slen, &blank_from, c->end);
}
/* append a formatted chunk to the output files (ordered flush, extract.c:514-535) */
static void emit_write(mdk_plan *p, emit_ctx *e) {
    int k;
    if(p->o.cytosine_report) { if(e->ob[0].l) fputs(e->ob[0].s, p->out[0]); }
    else for(k = 0; k < 3; k++) if(p->o.ctx_on[k] && e->ob[k].l) fputs(e->ob[k].s, p->out[k]);
    p->n_variant_positions += e->n_variant;
}

int mdk_plan_emit(mdk_plan *p, const mdk_chunk *c, const md_sites *s) {
    double te0 = now_s();
    if(c->index != p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    p->next_emit++;
    emit_format(p, c, s, &p->ec);
    emit_write(p, &p->ec);
    p->t_emit += now_s() - te0;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* extract_main's emitter: chunks are formatted by a few threads and written in chunk order          */
/* ------------------------------------------------------------------------------------------------ */
static int pwrite_all(int fd, const char *s, size_t n, int64_t off) {       /* 0, or the errno of the write that failed */
    while(n) { ssize_t w = pwrite(fd, s, n, (off_t)off); if(w < 0) { if(errno == EINTR) continue; return errno ? errno : EIO; } if(w == 0) return ENOSPC; s += w; n -= (size_t)w; off += w; }
    return 0;
}
static void *emitter_main(void *arg) {
    emitter *E = arg;
    for(;;) {
        ejob *j = NULL; int i; double t0;
        pthread_mutex_lock(&E->mu);
        for(;;) {
            for(i = 0; i < E->n_job; i++) if(E->job[i].state == EJ_READY && (!j || E->job[i].c.index < j->c.index)) j = &E->job[i];
            if(j || E->quit) break;
            pthread_cond_wait(&E->cv_job, &E->mu);
        }
        if(!j) { pthread_mutex_unlock(&E->mu); break; }
        j->state = EJ_BUSY;
        pthread_mutex_unlock(&E->mu);
        if(j->need_copy) {                       /* emitter_push_lazy: the copy out of the caller's buffer, by this thread instead of by the caller */
            memcpy(j->site, j->src_site, sizeof(md_site) * (size_t)j->s.n_sites); if(j->src_var) memcpy(j->var, j->src_var, sizeof(md_site_var) * (size_t)j->s.n_sites);
            pthread_mutex_lock(&E->mu); j->need_copy = 0; E->n_uncopied--; pthread_cond_broadcast(&E->cv_copied); pthread_mutex_unlock(&E->mu);
        }
        t0 = now_s();
        emit_format(E->p, &j->c, &j->s, &j->e);
        pthread_mutex_lock(&E->mu);
        E->t_format += now_s() - t0;
        while(E->next_write != j->c.index) pthread_cond_wait(&E->cv_turn, &E->mu);
        if(E->pw) {                              /* in turn: only the byte ranges; the copying into the files runs in parallel */
            int64_t at[3] = {0, 0, 0}; int k, bad = 0; const int nk = E->p->o.cytosine_report ? 1 : 3;
            for(k = 0; k < nk; k++) if(E->p->o.cytosine_report || E->p->o.ctx_on[k]) { at[k] = E->woff[k]; E->woff[k] += (int64_t)j->e.ob[k].l; }
            E->p->n_variant_positions += j->e.n_variant;
            E->next_write++; pthread_cond_broadcast(&E->cv_turn);
            pthread_mutex_unlock(&E->mu);
            for(k = 0; k < nk; k++) if((E->p->o.cytosine_report || E->p->o.ctx_on[k]) && j->e.ob[k].l && !bad) bad = pwrite_all(E->fd[k], j->e.ob[k].s, j->e.ob[k].l, at[k]);
            pthread_mutex_lock(&E->mu);
            if(bad && !E->failed) { E->failed = 1; fprintf(stderr, "[mdk] writing the output failed: %s\n", strerror(bad)); }
            j->state = EJ_FREE; pthread_cond_signal(&E->cv_free);
            pthread_mutex_unlock(&E->mu);
            continue;
        }
        emit_write(E->p, &j->e);                 /* in turn, so under the lock: nobody else may write now anyway */
        E->next_write++; j->state = EJ_FREE;
        pthread_cond_broadcast(&E->cv_turn); pthread_cond_signal(&E->cv_free);
        pthread_mutex_unlock(&E->mu);
    }
    return NULL;
}
MDK_LOCAL int emitter_start(emitter *E, mdk_plan *p, int n_th) {
    int i;
    memset(E, 0, sizeof(*E));
    E->p = p; E->n_th = n_th < 1 ? 1 : n_th; E->n_job = E->n_th + (getenv("MDK_EMIT_JOBS_EXTRA") ? atoi(getenv("MDK_EMIT_JOBS_EXTRA")) : 2); if(E->n_job < E->n_th + 1) E->n_job = E->n_th + 1; E->next_write = p->next_emit;
    E->job = xcalloc((size_t)E->n_job, sizeof(ejob)); E->th = xcalloc((size_t)E->n_th, sizeof(pthread_t));
    if(!E->job || !E->th) return -5;
    pthread_mutex_init(&E->mu, NULL); pthread_cond_init(&E->cv_job, NULL); pthread_cond_init(&E->cv_free, NULL); pthread_cond_init(&E->cv_turn, NULL); pthread_cond_init(&E->cv_copied, NULL);
    {   /* regular output files (what -o makes): what has been written so far (the header lines) is flushed, from here on the
         * emitter threads write at explicit offsets.  Anything else (a FIFO, /dev/null) keeps the ordered stream writes. */
        const int nk = p->o.cytosine_report ? 1 : 3; int k;
        E->pw = getenv("MDK_NO_PWRITE") ? 0 : 1;
        for(k = 0; k < nk && E->pw; k++) {
            struct stat st; off_t at;
            if(!(p->o.cytosine_report || p->o.ctx_on[k])) continue;
            if(!p->out[k] || fflush(p->out[k]) || (E->fd[k] = fileno(p->out[k])) < 0 || fstat(E->fd[k], &st) || !S_ISREG(st.st_mode) || (at = lseek(E->fd[k], 0, SEEK_CUR)) < 0) { E->pw = 0; break; }
            E->woff[k] = (int64_t)at;
        }
    }
    for(i = 0; i < E->n_th; i++) if(pthread_create(&E->th[i], NULL, emitter_main, E)) break;
    if(i == 0) { fprintf(stderr, "[mdk] cannot create an emitter thread\n"); return -5; }
    E->n_th = i;            /* fewer than asked for still drain the same queue */
    return 0;
}
/* hand a chunk and its sites over (both are copied: the caller's buffers are recycled).  lazy: a large site array is copied by the emitter
 * thread that takes the job, not here -- the caller hands over all the chunks of a group and then waits for the copies with
 * emitter_wait_copied before it recycles the buffers: the dense contexts' 6 MB per chunk, copied by the collector alone out of the pinned
 * buffers the device had just written, were 0.19 s of a 128 Mb run's 0.66 s (profiles/r06c3_profile.txt) */
static int emitter_push_ex(emitter *E, const mdk_chunk *c, const md_sites *s, int lazy);
MDK_LOCAL int emitter_push(emitter *E, const mdk_chunk *c, const md_sites *s) { return emitter_push_ex(E, c, s, 0); }
MDK_LOCAL int emitter_push_lazy(emitter *E, const mdk_chunk *c, const md_sites *s) { return emitter_push_ex(E, c, s, 1); }
static double g_t_free = 0, g_t_fill = 0, g_t_copied = 0;      /* the collector's time in here by what it waited for (MDK_HOST_PROFILE) */
MDK_LOCAL void emitter_wait_copied(emitter *E) { const double t0 = now_s(); pthread_mutex_lock(&E->mu); while(E->n_uncopied) pthread_cond_wait(&E->cv_copied, &E->mu); pthread_mutex_unlock(&E->mu); g_t_copied += now_s() - t0; }
static int emitter_push_ex(emitter *E, const mdk_chunk *c, const md_sites *s, int lazy) {
    ejob *j = NULL; int i;
    if(c->index != E->p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    if(E->failed) return -3;                     /* a write has failed (the reason was printed): the caller stops feeding the GPU for a file that cannot be completed */
    E->p->next_emit++;
    const double tf0 = now_s();
    pthread_mutex_lock(&E->mu);
    for(;;) { for(i = 0; i < E->n_job; i++) if(E->job[i].state == EJ_FREE) { j = &E->job[i]; break; } if(j) break; pthread_cond_wait(&E->cv_free, &E->mu); }
    j->state = EJ_BUSY;                          /* being filled */
    pthread_mutex_unlock(&E->mu);
    const double tf1 = now_s(); g_t_free += tf1 - tf0;
    j->c = *c; j->s = *s;
    if(s->n_sites > j->cap) {
        j->cap = s->n_sites + s->n_sites / 4 + 1024; free(j->site); free(j->var);
        j->site = malloc(sizeof(md_site) * (size_t)j->cap); j->var = malloc(sizeof(md_site_var) * (size_t)j->cap);
        if(!j->site || !j->var) { fprintf(stderr, "[mdk] out of memory while queueing a chunk for output\n"); abort(); }      /* nothing sensible can be written in order any more */
    }
    { static long lazy_min = -1; if(lazy_min < 0) lazy_min = getenv("MDK_LAZY_COPY_MIN") ? atol(getenv("MDK_LAZY_COPY_MIN")) : 16384;      /* (256 KB: below that the copy is not worth a hand-over; the variable is a test hook) */
      j->need_copy = lazy && s->n_sites >= lazy_min && s->n_sites > 0; }
    if(j->need_copy) { j->src_site = s->site; j->src_var = s->var; }
    else if(s->n_sites) { memcpy(j->site, s->site, sizeof(md_site) * (size_t)s->n_sites); if(s->var) memcpy(j->var, s->var, sizeof(md_site_var) * (size_t)s->n_sites); }
    j->s.site = j->site; j->s.var = s->var ? j->var : NULL;
    g_t_fill += now_s() - tf1;
    pthread_mutex_lock(&E->mu); if(j->need_copy) E->n_uncopied++; j->state = EJ_READY; pthread_cond_signal(&E->cv_job); pthread_mutex_unlock(&E->mu);
    return 0;
}
MDK_LOCAL void emitter_stop(emitter *E) {
    int i;
    if(!E->th) return;
    pthread_mutex_lock(&E->mu);
    while(E->next_write != E->p->next_emit) pthread_cond_wait(&E->cv_turn, &E->mu);       /* everything handed over has been written */
    E->quit = 1; pthread_cond_broadcast(&E->cv_job);
    pthread_mutex_unlock(&E->mu);
    for(i = 0; i < E->n_th; i++) pthread_join(E->th[i], NULL);
    if(E->pw) { int k; const int nk = E->p->o.cytosine_report ? 1 : 3; for(k = 0; k < nk; k++) if((E->p->o.cytosine_report || E->p->o.ctx_on[k]) && E->p->out[k]) (void)fseeko(E->p->out[k], (off_t)E->woff[k], SEEK_SET); }
    for(i = 0; i < E->n_job; i++) { int k; free(E->job[i].site); free(E->job[i].var); for(k = 0; k < 3; k++) free(E->job[i].e.ob[k].s); }
    E->p->t_emit += E->t_format;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] handing chunks to the emitter: waiting for a free job %.3fs, sizing and copying %.3fs, waiting for the emitter threads' copies %.3fs\n", g_t_free, g_t_fill, g_t_copied);
    free(E->job); free(E->th); E->th = NULL;
}


int mdk_plan_finish(mdk_plan *p) {
    int i;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk host] inflate+frame+admit+pack %.3fs (inflate alone %.3fs)  pairing %.3fs  segments %.3fs  emit %.3fs; records found in the inflate threads' tables %" PRIu64 ", by walking %" PRIu64 "; reader: scanning %.3fs, waiting for a free slot %.3fs; workers busy %.3fs idle %.3fs (sum over %d); pieces inflated by the host teams %" PRIu64 ", on the device %" PRIu64 " (of which read back: %" PRIu64 ")\n", p->t_collect, p->bam->t_inflate, p->t_pair, p->t_segs, p->t_emit, p->bam->n_fast, p->bam->n_slow, p->t_rfill, p->t_rwait, p->t_wbusy, p->t_widle, p->n_workers, p->bam->n_host_pieces, p->bam->n_dev_pieces, p->bam->n_materialized);
    if(p->n_variant_positions) printf("%" PRIu64 " positions were excluded due to likely being variants.\n", p->n_variant_positions);
    if(p->o.cytosine_report) { if(p->out[0]) fclose(p->out[0]); p->out[0] = p->out[1] = p->out[2] = NULL; }
    else for(i = 0; i < 3; i++) if(p->out[i]) { fclose(p->out[i]); p->out[i] = NULL; }
    return 0;
}


/*
 * mdk_extract.c -- extract_main, the drop-in entry point of the MI355X `MethylDackel extract` path (extract.c:706 in the reference).
 *
 * Division of labour:
 *   host  : options and inputs (mdk_plan.c), reading the file and part of the BGZF inflate (mdk_io.c), the reference's chunk
 *           schedule (extract.c:325-350, common.c:466-493) applied to member digests and record tables (mdk_pipeline.c), and the
 *           text post-pass (extract.c:443-510, 39-99; mdk_emit.c).  Per record or per base the host does nothing, except for a
 *           chunk the device hands back (MDK_ERR_PREP_HOST) and under MDK_HOST_PREP=1.
 *   device: the rest of the BGZF inflate and the record framing (k_inflate, k_walk), read admission (filter_func, common.c:416-444),
 *           strand (getStrand, common.c:84-116), read-name pairing (overlaps.c:121-147), CIGAR expansion, trimming, overlap
 *           resolution, context classification, counting.
 * Three threads move the chunks here: this one uploads and launches them in groups, a second collects the results and hands them
 * to the emitter, a third uploads the contigs' references ahead of the chunks that need them.
 * There is no CPU implementation of the per-base work in this library.
 */
#include <fcntl.h>
#include <sys/prctl.h>
#include "mdk_plan.h"

/* ------------------------------------------------------------------------------------------------ */
/* the drop-in entry point                                                                           */
/* ------------------------------------------------------------------------------------------------ */
/* The `MethylDackel` command asks (MDK_FAST_EXIT) to leave with _exit once the outputs are closed, skipping the unpinning
 * of buffers and the HIP shutdown.  Not under a profiler or another injected tool: those finalise at normal exit. */
static double rss_mb(int shared) {      /* resident set (or its file-backed/shared part) in MB: host profile only */
    long vm = 0, rss = 0, shr = 0; FILE *sf = fopen("/proc/self/statm", "r");
    if(sf) { if(fscanf(sf, "%ld %ld %ld", &vm, &rss, &shr) != 3) rss = shr = 0; fclose(sf); }
    return (shared ? shr : rss) * 4096e-6;
}
/* formatting threads: 8 are ahead of a CpG-only run; the dense contexts print ~20x the lines */
static int emit_threads(const mdk_plan *p) { const int cap = (p->o.ctx_on[1] || p->o.ctx_on[2]) ? 24 : 8; return p->o.n_threads >= cap ? cap : p->o.n_threads; }
MDK_LOCAL int fast_exit_wanted(void) {
    const char *pre = getenv("LD_PRELOAD");
    if(!getenv("MDK_FAST_EXIT")) return 0;
    if(getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCPROF_OUTPUT_PATH")) return 0;
    if(pre && (strstr(pre, "rocprof") || strstr(pre, "roctx") || strstr(pre, "rocm"))) return 0;
    return 1;
}
/* leave now: outputs are flushed and closed; nobody needs to wait for staging buffers to be unpinned one by one and for the
 * runtime's exit handlers (the `MethylDackel` command only: main.c sets MDK_FAST_EXIT) */
MDK_LOCAL void leave_fast(int ret) {
    const char *fdv = getenv("MDK_DONE_FD");
    fflush(stdout); fflush(stderr);
    if(fdv) {     /* the command runs as the child of a process that only waits for this word (main.c detach_teardown): outputs are closed, it may return */
        const int fd = atoi(fdv), nul = open("/dev/null", O_RDWR);
        unsigned char code = (unsigned char)(ret & 0xff);
        if(nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); if(nul > 2) close(nul); }
        (void)prctl(PR_SET_PDEATHSIG, 0);
        if(write(fd, &code, 1) != 1) { /* nobody is listening any more */ }
        close(fd);
    }
    mdk_cli_quiesce();
    _exit(ret & 0xff);
}
/* the HIP runtime takes 0.1-0.4 s to come up: start that before anything else (options, BAM header, FASTA), on its own thread */
static void *hipwarm_main(void *arg) { (void)arg; (void)md_dev_warm(getenv("MDK_DEVICE") ? atoi(getenv("MDK_DEVICE")) : 0); return NULL; }
/* only in the `MethylDackel` command, which always ends with _exit (leave_fast): a library caller whose bad command line makes
 * us return at once must not find a half-initialised runtime racing its exit handlers.  The thread is joined -- and with it the side
 * threads md_dev_warm starts (md_dev_quiesce) -- before the command leaves: mdk_cli_quiesce, called by leave_fast and by main.c after
 * the caller has been told that the outputs are closed. */
static pthread_t g_warm_th; static int g_warm_started = 0;
static mdk_plan *g_leaving = NULL;          /* the plan of a command on its way out: its reaper is waited for (mdk_io.h) */
MDK_LOCAL void hip_warm_up(void) {
    if(!fast_exit_wanted() || g_warm_started) return;
    if(pthread_create(&g_warm_th, NULL, hipwarm_main, NULL) == 0) g_warm_started = 1;
}
MDK_LOCAL void leave_fast_plan(mdk_plan *p, int ret) { g_leaving = p; leave_fast(ret); }      /* every command's way out: the plan's reaper and device teams are waited for */
void mdk_cli_quiesce(void) {
    if(g_leaving && g_leaving->bam) { mdk_bam_teams_leave(g_leaving->bam); mdk_bam_reap_wait(g_leaving->bam); }
    if(g_warm_started) { pthread_join(g_warm_th, NULL); g_warm_started = 0; }
    md_dev_quiesce();
}
/* md_dev_last_error is per thread: keep the text of a failed open for the thread that reports it */
MDK_LOCAL void *devopen_main(void *arg) { devopen_t *d = arg; d->rc = md_dev_open(d->device, &d->cfg, &d->dev); if(d->rc) snprintf(d->err, sizeof(d->err), "%s", md_dev_last_error()); return NULL; }

/* Chunks travel to the device in GROUPS of up to MDK_GROUP: a 1 Mb chunk alone is fewer than two workgroups per CU and pays every
 * launch boundary itself, so whatever the reader has ready when a group is opened (at least one chunk, at most eight) is uploaded
 * into the group's slots and prepared and piled up with one launch per kernel (md_dev_launch_group).  MDK_NGROUPS groups are in
 * flight, each on its own stream: while one computes, the next is uploaded by this thread and the one before is collected by the
 * collector thread (one wait and one round of copies per group, md_dev_download_group) and handed to the emitter in schedule
 * order.  The reference's unit of work is the chunk (extract.c:325-350); here it is the unit of scheduling and of output only. */
#define MDK_GROUP 8
#define MDK_NGROUPS_MAX 6
static int g_ngroups = 3;            /* groups in flight (MDK_GROUPS_IN_FLIGHT=n, 2..6) */
#define MDK_NGROUPS g_ngroups
enum { G_FREE = 0, G_FILL, G_LAUNCHED };
typedef struct { mdk_chunk ch[MDK_GROUP]; int slot[MDK_GROUP]; int n, launched[MDK_GROUP], inplace[MDK_GROUP], state, held, n_held, rel_slot[MDK_GROUP]; mdk_chunk rel_ch[MDK_GROUP]; } cgroup;      /* held: the host memory behind its records has not been given back yet; inplace: the device reads the chunk's records where the piece they were inflated in holds them (md_dev_upload_raw_inplace): that piece goes back when the chunk's results are in */
typedef struct {
    mdk_plan *p; md_dev *dev; emitter *em; cgroup G[MDK_NGROUPS_MAX];
    pthread_mutex_t mu; pthread_cond_t cv;
    int ret, up_done;                    /* (mu) first error; the uploader has launched its last group */
    uint64_t n_up, n_col;                /* (mu) groups launched / collected: group k lives in G[k % MDK_NGROUPS] */
    int *ref_state; int ref_quit, ref_done; int32_t ref_t0, ref_t1;        /* (mu) per contig: 0 not uploaded yet, 1 resident, < 0 the error its upload met; ref_done: the thread has left */
    double w_down, w_emit; int n_host_prep;
} xpipe;
/* Once a group's records have crossed the link (a few ms: they were queued before its kernels), the staging memory they came from goes back to
 * the inflate teams -- not when the results are in.  block = 0: only if the copies are done already; 1: wait for them (the reader is out of
 * chunks, perhaps for want of that very memory).  Called by the uploader thread only. */
static void release_uploaded(xpipe *X, int block) {
    int k, i;
    if(getenv("MDK_NO_EARLY_RELEASE")) return;
    for(k = 0; k < MDK_NGROUPS; k++) {
        cgroup *g = &X->G[(X->n_up + (uint64_t)k) % MDK_NGROUPS]; int last = -1;      /* oldest first: the group about to be refilled, ..., the one launched last */
        if(!g->held) continue;
        for(i = 0; i < g->n_held; i++) if(g->rel_slot[i] >= 0) last = g->rel_slot[i];
        if(last >= 0) { const int done = block ? (md_dev_upload_wait(X->dev, last) == 0) : md_dev_upload_done(X->dev, last) == 1; if(!done) return; }      /* (in order: a later group's copies are queued behind) */
        for(i = 0; i < g->n_held; i++) if(g->rel_slot[i] >= 0) (void)mdk_plan_release_records(X->p, &g->rel_ch[i]);
        g->held = 0; block = 0;
    }
}
static void xp_fail(xpipe *X, int ret) { pthread_mutex_lock(&X->mu); if(!X->ret) X->ret = ret; pthread_cond_broadcast(&X->cv); pthread_mutex_unlock(&X->mu); }

/* the contigs' bases (and BED runs, mappability tracks) go to the device ahead of the chunks, in schedule order */
static void *refs_main(void *arg) {
    xpipe *X = arg; mdk_plan *p = X->p; int32_t t;
    for(t = X->ref_t0; t < X->ref_t1 && t < p->bam->n_targets; t++) {
        int rc = 1, q;
        pthread_mutex_lock(&X->mu); q = X->ref_quit || X->ret; pthread_mutex_unlock(&X->mu);
        if(q) break;
        if(p->fa_of_tid[t] >= 0) { rc = mdk_plan_ensure_reference(p, X->dev, t); rc = rc ? (rc < 0 ? rc : -1) : 1; }
        pthread_mutex_lock(&X->mu); X->ref_state[t] = rc; pthread_cond_broadcast(&X->cv); pthread_mutex_unlock(&X->mu);
    }
    pthread_mutex_lock(&X->mu); X->ref_done = 1; pthread_cond_broadcast(&X->cv); pthread_mutex_unlock(&X->mu);
    return NULL;
}
static int ref_wait(xpipe *X, int32_t tid) {
    int rc;
    pthread_mutex_lock(&X->mu);
    while(!X->ref_state[tid] && !X->ret && !X->ref_done) pthread_cond_wait(&X->cv, &X->mu);
    rc = X->ref_state[tid];
    pthread_mutex_unlock(&X->mu);
    if(!rc && !X->ret) { rc = mdk_plan_ensure_reference(X->p, X->dev, tid); return rc; }       /* a contig the thread did not have on its list (it has left: nobody else uploads) */
    return rc == 1 ? 0 : rc ? rc : -1;
}

/* MDK_WATCHDOG=1 (diagnostics): once a second, where every stage of the pipeline stands -- for a run that stalls.  MDK_WATCHDOG=<ms> with
 * ms >= 2: the same every <ms> milliseconds in short form, a time series of the queues between the stages (which one runs empty, which one full) */
static volatile int g_up_phase, g_col_phase;
static void *watchdog_main(void *arg) {
    xpipe *X = arg; mdk_plan *p = X->p; mdk_bam *b = p->bam; int k; const double t0 = now_s();
    const int ms = atoi(getenv("MDK_WATCHDOG")) >= 2 ? atoi(getenv("MDK_WATCHDOG")) : 1000;
    for(;;) {
        int q, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if(ms >= 1000) { for(k = 0; k < 10; k++) { usleep(100000); pthread_mutex_lock(&X->mu); q = X->up_done && X->n_col == X->n_up; pthread_mutex_unlock(&X->mu); if(q) return NULL; } }
        else { usleep((useconds_t)ms * 1000); pthread_mutex_lock(&X->mu); q = X->up_done && X->n_col == X->n_up; pthread_mutex_unlock(&X->mu); if(q) return NULL; }
        for(k = 0; k < p->n_slot; k++) st[p->slot_state ? p->slot_state(p, k) & 7 : 0]++;
        if(ms < 1000) fprintf(stderr, "[wd] t=%.3f up=%d L=%" PRIu64 " C=%" PRIu64 " col=%d out=%u slots=%d/%d/%d/%d/%d/%d pieces=%" PRIu64 "/%" PRIu64 "/%d slabs=%d+%d/%d+%d\n",
                now_s() - t0, g_up_phase, X->n_up, X->n_col, g_col_phase, p->next_out, st[0], st[1], st[2], st[3], st[4], st[5], b->next_seq, b->pop_seq, b->n_ready, b->n_alloc, b->n_pool, b->n_dalloc, b->n_dpool);
        else
        fprintf(stderr, "[mdk watchdog] %.1fs: uploader phase %d groups launched %" PRIu64 " collected %" PRIu64 " collector phase %d | chunks handed out %u, slots free/fill/raw/work/done/held %d/%d/%d/%d/%d/%d | pieces handed %" PRIu64 " popped %" PRIu64 " ready %d, slabs host %d(+%d free) device %d(+%d free), io %d inf_done %d\n",
                now_s() - t0, g_up_phase, X->n_up, X->n_col, g_col_phase, p->next_out, st[0], st[1], st[2], st[3], st[4], st[5], b->next_seq, b->pop_seq, b->n_ready, b->n_alloc, b->n_pool, b->n_dalloc, b->n_dpool, b->io_status, b->inf_done);
    }
}

/* collects the launched groups in order: results to the emitter, the group back to the uploader */
static void *collector_main(void *arg) {
    xpipe *X = arg; mdk_plan *p = X->p; md_dev *dev = X->dev; int i;
    for(;;) {
        cgroup *g; int ls[MDK_GROUP], li[MDK_GROUP], nl = 0, rcs[MDK_GROUP], rc = 0, bad = 0; md_sites st[MDK_GROUP], sites[MDK_GROUP]; double ta;
        pthread_mutex_lock(&X->mu);
        while(X->n_col == X->n_up && !X->up_done && !X->ret) pthread_cond_wait(&X->cv, &X->mu);
        if(X->ret || X->n_col == X->n_up) { pthread_mutex_unlock(&X->mu); break; }
        g = &X->G[X->n_col % MDK_NGROUPS];
        pthread_mutex_unlock(&X->mu);
        memset(sites, 0, sizeof(sites));
        for(i = 0; i < g->n; i++) if(g->launched[i]) { ls[nl] = g->slot[i]; li[nl] = i; nl++; }
        ta = now_s(); g_col_phase = 1;
        if(nl) rc = md_dev_download_group(dev, ls, nl, st, rcs);
        g_col_phase = 2;
        if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); xp_fail(X, MDK_RC_DEVICE); break; }
        for(i = 0; i < nl && !bad; i++) {
            const int k = li[i];
            rc = rcs[i];
            if(rc == MDK_ERR_PREP_HOST) {          /* a read name the device preparation does not handle: this chunk the slow way */
                static int told = 0;
                if(!told) { told = 1; fprintf(stderr, "[mdk] note: a chunk holds a read name with more records than the device preparation handles (secondary/supplementary-rich or amplicon-like data); such chunks are prepared on the host, which is slower\n"); }
                rc = mdk_plan_host_prepare_from(p, &g->ch[k], dev, g->slot[k]);
                if(!rc) rc = md_dev_submit(dev, g->slot[k], &g->ch[k].batch);
                if(!rc) rc = md_dev_download(dev, g->slot[k], &st[i]);
                X->n_host_prep++;
            }
            if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
            if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); xp_fail(X, MDK_RC_DEVICE); bad = 1; break; }
            sites[k] = st[i];
            if(g->inplace[k] && !getenv("MDK_NO_EARLY_RELEASE")) (void)mdk_plan_release_records(p, &g->ch[k]);      /* the kernels that read the piece are done: it goes back to the inflate teams */
        }
        X->w_down += now_s() - ta;
        if(bad) break;
        ta = now_s();
        g_col_phase = 3;
        for(i = 0; i < g->n; i++) if(emitter_push_lazy(X->em, &g->ch[i], &sites[i])) { xp_fail(X, X->em->failed ? MDK_RC_OUTPUT : MDK_RC_DEVICE); bad = 1; break; }
        emitter_wait_copied(X->em);                  /* (the emitter threads copy the group's site arrays side by side; the slots' buffers are the device's again from here) */
        X->w_emit += now_s() - ta;
        if(bad) break;
        g_col_phase = 0;
        pthread_mutex_lock(&X->mu); g->n = 0; g->state = G_FREE; X->n_col++; pthread_cond_broadcast(&X->cv); pthread_mutex_unlock(&X->mu);
    }
    return NULL;
}

/* the slabs the host teams filled while the runtime was still starting: registered with it now, next to the uploader instead of by it */
static void *prereg_main(void *arg) { md_dev *dev = arg; (void)md_host_register_all(dev, 1); return NULL; }

/* opening the device on its own thread, from the moment the options are known: the inputs (BAM header, index, FASTA) are opened meanwhile */
typedef struct { devopen_t d; pthread_t th; int started; } xopen;
static void xopen_start(mdk_plan *p, void *arg) {
    xopen *o = arg;
    if(getenv("MDK_GROUPS_IN_FLIGHT")) { g_ngroups = atoi(getenv("MDK_GROUPS_IN_FLIGHT")); if(g_ngroups < 2) g_ngroups = 2; if(g_ngroups > MDK_NGROUPS_MAX) g_ngroups = MDK_NGROUPS_MAX; }
    mdk_plan_dev_cfg(p, &o->d.cfg);
    o->d.cfg.n_slots = MDK_NGROUPS * MDK_GROUP; o->d.cfg.n_streams = MDK_NGROUPS;
    if(getenv("MDK_DEVICE")) o->d.device = atoi(getenv("MDK_DEVICE"));
    o->started = pthread_create(&o->th, NULL, devopen_main, &o->d) == 0;
}

int extract_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; xpipe *X = NULL; int rc, ret = 0, more = 1, i, g_i; xopen dop; pthread_t cth, rth, preg; int cth_ok = 0, rth_ok = 0, preg_ok = 0; emitter em;
    double T0 = now_s(), t_open, t_dev, w_next = 0, w_sub = 0, w_group = 0, w_ref = 0, w_rel = 0, ta; uint64_t n_chunks = 0; int32_t ref_t0, ref_t1;
    if(getenv("MDK_HOST_PROFILE")) { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[mdk main] entered at epoch %.3f\n", ts.tv_sec + 1e-9 * ts.tv_nsec); }
    { int rk = 0, wd = 1, m = ranks_from_env(&rk, &wd); if(m < 0) return -1; if(m > 0) return extract_ranks(argc, argv, rk, wd); }       /* one process per GPU (mdk_ranks.c) */
    if(argc > 2) hip_warm_up();
    memset(&dop, 0, sizeof(dop));
    rc = plan_open_ex(argc, argv, &p, xopen_start, &dop);
    t_open = now_s() - T0;
    if(rc != 0 || !p) { if(dop.started) { pthread_join(dop.th, NULL); if(dop.d.dev) md_dev_close(dop.d.dev); } return rc; }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] resident after plan open %.0f MB\n", rss_mb(0));
    ref_t0 = p->o.region ? (int32_t)p->g_tid : 0; ref_t1 = (p->o.region && p->g_end) ? ref_t0 + 1 : p->bam->n_targets;       /* (before the reader moves the schedule) */
    /* the per-record work of a chunk (admission, strand, name pairing, CIGAR expansion) runs on the device; MDK_HOST_PREP=1 keeps
     * it on the host's chunk workers (the round-1 arrangement, and what a chunk the device gives up on falls back to) */
    if(!getenv("MDK_HOST_PREP")) mdk_plan_set_prep(p, 1);
    mdk_plan_set_hold(p, MDK_NGROUPS * MDK_GROUP + 2);
    if(!p->started && pipeline_start(p)) { if(dop.started) pthread_join(dop.th, NULL); if(dop.d.dev) md_dev_close(dop.d.dev); mdk_plan_close(p); return -5; }
    if(dop.started) pthread_join(dop.th, NULL); else { xopen_start(p, &dop); if(dop.started) pthread_join(dop.th, NULL); else devopen_main(&dop.d); }
    t_dev = now_s() - T0;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] resident at device ready %.0f MB\n", rss_mb(0));
    dev = dop.d.dev;
    if(dop.d.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `extract`.\n", user_device(dop.d.device), dop.d.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    (void)md_dev_reserve_contigs(dev, p->bam->n_targets);
    if(p->dev_prep) { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); }
    if(p->dev_prep) mdk_plan_attach_device(p, dev);      /* from here on the device inflates pieces of the file too */
    X = calloc(1, sizeof(*X));
    if(X) X->ref_state = calloc((size_t)p->bam->n_targets + 1, sizeof(int));
    if(!X || !X->ref_state || emitter_start(&em, p, emit_threads(p))) { if(X) free(X->ref_state); free(X); mdk_plan_detach_device(p); md_dev_close(dev); mdk_plan_close(p); return -5; }
    X->p = p; X->dev = dev; X->em = &em; X->ref_t0 = ref_t0; X->ref_t1 = ref_t1; pthread_mutex_init(&X->mu, NULL); pthread_cond_init(&X->cv, NULL);
    for(g_i = 0; g_i < MDK_NGROUPS; g_i++) for(i = 0; i < MDK_GROUP; i++) X->G[g_i].slot[i] = g_i * MDK_GROUP + i;
    preg_ok = !getenv("MDK_NO_PREREG") && pthread_create(&preg, NULL, prereg_main, dev) == 0;
    rth_ok = pthread_create(&rth, NULL, refs_main, X) == 0;
    cth_ok = pthread_create(&cth, NULL, collector_main, X) == 0;
    if(getenv("MDK_WATCHDOG")) { pthread_t wd; if(pthread_create(&wd, NULL, watchdog_main, X) == 0) pthread_detach(wd); }
    if(!rth_ok || !cth_ok) { fprintf(stderr, "[mdk] cannot create a thread\n"); ret = -5; more = 0; }
    while(more && !ret) {
        cgroup *g;
        /* the next group, once the collector has given it back */
        ta = now_s(); g_up_phase = 1;
        pthread_mutex_lock(&X->mu);
        g = &X->G[X->n_up % MDK_NGROUPS];
        while(g->state != G_FREE && !X->ret) pthread_cond_wait(&X->cv, &X->mu);
        ret = X->ret; g->state = G_FILL;
        pthread_mutex_unlock(&X->mu);
        w_group += now_s() - ta;
        if(ret) break;
        g_up_phase = 2; ta = now_s(); while(g->held) release_uploaded(X, 1); release_uploaded(X, 0); w_rel += now_s() - ta; g_up_phase = 3;
        /* the first chunk is waited for, the others are taken only if they are ready now */
        g->n = 0;
        while(more && g->n < MDK_GROUP) {
            mdk_chunk *c = &g->ch[g->n];
            ta = now_s();
            rc = mdk_plan_try_next_chunk(p, c);
            if(rc == 2 && g->n == 0) {      /* nothing ready: before waiting, give back what can be given back -- the reader may be short of that very memory */
                const double tb = now_s(); g_up_phase = 4; release_uploaded(X, 1); w_rel += now_s() - tb; ta += now_s() - tb;
                g_up_phase = 5; rc = mdk_plan_next_chunk(p, c); g_up_phase = 3;
            }
            w_next += now_s() - ta;
            if(rc == 2) break;
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) { more = 0; break; }
            g->launched[g->n] = 0; g->inplace[g->n] = 0;
            if(!c->skipped) {
                ta = now_s(); rc = c->prep ? 0 : ref_wait(X, c->tid); w_ref += now_s() - ta;       /* (raw records can cross the link before the contig's bases have) */
                ta = now_s(); g_up_phase = 6;
                g->inplace[g->n] = 0;
                if(!rc) { if(c->prep) { rc = md_dev_upload_raw_inplace(dev, g->slot[g->n], &c->raw); if(rc > 0) { g->inplace[g->n] = 1; rc = 0; } } else rc = md_dev_upload(dev, g->slot[g->n], &c->batch); }
                w_sub += now_s() - ta; g_up_phase = 3;
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                g->launched[g->n] = 1;
            }
            g->n++; n_chunks++;
        }
        if(ret) break;
        {   /* one launch per kernel for the group's chunks */
            int ls[MDK_GROUP], nl = 0;
            for(i = 0; i < g->n; i++) if(g->launched[i]) ls[nl++] = g->slot[i];
            ta = now_s(); g_up_phase = 7;
            for(i = 0, rc = 0; i < g->n && !rc; i++) if(g->launched[i] && g->ch[i].prep) rc = ref_wait(X, g->ch[i].tid);
            w_ref += now_s() - ta;
            if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            g_up_phase = 8;
            if(nl) { ta = now_s(); rc = md_dev_launch_group(dev, ls, nl); w_sub += now_s() - ta; if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; } }
        }
        g->n_held = g->n; g->held = 0;
        for(i = 0; i < g->n; i++) { g->rel_slot[i] = (g->launched[i] && g->ch[i].prep && !g->inplace[i]) ? g->slot[i] : -1; g->rel_ch[i] = g->ch[i]; if(g->rel_slot[i] >= 0 && !getenv("MDK_NO_EARLY_RELEASE")) g->held = 1; }
        pthread_mutex_lock(&X->mu);
        if(g->n) { g->state = G_LAUNCHED; X->n_up++; } else g->state = G_FREE;
        pthread_cond_broadcast(&X->cv);
        pthread_mutex_unlock(&X->mu);
        ta = now_s(); release_uploaded(X, 0); w_rel += now_s() - ta;
    }
    /* the last groups' records have crossed the link or are about to: their slabs go back too (to the reaper: the file has been read to its end),
     * instead of staying registered until the process ends */
    if(!ret) { int guard = 0; for(;;) { int any = 0; for(i = 0; i < MDK_NGROUPS; i++) any |= X->G[i].held; if(!any || ++guard > 2 * MDK_NGROUPS_MAX) break; release_uploaded(X, 1); } }
    if(ret) xp_fail(X, ret);
    pthread_mutex_lock(&X->mu); X->up_done = 1; X->ref_quit = 1; pthread_cond_broadcast(&X->cv); pthread_mutex_unlock(&X->mu);
    if(cth_ok) pthread_join(cth, NULL);
    if(rth_ok) pthread_join(rth, NULL);
    if(preg_ok) pthread_join(preg, NULL);
    if(!ret) ret = X->ret;
    { double tw = now_s(); emitter_stop(&em); X->w_emit += now_s() - tw; }
    if(em.failed && !ret) ret = MDK_RC_OUTPUT;
    if(getenv("MDK_HOST_PROFILE")) { double rs = 0; uint64_t rc2 = 0, rb = 0; md_host_profile(&rs, &rc2, &rb); fprintf(stderr, "[mdk main] staging blocks registered: %" PRIu64 " (%.0f MB) in %.3fs; %" PRIu64 " chunks in %" PRIu64 " group launches\n", rc2, rb / 1048576.0, rs, n_chunks, X->n_up); }
    if(getenv("MDK_HOST_PROFILE")) { char pt[1024]; if(md_dev_profile_text(pt, sizeof(pt)) == 0) fprintf(stderr, "[mdk hip] host threads inside the device library: %s\n", pt); }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] plan open %.3fs, device ready at %.3fs, uploader: wait-for-chunk %.3fs wait-for-reference %.3fs wait-for-group %.3fs submit %.3fs wait-for-uploads %.3fs; collector: download %.3fs emit %.3fs, total %.3fs; chunks prepared on the host after all: %d\n", t_open, t_dev, w_next, w_ref, w_group, w_sub, w_rel, X->w_down, X->w_emit, now_s() - T0, X->n_host_prep);
    if(ret == 0) mdk_plan_finish(p);
    if(getenv("MDK_HOST_PROFILE")) { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[mdk main] leaving at epoch %.3f (resident %.0f MB, of which file-backed/shared %.0f MB)\n", ts.tv_sec + 1e-9 * ts.tv_nsec, rss_mb(0), rss_mb(1)); }
    if(fast_exit_wanted()) leave_fast_plan(p, ret);
    { double tc = now_s(), td;
      pthread_mutex_destroy(&X->mu); pthread_cond_destroy(&X->cv); free(X->ref_state); free(X);
      mdk_plan_detach_device(p);
      md_dev_close(dev); td = now_s();
      mdk_plan_close(p);
      if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] device closed in %.3fs, plan (slabs, reference, mapped file) in %.3fs\n", td - tc, now_s() - td); }
    return ret;
}

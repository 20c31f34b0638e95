/*
 * mdk_extract.c -- host side of the MI355X `MethylDackel extract` path.
 *
 * Division of labour:
 *   host  : BGZF inflate + BAM record framing (mdk_io.c), read admission (the flag/tag/MAPQ tests of
 *           filter_func, common.c:416-444), strand determination (getStrand, common.c:84-116), the
 *           qname pairing that htslib's constructor/destructor callbacks perform (overlaps.c:121-147),
 *           packing into the SoA batch of include/mdk_hip.h, the reference's chunk schedule
 *           (extract.c:325-350, common.c:466-493) and the text post-pass (extract.c:443-510, 39-99).
 *   device: everything per base -- trimming, overlap resolution, context classification, counting.
 * There is no CPU implementation of the per-base work in this library.
 */
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
    fflush(stdout); fflush(stderr);
    _exit(ret & 0xff);
}
/* the HIP runtime takes 0.1-0.4 s to come up: start that before anything else (options, BAM header, FASTA), on its own thread */
static void *hipwarm_main(void *arg) { (void)arg; (void)md_dev_warm(getenv("MDK_DEVICE") ? atoi(getenv("MDK_DEVICE")) : 0); return NULL; }
/* only in the `MethylDackel` command, which always ends with _exit (leave_fast): a library caller whose bad command line makes
 * us return at once must not find a half-initialised runtime racing its exit handlers */
MDK_LOCAL void hip_warm_up(void) {
    pthread_t th;
    if(!fast_exit_wanted() || pthread_create(&th, NULL, hipwarm_main, NULL)) return;
    pthread_detach(th);
}
/* md_dev_last_error is per thread: keep the text of a failed open for the thread that reports it */
MDK_LOCAL void *devopen_main(void *arg) { devopen_t *d = arg; d->rc = md_dev_open(d->device, &d->cfg, &d->dev); if(d->rc) snprintf(d->err, sizeof(d->err), "%s", md_dev_last_error()); return NULL; }

/* Chunks travel to the device in GROUPS of up to MDK_GROUP: a 1 Mb chunk alone is fewer than two workgroups per CU and pays every
 * launch boundary itself, so whatever the reader has ready when a group is opened (at least one chunk, at most eight) is uploaded
 * into the group's slots and prepared and piled up with one launch per kernel (md_dev_launch_group); while that runs, the next
 * group is assembled, then the finished one is collected chunk by chunk in schedule order and handed to the emitter.  The
 * reference's unit of work is the chunk (extract.c:325-350); here it is the unit of scheduling and of output only. */
#define MDK_GROUP 8
typedef struct { mdk_chunk ch[MDK_GROUP]; int slot[MDK_GROUP]; int n, launched[MDK_GROUP]; } cgroup;

int extract_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; cgroup *G = NULL; int rc, ret = 0, more = 1, cur = 0, i; devopen_t dop; pthread_t dth; int dth_ok; emitter em;
    double T0 = now_s(), t_open, t_dev, w_next = 0, w_sub = 0, w_down = 0, w_emit = 0, ta; int n_host_prep = 0; uint64_t n_groups = 0, n_chunks = 0;
    if(getenv("MDK_HOST_PROFILE")) { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[mdk main] entered at epoch %.3f\n", ts.tv_sec + 1e-9 * ts.tv_nsec); }
    { int rk = 0, wd = 1, m = ranks_from_env(&rk, &wd); if(m < 0) return -1; if(m > 0) return extract_ranks(argc, argv, rk, wd); }       /* one process per GPU (mdk_ranks.c) */
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open(argc, argv, &p);
    t_open = now_s() - T0;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] resident after plan open %.0f MB\n", rss_mb(0));
    if(rc != 0 || !p) return rc;
    /* HIP initialisation takes a few hundred ms: do it while the host pipeline already inflates and packs */
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    dop.cfg.n_slots = 2 * MDK_GROUP;
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    /* the per-record work of a chunk (admission, strand, name pairing, CIGAR expansion) runs on the device; MDK_HOST_PREP=1 keeps
     * it on the host's chunk workers (the round-1 arrangement, and what a chunk the device gives up on falls back to) */
    if(!getenv("MDK_HOST_PREP")) mdk_plan_set_prep(p, 1);
    mdk_plan_set_hold(p, 2 * MDK_GROUP + 1);
    dth_ok = pthread_create(&dth, NULL, devopen_main, &dop) == 0;       /* no thread: open the device here, after the pipeline has started */
    if(!p->started && pipeline_start(p)) { if(dth_ok) pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    if(dth_ok) pthread_join(dth, NULL); else devopen_main(&dop);
    t_dev = now_s() - T0;
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] resident at device ready %.0f MB\n", rss_mb(0));
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `extract`.\n", dop.device, dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    if(p->dev_prep) { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); mdk_plan_attach_device(p, dev); }      /* from here on the device inflates pieces of the file too */
    G = calloc(2, sizeof(cgroup));
    if(!G || emitter_start(&em, p, emit_threads(p))) { free(G); mdk_plan_detach_device(p); md_dev_close(dev); mdk_plan_close(p); return -5; }
    for(i = 0; i < MDK_GROUP; i++) { G[0].slot[i] = i; G[1].slot[i] = MDK_GROUP + i; }
    while(more || G[0].n || G[1].n) {
        cgroup *g = &G[cur], *o = &G[cur ^ 1];
        /* open a group: the first chunk is waited for, the others are taken only if they are ready now */
        g->n = 0;
        while(more && g->n < MDK_GROUP) {
            mdk_chunk *c = &g->ch[g->n];
            ta = now_s();
            rc = g->n == 0 ? mdk_plan_next_chunk(p, c) : mdk_plan_try_next_chunk(p, c);
            w_next += now_s() - ta;
            if(rc == 2) break;
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) { more = 0; break; }
            g->launched[g->n] = 0;
            if(!c->skipped) {
                ta = now_s();
                rc = mdk_plan_ensure_reference(p, dev, c->tid);
                if(!rc) rc = c->prep ? md_dev_upload_raw(dev, g->slot[g->n], &c->raw) : md_dev_upload(dev, g->slot[g->n], &c->batch);
                w_sub += now_s() - ta;
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                g->launched[g->n] = 1;
            }
            g->n++; n_chunks++;
        }
        if(ret) break;
        {   /* one launch per kernel for the group's chunks */
            int ls[MDK_GROUP], nl = 0;
            for(i = 0; i < g->n; i++) if(g->launched[i]) ls[nl++] = g->slot[i];
            if(nl) { ta = now_s(); rc = md_dev_launch_group(dev, ls, nl); w_sub += now_s() - ta; n_groups++; if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; } }
        }
        /* collect the group before it, in schedule order */
        for(i = 0; i < o->n && !ret; i++) {
            md_sites sites; memset(&sites, 0, sizeof(sites));
            if(o->launched[i]) {
                ta = now_s();
                rc = md_dev_download(dev, o->slot[i], &sites);
                if(rc == MDK_ERR_PREP_HOST) {          /* a read name the device preparation does not handle: this chunk the slow way */
                    rc = mdk_plan_host_prepare_from(p, &o->ch[i], dev, o->slot[i]);
                    if(!rc) rc = md_dev_submit(dev, o->slot[i], &o->ch[i].batch);
                    if(!rc) rc = md_dev_download(dev, o->slot[i], &sites);
                    n_host_prep++;
                }
                w_down += now_s() - ta;
                if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            }
            ta = now_s();
            if(emitter_push(&em, &o->ch[i], &sites)) { ret = em.failed ? MDK_RC_OUTPUT : MDK_RC_DEVICE; break; }
            w_emit += now_s() - ta;
        }
        o->n = 0;
        if(ret) break;
        cur ^= 1;
    }
    { double tw = now_s(); emitter_stop(&em); w_emit += now_s() - tw; }
    if(em.failed && !ret) ret = MDK_RC_OUTPUT;
    if(getenv("MDK_HOST_PROFILE")) { double rs = 0; uint64_t rc2 = 0, rb = 0; md_host_profile(&rs, &rc2, &rb); fprintf(stderr, "[mdk main] staging blocks registered: %" PRIu64 " (%.0f MB) in %.3fs; %" PRIu64 " chunks in %" PRIu64 " group launches\n", rc2, rb / 1048576.0, rs, n_chunks, n_groups); }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] plan open %.3fs, device ready at %.3fs, loop: wait-for-chunk %.3fs submit %.3fs download %.3fs emit %.3fs, total %.3fs; chunks prepared on the host after all: %d\n", t_open, t_dev, w_next, w_sub, w_down, w_emit, now_s() - T0, n_host_prep);
    if(ret == 0) mdk_plan_finish(p);
    if(getenv("MDK_HOST_PROFILE")) { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); fprintf(stderr, "[mdk main] leaving at epoch %.3f (resident %.0f MB, of which file-backed/shared %.0f MB)\n", ts.tv_sec + 1e-9 * ts.tv_nsec, rss_mb(0), rss_mb(1)); }
    if(fast_exit_wanted()) leave_fast(ret);
    { double tc = now_s(), td;
      free(G);
      mdk_plan_detach_device(p);
      md_dev_close(dev); td = now_s();
      mdk_plan_close(p);
      if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] device closed in %.3fs, plan (slabs, reference, mapped file) in %.3fs\n", td - tc, now_s() - td); }
    return ret;
}

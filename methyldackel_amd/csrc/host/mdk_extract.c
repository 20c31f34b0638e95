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
MDK_LOCAL int fast_exit_wanted(void) {
    const char *pre = getenv("LD_PRELOAD");
    if(!getenv("MDK_FAST_EXIT")) return 0;
    if(getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCPROF_OUTPUT_PATH")) return 0;
    if(pre && (strstr(pre, "rocprof") || strstr(pre, "roctx") || strstr(pre, "rocm"))) return 0;
    return 1;
}
/* leave now: outputs are flushed and closed.  When the `MethylDackel` command runs the work in a child process (main.c),
 * MDK_DONE_FD names the pipe on which the parent waits for the result: it is told first, and the standard streams are
 * closed, so that nobody waits for the kernel to unpin ~1 GB of staging buffers and tear the GPU context down. */
MDK_LOCAL void leave_fast(int ret) {
    const char *fd = getenv("MDK_DONE_FD");
    fflush(stdout); fflush(stderr);
    if(fd) { int f = atoi(fd), rc = ret; if(f > 2 && write(f, &rc, sizeof(rc)) == (ssize_t)sizeof(rc)) { close(f); close(0); close(1); close(2); } }
    _exit(ret & 0xff);
}
/* the HIP runtime takes 0.1-0.4 s to come up: start that before anything else (options, BAM header, FASTA), on its own thread */
static void *hipwarm_main(void *arg) { (void)arg; (void)md_dev_warm(getenv("MDK_DEVICE") ? atoi(getenv("MDK_DEVICE")) : 0); return NULL; }
/* only in the `MethylDackel` command, which always ends with _exit (leave_fast): a library caller whose bad command line makes
 * us return at once must not find a half-initialised runtime racing its exit handlers */
MDK_LOCAL void hip_warm_up(void) {
    pthread_t th;
    if(!fast_exit_wanted() || pthread_create(&th, NULL, hipwarm_main, NULL)) return;
    if(getenv("MDK_INIT_FIRST")) pthread_join(th, NULL); else pthread_detach(th);      /* experiment: runtime first, inflate threads afterwards */
}
/* md_dev_last_error is per thread: keep the text of a failed open for the thread that reports it */
MDK_LOCAL void *devopen_main(void *arg) { devopen_t *d = arg; d->rc = md_dev_open(d->device, &d->cfg, &d->dev); if(d->rc) snprintf(d->err, sizeof(d->err), "%s", md_dev_last_error()); return NULL; }

int extract_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch[2]; int have[2] = {0, 0}; int rc, k = 0, ret = 0, more = 1; devopen_t dop; pthread_t dth; int dth_ok; emitter em;
    double T0 = now_s(), t_open, t_dev, w_next = 0, w_sub = 0, w_down = 0, w_emit = 0, ta; int n_host_prep = 0;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open(argc, argv, &p);
    t_open = now_s() - T0;
    if(rc != 0 || !p) return rc;
    /* HIP initialisation takes a few hundred ms: do it while the host pipeline already inflates and packs */
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    /* the per-record work of a chunk (admission, strand, name pairing, CIGAR expansion) runs on the device; MDK_HOST_PREP=1 keeps
     * it on the host's chunk workers (the round-1 arrangement, and what a chunk the device gives up on falls back to) */
    if(!getenv("MDK_HOST_PREP")) mdk_plan_set_prep(p, 1);
    dth_ok = pthread_create(&dth, NULL, devopen_main, &dop) == 0;       /* no thread: open the device here, after the pipeline has started */
    if(!p->started && pipeline_start(p)) { if(dth_ok) pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    if(dth_ok) pthread_join(dth, NULL); else devopen_main(&dop);
    t_dev = now_s() - T0;
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `extract`.\n", dop.device, dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    if(p->dev_prep) { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); }
    if(emitter_start(&em, p, p->o.n_threads >= 8 ? 8 : p->o.n_threads)) { md_dev_close(dev); mdk_plan_close(p); return -5; }
    /* two chunks in flight: build+submit chunk k while chunk k-1 finishes on the device, then hand k-1 to the emitter */
    while(more || have[0] || have[1]) {
        int cur = k & 1, prev = cur ^ 1;
        if(more) {
            ta = now_s();
            rc = mdk_plan_next_chunk(p, &ch[cur]);
            w_next += now_s() - ta;
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) more = 0;
            else {
                if(!ch[cur].skipped) {
                    ta = now_s();
                    rc = mdk_plan_ensure_reference(p, dev, ch[cur].tid);
                    if(!rc) rc = ch[cur].prep ? md_dev_submit_raw(dev, cur, &ch[cur].raw) : md_dev_submit(dev, cur, &ch[cur].batch);
                    w_sub += now_s() - ta;
                    if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                }
                have[cur] = 1;
            }
        }
        if(have[prev]) {
            md_sites sites; memset(&sites, 0, sizeof(sites));
            if(!ch[prev].skipped) {
                ta = now_s();
                rc = md_dev_download(dev, prev, &sites);
                if(rc == MDK_ERR_PREP_HOST) {          /* a read name the device preparation does not handle: this chunk the slow way */
                    rc = mdk_plan_host_prepare(p, &ch[prev]);
                    if(!rc) rc = md_dev_submit(dev, prev, &ch[prev].batch);
                    if(!rc) rc = md_dev_download(dev, prev, &sites);
                    n_host_prep++;
                }
                w_down += now_s() - ta;
                if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            }
            ta = now_s();
            if(emitter_push(&em, &ch[prev], &sites)) { ret = MDK_RC_DEVICE; break; }
            w_emit += now_s() - ta;
            have[prev] = 0;
        }
        k++;
        if(!more && !have[0] && !have[1]) break;
    }
    { double tw = now_s(); emitter_stop(&em); w_emit += now_s() - tw; }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk main] plan open %.3fs, device ready at %.3fs, loop: wait-for-chunk %.3fs submit %.3fs download %.3fs emit %.3fs, total %.3fs; chunks prepared on the host after all: %d\n", t_open, t_dev, w_next, w_sub, w_down, w_emit, now_s() - T0, n_host_prep);
    if(ret == 0) mdk_plan_finish(p);
    if(fast_exit_wanted()) leave_fast(ret);
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}


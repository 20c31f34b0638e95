/* mdk_cmd_perread.c -- the `perRead` command on the plan/pipeline of `extract` (see mdk_plan.h). */
#include "mdk_plan.h"

/* ------------------------------------------------------------------------------------------------ */
/* perRead (perRead.c): chunks without adjustBounds; the reads that start in a chunk and pass the    */
/* flag/MAPQ tests go to the device, which walks each CIGAR (k_perread); one text line per read.     */
/* ------------------------------------------------------------------------------------------------ */
static void perread_usage(void) {
    fputs("\nUsage: MethylDackel perRead [OPTIONS] <ref.fa> <input>\n", stderr);
    fputs("\nOutput columns: read name, chromosome, position, CpG methylation (%), number of informative bases.\n"
"Options (MI355X build; same option surface as MethylDackel 0.6.1):\n"
" -q INT, -p INT, -r STR, -l FILE, --keepStrand, -o STR, -F/--ignoreFlags INT (default 0),\n"
" -R/--requireFlags INT, -@ INT, --chunkSize INT, --version\n", stderr);
}

int mdk_plan_open_perread(int argc, char *argv[], mdk_plan **out) {
    static const struct option longopts[] = {            /* perRead.c:300-308; --ignoreNH is in the help text only */
        {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'}, {"chunkSize", required_argument, 0, 19}, {"keepStrand", no_argument, 0, 20},
        {"ignoreFlags", required_argument, 0, 'F'}, {"requireFlags", required_argument, 0, 'R'}, {0, 0, 0, 0}};
    mdk_plan *p; opts_t *o; int c;
    *out = NULL;
    p = calloc(1, sizeof(*p)); if(!p) return -5;
    o = &p->o;
    o->perread = 1;
    o->ctx_on[0] = 1; o->min_mapq = 10; o->min_phred = 5; o->min_depth = 1; o->ignore_flags = 0; o->n_threads = 1; o->chunk_size = 1000000;
    p->shard_rank = 0; p->shard_world = 1;
    p->last_tid = -1; p->last_pos = -1; p->carry_tid = -1;
    p->pr_out = stdout;
    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:o:@:r:l:F:R:", longopts, NULL)) >= 0) {
        switch(c) {
        case 'h': perread_usage(); plan_free(p); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MDK_VERSION, "none; methyldackel_amd MI355X build"); plan_free(p); return 0;
        case 'o':
            if(p->pr_out_owned) fclose(p->pr_out);
            if((p->pr_out = fopen(optarg, "w")) == NULL) { fprintf(stderr, "Couldn't open %s for writing\n", optarg); p->pr_out_owned = 0; plan_free(p); return 2; }
            p->pr_out_owned = 1;
            break;
        case 'q': o->min_mapq = atoi(optarg); break;
        case 'p': o->min_phred = atoi(optarg); break;
        case '@': o->n_threads = atoi(optarg); break;
        case 'r': o->region = optarg; break;
        case 'l': o->bed_name = optarg; break;
        case 'F': o->ignore_flags = atoi(optarg); break;
        case 'R': o->require_flags = atoi(optarg); break;
        case 19: o->chunk_size = strtoul(optarg, NULL, 10); if(o->chunk_size < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); plan_free(p); return 1; } break;
        case 20: o->keep_strand = 1; break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); perread_usage(); plan_free(p); return 1;
        }
    }
    if(argc == 1) { perread_usage(); plan_free(p); return 0; }
    if(argc - optind != 2) { fprintf(stderr, "You must supply a reference genome in fasta format and a BAM or CRAM file\n"); perread_usage(); plan_free(p); return -1; }
    if(o->min_phred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", o->min_phred); o->min_phred = 1; }
    if(o->min_mapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", o->min_mapq); o->min_mapq = 0; }
    /* the reference opens the FASTA first (-2 with the usage text), then the BAM (-4) (perRead.c:386-396) */
    { FILE *f = fopen(argv[optind], "r"); if(!f) { fprintf(stderr, "Couldn't open the index for %s!\n", argv[optind]); perread_usage(); plan_free(p); return -2; } fclose(f); }
    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}

int mdk_plan_emit_perread(mdk_plan *p, const mdk_chunk *c, const md_pr_count *counts, int64_t n) {
    const batchbuf *b; const char *chrom; int64_t i; char line[10000]; sbuf *ob;
    if(!p || !c || !p->o.perread) return -1;
    if(c->index != p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    p->next_emit++;
    if(c->skipped & ~MDK_CHUNK_NOREF) return 0;
    b = c->host;
    if(!b || (int64_t)b->n != c->pr.n_reads) return -2;
    if(counts && n != c->pr.n_reads) return -2;
    if(!counts && !(c->skipped & MDK_CHUNK_NOREF) && c->pr.n_reads) return -2;
    chrom = p->bam->target_name[c->tid];
    ob = &p->ob[0]; ob->l = 0;
    for(i = 0; i < c->pr.n_reads; i++) {             /* addRead, perRead.c:16-36 */
        uint32_t m = counts ? counts[i].nmeth : 0, u = counts ? counts[i].nunmeth : 0; int l;
        const char *qn = b->qn + b->ri[i].qn_off;
        if(m + u > 0) l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t%f\t%" PRIu32 "\n", qn, chrom, (int64_t)b->ri[i].pos, 100. * ((double)m) / (m + u), m + u);
        else l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t0.0\t%" PRIu32 "\n", qn, chrom, (int64_t)b->ri[i].pos, m + u);
        if(l >= (int)sizeof(line)) l = (int)sizeof(line) - 1;
        sb_put(ob, line, (size_t)l);
    }
    if(ob->l) fputs(ob->s, p->pr_out);
    ob->l = 0;
    return 0;
}

int mdk_plan_emit_perread_raw(mdk_plan *p, const mdk_chunk *c, const uint32_t *kept, const md_pr_count *counts, int64_t n) {
    const char *chrom; int64_t i; char line[10000]; sbuf *ob; int r = 0; uint64_t base = 0;
    if(!p || !c || !p->o.perread || !c->prep || (n && (!kept || !counts))) return -1;
    if(c->index != p->next_emit) { fprintf(stderr, "[mdk] chunks must be emitted in order\n"); return -2; }
    p->next_emit++;
    if(c->skipped) return 0;
    chrom = p->bam->target_name[c->tid];
    ob = &p->ob[0]; ob->l = 0;
    for(i = 0; i < n; i++) {                          /* addRead, perRead.c:16-36; the kept reads come in file order, so the range cursor only moves forward */
        uint32_t m = counts[i].nmeth, u = counts[i].nunmeth, off; const uint8_t *rec; int32_t pos; int l;
        if((int64_t)kept[i] >= c->raw.n_records) return -2;
        off = c->raw.rec_off[kept[i]];
        while(r < c->raw.n_ranges && (uint64_t)off >= base + c->raw.range[r].bytes) { base += c->raw.range[r].bytes; r++; }
        if(r >= c->raw.n_ranges) return -2;
        rec = c->raw.range[r].ptr + (off - base) + 4;
        memcpy(&pos, rec + 4, 4);
        if(m + u > 0) l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t%f\t%" PRIu32 "\n", (const char *)(rec + 32), chrom, (int64_t)pos, 100. * ((double)m) / (m + u), m + u);
        else l = snprintf(line, sizeof(line), "%s\t%s\t%" PRId64 "\t0.0\t%" PRIu32 "\n", (const char *)(rec + 32), chrom, (int64_t)pos, m + u);
        if(l >= (int)sizeof(line)) l = (int)sizeof(line) - 1;
        sb_put(ob, line, (size_t)l);
    }
    if(ob->l) fputs(ob->s, p->pr_out);
    ob->l = 0;
    return 0;
}

int perRead_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch[2]; int have[2] = {0, 0}; int rc, k = 0, ret = 0, more = 1; devopen_t dop; pthread_t dth; int dth_ok;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open_perread(argc, argv, &p);
    if(rc != 0 || !p) return rc;
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    if(!getenv("MDK_HOST_PREP")) mdk_plan_set_prep(p, 1);       /* the device selects the reads (perRead.c:178-183) and walks them where they lie in the records */
    dth_ok = pthread_create(&dth, NULL, devopen_main, &dop) == 0;       /* no thread: open the device here, after the pipeline has started */
    if(!p->started && pipeline_start(p)) { if(dth_ok) pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    if(dth_ok) pthread_join(dth, NULL); else devopen_main(&dop);
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `perRead`.\n", user_device(dop.device), dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    if(p->dev_prep) { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); }
    while(more || have[0] || have[1]) {       /* two chunks in flight, as in extract_main */
        int cur = k & 1, prev = cur ^ 1;
        if(more) {
            rc = mdk_plan_next_chunk(p, &ch[cur]);
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) more = 0;
            else {
                if(ch[cur].prep && !ch[cur].skipped) {
                    rc = mdk_plan_ensure_reference(p, dev, ch[cur].tid);
                    if(!rc) rc = md_dev_perread_submit_raw(dev, cur, &ch[cur].raw);
                    if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                } else if(!ch[cur].skipped && ch[cur].pr.n_reads) {
                    rc = mdk_plan_ensure_reference(p, dev, ch[cur].tid);
                    if(!rc) rc = md_dev_perread_submit(dev, cur, &ch[cur].pr);
                    if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                }
                have[cur] = 1;
            }
        }
        if(have[prev]) {
            const md_pr_count *cnt = NULL; int64_t n = 0;
            if(ch[prev].prep && !ch[prev].skipped) {
                const uint32_t *kept = NULL;
                rc = md_dev_perread_download_raw(dev, prev, &kept, &cnt, &n);
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                if(mdk_plan_emit_perread_raw(p, &ch[prev], kept, cnt, n)) { ret = MDK_RC_DEVICE; break; }
                have[prev] = 0; k++;
                if(!more && !have[0] && !have[1]) break;
                continue;
            }
            if(!ch[prev].skipped && ch[prev].pr.n_reads) {
                rc = md_dev_perread_download(dev, prev, &cnt, &n);
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            }
            if(mdk_plan_emit_perread(p, &ch[prev], cnt, n)) { ret = MDK_RC_DEVICE; break; }
            have[prev] = 0;
        }
        k++;
        if(!more && !have[0] && !have[1]) break;
    }
    fflush(p->pr_out);
    if(fast_exit_wanted()) { if(p->pr_out_owned) fclose(p->pr_out); leave_fast_plan(p, ret); }
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}


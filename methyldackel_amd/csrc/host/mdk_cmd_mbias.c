/* mdk_cmd_mbias.c -- the `mbias` command on the plan/pipeline of `extract` (see mdk_plan.h; report: mdk_mbias.c). */
#include "mdk_plan.h"

/* ------------------------------------------------------------------------------------------------ */
/* mbias (MBias.c): the same schedule, admission and segments; the device accumulates a histogram    */
/* over (strand, read number, position in read) across all chunks, which is read back once.          */
/* ------------------------------------------------------------------------------------------------ */
static void mbias_usage(void) {
    fputs("\nUsage: MethylDackel mbias [OPTIONS] <ref.fa> <sorted_alignments.bam> <output.prefix>\n", stderr);
    fputs("\nOptions (MI355X build; same option surface as MethylDackel 0.6.1):\n"
" -q INT, -p INT, -D INT(ignored), -r STR, -l FILE, --keepStrand, -@ INT, --chunkSize INT,\n"
" --keepDupes, --keepSingleton, --keepDiscordant, -F/--ignoreFlags INT, -R/--requireFlags INT,\n"
" --ignoreNH, --minConversionEfficiency FLOAT, --txt, --noSVG (implies --txt; no prefix needed),\n"
" --noCpG, --CHG, --CHH, --nOT/--nOB/--nCTOT/--nCTOB INT,INT,INT,INT, --version\n", stderr);
}

int mdk_plan_open_mbias(int argc, char *argv[], mdk_plan **out) {
    enum { M_NOCPG = 1, M_CHG, M_CHH, M_KEEPDUPES, M_KEEPSINGLETON, M_KEEPDISCORDANT, M_TXT, M_NOSVG, M_NOT, M_NOB, M_NCTOT, M_NCTOB,
           M_CHUNKSIZE, M_KEEPSTRAND, M_MINCONVEFF, M_IGNORENH };
    static const struct option longopts[] = {            /* MBias.c:330-352 */
        {"noCpG", no_argument, 0, M_NOCPG}, {"CHG", no_argument, 0, M_CHG}, {"CHH", no_argument, 0, M_CHH}, {"keepDupes", no_argument, 0, M_KEEPDUPES},
        {"keepSingleton", no_argument, 0, M_KEEPSINGLETON}, {"keepDiscordant", no_argument, 0, M_KEEPDISCORDANT}, {"txt", no_argument, 0, M_TXT},
        {"noSVG", no_argument, 0, M_NOSVG}, {"nOT", required_argument, 0, M_NOT}, {"nOB", required_argument, 0, M_NOB}, {"nCTOT", required_argument, 0, M_NCTOT},
        {"nCTOB", required_argument, 0, M_NCTOB}, {"chunkSize", required_argument, 0, M_CHUNKSIZE}, {"keepStrand", no_argument, 0, M_KEEPSTRAND},
        {"minConversionEfficiency", required_argument, 0, M_MINCONVEFF}, {"ignoreNH", no_argument, 0, M_IGNORENH},
        {"ignoreFlags", required_argument, 0, 'F'}, {"requireFlags", required_argument, 0, 'R'}, {"help", no_argument, 0, 'h'}, {"version", no_argument, 0, 'v'},
        {0, 0, 0, 0}};
    mdk_plan *p; opts_t *o; int c;
    *out = NULL;
    p = calloc(1, sizeof(*p)); if(!p) return -5;
    o = &p->o;
    o->mbias = 1; o->svg = 1;
    o->ctx_on[0] = 1; o->min_mapq = 10; o->min_phred = 5; o->min_depth = 1; o->ignore_flags = 0xF00; o->n_threads = 1; o->chunk_size = 1000000;
    p->shard_rank = 0; p->shard_world = 1;
    p->last_tid = -1; p->last_pos = -1; p->carry_tid = -1;
    optind = 1;
    while((c = getopt_long(argc, argv, "hvq:p:r:l:D:F:@:", longopts, NULL)) >= 0) {      /* NB no R: in the short options (MBias.c:353) */
        switch(c) {
        case 'h': mbias_usage(); plan_free(p); return 0;
        case 'v': printf("%s (using HTSlib version %s)\n", MDK_VERSION, "none; methyldackel_amd MI355X build"); plan_free(p); return 0;
        case 'D': break;
        case 'r': o->region = optarg; break;
        case 'l': o->bed_name = optarg; break;
        case M_NOCPG: o->ctx_on[0] = 0; break;
        case M_CHG: o->ctx_on[1] = 1; break;
        case M_CHH: o->ctx_on[2] = 1; break;
        case M_KEEPDUPES: o->keep_dupes = 1; break;       /* unlike extract, 0x400 stays in ignoreFlags, so this alone changes nothing */
        case M_KEEPSINGLETON: o->keep_singleton = 1; break;
        case M_KEEPDISCORDANT: o->keep_discordant = 1; break;
        case M_TXT: o->txt = 1; break;
        case M_NOSVG: o->svg = 0; o->txt = 1; break;
        case M_NOT: case M_NOB: case M_NCTOT: case M_NCTOB: parse_bounds(optarg, o->abs_bounds + 4 * (c - M_NOT)); break;
        case M_CHUNKSIZE: o->chunk_size = strtoul(optarg, NULL, 10); if(o->chunk_size < 1) { fprintf(stderr, "Error: The chunk size must be at least 1!\n"); plan_free(p); return 1; } break;
        case M_KEEPSTRAND: o->keep_strand = 1; break;
        case M_MINCONVEFF: o->min_conv_eff = (float)atof(optarg); break;
        case M_IGNORENH: o->ignore_nh = 1; break;
        case 'F': o->ignore_flags = atoi(optarg); break;
        case 'R': o->require_flags = atoi(optarg); break;
        case 'q': o->min_mapq = atoi(optarg); break;
        case 'p': o->min_phred = atoi(optarg); break;
        case '@': o->n_threads = atoi(optarg); break;
        default: fprintf(stderr, "Invalid option '%c'\n", c); mbias_usage(); plan_free(p); return 1;
        }
    }
    if(argc == 1) { mbias_usage(); plan_free(p); return 0; }
    if((o->svg && argc - optind != 3) || (!o->svg && argc - optind < 2)) {
        fprintf(stderr, "You must supply a reference genome in fasta format, an input BAM file, and an output prefix!!!\n");
        mbias_usage(); plan_free(p); return -1;
    }
    if(o->min_phred < 1) { fprintf(stderr, "-p %i is invalid. resetting to 1, which is the lowest possible value.\n", o->min_phred); o->min_phred = 1; }
    if(o->min_mapq < 0) { fprintf(stderr, "-q %i is invalid. Resetting to 0, which is the lowest possible value.\n", o->min_mapq); o->min_mapq = 0; }
    if(!(o->ctx_on[0] + o->ctx_on[1] + o->ctx_on[2])) {
        fprintf(stderr, "You haven't specified any metrics to output!\nEither don't use the --noCpG option or specify --CHG and/or --CHH.\n");
        plan_free(p); return -1;
    }
    if(o->svg) o->mb_opref = argv[optind + 2];
    { int rc = plan_attach_inputs(p, argv, optind); if(rc) return rc; }
    *out = p;
    return 0;
}
int mdk_plan_mbias_outputs(const mdk_plan *p, const char **opref, int *svg, int *txt, int *which) {
    if(!p || !p->o.mbias) return -1;
    if(opref) *opref = p->o.mb_opref;
    if(svg) *svg = p->o.svg;
    if(txt) *txt = p->o.txt;
    if(which) *which = p->o.ctx_on[0] + 2 * p->o.ctx_on[1] + 4 * p->o.ctx_on[2];
    return 0;
}

int mbias_main(int argc, char *argv[]) {
    mdk_plan *p = NULL; md_dev *dev = NULL; mdk_chunk ch; int rc, k = 0, ret = 0; devopen_t dop; pthread_t dth; int dth_ok; md_mbias hist;
    if(argc > 2) hip_warm_up();
    rc = mdk_plan_open_mbias(argc, argv, &p);
    if(rc != 0 || !p) return rc;
    memset(&dop, 0, sizeof(dop));
    mdk_plan_dev_cfg(p, &dop.cfg);
    if(getenv("MDK_DEVICE")) dop.device = atoi(getenv("MDK_DEVICE"));
    if(!getenv("MDK_HOST_PREP")) mdk_plan_set_prep(p, 1);       /* admission, strand and CIGAR expansion on the device, as in extract (no pairing: MBias.c:158-161) */
    dth_ok = pthread_create(&dth, NULL, devopen_main, &dop) == 0;       /* no thread: open the device here, after the pipeline has started */
    if(!p->started && pipeline_start(p)) { if(dth_ok) pthread_join(dth, NULL); if(dop.dev) md_dev_close(dop.dev); mdk_plan_close(p); return -5; }
    if(dth_ok) pthread_join(dth, NULL); else devopen_main(&dop);
    dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `mbias`.\n", user_device(dop.device), dop.err); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    if(p->dev_prep) { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); }
    if(p->dev_prep) mdk_plan_attach_device(p, dev);      /* from here on the device inflates pieces of the file too, as in extract */
    for(;; k++) {
        /* chunk k goes to slot k&1; the batch handed out two calls ago is recycled by the next call, so its upload must be over.  A submit queues
         * the chunk's upload and preparation and meanwhile sends the other slot's chunk -- whose preparation has reported by then -- on to the
         * histogram kernel (md_dev_mbias_submit_raw), so the records of chunk k cross the link while chunk k-1 is counted */
        if((rc = md_dev_slot_sync(dev, k & 1)) != 0) {
            if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }       /* (the slot's deferred histogram step reports here when the chunks in between were skipped) */
            fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break;
        }
        rc = mdk_plan_next_chunk(p, &ch);
        if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
        if(rc == 0) break;
        if(ch.skipped & MDK_CHUNK_NOREF) { ret = -4; break; }        /* the reference's worker gives up here and its caller then crashes (MBias.c:150-155,543) */
        if(ch.skipped) continue;
        rc = mdk_plan_ensure_reference(p, dev, ch.tid);
        if(!rc) rc = ch.prep ? md_dev_mbias_submit_raw(dev, k & 1, &ch.raw) : md_dev_mbias_submit(dev, k & 1, &ch.batch);
        if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
        if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
    }
    if(ret == 0) {
        rc = md_dev_mbias_read(dev, &hist);
        if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
        if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; }
        else if(mdk_mbias_report(&hist, p->o.mb_opref, p->o.svg, p->o.txt, p->o.ctx_on[0] + 2 * p->o.ctx_on[1] + 4 * p->o.ctx_on[2])) ret = -3;
    }
    if(fast_exit_wanted()) leave_fast_plan(p, ret);
    mdk_plan_detach_device(p);
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}


/* main.c -- `MethylDackel` command of the MI355X build: `extract`, `mbias`, `perRead` on the GPU and the `mergeContext`
 * text tool (the reference's dispatcher is main.c:39-62). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "mdk_extract.h"

static void usage_main(void) {
    fprintf(stderr, "MethylDackel (methyldackel_amd, MI355X build of the `extract` path)\n"
                    "Usage: MethylDackel <command> [options]\n\nCommands:\n"
                    "    extract  Extract methylation metrics from an alignment file in BAM format (GPU).\n"
                    "    mbias    Determine the position-dependent methylation bias in a dataset (GPU).\n"
                    "    perRead  Generate a per-read methylation summary (GPU).\n"
                    "    mergeContext   Combine single Cytosine metrics from 'MethylDackel extract' into per-CpG/CHG metrics.\n");
}
int main(int argc, char *argv[]) {
    if(argc == 1) { usage_main(); return 0; }
    if(!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage_main(); return 0; }
    if(!strcmp(argv[1], "-v") || !strcmp(argv[1], "--version")) { printf("0.6.1 (using HTSlib version none; methyldackel_amd MI355X build)\n"); return 0; }
    if(!strcmp(argv[1], "extract") || !strcmp(argv[1], "mbias") || !strcmp(argv[1], "perRead")) {
        /* stay on the CPUs next to the GPU (the inflate threads first-touch what the GPU uploads from); a rank of a several-GPU run
         * (csrc/host/mdk_ranks.c) takes the device its rank names */
        const char *dv = getenv("MDK_DEVICE"), *lr = getenv("LOCAL_RANK"), *mr = getenv("MDK_RANK");
        (void)mdk_bind_to_device_node(dv ? atoi(dv) : lr ? atoi(lr) : mr ? atoi(mr) : 0);
    }
    {   /* a process about to end need not unpin buffers and shut the runtime down politely (MDK_FAST_EXIT: the command leaves with
         * _exit once its outputs are closed).  A command that comes BACK here returned early -- a bad option, a missing input -- possibly
         * while the thread that warms the HIP runtime up is still inside hipInit: leave with _exit as well, so that no exit handler or
         * static destructor runs under that thread's feet and the command's own return code is what the caller sees. */
        int (*cmd)(int, char **) = !strcmp(argv[1], "extract") ? extract_main : !strcmp(argv[1], "mbias") ? mbias_main : !strcmp(argv[1], "perRead") ? perRead_main : NULL;
        if(cmd) {
            int rc;
            setenv("MDK_FAST_EXIT", "1", 0);
            rc = cmd(argc - 1, argv + 1);
            fflush(stdout); fflush(stderr);
            _exit(rc & 0xff);
        }
    }
    if(!strcmp(argv[1], "mergeContext")) return mergeContext_main(argc - 1, argv + 1);
    fprintf(stderr, "Unknown command!\n"); usage_main(); return -1;
}

/* main.c -- `MethylDackel` command of the MI355X build: `extract`, `mbias`, `perRead` on the GPU and the `mergeContext`
 * text tool (the reference's dispatcher is main.c:39-62). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
    if(!strcmp(argv[1], "extract")) {
        setenv("MDK_FAST_EXIT", "1", 0);      /* a process about to end need not unpin buffers and shut the runtime down politely */
        return extract_main(argc - 1, argv + 1);
    }
    if(!strcmp(argv[1], "mbias")) { setenv("MDK_FAST_EXIT", "1", 0); return mbias_main(argc - 1, argv + 1); }
    if(!strcmp(argv[1], "perRead")) { setenv("MDK_FAST_EXIT", "1", 0); return perRead_main(argc - 1, argv + 1); }
    if(!strcmp(argv[1], "mergeContext")) return mergeContext_main(argc - 1, argv + 1);
    fprintf(stderr, "Unknown command!\n"); usage_main(); return -1;
}

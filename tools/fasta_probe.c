/* fasta_probe.c -- TEST/MEASUREMENT TOOL: loads a FASTA through csrc/host/mdk_fasta.c (MDK_FASTA_THREADS decides how) and prints, per contig,
 * name, length and a hash of its bytes, and the time the load took.   fasta_probe ref.fa */
#include <stdio.h>
#include <stdint.h>
#include <time.h>
#include "../methyldackel_amd/csrc/host/mdk_io.h"
int main(int argc, char **argv) {
    mdk_fasta fa; struct timespec a, b; int i;
    if(argc < 2) return 2;
    clock_gettime(CLOCK_MONOTONIC, &a);
    if(mdk_fasta_load(argv[1], &fa)) { fprintf(stderr, "load failed\n"); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &b);
    for(i = 0; i < fa.n; i++) { uint64_t h = 1469598103934665603ull; int64_t k; for(k = 0; k < fa.len[i]; k++) { h ^= (unsigned char)fa.seq[i][k]; h *= 1099511628211ull; } printf("%s\t%lld\t%016llx\n", fa.name[i], (long long)fa.len[i], (unsigned long long)h); }
    fprintf(stderr, "loaded in %.3f s\n", (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec));
    mdk_fasta_free(&fa);
    return 0;
}

/* piece_bench.c -- TEST/BENCH INFRASTRUCTURE for the device BGZF inflate (include/mdk_hip.h md_piece_*), through the C ABI.
 *   piece_bench file.bam [piece_MB=64] [in_flight=3] [verify=1]
 * Splits the file into pieces of whole BGZF members, stages each piece in pinned memory, runs it through md_piece_submit / wait
 * with `in_flight` pieces queued, and reports: wall-clock of the whole file (staging copy + H2D + kernels + D2H of the digests),
 * the kernels alone on a resident piece (HIP events), and -- verify=1 -- whether every inflated byte
 * equals zlib's, every record offset equals the host's walk and every digest equals what csrc/host/mdk_io.c note_records leaves.
 * build: make tools   (gcc, links libmdk_hip.so) */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include "mdk_hip.h"

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct { size_t file_off, file_end; int m0, m1; uint64_t out_bytes; } piece_t;

int main(int argc, char **argv) {
    if(argc < 2) { fprintf(stderr, "usage: piece_bench file.bam [piece_MB] [in_flight] [verify]\n"); return 2; }
    const size_t piece_bytes = (size_t)(argc > 2 ? atof(argv[2]) : 64.0) * (1u << 20);
    int in_flight = argc > 3 ? atoi(argv[3]) : 3; const int verify = argc > 4 ? atoi(argv[4]) : 1;
    if(in_flight < 1) in_flight = 1;
    if(in_flight > 8) in_flight = 8;
    FILE *f = fopen(argv[1], "rb"); if(!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *raw = malloc(n + 64); if(!raw || fread(raw, 1, n, f) != n) return 2; fclose(f);
    /* members of the whole file; in_off is relative to the start of the piece a member ends up in (fixed below) */
    int nm = 0, cap = 1 << 16; md_inf_member *mem = malloc(sizeof(*mem) * cap); size_t *mfile = malloc(sizeof(size_t) * cap);
    for(size_t o = 0; o + 18 <= n;) {
        const uint32_t xlen = raw[o + 10] | (raw[o + 11] << 8), bs = (raw[o + 16] | (raw[o + 17] << 8)) + 1u;
        if(nm == cap) { cap *= 2; mem = realloc(mem, sizeof(*mem) * cap); mfile = realloc(mfile, sizeof(size_t) * cap); }
        mem[nm].in_off = o + 12 + xlen; mem[nm].in_len = bs - 12 - xlen - 8; mem[nm].out_len = le32(raw + o + bs - 4); mem[nm].crc32 = le32(raw + o + bs - 8); mem[nm].reserved = 0; mem[nm].out_off = 0; mfile[nm] = o; nm++;
        o += bs;
    }
    piece_t *pc = malloc(sizeof(piece_t) * (nm + 1)); int np = 0; uint64_t tot_out = 0;
    for(int i = 0; i < nm;) {
        piece_t *p = &pc[np++]; p->m0 = i; p->file_off = mfile[i]; p->out_bytes = 0;
        while(i < nm && (mfile[i] - p->file_off < piece_bytes || i == p->m0)) { mem[i].in_off -= p->file_off; mem[i].out_off = p->out_bytes; p->out_bytes += mem[i].out_len; i++; }
        p->m1 = i; p->file_end = i < nm ? mfile[i] : n; tot_out += p->out_bytes;
    }
    printf("{\"file_MB\": %.1f, \"members\": %d, \"inflated_MB\": %.1f, \"pieces\": %d, \"piece_MB\": %.0f, \"in_flight\": %d", n / 1048576.0, nm, tot_out / 1048576.0, np, piece_bytes / 1048576.0, in_flight);
    md_dev_cfg cfg; memset(&cfg, 0, sizeof(cfg)); cfg.keepCpG = 1; cfg.minPhred = 5;
    md_dev *dev = NULL; double t0 = now();
    if(md_dev_open(0, &cfg, &dev)) { fprintf(stderr, "md_dev_open: %s\n", md_dev_last_error()); return 1; }
    printf(", \"dev_open_s\": %.3f", now() - t0);
    md_piece *P[8]; uint8_t *stage[8]; size_t stage_cap = 0;
    for(int i = 0; i < np; i++) if(pc[i].file_end - pc[i].file_off > stage_cap) stage_cap = pc[i].file_end - pc[i].file_off;
    t0 = now();
    for(int k = 0; k < in_flight; k++) { if(md_piece_create(dev, &P[k])) { fprintf(stderr, "md_piece_create: %s\n", md_dev_last_error()); return 1; } stage[k] = md_host_alloc(stage_cap + 64); if(!stage[k]) return 1; }
    printf(", \"pinned_alloc_s\": %.3f, \"pinned_MB\": %.0f", now() - t0, in_flight * (stage_cap / 1048576.0));
    /* whole file, twice (the first pass grows the device buffers) */
    long bad_bytes = 0, bad_rec = 0, bad_dig = 0; uint64_t n_rec_total = 0;
    for(int pass = 0; pass < 2; pass++) {
        double t_copy = 0; t0 = now();
        for(int i = 0; i < np + in_flight; i++) {
            const int k = i % in_flight;
            if(i >= in_flight) {       /* collect piece i - in_flight, which used this slot */
                md_piece_info info; const piece_t *q = &pc[i - in_flight];
                if(md_piece_wait(P[k], &info)) { fprintf(stderr, "md_piece_wait: %s\n", md_dev_last_error()); return 1; }
                if(pass == 1 && verify) {
                    uint8_t *got = malloc(q->out_bytes + 64), *ref = malloc(q->out_bytes + 64); uint32_t *ro = malloc(sizeof(uint32_t) * (info.n_records + 1));
                    md_piece_read(P[k], 0, q->out_bytes, got); md_piece_read_records(P[k], 0, info.n_records, ro);
                    uint32_t r = 0;
                    for(int m = q->m0; m < q->m1; m++) {
                        const md_inf_member *M = &mem[m]; const md_inf_digest *D = &info.digest[m - q->m0];
                        if(M->out_len) {
                            z_stream zs; memset(&zs, 0, sizeof zs); zs.next_in = raw + q->file_off + M->in_off; zs.avail_in = M->in_len; zs.next_out = ref + M->out_off; zs.avail_out = M->out_len;
                            inflateInit2(&zs, -15); if(inflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "zlib failed\n"); return 1; } inflateEnd(&zs);
                            if(memcmp(ref + M->out_off, got + M->out_off, M->out_len)) bad_bytes++;
                        }
                        /* the host's walk of the member (mdk_io.c note_records) */
                        const uint8_t *d = ref + M->out_off; uint32_t L = M->out_len, o = 0, cnt = 0; int ok = 1, sorted = 1; int32_t tid0 = -1, pos0 = -1, tidN = -1, posN = -1, mn = 0x7fffffff, mx = (int32_t)0x80000000;
                        while(o + 4 <= L) {
                            const uint32_t bs = le32(d + o); const uint8_t *rr = d + o + 4;
                            if(bs < 32 || (uint64_t)o + 4 + bs > L) { ok = 0; break; }
                            const uint32_t lq = rr[8], nc = rr[12] | (rr[13] << 8); if(32u + lq + 4u * nc > bs) { ok = 0; break; }
                            int32_t rl = 0; for(uint32_t c = 0; c < nc; c++) { const uint32_t v = le32(rr + 32 + lq + 4 * c), op = v & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(v >> 4); }
                            const int32_t tid = (int32_t)le32(rr), pos = (int32_t)le32(rr + 4), endp = pos + (rl > 0 ? rl : 1);
                            if(cnt == 0) { tid0 = tid; pos0 = pos; } else if(tid < 0 || tid < tidN || (tid == tidN && pos < posN)) sorted = 0;
                            if(tid < 0) sorted = 0;
                            tidN = tid; posN = pos; if(endp < mn) mn = endp; if(endp > mx) mx = endp;
                            if(ok && (D->first_rec + cnt >= info.n_records || ro[D->first_rec + cnt] != M->out_off + o)) bad_rec++;
                            cnt++; o += 4 + bs;
                        }
                        ok = ok && o == L;
                        if(ok) { if(!D->ok || D->n_rec != cnt || D->first_rec != r || (cnt && (D->tid0 != tid0 || D->pos0 != pos0 || D->tidN != tidN || D->posN != posN || D->min_endp != mn || D->max_endp != mx || D->sorted != sorted))) bad_dig++; r += cnt; }
                        else if(D->ok) bad_dig++;
                    }
                    if(r != info.n_records) bad_dig++;
                    n_rec_total += info.n_records;
                    free(got); free(ref); free(ro);
                }
            }
            if(i < np) {
                const piece_t *q = &pc[i]; double tc = now();
                memcpy(stage[k], raw + q->file_off, q->file_end - q->file_off); t_copy += now() - tc;
                if(md_piece_submit(P[k], stage[k], q->file_end - q->file_off, mem + q->m0, q->m1 - q->m0)) { fprintf(stderr, "md_piece_submit: %s\n", md_dev_last_error()); return 1; }
            }
        }
        const double dt = now() - t0;
        if(pass == 1 && verify) printf(", \"verified\": {\"members_with_wrong_bytes\": %ld, \"wrong_record_offsets\": %ld, \"wrong_digests\": %ld, \"records\": %llu}", bad_bytes, bad_rec, bad_dig, (unsigned long long)n_rec_total);
        else printf(", \"pass%d\": {\"seconds\": %.4f, \"staging_copy_s\": %.4f, \"GBps_compressed\": %.2f, \"GBps_inflated\": %.2f}", pass, dt, t_copy, n / dt / 1e9, tot_out / dt / 1e9);
    }
    /* a timed pass without verification when the second one verified */
    if(verify) {
        t0 = now();
        for(int i = 0; i < np + in_flight; i++) {
            const int k = i % in_flight; md_piece_info info;
            if(i >= in_flight && md_piece_wait(P[k], &info)) return 1;
            if(i < np) { const piece_t *q = &pc[i]; memcpy(stage[k], raw + q->file_off, q->file_end - q->file_off); if(md_piece_submit(P[k], stage[k], q->file_end - q->file_off, mem + q->m0, q->m1 - q->m0)) return 1; }
        }
        const double dt = now() - t0;
        printf(", \"pass2\": {\"seconds\": %.4f, \"GBps_compressed\": %.2f, \"GBps_inflated\": %.2f}", dt, n / dt / 1e9, tot_out / dt / 1e9);
    }
    /* kernels alone on the largest resident piece */
    { int big = 0; for(int i = 1; i < np; i++) if(pc[i].m1 - pc[i].m0 > pc[big].m1 - pc[big].m0) big = i;
      const piece_t *q = &pc[big]; const size_t cb = q->file_end - q->file_off;
      printf(", \"kernel_only\": {\"piece_members\": %d, \"piece_comp_MB\": %.1f, \"piece_out_MB\": %.1f", q->m1 - q->m0, cb / 1048576.0, q->out_bytes / 1048576.0);
      memcpy(stage[0], raw + q->file_off, cb);
      {   /* (the decoder variant is chosen when the library is built: make B=dir HIPFLAGS=-DINF_VARIANT=n) */
          const int var = 0;
          md_piece *X; md_piece_info info; float a = 0, b = 0;
          if(md_piece_create(dev, &X) || md_piece_submit(X, stage[0], cb, mem + q->m0, q->m1 - q->m0) || md_piece_wait(X, &info)) { fprintf(stderr, "variant %d: %s\n", var, md_dev_last_error()); return 1; }
          if(md_piece_bench(X, 5, &a, &b)) { fprintf(stderr, "md_piece_bench: %s\n", md_dev_last_error()); return 1; }
          int same = 1;
          if(verify) { uint8_t *got = malloc(q->out_bytes + 64), *ref = malloc(q->out_bytes + 64); md_piece_read(X, 0, q->out_bytes, got);
              for(int m = q->m0; m < q->m1 && same; m++) { const md_inf_member *M = &mem[m]; if(!M->out_len) continue; z_stream zs; memset(&zs, 0, sizeof zs); zs.next_in = raw + q->file_off + M->in_off; zs.avail_in = M->in_len; zs.next_out = ref + M->out_off; zs.avail_out = M->out_len;
                  inflateInit2(&zs, -15); inflate(&zs, Z_FINISH); inflateEnd(&zs); if(memcmp(ref + M->out_off, got + M->out_off, M->out_len)) same = 0; }
              free(got); free(ref); }
          float c = 0; if(md_piece_bench_crc(X, 5, &c)) { fprintf(stderr, "md_piece_bench_crc: %s\n", md_dev_last_error()); if(verify) return 1; c = -1; }      /* (verify=0: experiment builds whose output is wrong on purpose are still timed) */
          printf(", \"v%d\": {\"inflate_ms\": %.3f, \"walk_ms\": %.3f, \"crc32_ms\": %.3f, \"comp_bytes\": %zu, \"out_bytes\": %llu, \"GBps_compressed\": %.2f, \"GBps_inflated\": %.2f, \"identical_to_zlib\": %d}", var, a, b, c, cb, (unsigned long long)q->out_bytes, cb / (a * 1e6), q->out_bytes / (a * 1e6), same);
          md_piece_destroy(X);
      }
      printf("}"); }
    printf("}\n");
    for(int k = 0; k < in_flight; k++) { md_piece_destroy(P[k]); md_host_free(stage[k]); }
    md_dev_close(dev);
    return (bad_bytes || bad_rec || bad_dig) ? 1 : 0;
}

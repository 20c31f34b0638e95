/* feed_harness.c -- TEST INFRASTRUCTURE: the BAM feed of the host pipeline (csrc/host/mdk_io.c: host inflate teams, device inflate
 * teams, the reorder buffer, slab accounting) exercised WITHOUT a GPU.  The reader under test is compiled in as it is; what it calls in
 * the device library (md_piece_*, md_host_*) is replaced by stand-ins in which "device memory" is host memory, k_inflate is zlib and
 * k_walk is the host's own record walk -- so the members, digests and record tables a piece hands back are what the real kernels hand
 * back (tests/test_gpu_inflate.py checks those against exactly this), and everything around them is the product's code.
 *
 *   feed_harness file.bam MODE HOLD [threads]
 *     MODE 0  host teams only (the reference pass)
 *     MODE 1  device teams attached right after the header (hybrid: both kinds of team share the piece counter)
 *     HOLD    the consumer keeps the last HOLD slabs referenced, like chunks that are still in flight
 *   prints: records, bytes and an order-sensitive 64-bit digest of every record (block_size word + body) in stream order, and how many
 *   pieces each side inflated.  Two runs over one file must print the same digest; a run that stands still is killed by the test's timeout.
 *   Small pieces and few slabs (MDK_GPU_PIECE_MB=0.25, MDK_SLAB_CAP=2) with a large HOLD make the scanner hold more slabs than the
 *   inflaters may allocate ahead, which is the situation that stopped the 128 Mb run in round 3. */
#define _GNU_SOURCE
#include "../methyldackel_amd/csrc/host/mdk_io.c"      /* the reader under test, statics and all */

/* ---- stand-ins for libmdk_hip ---- */
struct md_dev { int unused; };
struct md_piece { uint8_t *out; uint64_t out_cap; uint32_t *rec; uint64_t rec_cap; md_inf_digest *dig; int dig_cap; md_piece_info info; };
const char *md_dev_last_error(void) { return "feed_harness stand-in"; }
void *md_host_alloc(uint64_t bytes) { return malloc((size_t)bytes + 64); }
void md_host_free(void *p) { free(p); }
void md_host_register(md_dev *h, const void *ptr) { (void)h; (void)ptr; }
void md_dev_reserve_hint(uint64_t device_bytes) { (void)device_bytes; }
int md_piece_members_per_round(md_dev *h) { (void)h; return getenv("MDK_STANDIN_MEMBERS_PER_ROUND") ? atoi(getenv("MDK_STANDIN_MEMBERS_PER_ROUND")) : 0; }      /* (0: pieces are cut by bytes) */
int md_piece_create(md_dev *h, md_piece **out) { (void)h; *out = calloc(1, sizeof(**out)); return *out ? 0 : -6; }
void md_piece_destroy(md_piece *p) { if(!p) return; free(p->out); free(p->rec); free(p->dig); free(p); }
int md_piece_submit(md_piece *p, const uint8_t *comp, uint64_t comp_bytes, const md_inf_member *mem, int32_t n_mem) {
    uint64_t total = 0, nrec = 0; int i; sumbuf sb; memset(&sb, 0, sizeof(sb));
    for(i = 0; i < n_mem; i++) total += mem[i].out_len;
    if(p->out_cap < total + 64) { free(p->out); p->out_cap = total + (total >> 3) + 64; p->out = malloc(p->out_cap); }
    if(p->dig_cap < n_mem) { free(p->dig); p->dig_cap = n_mem + 64; p->dig = malloc(sizeof(md_inf_digest) * (size_t)p->dig_cap); }
    if(!p->out || !p->dig) return -6;
    for(i = 0; i < n_mem; i++) {
        blk_t b; md_inf_digest *g = &p->dig[i]; z_stream zs;
        memset(&b, 0, sizeof(b)); memset(g, 0, sizeof(*g));
        if(mem[i].in_off + mem[i].in_len > comp_bytes || mem[i].out_off + mem[i].out_len > total) return -3;
        b.out = p->out + mem[i].out_off; b.out_len = mem[i].out_len;
        g->first_rec = (uint32_t)sb.n; g->ok = 0;
        if(!b.out_len) { g->ok = 1; g->sorted = 1; continue; }      /* an empty member (the EOF marker) holds no record and ends where it starts */
        memset(&zs, 0, sizeof(zs));
        if(inflateInit2(&zs, -15) != Z_OK) return -1;
        zs.next_in = (Bytef *)(comp + mem[i].in_off); zs.avail_in = mem[i].in_len; zs.next_out = b.out; zs.avail_out = b.out_len;
        if(inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); return -1; }
        inflateEnd(&zs);
        note_records(&b, &sb, p->out);
        g->first_rec = (uint32_t)b.sum0;
        if(b.ok) { g->n_rec = b.n_sum; g->tid0 = b.tid0; g->pos0 = b.pos0; g->tidN = b.tidN; g->posN = b.posN; g->min_endp = b.min_endp; g->max_endp = b.max_endp; g->ok = 1; g->sorted = b.sorted; }
        else sb.n = b.sum0;                                         /* a member that is not ok contributes no records to the table */
    }
    nrec = sb.n;
    if(p->rec_cap < nrec + 1) { free(p->rec); p->rec_cap = nrec + (nrec >> 3) + 64; p->rec = malloc(sizeof(uint32_t) * p->rec_cap); if(!p->rec) return -6; }
    for(uint64_t k = 0; k < nrec; k++) p->rec[k] = sb.v[k].off;
    free(sb.v);
    p->info.n_mem = n_mem; p->info.digest = p->dig; p->info.n_records = (uint32_t)nrec; p->info.out_bytes = total; p->info.d_out = p->out; p->info.d_rec_off = p->rec;
    return 0;
}
int md_piece_wait(md_piece *p, md_piece_info *info) { *info = p->info; return 0; }
int md_piece_read(md_piece *p, uint64_t off, uint64_t bytes, uint8_t *dst) { if(off + bytes > p->info.out_bytes) return -3; memcpy(dst, p->out + off, (size_t)bytes); return 0; }
int md_piece_read_records(md_piece *p, uint32_t first, uint32_t n, uint32_t *dst) { if((uint64_t)first + n > p->info.n_records) return -3; memcpy(dst, p->rec + first, sizeof(uint32_t) * (size_t)n); return 0; }

/* ---- the consumer: every record in stream order ---- */
static uint64_t mix(uint64_t h, const uint8_t *p, size_t n) { for(size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ULL; return h; }

int main(int argc, char **argv) {
    if(argc < 4) { fprintf(stderr, "usage: feed_harness file.bam MODE HOLD [threads]\n"); return 2; }
    const int mode = atoi(argv[2]), hold = atoi(argv[3]), threads = argc > 4 ? atoi(argv[4]) : 8;
    mdk_bam *b = mdk_bam_open(argv[1], threads);
    static struct md_dev dev;
    if(!b) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    if(mode == 1 && mdk_bam_attach_device(b, &dev, 3)) { fprintf(stderr, "attach failed\n"); return 2; }
    uint64_t h = 0xcbf29ce484222325ULL, n = 0, bytes = 0, dev_members = 0;
    mdk_slab **held = calloc((size_t)hold + 1, sizeof(*held)); int nheld = 0; mdk_slab *last = NULL;
    for(;;) {
        mdk_rsum q; const uint8_t *raw; mdk_slab *s = NULL; int mi = 0, on_device;
        int rc = mdk_bam_at_device(b, &s, &mi);
        if(rc < 0) { fprintf(stderr, "error: %s\n", b->err); return 1; }
        on_device = rc == 1;
        if(on_device) {                                            /* a member of a slab inflated "on the device": its records from the piece's tables */
            const mdk_member *m = &s->mem[mi];
            for(uint32_t k = 0; k < m->n_sum; k++) {
                const uint8_t *r = s->d_buf + s->d_rec_off[m->sum0 + k]; uint32_t bs = le32(r);
                h = mix(h, r, 4 + (size_t)bs); n++; bytes += 4 + bs;
            }
            dev_members++;
        } else {
            rc = mdk_bam_peek_sum(b, &q, &raw);
            if(rc < 0) { fprintf(stderr, "error: %s\n", b->err); return 1; }
            if(rc == 0) break;
            if(rc == 2) continue;                                  /* (a device slab has just become current) */
            h = mix(h, raw - 4, 4 + (size_t)q.len); n++; bytes += 4 + q.len;
            { size_t off; s = mdk_bam_cur_slab(b, &off); }
        }
        if(s && s != last) {                                       /* a new slab: keep it referenced for a while, as a chunk in flight does */
            mdk_slab_ref(b, s); held[nheld++] = s; last = s;
            if(nheld > hold) { mdk_slab_unref(b, held[0]); memmove(held, held + 1, sizeof(*held) * (size_t)(nheld - 1)); nheld--; }
        }
        if(on_device) mdk_bam_dev_advance(b); else mdk_bam_advance_sum(b, &q);
    }
    for(int i = 0; i < nheld; i++) mdk_slab_unref(b, held[i]);
    printf("{\"records\": %llu, \"bytes\": %llu, \"digest\": \"%016llx\", \"host_pieces\": %llu, \"device_pieces\": %llu, \"device_members\": %llu, \"read_back\": %llu, \"spec_redo\": %llu}\n",
           (unsigned long long)n, (unsigned long long)bytes, (unsigned long long)h, (unsigned long long)b->n_host_pieces, (unsigned long long)b->n_dev_pieces, (unsigned long long)dev_members, (unsigned long long)b->n_materialized, (unsigned long long)b->n_spec_redo);
    if(mode == 1) mdk_bam_detach_device(b);
    mdk_bam_close(b);
    free(held);
    return 0;
}

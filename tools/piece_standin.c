/* piece_standin.c -- TEST INFRASTRUCTURE: the device library's BGZF pieces (md_piece_*, include/mdk_hip.h) without a device, as a library
 * to LD_PRELOAD in front of libmdk_hip.so: "device memory" is host memory, k_inflate is zlib, k_walk is a plain walk over block_size words.
 * With it the WHOLE host pipeline of `extract` in device-inflate mode (mdk_plan_attach_device: device teams, device slabs taken member by
 * member from their digests, chunks whose ranges point into "device" memory) runs on a CPU-only box and can be compared, chunk by chunk,
 * with the host-only pipeline (tests/test_raw_batch.py).  What the real kernels hand back for a piece is compared with zlib, with the
 * host walk and member by member on the GPU (tests/test_gpu_inflate.py); this stand-in returns the same things by construction.
 *   build: gcc -O2 -shared -fPIC -Iinclude -o tools/_build/libmdk_piece_standin.so tools/piece_standin.c -lz */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "mdk_hip.h"

struct md_piece { uint8_t *out; uint64_t out_cap; uint32_t *rec; uint64_t rec_cap, n_rec; md_inf_digest *dig; int dig_cap; md_piece_info info; };
static uint32_t u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

int md_piece_members_per_round(md_dev *h) { (void)h; return getenv("MDK_STANDIN_MEMBERS_PER_ROUND") ? atoi(getenv("MDK_STANDIN_MEMBERS_PER_ROUND")) : 0; }      /* (0: pieces are cut by bytes) */
int md_piece_create(md_dev *h, md_piece **out) { (void)h; *out = calloc(1, sizeof(**out)); return *out ? 0 : -6; }
void md_piece_destroy(md_piece *p) { if(!p) return; free(p->out); free(p->rec); free(p->dig); free(p); }
void md_host_register(md_dev *h, const void *ptr) { (void)h; (void)ptr; }
static int push(md_piece *p, uint32_t off) {
    if(p->n_rec == p->rec_cap) { p->rec_cap = p->rec_cap ? p->rec_cap * 2 : 4096; p->rec = realloc(p->rec, sizeof(uint32_t) * p->rec_cap); if(!p->rec) return -1; }
    p->rec[p->n_rec++] = off; return 0;
}
int md_piece_submit(md_piece *p, const uint8_t *comp, uint64_t comp_bytes, const md_inf_member *mem, int32_t n_mem) {
    uint64_t total = 0; int i;
    for(i = 0; i < n_mem; i++) total += mem[i].out_len;
    if(p->out_cap < total + 64) { free(p->out); p->out_cap = total + (total >> 3) + 64; p->out = malloc(p->out_cap); }
    if(p->dig_cap < n_mem) { free(p->dig); p->dig_cap = n_mem + 64; p->dig = malloc(sizeof(md_inf_digest) * (size_t)p->dig_cap); }
    if(!p->out || !p->dig) return -6;
    p->n_rec = 0;
    for(i = 0; i < n_mem; i++) {
        md_inf_digest *g = &p->dig[i]; uint8_t *d = p->out + mem[i].out_off; const uint32_t L = mem[i].out_len; uint32_t o = 0; const uint64_t first = p->n_rec; z_stream zs;
        memset(g, 0, sizeof(*g)); g->first_rec = (uint32_t)first; g->sorted = 1; g->min_endp = 0x7fffffff; g->max_endp = (int32_t)0x80000000; g->tid0 = g->tidN = g->pos0 = g->posN = -1;
        if(mem[i].in_off + mem[i].in_len > comp_bytes || mem[i].out_off + L > total) return -3;
        if(!L) { g->ok = 1; continue; }
        memset(&zs, 0, sizeof(zs));
        if(inflateInit2(&zs, -15) != Z_OK) return -1;
        zs.next_in = (Bytef *)(comp + mem[i].in_off); zs.avail_in = mem[i].in_len; zs.next_out = d; zs.avail_out = L;
        if(inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); return -1; }
        inflateEnd(&zs);
        if((uint32_t)crc32(0L, d, L) != mem[i].crc32) return -1;       /* as the device's k_crc32 */
        while(o + 4 <= L) {                                         /* the member's records, from its first byte */
            const uint32_t bs = u32(d + o); const uint8_t *r = d + o + 4; uint32_t lq, nc, k; int32_t rl = 0, tid, pos, endp;
            if(bs < 32 || (uint64_t)o + 4 + bs > L) break;
            lq = r[8]; nc = (uint32_t)r[12] | (uint32_t)r[13] << 8;
            if(32u + lq + 4u * nc > bs) break;
            for(k = 0; k < nc; k++) { const uint32_t v = u32(r + 32 + lq + 4 * k), op = v & 15; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += (int32_t)(v >> 4); }
            tid = (int32_t)u32(r); pos = (int32_t)u32(r + 4); endp = pos + (rl > 0 ? rl : 1);
            if(push(p, (uint32_t)(d + o - p->out))) return -6;
            if(g->n_rec == 0) { g->tid0 = tid; g->pos0 = pos; }
            else if(tid < 0 || tid < g->tidN || (tid == g->tidN && pos < g->posN)) g->sorted = 0;
            if(tid < 0) g->sorted = 0;
            g->tidN = tid; g->posN = pos;
            if(endp < g->min_endp) g->min_endp = endp;
            if(endp > g->max_endp) g->max_endp = endp;
            g->n_rec++; o += 4 + bs;
        }
        g->ok = o == L;
        if(!g->ok) { p->n_rec = first; g->n_rec = 0; }              /* a member that does not start and end on record boundaries has no place in the table */
    }
    p->info.n_mem = n_mem; p->info.digest = p->dig; p->info.n_records = (uint32_t)p->n_rec; p->info.out_bytes = total; p->info.d_out = p->out; p->info.d_rec_off = p->rec;
    return 0;
}
int md_piece_wait(md_piece *p, md_piece_info *info) { *info = p->info; return 0; }
int md_piece_read(md_piece *p, uint64_t off, uint64_t bytes, uint8_t *dst) { if(off + bytes > p->info.out_bytes) return -3; memcpy(dst, p->out + off, (size_t)bytes); return 0; }
int md_piece_read_records(md_piece *p, uint32_t first, uint32_t n, uint32_t *dst) { if((uint64_t)first + n > p->info.n_records) return -3; memcpy(dst, p->rec + first, sizeof(uint32_t) * (size_t)n); return 0; }

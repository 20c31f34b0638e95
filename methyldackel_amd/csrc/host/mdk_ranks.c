/* mdk_ranks.c -- `MethylDackel extract` as ONE PROCESS PER GPU: the multi-GPU form of the command.
 *
 * The path shards by interval with no data dependence between intervals: chunk k of the reference's schedule (extract.c:325-350)
 * belongs to rank k mod N.  Every rank opens the same inputs, seeks to its own chunks through the index (so the BGZF inflate and the
 * uploads shard with the GPUs: csrc/host/mdk_pipeline.c reader_fill), computes them on its GPU, and the per-interval site buffers
 * travel to rank 0, which alone writes the files, in schedule order (the reference's ordered flush, extract.c:514-535).
 *
 * Launch: N processes with MDK_RANK / MDK_WORLD (or, with MDK_TORCHRUN=1, torchrun's RANK / WORLD_SIZE), MASTER_ADDR (default 127.0.0.1) and MASTER_PORT;
 * LOCAL_RANK (or MDK_DEVICE) picks the GPU.  tools/extract_ranks.sh starts them on one node.
 *   control channel  one TCP connection from every rank to rank 0: the bootstrap (who sits on which device, the RCCL id) and, per
 *                    chunk, a 24-byte header with the sizes and the chunk's status;
 *   data channel     RCCL over xGMI (ncclSend / ncclRecv of the device-resident site records, variant evidence and tile segments:
 *                    libmdk_hip md_comm_result_*) when every rank has a GPU of its own; when two ranks share a physical device --
 *                    a single-GPU box, where RCCL refuses duplicate devices: the tests -- the sender downloads the chunk and the
 *                    ordered site records follow the header over the TCP connection.
 * A rank never needs anything from another rank to compute: there is no collective anywhere on the data path.
 *
 * MDK_CLAIM=1 (every rank): instead of k mod N the ranks CLAIM chunks as they get to them, as the reference's worker threads do under
 * positionMutex (extract.c:325-350): rank 0 keeps the counter and hands out the next chunk index over a second connection per rank, so a
 * rank whose chunks are deep gets fewer of them.  Rank 0 learns who computed a chunk from the counter's log. */
#include "mdk_plan.h"
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>

#include <poll.h>
typedef struct { int rank, world, *fd, *cfd, claim; } ranks_t;       /* rank 0: fd[r] = connection to rank r; the others: fd[0] = connection to rank 0; cfd: the same for the claims (MDK_CLAIM=1) */

static int wr_all(int fd, const void *b, size_t n) { const char *p = b; while(n) { ssize_t k = send(fd, p, n, MSG_NOSIGNAL); if(k <= 0) { if(k < 0 && errno == EINTR) continue; return -1; } p += k; n -= (size_t)k; } return 0; }
static int rd_all(int fd, void *b, size_t n) { char *p = b; while(n) { ssize_t k = recv(fd, p, n, 0); if(k <= 0) { if(k < 0 && errno == EINTR) continue; return -1; } p += k; n -= (size_t)k; } return 0; }

static void ranks_close(ranks_t *R) { int i; if(!R->fd) return; for(i = 0; i < R->world; i++) { if(R->fd[i] >= 0) close(R->fd[i]); if(R->cfd && R->cfd[i] >= 0) close(R->cfd[i]); } free(R->fd); free(R->cfd); R->fd = R->cfd = NULL; }
static int ranks_open(ranks_t *R) {
    const char *addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1"; const int port = getenv("MDK_PORT") ? atoi(getenv("MDK_PORT")) : getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) + (getenv("MDK_WORLD") ? 0 : 1) : 29517;      /* under torchrun MASTER_PORT is the launcher's own store: the ranks meet one port above it */
    int i, one = 1;
    R->fd = malloc(sizeof(int) * (size_t)R->world); R->cfd = malloc(sizeof(int) * (size_t)R->world); if(!R->fd || !R->cfd) return -1;
    for(i = 0; i < R->world; i++) R->fd[i] = R->cfd[i] = -1;
    if(R->rank == 0) {
        struct sockaddr_in a; int ls = socket(AF_INET, SOCK_STREAM, 0);
        if(ls < 0) return -1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        { struct timeval tv; tv.tv_sec = 120; tv.tv_usec = 0; setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)); }      /* accept() gives up on a peer that never shows */
        memset(&a, 0, sizeof(a)); a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
        if(bind(ls, (struct sockaddr *)&a, sizeof(a)) || listen(ls, R->world)) { fprintf(stderr, "[mdk] rank 0 cannot listen on port %d: %s\n", port, strerror(errno)); close(ls); return -1; }
        for(i = 1; i < (R->claim ? 2 : 1) * (R->world - 1) + 1; i++) {
            int c = accept(ls, NULL, NULL), r = -1, *slot;
            if(c < 0 || rd_all(c, &r, sizeof(r)) || (r & 0xffff) < 1 || (r & 0xffff) >= R->world || *(slot = (r & 0x10000) ? &R->cfd[r & 0xffff] : &R->fd[r & 0xffff]) >= 0 || ((r & 0x10000) && !R->claim)) { fprintf(stderr, "[mdk] rank 0: a peer failed to introduce itself\n"); if(c >= 0) close(c); close(ls); return -1; }
            setsockopt(c, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            *slot = c;
        }
        close(ls);
    } else {
        struct addrinfo hints, *res = NULL; char ps[16]; int tries, c = -1;
        memset(&hints, 0, sizeof(hints)); hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM; snprintf(ps, sizeof(ps), "%d", port);
        if(getaddrinfo(addr, ps, &hints, &res) || !res) { fprintf(stderr, "[mdk] rank %d cannot resolve %s\n", R->rank, addr); return -1; }
        for(int pass = 0; pass < (R->claim ? 2 : 1); pass++) {
            const int tag = R->rank | (pass ? 0x10000 : 0);
            for(tries = 0, c = -1; tries < 600; tries++) {          /* rank 0 may still be starting: up to a minute */
                c = socket(AF_INET, SOCK_STREAM, 0);
                if(c >= 0 && connect(c, res->ai_addr, res->ai_addrlen) == 0) break;
                if(c >= 0) close(c);
                c = -1; usleep(100000);
            }
            if(c < 0 || wr_all(c, &tag, sizeof(int))) { fprintf(stderr, "[mdk] rank %d cannot reach rank 0 at %s:%d\n", R->rank, addr, port); if(c >= 0) close(c); freeaddrinfo(res); return -1; }
            setsockopt(c, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            if(pass) R->cfd[0] = c; else R->fd[0] = c;
        }
        freeaddrinfo(res);
    }
    return 0;
}

/* which ranks mode the environment asks for: MDK_WORLD/MDK_RANK, or -- only when MDK_TORCHRUN=1 says that this command IS the torchrun worker --
 * torchrun's WORLD_SIZE/RANK.  (A command that merely inherits a torchrun worker's environment, e.g. started by one, runs alone.) */
MDK_LOCAL int ranks_from_env(int *rank, int *world) {
    const char *w = getenv("MDK_WORLD"), *r = getenv("MDK_RANK");
    if(!w && getenv("MDK_TORCHRUN") && getenv("WORLD_SIZE") && getenv("RANK") && getenv("MASTER_PORT") && !getenv("MDK_NO_RANKS")) { w = getenv("WORLD_SIZE"); r = getenv("RANK"); }
    if(!w || atoi(w) < 2) return 0;
    *world = atoi(w); *rank = r ? atoi(r) : 0;
    if(*rank < 0 || *rank >= *world || *world > 1024) { fprintf(stderr, "[mdk] bad rank %d of %d\n", *rank, *world); return -1; }
    return 1;
}

/* ---- MDK_CLAIM=1: chunks are claimed, not dealt ---- */
typedef struct {
    ranks_t *R; int have; uint32_t granted;                       /* every rank: the chunk index this rank holds a claim on */
    pthread_mutex_t mu; pthread_cond_t cv; uint32_t next; int32_t *owner; uint32_t cap_owner; pthread_t th; int th_ok, quit;       /* rank 0: the counter and its log */
} claimer;
static uint32_t grant(claimer *Q, int rank) {
    uint32_t g;
    pthread_mutex_lock(&Q->mu);
    g = Q->next++;
    if(g >= Q->cap_owner) { const uint32_t nc = Q->cap_owner ? Q->cap_owner * 2 : 4096; Q->owner = xrealloc(Q->owner, sizeof(int32_t) * nc); Q->cap_owner = nc; }
    Q->owner[g] = rank;
    pthread_cond_broadcast(&Q->cv);
    pthread_mutex_unlock(&Q->mu);
    return g;
}
/* called by the plan's reader thread for every chunk of the schedule, in order: is it this rank's?  A rank holds one claim at a time; the
 * chunks in front of it have been claimed by others (the counter only grows). */
static int claim_chunk(void *ctx, uint32_t index) {
    claimer *Q = ctx; ranks_t *R = Q->R;
    if(!Q->have) {
        if(R->rank == 0) Q->granted = grant(Q, 0);
        else { uint32_t ask = index; if(wr_all(R->cfd[0], &ask, 4) || rd_all(R->cfd[0], &Q->granted, 4)) Q->granted = 0xffffffffu; }      /* rank 0 is gone: nothing is ours any more (the run fails at the next result) */
        Q->have = 1;
    }
    if(index < Q->granted) return 0;
    Q->have = 0;
    return 1;
}
static void *dispenser_main(void *arg) {          /* rank 0: answers the other ranks' claims */
    claimer *Q = arg; ranks_t *R = Q->R; struct pollfd *pf = xcalloc((size_t)R->world, sizeof(*pf)); int i, open = R->world - 1;
    for(i = 1; i < R->world; i++) { pf[i].fd = R->cfd[i]; pf[i].events = POLLIN; }
    while(open > 0) {
        int q; pthread_mutex_lock(&Q->mu); q = Q->quit; pthread_mutex_unlock(&Q->mu);
        if(q) break;
        if(poll(pf + 1, (nfds_t)(R->world - 1), 200) <= 0) continue;
        for(i = 1; i < R->world; i++) if(pf[i].fd >= 0 && (pf[i].revents & (POLLIN | POLLHUP | POLLERR))) {
            uint32_t ask, g;
            if(rd_all(pf[i].fd, &ask, 4)) { pf[i].fd = -1; open--; continue; }      /* the rank has walked the whole schedule (or died: its missing result says so) */
            g = grant(Q, i);
            if(wr_all(pf[i].fd, &g, 4)) { pf[i].fd = -1; open--; }
        }
    }
    free(pf);
    return NULL;
}
static int owner_wait(claimer *Q, uint32_t index) {       /* rank 0: who claimed chunk `index` (waits until somebody has) */
    int r;
    pthread_mutex_lock(&Q->mu);
    while(index >= Q->next && !Q->quit) pthread_cond_wait(&Q->cv, &Q->mu);
    r = index < Q->next ? Q->owner[index] : -1;
    pthread_mutex_unlock(&Q->mu);
    return r;
}

typedef struct { md_site *site; md_site_var *var; int64_t cap; } hostbuf;
static int hb_need(hostbuf *h, int64_t n, int variant) {
    if(n <= h->cap) return 0;
    free(h->site); free(h->var); h->cap = n + n / 4 + 1024;
    h->site = malloc(sizeof(md_site) * (size_t)h->cap); h->var = variant ? malloc(sizeof(md_site_var) * (size_t)h->cap) : NULL;
    if(!h->site || (variant && !h->var)) { h->cap = 0; return -1; }
    return 0;
}

/* a finished chunk of this rank (slot `sl`): the header, then the payload over the data channel */
static int send_result(ranks_t *R, md_dev *dev, md_comm *comm, mdk_plan *p, mdk_chunk *c, int sl, int *n_host_prep) {
    md_result_hdr h; int rc;
    if(comm) {
        rc = md_comm_result_header(dev, sl, &h); if(rc) return rc;
        if(h.rc == MDK_ERR_PREP_HOST) {               /* a read name the device preparation does not handle: this chunk the slow way, here, where its records are */
            rc = mdk_plan_host_prepare_from(p, c, dev, sl);
            if(!rc) rc = md_dev_submit(dev, sl, &c->batch);
            if(!rc) rc = md_comm_result_header(dev, sl, &h);
            if(rc) return rc;
            (*n_host_prep)++;
        }
        if(wr_all(R->fd[0], &h, sizeof(h))) return -1;
        rc = md_comm_result_send(comm, sl, &h); if(rc) return rc;
        return md_comm_wait(comm);                    /* the slot is uploaded into again next */
    } else {
        md_sites s; memset(&s, 0, sizeof(s));
        rc = md_dev_download(dev, sl, &s);
        if(rc == MDK_ERR_PREP_HOST) {
            rc = mdk_plan_host_prepare_from(p, c, dev, sl);
            if(!rc) rc = md_dev_submit(dev, sl, &c->batch);
            if(!rc) rc = md_dev_download(dev, sl, &s);
            (*n_host_prep)++;
        }
        memset(&h, 0, sizeof(h)); h.rc = rc; h.n_slots = rc ? 0 : s.n_sites; h.variant = s.var ? 1 : 0;
        if(wr_all(R->fd[0], &h, sizeof(h))) return -1;
        if(rc) return rc;
        if(s.n_sites && wr_all(R->fd[0], s.site, sizeof(md_site) * (size_t)s.n_sites)) return -1;
        if(s.n_sites && s.var && wr_all(R->fd[0], s.var, sizeof(md_site_var) * (size_t)s.n_sites)) return -1;
        return 0;
    }
}
static int recv_result(ranks_t *R, md_comm *comm, int src, hostbuf *hb, md_sites *out) {
    md_result_hdr h;
    memset(out, 0, sizeof(*out));
    if(rd_all(R->fd[src], &h, sizeof(h))) { fprintf(stderr, "[mdk] rank 0 lost the connection to rank %d\n", src); return -1; }
    if(h.rc) return h.rc;
    if(comm) return md_comm_result_recv(comm, src, &h, out);
    if(hb_need(hb, h.n_slots, h.variant)) return -5;
    if(h.n_slots && rd_all(R->fd[src], hb->site, sizeof(md_site) * (size_t)h.n_slots)) return -1;
    if(h.n_slots && h.variant && rd_all(R->fd[src], hb->var, sizeof(md_site_var) * (size_t)h.n_slots)) return -1;
    out->n_sites = h.n_slots; out->site = hb->site; out->var = h.variant ? hb->var : NULL;
    return 0;
}

MDK_LOCAL int extract_ranks(int argc, char *argv[], int rank, int world) {
    mdk_plan *p = NULL; md_dev *dev = NULL; md_comm *comm = NULL; ranks_t R; devopen_t dop; emitter em; int have_em = 0, rc, ret = 0, i, n_host_prep = 0;
    /* rank 0 keeps F chunks in flight (its own computing, the others' arriving) and hands them on in schedule order.  Dealt k mod N, a round and a
     * bit; claimed, several rounds: rank 0 collects in order, and while it waits for the oldest chunk -- somebody else's -- it may only claim and
     * compute as far ahead as its ring is long, whereas the others never wait for anybody */
    mdk_chunk *ring = NULL; int *rslot = NULL; hostbuf *hb = NULL; const int F = getenv("MDK_CLAIM") ? 3 * world + 6 : world + 2; int head = 0, count = 0, more = 1; uint32_t n_own = 0;
    char pci[64] = ""; int use_rccl = 1; uint8_t id[MD_COMM_ID_BYTES]; claimer Q; uint64_t n_rec_own = 0;
    R.rank = rank; R.world = world; R.fd = R.cfd = NULL; R.claim = getenv("MDK_CLAIM") != NULL;
    memset(&Q, 0, sizeof(Q)); Q.R = &R; pthread_mutex_init(&Q.mu, NULL); pthread_cond_init(&Q.cv, NULL);
    if(rank != 0) setenv("MDK_NO_OUTPUT", "1", 1);              /* only rank 0 writes the files */
    rc = mdk_plan_open(argc, argv, &p);
    if(rc != 0 || !p) return rc;
    mdk_plan_set_prep(p, 1);
    mdk_plan_set_shard(p, rank, world);
    if(world > (getenv("MDK_CLAIM") ? 10 : 36)) { fprintf(stderr, "[mdk] at most %d ranks\n", getenv("MDK_CLAIM") ? 10 : 36); mdk_plan_close(p); return -1; }
    mdk_plan_set_hold(p, rank == 0 || !getenv("MDK_CLAIM") ? F + 3 : 4);
    memset(&dop, 0, sizeof(dop)); mdk_plan_dev_cfg(p, &dop.cfg); dop.cfg.n_slots = getenv("MDK_CLAIM") ? F : 2;      /* dealt k mod N, rank 0 never holds more than two chunks of its own among the F of its ring; claimed, all of them may be its own */
    { int nd = md_dev_count(); const char *lr = getenv("LOCAL_RANK"); dop.device = getenv("MDK_DEVICE") ? atoi(getenv("MDK_DEVICE")) : nd > 0 ? (lr ? atoi(lr) : rank) % nd : 0; }
    /* the control connections first: a rank that cannot get its device says so by hanging up, and nobody waits for it */
    if(ranks_open(&R)) { ranks_close(&R); mdk_plan_close(p); return MDK_RC_DEVICE; }
    if(R.claim) {          /* (before the first mdk_plan_next_chunk starts the reader) */
        p->claim = claim_chunk; p->claim_ctx = &Q;
        if(rank == 0) Q.th_ok = pthread_create(&Q.th, NULL, dispenser_main, &Q) == 0;
        if(rank == 0 && !Q.th_ok) { fprintf(stderr, "[mdk] cannot create a thread\n"); ranks_close(&R); mdk_plan_close(p); return -5; }
    }
    devopen_main(&dop); dev = dop.dev;
    if(dop.rc) { fprintf(stderr, "[mdk] rank %d cannot open MI355X device %d: %s\n[mdk] this build has no CPU path for `extract`.\n", rank, dop.device, dop.err); ranks_close(&R); mdk_plan_close(p); return MDK_RC_NODEVICE; }
    { md_prep_cfg pc; mdk_plan_prep_cfg(p, &pc); md_dev_set_prep(dev, &pc); mdk_plan_attach_device(p, dev); }
    /* bootstrap: everybody tells rank 0 which device it sits on; rank 0 decides the data channel and hands the RCCL id round */
    (void)md_dev_pci_bus_id(dev, pci, (int)sizeof(pci));
    if(rank == 0) {
        char (*all)[64] = calloc((size_t)world, 64); int j;
        if(!all) { ret = -5; goto out; }
        snprintf(all[0], 64, "%s", pci);
        for(i = 1; i < world; i++) if(rd_all(R.fd[i], all[i], 64)) { fprintf(stderr, "[mdk] rank %d left before the run began\n", i); free(all); ret = MDK_RC_DEVICE; goto out; }
        for(i = 0; i < world; i++) for(j = 0; j < i; j++) if(!strncmp(all[i], all[j], 64)) use_rccl = 0;     /* two ranks on one physical device (all on this node: torchrun --nnodes=1) */
        if(getenv("MDK_RANKS_TCP")) use_rccl = 0;
        free(all);
        memset(id, 0, sizeof(id));
        if(use_rccl && md_comm_unique_id(id)) { fprintf(stderr, "[mdk] RCCL is not usable (%s): site buffers travel over TCP\n", md_dev_last_error()); use_rccl = 0; }
        for(i = 1; i < world; i++) if(wr_all(R.fd[i], &use_rccl, sizeof(int)) || wr_all(R.fd[i], id, sizeof(id))) { ret = MDK_RC_DEVICE; goto out; }
    } else {
        char mine[64]; memset(mine, 0, sizeof(mine)); snprintf(mine, sizeof(mine), "%s", pci);
        if(wr_all(R.fd[0], mine, 64) || rd_all(R.fd[0], &use_rccl, sizeof(int)) || rd_all(R.fd[0], id, sizeof(id))) { ret = MDK_RC_DEVICE; goto out; }
    }
    if(use_rccl && md_comm_open_rank(dev, rank, world, id, &comm)) { fprintf(stderr, "[mdk] rank %d: %s\n", rank, md_dev_last_error()); ret = MDK_RC_DEVICE; goto out; }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk ranks] rank %d of %d on device %d (%s); site buffers travel over %s\n", rank, world, dop.device, pci, use_rccl ? "RCCL (ncclSend/ncclRecv)" : "the TCP connection (ranks share a device)");
    ring = calloc((size_t)F, sizeof(mdk_chunk)); rslot = calloc((size_t)F, sizeof(int)); hb = calloc((size_t)world, sizeof(hostbuf));
    if(!ring || !rslot || !hb) { ret = -5; goto out; }
    if(rank == 0) { if(emitter_start(&em, p, p->o.n_threads >= 8 ? 8 : p->o.n_threads)) { ret = -5; goto out; } have_em = 1; }
    /* every rank walks the whole schedule (foreign chunks cost nothing: with an index their records are not even read); own chunks
     * alternate between two slots */
    if(rank != 0) {
        /* compute chunk n while chunk n-1 is collected and leaves for rank 0 */
        mdk_chunk prev, c; int have_prev = 0, prev_slot = 0;
        for(;;) {
            rc = mdk_plan_next_chunk(p, &c);
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) break;
            if(c.skipped) continue;
            {
                const int sl = (int)(n_own & 1);
                rc = mdk_plan_ensure_reference(p, dev, c.tid);
                if(!rc) rc = md_dev_submit_raw(dev, sl, &c.raw);
                if(rc) { fprintf(stderr, "[mdk] rank %d: device error: %s\n", rank, md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                n_own++; n_rec_own += c.n_records_seen;
                if(have_prev) { rc = send_result(&R, dev, comm, p, &prev, prev_slot, &n_host_prep); have_prev = 0; if(rc) { if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); } fprintf(stderr, "[mdk] rank %d: %s\n", rank, md_dev_last_error()); ret = MDK_RC_DEVICE; break; } }
                prev = c; prev_slot = sl; have_prev = 1;
            }
        }
        if(!ret && have_prev) { rc = send_result(&R, dev, comm, p, &prev, prev_slot, &n_host_prep); if(rc) { if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); } fprintf(stderr, "[mdk] rank %d: %s\n", rank, md_dev_last_error()); ret = MDK_RC_DEVICE; } }
    } else
    /* rank 0 reads up to a round ahead, so that its own chunk computes while the others' arrive, and hands the chunks to the emitter in schedule order */
    while(more || count) {
        if(more && count < F) {
            const int at = (head + count) % F; mdk_chunk *c = &ring[at];
            rc = mdk_plan_next_chunk(p, c);
            if(rc < 0) { ret = rc == -5 ? -5 : -4; break; }
            if(rc == 0) { more = 0; continue; }
            rslot[at] = -1;
            if(!c->skipped) {                         /* an own chunk with records */
                const int sl = R.claim ? at : (int)(n_own & 1);
                rc = mdk_plan_ensure_reference(p, dev, c->tid);
                if(!rc) rc = md_dev_submit_raw(dev, sl, &c->raw);
                if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
                rslot[at] = sl; n_own++; n_rec_own += c->n_records_seen;
            }
            count++;
            continue;
        }
        {   /* the oldest chunk of the ring */
            mdk_chunk *c = &ring[head]; md_sites sites; memset(&sites, 0, sizeof(sites));
            if(rslot[head] >= 0) {                    /* rank 0's own */
                rc = md_dev_download(dev, rslot[head], &sites);
                if(rc == MDK_ERR_PREP_HOST) { rc = mdk_plan_host_prepare_from(p, c, dev, rslot[head]); if(!rc) rc = md_dev_submit(dev, rslot[head], &c->batch); if(!rc) rc = md_dev_download(dev, rslot[head], &sites); n_host_prep++; }
            } else if(c->skipped == MDK_CHUNK_FOREIGN) {      /* somebody else's, with records */
                const int src = R.claim ? owner_wait(&Q, c->index) : (int)(c->index % (uint32_t)world);
                if(src < 1 || src >= world) { fprintf(stderr, "[mdk] nobody claimed chunk %" PRIu32 "\n", c->index); ret = MDK_RC_DEVICE; break; }
                rc = recv_result(&R, comm, src, &hb[src], &sites);
            }
            else rc = 0;                              /* passed over by everybody (-l, a contig the FASTA lacks) */
            if(rc == MDK_ERR_STRAND0) { fprintf(stderr, "Can't determine the strand of a read!\n"); abort(); }
            if(rc) { fprintf(stderr, "[mdk] device error: %s\n", md_dev_last_error()); ret = MDK_RC_DEVICE; break; }
            if(emitter_push(&em, c, &sites)) { ret = em.failed ? MDK_RC_OUTPUT : MDK_RC_DEVICE; break; }
            head = (head + 1) % F; count--;
        }
    }
    if(have_em) { emitter_stop(&em); if(em.failed && !ret) ret = MDK_RC_OUTPUT; }
    if(getenv("MDK_HOST_PROFILE")) fprintf(stderr, "[mdk ranks] rank %d: %u own chunks (%s) holding %" PRIu64 " records, %d prepared on the host after all, rc %d\n", rank, n_own, R.claim ? "claimed" : "k mod N", n_rec_own, n_host_prep, ret);
    if(rank == 0 && ret == 0) mdk_plan_finish(p);
out:
    if(Q.th_ok) { pthread_mutex_lock(&Q.mu); Q.quit = 1; pthread_cond_broadcast(&Q.cv); pthread_mutex_unlock(&Q.mu); pthread_join(Q.th, NULL); }
    ranks_close(&R);
    if(fast_exit_wanted() && ret == 0) leave_fast_plan(p, ret);
    if(hb) { for(i = 0; i < world; i++) { free(hb[i].site); free(hb[i].var); } free(hb); }
    free(ring); free(rslot);
    if(comm) md_comm_close(comm);
    mdk_plan_detach_device(p);
    md_dev_close(dev);
    mdk_plan_close(p);
    return ret;
}

/*
 * mdk_hip.h -- C ABI of the MI355X device library (libmdk_hip.so) for the `MethylDackel extract`
 * hot path.  Plain pointers and sizes only; no HIP, torch or C++ types cross this boundary, so
 * the C host (and the reference's own C code) can include it directly.
 *
 * What this boundary replaces in the reference (the reference has no FFI layer; its seam for the
 * hot path is the htslib pileup engine + callbacks it drives from extractCalls):
 *   md_dev_open / md_dev_cfg      <- the subset of `Config` the per-base arithmetic reads
 *                                    (MethylDackel.h:90-126: keepCpG/CHG/CHH, minPhred, minOppositeDepth,
 *                                    bounds[16], absoluteBounds[16]; defaults extract.c:715-753)
 *   md_dev_set_reference          <- faidx_fetch_seq window handed to the column loop (extract.c:381,388-390)
 *   md_read_batch / md_dev_upload <- the reads bam_mplp64_auto pulls through filter_func for one chunk
 *                                    (extract.c:379,394-399; common.c:407-463) *after* admission, with the
 *                                    CIGAR -> (qpos, is_del, is_refskip) resolution htslib does per column
 *                                    (resolve_cigar2) done once per read on the host
 *   md_dev_launch                 <- the whole per-chunk pileup: trimming (common.c:137-208), mate-overlap
 *                                    resolution (overlaps.c:54-147), context classification
 *                                    (common.c:49-82, extract.c:407-418) and the per-read-base counting
 *                                    loop (extract.c:420-441, 225-239; common.c:118-134)
 *   md_sites / md_dev_download    <- the (pos, nmethyl, nunmethyl, nOff, nVariant) tuple that reaches the
 *                                    variant filter and writeCall/processLast (extract.c:444-491)
 *
 * All functions return 0 on success and a negative value on failure (never throw, never exit);
 * md_dev_last_error() gives a message for the calling thread's last failure.
 */
#ifndef MDK_HIP_H
#define MDK_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MDK_ERR_HIP      (-1)   /* a HIP runtime call failed */
#define MDK_ERR_NODEVICE (-2)   /* no usable MI355X/gfx950 device */
#define MDK_ERR_ARG      (-3)   /* bad argument */
#define MDK_ERR_NOREF    (-4)   /* reference for the batch's contig was never uploaded */
#define MDK_ERR_STRAND0  (-5)   /* a read of undeterminable strand reached a methylation call
                                   (the reference aborts there: common.c:122-125) */
#define MDK_ERR_NOMEM    (-6)
#define MDK_ERR_PREP_HOST (-7)  /* device chunk preparation met a read name it does not handle (more records of one name than a
                                   lane keeps): prepare this chunk on the host (md_dev_submit of a host-built batch) */

typedef struct md_dev md_dev;

typedef struct {
    int32_t keepCpG, keepCHG, keepCHH;   /* contexts to count (Config.keep*) */
    int32_t minPhred;                    /* -p, already normalised to >= 1 (extract.c:997-1000) */
    int32_t minOppositeDepth;            /* >0: also produce nOff / nVariant per site */
    int32_t bounds[16];                  /* --OT/--OB/--CTOT/--CTOB, index 4*(strand-1)+{R1 left,R1 right,R2 left,R2 right} */
    int32_t absoluteBounds[16];          /* --nOT/--nOB/--nCTOT/--nCTOB, same indexing */
    int32_t tile;                        /* reference positions per LDS tile; 0 = library default */
    int32_t n_slots;                     /* batches that may be in flight at once; 0 = 2 (double buffering) */
    int32_t n_streams;                   /* 0 = a stream per slot; k > 0 = k streams, slots 8j .. 8j+7 work on stream j % k: the slots of one
                                            md_dev_launch_group can share a stream -- creating one costs the runtime ~5 ms */
} md_dev_cfg;

/* The device sees every admitted alignment as one or more gapless SEGMENTS: the CIGAR is expanded on the host
 * (reference <-> query map), so the kernel never walks a CIGAR.  A segment is a run of M/=/X bases; it is split
 * further so that the read it is overlap-resolved against (overlaps.c:54-119) either covers the whole segment with
 * one of ITS gapless runs, or does not touch it.  32 bytes, coalesced one per lane.
 * The payload of a read lives at blob + 4*off4:
 *   uint8 seq[(l_qseq+1)/2]   BAM 4-bit codes, high nibble = even query index; padded to a multiple of 4 bytes
 *   uint8 qual[l_qseq]        raw phred bytes; padded to a multiple of 4 bytes
 * Segments are emitted in file (coordinate) order of their reads, ascending within a read. */
typedef struct {
    int32_t  rpos;       /* reference position of the segment's first base */
    uint32_t off4;       /* payload of the read the segment belongs to */
    uint32_t l_qseq;     /* length of that READ (trimming bounds are defined on the whole read, common.c:137-208) */
    uint32_t q0;         /* query index of the segment's first base */
    uint16_t len;        /* bases in the segment, >= 1 */
    uint8_t  sf;         /* MDK_SF_*: strand of origin (getStrand, common.c:84-116) and flags */
    uint8_t  msf;        /* partner: strand (bits 0-2) and read #2 (bit 3); valid iff sf & MDK_SF_PARTNER */
    uint32_t m_off4;     /* partner read's payload */
    uint32_t m_l_qseq;   /* partner read's length */
    uint32_t m_q0;       /* partner's query index of the base aligned to rpos */
} md_seg;
#define MDK_SF_STRAND  7u     /* bits 0-2: 1 OT, 2 OB, 3 CTOT, 4 CTOB, 0 undeterminable */
#define MDK_SF_READ2   8u     /* read #2 (BAM flag 0x80) */
#define MDK_SF_SECOND  16u    /* this read is the LATER-in-file member of its pair ('b' of overlaps.c:54) */
#define MDK_SF_PARTNER 32u    /* m_* valid: the partner covers the whole segment gaplessly */

/* The admitted reads of ONE interval [beg,end) of one contig -- what one chunk of extractCalls sees.
 * Host-owned; must stay valid until the slot is downloaded, waited for or synced (copies are asynchronous). */
typedef struct {
    int32_t  tid;
    int64_t  beg, end;          /* columns counted: beg <= pos < end (extract.c:400) */
    int32_t  n_segs;
    const md_seg *seg;          /* [n_segs] */
    const uint8_t *blob;
    uint64_t blob_bytes;
    int32_t  n_reads;           /* informational: admitted alignments behind the segments */
    uint64_t algo_bytes;        /* informational: sum over those reads of 16 + 4*n_cigar + ceil(l/2) + l */
} md_read_batch;

/* Result of one interval: every position with nmeth+nunmeth > 0 (or nOff > 0 when minOppositeDepth > 0),
 * ascending.  Pointers are host memory owned by the library, valid until the slot is reused. */
typedef struct { uint32_t pos, nmeth, nunmeth, meta; } md_site;   /* meta bit 0: reference base is G/g; bits 1-2: context 0 CpG / 1 CHG / 2 CHH */
typedef struct { uint32_t noff, nvar; } md_site_var;              /* opposite-strand depth / variant evidence (extract.c:225-239) */
typedef struct {
    int64_t n_sites;
    const md_site *site;
    const md_site_var *var;          /* NULL unless minOppositeDepth > 0 */
} md_sites;

/* Device-resident result of one interval, as the kernel leaves it: tile t's sites (ascending) occupy
 * site[seg[t].off .. seg[t].off + seg[t].cnt).  Tiles are in position order but their segments are not: a tile
 * reserves room for one site per kept context position with a single atomic, so segments appear in completion order
 * and a segment may be followed by unused slots.  Used for device-to-device exchange (RCCL gather). */
typedef struct { uint32_t off, cnt; } md_tile_seg;
typedef struct {
    int64_t n_slots;                 /* slots of d_site in use (>= number of sites) */
    int32_t n_tiles;
    const md_site *d_site; const md_site_var *d_var; const md_tile_seg *d_seg;   /* DEVICE pointers */
} md_sites_dev;

/* ---- chunk preparation on the device ----
 * Instead of a host-built md_read_batch the device can take the records of a chunk as they lie in the inflated BAM stream and
 * do the reference's per-record work itself: filter_func's admission tests (common.c:416-444) with the NH / XG aux walk,
 * getStrand (common.c:84-116), the mappability windows (common.c:277-335), the BED span test (common.c:432-439), the
 * conversion-efficiency filter (common.c:338-404), the read-name pairing of the overlap constructor/destructor callbacks
 * (overlaps.c:121-147, with htslib's buffer eviction) and the CIGAR -> segment expansion (overlaps.c:27-52).
 * md_prep_cfg: the part of `Config` those steps read (MethylDackel.h:90-126). */
typedef struct {
    int32_t min_mapq, ignore_flags, require_flags, keep_dupes, ignore_nh, keep_singleton, keep_discordant;
    int32_t min_phred;            /* -p, for the conversion-efficiency filter */
    float   min_conv_eff;         /* --minConversionEfficiency, 0 = off */
    int32_t map_on, min_mappable; /* -M/-B given; --minMappableBases */
    int32_t no_pairing;           /* mbias: no overlap handler is installed (MBias.c:158-161) */
    int32_t perread;              /* perRead: a record is kept iff it STARTS inside the chunk and passes -R / -F / -q (perRead.c:178-183) */
} md_prep_cfg;
/* whole records back to back, each as in the file: uint32 block_size, then block_size bytes.  Where the records of a range start is told in one
 * of three ways:
 *   d_rec_off != NULL   the range lies in DEVICE memory -- a run of members of a piece inflated on the device (md_piece_*): ptr and d_rec_off are
 *                       device pointers, d_rec_off[i] - rec_delta is the offset of the range's record i from ptr;
 *   h_rec_off != NULL   host memory with a table of its own (what the thread that inflated the bytes noted): h_rec_off[i] - rec_delta is the
 *                       offset of record i from ptr (host memory, read before md_dev_upload_raw returns or asynchronously when it is staging memory);
 *   neither             host memory; its records' offsets are the next n_records entries of the batch's rec_off array.
 * n_records must be filled in for every range as soon as one range has a table of either kind; a batch without any may leave it 0. */
typedef struct { const uint8_t *ptr; uint64_t bytes; const uint32_t *d_rec_off; uint32_t n_records, rec_delta; const uint32_t *h_rec_off; } md_raw_range;
/* The candidate records of ONE chunk: everything the region query [beg,end) of the chunk's contig returns (pos < end,
 * bam_endpos > beg), in file order.  Host ranges hold exactly those; a range in device memory is a run of whole BGZF members and may hold
 * records of the neighbouring chunk or contig at its ends, which the preparation drops (it redoes the query per record).  rec_off = for the records
 * of the ranges WITHOUT a table of their own, in order: offset of the record's block_size word in the concatenation of all ranges (less than 4 GiB in total).  woff/wlen: the reference window the chunk fetches (extract.c:381), which the
 * conversion-efficiency filter classifies inside.  Host-owned; valid until the slot is waited for. */
typedef struct {
    int32_t tid; int64_t beg, end;
    int32_t n_ranges; const md_raw_range *range;
    int32_t n_records; const uint32_t *rec_off;
    int64_t woff, wlen;
} md_raw_batch;
int  md_dev_set_prep(md_dev *h, const md_prep_cfg *cfg);
/* mappability of a contig, 1 bit per base (bit i%32 of word i/32; 1 = mappable), for the admission windows */
int  md_dev_set_mappability(md_dev *h, int32_t tid, const uint32_t *bits, int64_t n_bases);
/* md_dev_upload_raw = H2D of the ranges + the preparation kernels; the slot is then in the same state as after md_dev_upload
 * (launch / download / wait / bench work on it).  md_dev_submit_raw = upload_raw + launch.  md_dev_download / md_dev_wait
 * return MDK_ERR_PREP_HOST when the preparation gave up on the chunk (see above). */
int  md_dev_upload_raw(md_dev *h, int slot, const md_raw_batch *b);
/* The same for a caller that can keep its device memory: a batch that is ONE device-resident range (the usual chunk of a file inflated on the
 * device: a run of members of one piece) is not copied at all -- the preparation and the pileup read the records where md_piece_* left them,
 * with the piece's own record table.  Returns 1 then, and the range (the piece's buffers) must stay as it is until the slot's results have been
 * collected (md_dev_download / md_dev_download_group / md_dev_wait); 0 = copied as md_dev_upload_raw does (the memory may be reused after
 * md_dev_upload_wait); < 0 an error.  MDK_NO_INPLACE=1 in the environment: always copies. */
int  md_dev_upload_raw_inplace(md_dev *h, int slot, const md_raw_batch *b);
/* waits until the copies md_dev_upload_raw (or md_dev_upload) queued for the slot have read their host memory, which may then be reused */
int  md_dev_upload_wait(md_dev *h, int slot);
/* the same without waiting: 1 = they have, 0 = not yet, < 0 error */
int  md_dev_upload_done(md_dev *h, int slot);
int  md_dev_submit_raw(md_dev *h, int slot, const md_raw_batch *b);
/* the records of an uploaded slot back on the host, as the device holds them: the concatenation of the batch's ranges (bytes)
 * and every record's offset in it (rec_off[n_records]); for a chunk the device preparation gives up on (MDK_ERR_PREP_HOST)
 * whose records were inflated on the device and so never existed in host memory.  bytes/n_records: capacities in, sizes out. */
int  md_dev_read_raw(md_dev *h, int slot, uint8_t *bytes, uint64_t *n_bytes, uint32_t *rec_off, uint32_t *n_records);
/* The preparation kernels of a slot uploaded with md_dev_upload_raw, re-run `iters` times on the resident records and timed
 * with HIP events (the slot's segments are the same afterwards). */
int  md_dev_bench_prep(md_dev *h, int slot, int warmup, int iters, float *ms_per_chunk);
/* the same over `n` uploaded raw slots holding different intervals, `per_launch` chunks per launch of each preparation kernel (as
 * md_dev_launch_group prepares them), round robin on one stream: ms per LAUNCH when the records stream from HBM */
int  md_dev_bench_prep_rotate(md_dev *h, const int *slots, int n, int per_launch, int warmup, int iters, float *ms_per_launch);
/* Test hook: the segments the device built for an uploaded slot (off4 / m_off4 are BYTE offsets into the uploaded records,
 * qualities follow the sequence without padding), their number, and the number of admitted reads. */
int  md_dev_debug_segments(md_dev *h, int slot, md_seg *out, int64_t cap, int64_t *n_segs, int64_t *n_reads);

/* ---- BGZF inflate and BAM record framing on the device ----
 * What the reference gets from htslib inside sam_itr_next (common.c:413): bgzf_read_block's inflate of each <= 64 KiB member and
 * bam_read1's framing of the records in it.  The host hands over a PIECE of the file -- a run of whole BGZF members, compressed,
 * as they lie in the file -- with a table of its members (found by walking the 18-byte BGZF headers: BSIZE, and ISIZE from each
 * member's trailer); the device inflates every member (one wavefront each), checks its CRC32, walks the records of every member from its first
 * byte, and leaves: the inflated bytes, a table with the offset of every record, and per member a DIGEST (first/last record,
 * extent of the read ends, coordinate order inside) from which the host applies the reference's chunk schedule
 * (extract.c:325-350) without ever seeing a record.  Inflated bytes and record table stay in device memory; a chunk's records
 * are then handed to md_dev_upload_raw as device-resident ranges (md_raw_range.d_rec_off != NULL). */
typedef struct { uint64_t in_off; uint32_t in_len, out_len; uint64_t out_off; uint32_t crc32, reserved; } md_inf_member;   /* deflate stream at comp + in_off, in_len bytes; ISIZE; where its bytes go (running sum of ISIZE); the CRC32 of the member's trailer, checked on the device as htslib's bgzf_read_block checks it */
typedef struct {
    uint32_t n_rec, first_rec;           /* records in the member; index of its first record in the piece's record table */
    int32_t tid0, pos0, tidN, posN;      /* first and last record */
    int32_t min_endp, max_endp;          /* extent of bam_endpos over the member's records */
    int32_t ok, sorted;                  /* ok: the member starts and ends on record boundaries; sorted: placed records in coordinate order, none unplaced */
} md_inf_digest;
typedef struct md_piece md_piece;
typedef struct {
    int32_t n_mem; const md_inf_digest *digest;      /* host memory owned by the piece, valid until its next submit */
    uint32_t n_records; uint64_t out_bytes;
    const uint8_t *d_out; const uint32_t *d_rec_off;  /* DEVICE pointers: inflated bytes; offset (in d_out) of every record's block_size word */
} md_piece_info;
/* how many members the device inflates at once (the wavefronts k_inflate keeps resident, each on one member): a piece of a whole multiple of this
 * many members leaves no last, nearly empty round of them; <= 0: unknown */
int  md_piece_members_per_round(md_dev *h);
int  md_piece_create(md_dev *h, md_piece **out);     /* buffers grown on demand; its work is queued on one of a few streams the pieces of a handle share (four; MDK_PIECE_STREAMS=0: a stream of its own) */
void md_piece_destroy(md_piece *p);
/* asynchronous: H2D of `comp` (pinned memory makes it a DMA) and the member table, the kernels, D2H of the digests.  The member
 * table must be contiguous (out_off = running sum of out_len), out_len <= 65536.  comp/mem must stay valid until md_piece_wait. */
int  md_piece_submit(md_piece *p, const uint8_t *comp, uint64_t comp_bytes, const md_inf_member *mem, int32_t n_mem);
int  md_piece_wait(md_piece *p, md_piece_info *info);
/* inflated bytes / record offsets copied back to the host (tests; files whose records straddle members) */
int  md_piece_read(md_piece *p, uint64_t off, uint64_t bytes, uint8_t *dst);
int  md_piece_read_records(md_piece *p, uint32_t first, uint32_t n, uint32_t *dst);
/* the kernels alone, re-run on the resident piece and timed with HIP events on its stream */
int  md_piece_bench(md_piece *p, int iters, float *ms_inflate, float *ms_walk);
int  md_piece_bench_crc(md_piece *p, int iters, float *ms_crc);      /* k_crc32 alone on the resident piece (and its verdict) */

typedef struct {
    float ms_total;      /* one launch bracketed by HIP events on the slot's stream (includes lone-launch dispatch latency), mean over iters */
    float ms_pileup;     /* the pileup kernel: `iters` launches back to back between two HIP events, divided by iters */
    uint64_t algo_bytes; /* algorithmic bytes of one launch (SURVEY.md 8d formula) */
    uint64_t n_sites;
    int32_t tile, n_tiles, lds_bytes;   /* geometry the launch used: positions per tile, tiles, LDS bytes per workgroup */
} md_bench_result;

int  md_dev_count(void);                                       /* number of HIP devices, <0 on error */
/* optional: create the device context and load the kernels ahead of md_dev_open (e.g. on a thread, while options and
 * input headers are still being read) */
int  md_dev_warm(int device);
/* for a process about to leave without closing its handle (the command's _exit; also run at exit): joins the helper threads md_dev_warm started
 * (registration of staging blocks, streams and device blocks made ahead) and starts none again.  md_dev_close joins them too, but a later
 * md_dev_warm / md_dev_open in the same process starts its own again.  Idempotent. */
void md_dev_quiesce(void);
/* How much device memory the caller expects to use (inflated pieces, chunk slots, contigs), told as early as it knows (the size of the BAM):
 * md_dev_warm's helper thread makes that much carved memory ahead, while the copy engines are still idle.  An allocation made later, next to
 * the pieces' copies, costs its caller 10-30 ms and holds up every other call into the runtime meanwhile (profiles/r06pf_*).  No handle
 * needed; may be called before, during or after md_dev_warm; without it memory is made when first asked for. */
void md_dev_reserve_hint(uint64_t device_bytes);
int  md_dev_open(int device, const md_dev_cfg *cfg, md_dev **out);
/* room for the references of contigs 0..n-1, so that a thread may upload the next contig (md_dev_set_reference and what goes with it) while
 * others work on slots of contigs already uploaded; without it md_dev_set_reference must not run next to other calls on the handle */
int  md_dev_reserve_contigs(md_dev *h, int32_t n);
void md_dev_close(md_dev *h);
const char *md_dev_last_error(void);
int  md_dev_tile(const md_dev *h);

/* Upload (once) the bases of a contig; letters verbatim from the FASTA (case matters: C/c, G/g). */
int  md_dev_set_reference(md_dev *h, int32_t tid, const char *seq, int64_t len);

/* -l/--keepStrand: restrict the sites of a contig (whose reference is already uploaded) to these runs -- sorted,
 * disjoint, half-open, strand 0 = either, 1 = '+' (only OT/CTOT reads are seen there), 2 = '-' (only OB/CTOB reads).
 * Replaces posOverlapsBED + readStrandOverlapsBED inside the column loop (extract.c:402-405,425; bed.c:46-64).
 * n = 0 leaves the contig without sites.  Calling it again replaces the previous restriction only where a position
 * is still a site, so set_reference again first to widen. */
typedef struct { int32_t start, end, strand; } md_region;
int  md_dev_set_regions(md_dev *h, int32_t tid, const md_region *runs, int64_t n);

/* mbias (MBias.c:57-230): histogram of calls over (strand, read number, position in read) instead of per-position
 * counts; the reference keeps it as strandMeth{meth1,unmeth1,meth2,unmeth2}[qpos] per strand (MethylDackel.h:171-176).
 * count[q*16 + (strand-1)*4 + (read 2 ? 2 : 0) + (unmethylated ? 1 : 0)], 0 <= q < len; strand 1..4 = OT, OB, CTOT, CTOB.
 * md_dev_mbias_submit = md_dev_upload + the histogram kernel; the batch (built WITHOUT mate pairing: mbias installs no
 * overlap handler, MBias.c:159) must stay valid until md_dev_slot_sync(slot) or the next submit on that slot returns.
 * The histogram accumulates over submits in device memory; md_dev_mbias_read waits for all of them and returns it
 * (memory owned by the handle, valid until the next read/reset/close). */
typedef struct { int32_t len; const uint32_t *count; } md_mbias;
int  md_dev_mbias_submit(md_dev *h, int slot, const md_read_batch *b);
/* the same from the chunk's raw records (md_dev_set_prep with no_pairing = 1 first): H2D + device preparation are queued and the call
 * returns; the histogram kernel follows once the preparation has reported the longest admitted read (which sizes the histogram rows
 * kept in LDS) -- at the next submit on another slot, at md_dev_slot_sync or at md_dev_mbias_read, whichever comes first.  The ranges
 * must stay valid until md_dev_slot_sync(slot) or the next submit on that slot returns; MDK_ERR_PREP_HOST is not possible here (no
 * pairing). */
int  md_dev_mbias_submit_raw(md_dev *h, int slot, const md_raw_batch *b);
int  md_dev_mbias_read(md_dev *h, md_mbias *out);
int  md_dev_mbias_reset(md_dev *h);
int  md_dev_slot_sync(md_dev *h, int slot);

/* perRead (perRead.c): per-read CpG methylation.  One md_pr_read per alignment the command keeps (start inside the chunk,
 * -F/-R/-q passed), in file order; `cigar` holds the BAM CIGAR words of all reads back to back (cig_off/n_cigar index it);
 * the payload in `blob` is laid out as for md_read_batch (4*off4: seq nibbles padded to 4 bytes, then qualities).
 * Replaces processRead (perRead.c:38-94) for every read of a chunk; counts[i] belongs to read[i].
 * The reference must be resident WITHOUT md_dev_set_regions: perRead uses -l only to pass over whole chunks. */
typedef struct { int32_t pos; uint32_t off4, l_qseq, cig_off; uint16_t n_cigar; uint8_t strand, reserved; } md_pr_read;
typedef struct { int32_t tid; int64_t beg, end; int32_t n_reads; const md_pr_read *read; const uint32_t *cigar; uint64_t n_cigar;
                 const uint8_t *blob; uint64_t blob_bytes; } md_pr_batch;
typedef struct { uint32_t nmeth, nunmeth; } md_pr_count;
int  md_dev_perread_submit(md_dev *h, int slot, const md_pr_batch *b);                 /* H2D + kernel + D2H enqueued on the slot's stream */
int  md_dev_perread_download(md_dev *h, int slot, const md_pr_count **counts, int64_t *n);   /* waits; memory owned by the slot */
/* the same from the chunk's raw records (md_dev_set_prep with perread = 1): the device selects the reads (kept[i] = index into
 * the batch's rec_off of the i-th kept read, ascending) and walks them; n kept reads, counts[i] belongs to kept[i] */
int  md_dev_perread_submit_raw(md_dev *h, int slot, const md_raw_batch *b);
int  md_dev_perread_download_raw(md_dev *h, int slot, const uint32_t **kept, const md_pr_count **counts, int64_t *n);

/* slot in [0, n_slots): upload is H2D on the slot's stream; launch enqueues the kernels; download waits for
 * the slot and returns the sites.  md_dev_submit = upload + launch. */
int  md_dev_upload(md_dev *h, int slot, const md_read_batch *b);
int  md_dev_launch(md_dev *h, int slot);
int  md_dev_submit(md_dev *h, int slot, const md_read_batch *b);
/* One launch of each kernel over several uploaded slots (at most md_dev_group_max() = 8, all different): a 1 Mb chunk alone is
 * fewer than two workgroups per CU, so chunks are launched together -- the preparation kernels of the slots uploaded with
 * md_dev_upload_raw (queued here, not at upload) and the pileup; each chunk keeps its own reads, outputs and site counter, and
 * download / wait are per slot as after md_dev_launch. */
int  md_dev_launch_group(md_dev *h, const int *slots, int n);
int  md_dev_group_max(void);
int  md_dev_download(md_dev *h, int slot, md_sites *out);
/* md_dev_download for the slots of ONE md_dev_launch_group, with one wait and one round of copies for all of them instead of one per slot:
 * rc[i] is what md_dev_download(slots[i]) would have returned (0, MDK_ERR_PREP_HOST, MDK_ERR_STRAND0, ...) and out[i] its sites.  Returns 0 when
 * the collection itself worked, whatever the rc[i]. */
int  md_dev_download_group(md_dev *h, const int *slots, int n, md_sites *out, int *rc);
int  md_dev_sync(md_dev *h);

/* Make the kernels of `slot` write their result into caller-provided DEVICE buffers (e.g. torch tensors that are
 * then exchanged over RCCL) instead of library memory: d_site[cap_sites] (md_site), d_var[cap_sites] (md_site_var,
 * may be NULL when minOppositeDepth == 0), d_seg[cap_tiles] (md_tile_seg).  Pass all-NULL to unbind.
 * md_dev_wait then reports how many slots/tiles were used; MDK_ERR_ARG if a capacity was too small (cap_sites must
 * be at least the number of kept context positions of the interval; the interval length always suffices). */
int  md_dev_bind_output(md_dev *h, int slot, void *d_site, void *d_var, void *d_seg, int64_t cap_sites, int64_t cap_tiles);
/* wait for the slot's kernels; fills the device view (library or bound buffers) */
int  md_dev_wait(md_dev *h, int slot, md_sites_dev *out);
/* host-side helper: put a segmented result (copied to host memory, n_slots slots) into ascending order;
 * returns the number of sites written to out_site (>= 0) or a negative error */
int64_t md_sites_order(const md_site *site, const md_site_var *var, const md_tile_seg *seg, int32_t n_tiles, int64_t n_slots,
                       md_site *out_site, md_site_var *out_var);

/* Re-run the kernels of an uploaded slot `iters` times (inputs stay resident in HBM; results are identical
 * every time) and time them with HIP events on the slot's stream. */
int  md_dev_bench(md_dev *h, int slot, int warmup, int iters, md_bench_result *out);

/* The same for `n` uploaded slots holding DIFFERENT intervals, launched `per_launch` at a time (1 = md_dev_launch, more =
 * md_dev_launch_group; n must be a multiple) round robin on one stream, `iters` launches in all: with enough slots the
 * working set exceeds the 256 MiB Infinity Cache and every launch streams its inputs from HBM.  algo_bytes / n_sites /
 * n_tiles are per LAUNCH, averaged over the rotation; ms_total == ms_pileup. */
int  md_dev_bench_rotate(md_dev *h, const int *slots, int n, int per_launch, int warmup, int iters, md_bench_result *out);

/* ---- multi-GPU: the exchange step of the interval-sharded path (SURVEY.md 8b last row, 8e) ----
 * Chunk k of the reference's schedule belongs to GPU k mod N; per-interval site buffers travel to rank 0 (whose host writes
 * the files) with ncclSend/ncclRecv groups over xGMI -- a gather, never a reduction.  RCCL is loaded on first use.
 *   md_comm_open_rank   one process per GPU (torchrun-style launch, or the command's own ranks mode, csrc/host/mdk_ranks.c):
 *                       rank 0 makes an id, every rank gets it out of band
 * md_comm_gather: d_send/send_bytes have one entry (this rank's buffer), d_recv/recv_bytes are indexed by rank and only read
 * on rank 0 (entry 0 may be NULL to leave rank 0's own buffer where it is).  Sizes must agree on both sides.  Asynchronous:
 * the send buffers must be complete before the call, and md_comm_wait must return before either side is touched again. */
#define MD_COMM_ID_BYTES 128
typedef struct md_comm md_comm;
int  md_comm_unique_id(uint8_t *id /* [MD_COMM_ID_BYTES] */);
int  md_comm_open_rank(md_dev *h, int rank, int world, const uint8_t *id, md_comm **out);
/* one process per rank where ranks SHARE a physical device (tests of the multi-process path on a single GPU: RCCL refuses two ranks
 * on one device): no RCCL; rank 0's receive buffers are mapped into the peers with HIP IPC and a "send" is a device copy into the
 * mapping.  `oob` is the caller's out-of-band all-gather (every rank contributes `bytes` bytes, receives world*bytes in rank order;
 * bench.py gives torch.distributed over gloo), used to agree on sizes and to pass the IPC handles round.  Only md_bench_* uses such
 * a communicator. */
typedef int (*md_comm_oob_fn)(void *ctx, const void *send, void *recv, uint64_t bytes);
int  md_comm_open_rank_shared(md_dev *h, int rank, int world, md_comm_oob_fn oob, void *ctx, md_comm **out);
void md_comm_close(md_comm *c);
int  md_comm_world(const md_comm *c);
int  md_comm_gather(md_comm *c, const void *const *d_send, const uint64_t *send_bytes, void *const *d_recv, const uint64_t *recv_bytes);
int  md_comm_wait(md_comm *c);
/* One process per GPU (md_comm_open_rank): a finished chunk's result travels from the rank that computed it to rank 0, which writes the
 * files.  The sizes go first, out of band (the command's TCP connection between the ranks): md_comm_result_header waits for the slot's
 * kernels and describes what will be sent (rc = what md_dev_download would have returned: 0, MDK_ERR_PREP_HOST, MDK_ERR_STRAND0 ...);
 * md_comm_result_send then posts the ncclSend of the three arrays (site records, variant evidence, tile segments) and
 * md_comm_result_recv, on rank 0 with the header it was given, the matching ncclRecv, waits, copies to the host and orders the sites.
 * The returned arrays belong to the communicator and stay valid until the next md_comm_result_recv from the same rank. */
typedef struct { int64_t n_slots; int32_t n_tiles, variant, rc, reserved; } md_result_hdr;
int  md_comm_result_header(md_dev *h, int slot, md_result_hdr *hdr);
int  md_comm_result_send(md_comm *c, int slot, const md_result_hdr *hdr);
int  md_comm_result_recv(md_comm *c, int src, const md_result_hdr *hdr, md_sites *out);
/* PCI bus id of the handle's device ("0000:c1:00.0"): two ranks on the same physical device cannot be RCCL peers */
int  md_dev_pci_bus_id(const md_dev *h, char *buf, int cap);

/* The resident-input benchmark loop of bench.py: `n` uploaded slots holding different intervals are launched `group` at a
 * time (md_dev_launch_group; n a multiple of group, at least two groups) round robin, two launches in flight (launch g is
 * issued, then launch g-1 is collected, as extract_main does); the kernels write straight into a send buffer and, with a
 * communicator, the results of a launch travel to rank 0 in one exchange while the next launch is computed.
 * md_bench_run(launches): that many kernel launches.  md_bench_verify compares what the last launch left in the send buffer
 * with md_dev_download of the same intervals (and, on rank 0, checks that every peer's data arrived). */
typedef struct md_bench md_bench;
typedef struct { uint64_t launches, slots_last, exchanges, bytes_per_exchange; } md_bench_run_result;
int  md_bench_open(md_dev *h, md_comm *comm /* NULL: one GPU */, const int *slots, int n, int group, md_bench **out);
int  md_bench_run(md_bench *b, int64_t launches, md_bench_run_result *out);
/* on = 1 (default when the slots hold raw records): every launch of md_bench_run prepares its chunks again from their resident
 * records before the pileup -- the whole device work `extract` does per chunk; on = 0: the pileup alone over the resident segments */
int  md_bench_set_prep(md_bench *b, int on);
int  md_bench_verify(md_bench *b);
int64_t md_bench_region_bytes(const md_bench *b);
void md_bench_close(md_bench *b);

/* Test hook: effective (post-trim, post-overlap-resolution) base code and quality of every query base of
 * every read of an uploaded slot, written to host arrays laid out like the blob's qual/seq (one byte per
 * base, concatenated in read order; out_off[i] = start of read i). */
int  md_dev_debug_effective(md_dev *h, int slot, uint8_t *out_base, uint8_t *out_qual, const uint64_t *out_off);

/* Pinned host memory for staging buffers (so the C host never includes HIP headers). */
void *md_host_alloc(uint64_t bytes);
/* Pinning costs ~0.3 s per GB up front and again at process exit; a pageable buffer costs ~5 ms per 55 MB upload instead.
 * on = 0 makes later md_host_alloc calls return pageable memory (the host does this for inputs of a few dozen chunks). */
void  md_host_set_pinned(int on);
void  md_host_free(void *p);
/* what registering staging blocks has cost so far (seconds on the uploading threads, calls, bytes) */
void  md_host_profile(double *seconds, uint64_t *calls, uint64_t *bytes);
/* MDK_HOST_PROFILE=1: seconds and calls the host threads have spent inside the library per site (waiting for a slot, allocating, queueing
 * copies, launching, waiting for results, copying them back, ...), as one line of text */
int   md_dev_profile_text(char *buf, int cap);
/* a staging block is registered with the runtime (hipHostRegister) the first time an upload reads from it; a thread that has just FILLED
 * one may do that itself once a device is open (h: that device), so that the thread submitting uploads does not have to.  ptr: anywhere
 * inside the block */
void  md_host_register(md_dev *h, const void *ptr);
/* register every staging block that is not registered yet, with `threads` threads; returns how many there were.  For the moment the device
 * comes up: the blocks filled until then would otherwise be registered one by one by the thread that uploads from them */
int   md_host_register_all(md_dev *h, int threads);

#ifdef __cplusplus
}
#endif
#endif

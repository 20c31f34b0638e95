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
 *                                    (extract.c:379,394-399; common.c:407-463) *after* admission
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

typedef struct md_dev md_dev;

typedef struct {
    int32_t keepCpG, keepCHG, keepCHH;   /* contexts to count (Config.keep*) */
    int32_t minPhred;                    /* -p, already normalised to >= 1 (extract.c:997-1000) */
    int32_t minOppositeDepth;            /* >0: also produce nOff / nVariant per site */
    int32_t bounds[16];                  /* --OT/--OB/--CTOT/--CTOB, index 4*(strand-1)+{R1 left,R1 right,R2 left,R2 right} */
    int32_t absoluteBounds[16];          /* --nOT/--nOB/--nCTOT/--nCTOB, same indexing */
    int32_t tile;                        /* reference positions per LDS tile; 0 = library default */
    int32_t n_slots;                     /* batches that may be in flight at once; 0 = 2 (double buffering) */
} md_dev_cfg;

/* One admitted alignment (16 bytes).  The payload of read i lives at blob + 4*off4:
 *   uint32 cigar[n_cigar]            BAM encoding, len<<4|op, ops MIDNSHP=X = 0..8
 *   uint8  seq[(l_qseq+1)/2]         BAM 4-bit codes, high nibble = even query index; padded to a multiple of 4 bytes
 *   uint8  qual[l_qseq]              raw phred bytes; padded to a multiple of 4 bytes
 * Reads are in file (coordinate) order. */
typedef struct {
    int32_t  pos;        /* 0-based leftmost reference position (bam1_core_t.pos) */
    uint32_t off4;
    uint32_t l_qseq;
    uint16_t n_cigar;
    uint8_t  strand;     /* getStrand(): 1 OT, 2 OB, 3 CTOT, 4 CTOB, 0 undeterminable (common.c:84-116) */
    uint8_t  flags;      /* bit0: read #2 (BAM flag 0x80); bit1: this read is the LATER-in-file member of its pair */
} md_read_hdr;
#define MDK_RF_READ2  1u
#define MDK_RF_SECOND 2u

/* The admitted reads of ONE interval [beg,end) of one contig -- what one chunk of extractCalls sees.
 * Host-owned; must stay valid until md_dev_upload returns (the copy is staged internally). */
typedef struct {
    int32_t  tid;
    int64_t  beg, end;          /* columns counted: beg <= pos < end (extract.c:400) */
    int32_t  n_reads;
    const md_read_hdr *hdr;     /* [n_reads] */
    const int32_t *rend;        /* [n_reads] pos + raw reference length of the CIGAR (htslib lbnode end) */
    const int32_t *mate;        /* [n_reads] index of the read this one is overlap-resolved against
                                   (the pairing custom_overlap_constructor would make, overlaps.c:121-139) or -1 */
    const uint8_t *blob;
    uint64_t blob_bytes;
} md_read_batch;

/* Result of one interval: every position with nmeth+nunmeth > 0 (or nOff > 0 when minOppositeDepth > 0),
 * ascending.  Pointers are host memory owned by the library, valid until the slot is reused. */
typedef struct {
    int64_t n_sites;
    const uint32_t *pos, *nmeth, *nunmeth;
    const uint32_t *noff, *nvar;     /* NULL unless minOppositeDepth > 0 */
    const uint8_t  *meta;            /* bits 1-2: context 0 CpG / 1 CHG / 2 CHH; bit 0: reference base is G/g */
} md_sites;

typedef struct {
    float ms_total;      /* all kernels of one launch, averaged over iters */
    float ms_pileup;     /* the pileup kernel alone, averaged over iters */
    uint64_t algo_bytes; /* algorithmic bytes of one launch (SURVEY.md 8d formula) */
    uint64_t n_sites;
} md_bench_result;

int  md_dev_count(void);                                       /* number of HIP devices, <0 on error */
int  md_dev_open(int device, const md_dev_cfg *cfg, md_dev **out);
void md_dev_close(md_dev *h);
const char *md_dev_last_error(void);
int  md_dev_tile(const md_dev *h);

/* Upload (once) the bases of a contig; letters verbatim from the FASTA (case matters: C/c, G/g). */
int  md_dev_set_reference(md_dev *h, int32_t tid, const char *seq, int64_t len);

/* slot in [0, n_slots): upload is H2D on the slot's stream; launch enqueues the kernels; download waits for
 * the slot and returns the sites.  md_dev_submit = upload + launch. */
int  md_dev_upload(md_dev *h, int slot, const md_read_batch *b);
int  md_dev_launch(md_dev *h, int slot);
int  md_dev_submit(md_dev *h, int slot, const md_read_batch *b);
int  md_dev_download(md_dev *h, int slot, md_sites *out);
int  md_dev_sync(md_dev *h);

/* Write the sites of a finished slot into caller-provided DEVICE buffers (e.g. torch tensors used for the
 * RCCL gather) instead of library host memory.  cap = capacity in sites of every buffer; noff/nvar may be NULL.
 * Returns the number of sites (>=0) or a negative error; MDK_ERR_ARG if cap is too small. */
int64_t md_dev_sites_to_device(md_dev *h, int slot, uint32_t *d_pos, uint32_t *d_nmeth, uint32_t *d_nunmeth,
                               uint32_t *d_noff, uint32_t *d_nvar, uint8_t *d_meta, int64_t cap);

/* Re-run the kernels of an uploaded slot `iters` times (inputs stay resident in HBM; results are identical
 * every time) and time them with HIP events on the slot's stream. */
int  md_dev_bench(md_dev *h, int slot, int warmup, int iters, md_bench_result *out);

/* Test hook: effective (post-trim, post-overlap-resolution) base code and quality of every query base of
 * every read of an uploaded slot, written to host arrays laid out like the blob's qual/seq (one byte per
 * base, concatenated in read order; out_off[i] = start of read i). */
int  md_dev_debug_effective(md_dev *h, int slot, uint8_t *out_base, uint8_t *out_qual, const uint64_t *out_off);

/* Pinned host memory for staging buffers (so the C host never includes HIP headers). */
void *md_host_alloc(uint64_t bytes);
void  md_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif

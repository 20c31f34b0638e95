/* mdk_io.h -- host I/O substrate (plain C, zlib only): BGZF/BAM streaming reader with parallel inflate and a
 * FASTA loader.  htslib is not available in this image, and on the MI355X path the host's only jobs are
 * "inflate, find record boundaries, admit, pack" -- the reference gets these from htslib
 * (sam_itr_next at common.c:413, faidx_fetch_seq at extract.c:381). */
#ifndef MDK_IO_H
#define MDK_IO_H
#include <stdint.h>
#include <stdio.h>

typedef struct {
    FILE *f;
    int nthreads;
    uint8_t *cbuf; size_t ccap, clen;        /* compressed bytes not yet inflated */
    uint8_t *ubuf; size_t ucap, ulen, uoff;  /* inflated bytes; records are parsed at uoff */
    int file_eof;
    int32_t n_targets; char **target_name; uint32_t *target_len;
    char *text; uint32_t l_text;
    uint64_t n_records;
    double t_inflate;                        /* seconds spent in refill (read + inflate) */
    char err[256];
} mdk_bam;

/* a decoded view of one BAM record (pointers into the reader's buffer; valid until the next mdk_bam_peek
 * after mdk_bam_advance) */
typedef struct {
    int32_t tid, pos, l_qseq, mtid, mpos;
    uint16_t flag, n_cigar; uint8_t mapq, l_qname;
    const char *qname; const uint8_t *cigar, *seq, *qual, *aux; int32_t aux_len;
    const uint8_t *raw; uint32_t raw_len;    /* whole record after block_size */
} mdk_rec;

mdk_bam *mdk_bam_open(const char *fn, int nthreads);
void mdk_bam_close(mdk_bam *b);
/* 1 = record available, 0 = end of file, <0 = error (b->err) */
int mdk_bam_peek(mdk_bam *b, mdk_rec *r);
void mdk_bam_advance(mdk_bam *b, const mdk_rec *r);
/* decode a raw record (bytes after block_size) */
int mdk_rec_parse(const uint8_t *raw, uint32_t len, mdk_rec *r);

typedef struct { int n; char **name; char **seq; int64_t *len; char *pool; } mdk_fasta;
int mdk_fasta_load(const char *fn, mdk_fasta *fa);
void mdk_fasta_free(mdk_fasta *fa);
int mdk_fasta_find(const mdk_fasta *fa, const char *name);

#endif

// mdk_crc32_core.h -- CRC-32 (IEEE 802.3 / zlib: reflected, polynomial 0xEDB88320) of one inflated BGZF member by ONE WAVEFRONT: the parts a
// lane runs on its own.  What htslib's bgzf_read_block checks for every block the reference reads (behind sam_itr_next, common.c:413).
//
// In the reflected register bit 31-k is the coefficient of x^k, and feeding the register a zero byte multiplies it by x^8 mod P.  The CRC is
// therefore linear in the message bytes, crc(M) = ~( ~0 * x^(8|M|) + raw(M) ) with raw() the register run from zero, and leading zero bytes
// leave a zero register alone.  The member is laid out RIGHT-aligned in blocks of 1024 bytes (the first block padded in front with
// zeros); lane i of the wavefront owns bytes [16i, 16i+16) of every block -- one coalesced 16-byte load per lane and block -- and runs its
// register over "its piece, 1008 zero bytes, its next piece, ...": four slice-by-4 steps per piece and one table step (Z) for the zero bytes.
// At the end lane i's register still lacks the 16(63-i) bytes to its right: the 64 registers are merged pairwise (mdk_inflate.hip, with
// wave shuffles; tools/inflate_emu.cpp --crc, over an array), one GF(2) multiplication by x^(128 * 2^level) mod P per level.
// Plain C++: compiles for the device and for the host emulation, like mdk_inflate_core.h.  Written from the definition of the code; nothing
// is taken from zlib's crc32.c.
#ifndef MDK_CRC32_CORE_H
#define MDK_CRC32_CORE_H
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define MDK_CRC_HD __host__ __device__ __forceinline__
#else
#define MDK_CRC_HD static inline
#endif
#define CRC_POLY 0xEDB88320u
struct CrcConst { uint32_t T[4][256]; uint32_t Z[4][256]; uint32_t lvl[6]; uint32_t p8[17]; };       // Z: "1008 zero bytes"; lvl[l] = x^(128 * 2^l); p8[k] = x^(8 * 2^k)
MDK_CRC_HD uint32_t crc_mulx(uint32_t a) { return (a & 1u) ? (a >> 1) ^ CRC_POLY : a >> 1; }
MDK_CRC_HD uint32_t crc_mul(uint32_t a, uint32_t b) {       // a * b mod P
    uint32_t r = 0;
    for(int k = 0; k < 32; k++) { if(a & (0x80000000u >> k)) r ^= b; b = crc_mulx(b); }
    return r;
}
static inline void crc_make_const(CrcConst &K) {
    for(uint32_t i = 0; i < 256; i++) { uint32_t c = i; for(int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1; K.T[0][i] = c; }
    for(int t = 1; t < 4; t++) for(uint32_t i = 0; i < 256; i++) K.T[t][i] = (K.T[t - 1][i] >> 8) ^ K.T[0][K.T[t - 1][i] & 255u];
    uint32_t xp = 0x80000000u;                                              // x^0
    for(int k = 0; k < 8; k++) xp = crc_mulx(xp);                           // x^8
    K.p8[0] = xp; for(int k = 1; k < 17; k++) K.p8[k] = crc_mul(K.p8[k - 1], K.p8[k - 1]);
    for(int l = 0; l < 6; l++) K.lvl[l] = K.p8[4 + l];                      // x^(128 * 2^l) = x^(8 * 2^(4+l))
    uint32_t z = 0x80000000u;                                               // x^(8 * 1008)
    for(int k = 0; k < 17; k++) if((1008u >> k) & 1u) z = crc_mul(z, K.p8[k]);
    for(int t = 0; t < 4; t++) for(uint32_t i = 0; i < 256; i++) K.Z[t][i] = crc_mul(i << (8 * t), z);
}
MDK_CRC_HD uint32_t crc_step4(const uint32_t (*T)[256], uint32_t c, uint32_t w) {
    c ^= w;
    return T[3][c & 255u] ^ T[2][(c >> 8) & 255u] ^ T[1][(c >> 16) & 255u] ^ T[0][c >> 24];
}
// the register of lane `lane` over its column of the member d[0..L)
MDK_CRC_HD uint32_t crc_lane(const uint32_t (*T)[256], const uint32_t (*Z)[256], const uint8_t *d, uint32_t L, int lane) {
    const int nblk = (int)((L + 1023u) >> 10); const int pad = nblk * 1024 - (int)L;
    uint32_t c = 0;
    for(int k = 0; k < nblk; k++) {
        const int m0 = k * 1024 + 16 * lane - pad;                         // offset in the member of this lane's piece of block k
        uint32_t w[4] = {0, 0, 0, 0};
        if(m0 >= 0) memcpy(w, d + m0, 16);
        else if(m0 > -16) for(int j = -m0; j < 16; j++) w[j >> 2] |= (uint32_t)d[m0 + j] << (8 * (j & 3));
        if(k) c = Z[0][c & 255u] ^ Z[1][(c >> 8) & 255u] ^ Z[2][(c >> 16) & 255u] ^ Z[3][c >> 24];
        c = crc_step4(T, c, w[0]); c = crc_step4(T, c, w[1]); c = crc_step4(T, c, w[2]); c = crc_step4(T, c, w[3]);
    }
    return c;
}
// from the merged register (lane 63's after the six levels) to the CRC32 of the L-byte member
MDK_CRC_HD uint32_t crc_finish(uint32_t raw, uint32_t L, const uint32_t *p8) {
    uint32_t xl = 0x80000000u;
    for(int k = 0; k < 17; k++) if((L >> k) & 1u) xl = crc_mul(xl, p8[k]);
    return ~(crc_mul(0xffffffffu, xl) ^ raw);
}
#endif

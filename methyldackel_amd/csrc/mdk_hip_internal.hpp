// mdk_hip_internal.hpp -- shared between the translation units of libmdk_hip.so (not part of the C ABI).
#ifndef MDK_HIP_INTERNAL_HPP
#define MDK_HIP_INTERNAL_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mdk_hip.h"

#define WG 512
#define WAVES (WG / 64)
#define RING 64                    // site counters: launch i uses counter i%RING and clears the next one

#define MDK_HIDDEN __attribute__((visibility("hidden")))
extern MDK_HIDDEN thread_local char g_err[512];
MDK_HIDDEN int fail(int code, const char *what, hipError_t e);
#define HIPCHK(call) do { hipError_t e_ = (call); if(e_ != hipSuccess) return fail(MDK_ERR_HIP, #call, e_); } while(0)

struct TileEnt { int first, last; };     // segments [first,last) of the batch overlap the tile (one contiguous run: the batch is coordinate sorted)

template <typename T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    int need(size_t n) {
        if(n <= cap) return 0;
        if(p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipMalloc", e);
        cap = want; return 0;
    }
    void release() { if(p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T> struct HBuf {
    T *p = nullptr; size_t cap = 0;
    int need(size_t n) {
        if(n <= cap) return 0;
        if(p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if(e != hipSuccess) return fail(MDK_ERR_NOMEM, "hipHostMalloc", e);
        cap = want; return 0;
    }
    void release() { if(p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Slot {
    hipStream_t stream = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr, k0 = nullptr, k1 = nullptr;
    DBuf<md_seg> d_seg_in; DBuf<uint8_t> d_blob;
    DBuf<TileEnt> d_tiles; HBuf<TileEnt> h_tiles;
    DBuf<md_site> d_site; DBuf<md_site_var> d_var; DBuf<md_tile_seg> d_seg; DBuf<uint32_t> d_total; DBuf<int> d_err;
    HBuf<md_site> h_site, h_sorted; HBuf<md_site_var> h_var, h_vsorted; HBuf<md_tile_seg> h_seg; HBuf<uint32_t> h_total; HBuf<int> h_err;
    DBuf<md_pr_read> d_pr; DBuf<uint32_t> d_cig; DBuf<md_pr_count> d_prc; HBuf<md_pr_count> h_prc; int pr_n = -1;      // perRead
    // caller-bound output (device memory owned by the caller)
    md_site *b_site = nullptr; md_site_var *b_var = nullptr; md_tile_seg *b_seg = nullptr; int64_t b_cap_sites = 0, b_cap_tiles = 0;
    int n_segs = 0, n_reads = 0, ntiles = 0, tid = -1, tile = 0, lds_bytes = 0; int64_t beg = 0, end = 0; uint64_t read_bytes = 0;
    bool uploaded = false, launched = false; unsigned ring = 0;
};

struct md_dev {
    int device; md_dev_cfg cfg; int tile, n_slots; bool variant;
    std::vector<Slot> slots;
    std::vector<char *> ref; std::vector<uint8_t *> refcode; std::vector<int64_t> reflen;
    uint32_t *d_hist = nullptr; int hist_cap = 0, hist_len = 0; std::vector<uint32_t> h_hist;      // mbias: rows [q][16], q < hist_cap
};


MDK_HIDDEN Slot *get_slot(md_dev *h, int slot);
MDK_HIDDEN int launch_kernels(md_dev *h, Slot *s, bool time_pileup, hipStream_t on = nullptr);
MDK_HIDDEN int64_t finish_count(md_dev *h, Slot *s);
#endif

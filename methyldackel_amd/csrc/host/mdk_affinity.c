/* mdk_affinity.c -- keep the command's threads (and with them the first touch of the staging memory the GPU uploads from) on
 * the CPUs next to the GPU.  On the two-socket MI355X box the device hangs off NUMA node 0; `extract` on the 32 Mb sample takes
 * 0.176-0.188 s bound to that node, 0.192-0.215 s unbound, 0.189-0.221 s bound to the other one (profiles/r02h_numa32mb.txt).
 * The reference has no counterpart (htslib threads float). */
#define _GNU_SOURCE
#include <dirent.h>
#include <limits.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mdk_extract.h"

typedef struct { char pci[PATH_MAX]; char dir[PATH_MAX]; } gpu_ent;
static int by_pci(const void *a, const void *b) { return strcmp(((const gpu_ent *)a)->pci, ((const gpu_ent *)b)->pci); }

static int read_line(const char *path, char *buf, size_t cap) {
    FILE *f = fopen(path, "r"); size_t n;
    if(!f) return -1;
    n = fread(buf, 1, cap - 1, f); fclose(f); buf[n] = 0;
    while(n && (buf[n - 1] == '\n' || buf[n - 1] == ' ')) buf[--n] = 0;
    return (int)n;
}

/* "0-63,128-191" -> set; -1 on anything that is not a cpu list */
static int parse_cpulist(const char *s, cpu_set_t *set) {
    int n = 0;
    CPU_ZERO(set);
    while(*s) {
        char *end; long a = strtol(s, &end, 10), b;
        if(end == s || a < 0) return -1;
        b = a;
        if(*end == '-') { s = end + 1; b = strtol(s, &end, 10); if(end == s || b < a) return -1; }
        if(b >= CPU_SETSIZE) return -1;
        for(; a <= b; a++) { if(!CPU_ISSET((int)a, set)) n++; CPU_SET((int)a, set); }
        if(*end == ',') end++; else if(*end) return -1;
        s = end;
    }
    return n;
}

int mdk_bind_to_device_node(int index) {
    const char *root = getenv("MDK_SYSFS_DRM") ? getenv("MDK_SYSFS_DRM") : "/sys/class/drm";
    DIR *d; struct dirent *e; gpu_ent *g = NULL; int ng = 0, cap = 0, bound = 0; char path[PATH_MAX + 64], line[4096];
    cpu_set_t local, cur, both; int n_cur, n_both;
    /* a *_VISIBLE_DEVICES variable renumbers (and may hide) devices: the HIP index then says nothing about the sysfs order, and a
     * wrong guess pins every thread of the command to the far socket -- float instead */
    { static const char *const vis[] = {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"}; int k;
      for(k = 0; k < 4; k++) { const char *v = getenv(vis[k]); if(v && *v) return 0; } }
    if(getenv("MDK_NO_BIND") || index < 0 || !(d = opendir(root))) return 0;
    while((e = readdir(d)) != NULL) {          /* card<N> (not card<N>-<connector>) of vendor 0x1002, in PCI address order = HIP's default order */
        const char *p = e->d_name; gpu_ent x;
        if(strncmp(p, "card", 4) || !p[4] || strspn(p + 4, "0123456789") != strlen(p + 4)) continue;
        snprintf(path, sizeof(path), "%s/%s/device/vendor", root, p);
        if(read_line(path, line, sizeof(line)) <= 0 || strcmp(line, "0x1002")) continue;
        snprintf(x.dir, sizeof(x.dir), "%s/%s/device", root, p);
        if(!realpath(x.dir, x.pci)) snprintf(x.pci, sizeof(x.pci), "%s", x.dir);
        if(ng == cap) { gpu_ent *q = realloc(g, sizeof(*g) * (size_t)(cap = cap ? cap * 2 : 8)); if(!q) { free(g); closedir(d); return 0; } g = q; }
        g[ng++] = x;
    }
    closedir(d);
    if(index < ng) {
        qsort(g, (size_t)ng, sizeof(*g), by_pci);
        snprintf(path, sizeof(path), "%s/local_cpulist", g[index].dir);
        if(read_line(path, line, sizeof(line)) > 0 && parse_cpulist(line, &local) > 0 && sched_getaffinity(0, sizeof(cur), &cur) == 0) {
            CPU_AND(&both, &local, &cur);
            n_cur = CPU_COUNT(&cur); n_both = CPU_COUNT(&both);
            /* only a node that still offers at least half of what the process may use: on a machine cut into many small NUMA
             * nodes the threads are better left to float; and never below two CPUs */
            if(n_both >= 2 && 2 * n_both >= n_cur && n_both < n_cur && sched_setaffinity(0, sizeof(both), &both) == 0) bound = n_both;
        }
    }
    free(g);
    return bound;
}

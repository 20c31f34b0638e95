/* main.c -- `MethylDackel` command of the MI355X build: `extract`, `mbias`, `perRead` on the GPU and the `mergeContext`
 * text tool (the reference's dispatcher is main.c:39-62). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include "mdk_extract.h"

static void usage_main(void) {
    fprintf(stderr, "MethylDackel (methyldackel_amd, MI355X build of the `extract` path)\n"
                    "Usage: MethylDackel <command> [options]\n\nCommands:\n"
                    "    extract  Extract methylation metrics from an alignment file in BAM format (GPU).\n"
                    "    mbias    Determine the position-dependent methylation bias in a dataset (GPU).\n"
                    "    perRead  Generate a per-read methylation summary (GPU).\n"
                    "    mergeContext   Combine single Cytosine metrics from 'MethylDackel extract' into per-CpG/CHG metrics.\n");
}
/* A finished run still holds pinned staging blocks, device memory and queues, and the kernel takes ~0.2 s to take that address space
 * down (measured: tools/exit_stack_probe.py -- between _exit and the moment the parent can reap, one task is left, the one inside exit_mmap) --
 * AFTER every output is complete and closed.  With MDK_DETACH=1, as the mold linker does, the work is done by a child and this process, which holds
 * nothing, waits only for the child's word that its outputs are closed: it returns the child's code at once and the teardown goes on behind it.  The
 * child says so in leave (below; mdk_extract.c leave_fast) through the descriptor MDK_DONE_FD names, having closed its standard streams first so
 * that a caller reading our stdout/stderr through pipes sees their end when we return.  A child that dies without a word is waited for and
 * its fate is ours.  This is OPT-IN (MDK_DETACH=1): by default -- and under a profiler, which follows the process it started -- everything runs in
 * place, and the wall clock a caller sees includes the teardown, as the CPU path's does. */
static pid_t g_child = -1;
static int open_null(void) { return open("/dev/null", O_RDWR); }
static void forward_signal(int sig) { if(g_child > 0) kill(g_child, sig); }
static int profiler_present(void) {
    const char *pre = getenv("LD_PRELOAD");
    if(getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || getenv("ROCPROF_OUTPUT_PATH")) return 1;
    return pre && (strstr(pre, "rocprof") || strstr(pre, "roctx") || strstr(pre, "rocm"));
}
static void say_done(int rc) {              /* the child's last act before it leaves (the same few lines as mdk_extract.c leave_fast) */
    const char *fdv = getenv("MDK_DONE_FD");
    fflush(stdout); fflush(stderr);
    if(fdv) {
        const int fd = atoi(fdv), nul = open_null();
        unsigned char code = (unsigned char)(rc & 0xff);
        if(nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); if(nul > 2) close(nul); }
        (void)prctl(PR_SET_PDEATHSIG, 0);
        if(write(fd, &code, 1) != 1) { /* the parent is gone: nothing to tell */ }
        close(fd);
    }
}
/* returns in the child (the work is its to do) or, when nothing was forked, in the only process; the parent never returns */
static void detach_teardown(void) {
    int pfd[2]; pid_t c; char num[16]; const pid_t me = getpid();
    static const int sigs[] = {SIGINT, SIGTERM, SIGHUP, SIGQUIT, SIGUSR1, SIGUSR2};
    if(!getenv("MDK_DETACH") || getenv("MDK_NO_DETACH") || profiler_present() || pipe(pfd)) return;      /* opt-in since round 5: by default the process that does the work is the one the caller waits for */
    fflush(stdout); fflush(stderr);
    c = fork();
    if(c < 0) { close(pfd[0]); close(pfd[1]); return; }
    if(c == 0) {
        close(pfd[0]);
        (void)prctl(PR_SET_PDEATHSIG, SIGKILL);                  /* nobody to answer to: no work without a parent */
        if(getppid() != me) _exit(1);                            /* (the parent went between fork and prctl; PID 1 is a parent like any other) */
        snprintf(num, sizeof(num), "%d", pfd[1]); setenv("MDK_DONE_FD", num, 1);
        return;
    }
    {   unsigned char code = 0; ssize_t n; size_t i; int st = 0;
        close(pfd[1]); g_child = c;
        for(i = 0; i < sizeof(sigs) / sizeof(sigs[0]); i++) { struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_handler = forward_signal; sigaction(sigs[i], &sa, NULL); }
        do n = read(pfd[0], &code, 1); while(n < 0 && errno == EINTR);
        if(n == 1) _exit(code);
        while(waitpid(c, &st, 0) < 0 && errno == EINTR) { }
        if(WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);
    }
}
/* The HIP runtime brings up every GPU it can see (hipInit on a node with eight of them does eight times the work) and this process uses one:
 * before the runtime is loaded into the process that does the work, it is told to see that one only (ROCR_VISIBLE_DEVICES; the device is then
 * number 0).  Not when the caller has restricted the devices already, and not for a rank of a several-GPU run (its peers' devices stay visible to
 * the exchange layer).  MDK_NO_RESTRICT=1 leaves the environment alone. */
static void see_own_device_only(void) {
    const char *dv = getenv("MDK_DEVICE"), *w = getenv("MDK_WORLD"); const int d = dv ? atoi(dv) : 0; char num[16];
    if(getenv("MDK_NO_RESTRICT") || getenv("ROCR_VISIBLE_DEVICES") || getenv("HIP_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES") || getenv("GPU_DEVICE_ORDINAL")) return;
    if((w && atoi(w) > 1) || getenv("MDK_TORCHRUN") || d < 0) return;
    snprintf(num, sizeof(num), "%d", d);
    setenv("ROCR_VISIBLE_DEVICES", num, 1); setenv("MDK_DEVICE_USER", num, 1); setenv("MDK_DEVICE", "0", 1);      /* (messages name the device as the user did) */
}
int main(int argc, char *argv[]) {
    if(argc == 1) { usage_main(); return 0; }
    if(!strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) { usage_main(); return 0; }
    if(!strcmp(argv[1], "-v") || !strcmp(argv[1], "--version")) { printf("0.6.1 (using HTSlib version none; methyldackel_amd MI355X build)\n"); return 0; }
    if(!strcmp(argv[1], "extract") || !strcmp(argv[1], "mbias") || !strcmp(argv[1], "perRead")) {
        /* stay on the CPUs next to the GPU (the inflate threads first-touch what the GPU uploads from); a rank of a several-GPU run
         * (csrc/host/mdk_ranks.c) takes the device its rank names */
        const char *dv = getenv("MDK_DEVICE"), *lr = getenv("LOCAL_RANK"), *mr = getenv("MDK_RANK");
        (void)mdk_bind_to_device_node(dv ? atoi(dv) : lr ? atoi(lr) : mr ? atoi(mr) : 0);
    }
    {   /* a process about to end need not unpin buffers and shut the runtime down politely (MDK_FAST_EXIT: the command leaves with
         * _exit once its outputs are closed).  A command that comes BACK here returned early -- a bad option, a missing input -- possibly
         * while the thread that warms the HIP runtime up is still inside hipInit: leave with _exit as well, so that no exit handler or
         * static destructor runs under that thread's feet and the command's own return code is what the caller sees. */
        int (*cmd)(int, char **) = !strcmp(argv[1], "extract") ? extract_main : !strcmp(argv[1], "mbias") ? mbias_main : !strcmp(argv[1], "perRead") ? perRead_main : NULL;
        if(cmd) {
            int rc;
            setenv("MDK_FAST_EXIT", "1", 0);
            see_own_device_only();
            detach_teardown();
            rc = cmd(argc - 1, argv + 1);
            say_done(rc);
            mdk_cli_quiesce();                                   /* no thread of ours is left inside the HIP runtime when the process goes */
            if(profiler_present()) return rc & 0xff;             /* (a profiler writes its files from an exit handler: under one the process leaves the ordinary way) */
            _exit(rc & 0xff);
        }
    }
    if(!strcmp(argv[1], "mergeContext")) return mergeContext_main(argc - 1, argv + 1);
    fprintf(stderr, "Unknown command!\n"); usage_main(); return -1;
}

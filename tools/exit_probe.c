/* MEASUREMENT TOOL: how long the kernel takes over the exit of a process that holds `gb` GB of touched anonymous memory, with and without
 * transparent huge pages, and when the memory is given back (MADV_DONTNEED) before the exit.  usage: exit_probe GB thp(0|1) predrop(0|1) */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(int argc,char**argv){
  size_t gb = argc>1?atol(argv[1]):3; int thp = argc>2?atoi(argv[2]):1; int pre = argc>3?atoi(argv[3]):0;
  int pfd[2]; if(pipe(pfd)) return 1;
  pid_t c=fork();
  if(!c){
    size_t n=64, each=gb*(1ull<<30)/n; void*p[64];
    for(size_t i=0;i<n;i++){ if(posix_memalign(&p[i],2<<20,each)) return 1; if(thp) madvise(p[i],each,MADV_HUGEPAGE); else madvise(p[i],each,MADV_NOHUGEPAGE); memset(p[i],1,each);} 
    double t0=now();
    if(pre){ for(size_t i=0;i<n;i++) madvise(p[i],each,MADV_DONTNEED); }
    double t=now(); if(write(pfd[1],&t,sizeof t)<0||write(pfd[1],&t0,sizeof t0)<0) _exit(1); _exit(0);
  }
  int st; waitpid(c,&st,0); double t1=now(),t,t0; if(read(pfd[0],&t,sizeof t)<0||read(pfd[0],&t0,sizeof t0)<0) return 1;
  printf("gb=%zu thp=%d pre=%d: predrop %.3f s, exit->reaped %.3f s\n",gb,thp,pre,t-t0,t1-t);
}

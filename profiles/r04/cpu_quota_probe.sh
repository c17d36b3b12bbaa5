#!/bin/bash
# what the GPU box gives a process in CPU time: cgroup quota, visible CPUs, and the wall time of N threads of fixed private work
echo "nproc: $(nproc); cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null); cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr "\n" " "; echo
cat > /tmp/spin.c <<'C'
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static void *work(void *p) { volatile unsigned long x = 1; for (unsigned long i = 0; i < 600000000ul; ++i) x = x * 6364136223846793005ul + 1442695040888963407ul; return (void *)x; }
int main(int argc, char **argv) {
    int n = atoi(argv[1]); pthread_t t[512]; struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < n; ++i) pthread_create(&t[i], 0, work, 0);
    for (int i = 0; i < n; ++i) pthread_join(t[i], 0);
    clock_gettime(CLOCK_MONOTONIC, &b);
    printf("%d threads: %.3f s\n", n, (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) * 1e-9);
}
C
gcc -O1 -pthread /tmp/spin.c -o /tmp/spin
for n in 1 8 16 32 48 64 96 128 256; do /tmp/spin $n; done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr "\n" " "; echo

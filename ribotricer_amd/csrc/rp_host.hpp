// Host-side helpers shared by the text / file ends of the path (index parser, BAM reader, TSV
// renderer, rp_phase_score_csr_host).
#pragma once

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <thread>

namespace rphost {

// Threads this process may really use: the scheduler affinity, capped by the cgroup CPU quota
// (a GPU box shows 256 CPUs to std::thread::hardware_concurrency() and owns 16 of them).  The
// same rule as ribotricer_amd/_lib.py usable_cores().
inline int usable_threads()
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char quota[32];
        long period = 0;
        if (std::fscanf(fh, "%31s %ld", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
            long q = std::atol(quota);
            if (q > 0) {
                long cap = (q + period - 1) / period;
                if (cap >= 1 && cap < n) n = (int)cap;
            }
        }
        std::fclose(fh);
    }
    return n;
}

}  // namespace rphost

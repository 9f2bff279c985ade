// AddressSanitizer / UBSan driver for the BAM front end (rp_bam.hpp): reads a well-formed BAM,
// then 1 500 truncated / byte-mutated copies -- the reader must fail cleanly or succeed, never
// touch memory it does not own.  Built and run by tests/test_host_sanitizers_cpu.py.
#include "rp_bam.hpp"
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
int main(int argc, char **argv)
{
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    const std::string data = ss.str();
    rpbam::Split sp;
    int rc = rpbam::split_bam(argv[1], 0, nullptr, 0, sp);
    printf("split rc=%d rows=%zu total=%lld valid=%lld\n", rc, sp.pos.size(), (long long)sp.total, (long long)sp.valid);
    if (rc != 0) return 1;
    std::mt19937 rng(11);
    const std::string tmp = std::string(argv[2]);
    int ok = 0, bad = 0;
    for (int rep = 0; rep < 600; ++rep) {
        std::string t = rep % 3 == 0 ? data.substr(0, rng() % (data.size() + 1)) : data;
        const int flips = rep % 3 == 1 ? 1 + rng() % 4 : (rep % 3 == 2 ? 40 : 0);
        for (int k = 0; k < flips && !t.empty(); ++k) t[rng() % t.size()] = (char)(rng() & 0xff);
        FILE *o = fopen(tmp.c_str(), "wb");
        fwrite(t.data(), 1, t.size(), o);
        fclose(o);
        rpbam::Split s2;
        const int32_t want[3] = {28, 29, 30};
        const int r2 = rpbam::split_bam(tmp.c_str(), rep & 1, rep % 5 == 0 ? want : nullptr, rep % 5 == 0 ? 3 : 0, s2);
        (r2 == 0 ? ok : bad) += 1;
    }
    printf("ok mutated: %d readable, %d rejected\n", ok, bad);
    // a damaged ISIZE trailer (0xF0000000 claimed for the first block) must be rejected as a format
    // error, not used to size a 3.9 GB inflate buffer (BGZF blocks hold <= 64 KiB)
    {
        std::string t = data;
        const size_t bsize = (size_t)(unsigned char)t[16] | ((size_t)(unsigned char)t[17] << 8);  // BC subfield of block 0
        const size_t isize_at = bsize + 1 - 4;
        t[isize_at] = 0; t[isize_at + 1] = 0; t[isize_at + 2] = 0; t[isize_at + 3] = (char)0xF0;
        FILE *o = fopen(tmp.c_str(), "wb");
        fwrite(t.data(), 1, t.size(), o);
        fclose(o);
        rpbam::Split s3;
        const int r3 = rpbam::split_bam(tmp.c_str(), 0, nullptr, 0, s3);
        printf("huge isize rc=%d\n", r3);
        if (r3 != rpbam::kFormat) return 2;
    }
    return 0;
}

// AddressSanitizer / UBSan driver for the host-side C++ (rp_index.hpp, rp_format.hpp): parses the
// fixture index and 2 000 randomly mutated / truncated copies, renders rows into buffers that are
// too small, prints random doubles.  Built and run by tests/test_host_sanitizers_cpu.py.
#include "rp_format.hpp"
#include "rp_index.hpp"
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
int main(int argc, char **argv)
{
    // index parser on the fixture + mutated variants
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    std::string text = ss.str();
    rpidx::Index ix;
    int rc = rpidx::parse(text.data(), text.size(), true, ix);
    printf("parse rc=%d n=%zu iv=%zu groups=%zu\n", rc, ix.length.size(), ix.iv_start.size(), ix.group_lo.size());
    std::mt19937 rng(7);
    for (int rep = 0; rep < 2000; ++rep) {  // random byte mutations / truncations must never crash
        std::string t = text.substr(0, rng() % (text.size() + 1));
        for (int k = 0; k < 5 && !t.empty(); ++k) t[rng() % t.size()] = "\t\n-,0x9 \r"[rng() % 9];
        rpidx::Index jx;
        rpidx::parse(t.data(), t.size(), rep & 1, jx);
    }
    // formatter: random rows, tight buffers
    const long long n = ix.length.size();
    std::vector<int64_t> off(n + 1, 0);
    for (long long i = 0; i < n; ++i) off[i + 1] = off[i] + ix.length[i];
    std::vector<int32_t> counts(off[n]);
    for (auto &c : counts) c = (int32_t)(rng() % 7 == 0 ? (rng() % 3 ? rng() % 100 : -(int)(rng() % 1000000000)) : 0);
    std::vector<double> phase(n); std::vector<int32_t> valid(n); std::vector<int64_t> rc64(n); std::vector<uint8_t> st(n);
    for (long long i = 0; i < n; ++i) { phase[i] = (double)rng() / 4294967296.0 * (rng() % 2 ? 1e-9 : 1.0); valid[i] = rng() % 500; rc64[i] = (int64_t)rng() * (rng() % 1000); st[i] = rng() % 2; }
    rpfmt::RowInputs in{counts.data(), off.data(), phase.data(), valid.data(), rc64.data(), st.data(), ix.head.data(), ix.head_off.data(), ix.tail.data(), ix.tail_off.data()};
    size_t total = 0;
    for (size_t cap : {size_t(64), size_t(700), size_t(5000), size_t(1 << 20)}) {
        std::vector<char> buf(cap);
        long long cur = 0; size_t len = 0, need = 0; int grow = 0;
        while (cur < n) {
            long long nx = rpfmt::format_rows(in, n, true, cur, buf.data(), buf.size(), &len, &need);
            if (need) { buf.resize(need); ++grow; cur = nx; continue; }
            total += len; cur = nx;
        }
        printf("cap=%zu grow=%d\n", cap, grow);
    }
    char b[32];
    for (int k = 0; k < 300000; ++k) { uint64_t bits = ((uint64_t)rng() << 32) | rng(); double v; memcpy(&v, &bits, 8); int l = rpfmt::double_repr(v, b); if (l <= 0 || l > 25) { printf("bad len %d\n", l); return 1; } }
    printf("ok total=%zu\n", total);
    return 0;
}

// gen_big_index.cpp -- a human-scale synthetic ribotricer index for end-to-end timing
// (scripts/bench_export_big.py).  Transcripts of 1-8 exons on 24 chromosomes x 2 strands, several
// candidate ORFs per transcript (one `annotated`, the others uORF / dORF / overlapping: nested
// sub-ranges of the transcript, as prepare-orfs emits them, so ORFs of a transcript share coverage),
// 65 % of the ORFs 60-150 nt.  Writes
//   <prefix>_candidate_orfs.tsv   the index (11 columns, format of prepare_orfs.py:370-404)
//   <prefix>_exons.bin            per exon of every ORF: int32 chrom, int32 strand, int64 start, int64 end
// usage: gen_big_index <prefix> <n_orfs> [seed]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    const std::string prefix = argv[1];
    const long long n_orfs = atoll(argv[2]);
    std::mt19937_64 rng(argc > 3 ? atoll(argv[3]) : 7);
    FILE *tsv = fopen((prefix + "_candidate_orfs.tsv").c_str(), "wb");
    FILE *bin = fopen((prefix + "_exons.bin").c_str(), "wb");
    if (!tsv || !bin) return 2;
    static char buf[1 << 20];
    setvbuf(tsv, buf, _IOFBF, sizeof(buf));
    fputs("ORF_ID\tORF_type\ttranscript_id\ttranscript_type\tgene_id\tgene_name\tgene_type\tchrom\tstrand\tstart_codon\tcoordinate\n", tsv);
    const char *types[] = {"uORF", "dORF", "overlap_uORF", "overlap_dORF", "super_uORF", "super_dORF", "novel"};
    long long cursor[24][2];
    for (auto &c : cursor) c[0] = c[1] = 10000;
    auto uni = [&](long long lo, long long hi) { return lo + (long long)(rng() % (unsigned long long)(hi - lo + 1)); };
    long long made = 0, tx = 0;
    std::vector<long long> ex_s, ex_e;
    std::string line;
    while (made < n_orfs) {
        const int chrom = (int)(tx % 24), strand = (int)((tx / 24) % 2);
        const int n_ex = (int)uni(1, 8);
        ex_s.clear();
        ex_e.clear();
        long long pos = cursor[chrom][strand], tlen = 0;
        for (int k = 0; k < n_ex; ++k) {
            const long long len = uni(60, 420);
            ex_s.push_back(pos);
            ex_e.push_back(pos + len - 1);
            tlen += len;
            pos += len + uni(80, 3000);
        }
        cursor[chrom][strand] = pos + uni(200, 5000);
        const int n_here = (int)uni(1, 10);
        for (int o = 0; o < n_here && made < n_orfs; ++o) {
            // ORF in transcript coordinates [a, a + len)
            long long len = (rng() % 100 < 65) ? 3 * uni(20, 50) : 3 * uni(51, 600);
            if (len > tlen) len = tlen / 3 * 3;
            if (len < 60) continue;
            const long long a = uni(0, tlen - len);
            // genomic intervals: walk the exons (for '-' the transcript runs right to left)
            long long t0 = strand ? tlen - (a + len) : a, left = len, at = 0;
            line.clear();
            char tmp[64];
            int first = 1;
            std::string coords;
            for (int k = 0; k < n_ex && left > 0; ++k) {
                const long long el = ex_e[k] - ex_s[k] + 1;
                if (t0 >= at + el) { at += el; continue; }
                const long long s = ex_s[k] + (t0 > at ? t0 - at : 0);
                long long e = ex_e[k];
                if (e - s + 1 > left) e = s + left - 1;
                left -= e - s + 1;
                at += el;
                t0 = at;
                snprintf(tmp, sizeof(tmp), "%s%lld-%lld", first ? "" : ",", s, e);
                coords += tmp;
                first = 0;
                const int32_t hdr[2] = {chrom, strand};
                const int64_t se[2] = {s, e};
                fwrite(hdr, 4, 2, bin);
                fwrite(se, 8, 2, bin);
            }
            const char *type = o == 0 ? "annotated" : types[rng() % 7];
            fprintf(tsv, "x\t%s\tENST%011lld\tprotein_coding\tENSG%011lld\tGENE%lld\tprotein_coding\tchr%d\t%c\tATG\t%s\n", type, tx, tx / 3,
                    tx / 3, chrom + 1, strand ? '-' : '+', coords.c_str());
            ++made;
        }
        ++tx;
    }
    fclose(tsv);
    fclose(bin);
    fprintf(stderr, "%lld ORFs on %lld transcripts\n", made, tx);
    return 0;
}

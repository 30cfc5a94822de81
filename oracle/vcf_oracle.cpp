// vcf_oracle.cpp — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// Restatement of the reference's FASTA lookup (src/fasta_query.cpp:18-185) and VCF writer
// (SVCaller::saveToVCF + getReadDepth, src/sv_caller.cpp:1067-1344), one record at a time with host
// depth vectors, in the straight-line form of the reference so that the product writer (batched,
// depth gathered on the device) has an independent checker.
//
// PARITY UNPINNED by a reference build: both TUs include utils.h -> <htslib/sam.h> (absent here).
// Pinned by the record the reference's own test quotes (tests/test_general.py:124; see
// tests/test_vcf_writer.py::test_reference_test_record) and by line-by-line restatement.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Genome {
    std::string path;
    std::vector<std::string> names;                 // insertion order, then sorted (:78)
    std::map<std::string, std::string> seq;         // keyed lookups only, so the container kind is not observable
};

// fasta_query.cpp:18-81
bool load_genome(const char *path, Genome &g)
{
    g.path = path;
    std::ifstream in(path);
    if (!in.is_open()) return false;
    std::string chr, seq, line;
    while (std::getline(in, line)) {
        if (line[0] == '>') {                       // line[0] of an empty std::string is '\0'
            if (chr != "") { g.names.push_back(chr); g.seq[chr] = seq; seq = ""; }
            chr = line.substr(1);
            size_t sp = chr.find(" ");
            if (sp != std::string::npos) chr.erase(sp);
        } else {
            seq += line;
        }
    }
    if (chr != "") { g.names.push_back(chr); g.seq[chr] = seq; }
    std::sort(g.names.begin(), g.names.end());
    return true;
}

// fasta_query.cpp:88-102; found=false when the contig is unknown (unordered_map::at throws there)
std::string query(const Genome &g, const std::string &chr, uint32_t a, uint32_t b, bool *found)
{
    auto it = g.seq.find(chr);
    *found = it != g.seq.end();
    if (!*found) return "";
    a--; b--;
    if (b >= it->second.length() || a > b) return "";
    return it->second.substr(a, (size_t)(b - a) + 1);
}

// fasta_query.cpp:139-162
std::string contig_header(const Genome &g)
{
    std::string out;
    for (const auto &kv : g.seq) out += "##contig=<ID=" + kv.first + ",length=" + std::to_string(kv.second.length()) + ">\n";   // std::map iterates in std::sort order
    if (!out.empty()) out.pop_back();
    return out;
}

const char *type_name(int t)    // sv_types.h:28-37
{
    switch (t) { case -1: return "UNKNOWN"; case 0: return "DEL"; case 1: return "DUP"; case 2: return "INV"; case 3: return "INS";
                 case 4: return "BND"; case 5: return "NEUTRAL"; case 6: return "LOH"; }
    return "?";
}
const char *gt_name(int g)      // sv_types.h:50-55
{
    switch (g) { case 0: return "0/0"; case 1: return "0/1"; case 2: return "1/1"; case 3: return "./."; }
    return "?";
}
std::string aln_names(uint32_t flags)   // sv_types.h:112-123
{
    static const char *names[10] = {"CIGARINS", "CIGARDEL", "CIGARCLIP", "SPLIT", "SPLITDIST1", "SPLITDIST2", "SPLITINV", "SUPPINV", "HMM", "UNKNOWN"};
    std::string out;
    for (int i = 0; i < 10; i++) if (flags >> i & 1) { if (!out.empty()) out += ","; out += names[i]; }
    return out;
}

}  // namespace

extern "C" {

struct orc_vcf_call {           // same layout as the host mirror's test POD (48 bytes)
    uint32_t start, end;
    int32_t  sv_type, cluster_size;
    double   hmm_likelihood;
    int64_t  id;
    uint32_t aln_flags;
    int32_t  genotype, cn_state, aln_offset;
};

// ReferenceGenome::query; returns the length (copying at most cap bytes), -1 unknown contig, -2 unreadable file
int64_t orc_fasta_query(const char *fasta, const char *chr, uint32_t a, uint32_t b, char *buf, uint64_t cap)
{
    Genome g;
    if (!load_genome(fasta, g)) return -2;
    bool found;
    std::string s = query(g, chr, a, b, &found);
    if (!found) return -1;
    if (buf && cap) memcpy(buf, s.data(), (size_t)std::min<uint64_t>(cap, s.size()));
    return (int64_t)s.size();
}

// getContigHeader + '\x01' + '\n'-joined getChromosomes
int64_t orc_fasta_describe(const char *fasta, char *buf, uint64_t cap)
{
    Genome g;
    if (!load_genome(fasta, g)) return -2;
    std::string s = contig_header(g) + '\x01';
    for (size_t i = 0; i < g.names.size(); i++) { if (i) s += '\n'; s += g.names[i]; }
    if (buf && cap) memcpy(buf, s.data(), (size_t)std::min<uint64_t>(cap, s.size()));
    return (int64_t)s.size();
}

// saveToVCF over contigs in the order given (the reference iterates an unordered_map; order is the caller's business).
// depth[t] may be null: a contig without a depth map, which is an error (-3) as soon as one of its records reaches the
// depth lookup (chr_pos_depth_map.at(chr), :1306). Returns 0, or a negative code; counts = {total, unclassified, gap-filtered}.
int orc_save_vcf(const char *out_path, const char *fasta, const char *gap_path, const char *file_date, int n_contigs,
                 const char *const *chr_names, const uint64_t *call_off, const orc_vcf_call *calls, const char *const *alts,
                 const uint32_t *const *depth, const uint64_t *depth_len, int32_t *counts)
{
    Genome g;
    if (!load_genome(fasta, g)) return -2;
    std::map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> gaps;
    const bool have_gaps = gap_path && *gap_path;
    if (have_gaps) {                                                        // :1073-1099
        std::ifstream gs(gap_path);
        if (!gs.is_open()) return -4;
        std::string line;
        while (std::getline(gs, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream iss(line);
            std::string chr; uint32_t s, e;
            if (!(iss >> chr >> s >> e)) continue;
            gaps[chr].emplace_back(s, e);
        }
    }
    std::ofstream out(out_path);
    if (!out.is_open()) return -5;
    const std::string method = "ContextSV v1.0.0";                           // :1163 with include/version.h
    out << "##fileformat=VCFv4.2" << std::endl;                              // :1150-1151
    if (file_date && *file_date) out << "##fileDate=" << file_date << std::endl;
    else { char b[80]; time_t t; time(&t); strftime(b, sizeof b, "%Y%m%d", localtime(&t)); out << "##fileDate=" << b << std::endl; }
    out << "##source=" << method << std::endl;
    out << "##reference=" << g.path << std::endl;                            // :1127-1146
    out << contig_header(g) << std::endl;
    out << "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">" << std::endl;
    out << "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">" << std::endl;
    out << "##INFO=<ID=SVLEN,Number=1,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">" << std::endl;
    out << "##INFO=<ID=SVMETHOD,Number=1,Type=String,Description=\"Method used to call the structural variant\">" << std::endl;
    out << "##INFO=<ID=ALN,Number=1,Type=String,Description=\"Feature used to identify the structural variant\">" << std::endl;
    out << "##INFO=<ID=HMM,Number=1,Type=Float,Description=\"HMM likelihood\">" << std::endl;
    out << "##INFO=<ID=LOH,Number=0,Type=Flag,Description=\"Site shows loss of heterozygosity\">" << std::endl;
    out << "##INFO=<ID=SUPPORT,Number=1,Type=Integer,Description=\"Number of reads supporting the variant\">" << std::endl;
    out << "##INFO=<ID=CLUSTER,Number=1,Type=Integer,Description=\"Cluster size\">" << std::endl;
    out << "##INFO=<ID=CN,Number=1,Type=Integer,Description=\"Copy number state\">" << std::endl;
    out << "##INFO=<ID=ALNOFFSET,Number=1,Type=Integer,Description=\"Read vs. reference alignment offset\">" << std::endl;
    out << "##FILTER=<ID=PASS,Description=\"All filters passed\">" << std::endl;
    out << "##FILTER=<ID=LowQual,Description=\"Low quality\">" << std::endl;
    out << "##FILTER=<ID=AssemblyGap,Description=\"Assembly gap\">" << std::endl;
    out << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">" << std::endl;
    out << "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth at the variant site (sum of start and end positions)\">" << std::endl;
    out << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE" << std::endl;   // :1171-1172

    int total = 0, unclassified = 0, gap_filtered = 0;
    for (int t = 0; t < n_contigs; t++) {
        const std::string chr = chr_names[t];
        for (uint64_t i = call_off[t]; i < call_off[t + 1]; i++) {
            const orc_vcf_call &c = calls[i];
            uint32_t start = c.start, end = c.end;
            int len = (int)(end - start + 1);                                 // :1187
            std::string ref = ".", alt = alts[i];
            std::string filter = "PASS";
            if (c.cn_state < 0 || c.cn_state > 6) return -6;                  // CNVTypeMap.at throws (:1199)
            const bool loh = c.cn_state == 4;                                 // sv_types.h:96-104
            if (c.sv_type == -1 || c.sv_type == 5) { unclassified++; continue; }   // :1203-1208
            total++;
            if (have_gaps && gaps.count(chr)) {                               // :1211-1239
                bool inside = false;
                for (const auto &gp : gaps[chr]) {
                    uint32_t os = std::max(start, gp.first + 1), oe = std::min(end, gp.second + 1);
                    if (os <= oe) {
                        uint32_t olen = oe - os + 1;
                        if ((double)olen / (double)len > 0.2) { inside = true; break; }
                    }
                }
                if (inside) { filter = "AssemblyGap"; gap_filtered++; }
            }
            bool found = true;
            if (c.sv_type == 0) {                                             // DEL :1242-1260
                uint32_t prev = (uint32_t)std::max(1, (int)start - 1);
                ref = query(g, chr, prev, end, &found);
                if (!found) return -7;
                if (ref != "") alt = std::string(1, ref.at(0));
                else { ref = "N"; alt = "<DEL>"; }
                len = -1 * len;
                start = prev;
            } else if (c.sv_type == 3) {                                      // INS :1265-1287
                if ((int)start > 1) {
                    uint32_t prev = start - 1;
                    ref = query(g, chr, prev, prev, &found);
                    if (!found) return -7;
                    start = prev;
                    if (ref != "") { if (alt != "<INS>") alt.insert(0, ref); }
                    else { ref = "N"; alt = "<INS>"; }
                } else {
                    continue;
                }
                end = start;
            } else {
                ref = "N";                                                    // :1289-1291
            }
            for (char &b : ref)                                               // :1295-1305
                if (strchr("RYKMSWBDHVrykmswbdhv", b) && b != '\0') b = 'N';
            if (!depth[t]) return -3;                                         // :1306
            int dp = 0;                                                       // :1332-1344
            if ((uint64_t)start < depth_len[t]) dp += (int)depth[t][start];
            char hmm[64];
            snprintf(hmm, sizeof hmm, "%f", c.hmm_likelihood);                // std::to_string(double)
            std::string info = "END=" + std::to_string(end) + ";SVTYPE=" + type_name(c.sv_type) + ";SVLEN=" + std::to_string(len) + ";SVMETHOD=" + method +
                               ";ALN=" + aln_names(c.aln_flags) + ";HMM=" + hmm + ";SUPPORT=" + std::to_string(dp) + ";CLUSTER=" + std::to_string(c.cluster_size) +
                               ";ALNOFFSET=" + std::to_string(c.aln_offset) + ";CN=" + std::to_string(c.cn_state) + (loh ? ";LOH" : "");
            out << chr << "\t" << start << "\t" << "." << "\t" << ref << "\t" << alt << "\t" << "." << "\t" << filter << "\t" << info << "\t" << "GT:DP" << "\t"
                << gt_name(c.genotype) << ":" << dp << std::endl;             // :1317
        }
    }
    out.close();
    if (counts) { counts[0] = total; counts[1] = unclassified; counts[2] = gap_filtered; }
    return 0;
}

}  // extern "C"

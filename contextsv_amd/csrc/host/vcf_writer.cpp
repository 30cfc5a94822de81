#include "vcf_writer.h"

#include <algorithm>
#include <ctime>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "log.h"
#include "sv_types.h"

using namespace sv_types;

namespace {
constexpr int kVersionMajor = 1, kVersionMinor = 0, kVersionPatch = 0;   // include/version.h:5-7

// IUPAC ambiguity codes, either case, become N in REF (:1295-1305)
struct AmbiguityTable {
    bool amb[256] = {};
    AmbiguityTable()
    {
        for (const char *c = "RYKMSWBDHV"; *c; c++) {
            amb[(unsigned char)*c] = true;
            amb[(unsigned char)(*c + ('a' - 'A'))] = true;
        }
    }
};
const AmbiguityTable kAmb;

// one output line, split around the two places the depth value goes
struct Record {
    std::string head;     // CHROM .. ";HMM=<x>;SUPPORT="
    std::string tail;     // ";CLUSTER=.." .. "\tGT:DP\t<GT>:"
    uint32_t pos;         // POS after the preceding-base shift: where SUPPORT / DP is read
};

const char *const kHeaderLines[] = {
    "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the variant described in this record\">",
    "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">",
    "##INFO=<ID=SVLEN,Number=1,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">",
    "##INFO=<ID=SVMETHOD,Number=1,Type=String,Description=\"Method used to call the structural variant\">",
    "##INFO=<ID=ALN,Number=1,Type=String,Description=\"Feature used to identify the structural variant\">",
    "##INFO=<ID=HMM,Number=1,Type=Float,Description=\"HMM likelihood\">",
    "##INFO=<ID=LOH,Number=0,Type=Flag,Description=\"Site shows loss of heterozygosity\">",
    "##INFO=<ID=SUPPORT,Number=1,Type=Integer,Description=\"Number of reads supporting the variant\">",
    "##INFO=<ID=CLUSTER,Number=1,Type=Integer,Description=\"Cluster size\">",
    "##INFO=<ID=CN,Number=1,Type=Integer,Description=\"Copy number state\">",
    "##INFO=<ID=ALNOFFSET,Number=1,Type=Integer,Description=\"Read vs. reference alignment offset\">",
    "##FILTER=<ID=PASS,Description=\"All filters passed\">",
    "##FILTER=<ID=LowQual,Description=\"Low quality\">",
    "##FILTER=<ID=AssemblyGap,Description=\"Assembly gap\">",
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth at the variant site (sum of start and end positions)\">",
};
}  // namespace

std::string svMethodString()
{
    return "ContextSV v" + std::to_string(kVersionMajor) + "." + std::to_string(kVersionMinor) + "." + std::to_string(kVersionPatch);
}

void HostDepthSource::depthAt(const std::string &chr, const std::vector<uint32_t> &pos, std::vector<int32_t> &out) const
{
    const std::vector<uint32_t> &d = map.at(chr);
    out.resize(pos.size());
    for (size_t i = 0; i < pos.size(); i++) out[i] = pos[i] < d.size() ? (int32_t)d[pos[i]] : -1;
}

void ShardDepthSource::depthAt(const std::string &chr, const std::vector<uint32_t> &pos, std::vector<int32_t> &out) const
{
    csv_shard *sh = shards.at(chr);
    out.resize(pos.size());
    if (pos.empty()) return;
    if (csvgpu_depth_lookup_resident(ctx, sh, pos.data(), pos.size(), out.data()) != CSV_OK)
        throw std::runtime_error(std::string("depth lookup: ") + csvgpu_last_error(ctx));
}

bool loadAssemblyGaps(const std::string &path, std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> &gaps)
{
    std::ifstream in(path);
    if (!in.is_open()) return false;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream iss(line);
        std::string chr;
        uint32_t start, end;
        if (!(iss >> chr >> start >> end)) {
            printError("Failed to parse assembly gap file line: " + line);
            continue;
        }
        gaps[chr].emplace_back(start, end);
    }
    return true;
}

VCFCounts writeVCF(std::ostream &out, const std::vector<std::pair<std::string, const std::vector<SVCall> *>> &contigs, const VCFOptions &opt,
                   const std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> &gaps, const ReferenceGenome &ref_genome,
                   const DepthSource &depth)
{
    VCFCounts counts;
    const std::string sv_method = svMethodString();

    std::string date = opt.file_date;
    if (date.empty()) {
        char buf[80];
        time_t raw;
        time(&raw);
        strftime(buf, sizeof buf, "%Y%m%d", localtime(&raw));
        date = buf;
    }
    std::string text;
    text += "##fileformat=VCFv4.2\n";
    text += "##fileDate=" + date + "\n";
    text += "##source=" + sv_method + "\n";
    text += "##reference=" + ref_genome.getFilepath() + "\n";
    text += ref_genome.getContigHeader() + "\n";
    for (const char *line : kHeaderLines) { text += line; text += '\n'; }
    text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n";
    out << text;

    std::vector<Record> recs;
    std::vector<uint32_t> positions;
    std::vector<int32_t> depths;
    for (const auto &entry : contigs) {
        const std::string &chr = entry.first;
        printMessage("Saving SV calls for " + chr + "...");
        recs.clear();
        positions.clear();
        const auto gap_it = gaps.find(chr);
        for (const SVCall &sv : *entry.second) {
            uint32_t start = sv.start, end = sv.end;
            int sv_length = (int)(end - start + 1);
            std::string ref_allele = ".", alt_allele = sv.alt_allele;
            const std::string genotype = getGenotypeString(sv.genotype);
            const std::string aln = getSVAlignmentTypeString(sv.aln_type);
            const char *filter = "PASS";
            const char *loh = getSVTypeFromCNState(sv.cn_state) == SVType::LOH ? ";LOH" : "";   // throws for states outside 0..6, as the table lookup does (:1199)

            if (sv.sv_type == SVType::UNKNOWN || sv.sv_type == SVType::NEUTRAL) { counts.unclassified++; continue; }
            counts.total++;

            // more than 20 % of the call inside one assembly gap (BED rows are 0-based; :1212-1239)
            if (gap_it != gaps.end()) {
                for (const auto &gap : gap_it->second) {
                    const uint32_t ov_start = std::max(start, gap.first + 1), ov_end = std::min(end, gap.second + 1);
                    if (ov_start <= ov_end && (double)(ov_end - ov_start + 1) / (double)sv_length > 0.2) {
                        filter = "AssemblyGap";
                        counts.assembly_gap_filtered++;
                        break;
                    }
                }
            }

            if (sv.sv_type == SVType::DEL) {
                // REF = preceding base + deleted bases, ALT = preceding base, POS on the preceding base (:1242-1260)
                const uint32_t preceding = (uint32_t)std::max(1, (int)start - 1);
                ref_allele = ref_genome.query(chr, preceding, end);
                if (ref_allele != "") {
                    alt_allele = ref_allele.substr(0, 1);
                } else {
                    ref_allele = "N";
                    alt_allele = "<DEL>";
                    std::cerr << "Warning: Reference allele is empty for deletion at " << chr << ":" << start << "-" << end << std::endl;
                }
                sv_length = -sv_length;
                start = preceding;
            } else if (sv.sv_type == SVType::INS) {
                // POS and END on the preceding base, its letter in front of a sequence ALT (:1265-1287)
                if ((int)start > 1) {
                    start -= 1;
                    ref_allele = ref_genome.query(chr, start, start);
                    if (ref_allele != "") {
                        if (alt_allele != "<INS>") alt_allele.insert(0, ref_allele);
                    } else {
                        ref_allele = "N";
                        alt_allele = "<INS>";
                        std::cerr << "Warning: Reference allele is empty for insertion at " << chr << ":" << start << "-" << end << std::endl;
                    }
                } else {
                    std::cerr << "Error: Insertion at the first position " << chr << ":" << start << "-" << end << std::endl;
                    continue;
                }
                end = start;
            } else {
                ref_allele = "N";                 // DUP / INV / BND convention (:1289-1291)
            }
            for (char &base : ref_allele) if (kAmb.amb[(unsigned char)base]) base = 'N';

            Record r;
            r.pos = start;
            r.head.reserve(ref_allele.size() + alt_allele.size() + 192);
            r.head += chr; r.head += '\t'; r.head += std::to_string(start); r.head += "\t.\t"; r.head += ref_allele; r.head += '\t';
            r.head += alt_allele; r.head += "\t.\t"; r.head += filter;
            r.head += "\tEND=" + std::to_string(end) + ";SVTYPE=" + getSVTypeString(sv.sv_type) + ";SVLEN=" + std::to_string(sv_length) +
                      ";SVMETHOD=" + sv_method + ";ALN=" + aln + ";HMM=" + std::to_string(sv.hmm_likelihood) + ";SUPPORT=";
            r.tail = ";CLUSTER=" + std::to_string(sv.cluster_size) + ";ALNOFFSET=" + std::to_string(sv.aln_offset) + ";CN=" +
                     std::to_string(sv.cn_state) + loh + "\tGT:DP\t" + genotype + ":";
            positions.push_back(r.pos);
            recs.push_back(std::move(r));
        }
        if (recs.empty()) continue;

        depth.depthAt(chr, positions, depths);    // one gather per contig
        text.clear();
        for (size_t i = 0; i < recs.size(); i++) {
            int read_depth = depths[i];
            if (read_depth < 0) {                 // getReadDepth's out_of_range branch (:1337-1341)
                printError("Warning: Read depth for position " + std::to_string(recs[i].pos) + " is out of range of the depth map");
                read_depth = 0;
            }
            const std::string d = std::to_string(read_depth);
            text += recs[i].head; text += d; text += recs[i].tail; text += d; text += '\n';
        }
        out << text;
    }
    out.flush();
    return counts;
}

bool saveToVCF(const std::unordered_map<std::string, std::vector<SVCall>> &sv_calls, const VCFOptions &opt, const ReferenceGenome &ref_genome,
               const DepthSource &depth, VCFCounts *counts_out)
{
    std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> gaps;
    if (!opt.assembly_gaps.empty()) {
        printMessage("Loading assembly gap file: " + opt.assembly_gaps);
        if (!loadAssemblyGaps(opt.assembly_gaps, gaps)) {
            printError("Failed to open assembly gap file: " + opt.assembly_gaps);
            return false;
        }
        printMessage("Loaded " + std::to_string(gaps.size()) + " assembly gaps.");
    }
    const std::string path = opt.output_dir + "/output.vcf";
    printMessage("Writing VCF file to " + path);
    std::ofstream vcf(path);
    if (!vcf.is_open()) {
        printError("Failed to open VCF file for writing.");
        return false;
    }
    std::vector<std::pair<std::string, const std::vector<SVCall> *>> order;
    order.reserve(sv_calls.size());
    for (const auto &entry : sv_calls) order.emplace_back(entry.first, &entry.second);
    const VCFCounts c = writeVCF(vcf, order, opt, gaps, ref_genome, depth);
    vcf.close();
    printMessage("Finished writing VCF file. Total records: " + std::to_string(c.total));
    if (c.unclassified > 0) printMessage("Total unclassified SVs: " + std::to_string(c.unclassified));
    printMessage("Total filtered assembly gaps: " + std::to_string(c.assembly_gap_filtered));
    if (counts_out) *counts_out = c;
    return true;
}

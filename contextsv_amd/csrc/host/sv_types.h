// sv_types.h — output surface kept from the reference (include/sv_types.h:16-155): same enum names and
// numeric values, same strings, same flag bit positions (they are observable through the VCF ALN= field),
// same throwing behaviour (std::out_of_range for values outside the tables). Implemented with switches
// instead of static hash maps.
#pragma once
#include <bitset>
#include <stdexcept>
#include <string>

namespace sv_types {

enum class SVType { UNKNOWN = -1, DEL = 0, DUP = 1, INV = 2, INS = 3, BND = 4, NEUTRAL = 5, LOH = 6 };
enum class Genotype { HOMOZYGOUS_REF = 0, HETEROZYGOUS = 1, HOMOZYGOUS_ALT = 2, UNKNOWN = 3 };
enum class SVDataType { CIGARINS = 0, CIGARDEL = 1, CIGARCLIP = 2, SPLIT = 3, SPLITDIST1 = 4, SPLITDIST2 = 5,
                        SPLITINV = 6, SUPPINV = 7, HMM = 8, UNKNOWN = 9 };
using SVEvidenceFlags = std::bitset<10>;

inline std::string getSVTypeString(SVType t)
{
    switch (t) {
        case SVType::UNKNOWN: return "UNKNOWN"; case SVType::DEL: return "DEL"; case SVType::DUP: return "DUP";
        case SVType::INV: return "INV"; case SVType::INS: return "INS"; case SVType::BND: return "BND";
        case SVType::NEUTRAL: return "NEUTRAL"; case SVType::LOH: return "LOH";
    }
    throw std::out_of_range("SVTypeString");
}

inline std::string getSVTypeSymbol(SVType t)
{
    switch (t) {
        case SVType::UNKNOWN: return "."; case SVType::DEL: return "<DEL>"; case SVType::DUP: return "<DUP>";
        case SVType::INV: return "<INV>"; case SVType::INS: return "<INS>"; case SVType::BND: return "<BND>";
        default: break;                       // NEUTRAL / LOH have no symbol in the reference table (:40-47)
    }
    throw std::out_of_range("SVTypeSymbol");
}

inline std::string getGenotypeString(Genotype g)
{
    switch (g) {
        case Genotype::HOMOZYGOUS_REF: return "0/0"; case Genotype::HETEROZYGOUS: return "0/1";
        case Genotype::HOMOZYGOUS_ALT: return "1/1"; case Genotype::UNKNOWN: return "./.";
    }
    throw std::out_of_range("GenotypeString");
}

inline const char *svDataTypeName(int bit)
{
    static const char *names[10] = {"CIGARINS", "CIGARDEL", "CIGARCLIP", "SPLIT", "SPLITDIST1", "SPLITDIST2",
                                    "SPLITINV", "SUPPINV", "HMM", "UNKNOWN"};
    if (bit < 0 || bit > 9) throw std::out_of_range("SVDataTypeString");
    return names[bit];
}

// comma-joined names of the set bits, ascending bit order (:112-123)
inline std::string getSVAlignmentTypeString(SVEvidenceFlags f)
{
    std::string out;
    for (int i = 0; i < 10; i++)
        if (f.test((size_t)i)) { if (!out.empty()) out += ","; out += svDataTypeName(i); }
    return out;
}

// copy-number state 0..6 -> SV type (:96-104)
inline SVType getSVTypeFromCNState(int cn_state)
{
    switch (cn_state) {
        case 0: return SVType::UNKNOWN; case 1: case 2: return SVType::DEL; case 3: return SVType::NEUTRAL;
        case 4: return SVType::LOH; case 5: case 6: return SVType::DUP;
    }
    throw std::out_of_range("CNVTypeMap");
}

// (:146-155)
inline bool isValidCopyNumberUpdate(SVType sv_type, SVType updated)
{
    if (updated == SVType::UNKNOWN) return false;
    if (sv_type == SVType::DEL && updated != SVType::DEL) return false;
    if (sv_type == SVType::INS && updated != SVType::DUP) return false;
    return true;
}

}  // namespace sv_types

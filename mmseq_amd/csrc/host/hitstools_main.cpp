// hitstools -- inspect / convert hits files (equivalent of the reference's src/hitstools.cpp:42-159,
// built purely on hitsio).  Usage: hitstools inspect|header|t|b hits_file      (output on stdout)
#include <cstdlib>
#include <iostream>
#include <vector>

#include "hitsio.hpp"

static void usage()
{
    std::cerr << "Usage: hitstools COMMAND hits_file\n\n"
              << "Commands:\n"
              << "  inspect   print the hits file in text format\n"
              << "  header    print only the header in text format\n"
              << "  t         convert to text format\n"
              << "  b         convert to binary format\n";
}

int main(int argc, char **argv)
{
    if (argc != 3) { usage(); return 1; }
    const std::string cmd = argv[1];
    if (cmd == "hitsets") {
        // (additive, for tests: the records as the mmseq CLI reads them -- HitsfileReader::readReadMapRecordsBulk, no read names -- one
        // line per read: its transcripts' header indices)
        HitsfileReader reader(argv[2]);
        std::vector<std::string> names;
        std::map<std::string, double> efflen;
        std::map<std::string, int> truelen;
        std::map<std::string, std::vector<std::string>> genes;
        std::vector<std::vector<std::string>> identical;
        reader.readHeader(&names, &efflen, &truelen, &genes, &identical);
        std::vector<uint32_t> len, idx;
        for (bool more = true; more;) {
            len.clear(); idx.clear();
            more = reader.readReadMapRecordsBulk(len, idx, 1000);
            size_t at = 0;
            for (uint32_t l : len) {
                for (uint32_t j = 0; j < l; ++j) std::cout << (j ? " " : "") << idx[at + j];
                std::cout << "\n";
                at += l;
            }
        }
        return 0;
    }
    if (cmd != "inspect" && cmd != "header" && cmd != "t" && cmd != "b") { usage(); return 1; }
    HitsfileReader reader(argv[2]);
    HitsfileWriter writer(cmd == "b" ? "b" : "t");
    std::vector<std::string> names;
    std::map<std::string, double> efflen;
    std::map<std::string, int> truelen;
    std::map<std::string, std::vector<std::string>> genes;
    std::vector<std::vector<std::string>> identical;
    reader.readHeader(&names, &efflen, &truelen, &genes, &identical);
    for (auto &n : names) writer.addTranscriptMetaData(n, efflen[n], truelen[n]);
    for (auto &g : genes) {
        writer.addGeneIsoformRecord(g.first);
        for (auto &t : g.second) writer.addTranscriptToGeneIsoformRecord(t);
    }
    for (auto &set : identical) {
        writer.addIdenticalTranscriptsRecord();
        for (auto &t : set) writer.addTranscriptToIdenticalTranscriptsRecord(t);
    }
    writer.writeHeader();
    if (cmd != "header") {
        std::string id, tid;
        if (reader.schema() == 1) {
            // a binary file stores header indices, and the writer's index is the order of the addTranscriptMetaData calls above --
            // the header's: the records pass through without a name lookup per hit (10^9 of them in a 50 M-read file)
            std::vector<uint32_t> idx;
            while (reader.readReadMapRecordReadID(id)) {
                writer.addReadMapRecord(id);
                idx.clear();
                reader.readReadMapRecordTranscriptIndices(idx);
                for (uint32_t i : idx) writer.addTranscriptIndexToReadMapRecord(i);
                writer.writeReadMapRecord();
            }
        } else {
            while (reader.readReadMapRecordReadID(id)) {
                writer.addReadMapRecord(id);
                while (reader.readReadMapRecordTranscriptID(tid)) writer.addTranscriptToReadMapRecord(tid);
                writer.writeReadMapRecord();
            }
        }
    }
    writer.close();
    return 0;
}

// hitsio.hpp -- reading and writing mmseq hits files (text schema 0, zlib-binary schema 1).
//
// Same class interface as the reference's src/hitsio.hpp:33-95 (HitsfileWriter / HitsfileReader,
// same method names and argument meaning, errors print to stderr and exit(1) like the reference),
// implemented from the file-format description on plain zlib instead of Boost.Iostreams:
//   text   : src/hitsio.cpp:162-187 (writer), :286-347 (reader)
//   binary : one zlib stream; "MMSEQ_HITSFILE\n", u32 schema, header tables, then records of
//            delta-encoded read name + u32 count + u32 transcript indices
//            (src/hitsio.cpp:189-213, :232-240, :349-398, :413-439; delta coding :77-115)
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <unordered_map>
#include <memory>
#include <string>
#include <vector>
#include <cstdlib>
#include <iostream>

// Fatal input errors leave through here: the readers run on pipeline threads of the CLI, where std::exit would run static
// destructors and stream flushes under threads that are still working.  Flush what was said, leave without unwinding.
[[noreturn]] inline void hits_die() { std::cerr.flush(); std::cout.flush(); std::_Exit(1); }

#define MMSEQ_HEADER "MMSEQ_HITSFILE"

namespace hitsio_detail {
class ByteSource;
class ByteSink;
}

class HitsfileWriter {
public:
    // 't...' = text (schema 0), anything else = binary (schema 1), like src/hitsio.cpp:117-129.
    // The reference always writes to stdout; `out` (default stdout) is an additive extension.
    explicit HitsfileWriter(std::string argHitsfileFormat, FILE *out = stdout);
    ~HitsfileWriter();
    void addTranscriptMetaData(std::string transcriptName, double effectiveLength, int trueLength);
    void addGeneIsoformRecord(std::string geneName);
    void addTranscriptToGeneIsoformRecord(std::string transcriptName);
    void addIdenticalTranscriptsRecord();
    void addTranscriptToIdenticalTranscriptsRecord(std::string transcriptName);
    void writeHeader();
    void addReadMapRecord(std::string readName);
    void addTranscriptToReadMapRecord(std::string transcriptName);
    // Additive fast path (mirror of HitsfileReader::readReadMapRecordTranscriptIndex): the transcript by its index in the order of
    // the addTranscriptMetaData calls -- what the binary schema stores (src/hitsio.cpp:240).  Not to be mixed with names in one record.
    void addTranscriptIndexToReadMapRecord(uint32_t transcriptIndex);
    void writeReadMapRecord();
    void close(); // flushes the compressor (the reference relies on destructor order)

private:
    void writeHeaderSchema0();
    void writeHeaderSchema1();
    void writeReadMapRecordSchema0();
    void writeReadMapRecordSchema1();
    std::unique_ptr<hitsio_detail::ByteSink> sink;
    int hitsfileSchema;
    std::vector<std::string> transcriptName;
    std::map<std::string, double> transcriptEffectiveLength;
    std::map<std::string, int> transcriptTrueLength;
    std::map<std::string, uint32_t> transcriptToIndex;
    std::map<std::string, std::vector<std::string>> geneIsoforms;
    std::string currentGeneName;
    std::vector<std::vector<std::string>> identicalTranscripts;
    std::string currentReadName;
    std::vector<std::string> currentReadTranscripts;
    std::vector<uint32_t> currentReadIndices;
    std::string deltaBuffer;
};

class HitsfileReader {
public:
    explicit HitsfileReader(std::string fileName);
    ~HitsfileReader();
    void readHeader(std::vector<std::string> *transcriptName, std::map<std::string, double> *transcriptEffectiveLength,
                    std::map<std::string, int> *transcriptTrueLength,
                    std::map<std::string, std::vector<std::string>> *geneIsoforms,
                    std::vector<std::vector<std::string>> *identicalTranscripts);
    bool readReadMapRecordReadID(std::string &readID);
    bool readReadMapRecordTranscriptID(std::string &transcriptID);
    // Additive fast path: the transcript's index in header order instead of its name
    // (the binary schema stores exactly this, src/hitsio.cpp:435-436).  -1 when the record is done.
    bool readReadMapRecordTranscriptIndex(uint32_t &index);
    // Additive bulk forms for consumers that need neither read names nor one call per hit (a 50 M-read file has 10^9 of those):
    // the next record's header without materialising its name -- not to be mixed with readReadMapRecordReadID on one reader,
    // the delta coding of the names (src/hitsio.cpp:102-115) is not tracked -- and all remaining transcript indices of the record,
    // appended to `out`.
    bool skipReadMapRecordReadID();
    bool readReadMapRecordTranscriptIndices(std::vector<uint32_t> &out);
    // Additive: up to max_records whole records at once -- record r contributes len[r] indices to idx, both appended.  What the two
    // calls above do record by record, but parsed straight from the inflated buffer while a record lies inside it (binary schema).
    // Returns false when the file is exhausted (records read before that are in len / idx).
    bool readReadMapRecordsBulk(std::vector<uint32_t> &len, std::vector<uint32_t> &idx, size_t max_records);
    int schema() const { return hitsfileSchema; }

private:
    void readHeaderSchema0(std::vector<std::string> *, std::map<std::string, double> *, std::map<std::string, int> *,
                           std::map<std::string, std::vector<std::string>> *, std::vector<std::vector<std::string>> *);
    void readHeaderSchema1(std::vector<std::string> *, std::map<std::string, double> *, std::map<std::string, int> *,
                           std::map<std::string, std::vector<std::string>> *, std::vector<std::vector<std::string>> *);
    std::unique_ptr<hitsio_detail::ByteSource> src;
    std::vector<std::string> headerTranscriptName;
    std::unordered_map<std::string, uint32_t> headerIndex; // text schema: name -> header index
    int hitsfileSchema;
    uint32_t countReadMapRecord;
    std::string deltaBuffer;
};

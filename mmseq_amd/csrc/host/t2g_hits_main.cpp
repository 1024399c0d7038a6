// t2g_hits -- transcript-level hits file -> gene-level read records (equivalent of the reference's
// src/t2g_hits.cpp:33-121): every read's transcripts are replaced by their genes (from @GeneIsoforms),
// de-duplicated and sorted; only the read records are written, to stdout, in text form.
// Built on hitsio, so it also accepts the binary schema (the reference reads text only).
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "hitsio.hpp"

using namespace std;

int main(int argc, char **argv)
{
    if (argc != 2) {
        cerr << "Usage: t2g_hits hits_file > gene_hits_file" << endl
             << endl
             << "Mandatory arguments:" << endl
             << "  hits_file          hits file generated with `bam2hits`\n"
             << endl;
        return 1;
    }
    HitsfileReader reader(argv[1]);
    vector<string> names;
    map<string, double> efflen;
    map<string, int> truelen;
    map<string, vector<string>> genes;
    vector<vector<string>> identical;
    reader.readHeader(&names, &efflen, &truelen, &genes, &identical);
    map<string, string> t2g; // src/t2g_hits.cpp:88-91
    for (auto &g : genes)
        for (auto &t : g.second) t2g[t] = g.first;
    string id, tid, out;
    vector<string> comb;
    if (reader.schema() == 1) {
        // a binary file stores header indices: gene of a transcript = a table lookup, the genes of a read sorted by their rank in
        // name order (byte-wise, what std::sort does with the strings below; "" -- a transcript without a gene -- sorts first)
        vector<string> gname{""};
        for (auto &g : genes) gname.push_back(g.first);           // std::map: already in name order
        map<string, uint32_t> rank_of;
        for (uint32_t r = 0; r < gname.size(); ++r) rank_of[gname[r]] = r;
        vector<uint32_t> rank(names.size());
        for (size_t t = 0; t < names.size(); ++t) {
            auto it = t2g.find(names[t]);
            rank[t] = it == t2g.end() ? 0u : rank_of[it->second];
        }
        vector<uint32_t> idx, rs;
        out.reserve(1 << 20);
        while (reader.readReadMapRecordReadID(id)) {
            out += ">"; out += id; out += "\n";
            idx.clear(); rs.clear();
            reader.readReadMapRecordTranscriptIndices(idx);
            for (uint32_t i : idx) {
                if (i >= rank.size()) { cout.write(out.data(), (streamsize)out.size()); cerr << "Hits file looks malformed.\n"; hits_die(); } // (what the name API says to an index outside the header)
                rs.push_back(rank[i]);
            }
            sort(rs.begin(), rs.end());
            rs.erase(unique(rs.begin(), rs.end()), rs.end());
            for (uint32_t r : rs) { out += gname[r]; out += "\n"; }
            if (out.size() > (1 << 20) - 4096) { cout.write(out.data(), (streamsize)out.size()); out.clear(); }
        }
        cout.write(out.data(), (streamsize)out.size());
        return 0;
    }
    while (reader.readReadMapRecordReadID(id)) {
        out = ">" + id + "\n";
        comb.clear();
        while (reader.readReadMapRecordTranscriptID(tid)) {
            const string &g = t2g[tid]; // an unknown transcript maps to the empty gene id, as in the reference (:109)
            if (find(comb.begin(), comb.end(), g) == comb.end()) comb.push_back(g);
        }
        sort(comb.begin(), comb.end());
        for (auto &g : comb) { out += g; out += "\n"; }
        cout << out;
    }
    return 0;
}

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

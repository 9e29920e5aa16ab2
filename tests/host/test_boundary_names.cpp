// test_boundary_names.cpp -- TEST INFRASTRUCTURE.  A caller written only with the names the reference's own sources use
// at the drop-in boundary (src/DSK.hpp:39-54, src/DSK.cpp:32-34,51-68,80-86,100-103,
// src/main.cpp:34-46, utils/dsk2ascii.cpp:16-22,31-41,58-77,87-104,114-116) must compile against this repo's host layer
// and behave: it writes a small storage through Storage/Group/Partition, reads it back the way dsk2ascii does
// (getGroup("dsk").getPartition<Count>("solid"), tool.createIterator(..), LOCAL, model.toString) and dispatches on the
// k-mer size with Integer::apply<Functor,Parameter>.  Own code, not a copy of either reference file.
#include <cstdio>
#include <sstream>

#include "../../dsk_amd/host/dsk.hpp"

using namespace dsk;

namespace {

struct Probe : public Tool {
    Probe() : Tool("probe") {
        getParser()->push_front(new OptionOneParam(STR_URI_FILE, "storage to write and read back", true));
        getParser()->push_back(new OptionOneParam(STR_KMER_SIZE, "size of a kmer", false, "31"));
        getParser()->push_back(new OptionNoParam("-quiet", "print nothing", false));
        IOptionsParser* renamed = getParser()->getParser(STR_KMER_SIZE);
        if (renamed == nullptr) throw Exception("getParser(name) lost an option");
    }
    std::ostringstream lines;

    struct Parameter {
        Parameter(Probe& tool, Storage* storage, size_t kmerSize) : tool(tool), storage(storage), kmerSize(kmerSize) {}
        Probe& tool; Storage* storage; size_t kmerSize;
    };

    template <size_t span> struct Functor {
        void operator()(Parameter parameter) {
            typedef typename Kmer<span>::Count Count;
            typedef typename Kmer<span>::Type Type;
            Probe& tool = parameter.tool;
            Storage* storage = parameter.storage;
            typename Kmer<span>::ModelCanonical model(parameter.kmerSize);
            // write three rows into two partitions
            Partition<Count>& out = storage->getGroup("dsk").template getPartition<Count>("solid", 2);
            std::string a(parameter.kmerSize, 'A'), c(parameter.kmerSize, 'C'), t(parameter.kmerSize, 'T');
            c[parameter.kmerSize - 1] = 'A';
            Count rows0[2] = {Count(model.canonical(model.codeSeed(a.c_str())), 7), Count(model.canonical(model.codeSeed(c.c_str())), 3)};
            Count rows1[1] = {Count(model.canonical(model.codeSeed(t.c_str())), 2)};
            out.insert(0, rows0, 2);
            out.insert(1, rows1, 1);
            storage->getGroup("dsk").setProperty("kmer_size", std::to_string(parameter.kmerSize));
            // read back exactly as utils/dsk2ascii.cpp:61-104 does
            Partition<Count>& solidKmers = storage->getGroup("dsk").template getPartition<Count>("solid");
            Iterator<Count>* itKmers = tool.createIterator(solidKmers.iterator(), solidKmers.getNbItems(), "parsing");
            LOCAL(itKmers);
            for (itKmers->first(); !itKmers->isDone(); itKmers->next()) {
                const Count& count = itKmers->item();
                char buf[256];
                snprintf(buf, sizeof(buf), "%s %i\n", model.toString(count.value).c_str(), count.abundance);
                tool.lines << buf;
            }
            Type zero; (void)zero;
        }
    };

    void execute() override {
        Storage* storage = StorageFactory(STORAGE_HDF5).create(getInput()->getStr(STR_URI_FILE), true, false);
        LOCAL(storage);
        const size_t kmerSize = (size_t)getInput()->getInt(STR_KMER_SIZE);
        Integer::apply<Functor, Parameter>(kmerSize, Parameter(*this, storage, kmerSize));
        if (storage->getGroup("dsk").getProperty("kmer_size") != std::to_string(kmerSize)) throw Exception("kmer_size attribute lost");
        getInfo()->add(1, "rows", "%d", 3);
        (void)getInfo()->getXML();
    }
};

int fail(const char* what) { std::printf("FAILED: %s\n", what); return 1; }

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 2) return fail("usage: test_boundary_names <tmp storage name>");
    for (const char* k : {"5", "31", "40", "70", "100"}) {
        Probe tool;
        char* args[] = {argv[0], (char*)"-file", argv[1], (char*)"-kmer-size", (char*)k, (char*)"-verbose", (char*)"0", (char*)"-quiet"};
        try {
            tool.run(8, args);
            if (!tool.getParser()->saw("-quiet")) return fail("saw()");
        } catch (OptionFailure& e) { return e.displayErrors(std::cout); }
        catch (Exception& e) { std::cout << "EXCEPTION: " << e.getMessage() << std::endl; return EXIT_FAILURE; }
        const size_t n = (size_t)atoi(k);
        // A..A is its own minimum; C..CA vs its reverse complement TG..G: C < T so the forward form stays; T..T -> A..A (complement)
        std::string want = std::string(n, 'A') + " 7\n" + std::string(n - 1, 'C') + "A 3\n" + std::string(n, 'A') + " 2\n";
        if (tool.lines.str() != want) { std::printf("k=%s got:\n%s", k, tool.lines.str().c_str()); return fail("rows read back"); }
    }
    // errors keep the reference's shape: a missing mandatory option is an OptionFailure with a usage text
    try { Probe t; char* a[] = {argv[0]}; t.run(1, a); return fail("missing -file accepted"); }
    catch (OptionFailure& e) { std::ostringstream os; if (e.displayErrors(os) == 0) return fail("displayErrors code"); }
    // 128 and up: no compiled span (KSIZE_LIST 32 64 96 128)
    try { Probe t; char* a[] = {argv[0], (char*)"-file", argv[1], (char*)"-kmer-size", (char*)"128", (char*)"-verbose", (char*)"0"}; t.run(7, a); return fail("k=128 accepted"); }
    catch (std::exception&) {}
    std::printf("ALL OK\n");
    return 0;
}

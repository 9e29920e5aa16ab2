// dsk.cpp -- the `dsk` tool: option wiring + span dispatch around the counting engine.
// Plays the role of src/DSK.cpp in the reference (parser splice and -in -> -file rename at :80-87,
// k-span dispatch at :97-104, run + statistics + "xml" property at :45-70) on this repo's host layer.
#include "dsk.hpp"

namespace dsk {

DSK::DSK() : Tool("dsk") {
    // every option of the counting algorithm becomes an option of the tool ...
    OptionsParser* mine = getParser();
    mine->push_back(SortingCountAlgorithm<>::getOptionsParser(), 1);
    // ... except that the reads are given with -file instead of -in
    IOptionsParser* reads = mine->getParser(STR_URI_INPUT);
    if (reads != nullptr) reads->setName(STR_URI_FILE);
}

// Count with the k-mer integer width `span` bits / 2 bases that fits the requested k.
template <size_t span>
static void countWithSpan(DSK& tool, IProperties& options) {
    std::unique_ptr<IBank> reads(Bank::open(options.getStr(STR_URI_FILE)));
    SortingCountAlgorithm<span> counter(reads.get(), &options);
    counter.getInput()->set(STR_VERBOSE, options.getStr(STR_VERBOSE));

    counter.execute();      // reads -> HBM -> (kmer, abundance) partitions + histogram -> HDF5

    // what -verbose 1 prints at exit, and what is kept inside the output file
    IProperties* report = tool.getInfo();
    report->add(1, counter.getConfig().getProperties());
    report->add(1, counter.getInfo());
    Group& home = counter.getStorage()->getGroup(counter.getName());
    home.setProperty("xml", "\n" + counter.getInfo()->getXML());
}

void DSK::execute() {
    IProperties& options = *getInput();
    const size_t k = (size_t)options.getInt(STR_KMER_SIZE);
    try {
        Integer::dispatch(k, [&](auto width) { countWithSpan<decltype(width)::value>(*this, options); });
    } catch (std::runtime_error& problem) {
        throw Exception(std::string(problem.what()));
    }
}

}  // namespace dsk

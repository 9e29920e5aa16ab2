// dsk.cpp -- see dsk.hpp.  Same control flow as src/DSK.cpp:45-104, written
// against this repo's host layer.
#include "dsk.hpp"

namespace dsk {

namespace {
struct Parameter {
    Parameter(DSK& d, IProperties* p) : dsk(d), props(p) {}
    DSK& dsk; IProperties* props;
};

template <size_t span>
struct Functor {
    void operator()(Parameter parameter) {
        DSK& tool = parameter.dsk;
        IProperties* props = parameter.props;

        IBank* bank = Bank::open(props->getStr(STR_URI_FILE));          // src/DSK.cpp:51
        LOCAL(bank);

        SortingCountAlgorithm<span> sortingCount(bank, props);           // src/DSK.cpp:55
        sortingCount.getInput()->set(STR_VERBOSE, props->getStr(STR_VERBOSE));
        sortingCount.execute();                                          // src/DSK.cpp:60  (GPU engine)

        tool.getInfo()->add(1, sortingCount.getConfig().getProperties());  // src/DSK.cpp:63-64
        tool.getInfo()->add(1, sortingCount.getInfo());

        // src/DSK.cpp:68: run info stored as the "xml" attribute of group "dsk"
        sortingCount.getStorage()->getGroup(sortingCount.getName()).setProperty("xml", std::string("\n") + sortingCount.getInfo()->getXML());
    }
};
}  // namespace

DSK::DSK() : Tool("dsk") {
    getParser()->push_back(SortingCountAlgorithm<>::getOptionsParser(), 1);          // src/DSK.cpp:83
    if (IOptionsParser* input = getParser()->getParser(STR_URI_INPUT)) input->setName(STR_URI_FILE);   // src/DSK.cpp:86
}

void DSK::execute() {
    size_t kmerSize = (size_t)getInput()->getInt(STR_KMER_SIZE);                     // src/DSK.cpp:100
    try { Integer::apply<Functor, Parameter>(kmerSize, Parameter(*this, getInput())); }   // src/DSK.cpp:103
    catch (std::runtime_error& e) { throw Exception(std::string(e.what())); }
}

}  // namespace dsk

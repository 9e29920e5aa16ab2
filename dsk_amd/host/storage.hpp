// storage.hpp -- HDF5 storage with the layout DSK writes and dsk2ascii reads:
//   /dsk                      group, string attributes "kmer_size", "xml"      (src/DSK.cpp:68; utils/dsk2ascii.cpp:34)
//   /dsk/solid                group, attribute "nb_partitions", datasets "0".."P-1" of
//                             compound { value, abundance }                     (utils/dsk2ascii.cpp:61,77,104; README.md:70-75)
//   /histogram/histogram      compound { index:u16, abundance:u64 }, rows 1..histo_max
//                                                                               (README.md:73,78; scripts/simple_test.sh:37; test/k27.histo)
// Names Storage / Group / Partition / StorageFactory / Iterator follow the calls
// in the reference sources; the implementation is plain HDF5 C API (1.10).
#pragma once
#include <hdf5.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kmer.hpp"
#include "tool.hpp"

namespace dsk {

enum StorageMode_e { STORAGE_HDF5 = 0 };

struct HistoEntry { uint16_t index; uint64_t abundance; };

template <size_t span> hid_t count_type();
// HDF5 memory type of a row type (caller H5Tclose's it): Kmer<span>::Count by default, HistoEntry below
template <class T> struct H5Row { static hid_t make() { return count_type<T::SPAN>(); } };
template <> struct H5Row<HistoEntry> {
    static hid_t make() {
        hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(HistoEntry));
        H5Tinsert(t, "index", HOFFSET(HistoEntry, index), H5T_NATIVE_UINT16);
        H5Tinsert(t, "abundance", HOFFSET(HistoEntry, abundance), H5T_NATIVE_UINT64);
        return t;
    }
};
template <size_t span> hid_t count_type() {
    typedef typename Kmer<span>::Count Count;
    hid_t t = H5Tcreate(H5T_COMPOUND, sizeof(Count));
    if (Kmer<span>::WORDS == 1) H5Tinsert(t, "value", HOFFSET(Count, value), H5T_NATIVE_UINT64);
    else {
        hsize_t dims[1] = {Kmer<span>::WORDS};
        hid_t arr = H5Tarray_create2(H5T_NATIVE_UINT64, 1, dims);   // words, least significant first
        H5Tinsert(t, "value", HOFFSET(Count, value), arr);
        H5Tclose(arr);
    }
    H5Tinsert(t, "abundance", HOFFSET(Count, abundance), H5T_NATIVE_INT32);
    return t;
}

template <class T>
class Iterator {
public:
    virtual ~Iterator() {}
    virtual void first() = 0;
    virtual void next() = 0;
    virtual bool isDone() = 0;
    virtual const T& item() = 0;
};

class Storage;
template <class T> class Partition;

class Group {
public:
    Group(Storage* st, const std::string& path) : st_(st), path_(path) {}
    void setProperty(const std::string& key, const std::string& value);
    std::string getProperty(const std::string& key);         // "" when absent
    Group& getGroup(const std::string& name);
    const std::string& path() const { return path_; }
    Storage* storage() const { return st_; }
    // raw dataset helpers (1-D arrays of a compound/atomic memory type)
    void writeDataset(const std::string& name, hid_t memtype, const void* rows, uint64_t n, int compress = 0);
    // A dataset of n rows with its space in the file allocated at once (contiguous, never filled) -> the file offset of row 0:
    // the caller writes the rows itself (Storage::rawWrite, from several threads), which is what a big partition needs --
    // H5Dwrite copies the rows into the page cache from one thread.  -1 when the library gives no offset.
    long long reserveDataset(const std::string& name, hid_t memtype, uint64_t n);
    uint64_t datasetSize(const std::string& name);
    void readDataset(const std::string& name, hid_t memtype, void* rows, uint64_t offset, uint64_t n);
    bool exists(const std::string& name);
    // The collection of datasets "0".."P-1" under <this group>/<name> with rows of type T, as read by
    // `storage->getGroup("dsk").getPartition<Count>("solid")` (utils/dsk2ascii.cpp:61).  nbPartitions = 0: an existing
    // collection (its "nb_partitions" attribute says how many); > 0: a collection being written.  Owned by the group.
    template <class T> Partition<T>& getPartition(const std::string& name, size_t nbPartitions = 0);
private:
    Storage* st_; std::string path_;
    std::map<std::string, std::unique_ptr<Group>> subs_;
    std::map<std::string, std::shared_ptr<void>> parts_;
};

// Partition<T>: the datasets "0".."P-1" under <group>/<name>, iterated in index order.
template <class T>
class Partition {
public:
    Partition(Group& parent, const std::string& name, hid_t memtype, size_t nbPartitions = 0)
        : grp_(parent.getGroup(name)), type_(memtype), nb_(nbPartitions) {
        if (nb_ == 0) { std::string s = grp_.getProperty("nb_partitions"); nb_ = s.empty() ? 0 : (size_t)atoll(s.c_str()); }
        else grp_.setProperty("nb_partitions", std::to_string(nb_));
    }
    ~Partition() { H5Tclose(type_); }
    size_t size() const { return nb_; }
    void insert(size_t p, const T* rows, uint64_t n, int compress = 0) { grp_.writeDataset(std::to_string(p), type_, rows, n, compress); }
    // partition p as n rows the caller writes itself with Storage::rawWrite at the returned file offset (-1: use insert)
    long long reserve(size_t p, uint64_t n) { return grp_.reserveDataset(std::to_string(p), type_, n); }
    Storage* storage() const { return grp_.storage(); }
    uint64_t partitionSize(size_t p) { return grp_.datasetSize(std::to_string(p)); }
    uint64_t getNbItems() { uint64_t t = 0; for (size_t p = 0; p < nb_; ++p) t += partitionSize(p); return t; }
    void read(size_t p, std::vector<T>& out) {
        uint64_t n = partitionSize(p); out.resize(n);
        if (n) grp_.readDataset(std::to_string(p), type_, out.data(), 0, n);
    }
    // all partitions in index order, rows in stored order (utils/dsk2ascii.cpp:77,85)
    Iterator<T>* iterator() { return new It(*this); }
private:
    class It : public Iterator<T> {
    public:
        explicit It(Partition& p) : part_(p), p_(0), i_(0) {}
        void first() override { p_ = 0; i_ = 0; buf_.clear(); load(); }
        void next() override { ++i_; if (i_ >= buf_.size()) { ++p_; load(); } }
        bool isDone() override { return p_ >= part_.size(); }
        const T& item() override { return buf_[i_]; }
    private:
        void load() { i_ = 0; buf_.clear(); while (p_ < part_.size()) { part_.read(p_, buf_); if (!buf_.empty()) break; ++p_; } }
        Partition& part_; size_t p_, i_; std::vector<T> buf_;
    };
    Group& grp_; hid_t type_; size_t nb_;
};

template <class T>
Partition<T>& Group::getPartition(const std::string& name, size_t nbPartitions) {
    std::shared_ptr<void>& slot = parts_[name];
    if (!slot || nbPartitions) slot = std::shared_ptr<void>(new Partition<T>(*this, name, H5Row<T>::make(), nbPartitions),
                                                            [](void* p) { delete static_cast<Partition<T>*>(p); });
    return *static_cast<Partition<T>*>(slot.get());
}

class Storage {
public:
    ~Storage();
    Group& getGroup(const std::string& name);     // created on demand when the file is writable
    Group& root() { return *root_; }
    hid_t file() const { return fid_; }
    bool writable() const { return writable_; }
    const std::string& filename() const { return filename_; }
    // bytes of a reserved dataset, straight into the file (thread-safe: positional writes on a descriptor of its own)
    void rawWrite(long long offset, const void* data, size_t bytes);
    template <size_t span>
    Partition<typename Kmer<span>::Count>* solidPartition(size_t nbPartitions = 0) {
        return new Partition<typename Kmer<span>::Count>(getGroup("dsk"), "solid", count_type<span>(), nbPartitions);
    }
private:
    friend class StorageFactory;
    Storage() : fid_(-1), writable_(false) {}
    hid_t fid_; bool writable_; std::string filename_;
    int raw_fd_ = -1; std::mutex raw_mu_;
    std::unique_ptr<Group> root_;
};

class StorageFactory {
public:
    explicit StorageFactory(StorageMode_e mode) : mode_(mode) {}
    // create <uri>[.h5] (truncate); load accepts the name with or without ".h5"
    // (scripts/simple_test.sh:89 passes "-file test_short").
    Storage* create(const std::string& uri, bool deleteIfExist = true, bool autoRemove = false);
    Storage* load(const std::string& uri);
    static std::string h5name(const std::string& uri);
private:
    StorageMode_e mode_;
};

}  // namespace dsk

// storage.cpp -- see storage.hpp.
#include "storage.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

namespace dsk {

namespace {
struct SilenceH5 { SilenceH5() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); } };
void silence() { static SilenceH5 s; }
bool link_exists(hid_t loc, const std::string& path) {
    // every intermediate component must exist for H5Lexists
    size_t pos = 0; std::string cur;
    while (pos < path.size()) {
        size_t s = path.find('/', pos);
        if (s == std::string::npos) s = path.size();
        if (s > pos) {
            cur += (cur.empty() ? "" : "/") + path.substr(pos, s - pos);
            if (H5Lexists(loc, cur.c_str(), H5P_DEFAULT) <= 0) return false;
        }
        pos = s + 1;
    }
    return true;
}
}  // namespace

std::string StorageFactory::h5name(const std::string& uri) {
    if (uri.size() >= 3 && uri.compare(uri.size() - 3, 3, ".h5") == 0) return uri;
    return uri + ".h5";
}

Storage* StorageFactory::create(const std::string& uri, bool /*deleteIfExist*/, bool /*autoRemove*/) {
    silence();
    std::unique_ptr<Storage> st(new Storage());
    st->filename_ = h5name(uri);
    st->fid_ = H5Fcreate(st->filename_.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (st->fid_ < 0) throw Exception("unable to create storage '%s'", st->filename_.c_str());
    st->writable_ = true;
    st->root_.reset(new Group(st.get(), ""));
    return st.release();
}

Storage* StorageFactory::load(const std::string& uri) {
    silence();
    struct stat sb;
    std::string name = uri;
    if (stat(name.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) name = h5name(uri);
    if (stat(name.c_str(), &sb) != 0) throw Exception("unable to open storage '%s'", uri.c_str());
    std::unique_ptr<Storage> st(new Storage());
    st->filename_ = name;
    st->fid_ = H5Fopen(name.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (st->fid_ < 0) throw Exception("unable to open storage '%s' (not an HDF5 file?)", name.c_str());
    st->writable_ = false;
    st->root_.reset(new Group(st.get(), ""));
    return st.release();
}

Storage::~Storage() { root_.reset(); if (raw_fd_ >= 0) ::close(raw_fd_); if (fid_ >= 0) H5Fclose(fid_); }

Group& Storage::getGroup(const std::string& name) { return root_->getGroup(name); }

Group& Group::getGroup(const std::string& name) {
    auto it = subs_.find(name);
    if (it != subs_.end()) return *it->second;
    std::string p = path_.empty() ? name : path_ + "/" + name;
    if (!link_exists(st_->file(), p)) {
        if (!st_->writable()) throw Exception("group '%s' not found in '%s'", p.c_str(), st_->filename().c_str());
        hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
        H5Pset_create_intermediate_group(lcpl, 1);
        hid_t g = H5Gcreate2(st_->file(), p.c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT);
        H5Pclose(lcpl);
        if (g < 0) throw Exception("unable to create group '%s'", p.c_str());
        H5Gclose(g);
    }
    subs_[name].reset(new Group(st_, p));
    return *subs_[name];
}

void Group::setProperty(const std::string& key, const std::string& value) {
    hid_t obj = path_.empty() ? H5Gopen2(st_->file(), "/", H5P_DEFAULT) : H5Gopen2(st_->file(), path_.c_str(), H5P_DEFAULT);
    if (obj < 0) throw Exception("unable to open group '%s'", path_.c_str());
    if (H5Aexists(obj, key.c_str()) > 0) H5Adelete(obj, key.c_str());
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, value.size() + 1);
    H5Tset_strpad(t, H5T_STR_NULLTERM);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, key.c_str(), t, sp, H5P_DEFAULT, H5P_DEFAULT);
    if (a < 0) { H5Sclose(sp); H5Tclose(t); H5Gclose(obj); throw Exception("unable to write attribute '%s'", key.c_str()); }
    H5Awrite(a, t, value.c_str());
    H5Aclose(a); H5Sclose(sp); H5Tclose(t); H5Gclose(obj);
}

std::string Group::getProperty(const std::string& key) {
    hid_t obj = path_.empty() ? H5Gopen2(st_->file(), "/", H5P_DEFAULT) : H5Gopen2(st_->file(), path_.c_str(), H5P_DEFAULT);
    if (obj < 0) return "";
    std::string out;
    if (H5Aexists(obj, key.c_str()) > 0) {
        hid_t a = H5Aopen(obj, key.c_str(), H5P_DEFAULT);
        hid_t ft = H5Aget_type(a);
        if (H5Tis_variable_str(ft) > 0) {
            char* s = nullptr;
            hid_t mt = H5Tcopy(H5T_C_S1); H5Tset_size(mt, H5T_VARIABLE);
            if (H5Aread(a, mt, &s) >= 0 && s) { out = s; free(s); }
            H5Tclose(mt);
        } else {
            size_t n = H5Tget_size(ft);
            std::vector<char> buf(n + 1, 0);
            hid_t mt = H5Tcopy(H5T_C_S1); H5Tset_size(mt, n);
            if (H5Aread(a, mt, buf.data()) >= 0) out = std::string(buf.data(), strnlen(buf.data(), n));
            H5Tclose(mt);
        }
        H5Tclose(ft); H5Aclose(a);
    }
    H5Gclose(obj);
    return out;
}

bool Group::exists(const std::string& name) { return link_exists(st_->file(), path_.empty() ? name : path_ + "/" + name); }

void Group::writeDataset(const std::string& name, hid_t memtype, const void* rows, uint64_t n, int compress) {
    std::string p = path_.empty() ? name : path_ + "/" + name;
    if (link_exists(st_->file(), p)) H5Ldelete(st_->file(), p.c_str(), H5P_DEFAULT);
    hsize_t dims[1] = {n}, maxd[1] = {H5S_UNLIMITED};
    hsize_t chunk[1] = {std::max<hsize_t>(1, std::min<hsize_t>(n, 1 << 16))};
    hid_t sp = H5Screate_simple(1, dims, maxd);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 1, chunk);
    if (compress > 0) H5Pset_deflate(pl, (unsigned)std::min(compress, 9));
    hid_t ds = H5Dcreate2(st_->file(), p.c_str(), memtype, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    if (ds < 0) { H5Pclose(pl); H5Sclose(sp); throw Exception("unable to create dataset '%s'", p.c_str()); }
    if (n && H5Dwrite(ds, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, rows) < 0) {
        H5Dclose(ds); H5Pclose(pl); H5Sclose(sp); throw Exception("unable to write dataset '%s'", p.c_str());
    }
    H5Dclose(ds); H5Pclose(pl); H5Sclose(sp);
}

long long Group::reserveDataset(const std::string& name, hid_t memtype, uint64_t n) {
    if (n == 0) return -1;
    std::string p = path_.empty() ? name : path_ + "/" + name;
    if (link_exists(st_->file(), p)) H5Ldelete(st_->file(), p.c_str(), H5P_DEFAULT);
    hsize_t dims[1] = {n};
    hid_t sp = H5Screate_simple(1, dims, nullptr);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_layout(pl, H5D_CONTIGUOUS);
    H5Pset_alloc_time(pl, H5D_ALLOC_TIME_EARLY);       // the rows get their place in the file now ...
    H5Pset_fill_time(pl, H5D_FILL_TIME_NEVER);         // ... and the library writes nothing there
    hid_t ds = H5Dcreate2(st_->file(), p.c_str(), memtype, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    long long off = -1;
    if (ds >= 0) {
        const haddr_t a = H5Dget_offset(ds);
        if (a != HADDR_UNDEF) off = (long long)a;
        H5Dclose(ds);
        if (off < 0) H5Ldelete(st_->file(), p.c_str(), H5P_DEFAULT);
    }
    H5Pclose(pl); H5Sclose(sp);
    return off;
}

void Storage::rawWrite(long long offset, const void* data, size_t bytes) {
    {
        std::lock_guard<std::mutex> g(raw_mu_);
        if (raw_fd_ < 0) {
            raw_fd_ = ::open(filename_.c_str(), O_WRONLY);
            if (raw_fd_ < 0) throw Exception("unable to open '%s' for the partition rows", filename_.c_str());
        }
    }
    const char* p = static_cast<const char*>(data);
    while (bytes) {
        const ssize_t w = ::pwrite(raw_fd_, p, bytes, (off_t)offset);
        if (w <= 0) throw Exception("write error in '%s'", filename_.c_str());
        p += w; offset += w; bytes -= (size_t)w;
    }
}

uint64_t Group::datasetSize(const std::string& name) {
    std::string p = path_.empty() ? name : path_ + "/" + name;
    if (!link_exists(st_->file(), p)) return 0;
    hid_t ds = H5Dopen2(st_->file(), p.c_str(), H5P_DEFAULT);
    if (ds < 0) return 0;
    hid_t sp = H5Dget_space(ds);
    hsize_t dims[1] = {0};
    H5Sget_simple_extent_dims(sp, dims, nullptr);
    H5Sclose(sp); H5Dclose(ds);
    return dims[0];
}

void Group::readDataset(const std::string& name, hid_t memtype, void* rows, uint64_t offset, uint64_t n) {
    std::string p = path_.empty() ? name : path_ + "/" + name;
    hid_t ds = H5Dopen2(st_->file(), p.c_str(), H5P_DEFAULT);
    if (ds < 0) throw Exception("dataset '%s' not found", p.c_str());
    hid_t fs = H5Dget_space(ds);
    hsize_t start[1] = {offset}, cnt[1] = {n};
    H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, nullptr, cnt, nullptr);
    hid_t ms = H5Screate_simple(1, cnt, nullptr);
    herr_t e = H5Dread(ds, memtype, ms, fs, H5P_DEFAULT, rows);
    H5Sclose(ms); H5Sclose(fs); H5Dclose(ds);
    if (e < 0) throw Exception("unable to read dataset '%s'", p.c_str());
}

}  // namespace dsk

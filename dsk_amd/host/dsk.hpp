// dsk.hpp -- the DSK tool wrapper (counterpart of src/DSK.hpp / src/DSK.cpp):
// option wiring, "-in" renamed to "-file", k-span dispatch, run + info + xml.
#pragma once
#include "sorting_count.hpp"

namespace dsk {

class DSK : public Tool {
public:
    DSK();
    static StorageMode_e getStorageMode() { return STORAGE_HDF5; }   // src/DSK.hpp:48
private:
    void execute() override;
};

}  // namespace dsk

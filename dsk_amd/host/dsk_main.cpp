// dsk_main.cpp -- `dsk` executable.  Exit-code / message mapping of src/main.cpp:28-49:
// OptionFailure -> usage on stdout, its code; Exception -> "EXCEPTION: msg" on stderr, EXIT_FAILURE.
#include "dsk.hpp"

int main(int argc, char* argv[]) {
    dsk::setBackendFactory(dsk::createGpuBackend);   // the only backend this binary knows: the HIP engine
    try {
        dsk::DSK().run(argc, argv);
    } catch (dsk::OptionFailure& e) {
        return e.displayErrors(std::cout);
    } catch (dsk::Exception& e) {
        std::cerr << "EXCEPTION: " << e.getMessage() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

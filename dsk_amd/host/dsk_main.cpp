// dsk_main.cpp -- `dsk` executable.  Exit-code / message mapping of src/main.cpp:28-49:
// OptionFailure -> usage on stdout, its code; Exception -> "EXCEPTION: msg" on stderr, EXIT_FAILURE.
#include "dsk.hpp"

#include <sys/time.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
static double wall_s() { struct timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }

int main(int argc, char* argv[]) {
    const double t0 = wall_s();
    dsk::setBackendFactory(dsk::createGpuBackend);   // the only backend this binary knows: the HIP engine
    try {
        dsk::DSK().run(argc, argv);
    } catch (dsk::OptionFailure& e) {
        return e.displayErrors(std::cout);
    } catch (dsk::Exception& e) {
        std::cerr << "EXCEPTION: " << e.getMessage() << std::endl;
        return EXIT_FAILURE;
    }
    // The output file is closed and everything printed: leave without the device runtime's exit handlers (0.1 s of a 0.4 s run
    // on the E. coli-sized input: they unload code objects and unmap the device for a process that is gone a moment later).
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    if (getenv("DSK_PHASE_TIMES")) fprintf(stderr, "[dsk] main() took %.3f s\n", wall_s() - t0);
    _exit(EXIT_SUCCESS);
}

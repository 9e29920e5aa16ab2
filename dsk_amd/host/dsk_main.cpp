// dsk_main.cpp -- `dsk` executable.  Exit-code / message mapping of src/main.cpp:28-49:
// OptionFailure -> usage on stdout, its code; Exception -> "EXCEPTION: msg" on stderr, EXIT_FAILURE.
#include "dsk.hpp"

#include <hdf5.h>
#include <sys/time.h>
#include <cstring>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
static double wall_s() { struct timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }

int main(int argc, char* argv[]) {
    const double t0 = wall_s();
    // One GPU of a node with several: let the device runtime bring up only that one (its start-up is most of a small run and grows
    // with the devices it enumerates).  Decided HERE, before any thread exists (setenv next to the parser threads' getenv was a data
    // race: ADVICE r03), for the command line only -- a host that embeds the backend selects its device with -device / hipSetDevice.
    {
        int device = 0; bool multi = false;
        for (int i = 1; i + 1 < argc; ++i) {
            if (!std::strcmp(argv[i], "-device")) device = atoi(argv[i + 1]);
            if (!std::strcmp(argv[i], "-nb-gpus") && atoi(argv[i + 1]) > 1) multi = true;
        }
        if (!multi && device >= 0 && !getenv("ROCR_VISIBLE_DEVICES") && !getenv("HIP_VISIBLE_DEVICES") && !getenv("CUDA_VISIBLE_DEVICES")) {
            char dev[16]; snprintf(dev, sizeof dev, "%d", device);
            setenv("ROCR_VISIBLE_DEVICES", dev, 1);
            setenv("DSK_DEVICE_REMAPPED", dev, 1);          // the backend maps `-device <dev>` to ordinal 0 of what is visible now
        }
    }
    dsk::setBackendFactory(dsk::createGpuBackend);   // the only backend this binary knows: the HIP engine
    dsk::setProcessExitsAfterRun(true);              // (this process ends with _exit below: no buffer-by-buffer teardown of the engine)
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
    // (HDF5 first: H5close flushes and closes whatever the library still holds -- with the default weak close degree a leaked
    //  handle would otherwise leave the .h5 unflushed behind an exit code of 0)
    H5close();
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    if (getenv("DSK_PHASE_TIMES")) {
        fprintf(stderr, "[dsk] main() took %.3f s (entered at %.6f, leaves at %.6f: against the caller's clock around the process this tells start-up and exit apart)\n", wall_s() - t0, t0, wall_s());
        if (FILE* f = fopen("/proc/self/status", "r")) {       // what the kernel still has to take apart after _exit (it is part of the wall clock a caller sees)
            char line[256];
            while (fgets(line, sizeof line, f))
                if (!strncmp(line, "VmRSS", 5) || !strncmp(line, "RssAnon", 7) || !strncmp(line, "RssFile", 7) || !strncmp(line, "RssShmem", 8) || !strncmp(line, "VmPTE", 5) || !strncmp(line, "Threads", 7))
                    fprintf(stderr, "[dsk] at exit: %s", line);
            fclose(f);
        }
    }
    _exit(EXIT_SUCCESS);
}

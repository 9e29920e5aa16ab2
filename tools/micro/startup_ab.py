import subprocess, time, os
for i in range(4):
    for d in ("1", "0"):
        env = dict(os.environ, HIP_ENABLE_DEFERRED_LOADING=d)
        t0 = time.time(); subprocess.run(["tools/micro/hip_startup", "lib", "dsk_amd/libdskgpu.so"], env=env, stdout=subprocess.DEVNULL); dt = time.time() - t0
        print(f"deferred={d} whole process {dt:.3f} s")

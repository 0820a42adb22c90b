timeout -k 10 420 python tools/fuzz_parity.py 4000 940000 > gpurun_out/r3h_fuzz_a.txt 2>&1; tail -1 gpurun_out/r3h_fuzz_a.txt
timeout -k 10 300 python tools/fuzz_parity.py 2000 950000 brick > gpurun_out/r3h_fuzz_b.txt 2>&1; tail -1 gpurun_out/r3h_fuzz_b.txt
SVR_LIB=_ab/libs/exp.so SVR_FORCE_BIG=1 SVR_FORCE_ZSPLIT=5 timeout -k 10 300 python tools/fuzz_parity.py 2500 960000 > gpurun_out/r3h_fuzz_c.txt 2>&1; tail -1 gpurun_out/r3h_fuzz_c.txt
SVR_LIB=_ab/libs/exp.so SVR_FORCE_BIG=1 SVR_FORCE_ZSPLIT=9 timeout -k 10 240 python tools/fuzz_parity.py 1200 970000 brick > gpurun_out/r3h_fuzz_d.txt 2>&1; tail -1 gpurun_out/r3h_fuzz_d.txt

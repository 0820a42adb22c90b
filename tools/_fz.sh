timeout -k 10 300 python tools/fuzz_parity.py 3000 980000 > gpurun_out/r3l_fuzz_a.txt 2>&1; tail -1 gpurun_out/r3l_fuzz_a.txt
timeout -k 10 240 python tools/fuzz_parity.py 1500 985000 brick > gpurun_out/r3l_fuzz_b.txt 2>&1; tail -1 gpurun_out/r3l_fuzz_b.txt
SVR_LIB=_ab/libs/exp.so SVR_FORCE_BIG=1 SVR_FORCE_ZSPLIT=5 timeout -k 10 240 python tools/fuzz_parity.py 2000 990000 > gpurun_out/r3l_fuzz_c.txt 2>&1; tail -1 gpurun_out/r3l_fuzz_c.txt
SVR_LIB=_ab/libs/exp.so SVR_FORCE_BIG=1 SVR_FORCE_ZSPLIT=9 timeout -k 10 200 python tools/fuzz_parity.py 1000 995000 brick > gpurun_out/r3l_fuzz_d.txt 2>&1; tail -1 gpurun_out/r3l_fuzz_d.txt

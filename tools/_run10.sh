python -m pytest tests -m gpu -q > gpurun_out/t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t10.log; tail -3 gpurun_out/t10.log
bash tools/profile_bench.sh r02b r02 > gpurun_out/profile_r02b.log 2>&1; tail -5 gpurun_out/profile_r02b.log
mkdir -p profiles/r02 && cp gpurun_out/profiles_r02/traffic.json profiles/r02/traffic.json
python bench.py --steps 20 --warmup 5 --check > gpurun_out/b10.log 2>&1; tail -c 300 gpurun_out/b10.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --source-dtype uint16 > gpurun_out/b10_u16.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ring-storage float32 > gpurun_out/b10_f32.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-collective --check --planes all --tiling 1x1 > gpurun_out/b10_fc.log 2>&1; tail -c 600 gpurun_out/b10_fc.log
python tools/exp_variants.py 1024 0,8 K1,K2,-x,-y,-z,diag > gpurun_out/var10.log 2>&1

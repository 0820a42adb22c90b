python -m pytest tests -m gpu -q > gpurun_out/t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t14.log; tail -3 gpurun_out/t14.log
bash tools/profile_bench.sh r02c r02 > gpurun_out/profile_r02c.log 2>&1; tail -3 gpurun_out/profile_r02c.log
cp gpurun_out/profiles_r02/traffic.json profiles/r02/traffic.json
python bench.py --steps 20 --warmup 5 --check > gpurun_out/b14.log 2>&1; tail -c 300 gpurun_out/b14.log
SVR_STATIC_PLACEMENT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b14_static.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --source-dtype uint16 > gpurun_out/b14_u16.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ring-storage float32 > gpurun_out/b14_f32.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-collective --check --planes all --tiling 1x1 > gpurun_out/b14_fc.log 2>&1
python bench.py --config C5 --no-cpu-baseline > gpurun_out/b14_c5.log 2>&1
python bench.py --config C4 --steps 240 --no-cpu-baseline --blocking-too > gpurun_out/b14_c4.log 2>&1; tail -c 300 gpurun_out/b14_c4.log
python tools/exp_variants.py 1024 0,8 K1,K2,-x,-y,-z,diag > gpurun_out/var14.log 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kernels14 -- python3 $GRAFT_REPO_ROOT/tools/exp_kernels.py > $GRAFT_REPO_ROOT/gpurun_out/kernels14.log 2>&1

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 0 1; do
GECCO_IMGPROJ16=$v python -m pytest tests/test_hip_fullsize.py tests/test_hip_modules.py -q -s -k "(c3_full_size or c4_shape or conditional or ray) and not training" 2>&1 | grep -E "C3|C4|passed|failed|Error" > gpurun_out/r06h_c3_par_$v.log
GECCO_IMGPROJ16=$v python bench.py --config C3 --no-extras --no-cpu-baseline --no-sampler --steps 20 --warmup 5 > gpurun_out/r06h_c3_bench_$v.json 2>gpurun_out/r06h_c3_bench_$v.err
done
GECCO_FWD_STREAMS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/r06h_prof_2 -o c3 --output-format csv -- python3 bench.py --config C3 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-sampler > gpurun_out/r06h_prof_2.log 2>&1
head -n 14 gpurun_out/r06h_prof_2/c3_kernel_stats.csv | cut -c1-160
cat gpurun_out/r06h_c3_par_*.log; for v in 0 1; do python -c "import json;d=json.loads(open('gpurun_out/r06h_c3_bench_$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"; done

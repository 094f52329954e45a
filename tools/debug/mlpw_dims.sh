python tools/debug/mlpw_dims_time.py > gpurun_out/r06i_mlpw_dims_time.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-sampler --steps 30 --warmup 5 > gpurun_out/r06i_bench.json 2> gpurun_out/r06i_bench.err
cat gpurun_out/r06i_mlpw_dims_time.log; python -c "import json;d=json.loads(open('gpurun_out/r06i_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['roofline'])"

python -m pytest tests/test_hip_ops.py -q -x -k "mlp_fused_w" 2>&1 | tail -n 1 > gpurun_out/r06m_mlpw_dims_time.log
python tools/debug/mlpw_dims_time.py 2>&1 | grep "d=" >> gpurun_out/r06m_mlpw_dims_time.log
for v in 1 0; do GECCO_MLPW=$v python bench.py --config C4 --no-extras --no-cpu-baseline --no-sampler --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 mlpw=$v', d['ms_per_step'], d['value'])"; done >> gpurun_out/r06m_mlpw_dims_time.log
cat gpurun_out/r06m_mlpw_dims_time.log

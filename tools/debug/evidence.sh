# One call, one box: counter passes + one-stream kernel trace of the headline forward, the C3 / C4 conditional traces, the training trace,
# the whole GPU suite, smoke, and the driver's bench command.  bash tools/debug/evidence.sh <tag>
TAG=${1:-r06y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
PMC_WRITE=1 bash $R/tools/pmc_collect.sh $TAG > $R/gpurun_out/$TAG/pmc.log 2>&1
bash $R/tools/prof_fwd.sh $TAG > $R/gpurun_out/$TAG/prof_fwd.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in C3 C4; do
GECCO_FWD_STREAMS=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/$c --output-format csv -- python3 $R/bench.py --config $c --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-sampler > $R/gpurun_out/$TAG/$c.log 2>&1
done
bash $R/tools/prof_amp.sh $TAG > $R/gpurun_out/$TAG/prof_amp.log 2>&1
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver_cmd.json 2> gpurun_out/$TAG/bench.err
tail -n 3 gpurun_out/$TAG/gpu_tests.log; tail -n 2 gpurun_out/$TAG/smoke.log
python -c "
import json;d=json.loads(open('gpurun_out/$TAG/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_stale'))
for k in ('train','other_configs','extras'):
    if k in d: print(k, json.dumps(d[k])[:600])
"

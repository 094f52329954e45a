run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-sampler --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms',round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
run X=base
run GECCO_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/var/libgecco_aregplain.so
done

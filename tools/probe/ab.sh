cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_training.py -m gpu -x -q -k "attention or attn or golden" 2>&1 | tail -2
run() { env "$@" timeout 300 python bench.py --train --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*','ms',round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
run X=new
run GECCO_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/var/libgecco_old.so
done

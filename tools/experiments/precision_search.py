"""Per-site precision SEARCH for the denoiser forward: every product site of a layer gets one of the
arithmetic schemes the kernels of this repository implement (or could), the F_x error of the whole C2 network (N = 2048,
d = 384, L = 6) is emulated on the CPU oracle with per-site operand rounding (tools/experiments/fp16_site_sensitivity.py's
machinery), and the recipes are ranked by the per-layer cost model below against the constraint F_x <= 3.3e-4 (1e-3 bar, 3x
margin).  The survivors are re-checked at sigma in {0.002, 1, 165} and on the deeper networks (d, L) = (384, 8), (256, 10),
(128, 14) that tests/test_hip_fullsize.py runs.

  python tools/experiments/precision_search.py [workers] > profiles/r03_precision_search.txt

Schemes per site:  f16 = one fp16 term per operand (1 matrix-pipe unit);  x2a = fp16 A x two-term fp16 W (2 units; `lo8` form:
fp16 + fp8 second term, 1.5);  h8 = fp16 main product + both cross terms on the fp8 instruction (2 units, 3 B / element);
x3 = split-bf16 (3 units, 4 B / element).  Cost model: measured kernel times of this tree in microseconds per layer at C2
(profiles/r03g_fwd_kernel_stats_one_stream.csv for h8 / lo8 / x3-chain, profiles/r02ze_* for the x3 and fp16-mode kernels;
`est` = interpolated where no kernel exists)."""
import itertools
import multiprocessing as mp
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

COST = {   # us per layer at C2 (B 64 x N 2048, d 384)
    "kv_proj": {"f16": 172.0, "x2a_v8": 184.0},
    "chain": {"f16": 60.0, "x2a": 80.0, "x3": 105.0},                        # x2a: est (the fp16 chain kernel with a second weight stream)
    "out_proj": {"f16": 149.0, "x2a": 185.0, "h8": 139.0 + 75.0, "x3": 157.0 + 67.0},   # incl. the unpool attention (fused in f16); x2a est
    "mlp0": {"f16": 136.0, "x2a": 172.0, "h8": 210.0, "x3": 322.0},          # f16: half of the fused MLP; x2a est
    "mlp2": {"f16": 131.0, "x2a": 170.0, "h8": 202.0, "x3": 234.0},
}
FIXED = 59.0 + 12.0   # pool attention, coefficient launches
SITES = ["kv_proj", "chain", "out_proj", "mlp0", "mlp2"]
_STATE = {}


def _setup(d, L, N, sigma, seed=5):
    import fp16_site_sensitivity as fs
    import torch.nn.functional as F
    from oracle import cases, cpu_ref
    from oracle import weights as W
    key = (d, L, N, sigma)
    if key in _STATE:
        return _STATE[key]
    p = W.linear_lift_state_dict(3, d, L, cases.I, cases.H)
    fs.SITE_OF.clear()
    for k, v in p.items():
        if k.endswith("in_proj_weight"):
            C = v.shape[1]
            fs.SITE_OF[v[:C].data_ptr()] = "q_proj"
            fs.SITE_OF[v[C:2 * C].data_ptr()] = "chain"
            fs.SITE_OF[v[2 * C:].data_ptr()] = "chain"
        elif k.endswith("kv_proj.weight"):
            fs.SITE_OF[v.data_ptr()] = "kv_proj"
        elif "broadcast.pool.out_proj" in k or ("broadcast.mlp." in k and k.endswith("weight")):
            fs.SITE_OF[v.data_ptr()] = "chain"
        elif k.endswith("unpool.out_proj.weight"):
            fs.SITE_OF[v.data_ptr()] = "out_proj"
        elif k.endswith(".mlp.0.weight"):
            fs.SITE_OF[v.data_ptr()] = "mlp0"
        elif k.endswith(".mlp.2.weight"):
            fs.SITE_OF[v.data_ptr()] = "mlp2"
    rs = np.random.RandomState(seed)
    data = torch.from_numpy(rs.randn(1, N, 3).astype(np.float32))
    x = data + sigma * torch.from_numpy(rs.randn(1, N, 3).astype(np.float32))
    if not getattr(fs, "_patched", False):
        F.linear = fs.emu_linear
        torch.matmul = fs.emu_matmul
        cpu_ref.attention_pool = fs.wrap_attn(cpu_ref.attention_pool, "pool")
        cpu_ref.mha_unpool = fs.wrap_attn(cpu_ref.mha_unpool, "unpool")
        fs._patched = True
    D = cpu_ref.uncond_denoiser(p, "", cases.H)

    def run(scheme):
        fs.SCHEME.clear()
        fs.SCHEME.update(scheme)
        with torch.no_grad():
            return D(x, torch.tensor([sigma]), return_raw=True)[1]
    raw0 = run({})
    _STATE.clear()   # one network per worker at a time (memory)
    _STATE[key] = (run, raw0, cpu_ref)
    return _STATE[key]


def evaluate(job):
    recipe, d, L, N, sigma = job
    torch.set_num_threads(1)
    run, raw0, cpu_ref = _setup(d, L, N, sigma)
    sch = {"q_proj": "f16", "pool.qk": "f16", "pool.pv": "f16", "unpool.qk": "f16", "unpool.pv": "f16"}
    sch.update(dict(zip(SITES, recipe)))
    e = cpu_ref.rel_err(run(sch), raw0)
    return recipe, (d, L, N, sigma), float(e[0])


def cost(recipe):
    return FIXED + sum(COST[s][r] for s, r in zip(SITES, recipe))


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    space = list(itertools.product(*[list(COST[s]) for s in SITES]))
    print(f"# {len(space)} recipes over sites {SITES}; constraint F_x <= 3.3e-4; cost model in us per layer (x 6 layers per evaluation)")
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(evaluate, [(r, 384, 6, 2048, 0.002) for r in space], chunksize=8)
        table = sorted(((cost(r), r, e) for r, _, e in res))
        print("\n## all recipes at C2 (N 2048, d 384, L 6), sigma = 0.002, sorted by modelled cost")
        print(f"{'us/layer':>9s} {'ms/eval':>8s}  {'F_x max-rel':>11s}  ok   " + "  ".join(f"{s:>8s}" for s in SITES))
        for c, r, e in table:
            print(f"{c:9.0f} {6 * c / 1e3:8.2f}  {e:11.2e}  {'yes' if e <= 3.3e-4 else ' no'}  " + "  ".join(f"{x:>8s}" for x in r))
        feas = [(c, r, e) for c, r, e in table if e <= 3.3e-4][:12]
        print("\n## the 12 cheapest feasible recipes re-checked: sigma in {0.002, 1, 165} at C2 and the deeper networks (sigma = 0.002, N = 1024)")
        jobs = []
        for c, r, e in feas:
            jobs += [(r, 384, 6, 2048, s) for s in (1.0, 165.0)] + [(r, 384, 8, 1024, 0.002), (r, 256, 10, 1024, 0.002), (r, 128, 14, 1024, 0.002)]
        jobs.sort(key=lambda j: j[1:])      # group by network: a worker keeps one network at a time
        chk = {}
        for r, key, e in pool.map(evaluate, jobs, chunksize=max(1, len(feas))):
            chk.setdefault(r, {})[key] = e
        for c, r, e in feas:
            worst = max([e] + list(chk[r].values()))
            print(f"{c:9.0f} us  " + "  ".join(f"{x:>8s}" for x in r) + f"   sigma .002: {e:.2e}  " +
                  "  ".join(f"{k[0]}/{k[1]}/s{k[3]:g}: {v:.2e}" for k, v in sorted(chk[r].items())) + f"   worst {worst:.2e} {'OK' if worst <= 3.3e-4 else 'FAILS'}")
    print("\n# shipped this round (mixed): kv_proj x2a_v8, chain x3, out_proj h8, mlp0 h8, mlp2 h8")


if __name__ == "__main__":
    main()

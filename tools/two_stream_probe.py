"""Probe: the C2 evaluation as two half-batches on two streams (one captured graph with a fork / join) against one batch."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from gecco_amd import hip_ops as ops
import __graft_entry__ as ge
ge.build()
ops.set_default_precision("mixed")
dev = torch.device("cuda:0")
model = bench.build_model(bench.random_state_dict(3)).to(dev).eval()
x, sigma = (t.to(dev) for t in bench.synthetic_cloud(0))
with torch.no_grad():
    run1 = model.graphed_forward(x, sigma, None)
    t1 = bench.time_events(run1, 30, warmup=5)
    print(f"1 x 64 on one stream: {t1:.3f} ms")
    for parts in (2, 4):
        bs = bench.B // parts
        models = [model] + [copy.deepcopy(model) for _ in range(parts - 1)]
        xs = [x[i * bs:(i + 1) * bs].contiguous() for i in range(parts)]
        ss = [sigma[i * bs:(i + 1) * bs].contiguous() for i in range(parts)]
        outs = [torch.empty_like(v) for v in xs]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        def step():
            main = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(main)
            for m, st, a, b, o in zip(models, streams, xs, ss, outs):
                with torch.cuda.stream(st):
                    m.forward(a, b, None, out=o)
            for st in streams:
                main.wait_stream(st)
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        t = bench.time_events(g.replay, 30, warmup=5)
        ref = run1()
        err = max(float((o - ref[i * bs:(i + 1) * bs]).abs().max()) for i, o in enumerate(outs))
        print(f"{parts} x {bs} on {parts} streams: {t:.3f} ms (max abs diff vs one batch {err:.2e})")

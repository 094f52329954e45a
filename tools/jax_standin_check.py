"""SUPPORTING evidence for the "parity unpinned" rows a14 / f4 (it pins nothing: jax, kornia and ott are absent from the image).

gecco-jax carries its own JAX ports of the camera functions the torch reference takes from kornia (geometry.py:27-83:
convert_points_from_homogeneous with kornia's `where(|z| > eps, 1 / (z + eps), 1)` convention, project_points, unproject_points)
and the metrics (geometry.py:8-24 distance_matrix; metrics.py:92-139 chamfer_distance, scipy_emd).  Those functions only use
`jax.numpy` as an array namespace, so this script EXECUTES THE REFERENCE'S OWN SOURCE (read from /root/reference at run time,
never copied into the repository) on a numpy stand-in for `jax.numpy` and compares the results with oracle/cpu_ref.py's
restatements in float64.  Build container only (the reference does not exist on the GPU box).

  python tools/jax_standin_check.py  > profiles/r03_jax_standin_check.txt
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/gecco-jax/src/gecco_jax"


def install_standins():
    jnp = types.ModuleType("jax.numpy")
    for name in dir(np):
        if not name.startswith("_"):
            setattr(jnp, name, getattr(np, name))
    jnp.linalg = np.linalg
    jax = types.ModuleType("jax")
    jax.numpy = jnp
    jax.pure_callback = lambda fn, shapes, *args, **kw: fn(*args)
    jax.lax = types.SimpleNamespace(stop_gradient=lambda x: x)
    dim = types.ModuleType("torch_dimcheck")
    dim.dimchecked = lambda f: f

    class _A:
        def __class_getitem__(cls, item):
            return cls
    dim.A = _A
    sys.modules.update({"jax": jax, "jax.numpy": jnp, "torch_dimcheck": dim})
    return jax, jnp


def load_geometry():
    spec = importlib.util.spec_from_file_location("gecco_jax_geometry_ref", os.path.join(REF, "geometry.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_metric_functions(geometry, jax, jnp):
    """chamfer_distance, chamfer_distance_squared, _scipy_lsa, scipy_emd: metrics.py:92-139 executed as they stand (the module's
    other imports — equinox, the model classes — are not needed by these four functions)."""
    lines = open(os.path.join(REF, "metrics.py")).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith("def chamfer_distance(")) - 1     # its @dimchecked line
    end = next(i for i, ln in enumerate(lines) if ln.startswith("def sinkhorn_emd(")) - 1
    from scipy.optimize import linear_sum_assignment
    from typing import Tuple
    ns = {"jnp": jnp, "jax": jax, "np": np, "linear_sum_assignment": linear_sum_assignment, "Tuple": Tuple,
          "dimchecked": lambda f: f, "A": sys.modules["torch_dimcheck"].A, "distance_matrix": geometry.distance_matrix}
    exec(compile("\n".join(lines[start:end]), os.path.join(REF, "metrics.py"), "exec"), ns)
    return ns


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference is not present: this check runs in the build container only")
    jax, jnp = install_standins()
    geo = load_geometry()
    met = load_metric_functions(geo, jax, jnp)
    from oracle import cpu_ref
    rs = np.random.RandomState(0)
    worst = 0.0
    print("# reference source executed on a numpy stand-in for jax.numpy vs oracle/cpu_ref.py (float64); max-norm relative differences")
    for n, m in ((257, 300), (1024, 1024)):
        a, b = rs.randn(n, 3), rs.randn(m, 3) * 1.3 + 0.2
        for sq in (False, True):
            e = rel(cpu_ref.distance_matrix(torch.from_numpy(a), torch.from_numpy(b), sq).numpy(), geo.distance_matrix(a, b, squared=sq))
            worst = max(worst, e)
            print(f"distance_matrix  N={n} M={m} squared={sq}: {e:.2e}")
    for n in (256, 2048):
        a, b = rs.randn(n, 3), rs.randn(n, 3) * 0.9
        for sq in (False, True):
            e = rel(float(cpu_ref.chamfer_distance(torch.from_numpy(a), torch.from_numpy(b), sq)), met["chamfer_distance"](a, b, squared=sq))
            worst = max(worst, e)
            print(f"chamfer_distance N={n} squared={sq}: {e:.2e}")
    # exact EMD: the reference's scipy_emd against linear_sum_assignment on the oracle's distance matrix (what gecco_amd/metrics.py does
    # with the device's matrix)
    from scipy.optimize import linear_sum_assignment
    for match, avg in (("l1", "l1"), ("l2", "l2"), ("l2", "l1")):
        a, b = rs.randn(200, 3), rs.randn(200, 3)
        md = cpu_ref.distance_matrix(torch.from_numpy(a), torch.from_numpy(b), match == "l2").numpy()
        ad = md if avg == match else cpu_ref.distance_matrix(torch.from_numpy(a), torch.from_numpy(b), avg == "l2").numpy()
        r, c = linear_sum_assignment(md)
        e = rel(ad[r, c].mean(), met["scipy_emd"](a, b, match=match, average=avg))
        worst = max(worst, e)
        print(f"scipy_emd match={match} average={avg}: {e:.2e}")
    # the pinhole camera: K with zero skew (every shipped config), points in front of and AT the camera plane (|z| <= eps branch)
    K = np.array([[1.1, 0.0, 0.5], [0.0, 1.3, 0.45], [0.0, 0.0, 1.0]])
    xyz = rs.randn(500, 3)
    xyz[:, 2] = np.abs(xyz[:, 2]) + 0.2
    xyz[:5, 2] = [0.0, 1e-9, -1e-9, 5e-9, 1e-7]
    got = cpu_ref.project_points(torch.from_numpy(xyz)[None], torch.from_numpy(K)[None])[0].numpy()
    e = rel(got, geo.project_points(xyz, K))
    # not counted in `worst`: the JAX port divides K xyz by (z + eps) — the principal point enters as c z / (z + eps) — while
    # kornia (and the oracle) divides xyz first and adds c: the two differ by c eps / (z + eps) ~ 1e-8 by construction
    print(f"project_points (incl. |z| <= 1e-8 rows): {e:.2e}   (eps convention: c eps / (z + eps) apart by construction; bar 1e-7)")
    assert e < 1e-7, e
    uv, depth = rs.rand(500, 2), rs.rand(500) * 3 + 0.1
    for normalized in (True, False):
        got = cpu_ref.unproject_points(torch.from_numpy(uv)[None], torch.from_numpy(depth)[None, :, None], torch.from_numpy(K)[None],
                                       normalized)[0].numpy()
        e = rel(got, geo.unproject_points(uv, depth, K, normalized))
        worst = max(worst, e)
        print(f"unproject_points normalized={normalized}: {e:.2e}")
    print(f"# worst {worst:.2e}")
    assert worst < 1e-12, worst
    print("# OK: the oracle's restatements agree with the reference's JAX ports to float64 rounding (supporting evidence only)")


if __name__ == "__main__":
    main()

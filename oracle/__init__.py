"""CPU oracle for the GECCO denoiser hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference
`gecco_torch` forward path.  It exists to *check* the HIP product path:

* only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
  `bench.py` may import it;
* nothing under `gecco_amd/` imports it, and the product path raises when the
  HIP library is missing instead of falling back to this code.

Pinning: `tools/make_golden.py` imports the real reference from
`/root/reference` (only possible in the build container), checks this
restatement against it and writes `tests/golden/*.npz`.  The unconditional path
(a1-a10, a16-a18 of SURVEY.md section 8) is pinned by those vectors.  The two
kornia functions (`project_points`, `unproject_points`) are NOT in
`/root/reference` (unpinned third-party dependency, `gecco-torch/pyproject.toml:25`):
for them this oracle *is* the definition -> "parity unpinned" for a14 only.
"""

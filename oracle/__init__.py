"""CPU oracle for the MEH/HUA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain torch-fp32 / numpy restatement of the reference algorithm for every row
of SURVEY.md section 8(a), each function citing the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package -- and there only as the checker / the reported CPU baseline.
`aod_meh_hua_amd` (the product) never imports it; the product path fails loudly
if its HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * pinned against outputs of the reference itself, captured in this container by
    tools/golden/make_golden.py (reference imported under tools/golden/mmcv_shim.py)
    and committed as tests/golden/*.npz; tests/test_oracle_golden.py checks every one.
  * pinned against the reference's docstring known-answer examples
    (AnchorGenerator, MaxIoUAssigner, delta2bbox, l1_loss).
  * mmcv-full 1.3.8 `sigmoid_focal_loss` and `nms` are NOT in /root/reference:
    "parity unpinned" at that boundary (restated from the published kernels and
    cross-checked against the in-tree py_sigmoid_focal_loss formula,
    mmdet/models/losses/focal_loss.py:11-56).
"""

"""INTEGRATION.md section 2 is code a maintainer of the reference is told to paste: the first python block there is
executed as written (only the library path is filled in) and must train exactly like mfcd.engine does."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "mfcd_binding.py" in b)
    return stub.replace('ctypes.CDLL("libmfcd_hip.so")', f'ctypes.CDLL({os.path.join(PKG, "libmfcd_hip.so")!r})')


def test_integration_stub_names_only_exported_entries():
    import ctypes
    src = _stub_source()
    lib = ctypes.CDLL(os.path.join(PKG, "libmfcd_hip.so"))
    names = set(re.findall(r"L\.(mfcd_\w+)", src))
    assert {"mfcd_train_workspace_bytes", "mfcd_train_workspace_init", "mfcd_train_steps", "mfcd_eval_batches"} <= names
    for nm in names:
        assert hasattr(lib, nm), nm
    compile(src, "INTEGRATION.md:mfcd_binding", "exec")


@pytest.mark.gpu
def test_integration_stub_trains_like_the_engine():
    import structure as S
    from mfcd import engine
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md:mfcd_binding", "exec"), ns)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    n, m, d, B, N = 300, 200, 16, 64, 64 * 9 + 17

    class DS:       # what BTLPreferenceDataset exposes: .data = list of (u, i, j, z)
        data = [(int(rng.integers(0, n)), int(rng.integers(0, m)), int(rng.integers(0, m)), float(rng.integers(0, 2)))
                for _ in range(N)]

    torch.manual_seed(3)
    a = S.MatrixFactorization(n, m, d).to(dev)
    b = S.MatrixFactorization(n, m, d).to(dev)
    b.load_state_dict(a.state_dict())
    oa = torch.optim.Adam(a.parameters(), lr=1e-3, weight_decay=1e-5)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=1e-5)
    rec = ns["records"](DS, dev)
    ws = ns["plan_workspace"](N, B, a)
    la = torch.cat([ns["fused_epoch"](a, oa, rec, B, ws) for _ in range(2)])
    assert int(ws[:4].view(torch.int32).item()) == 0
    bind = engine.AdamBinding(b, ob)
    lb = torch.cat([engine.train_steps(bind, rec, B).clone() for _ in range(2)])
    assert torch.equal(la, lb)
    assert torch.equal(a.U.data, b.U.data) and torch.equal(a.V.data, b.V.data)
    assert float(oa.state[a.U]["step"]) == float(ob.state[b.U]["step"]) == 2 * ((N + B - 1) // B)

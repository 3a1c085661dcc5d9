"""pdegym.common.transforms against values captured from the reference (transforms.py)."""
import numpy as np
import pytest
import torch

from pdegym.common import transforms as T
from pdecontrol.mbrl.types import Sample


def test_normalize_running_statistics(sur_golden):
    n = T.Normalize(aggregate=True, batched=True)
    n.update(torch.from_numpy(sur_golden["norm_b1"]))
    n.update(torch.from_numpy(sur_golden["norm_b2"]))
    assert n.count == int(sur_golden["norm_count"]) and n.mean.shape == (1, 1, 1)
    np.testing.assert_array_equal(n.mean.numpy(), sur_golden["norm_mean"])
    np.testing.assert_array_equal(n.var.numpy(), sur_golden["norm_var"])
    np.testing.assert_array_equal(n(torch.from_numpy(sur_golden["norm_b1"])).numpy(), sur_golden["norm_apply"])
    np.testing.assert_array_equal(n.Inverse(sur_golden["norm_b1"]), sur_golden["norm_inverse"])  # numpy in/out
    assert n.Inverse.Inverse is n
    n.reset()
    assert n.mean is None and n.count == 0
    f = T.Normalize(frozen=True)
    f.update(torch.zeros(2, 1, 4))
    assert f.mean is None


def test_normalize_dims():
    assert T.Normalize(aggregate=True, batched=True).dim == (0, 1, 2)
    assert T.Normalize(aggregate=True).dim == (0, 1)
    assert T.Normalize(batched=True).dim == (0, 1)
    assert T.Normalize().dim == (0,)


def test_inverse_update_feeds_forward_statistics():
    n = T.Normalize(aggregate=True, batched=True)
    n.mean, n.var, n.count = torch.full((1, 1, 1), 2.0), torch.full((1, 1, 1), 4.0), 10
    x = torch.randn(3, 1, 8)
    ref = T.Normalize(aggregate=True, batched=True)
    ref.mean, ref.var, ref.count = n.mean.clone(), n.var.clone(), 10
    ref.update(n.Inverse(x))
    n.Inverse.update(x)
    torch.testing.assert_close(n.mean, ref.mean)
    assert n.count == 13


def test_scale_transform_roundtrip_and_update():
    s = T.ScaleTransform(scale=(-1.0, 1.0), aggregate=True, batched=True)
    x = torch.tensor([[[0.0, 2.0, 4.0]], [[-2.0, 1.0, 3.0]]])
    s.update(x)
    assert s.vmin.shape == (1, 1, 1) and float(s.vmin) == -2.0 and float(s.vmax) == 4.0
    y = s(x)
    assert float(y.min()) == -1.0 and float(y.max()) == 1.0
    torch.testing.assert_close(s.Inverse(y), x)
    s.update(torch.full((1, 1, 3), 10.0))
    assert float(s.vmax) == 10.0 and float(s.vmin) == -2.0
    b = T.ScaleTransform(scale=(0.0, 1.0), bounds=(-1.0, 1.0), frozen=True)
    b.update(torch.full((1, 1, 3), 10.0))
    assert float(b.vmax) == 1.0
    np.testing.assert_allclose(b(np.array([0.0], dtype=np.float32)), [0.5])
    # 0-d inputs (np.float32, 0-d arrays, torch scalars) as the reference's expression handles them
    assert float(b(np.float32(0.5))) == 0.75 and float(b(np.asarray(-1.0, dtype=np.float32))) == 0.0
    assert float(b(torch.tensor(0.5))) == 0.75 and float(b.Inverse(torch.tensor(0.75))) == 0.5


def test_sensor_func_identity_operation_batch():
    x = torch.arange(16.0).reshape(1, 1, 16)
    st = T.SensorTransform(4)
    assert st(x).flatten().tolist() == [2.0, 6.0, 10.0, 14.0]
    with pytest.raises(NotImplementedError):
        st.Inverse(x)
    assert T.SensorTransform(1).Inverse(x) is x
    f = T.FuncTransform(lambda a, b: a + b, inverse=lambda a, b: a - b)
    out = f(np.ones(3), np.ones(3))
    assert isinstance(out, np.ndarray) and out.tolist() == [2, 2, 2]
    assert f.Inverse(torch.ones(3), torch.ones(3)).tolist() == [0, 0, 0]
    ident = T.Identity()
    assert ident(x) is x and ident.Inverse(x) is x
    n = T.Normalize(aggregate=True, batched=True)
    n.mean, n.var = torch.full((1, 1, 1), 1.0), torch.full((1, 1, 1), 3.0)
    op = T.Operation([T.BatchTransform(n), T.BatchTransform(T.SensorTransform(1))])
    xb = torch.randn(4, 5, 1, 16)
    per_item = torch.stack([n(item) for item in xb])
    torch.testing.assert_close(op(xb), per_item, rtol=0, atol=0)
    torch.testing.assert_close(op.Inverse(op(xb)), xb, rtol=1e-6, atol=1e-6)
    # a non-batchable inner transform falls back to the per-item loop
    loop = T.BatchTransform(T.FuncTransform(lambda a: a * 2))
    torch.testing.assert_close(loop(xb), xb * 2)
    # BatchTransform.update feeds items one by one (running statistics)
    m1, m2 = T.Normalize(aggregate=True, batched=True), T.Normalize(aggregate=True, batched=True)
    T.BatchTransform(m1).update(xb)
    for item in xb:
        m2.update(item)
    torch.testing.assert_close(m1.mean, m2.mean, rtol=0, atol=0)
    assert m1.count == 4 * 5


def test_sample_transform_and_types():
    n = T.Normalize(aggregate=True, batched=True)
    n.mean, n.var = torch.zeros(1, 1, 1), torch.ones(1, 1, 1)
    stf = T.SampleTransform(otransf=T.BatchTransform(n))
    s = Sample(np.ones((2, 3, 1, 4), np.float32), np.ones((2, 3, 1, 4), np.float32), np.ones((2, 3, 1, 4), np.float32),
               np.zeros((2, 3)), np.zeros((2, 3), bool), np.zeros((2, 3), bool), np.zeros((2, 3), np.int32))
    out = stf(s)
    assert isinstance(out, Sample) and len(list(out)) == 7
    np.testing.assert_allclose(out.obs, 1 / np.sqrt(1 + 1e-4), rtol=1e-6)
    np.testing.assert_array_equal(out.actions, s.actions)
    back = stf.Inverse(out)
    np.testing.assert_allclose(back.obs, s.obs, rtol=1e-6)
    parts = s.split(axis=0)
    assert len(parts) == 2 and parts[0].obs.shape == (3, 1, 4)
    t = Sample(*[np.asarray(v) for v in s]).totorch()
    assert t.obs.dtype == torch.float32 and t.truncated.dtype == torch.bool and t.steps.dtype == torch.int32
    assert t.tonumpy().obs.dtype == np.float32


def test_device_aware_statistics():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = T.Normalize(aggregate=True, batched=True)
    n.mean, n.var = torch.full((1, 1, 1), 0.5), torch.full((1, 1, 1), 2.0)
    x = torch.randn(2, 3, 1, 8, device="cuda")
    y = T.BatchTransform(n).Inverse(x)
    assert y.device.type == "cuda"
    torch.testing.assert_close(y.cpu(), n.Inverse(x.cpu()))

"""torch.ops.silent_speech.* (torch_ops.py): registered, shape-inferable through their fake implementations, differentiable where the
reference differentiates, and actually on the path of the drop-in entry points."""
import numpy as np
import pytest
import torch

from oracle import dtw_ref
from silent_speech_amd import torch_ops
from silent_speech_amd.architecture import Model
from tests.backend import dev, is_emu  # noqa: F401


def test_ops_are_registered_with_schemas():
    for name in torch_ops.OPS:
        op = getattr(torch.ops.silent_speech, name)
        schema = str(op.default._schema)
        assert schema.startswith('silent_speech::' + name), schema
    assert 'Tensor(a0!) p' in str(torch.ops.silent_speech.fused_adamw.default._schema)        # the optimiser step declares what it mutates


def test_dtw_align_op_matches_oracle_and_has_a_fake(dev):
    rng = np.random.default_rng(4)
    c = rng.random((37, 52), dtype=np.float32)
    t = torch.from_numpy(c).to(dev)
    got = torch.ops.silent_speech.dtw_align(t)
    assert got.dtype == torch.int32 and got.tolist() == list(dtw_ref.align_from_distances_c(c))
    gt = torch.ops.silent_speech.dtw_align(t.t())                     # a strided view, read in place (transduction_model.py:126 passes costs.T)
    assert gt.tolist() == list(dtw_ref.align_from_distances_c(np.ascontiguousarray(c.T)))
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        f = torch.ops.silent_speech.dtw_align(torch.empty(37, 52))
        assert tuple(f.shape) == (37,) and f.dtype == torch.int32
        m = torch.ops.silent_speech.stft_logmel(torch.empty(3, 4096), 1024, 80, 22050, 256, 1024, 0, 8000, False)
        assert tuple(m.shape) == (3, 80, 1 + (4096 + 768 - 1024) // 256)


def test_functional_ops_pass_opcheck(dev):
    """torch.library.opcheck on the ops that declare no mutation (schema vs behaviour, fake implementation vs real shapes / dtypes, AOT dispatch):
    what torch.compile / functionalization rely on.  silent_speech::model_forward is functional since round 5 (the shifted input comes back as an
    output; Model.forward writes it into x_raw like architecture.py:67-68)."""
    rng = np.random.default_rng(9)
    c = torch.from_numpy(rng.random((23, 31), dtype=np.float32)).to(dev)
    torch.library.opcheck(torch.ops.silent_speech.dtw_align.default, (c,), test_utils=('test_schema', 'test_faketensor'))
    y = torch.from_numpy(rng.standard_normal((2, 2048)).astype(np.float32) * 0.1).to(dev)
    torch.library.opcheck(torch.ops.silent_speech.stft_logmel.default, (y, 1024, 80, 22050, 256, 1024, 0, 8000, False), test_utils=('test_schema', 'test_faketensor'))
    torch.manual_seed(0)
    model = Model(112, 80, 48, model_size=16, num_layers=1, dropout=0.0, compute_dtype=torch.float32).to(dev).eval()
    x = torch.randn(2, 8 * 16, 8).to(dev)
    before = x.clone()
    args = (x, model.w_out.weight, torch_ops.model_handle(model), False, 0, 1)
    torch.library.opcheck(torch.ops.silent_speech.model_forward.default, args, test_utils=('test_schema', 'test_faketensor'))
    assert torch.equal(x, before)


def test_model_forward_is_a_dispatcher_op_with_autograd(dev):
    torch.manual_seed(0)
    model = Model(112, 80, 48, model_size=16, num_layers=1, dropout=0.0, compute_dtype=torch.float32).to(dev).train()
    x = torch.randn(2, 8 * 16, 8).to(dev)
    pred, aux = model(None, x, None)
    node, names = pred.grad_fn, []
    while node is not None and len(names) < 8:
        names.append(type(node).__name__)
        node = node.next_functions[0][0] if node.next_functions else None
    assert any('silent_speech' in n and 'model_forward' in n for n in names), names
    (pred.sum() + aux.sum()).backward()
    g = model.w_out.weight.grad
    assert g is not None and float(g.abs().sum()) > 0
    with pytest.raises(RuntimeError):                                 # the saved context is consumed by its one backward
        torch.ops.silent_speech.model_backward(torch.zeros(2 * 16, 128).to(dev), torch_ops.model_handle(model), model.last_seed)
    # a training-mode forward that is never back-propagated does not pin workspaces without bound
    for _ in range(5):
        model(None, x, None)
    assert len(model._saved_ctx) <= 2
    model.eval()
    with torch.no_grad():
        p2, _ = model(None, x, None)
    assert p2.grad_fn is None


def test_fused_adamw_op_matches_torch_adamw(dev):
    g = torch.Generator().manual_seed(2)
    n = 1000
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.1)
    ref.grad = gr.clone()
    opt.step()
    p, m, v = p0.clone().to(dev), torch.zeros(n).to(dev), torch.zeros(n).to(dev)
    torch.ops.silent_speech.fused_adamw(p, gr.to(dev), m, v, n, 1e-2, 1, 0.9, 0.999, 1e-8, 0.1, 1.0)
    assert float((p.cpu() - ref.detach()).abs().max()) < 1e-6

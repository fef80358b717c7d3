"""Learner-side HIP kernels against the torch expressions that define them (which test_learner_cpu.py pins to the reference)."""
import copy

import numpy as np
import pytest
import torch

from phc_amd.learning.running_mean_std import RunningMeanStd

pytestmark = pytest.mark.gpu


def _pair(cols, seed=0):
    g = torch.Generator().manual_seed(seed)
    cpu = RunningMeanStd(cols)
    cpu.running_mean.copy_(torch.randn(cols, generator=g, dtype=torch.float64) * 0.5)
    cpu.running_var.copy_(torch.rand(cols, generator=g, dtype=torch.float64) * 2 + 0.05)
    cpu.count.fill_(1234.0)
    dev = copy.deepcopy(cpu).cuda()
    return cpu, dev, g


@pytest.mark.parametrize("rows,cols", [(5000, 934), (4096, 1960), (70, 3), (1, 934)])
def test_running_norm_kernel_equals_torch_path(rows, cols):
    """phc_running_norm == RunningMeanStd.forward (running_mean_std.py:69-111): normalised + clamped output bit-equal to the IEEE
    fp32 evaluation of the reference expression (numpy; torch's AVX-512 CPU kernels deviate from it by up to 2 ulp themselves), bf16
    output == its rounding, running statistics after three train-mode batches to fp64 round-off of the fp32 batch moments."""
    cpu, dev, g = _pair(cols)
    cpu.train(); dev.train()
    for it in range(3):
        x = torch.randn(rows, cols, generator=g) * (1 + it) + 0.3 * it
        x[0, 0] = 40.0   # clamped
        m32, v32 = cpu.running_mean.numpy().astype(np.float32), cpu.running_var.numpy().astype(np.float32)
        exact = np.clip((x.numpy() - m32) / np.sqrt(v32 + np.float32(1e-5)), -5.0, 5.0)
        y_ref = cpu(x)
        y = dev(x.cuda())
        np.testing.assert_array_equal(y.cpu().numpy(), exact)
        torch.testing.assert_close(y.cpu(), y_ref, rtol=3e-7, atol=1e-6)
        if rows > 1:
            np.testing.assert_allclose(dev.running_mean.cpu().numpy(), cpu.running_mean.numpy(), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(dev.running_var.cpu().numpy(), cpu.running_var.numpy(), rtol=2e-5)
        assert float(dev.count) == float(cpu.count)
    # eval mode: statistics untouched, bf16 output == rounding of the fp32 output
    cpu.eval(); dev.eval()
    before = dev.running_mean.clone()
    x = torch.randn(rows, cols, generator=g)
    yb = dev(x.cuda(), out_dtype=torch.bfloat16)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb.cpu(), dev(x.cuda()).cpu().to(torch.bfloat16)) and torch.equal(before, dev.running_mean)


def test_running_norm_frozen_source_and_update_only():
    """`norm_from` (amp_agent.py:527-532: the output comes from the frozen epoch-start copy while the live statistics update) and
    the store-free update (`want_output=False`)."""
    cpu, dev, g = _pair(358, seed=3)
    frozen_cpu, frozen_dev = copy.deepcopy(cpu), copy.deepcopy(dev)
    frozen_cpu.freeze(); frozen_dev.freeze()
    cpu.train(); dev.train()
    x = torch.randn(3000, 358, generator=g) * 3
    out_ref = frozen_cpu(x)
    cpu(x)
    out = dev(x.cuda(), norm_from=frozen_dev)
    torch.testing.assert_close(out.cpu(), out_ref, rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(dev.running_mean.cpu().numpy(), cpu.running_mean.numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(frozen_dev.running_mean.cpu(), frozen_cpu.running_mean)
    cpu(x)
    assert dev(x.cuda(), want_output=False) is None
    np.testing.assert_allclose(dev.running_var.cpu().numpy(), cpu.running_var.numpy(), rtol=2e-5)
    x[5, 7] = float("nan")
    dev.eval()
    assert torch.isnan(dev(x.cuda())[5, 7])

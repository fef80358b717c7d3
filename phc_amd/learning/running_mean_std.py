"""P1: running mean / variance normaliser with fp64 statistics.

Same behaviour as the reference `RunningMeanStd` (phc/utils/running_mean_std.py:10-111): fp64 buffers `running_mean`,
`running_var`, `count`; forward normalises and clamps to +-5 (or un-normalises with `unnorm=True`), and in train mode
(unless frozen) folds the batch moments in with the parallel-variance update (:56-67) AFTER computing the output.
`sync()` averages the moments across ranks (the reference's `hvd.sync_stats`, common_agent.py:126-127).

On the device the whole forward -- normalise + clamp, batch mean / variance, the fp64 moment update -- is one HIP pass over the
batch plus a one-block finish (`phc_running_norm`, csrc/phc_learn.hip) instead of ~40 torch launches; `norm_from` lets the
output come from a frozen copy while this module's statistics keep updating (amp_agent.py:527-532), `out_dtype=torch.bfloat16`
writes the tensor the bf16 GEMMs read.  The torch expressions below remain the CPU path and the definition the kernel is tested
against.
"""
import torch
from torch import nn


class RunningMeanStd(nn.Module):
    def __init__(self, insize, epsilon=1e-05, per_channel=False, norm_only=False):
        super().__init__()
        self.insize = insize
        self.mean_size = insize[0] if isinstance(insize, (tuple, list)) else insize
        self.epsilon = epsilon
        self.norm_only = norm_only
        self.per_channel = per_channel
        if per_channel:
            raise NotImplementedError("per_channel normalisation is not used on this path")
        self.axis = [0]
        self.register_buffer("running_mean", torch.zeros(self.mean_size, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(self.mean_size, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))
        self.forzen = False  # (sic) attribute names of the reference
        self.forzen_partial = False

    def freeze(self):
        self.forzen = True

    def unfreeze(self):
        self.forzen = False

    def freeze_partial(self, diff):
        self.forzen_partial = True
        self.diff = diff

    @staticmethod
    def _update_mean_var_count_from_moments(mean, var, count, batch_mean, batch_var, batch_count):
        delta = batch_mean - mean
        tot_count = count + batch_count
        new_mean = mean + delta * batch_count / tot_count
        m_a = var * count
        m_b = batch_var * batch_count
        M2 = m_a + m_b + delta ** 2 * count * batch_count / tot_count
        return new_mean, M2 / tot_count, tot_count

    def _fused_ok(self, input, unnorm):
        return (input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and input.is_contiguous() and not unnorm
                and not self.norm_only and not self.forzen_partial and input.shape[1] == self.mean_size)

    def _forward_fused(self, input, src, out_dtype, want_output, row_index=None, out=None):
        from .. import _lib as L
        lib = L.load()
        rows, cols = input.shape
        if row_index is not None:
            assert row_index.dtype == torch.int64 and row_index.is_contiguous() and row_index.device == input.device
            rows = row_index.numel()
        update = self.training and not self.forzen
        if out is not None:   # (may be the left `cols` columns of a wider, K-padded buffer: rows `out.stride(0)` apart)
            assert out.shape == (rows, cols) and out.dtype == out_dtype and out.stride(1) == 1 and out.device == input.device
        elif want_output:
            out = torch.empty((rows, cols), dtype=out_dtype, device=input.device)
        ws = None
        if update:
            need = lib.phc_running_norm_workspace(rows, cols) // 8
            if getattr(self, "_ws_need", None) != need or self._ws.device != input.device:
                # exact size: the kernel's ticket counter sits in the last 8 bytes (zero-initialised here, left at zero by the kernel)
                self._ws, self._ws_need = torch.zeros(need, dtype=torch.float64, device=input.device), need
            ws = self._ws
        ptr = lambda t: None if t is None else t.data_ptr()
        L.check(lib.phc_running_norm(input.data_ptr(), ptr(row_index), rows, cols, src.running_mean.data_ptr(), src.running_var.data_ptr(), float(src.epsilon), 5.0,
                                     ptr(out), int(out_dtype == torch.bfloat16), 0 if out is None else int(out.stride(0)), ptr(self.running_mean if update else None),
                                     ptr(self.running_var if update else None), ptr(self.count if update else None), ptr(ws),
                                     torch.cuda.current_stream(input.device).cuda_stream), "phc_running_norm")
        return out

    def forward(self, input, unnorm=False, norm_from=None, out_dtype=None, want_output=True, row_index=None, out=None):
        """`norm_from`: module whose statistics produce the output (default: this one, before its update); `out_dtype`: fp32
        (default) or bf16; `want_output=False`: only fold the batch into the statistics (device path skips the store);
        `row_index`: operate on input[row_index] (the device pass reads the rows in place); `out`: device path only, the tensor
        (e.g. a row block of a larger buffer) that receives the output."""
        src = norm_from if norm_from is not None else self
        if self._fused_ok(input, unnorm):
            return self._forward_fused(input, src, out_dtype or torch.float32, want_output, row_index, out)
        if row_index is not None:
            input = input[row_index]
        mean, var = src.running_mean, src.running_var
        if unnorm:
            y = torch.clamp(input, min=-5.0, max=5.0)
            y = torch.sqrt(var.float() + self.epsilon) * y + mean.float()
        elif self.norm_only:
            y = input / torch.sqrt(var.float() + self.epsilon)
        else:
            y = (input - mean.float()) / torch.sqrt(var.float() + self.epsilon)
            y = torch.clamp(y, min=-5.0, max=5.0)
        if self.training and not self.forzen:
            with torch.no_grad():
                bm = input.mean(self.axis)
                bv = input.var(self.axis)
                nm, nv, nc = self._update_mean_var_count_from_moments(self.running_mean, self.running_var, self.count, bm, bv, input.size()[0])
                if self.forzen_partial:
                    self.running_mean[-self.diff:], self.running_var[-self.diff:] = nm[-self.diff:], nv[-self.diff:]
                    self.count.copy_(nc)
                else:
                    self.running_mean.copy_(nm)
                    self.running_var.copy_(nv)
                    self.count.copy_(nc)
        return y if out_dtype is None else y.to(out_dtype)

    @torch.no_grad()
    def sync(self, dist):
        """Average (mean, var, count) over ranks so that every replica normalises identically."""
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return
        flat = torch.cat([self.running_mean, self.running_var, self.count.reshape(1)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        n = self.mean_size
        self.running_mean.copy_(flat[:n])
        self.running_var.copy_(flat[n:2 * n])
        self.count.copy_(flat[2 * n])

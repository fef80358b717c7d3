"""Device-side pieces of the PPO / AMP update that are not GEMMs (csrc/phc_learn.hip) and the layers that use them.

* `FastLinear` -- `nn.Linear` (same parameters, same state-dict keys) whose TRAINING pass under bf16 autocast on the device is one
  autograd node built for this problem shape (16 384-row batches, 10^2..10^3 features): bf16 addmm forward; weight gradient dY^T X
  as a batched GEMM over 8 row chunks + fp32 sum (the library's pick for a 16 384-long reduction is a 240-workgroup kernel without
  split-K: 105-115 us vs 44-68 us, scripts/probes/gemm_probe2.py); bias gradient by `phc_colsum_bf16`; one-output layers (the value head)
  by `phc_linear1_*`.  Used for actor, critic, PNN columns.
* `FastLinearDD` -- the same, differentiable twice, for the discriminator MLP whose gradient penalty differentiates the backward pass.
* `ppo_loss`, `disc_bce`, `weighted_sumsq` -- the loss terms with their gradients as kernels (unit-weight convention, see `_PPOLossFn`).
* `policy_sample` -- the rollout's sampling / neglogp / value un-normalisation in one kernel.
* `adam_clip_step` -- clip_grad_norm_ + torch.optim.Adam.step on the flat parameter in two launches, also maintaining the bf16
  parameter copy the layers read inside `FlatGradBucket.shadow_scope()`.
Anywhere else (CPU, fp32, frozen columns) the layers are exactly nn.Linear and the agent uses the torch expressions the kernels are
tested against (tests/test_learn_gpu.py)."""
import ctypes as C

import os

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from .. import _lib as L

_ws = {}
_ws_retired = []   # outgrown workspaces are kept alive: a captured hipGraph may still hold their address


_lanes = {}        # raw stream handle -> lane name (register_lane): kernels on a branch stream get their OWN scratch buffers


def register_lane(stream, name):
    """The optimizer step forks independent networks onto side streams (IMAmpAgent._fwd_bwd): launches issued while `stream` is current
    -- the forward under `torch.cuda.stream(stream)`, the backward because autograd runs a node on its forward's stream -- take their
    workspaces under `name`, so that two branches never share a reduction scratch buffer."""
    _lanes[stream.cuda_stream] = name


def _workspace(key, nbytes, device, dtype):
    n = (nbytes + dtype.itemsize - 1) // dtype.itemsize
    if _lanes and torch.device(device).type == "cuda":
        lane = _lanes.get(torch.cuda.current_stream(device).cuda_stream)
        if lane is not None:
            key = (key, lane)
    t = _ws.get((key, device))
    if t is None or t.numel() < n:
        if t is not None:
            _ws_retired.append(t)
        t = torch.empty(max(n, 1), dtype=dtype, device=device)   # (long-lived: which stream's pool it comes from does not matter)
        _ws[(key, device)] = t
    return t


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _first_write(p):
    """True when parameter `p` lives in a FlatGradBucket and nothing has been written into its `.grad` since the bucket was last
    zeroed: the layer's backward may then store its result there directly and return None instead of handing autograd a tensor to add
    (one launch per parameter and step saved); later contributions in the same step take the normal accumulate path."""
    b = getattr(p, "_bucket", None)
    if b is None or p.grad is None:
        return False
    if getattr(p, "_grad_gen", -1) == b.gen:
        # a SECOND contribution to this gradient inside one pass (a module applied twice): the first one may still be a pending column-sum
        # job whose finishing launch STORES -- finish it now, so that the store lands before this contribution is added
        if _pending and p.grad.data_ptr() in _pending_dst:
            flush_colsums(p.grad.device)
        return False
    p._grad_gen = b.gen
    return True


# ---- deferred second stage of the column sums (round 5) ---------------------------------------------------------------------------------
# Inside `deferred_colsums()` a column sum that goes straight into a bucket gradient (`out` given) only runs its first stage; the per-chunk
# partials wait in a workspace of their own (keyed by the destination) and ONE `phc_colsum_finish_batch` launch at the end of the pass finishes
# all of them -- the agent's passes end with it (IMAmpAgent._policy_pass / _disc_pass / _fwd_bwd).  Nobody reads a bias gradient before clip + Adam.
# Module-level state, like `_INPUT_GRAD_ONLY`: the backward runs on the autograd engine's thread; the pending list is per lane (stream).
_DEFER = [0]
_pending = {}
_pending_dst = set()    # data pointers of the gradients (and gradient blocks) a pending job will write


def _lane_of(device):
    return _lanes.get(torch.cuda.current_stream(device).cuda_stream) if _lanes else None


class deferred_colsums:
    def __enter__(self):
        _DEFER[0] += 1
        return self

    def __exit__(self, et, ev, tb):
        _DEFER[0] -= 1
        if et is None:
            flush_colsums()
        else:
            _pending.clear()
            _pending_dst.clear()


def _defer(out):
    return _DEFER[0] > 0 and out is not None and out.is_cuda and not os.environ.get("PHC_NO_DEFER_COLSUM")


def _pend(ws, out, nchunks, cols, accumulate=0):
    if out.data_ptr() in _pending_dst:   # (two stores into one destination in one batch would race: finish the first)
        flush_colsums(out.device)
    _pending_dst.add(out.data_ptr())
    _pending.setdefault((out.device, _lane_of(out.device)), []).append((ws, out, int(nchunks), int(cols), int(accumulate)))


def flush_colsums(device=None):
    """Finish every pending column sum of the current lane (stream) in one launch per 16 of them."""
    if not _pending:
        return
    for key in list(_pending):
        dev, lane = key
        if (device is not None and torch.device(device) != dev) or lane != _lane_of(dev):
            continue
        jobs = _pending.pop(key)
        for j in jobs:
            _pending_dst.discard(j[1].data_ptr())
        arr = (L.ColsumJob * len(jobs))()
        for i, (ws, out, nch, cols, acc) in enumerate(jobs):
            arr[i].partial, arr[i].out, arr[i].nchunks, arr[i].cols, arr[i].accumulate = ws.data_ptr(), out.data_ptr(), nch, cols, acc
        L.check(L.load().phc_colsum_finish_batch(len(jobs), arr, _stream(dev)), "phc_colsum_finish_batch")


def colsum_bf16(x, out=None):
    """x bf16 [rows, cols] (contiguous, device) -> fp32 [cols] (written into `out` when given)"""
    lib = L.load()
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    else:
        assert out.numel() == cols and out.dtype == torch.float32 and out.is_contiguous()
        if _defer(out):
            ws = _workspace(("colsum", out.data_ptr()), lib.phc_colsum_workspace(rows, cols), x.device, torch.float32)
            L.check(lib.phc_colsum_bf16(x.data_ptr(), rows, cols, None, ws.data_ptr(), _stream(x.device)), "phc_colsum_bf16")
            _pend(ws, out, lib.phc_colsum_chunks(rows), cols)
            return out
    ws = _workspace("colsum", lib.phc_colsum_workspace(rows, cols), x.device, torch.float32)
    L.check(lib.phc_colsum_bf16(x.data_ptr(), rows, cols, out.data_ptr(), ws.data_ptr(), _stream(x.device)), "phc_colsum_bf16")
    return out


def colsum_relu_bf16(gy, y, out=None):
    """(gy, y) bf16 [rows, cols] -> (gm = gy masked by y > 0, bf16 [rows, cols]; column sums of gm, fp32 [cols], into `out` when given)"""
    lib = L.load()
    rows, cols = gy.shape
    gm = torch.empty_like(gy)
    if _defer(out):
        assert out.numel() == cols and out.dtype == torch.float32 and out.is_contiguous()
        ws = _workspace(("colsum", out.data_ptr()), lib.phc_colsum_workspace(rows, cols), gy.device, torch.float32)
        L.check(lib.phc_colsum_relu_bf16(gy.data_ptr(), y.data_ptr(), rows, cols, gm.data_ptr(), None, ws.data_ptr(), _stream(gy.device)), "phc_colsum_relu_bf16")
        _pend(ws, out, lib.phc_colsum_chunks(rows), cols)
        return gm, out
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=gy.device)
    ws = _workspace("colsum", lib.phc_colsum_workspace(rows, cols), gy.device, torch.float32)
    L.check(lib.phc_colsum_relu_bf16(gy.data_ptr(), y.data_ptr(), rows, cols, gm.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream(gy.device)),
            "phc_colsum_relu_bf16")
    return gm, out


SPLIT_K = 8


# K-padded first-layer weights (FlatGradBucket): `weight` is the strided [N, K] view of a stored [N, Kp] matrix, its bf16 shadow is the
# padded matrix, and the layer input may arrive K-padded (the normaliser wrote it into a [rows, Kp] buffer whose pad columns are zero).
def _match_cols(wb, xb):
    """Bring a (maybe padded) bf16 weight [N, K or Kp] and a (maybe padded) input [M, K or Kp] to the same K."""
    if wb.shape[1] > xb.shape[1]:
        wb = wb[:, :xb.shape[1]]
    elif wb.shape[1] < xb.shape[1]:
        xb = xb[:, :wb.shape[1]]
    return wb, xb


def _pad_like(gx, x):
    """An input gradient computed against the UNPADDED weight (outside FlatGradBucket.shadow_scope() the bf16 copy of a K-padded first layer
    has K columns) brought to the K-padded input's width: the pad columns of the input are constants (zero), their gradient is zero."""
    if gx is None or gx.dim() != 2 or x.dim() != 2 or gx.shape[1] >= x.shape[1]:
        return gx
    return torch.nn.functional.pad(gx, (0, x.shape[1] - gx.shape[1]))


def _zero_rows(t, r0):
    """PHC_DEBUG_ZERO_FILL=1: rows [0, r0) of a gradient tensor that `input_grad_only(row_start)` leaves unwritten are zeroed, so that anomaly
    detection / NaN checks over whole tensors do not trip over uninitialised memory (off by default: a memset per layer and step)."""
    if r0 > 0 and os.environ.get("PHC_DEBUG_ZERO_FILL"):
        t[:r0].zero_()
    return t


def _logical(g, weight):
    """A weight gradient computed against a padded input, cut to the parameter's own shape."""
    return g if g is None or g.shape[1] == weight.shape[1] else g[:, :weight.shape[1]]


def _grad_out(weight, cols):
    """The tensor a first-write weight gradient of `cols` columns goes to: the parameter's gradient, or its padded storage."""
    gp = getattr(weight, "_grad_padded", None)
    return gp if (gp is not None and gp.shape[1] == cols) else weight.grad


def wgrad_split_k(gy, x, out=None, accumulate=False):
    """gy^T x for gy [B, N], x [B, K] bf16 -> fp32 [N, K] (into `out` when given, added to it with `accumulate`); the batch is cut into
    SPLIT_K chunks that run as one batched GEMM, whose slabs `phc_sum_slabs_bf16` sums (and accumulates) in one pass."""
    B, N = gy.shape
    K = x.shape[1]
    if B % SPLIT_K == 0 and B >= 2048 and N >= 16:   # (a 1-row batched GEMM -- the value head -- stalls the host for 11 ms in hipBLASLt)
        part = torch.bmm(gy.view(SPLIT_K, B // SPLIT_K, N).transpose(1, 2), x.view(SPLIT_K, B // SPLIT_K, K))
        if out is not None:
            if (part.dtype == torch.bfloat16 and part.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous() and out.data_ptr() % 16 == 0
                    and part.data_ptr() % 16 == 0):
                L.check(L.load().phc_sum_slabs_bf16(part.data_ptr(), SPLIT_K, N * K, out.data_ptr(), int(accumulate), _stream(out.device)), "phc_sum_slabs_bf16")
                return out
            if accumulate:
                return out.add_(part.sum(0, dtype=torch.float32))
            return torch.sum(part, 0, dtype=torch.float32, out=out)
        return part.sum(0, dtype=torch.float32)
    if out is not None:
        return out.add_(gy.t() @ x) if accumulate else out.copy_(gy.t() @ x)
    return (gy.t() @ x).float()


def _wgrad_into(weight, gy, xb):
    """The weight gradient gy^T xb of a bucket parameter, put where it belongs: STORED into the parameter's gradient when that holds nothing
    of this step yet, ADDED to it otherwise (one kernel, no AccumulateGrad launch) -> None; parameters outside a FlatGradBucket get the tensor
    back for autograd to accumulate."""
    b = getattr(weight, "_bucket", None)
    if b is None or weight.grad is None or gy.dtype != torch.bfloat16:
        return _logical(wgrad_split_k(gy, xb), weight)
    out = _grad_out(weight, xb.shape[1])
    if out.shape[1] != xb.shape[1] or not out.is_contiguous():
        return _logical(wgrad_split_k(gy, xb), weight)
    wgrad_split_k(gy, xb, out=out, accumulate=not _first_write(weight))
    return None


class _LinearFn(torch.autograd.Function):
    """`relu=True`: the layer AND the ReLU that follows it (round 2): hipBLASLt's epilogue applies it (`torch._addmm_activation`:
    bit-identical to addmm + relu, 36 -> 40 us instead of 52 for 16384 x 934 x 1024), the backward masks the incoming gradient with the
    saved output before the weight / bias / input gradients are formed."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        with torch.autocast("cuda", enabled=False):
            xb = x.to(torch.bfloat16)
            # inside FlatGradBucket.shadow_scope() the optimizer kernel keeps a bf16 copy of every parameter up to date
            live = getattr(weight, "_shadow_live", None)
            if live is not None and live[0]:
                wb, bb = weight._bf16_shadow, bias._bf16_shadow
            else:
                wb, bb = weight.to(torch.bfloat16), bias.to(torch.bfloat16)
            wb, xb = _match_cols(wb, xb)
            y = torch._addmm_activation(bb, xb, wb.t()) if relu else torch.addmm(bb, xb, wb.t())
        if relu:
            ctx.save_for_backward(xb, wb, y)
        else:
            ctx.save_for_backward(xb, wb)
        ctx.x_dtype, ctx.params, ctx.relu = x.dtype, (weight, bias), relu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        if ctx.relu:
            xb, wb, y = ctx.saved_tensors
            weight, bias = ctx.params
            gy = gy.contiguous()
            if ctx.needs_input_grad[2] and gy.dtype == torch.bfloat16:
                # ReLU mask and bias gradient in ONE pass over the output gradient (phc_colsum_relu_bf16)
                direct = _first_write(bias)
                gm, gb = colsum_relu_bf16(gy, y, out=bias.grad if direct else None)
                gx, gw, _ = _LinearFn._grads(ctx, gm, xb, wb, skip_bias=True)
                return gx, gw, (None if direct else gb), None
            gy = torch.ops.aten.threshold_backward(gy, y, 0.0)
            gx, gw, gb = _LinearFn._grads(ctx, gy, xb, wb)
            return gx, gw, gb, None
        xb, wb = ctx.saved_tensors
        return _LinearFn._grads(ctx, gy.contiguous(), xb, wb) + ((None,) if len(ctx.needs_input_grad) > 3 else ())

    @staticmethod
    def _grads(ctx, gy, xb, wb, skip_bias=False):
        weight, bias = ctx.params
        gx = (gy @ wb).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = _wgrad_into(weight, gy, xb)
        if ctx.needs_input_grad[2] and not skip_bias:
            if _first_write(bias):
                colsum_bf16(gy, out=bias.grad)
            else:
                gb = colsum_bf16(gy)
        return gx, gw, gb


def _bf16_params(weight, bias):
    live = getattr(weight, "_shadow_live", None)
    if live is not None and live[0]:
        return weight._bf16_shadow, bias._bf16_shadow
    return weight.to(torch.bfloat16), bias.to(torch.bfloat16)


_INPUT_GRAD_ONLY = [False, 0]   # [active, first row of the cotangent's non-zero block]


class input_grad_only:
    """Context for a `torch.autograd.grad(outputs, inputs=<activations>)` call (the gradient penalty): tells the twice-differentiable
    layers that no parameter gradient is asked for -- a custom Function cannot see which of its gradients the engine needs, and
    computing the weight / bias gradients there cost 10 ms per update.  (Module-level flag: the backward runs on the autograd thread.)
    `row_start` (round 2): the cotangent is zero in rows [0, row_start) -- the penalty differentiates the DEMO logits only, which are
    the last third of the discriminator's [agent; replay; demo] batch -- and only rows [row_start, n) of the input gradient are read by
    the caller: the layers then work on that row block alone, here and in the second-order pass (GEMMs over m instead of 3m rows); the
    other rows of the full-size gradient tensors they hand to autograd are left UNWRITTEN."""

    def __init__(self, row_start=0):
        self.row_start = int(row_start)

    def __enter__(self):
        _INPUT_GRAD_ONLY[0], _INPUT_GRAD_ONLY[1] = True, self.row_start

    def __exit__(self, *a):
        _INPUT_GRAD_ONLY[0], _INPUT_GRAD_ONLY[1] = False, 0


_PARAM_GRAD_ONLY = [False]


class param_grad_only:
    """Context for the `backward()` of the total loss: nobody reads the gradient w.r.t. the network INPUT (the discriminator's demo
    rows are a leaf only for the penalty's `autograd.grad`), so the first twice-differentiable layer skips its input-gradient GEMM
    (12288 x 1024 x 1960 per step)."""

    def __enter__(self):
        _PARAM_GRAD_ONLY[0] = True

    def __exit__(self, *a):
        _PARAM_GRAD_ONLY[0] = False


_placeholders = {}


def _placeholder(like):
    """A constant zero-dim tensor for Function outputs nobody reads (a fresh alias per call: no fill launch)."""
    key = (like.dtype, like.device)
    if key not in _placeholders:
        _placeholders[key] = torch.zeros((), dtype=like.dtype, device=like.device)
    return _placeholders[key].detach()


def _relu_mask(g, y, out=None):
    """g where y > 0 else 0 (the ReLU's backward), optionally into `out`."""
    if out is None:
        return torch.ops.aten.threshold_backward(g, y, 0.0)
    return torch.ops.aten.threshold_backward.grad_input(g, y, 0.0, grad_input=out)


class _LinearDDFn(torch.autograd.Function):
    """The same layer for the discriminator, whose gradient penalty differentiates the backward pass (create_graph=True): the
    backward is itself an autograd node (`_LinearDDBwdFn`) with an explicit second-order rule, so both the first- and the
    second-order weight gradients -- four reductions over the 12 288-row batch per step -- run as split-K batched GEMMs.
    `relu=True` (round 2): the ReLU behind the layer rides along -- epilogue in the forward, its mask m = [y > 0] inside the backward
    node (piecewise constant: the second-order rule only gains `d gy = m * (...)` and uses the masked gradient everywhere else)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        wb, bb = _bf16_params(weight, bias)
        wb, xb = _match_cols(wb, x.to(torch.bfloat16))
        y = torch._addmm_activation(bb, xb, wb.t()) if relu else torch.addmm(bb, xb, wb.t())
        ctx.relu = relu
        ctx.x_is_net_input = type(x.grad_fn).__name__.startswith("_RowsWithGradFn")
        if relu:
            ctx.save_for_backward(x, weight, bias, y)
        else:
            ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.relu:
            x, weight, bias, y = ctx.saved_tensors
        else:
            (x, weight, bias), y = ctx.saved_tensors, None
        only_x, r0 = _INPUT_GRAD_ONLY
        need_gx = ctx.needs_input_grad[0] and not (_PARAM_GRAD_ONLY[0] and ctx.x_is_net_input)
        gx, gw, gb = _LinearDDBwdFn.apply(gy, x, weight, bias, need_gx, only_x, y, r0 if only_x else 0)
        out = (gx if need_gx else None, None if (only_x or gw.dim() != 2) else gw, None if (only_x or gb.dim() != 1) else gb)   # (0-dim gw / gb: written in place)
        return out + ((None,) if len(ctx.needs_input_grad) > 3 else ())


class _LinearDDBwdFn(torch.autograd.Function):
    """(gy, x, W[, y]) -> gz = gy (masked by y > 0 when the layer carries its ReLU), gx = gz W, gW = gz^T x, gb = 1^T gz; its own
    backward for cotangents (ggx, ggW, ggb):  d gz = ggx W^T + x ggW^T + ggb,  d gy = mask * d gz,  d x = gz ggW,  d W = gz^T ggx.
    `only_x` (inside input_grad_only): gx alone, over the rows [r0, n) -- see input_grad_only."""

    @staticmethod
    def forward(ctx, gy, x, weight, bias, need_gx, only_x, y=None, r0=0):
        gy = gy.contiguous()
        wb, _ = _bf16_params(weight, bias)
        wb, _ = _match_cols(wb, x)          # (x may be K-padded; its gradient then is, too)
        ctx.w_cols, ctx.weight = weight.shape[1], weight
        ctx.x_dtype, ctx.need_gx, ctx.only_x, ctx.masked, ctx.r0, ctx.rows = x.dtype, need_gx, only_x, y is not None, r0, gy.shape[0]
        ctx.set_materialize_grads(False)   # cotangents nobody produced arrive as None, not as zeros
        if only_x:
            gz = gy[r0:]
            if y is not None:
                y = y[r0:]
                gz = _relu_mask(gz, y)
            ctx.save_for_backward(gz, wb, *([y] if y is not None else []))
            if need_gx:
                gx = _zero_rows(torch.empty((gy.shape[0], wb.shape[1]), dtype=torch.bfloat16, device=gy.device), r0)
                torch.mm(gz, wb, out=gx[r0:])
                gx = _pad_like(gx.to(x.dtype), x)
            else:
                gx = _placeholder(gy)
            pw, pb = _placeholder(gy), _placeholder(gy)
            ctx.mark_non_differentiable(pw, pb)
            return gx, pw, pb
        _, xb = _match_cols(wb, x.to(torch.bfloat16))
        gb = None
        # the bias receives ONE contribution per step (the penalty path asks for no parameter gradient, the second-order rule has no bias term): when it is
        # the first write into a bucket gradient the column sums are stored there directly -- no tensor for autograd to add (one launch per layer, round 5)
        direct_b = (gy.dtype == torch.bfloat16 and bias.grad is not None and bias.grad.is_contiguous() and not os.environ.get("PHC_NO_DIRECT_DD_BIAS")
                    and _first_write(bias))
        if y is not None:
            if gy.dtype != torch.bfloat16:
                gy = _relu_mask(gy, y)
            else:
                gy, gb = colsum_relu_bf16(gy, y, out=bias.grad if direct_b else None)          # mask + bias gradient in one pass
            ctx.save_for_backward(gy, wb, xb, y)
        else:
            ctx.save_for_backward(gy, wb, xb)
        gx = _pad_like((gy @ wb).to(x.dtype), x) if need_gx else _placeholder(gy)
        gw = _wgrad_into(weight, gy, xb)
        if gw is None:        # stored / added in place
            gw = _placeholder(gy)
            ctx.mark_non_differentiable(gw)
        if gb is None:
            gb = colsum_bf16(gy, out=bias.grad if direct_b else None)
        if direct_b:
            gb = _placeholder(gy)
            ctx.mark_non_differentiable(gb)
        return gx, gw, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx, ggw, ggb):
        saved = ctx.saved_tensors
        gy, wb = saved[0], saved[1]
        y = saved[-1] if ctx.masked else None
        r0 = ctx.r0
        d_gy = d_x = d_w = None
        if ctx.only_x:   # the weight / bias outputs were placeholders; gy, y are the row block [r0, n)
            if ggx is not None and ctx.need_gx:
                ggx = ggx[r0:].to(torch.bfloat16).contiguous()
                d_w = _wgrad_into(ctx.weight, gy, ggx)
                d_gy = _zero_rows(torch.empty((ctx.rows, wb.shape[0]), dtype=torch.bfloat16, device=gy.device), r0)   # rows [0, r0) stay unwritten (PHC_DEBUG_ZERO_FILL)
                if y is not None:
                    _relu_mask(ggx @ wb.t(), y, out=d_gy[r0:])
                else:
                    torch.mm(ggx, wb.t(), out=d_gy[r0:])
            return d_gy, None, d_w, None, None, None, None, None
        xb = saved[2]
        if ggx is not None and ctx.need_gx:
            ggx = ggx.to(torch.bfloat16).contiguous()
            d_gy = ggx @ wb.t()
            d_w = _wgrad_into(ctx.weight, gy, ggx)
        if ggw is not None:
            gwb = ggw.to(torch.bfloat16)
            t = xb @ gwb.t()
            d_gy = t if d_gy is None else d_gy + t
            d_x = (gy @ gwb).to(ctx.x_dtype)
        if ggb is not None:
            t = ggb.to(torch.bfloat16).expand_as(gy)
            d_gy = t if d_gy is None else d_gy + t
        if y is not None and d_gy is not None:
            d_gy = _relu_mask(d_gy.contiguous(), y)
        return d_gy, d_x, d_w, None, None, None, None, None


def _device_training_pass(mod, x):
    return (x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and mod.weight.requires_grad and mod.bias is not None
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.is_contiguous())


class FastLinearDD(nn.Linear):
    """nn.Linear for layers that are differentiated twice (the discriminator MLP): see _LinearDDFn."""
    fuse_relu = False
    _fused_now = False

    def forward(self, x):
        self._fused_now = False
        if _device_training_pass(self, x):
            self._fused_now = self.fuse_relu
            return _LinearDDFn.apply(x, self.weight, self.bias, True) if self.fuse_relu else _LinearDDFn.apply(x, self.weight, self.bias)
        return nn.functional.linear(x[..., :self.in_features] if x.shape[-1] > self.in_features else x, self.weight, self.bias)   # (x may be K-padded)


def _linear1_forward(xb, wb, bb):
    lib = L.load()
    rows, cols = xb.shape
    y = torch.empty((rows, 1), dtype=torch.bfloat16, device=xb.device)
    L.check(lib.phc_linear1_forward(xb.data_ptr(), wb.data_ptr(), bb.data_ptr(), rows, cols, y.data_ptr(), _stream(xb.device)), "phc_linear1_forward")
    return y


class _Linear1Fn(torch.autograd.Function):
    """FastLinear with a single output (the value head): dot product per row, scaled copy, weighted column sum (phc_linear1_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xb = x.to(torch.bfloat16)
        wb, bb = _bf16_params(weight, bias)
        ctx.save_for_backward(xb, wb)
        ctx.x_dtype, ctx.params = x.dtype, (weight, bias)
        return _linear1_forward(xb, wb, bb)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xb, wb = ctx.saved_tensors
        lib = L.load()
        rows, cols = xb.shape
        gy = gy.contiguous()
        gx = torch.empty_like(xb) if ctx.needs_input_grad[0] else None
        weight, bias = ctx.params
        # weight [1, cols] and bias [1] sit next to each other in the flat gradient buffer: the kernel's [cols + 1] result goes there
        b = getattr(weight, "_bucket", None)
        direct = (b is not None and weight.grad is not None and bias.grad is not None and bias.grad.data_ptr() == weight.grad.data_ptr() + 4 * cols
                  and getattr(weight, "_grad_gen", -1) != b.gen and getattr(bias, "_grad_gen", -1) != b.gen)
        if direct:
            weight._grad_gen = bias._grad_gen = b.gen
        elif _pending and weight.grad is not None and weight.grad.data_ptr() in _pending_dst:
            flush_colsums(weight.grad.device)   # (a second application of the layer in one pass: the first one's pending STORE lands before this one is added)
        gwb = None if direct else torch.empty(cols + 1, dtype=torch.float32, device=xb.device)
        if direct and _defer(weight.grad):   # first stage only: the [cols + 1] result is finished with the pass's other column sums
            ws = _workspace(("lin1", weight.grad.data_ptr()), lib.phc_linear1_workspace(rows, cols), xb.device, torch.float32)
            L.check(lib.phc_linear1_backward(xb.data_ptr(), wb.data_ptr(), gy.data_ptr(), rows, cols, None if gx is None else gx.data_ptr(), None, ws.data_ptr(),
                                             _stream(xb.device)), "phc_linear1_backward")
            _pend(ws, weight.grad, lib.phc_linear1_chunks(rows), cols + 1)
            gx = gx.to(ctx.x_dtype) if gx is not None else None
            return gx, None, None
        ws = _workspace("lin1", lib.phc_linear1_workspace(rows, cols), xb.device, torch.float32)
        L.check(lib.phc_linear1_backward(xb.data_ptr(), wb.data_ptr(), gy.data_ptr(), rows, cols, None if gx is None else gx.data_ptr(),
                                         weight.grad.data_ptr() if direct else gwb.data_ptr(), ws.data_ptr(), _stream(xb.device)), "phc_linear1_backward")
        gx = gx.to(ctx.x_dtype) if gx is not None else None
        return (gx, None, None) if direct else (gx, gwb[:cols].view(1, cols), gwb[cols:])


class _Linear1DDFn(torch.autograd.Function):
    """The discriminator's logit layer (one output, differentiated twice by the gradient penalty) on the phc_linear1_* kernels:
    hipBLASLt ran this 512 -> 1 layer as five skinny GEMMs per step (~175 us: 12288 x 512 x 1 and its transposes)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        wb, bb = _bf16_params(weight, bias)
        ctx.save_for_backward(x, weight, bias)
        return _linear1_forward(x.to(torch.bfloat16), wb, bb)

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias = ctx.saved_tensors
        only_x, r0 = _INPUT_GRAD_ONLY
        gx, gw, gb = _Linear1DDBwdFn.apply(gy, x, weight, bias, only_x, r0 if only_x else 0)
        return gx if ctx.needs_input_grad[0] else None, None if only_x else gw, None if only_x else gb


class _Linear1DDBwdFn(torch.autograd.Function):
    """(gy [n, 1], x [n, K], w [1, K]) -> gx = gy w, gw = gy^T x, gb = 1^T gy; second order (the penalty path only: cotangent ggx on
    gx):  d w = gy^T ggx,  d gy = ggx w^T."""

    @staticmethod
    def forward(ctx, gy, x, weight, bias, only_x, r0=0):
        lib = L.load()
        gy = gy.to(torch.bfloat16).contiguous()
        wb, _ = _bf16_params(weight, bias)
        rows, cols = x.shape
        ctx.only_x, ctx.r0, ctx.rows, ctx.x_dtype = only_x, r0, rows, x.dtype
        ctx.set_materialize_grads(False)
        if only_x:
            gys = gy[r0:]
            ctx.save_for_backward(gys, wb)
            gx = _zero_rows(torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device), r0)   # rows [0, r0) stay unwritten (PHC_DEBUG_ZERO_FILL)
            torch.mul(gys, wb, out=gx[r0:])
            pw, pb = _placeholder(gy), _placeholder(gy)
            ctx.mark_non_differentiable(pw, pb)
            return gx.to(x.dtype), pw, pb
        ctx.save_for_backward(gy, wb)
        xb = x.to(torch.bfloat16)
        gx = torch.empty_like(xb)
        gwb = torch.empty(cols + 1, dtype=torch.float32, device=x.device)
        ws = _workspace("lin1", lib.phc_linear1_workspace(rows, cols), x.device, torch.float32)
        L.check(lib.phc_linear1_backward(xb.data_ptr(), wb.data_ptr(), gy.data_ptr(), rows, cols, gx.data_ptr(), gwb.data_ptr(), ws.data_ptr(),
                                         _stream(x.device)), "phc_linear1_backward")
        return gx.to(x.dtype), gwb[:cols].view(1, cols), gwb[cols:]

    @staticmethod
    @once_differentiable
    def backward(ctx, ggx, ggw, ggb):
        if ggw is not None or ggb is not None:
            raise NotImplementedError("second-order rule through the logit layer's weight / bias gradient")
        if ggx is None:
            return (None,) * 6
        lib = L.load()
        gy, wb = ctx.saved_tensors       # (only_x: the row block [r0, n))
        g = ggx[ctx.r0:].to(torch.bfloat16).contiguous()
        m, cols = g.shape
        gwb = torch.empty(cols + 1, dtype=torch.float32, device=g.device)
        ws = _workspace("lin1", lib.phc_linear1_workspace(m, cols), g.device, torch.float32)
        L.check(lib.phc_linear1_backward(g.data_ptr(), wb.data_ptr(), gy.data_ptr(), m, cols, None, gwb.data_ptr(), ws.data_ptr(), _stream(g.device)),
                "phc_linear1_backward")
        d_gy = None
        if ctx.needs_input_grad[0]:
            d_gy = torch.zeros((ctx.rows, 1), dtype=torch.bfloat16, device=g.device)
            d_gy[ctx.r0:] = (g.float() * wb.float()).sum(-1, keepdim=True).to(torch.bfloat16)
        return d_gy, None, gwb[:cols].view(1, cols), None, None, None


class FastLinear1DD(nn.Linear):
    """nn.Linear(K, 1) that is differentiated twice (the discriminator's `_disc_logits`): see _Linear1DDFn."""

    def forward(self, x):
        if self.out_features == 1 and _device_training_pass(self, x) and x.dtype == torch.bfloat16:
            return _Linear1DDFn.apply(x, self.weight, self.bias)
        return nn.functional.linear(x[..., :self.in_features] if x.shape[-1] > self.in_features else x, self.weight, self.bias)   # (x may be K-padded)


# ---- split-bf16 layers (round 6, `+learning.params.config.actor_precision=split_bf16`) --------------------------------------------------------------------------
# A precision option between bf16 GEMMs and the 4.8 x slower fp32 ones for the ACTOR: every operand is cut into a bf16 head and a bf16 tail (x = xh + xl with
# xl = bf16(x - xh): 16 mantissa bits together), a product is (ah + al)(bh + bl) ~ ah bh + al bh + ah bl (the tail x tail term, 2^-18, is dropped) and the activations stay
# fp32 between the layers.  The three partial products of one product are ONE MFMA GEMM with a three times longer reduction -- chunks (ah, al, ah) against (bh, bh, bl) --
# with an fp32 result (`torch.mm(..., out_dtype=torch.float32)`, aten::mm.dtype).  `phc_split3_bf16` writes an operand's three chunks in one pass (reads the fp32 tensor
# once, applies the ReLU mask of a backward pass on the way, pads the reduction length to a multiple of 32, carries the ones / bias column that makes the bias part of the
# product and the bias gradient a column of the weight gradient); the SAME [rows, 3, cols] buffer is the [rows, 3 cols] operand
# of the forward / input-gradient product and the [3 rows, cols] operand of the weight gradient (reduction over rows AND chunks), so every tensor is split once.
def _pad32(n):
    return (n + 31) // 32 * 32


def _split3(x, order, gate=None, rows_pad=None, cols_pad=None, chunk_major=False, extra=None, ones=False):
    """x fp32 [R, C] (unit column stride) -> bf16 [Rp, 3, Cp] (chunks (h, h, l) for order 0, (h, l, h) for order 1), or [3, Rp, Cp] with `chunk_major`;
    `ones`: column C of the valid rows is 1, `extra` (fp32 [R]): column C holds it."""
    R, C = x.shape
    Rp, Cp = rows_pad or R, cols_pad or _pad32(C)
    assert x.dtype == torch.float32 and x.stride(1) == 1 and (gate is None or (gate.dtype == torch.float32 and gate.stride(1) == 1 and gate.shape == x.shape))
    assert extra is None or (extra.dtype == torch.float32 and extra.is_contiguous() and extra.numel() == R)
    out = torch.empty((3, Rp, Cp) if chunk_major else (Rp, 3, Cp), dtype=torch.bfloat16, device=x.device)
    row_stride, chunk_stride = (Cp, Rp * Cp) if chunk_major else (3 * Cp, Cp)
    L.check(L.load().phc_split3_bf16(x.data_ptr(), x.stride(0), gate.data_ptr() if gate is not None else None, gate.stride(0) if gate is not None else 0, R, C, Rp, Cp,
                                     extra.data_ptr() if extra is not None else None, 2 if extra is not None else int(bool(ones)),
                                     out.data_ptr(), row_stride, chunk_stride, order, _stream(x.device)), "phc_split3_bf16")
    return out


def _split_forward(x, weight, bias, relu):
    """y = x W^T + b with split operands: xc [B, 3, Kp] order 1 (with the ones column) against wc [Np, 3, Kp] order 0 (with the bias column) -> fp32 [B, N]."""
    N, K = weight.shape
    Np, Kp = _pad32(N), _pad32(K + 1)
    xc = _split3(x, 1, cols_pad=Kp, ones=True)
    wc = _split3(weight, 0, rows_pad=Np, cols_pad=Kp, extra=bias)
    y = torch.mm(xc.view(-1, 3 * Kp), wc.view(Np, 3 * Kp).t(), out_dtype=torch.float32)
    if relu:
        torch.relu_(y)
    return (y if Np == N else y[:, :N].contiguous()), xc


class _SplitLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        with torch.autocast("cuda", enabled=False):
            w = weight.detach()
            y, xc = _split_forward(x, w, bias.detach().float().contiguous(), relu)
        ctx.save_for_backward(xc, w, *((y,) if relu else ()))
        ctx.relu = relu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, w = ctx.saved_tensors[:2]
        N, K = w.shape
        Np, Kp = _pad32(N), xc.shape[2]
        B = xc.shape[0]
        with torch.autocast("cuda", enabled=False):
            gy = gy.float()
            if gy.stride(1) != 1:
                gy = gy.contiguous()
            gc = _split3(gy, 0, gate=ctx.saved_tensors[2] if ctx.relu else None, cols_pad=Np)           # [B, 3, Np] chunks (gh, gh, gl): the ReLU mask applied on the way
            gx = gw = gb = None
            if ctx.needs_input_grad[0]:
                wr = _split3(w, 1, rows_pad=Np, cols_pad=K if K % 4 == 0 else _pad32(K), chunk_major=True)   # [3, Np, K] chunks (wh, wl, wh)
                gx = torch.mm(gc.view(B, 3 * Np), wr.view(3 * Np, -1), out_dtype=torch.float32)
                if gx.shape[1] != K:
                    gx = gx[:, :K]
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                # dW = gh^T xh + gh^T xl + gl^T xh: one reduction over rows AND chunks = [3 B, Np]^T [3 B, Kp], cut into SPLIT_K slabs like the bf16 layers' (wgrad_split_k);
                # column K of it -- the gradient against the ones column -- is the bias gradient
                rows = 3 * B
                if rows % SPLIT_K == 0 and rows >= 2048:
                    g = torch.bmm(gc.view(SPLIT_K, rows // SPLIT_K, Np).transpose(1, 2), xc.view(SPLIT_K, rows // SPLIT_K, Kp), out_dtype=torch.float32).sum(0)
                else:
                    g = torch.mm(gc.view(rows, Np).t(), xc.view(rows, Kp), out_dtype=torch.float32)
                gw, gb = g[:N, :K], g[:N, K]
        return gx, gw, gb, None


class FastLinear(nn.Linear):
    fuse_relu = False      # set by network.build_mlp when a ReLU follows: the device passes apply it in the GEMM epilogue
    _fused_now = False     # did the last forward() apply it?  (read by the FusedReLU module that follows in the nn.Sequential)
    split_precision = False   # IMAmpAgent sets it on the actor's layers for `actor_precision=split_bf16`: fp32 activations in, fp32 out, three bf16 GEMMs per product

    def forward(self, x):
        self._fused_now = False
        if self.split_precision and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
            self._fused_now = self.fuse_relu
            if torch.is_grad_enabled() and self.weight.requires_grad:
                return _SplitLinearFn.apply(x[:, :self.in_features] if x.shape[1] > self.in_features else x, self.weight, self.bias, self.fuse_relu)
            with torch.autocast("cuda", enabled=False):
                return _split_forward(x[:, :self.in_features] if x.shape[1] > self.in_features else x, self.weight.detach(), self.bias.detach().float().contiguous(), self.fuse_relu)[0]
        if self.out_features == 1 and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and x.is_contiguous() and self.bias is not None:
            if _device_training_pass(self, x):
                return _Linear1Fn.apply(x, self.weight, self.bias)
            live = getattr(self.weight, "_shadow_live", None)
            if not torch.is_grad_enabled() and live is not None and live[0]:
                return _linear1_forward(x, self.weight._bf16_shadow, self.bias._bf16_shadow)
        if (x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and self.weight.requires_grad and self.bias is not None
                and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 and x.is_contiguous()):
            self._fused_now = self.fuse_relu
            return _LinearFn.apply(x, self.weight, self.bias, True) if self.fuse_relu else _LinearFn.apply(x, self.weight, self.bias)
        live = getattr(self.weight, "_shadow_live", None)
        if live is not None and live[0] and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and self.bias is not None:
            # rollout inference inside FlatGradBucket.shadow_scope(): the bf16 parameter copies are current, no per-call casts
            wb = self.weight._bf16_shadow
            if x.dim() == 2:
                wb, x = _match_cols(wb, x)
            if self.fuse_relu and x.dim() == 2:
                self._fused_now = True
                return torch._addmm_activation(self.bias._bf16_shadow, x, wb.t())
            return nn.functional.linear(x, wb, self.bias._bf16_shadow)
        return nn.functional.linear(x[..., :self.in_features] if x.shape[-1] > self.in_features else x, self.weight, self.bias)   # (x may be K-padded)


class FusedReLU(nn.ReLU):
    """The ReLU behind a FastLinear in an nn.Sequential (same position, no parameters: state-dict keys unchanged): a no-op when the
    layer in front of it has already applied it in its GEMM epilogue."""

    def __init__(self, prev):
        super().__init__()
        self._prev = [prev]     # (a list: not registered as a sub-module)

    def forward(self, x):
        if self._prev[0]._fused_now:
            self._prev[0]._fused_now = False
            return x
        return super().forward(x)


class _PPOLossFn(torch.autograd.Function):
    """Actor + critic part of the PPO loss (`phc_ppo_loss`): forward computes the loss, its statistics AND the gradients w.r.t. the
    two network heads; backward hands those out.  The loss must enter the total with weight one (`unit_grad`): the incoming
    gradient is then 1 and the stored gradients are returned as they are (no extra pass over [B, D])."""

    @staticmethod
    def forward(ctx, mu, value, logstd, actions, old_neglogp, adv, returns, old_values, old_mu, old_sigma, prm, unit_grad, row_index, out=None):
        lib = L.load()
        B, D = mu.shape
        assert value.numel() == B and mu.dtype == value.dtype and mu.dtype in (torch.bfloat16, torch.float32)
        assert mu.is_contiguous() and value.is_contiguous()
        f32 = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        logstd, actions, old_neglogp, adv, returns, old_mu, old_sigma = map(f32, (logstd, actions, old_neglogp, adv, returns, old_mu, old_sigma))
        old_values = f32(old_values) if old_values is not None else None
        gmu, gval = torch.empty_like(mu), torch.empty_like(value)
        buf = torch.empty(7, dtype=torch.float32, device=mu.device) if out is None else out
        assert buf.numel() == 7 and buf.dtype == torch.float32 and buf.is_contiguous()
        ws = _workspace("ppo", lib.phc_ppo_loss_workspace(), mu.device, torch.float64)
        p = L.PpoParams(float(prm["e_clip"]), float(prm["critic_coef"]), float(prm["entropy_coef"]), float(prm["bounds_loss_coef"]), int(prm["clip_value"]))
        L.check(lib.phc_ppo_loss(mu.data_ptr(), value.data_ptr(), int(mu.dtype == torch.bfloat16), logstd.data_ptr(), actions.data_ptr(),
                                 old_neglogp.data_ptr(), adv.data_ptr(), returns.data_ptr(), None if old_values is None else old_values.data_ptr(),
                                 old_mu.data_ptr(), old_sigma.data_ptr(), None if row_index is None else row_index.data_ptr(), B, D, C.byref(p), gmu.data_ptr(), gval.data_ptr(), buf.data_ptr(),
                                 ws.data_ptr(), _stream(mu.device)), "phc_ppo_loss")
        ctx.save_for_backward(gmu, gval)
        ctx.unit_grad = unit_grad
        ctx.set_materialize_grads(False)
        loss, stats = buf.narrow(0, 0, 1).view(()), buf.narrow(0, 1, 6)   # disjoint views of the kernel's output
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, _g_stats):
        gmu, gval = ctx.saved_tensors
        if not ctx.unit_grad:
            gmu, gval = gmu * g_loss.to(gmu.dtype), gval * g_loss.to(gval.dtype)
        return (gmu, gval) + (None,) * 12


def ppo_loss(mu, value, logstd, actions, old_neglogp, adv, returns, old_values, old_mu, old_sigma, e_clip, critic_coef, entropy_coef,
             bounds_loss_coef, clip_value, unit_grad=False, row_index=None, out=None):
    """-> (loss, stats[6] = a_loss, c_loss, b_loss, entropy, kl, clip fraction); mu [B, D] / value [B, 1] are the (bf16 or fp32) network heads;
    with `row_index` [B] the rollout tensors (actions ... old_sigma) are the whole dataset and row r of the minibatch is row_index[r];
    `out`: fp32 [7] that receives [loss, stats] (the results are views of it)."""
    prm = dict(e_clip=e_clip, critic_coef=critic_coef, entropy_coef=entropy_coef, bounds_loss_coef=bounds_loss_coef or 0.0, clip_value=bool(clip_value))
    return _PPOLossFn.apply(mu, value, logstd, actions, old_neglogp, adv, returns, old_values if clip_value else None, old_mu, old_sigma, prm, unit_grad,
                            row_index, out)


class _TakeRowsFn(torch.autograd.Function):
    """g[start:] whose backward leaves rows [0, start) of the full-size gradient unwritten (see input_grad_only(row_start))."""

    @staticmethod
    def forward(ctx, g, start):
        ctx.start, ctx.rows = start, g.shape[0]
        return g[start:]

    @staticmethod
    @once_differentiable
    def backward(ctx, gg):
        full = _zero_rows(torch.empty((ctx.rows,) + tuple(gg.shape[1:]), dtype=gg.dtype, device=gg.device), ctx.rows - gg.shape[0])
        full[ctx.start:] = gg
        return full, None


class _RowsWithGradFn(torch.autograd.Function):
    """`buf` [n, A] already holds all rows (written in place by the normaliser); rows [start:] additionally ARE the leaf `rows_leaf`
    (same memory).  Forward is a view of `buf` -- no `torch.cat` of the discriminator's three input batches -- and backward hands the
    leaf its row block of the incoming gradient (differentiable again: the gradient penalty differentiates through it)."""

    @staticmethod
    def forward(ctx, buf, rows_leaf, start):
        ctx.start = start
        ctx.set_materialize_grads(False)
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        if g is None:     # (param_grad_only: the first layer formed no input gradient)
            return None, None, None
        if _INPUT_GRAD_ONLY[0] and _INPUT_GRAD_ONLY[1] == ctx.start and ctx.start > 0:
            return None, _TakeRowsFn.apply(g, ctx.start), None
        return None, g[ctx.start:], None


def rows_with_grad(buf, rows_leaf, start):
    return _RowsWithGradFn.apply(buf, rows_leaf, start)


class _DiscBCEFn(torch.autograd.Function):
    """0.5 (BCEWithLogits(agent rows, 0) + BCEWithLogits(demo rows, 1)) * scale, the two accuracies, and the gradient w.r.t. the
    logits in one launch (`phc_disc_bce`).  Unit-weight convention as `_PPOLossFn`: the result is added to the total loss as it is."""

    @staticmethod
    def forward(ctx, logits, n_agent, scale, out=None):
        lib = L.load()
        n = logits.shape[0]
        assert logits.is_contiguous() and logits.numel() == n and logits.dtype in (torch.bfloat16, torch.float32)
        grad = torch.empty_like(logits)
        stats = torch.empty(5, dtype=torch.float32, device=logits.device) if out is None else out
        assert stats.numel() == 5 and stats.dtype == torch.float32 and stats.is_contiguous()
        ctx.set_materialize_grads(False)
        L.check(lib.phc_disc_bce(logits.data_ptr(), int(logits.dtype == torch.bfloat16), n_agent, n - n_agent, float(scale), grad.data_ptr(), stats.data_ptr(),
                                 _stream(logits.device)), "phc_disc_bce")
        ctx.save_for_backward(grad)
        acc = stats.narrow(0, 1, 4)
        ctx.mark_non_differentiable(acc)
        return stats.narrow(0, 0, 1).view(()), acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _):
        (grad,) = ctx.saved_tensors
        return grad, None, None, None


def disc_bce(logits, n_agent, scale=1.0, out=None):
    """-> (scale * 0.5 (bce(agent, 0) + bce(demo, 1)), [agent_acc, demo_acc, mean agent logit, mean demo logit]); logits [n, 1]: agent (+ replay) rows
    first, demo rows last; `out`: fp32 [5] that receives [loss, the four statistics] (the results are views of it)."""
    return _DiscBCEFn.apply(logits, n_agent, scale, out)


def _weighted_sumsq(tensors, coefs, out=None):
    """-> fp32 [1 + n]: [sum_i coefs[i] |t_i|^2, |t_0|^2, ...] (into `out` when given)"""
    lib = L.load()
    dev = tensors[0].device
    is_bf16 = tensors[0].dtype == torch.bfloat16
    for t in tensors:
        assert t.is_contiguous() and t.device == dev and t.dtype == tensors[0].dtype and t.dtype in (torch.bfloat16, torch.float32)
    n = len(tensors)
    if out is None:
        out = torch.empty(1 + n, dtype=torch.float32, device=dev)
    assert out.numel() == 1 + n and out.dtype == torch.float32 and out.is_contiguous()
    ws = _workspace("sumsq", lib.phc_sumsq_workspace(), dev, torch.float64)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (C.c_int64 * n)(*[t.numel() for t in tensors])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    L.check(lib.phc_weighted_sumsq(n, ptrs, sizes, cf, int(is_bf16), out.data_ptr(), ws.data_ptr(), _stream(dev)), "phc_weighted_sumsq")
    return out


class _WeightedSumsqFn(torch.autograd.Function):
    """sum_i coefs[i] |t_i|^2 over up to four tensors in two launches (`phc_weighted_sumsq`); gradient 2 coefs[i] t_i.  Unit-weight
    convention: the result enters the total loss as it is.  `preloaded`: the caller has already put 2 coefs[i] t_i into the
    gradient buffers (FlatGradBucket.zero(decay=...)), backward returns nothing."""

    @staticmethod
    def forward(ctx, coefs, out, preloaded, *tensors):
        ctx.save_for_backward(*tensors)
        ctx.coefs, ctx.preloaded = [float(c) for c in coefs], preloaded
        ctx.set_materialize_grads(False)
        buf = _weighted_sumsq(tensors, coefs, out)
        total, parts = buf.narrow(0, 0, 1).view(()), buf.narrow(0, 1, len(tensors))
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _):
        ts = ctx.saved_tensors
        if ctx.preloaded:
            return (None,) * (3 + len(ts))
        return (None, None, None) + tuple(t * (2.0 * c) for t, c in zip(ts, ctx.coefs))   # (torch._foreach_mul: 49 us for these three tensors)


def weighted_sumsq(tensors, coefs, out=None, preloaded=False, parts=False):
    total, p = _WeightedSumsqFn.apply(list(coefs), out, preloaded, *tensors)
    return (total, p) if parts else total


def policy_sample(mu, value, logstd, value_norm, out_actions, out_mus, out_sigmas, out_neglogp, out_values, mask=None):
    """One rollout policy step on the device (`phc_policy_sample`): samples the action (torch.randn noise, the generator stream of
    `torch.randn_like(mu)`) and writes action / mu / sigma / neglogp / un-normalised value straight into rows of the experience
    buffer.  `mu=None`: only the value part (next_values), optionally masked.  `value_norm`: the value RunningMeanStd or None."""
    lib = L.load()
    head = mu if mu is not None else value
    N = head.shape[0]
    D = mu.shape[1] if mu is not None else 0
    ptr = lambda t: None if t is None else t.data_ptr()
    noise = torch.randn((N, D), dtype=torch.float32, device=head.device) if mu is not None else None
    for t in (out_actions, out_mus, out_sigmas, out_neglogp, out_values):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    assert (mu is None or mu.is_contiguous()) and (value is None or value.is_contiguous())
    L.check(lib.phc_policy_sample(ptr(mu), ptr(value), int(head.dtype == torch.bfloat16), ptr(logstd), ptr(noise),
                                  ptr(value_norm.running_mean) if value_norm is not None else None,
                                  ptr(value_norm.running_var) if value_norm is not None else None,
                                  float(value_norm.epsilon) if value_norm is not None else 0.0, ptr(mask), N, D, ptr(out_actions), ptr(out_mus),
                                  ptr(out_sigmas), ptr(out_neglogp), ptr(out_values), _stream(head.device)), "phc_policy_sample")


def adam_state(optimizer, flat_param):
    """The optimizer's state of the flat parameter, created (zero moments, step 0) if it does not exist yet -- callers that capture the step
    in a graph create it BEFORE the capture (tensors made inside a capture live in the graph's pool and their zero-fill would replay)."""
    st = optimizer.state[flat_param]
    if len(st) == 0:
        st["step"] = torch.tensor(0.0, dtype=torch.float32)
        st["exp_avg"] = torch.zeros_like(flat_param)
        st["exp_avg_sq"] = torch.zeros_like(flat_param)
    return st


def adam_clip_step(optimizer, flat_param, flat_grad, max_norm, shadow=None, step_device=None, count_host=True):
    """`clip_grad_norm_(max_norm)` (None / <= 0: no clipping) + `optimizer.step()` for a torch.optim.Adam that holds the single flat
    parameter, on the device: its state (`step`, `exp_avg`, `exp_avg_sq`) stays the optimizer's, so checkpoints are unchanged.
    `shadow`: bf16 tensor of the same length that receives the updated parameter (FlatGradBucket.shadow_scope).
    `step_device` (int64 device scalar): the kernels count the step there and derive the bias corrections from it -- the form a captured
    graph needs; `count_host=False` leaves the optimizer's host-side `step` to the caller (one increment per replay)."""
    lib = L.load()
    group = optimizer.param_groups[0]
    assert len(optimizer.param_groups) == 1 and len(group["params"]) == 1 and group["params"][0] is flat_param
    assert not group.get("amsgrad", False) and not group.get("maximize", False)
    st = adam_state(optimizer, flat_param)
    if st["step"].is_cuda:   # state restored from a checkpoint written by a fused / capturable optimizer
        st["step"] = st["step"].detach().cpu()
    if count_host:
        st["step"] += 1
    b1, b2 = group["betas"]
    ws = _workspace("adam", lib.phc_adam_workspace(), flat_param.device, torch.float64)
    L.check(lib.phc_adam_clip_step(flat_param.data_ptr(), flat_grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                   flat_param.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                   int(st["step"].item()), float(max_norm) if max_norm else 0.0, ws.data_ptr(), None,
                                   None if shadow is None else shadow.data_ptr(), None if step_device is None else step_device.data_ptr(),
                                   _stream(flat_param.device)),
            "phc_adam_clip_step")

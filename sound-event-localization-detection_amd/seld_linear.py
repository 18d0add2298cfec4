"""``nn.Linear`` with the reference's parameter names whose WEIGHT GRADIENT is shaped for the GPU.

Every Linear of the SELD models sees B*T = 8000 rows (model_crnn.py:77-83, model_conformer.py:28-51,117-127,
resnet50_model.py:80-91).  Its weight gradient  dW = dY^T X  is then a [out, in] matrix -- as small as 256 x 256 --
reduced over 8000 rows: one library GEMM launches a handful of 64 x 64 tiles that each walk the whole reduction
(measured 54 us for 4 GFLOP; 18 such GEMMs per Conformer iteration).  ``tall_product`` splits the rows into chunks
multiplied as ONE batched GEMM on transposed views (no copies) and adds the partial products in fp32.  Forward and
input gradient are the stock GEMMs.  On a CPU tensor (the reference's plumbing case) this is ``nn.Linear`` itself.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import seld_overlap

enabled = True

# ---- reductions queued during a captured step's backward pass ---------------------------------------------------------
# Every Linear's backward ends in two latency-sized reductions (the chunk sum of the split-K weight gradient, the column
# sum behind the bias gradient): 23 Linears per Conformer iteration, 43 per ResNet50-Conformer one -- 69 / 129 launches of
# ~6 us for a few MB each.  Nothing consumes a weight or bias gradient before the optimiser, so while a batch is open
# (``begin_batch``: seld_graph.GraphedTrainStep around its backward pass; never under DistributedDataParallel, whose reducer
# copies gradients as autograd delivers them) they are queued and ``flush_batch`` runs ONE multi-tensor launch of each kind
# (csrc/glue.hip).  The queued outputs are handed to autograd as fresh aliases: autograd clones a gradient tensor that
# is referenced elsewhere -- here, before it has been filled.
_batch = None


def begin_batch():
    global _batch
    _batch = {"sums": [], "cols": []}


def flush_batch(close=True):
    """Run the queued reductions on the current stream; ``close``: stop queueing."""
    global _batch
    batch = _batch
    if close:
        _batch = None
    elif batch is not None:
        _batch = {"sums": [], "cols": []}
    if batch is None:
        return 0
    import seld_native
    seld_native.multi_sum_chunks(batch["sums"])
    seld_native.multi_column_sums(batch["cols"])
    return len(batch["sums"]) + len(batch["cols"])


def abandon_batch():
    global _batch
    _batch = None


def tall_product(a, c, out_dtype=torch.float32, out=None, queue=False):
    """a^T c for tall operands (a [N, G], c [N, K]) -> [G, K] in ``out_dtype``.  Row strides may be anything (column
    stride 1).  The library's transposed-A GEMM with an 8000-row reduction runs at a fraction of its usual rate, the
    more so the smaller the output (measured, bf16, N = 8000, tools/bench_tall_product.py: 9072 x 512 125 us as one
    GEMM vs 95 us in 5 row chunks; 1536 x 2048 93 vs 76; 512 x 512 52 vs 27 in 8): the rows are split into chunks
    multiplied as ONE batched GEMM on transposed views (no copies) and the partial products are added by a reduction
    that accumulates in fp32 and writes ``out_dtype`` directly (into ``out`` if given)."""
    n, g = a.shape
    k = c.shape[1]
    chunks = chunk_count(n, g, k)
    if chunks == 1:
        prod = a.t() @ c
        if out is not None:
            return out.copy_(prod)
        return prod if prod.dtype == out_dtype else prod.to(out_dtype)
    partial = tall_chunks(a, c, chunks)                                           # [chunks, G, K]
    if out is None:
        out = torch.empty((g, k), dtype=out_dtype, device=a.device)
    if partial.is_cuda and out.is_contiguous() and partial.dtype in (torch.float32, torch.bfloat16) \
            and out.dtype in (torch.float32, torch.bfloat16):
        if queue and _batch is not None:
            _batch["sums"].append((partial, out))            # filled by flush_batch (``queue``: the caller allows it)
            return out
        import seld_native
        return seld_native.sum_chunks(partial, out)          # one launch (torch.sum(out=) is fill + reduce + copy)
    return torch.sum(partial, dim=0, dtype=out.dtype, out=out)


def chunk_count(n, g, k):
    """Row chunks for ``tall_chunks`` (from tools/bench_tall_product.py): 5 when the output has >= 150 tiles of 128 x 128,
    else 8; reduced until it divides n with at least 256 rows per chunk."""
    tiles = ((g + 127) // 128) * ((k + 127) // 128)
    chunks = 5 if tiles >= 150 else 8
    while chunks > 1 and (n % chunks or n // chunks < 256):
        chunks -= 1
    return chunks


def tall_chunks(a, c, chunks=None):
    """The partial products of ``tall_product``: [chunks, G, K] in the operands' dtype, ONE batched GEMM on views."""
    n = a.shape[0]
    if chunks is None:
        chunks = chunk_count(n, a.shape[1], c.shape[1])
    av = a.unflatten(0, (chunks, n // chunks)).transpose(1, 2)                    # [chunks, G, N/chunks] view
    cv = c.unflatten(0, (chunks, n // chunks))
    return torch.bmm(av, cv)


def column_sum(g2, out, queue=False):
    """out[n] = sum_r g2[r, n] with fp32 accumulation: the bias gradient.  csrc/glue.hip (two launches, ~1000 workgroups
    streaming the matrix) where it applies -- the framework's reduction takes 16 - 36 us for these 8000-row shapes whatever
    their width, and a Conformer iteration has 19 of them."""
    import os
    import seld_native
    if os.environ.get("SELD_COLUMN_SUMS", "1") != "0" and seld_native.column_sums_supported(g2, out):      # developer A/B switch
        if queue and _batch is not None:
            _batch["cols"].append((g2, out))
            return out
        return seld_native.column_sums(g2, out)
    return torch.sum(g2, dim=0, dtype=out.dtype, out=out)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, overlap=False):
        ctx.overlap = overlap
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        with torch.autocast(device_type="cuda", enabled=False):
            xc, wc = x.to(cdt), weight.to(cdt)
            y = F.linear(xc, wc, None if bias is None else bias.to(cdt))
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.in_dtype = x.dtype
        ctx.w_dtype = weight.dtype
        ctx.b_dtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, grad):
        xc, wc = ctx.saved_tensors
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]

        with torch.autocast(device_type="cuda", enabled=False):
            g2 = grad.reshape(-1, grad.shape[-1]).to(xc.dtype)
            x2 = xc.reshape(-1, xc.shape[-1])
            dx = (g2 @ wc).view_as(xc) if ctx.needs_input_grad[0] else None
            dw = db = None
            if ctx.overlap and (want_w or want_b):
                # side stream, started with the next BiGRU recurrence (seld_overlap): outputs are allocated now
                if want_w:
                    dw = torch.empty((g2.shape[1], x2.shape[1]), dtype=ctx.w_dtype, device=grad.device)
                if want_b:
                    db = torch.empty((g2.shape[1],), dtype=ctx.b_dtype, device=grad.device)

                def job():
                    if dw is not None:
                        tall_product(g2, x2, out=dw)
                    if db is not None:
                        column_sum(g2, db)

                seld_overlap.submit(grad.device, [t for t in (g2, x2, dw, db) if t is not None], job)
            else:
                dw = tall_product(g2, x2, out_dtype=ctx.w_dtype, queue=True) if want_w else None
                db = column_sum(g2, torch.empty((g2.shape[1],), dtype=ctx.b_dtype, device=grad.device), queue=True) \
                    if want_b else None
                if _batch is not None:                      # possibly queued: fresh aliases for autograd (see above)
                    dw = dw.view_as(dw) if dw is not None else None
                    db = db.view_as(db) if db is not None else None
        if dx is not None and dx.dtype != ctx.in_dtype:
            dx = dx.to(ctx.in_dtype)
        return dx, dw, db, None


class SeldLinear(nn.Linear):
    def forward(self, x):
        late = self.__dict__.pop("_deferred", None)              # seld_overlap.defer_linear: this forward pass only
        if enabled and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() >= 2:
            if late is not None and torch.is_grad_enabled():
                return _Linear.apply(x, late[0], late[1] if len(late) > 1 else None, True)
            return _Linear.apply(x, self.weight, self.bias)
        return super().forward(x)


def linear_on_channels_last_features(linear, y):
    """``linear(features)`` for an encoder output y [B, C, T, F] whose MEMORY is channels-last, i.e. [B][T][F][C]: the
    reference flattens the features channel-major -- ``y.permute(0, 2, 1, 3).reshape(B, T, C * F)``, model_conformer.py:196-200,
    resnet50_model.py:128-133: a 65 MB copy each way at batch 32 -- but the (frequency, channel)-ordered vector is a VIEW of
    that memory, so the COLUMNS of the (small) weight are permuted instead; same dot products.  Returns None when the
    layout does not allow it (the caller then takes the reference's order)."""
    if not (enabled and y.is_cuda and y.dim() == 4 and y.is_contiguous(memory_format=torch.channels_last)
            and y.dtype in (torch.float32, torch.bfloat16)):
        return None
    b, c, t, f = y.shape
    if linear.in_features != c * f:
        return None
    feats = y.permute(0, 2, 3, 1).reshape(b, t, f * c)                       # a view: (f, c) order
    w = linear.weight.view(linear.out_features, c, f).transpose(1, 2).reshape(linear.out_features, f * c)
    return _Linear.apply(feats, w, linear.bias)


conv1x1_as_gemm = True        # Config.CONV1X1_AS_GEMM via trainer.prepare_model_for_device


def conv1x1(conv, x):
    """``conv(x)`` for a 1x1 / stride-1 / bias-free nn.Conv2d on a channels-last activation (the first and third
    convolution of every ResNet bottleneck, resnet50_model.py:30-52).  In channels-last memory x [B, C, T, F] IS a row-major
    [B*T*F, C] matrix, so the convolution is a plain GEMM on views: forward and data gradient through hipBLASLt, the
    weight gradient as the split-K product of ``tall_product`` with its chunk sum queued with the Linears' (MIOpen runs
    these as CK batched split-K GEMMs with a memset each: 27 + 65 launches, 1.8 ms of a ResNet50-Conformer iteration)."""
    if not (conv1x1_as_gemm and enabled and x.is_cuda and x.dim() == 4 and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and x.is_contiguous(memory_format=torch.channels_last) and x.dtype in (torch.float32, torch.bfloat16)):
        return conv(x)
    b, c, t, f = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(b * t * f, c)                       # a view of the channels-last memory
    y2 = _Linear.apply(x2, conv.weight.reshape(conv.out_channels, c), None)
    return y2.view(b, t, f, conv.out_channels).permute(0, 3, 1, 2)         # channels-last [B, C_out, T, F]

"""The glue kernels of the training iteration (csrc/glue.hip) against the framework op chains they replace
(clone / fill / add / cast around nn.GRU's biases, fill + reduce + copy behind split-K weight gradients, the slice
copies that assemble dW_hh, flip + copy for the transposed convolution weights).  Index work is bit-exact; sums
accumulate in fp32 in a fixed order and are compared with a float64 sum at fp32 / bf16 rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 256


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gru_fold_bias_matches_the_framework_chain(gpu_device, dtype):
    import seld_native
    g = torch.Generator().manual_seed(0)
    b_ih = torch.randn(6 * H, generator=g).to(gpu_device)
    b_hh = torch.randn(2, 3 * H, generator=g).to(gpu_device)
    gi_bias, b_hn = seld_native.gru_fold_bias(b_ih, b_hh.reshape(-1), dtype)
    fold = b_hh.clone()
    fold[:, 2 * H:] = 0
    want = (b_ih + fold.reshape(-1)).to(dtype)
    assert gi_bias.dtype == dtype and torch.equal(gi_bias, want)
    assert torch.equal(b_hn, b_hh[:, 2 * H:])


@pytest.mark.parametrize("tiles", [1, 3, 8])
def test_gru_bias_grads_matches_the_framework_chain(gpu_device, tiles):
    import seld_native
    g = torch.Generator().manual_seed(tiles)
    partial = torch.randn(tiles, 2, 4, H, generator=g).to(gpu_device)
    db_ih, db_hh = seld_native.gru_bias_grads(partial)
    total = partial.double().sum(dim=0)
    want_ih = total[:, :3].reshape(-1)
    want_hh = torch.cat((total[:, :2], total[:, 3:]), dim=1).reshape(-1)
    assert tuple(db_ih.shape) == (6 * H,) and tuple(db_hh.shape) == (6 * H,)
    assert (db_ih.double() - want_ih).abs().max().item() <= 1e-5
    assert (db_hh.double() - want_hh).abs().max().item() <= 1e-5


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                                (torch.float32, torch.float32), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("shape", [(5, 9072, 512), (8, 512, 512), (3, 7, 5), (1, 33, 3)])
def test_sum_chunks(gpu_device, in_dtype, out_dtype, shape):
    import seld_native
    g = torch.Generator().manual_seed(shape[0])
    partial = torch.randn(*shape, generator=g).to(in_dtype).to(gpu_device)
    out = torch.empty(shape[1:], dtype=out_dtype, device=gpu_device)
    seld_native.sum_chunks(partial, out)
    want = partial.double().sum(dim=0)
    tol = (2.0 ** -8 if out_dtype == torch.bfloat16 else 2e-6) * (want.abs() + 1.0)
    assert ((out.double() - want).abs() <= tol).all()


def test_tall_product_uses_it_and_matches_the_plain_product(gpu_device):
    from seld_linear import tall_product
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(8000, 512, generator=g) * 0.1).to(torch.bfloat16).to(gpu_device)
    c = torch.randn(8000, 384, generator=g).to(torch.bfloat16).to(gpu_device)
    want = a.double().t() @ c.double()
    for out_dtype in (torch.float32, torch.bfloat16):
        got = tall_product(a, c, out_dtype=out_dtype)
        out = torch.empty(512, 384, dtype=out_dtype, device=gpu_device)
        assert tall_product(a, c, out=out) is out and torch.equal(out, got)
        # the partial products are bf16: each of the 8 chunks is rounded once before the fp32 sum
        assert (got.double() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()


@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32),
                                             (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("chunks", [1, 5])
def test_gru_dwhh_finish_extracts_the_matching_direction_blocks(gpu_device, dtype, out_dtype, chunks):
    import seld_native
    g = torch.Generator().manual_seed(chunks)
    p_gi = torch.randn(chunks, 6 * H, 2 * H, generator=g).to(dtype).to(gpu_device)
    p_n = torch.randn(chunks, 2 * H, 2 * H, generator=g).to(dtype).to(gpu_device)
    out = torch.empty(2, 3 * H, H, dtype=out_dtype, device=gpu_device)
    seld_native.gru_dwhh_finish(p_gi, p_n, out)
    s_gi = p_gi.double().sum(dim=0).view(2, 3, H, 2, H)
    s_n = p_n.double().sum(dim=0).view(2, H, 2, H)
    want = torch.empty(2, 3 * H, H, dtype=torch.float64, device=gpu_device)
    for d in range(2):
        want[d, :2 * H].view(2, H, H).copy_(s_gi[d, :2, :, d])
        want[d, 2 * H:].copy_(s_n[d, :, d])
    tol = (2.0 ** -8 if out_dtype == torch.bfloat16 else 2e-6) * (want.abs() + 1.0)
    assert ((out.double() - want).abs() <= tol).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("o,i", [(64, 4), (128, 64), (512, 256), (16, 16)])
def test_conv_weight_flip_transpose_is_the_framework_expression(gpu_device, dtype, o, i):
    import seld_native
    g = torch.Generator().manual_seed(o)
    w = torch.randn(o, i, 3, 3, generator=g).to(dtype).to(gpu_device).contiguous(memory_format=torch.channels_last)
    got = seld_native.conv_weight_flip_transpose(w)
    want = w.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)
    assert got.shape == want.shape and got.stride() == want.stride() and torch.equal(got, want)


def test_conv3x3_data_gradient_still_matches_autograd(gpu_device):
    from model_crnn import _Conv3x3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 9, 8, generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(32, 16, 3, 3, generator=g) * 0.1).to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    go = torch.randn(2, 32, 9, 8, generator=g).to(gpu_device)
    _Conv3x3.apply(x, w).backward(go)
    dx, dw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    torch.nn.functional.conv2d(x, w, padding=1).backward(go)
    assert (dx - x.grad).abs().max().item() <= 1e-4 * x.grad.abs().max().item()
    assert (dw - w.grad).abs().max().item() <= 1e-4 * w.grad.abs().max().item()


def test_multi_tensor_reductions_equal_the_single_ones(gpu_device):
    """``multi_sum_chunks`` / ``multi_column_sums`` (one launch for every queued reduction of a backward pass) against
    ``sum_chunks`` on the same tensors (same summation order: IDENTICAL) and a float64 column sum; run
    twice (the column kernel's arrival counters must come back to zero) and with more pairs than one launch holds."""
    import seld_native
    g = torch.Generator().manual_seed(5)
    shapes = [(5, 9072, 512), (8, 512, 512), (3, 7, 5), (1, 33, 3), (8, 256, 1024), (2, 1024, 256)] * 12     # 72 > 64
    sums = []
    for i, shape in enumerate(shapes):
        in_dtype = torch.bfloat16 if i % 3 else torch.float32
        out_dtype = torch.float32 if i % 2 else torch.bfloat16
        partial = torch.randn(*shape, generator=g).to(in_dtype).to(gpu_device)
        sums.append((partial, torch.empty(shape[1:], dtype=out_dtype, device=gpu_device)))
    cols = []
    for i, (rows, n) in enumerate([(8000, 256), (8000, 9072), (8000, 1024), (750, 512), (3, 8), (8000, 512)] * 8):    # 48 > 40
        in_dtype = torch.bfloat16 if i % 3 else torch.float32
        out_dtype = torch.bfloat16 if i % 2 else torch.float32
        m = (torch.randn(rows, n, generator=g) * 0.3).to(in_dtype).to(gpu_device)
        cols.append((m, torch.empty(n, dtype=out_dtype, device=gpu_device)))
    for _ in range(2):
        for _, out in sums + cols:
            out.fill_(float("nan"))
        seld_native.multi_sum_chunks(sums)
        seld_native.multi_column_sums(cols)
        for partial, out in sums:
            assert torch.equal(out, seld_native.sum_chunks(partial, torch.empty_like(out)))
        for m, out in cols:                                # (row blocks added in a different fixed order than column_sums)
            want = m.double().sum(dim=0)
            tol = (2.0 ** -8 if out.dtype == torch.bfloat16 else 1e-5) * (want.abs() + 1.0)
            assert ((out.double() - want).abs() <= tol).all()


def test_linear_backward_queues_its_reductions_inside_a_batch(gpu_device):
    """seld_linear.begin_batch / flush_batch (what the captured step does around its backward pass): the weight and bias
    gradients of SeldLinear layers are the same tensors' worth of numbers as without the batch, bit for bit."""
    import seld_linear
    from seld_linear import SeldLinear
    torch.manual_seed(0)
    layers = [SeldLinear(256, 1024), SeldLinear(1024, 256), SeldLinear(256, 9072)]
    for m in layers:
        m.to(gpu_device).to(torch.bfloat16)
    x = torch.randn(8000, 256, device=gpu_device).to(torch.bfloat16)

    def run(batched):
        for m in layers:
            m.weight.grad = m.bias.grad = None
        y = layers[2](layers[1](torch.relu(layers[0](x))))
        if batched:
            seld_linear.begin_batch()
        y.float().square().mean().backward()
        if batched:
            assert seld_linear.flush_batch() == 6
        return [p.grad.clone() for m in layers for p in (m.weight, m.bias)]

    plain, queued = run(False), run(True)
    assert seld_linear._batch is None
    for i, (a, b) in enumerate(zip(plain, queued)):
        assert torch.isfinite(a.float()).all()
        if i % 2 == 0:                                     # weights: same chunk order
            assert torch.equal(a, b)
        else:                                              # biases: the row blocks are added in a different fixed order
            assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -7 * a.float().abs().max().item()

"""bf16 MFMA Dense layer (hand-written v_mfma_f32_32x32x16_bf16 kernel) vs a plain PyTorch fp32 reference of
the same op on the same bf16-rounded operands (tolerance = bf16 output rounding / f32 accumulation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("batch,k,n,act", [(128, 16, 128, "identity"), (256, 128, 128, "relu"), (4096, 256, 256, "relu"),
                                          (1024, 64, 384, "tanh"), (32768, 128, 128, "relu")])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_dense_bf16_mfma_forward(batch, k, n, act, out_dtype):
    from rlhip import ops

    g = torch.Generator(device="cpu").manual_seed(batch + k + n)
    # ASYMMETRIC operands (a symmetric B would hide a transposed C write -- MI355X guide section 3)
    x = (torch.randn((k, batch), generator=g) * 0.5).cuda()                 # SoA f32 activations
    w_flux = (torch.randn(n * k, generator=g) / np.sqrt(k)).cuda()          # (n x k) column-major flat
    bias = torch.randn(n, generator=g).cuda()
    xr = ops.soa_to_bf16_rows(x)
    wt = ops.dense_pack_weight_bf16(w_flux, k, n)
    assert xr.shape == (batch, k) and wt.shape == (n, k)
    # converters are exact bf16 round-to-nearest-even
    assert torch.equal(xr, x.t().contiguous().to(torch.bfloat16))
    W = w_flux.reshape(k, n).t().contiguous()                                # W[o, i]
    assert torch.equal(wt, W.to(torch.bfloat16))
    y = ops.dense_bf16_forward(xr, wt, bias, act, out_dtype)
    ref = xr.float() @ wt.float().t() + bias
    ref = {"relu": torch.relu, "tanh": torch.tanh, "identity": lambda t: t}[act](ref)
    if out_dtype == torch.bfloat16:
        torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)
    else:
        torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    back = ops.bf16_rows_to_soa(y, n) if out_dtype == torch.bfloat16 else None
    if back is not None:
        assert torch.equal(back, y.float().t().contiguous())


def test_dense_mfma_identity_weight_catches_layout_errors():
    """A = I check with an asymmetric operand: Y must equal X exactly (bf16 values, f32 accumulate)."""
    from rlhip import ops

    k = n = 128
    batch = 256
    x = torch.arange(batch * k, dtype=torch.float32).reshape(batch, k).remainder(251).sub(125).cuda()  # exact in bf16
    xr = x.to(torch.bfloat16).contiguous()
    eye = torch.eye(n, dtype=torch.bfloat16, device="cuda").contiguous()
    y = ops.dense_bf16_forward(xr, eye, None, "identity", torch.float32)
    assert torch.equal(y, xr.float())


def test_dense_mfma_argument_validation():
    from rlhip import ops
    from rlhip._lib import RLHipArgumentError

    x = torch.zeros((100, 16), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((128, 16), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RLHipArgumentError):
        ops.dense_bf16_forward(x, w)  # batch not a multiple of 128


@pytest.mark.parametrize("batch,k,n,act", [(128, 16, 128, "identity"), (256, 128, 128, "relu"), (4096, 256, 256, "relu"),
                                          (1024, 64, 384, "tanh"), (2048, 512, 512, "relu"), (32768, 128, 256, "relu")])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_dense_bf16_mfma_tiled_forward(batch, k, n, act, out_dtype):
    """the LDS-staged tiled kernel (fragment-ordered weights) against the same torch reference, and bit-for-bit
    against the simple kernel for the f32 output (same bf16 operands, same k order inside the MFMA chain)."""
    from rlhip import ops

    g = torch.Generator(device="cpu").manual_seed(batch + k + n + 1)
    xr = (torch.randn((batch, k), generator=g) * 0.5).cuda().to(torch.bfloat16)
    wt = (torch.randn((n, k), generator=g) / np.sqrt(k)).cuda().to(torch.bfloat16)
    bias = torch.randn(n, generator=g).cuda()
    wf = ops.dense_frag_weight_bf16(wt)
    # fragment order: fragment (ks, tg), lane l, element u  <-  Wt[32 tg + (l & 31)][16 ks + 8 (l >> 5) + u]
    q = torch.arange(n * k, device="cuda")
    u, l, f = q & 7, (q >> 3) & 63, q >> 9
    tg, ks = f % (n // 32), f // (n // 32)
    assert torch.equal(wf, wt[32 * tg + (l & 31), 16 * ks + 8 * (l >> 5) + u])
    y = ops.dense_bf16_forward_tiled(xr, wf, n, bias, act, out_dtype)
    ref = xr.float() @ wt.float().t() + bias
    ref = {"relu": torch.relu, "tanh": torch.tanh, "identity": lambda t: t}[act](ref)
    if out_dtype == torch.bfloat16:
        torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)
    else:
        torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
        assert torch.equal(y, ops.dense_bf16_forward(xr, wt, bias, act, out_dtype))


def test_dense_tiled_identity_weight():
    from rlhip import ops

    k = n = 256
    batch = 384
    x = torch.arange(batch * k, dtype=torch.float32).reshape(batch, k).remainder(251).sub(125).cuda()
    xr = x.to(torch.bfloat16).contiguous()
    wf = ops.dense_frag_weight_bf16(torch.eye(n, dtype=torch.bfloat16, device="cuda").contiguous())
    for dt in (torch.float32, torch.bfloat16):
        y = ops.dense_bf16_forward_tiled(xr, wf, n, None, "identity", dt)
        assert torch.equal(y.float(), xr.float())


@pytest.mark.parametrize("n", [256, 128])
@pytest.mark.parametrize("batch,k,act", [(128, 256, "relu"), (384, 128, "identity"), (33 * 128, 256, "tanh"),
                                         (1025 * 128, 256, "relu"), (600 * 128, 128, "relu")])
def test_dense_persistent_kernel_equals_the_tiled_kernel_bitwise(batch, k, act, n):
    """N = 256 / 128 with bf16 output takes the persistent kernel (weights in registers, double-buffered X tiles, one
    workgroup per CU walking tiles b, b + 256, ...): same MFMA chain order as the tiled kernel, so its bf16 result
    equals the tiled kernel's f32 result rounded to bf16, bit for bit -- including uneven tile counts per workgroup"""
    from rlhip import ops

    g = torch.Generator(device="cpu").manual_seed(batch + k)
    xr = (torch.randn((batch, k), generator=g) * 0.5).cuda().to(torch.bfloat16)
    wt = (torch.randn((n, k), generator=g) / np.sqrt(k)).cuda().to(torch.bfloat16)
    bias = torch.randn(n, generator=g).cuda()
    wf = ops.dense_frag_weight_bf16(wt)
    y16 = ops.dense_bf16_forward_tiled(xr, wf, n, bias, act, torch.bfloat16)
    y32 = ops.dense_bf16_forward_tiled(xr, wf, n, bias, act, torch.float32)
    assert torch.equal(y16, y32.to(torch.bfloat16))
    y16b = ops.dense_bf16_forward_tiled(xr, wf, n, None, act, torch.bfloat16)      # no bias
    assert torch.equal(y16b, ops.dense_bf16_forward_tiled(xr, wf, n, None, act, torch.float32).to(torch.bfloat16))


@pytest.mark.parametrize("batch,k,n,act", [(33 * 128, 512, 512, "relu"), (1024, 512, 256, "tanh"), (8192, 256, 512, "relu"),
                                           (2048, 128, 1024, "identity"), (640, 512, 128, "relu"),
                                           (300 * 128, 512, 1024, "relu")])
def test_dense_persistent_kernel_wide_layers_bitwise(batch, k, n, act):
    """K = 512 (256 VGPRs of register-resident fragments, one workgroup per CU) and N > 256 (column blocks of 256 over
    blockIdx.y): the bf16 result equals the tiled kernel's f32 result rounded to bf16, bit for bit"""
    from rlhip import ops

    g = torch.Generator(device="cpu").manual_seed(batch + k + n)
    xr = (torch.randn((batch, k), generator=g) * 0.5).cuda().to(torch.bfloat16)
    wt = (torch.randn((n, k), generator=g) / np.sqrt(k)).cuda().to(torch.bfloat16)
    bias = torch.randn(n, generator=g).cuda()
    wf = ops.dense_frag_weight_bf16(wt)
    y16 = ops.dense_bf16_forward_tiled(xr, wf, n, bias, act, torch.bfloat16)
    y32 = ops.dense_bf16_forward_tiled(xr, wf, n, bias, act, torch.float32)
    assert torch.equal(y16, y32.to(torch.bfloat16))
    ref = xr.float() @ wt.float().t() + bias
    ref = {"relu": torch.relu, "tanh": torch.tanh, "identity": lambda t: t}[act](ref)
    torch.testing.assert_close(y16.float(), ref, rtol=2e-2, atol=2e-2)

"""GPU: the fp8-weight inference path (QST_PREC_FP8W; BASELINE configs[4] "fp8 weights").
Weights are e4m3 (OCP) with one fp32 scale per output row, activations bf16, accumulation fp32. The oracle applies the
SAME quantisation (oracle/torch_ref.fp8_weight_arena, torch.float8_e4m3fn) and then runs the bf16-operand reference,
so the comparison isolates the kernels: tolerances are those of the bf16 path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd import _lib  # noqa: E402
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout  # noqa: E402
from quadruplet_sentence_transformer_amd.encoder import HipEncoder  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from oracle import torch_ref as R  # noqa: E402


def stream():
    return _lib.current_stream_ptr()


@pytest.mark.parametrize("rows,cols", [(5, 64), (192, 384), (1000, 1536), (3, 4)])
def test_row_quantisation_is_bit_exact_against_torch_float8(rows, cols):
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + cols)
    w = torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 0.1
    w[0] = 0.0                                               # an all-zero row keeps scale 1
    wd = w.cuda()
    q = torch.empty(rows, cols, dtype=torch.uint8, device="cuda")
    sc = torch.empty(rows, device="cuda")
    _lib.check(lib.qst_quant_rows_fp8(wd.data_ptr(), rows, cols, q.data_ptr(), sc.data_ptr(), stream()))
    amax = w.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    want = (w / scale).to(torch.float8_e4m3fn)
    torch.testing.assert_close(sc.cpu(), scale.reshape(-1), rtol=0, atol=0)
    got = q.cpu().view(torch.float8_e4m3fn).to(torch.float32)
    # +0 / -0 are distinct bytes with the same value: compare values
    torch.testing.assert_close(got, want.to(torch.float32), rtol=0, atol=0)
    assert float(got.abs().max()) <= 448.0


@pytest.mark.parametrize("M,N,K", [(128, 192, 64), (300, 384, 384), (1000, 1152, 384), (4096, 384, 1536), (64, 64, 128)])
def test_gemm_with_fp8_weights(M, N, K):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = torch.randn(N, K, generator=g) * 0.05
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Wd = W.cuda()
    q = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    sc = torch.empty(N, device="cuda")
    _lib.check(lib.qst_quant_rows_fp8(Wd.data_ptr(), N, K, q.data_ptr(), sc.data_ptr(), stream()))
    Wq = q.cpu().view(torch.float8_e4m3fn).to(torch.float32) * sc.cpu()[:, None]
    ref = A.float() @ Wq.t() + bias
    Ad, bd, rd = A.cuda(), bias.cuda(), resid.cuda()

    def args(**kw):
        a = _lib.QstGemmArgs()
        a._keep = [v for v in kw.values() if torch.is_tensor(v)]
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
        return a
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.qst_gemm_nt_w8(args(A=Ad, B=q, C=Cb, bias=bd, bscale=sc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N), 0, stream()))
    torch.testing.assert_close(Cb.float().cpu(), ref, rtol=8e-3, atol=2e-2)
    Cf = torch.empty(M, N, device="cuda")
    _lib.check(lib.qst_gemm_nt_w8(args(A=Ad, B=q, C=Cf, bias=bd, resid=rd, bscale=sc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N,
                                       ldr=N), 1, stream()))
    torch.testing.assert_close(Cf.cpu(), ref + resid, rtol=1e-4, atol=1e-3)
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.qst_gemm_nt_w8(args(A=Ad, B=q, C=Cb, C2=C2, bias=bd, bscale=sc, M=M, N=N, K=K, lda=K, ldb=K, ldc=N), 2,
                                  stream()))
    torch.testing.assert_close(C2.float().cpu(), torch.nn.functional.gelu(ref), rtol=8e-3, atol=2e-2)
    # missing scales / unsupported leading dimension are refused
    assert lib.qst_gemm_nt_w8(args(A=Ad, B=q, C=Cb, M=M, N=N, K=K, lda=K, ldb=K, ldc=N), 0, stream()) == -1
    assert lib.qst_gemm_nt_w8(args(A=Ad, B=q, C=Cb, bscale=sc, M=M, N=N, K=K, lda=K, ldb=K + 8, ldc=N), 0, stream()) == -2


@pytest.mark.parametrize("name,B,L,wkw", [("tiny-bert", 6, 32, dict(std=0.05, bias_std=0.02, ln_jitter=0.05)),
                                          ("tiny-mpnet", 4, 64, dict(std=0.05, bias_std=0.02, ln_jitter=0.05)),
                                          ("all-MiniLM-L6-v2", 2, 128, dict(std=0.02)),
                                          ("bert-base-uncased", 1, 64, dict(std=0.02))])
def test_fp8w_encoder_matches_the_oracle_with_the_same_quantised_weights(name, B, L, wkw):
    cfg = PRESETS[name]
    arena = synthetic_params(cfg, seed=14, **wkw)
    ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)
    ids_t, mask_t, types_t = (torch.from_numpy(x).view(4 * B, L) for x in (ids, mask, types))
    arena_q, _ = R.fp8_weight_arena(arena, cfg)
    with torch.no_grad():
        ref = R.sentence_embeddings(R.arena_to_dict(arena_q, cfg), cfg, ids_t, mask_t, types_t, bf16_operands=True)
        ref_bf16 = R.sentence_embeddings(R.arena_to_dict(arena, cfg), cfg, ids_t, mask_t, types_t, bf16_operands=True)
    enc = HipEncoder(cfg)
    enc.load_arena(arena)
    tt = types_t.cuda() if cfg.type_vocab_size else None
    emb, _, _ = enc.forward(ids_t.cuda(), mask_t.cuda(), tt, precision="fp8w")
    sc = 1.0 if cfg.normalize else float(ref.norm(dim=-1).mean())
    torch.testing.assert_close(emb.cpu(), ref, rtol=1e-3, atol=1.5e-3 * sc)
    # and the quantisation itself is what moves the embeddings: the un-quantised reference is measurably further away
    d_q = (emb.cpu() - ref).abs().max().item()
    d_full = (emb.cpu() - ref_bf16).abs().max().item()
    assert d_full > 2 * d_q
    # the bf16 path on the same encoder is untouched by the fp8 shadow
    emb16, _, _ = enc.forward(ids_t.cuda(), mask_t.cuda(), tt)
    torch.testing.assert_close(emb16.cpu(), ref_bf16, rtol=1e-3, atol=1.5e-3 * sc)
    with pytest.raises(_lib.QstError):
        enc.forward(ids_t.cuda(), mask_t.cuda(), tt, training=True, precision="fp8w")

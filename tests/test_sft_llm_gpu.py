"""Gradient of a loss on the trajectory-token hidden states w.r.t. `latent_queries` through the frozen decoder (internnav_amd.sft_llm):
the cached B * n_query-row backward against torch autograd of the fp32 oracle over the WHOLE sequence (oracle/qwen_vl.generate_latents
= the reference's full forward with the TRAJ rows overwritten, internvla_n1.py:166-172,320-347), bf16-autocast autograd as yardstick."""
import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_grad(sd0, cfg, inp, G, autocast):
    from oracle import qwen_vl as o_q

    sd = {k: v.float() for k, v in sd0.items()}
    lq = sd["model.latent_queries"].clone().requires_grad_(True)
    sd["model.latent_queries"] = lq
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        h = o_q.generate_latents(sd, cfg, inp["input_ids"], inp["pixel_values"].float(), inp["grid_thw"])
    (h.float() * G).sum().backward()
    return h.detach().float(), lq.grad.reshape(-1, lq.shape[-1])


@pytest.mark.parametrize("ragged", [False, True])
def test_latent_query_gradient(built_lib, ragged):
    from internnav_amd.qwen_vl import QwenVLEngine
    from internnav_amd.sft_llm import LatentQueryGrad

    cfg = W.QWEN_TEST_CFG
    B = 2
    sd = W.qwen_state_dict(seed=11, cfg=cfg)
    inp = W.qwen_inputs(B, 2, seed=9, cfg=cfg)
    eng = QwenVLEngine(sd, cfg, DEV, max_seqs=B, max_seq_len=512, max_patches=inp["pixel_values"].shape[0])
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    nq, H = cfg["n_query"], cfg["t_hidden"]
    G = torch.randn(B, nq, H, generator=g)
    if ragged:
        # the second sequence is shorter: right-padded batch, per-sequence cache rows / key lengths
        S = inp["input_ids"].shape[1]
        lens = [S, S - 7]
        ids = inp["input_ids"].clone()
        state = eng.prefill(ids, pv, inp["grid_thw"], seq_lens=lens)
    else:
        state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    lq = LatentQueryGrad(eng)
    hq = lq.forward(state)
    d_lat = lq.backward(G.to(DEV, torch.bfloat16)).cpu()
    if ragged:
        href, gref, gy = [], 0, 0
        for b in range(B):
            one = dict(input_ids=inp["input_ids"][b:b + 1, : lens[b]], pixel_values=inp["pixel_values"], grid_thw=inp["grid_thw"])
            # per-sequence pixel rows: qwen_inputs lays the images of sequence b out contiguously
            per = inp["pixel_values"].shape[0] // B
            one["pixel_values"] = inp["pixel_values"][b * per:(b + 1) * per]
            one["grid_thw"] = inp["grid_thw"][b * (inp["grid_thw"].shape[0] // B):(b + 1) * (inp["grid_thw"].shape[0] // B)]
            h, gr = _oracle_grad(sd, cfg, one, G[b:b + 1], False)
            _, gry = _oracle_grad(sd, cfg, one, G[b:b + 1], True)
            href.append(h)
            gref = gref + gr
            gy = gy + gry
        href = torch.cat(href)
    else:
        href, gref = _oracle_grad(sd, cfg, inp, G, False)
        _, gy = _oracle_grad(sd, cfg, inp, G, True)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    e_h = rel(hq.float().cpu(), href)
    e_g, y_g = rel(d_lat, gref), rel(gy, gref)
    print(f"hidden states rel err {e_h:.3e}; d latent_queries: engine {e_g:.3e}, bf16 autocast PyTorch {y_g:.3e}")
    assert e_h < 2e-2
    assert e_g <= 1.25 * y_g + 1e-3, (e_g, y_g)
    # the training forward equals the inference latent-query pass (same kernels but an unfused SwiGLU)
    state2 = eng.prefill(inp["input_ids"], pv, inp["grid_thw"], **(dict(seq_lens=lens) if ragged else {}))
    inf = eng.latents(state2, None)
    assert rel(hq.float().cpu(), inf.float().cpu()) < 1e-2

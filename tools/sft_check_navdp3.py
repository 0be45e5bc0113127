import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from test_sft_navdp_gpu import _inputs, _rel
from internnav_amd import sft as E, synthetic as S
import oracle.sft as O
import oracle.navdp as ON
dev = torch.device("cuda:0")
cfg = S.N1_NAVDP_CFG
sd0 = {k: v.float() for k, v in S.materialize(S.n1_navdp_spec(), 3).items()}
inp = _inputs(2, 2)
rec = []
orig = ON.decoder_layer
def dl(x, mem, sd, p, nhead, norm_first, act, tgt_mask=None, memory_mask=None, eps=1e-5):
    if p == "decoder.layers.0": rec.append(("cond", mem.detach().float().clone())); rec.append(("x0", x.detach().float().clone()))
    y = orig(x, mem, sd, p, nhead, norm_first, act, tgt_mask, memory_mask, eps)
    if p.startswith("decoder.layers."): rec.append((f"layer{p.split('.')[-1]}", y.detach().float().clone()))
    return y
ON.decoder_layer = dl
orig_lin = O.linear
def cap_linear(x, sd, p):
    y = orig_lin(x, sd, p)
    if p == "action_head": rec.append(("lnout", x.detach().float().clone())); rec.append(("out", y.detach().float().clone()))
    return y
O.linear = cap_linear
def run(ac):
    rec.clear()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
        O.navdp_sft_loss(sd0, inp["hidden_q"], inp["traj_images"], inp["traj_depths"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"], cfg)
    return dict(rec)
r32, r16 = run(False), run(True)
head = E.NavDPSftHead(sd0, dev, cfg)
head.taps = []
head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"])
for name, v in head.taps:
    a = v.cpu().view_as(r32[name])
    print(f"{name:8s} engine {_rel(a, r32[name]):.3e}  autocast {_rel(r16[name], r32[name]):.3e}   |x| {r32[name].norm().item():.2f}")

oe = head.last_prediction.cpu().view_as(r32["out"]); o32 = r32["out"]; o16 = r16["out"]
print("out: engine", _rel(oe, o32), "autocast", _rel(o16, o32), "|out| rms", o32.pow(2).mean().sqrt().item())
for nm, o in (("engine", oe), ("autocast", o16)):
    d = (o - o32)
    print(nm, "mean err per column", d.mean(0).view(-1).tolist(), "rms err", d.pow(2).mean().sqrt().item(), "corr(err, out)", (d * o32).sum().item() / (o32.pow(2).sum().item()))

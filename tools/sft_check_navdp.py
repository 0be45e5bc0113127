import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from test_sft_navdp_gpu import _inputs, _oracle, _rel
from internnav_amd import sft as E, synthetic as S
dev = torch.device("cuda:0")
cfg = S.N1_NAVDP_CFG
sd0 = {k: v.float() for k, v in S.materialize(S.n1_navdp_spec(), 3).items()}
inp = _inputs(2, 2)
l32, dh32, g32 = _oracle(sd0, inp, cfg, False)
l16, dh16, g16 = _oracle(sd0, inp, cfg, True)
head = E.NavDPSftHead(sd0, dev, cfg)
loss, dh = head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"])
rows = []
for k, ref in g32.items():
    if "rgb_model" in k or ref.norm() < 1e-9: continue
    rows.append((_rel(head.P.grad(k).cpu().view_as(ref), ref), _rel(g16[k], ref), ref.norm().item(), k))
groups = {}
for e, y, n, k in rows:
    grp = ".".join(k.split(".")[:3]) if k.startswith(("decoder", "rgbd_encoder.depth")) else ".".join(k.split(".")[:4])
    if k.startswith("decoder.layers"): grp = "decoder.layers." + k.split(".")[2]
    if k.startswith("rgbd_encoder.depth_model.blocks"): grp = "depth.blocks"
    groups.setdefault(grp, []).append((e, y, n))
for grp, v in groups.items():
    print(f"{grp:60s} n={len(v):3d} engine {sum(a for a,_,_ in v)/len(v):.3e} bf16 {sum(b for _,b,_ in v)/len(v):.3e} |g| {sum(c for _,_,c in v)/len(v):.3e}")
print("loss engine", loss.item(), "oracle", l32, "bf16", l16, " dh:", _rel(dh.float().cpu().view_as(dh32), dh32), _rel(dh16, dh32))
for depth in (1, 4):
    cfg2 = dict(cfg, temporal_depth=depth)
    l32b, _, g32b = _oracle(sd0, inp, cfg2, False)
    l16b, _, g16b = _oracle(sd0, inp, cfg2, True)
    head2 = E.NavDPSftHead(sd0, dev, cfg2)
    loss2, _ = head2.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"])
    k = "action_head.weight"
    print(f"depth {depth}: loss {loss2.item():.6f} / {l32b:.6f} / bf16 {l16b:.6f}; action_head.weight engine {_rel(head2.P.grad(k).cpu(), g32b[k]):.3e} bf16 {_rel(g16b[k], g32b[k]):.3e};"
          f" layernorm.bias {_rel(head2.P.grad('layernorm.bias').cpu(), g32b['layernorm.bias']):.3e} / {_rel(g16b['layernorm.bias'], g32b['layernorm.bias']):.3e}")

import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from test_sft_navdp_gpu import _inputs, _rel
from internnav_amd import sft as E, synthetic as S
import oracle.sft as O
import torch.nn.functional as F
dev = torch.device("cuda:0")
cfg0 = S.N1_NAVDP_CFG
sd0 = {k: v.float() for k, v in S.materialize(S.n1_navdp_spec(), 3).items()}
inp = _inputs(2, 2)
# oracle prediction: patch the loss function to capture pred
import oracle.nn_ref as NR
cap = {}
orig_linear = O.linear
def cap_linear(x, sd, p):
    y = orig_linear(x, sd, p)
    if p == "action_head": cap["pred"] = y.detach().float(); cap["ln"] = x.detach().float()
    return y
O.linear = cap_linear
for depth in (1, 4, 16):
    cfg = dict(cfg0, temporal_depth=depth)
    with torch.no_grad():
        O.navdp_sft_loss(sd0, inp["hidden_q"], inp["traj_images"], inp["traj_depths"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"], cfg)
        p32, ln32 = cap["pred"].clone(), cap["ln"].clone()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            O.navdp_sft_loss(sd0, inp["hidden_q"], inp["traj_images"], inp["traj_depths"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"], cfg)
        p16, ln16 = cap["pred"].clone(), cap["ln"].clone()
    head = E.NavDPSftHead(sd0, dev, cfg)
    head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["timesteps"])
    pe = head.last_prediction.cpu().view_as(p32)
    print(f"depth {depth}: prediction rel err engine {_rel(pe, p32):.3e} bf16 autocast {_rel(p16, p32):.3e}; LN-out autocast {_rel(ln16, ln32):.3e}")

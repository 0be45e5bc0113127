"""Engine SFT gradients vs oracle autograd (fp32) and vs pure-bf16 PyTorch autograd of the same oracle (yardstick). GPU box."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import synthetic as S
from internnav_amd import sft as E
from oracle import sft as O

def inputs(B, T, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(hidden_q=torch.randn(B, 4, 3584, generator=g).bfloat16().float(), traj_images=torch.rand(B, T, 224, 224, 3, generator=g),
                traj_poses=torch.randn(B, T, 32, 3, generator=g), video_frame_num=torch.tensor([T] + [max(1, T - 1)] * (B - 1)),
                noise=torch.randn(B * T, 32, 3, generator=g), t_index=torch.randint(0, 1000, (B * T,), generator=g))

def oracle_grads(sd0, inp, dtype):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    hq = inp["hidden_q"].clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):      # yardstick: the same functions under bf16 autocast
        loss = O.nextdit_sft_loss(sd, hq, inp["traj_images"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["t_index"])
    loss.backward()
    return loss.item(), hq.grad.float(), {k: (v.grad.float() if v.grad is not None else None) for k, v in sd.items()}

if __name__ == "__main__":
    B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    sd0 = {k: v.float() for k, v in S.materialize(S.n1_nextdit_spec(), 3).items()}
    inp = inputs(B, T)
    t0 = time.time(); l32, dh32, g32 = oracle_grads(sd0, inp, torch.float32); print("oracle fp32", l32, time.time() - t0, flush=True)
    t0 = time.time(); l16, dh16, g16 = oracle_grads(sd0, inp, torch.bfloat16); print("oracle bf16", l16, time.time() - t0, flush=True)
    head = E.NextDiTSftHead(sd0, dev)
    t0 = time.time()
    loss, dh = head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["t_index"])
    torch.cuda.synchronize(); print("engine", loss.item(), time.time() - t0, flush=True)
    def rel(a, b): return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
    print(f"dh: engine {rel(dh.float().cpu().view_as(dh32), dh32):.3e} bf16 {rel(dh16, dh32):.3e}")
    worse = []
    groups = {}
    for k, ref in g32.items():
        if ref is None or k not in head.P or k.endswith('norm_k.bias'): continue   # d/d(norm_k.bias) is identically 0 (softmax shift invariance)
        e = rel(head.P.grad(k).cpu().view_as(ref), ref); y = rel(g16[k], ref)
        grp = ".".join(k.split(".")[:2]) if not k.startswith("traj_dit") else "traj_dit." + (k.split(".")[2] if "layers" not in k else "layers")
        groups.setdefault(grp, []).append((e, y, k))
        if e > y: worse.append((e, y, k))
    for grp, v in groups.items():
        print(f"{grp:40s} n={len(v):3d} engine mean {sum(a for a,_,_ in v)/len(v):.3e} max {max(a for a,_,_ in v):.3e} | bf16 mean {sum(b for _,b,_ in v)/len(v):.3e} max {max(b for _,b,_ in v):.3e}")
    print("params where engine error > bf16-PyTorch error:", len(worse))
    for e, y, k in sorted(worse, reverse=True)[:25]: print(f"   {k:70s} {e:.3e} vs {y:.3e}")

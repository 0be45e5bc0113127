"""CPU only: the bf16-autocast PyTorch yardstick of the NavDPNet B = 64 spot check (tests/test_b64_spotcheck_gpu.py) - the fp32 oracle against the
same oracle under torch.autocast(bfloat16), per env of the seeded 64-env batch. Usage: python tools/navdp_yardstick.py > profiles/<tag>_navdp_bf16_yardstick_cpu.log"""
import time, torch, sys, json
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from oracle import navdp as o_navdp
from oracle import weights as W
torch.set_num_threads(8)
B, cfg = 64, W.NAVDPNET_CFG
sd = W.navdpnet_state_dict(seed=21)
inp = W.navdpnet_inputs(B, seed=21)
rows=[]
for b in range(0,64):
    args=(sd, inp["goal"][b:b+1], inp["images"][b:b+1], inp["depths"][b:b+1], inp["x_init"][b:b+1], inp["step_noise"][:, b:b+1], cfg)
    with torch.no_grad():
        _,_,f32,c32,_ = o_navdp.navdpnet_pointgoal(*args, return_all=True)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        _,_,f16,c16,_ = o_navdp.navdpnet_pointgoal(*args, return_all=True)
    e=(f16.float()-f32).abs(); ec=(c16.float()-c32).abs()
    rows.append((b, e.mean().item(), e.max().item(), ec.max().item(), c32.abs().max().item()))
    print(f"env {b:2d}: samples mean|err| {e.mean():.3e} max {e.max():.3e}; critic max|err| {ec.max():.3e} (range {c32.abs().max():.2f})", flush=True)
import statistics
mx=[r[2] for r in rows]; mn=[r[1] for r in rows]
print(f"# over 64 envs: mean|err| median {statistics.median(mn):.3e} max {max(mn):.3e}; max|err| median {statistics.median(mx):.3e}, 90th pct {sorted(mx)[57]:.3e}, max {max(mx):.3e}; envs with max|err| > 5e-2: {[r[0] for r in rows if r[2]>5e-2]}")

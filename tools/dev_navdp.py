import sys, time, torch
sys.path.insert(0, ".")
from oracle import weights as W
from internnav_amd.navdp import NavDPNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = W.navdpnet_state_dict(seed=0)
inp = {k: v.cuda() for k, v in W.navdpnet_inputs(B, seed=0).items()}
net = NavDPNet(sd, W.NAVDPNET_CFG, "cuda:0", max_envs=B)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    neg, pos = net.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    torch.cuda.synchronize(); print("eager call", B, "envs:", time.time() - t0, "s", flush=True)

"""CPU: the drop-in boundary exercised through the REFERENCE'S OWN registry code (VERDICT r1 items b / b2).

`internnav/agent/base.py`, `internnav/model/__init__.py` and `internnav/configs/agent` are imported from /root/reference (package
`__init__`s that pull gym / habitat are bypassed with namespace stubs, as oracle/ref_loader.py does); `internnav_amd.register_all()`
installs this package into them and `Agent.init(AgentCfg(...))` - the call the evaluator and the AgentServer make
(internnav/agent/base.py:40-45, utils/comm_utils/server.py:43-49) - must produce a working agent from the config ALONE.
The HIP engines need a GPU, so here the checkpoint loader is replaced by a scripted model; the GPU twin of this test
(tests/test_agent_gpu.py) loads a real synthetic checkpoint from disk with `from_pretrained`.
Config #1 of BASELINE.json (CMA-style baseline, batch 1, CPU, "plumbing"): a small ResNet-ish + GRU policy with the reference's
`forward(batch)` contract (cma_policy.py:331-341) is pushed through the same registry and driven by `agent.step` and the Habitat
default evaluator's `agent.act(obs, env, info)` (habitat_default_evaluator.py:117).
"""
import importlib
import importlib.util
import json
import sys
import types
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ref_loader as R

pytestmark = pytest.mark.skipif(not R.available(), reason="/root/reference is only mounted in the build container")


@pytest.fixture(scope="module")
def ref():
    """the reference's real registry modules."""
    R.setup()
    r = R.REF / "internnav"
    spec = importlib.util.spec_from_file_location("internnav.model", str(r / "model" / "__init__.py"), submodule_search_locations=[str(r / "model")])
    real_model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(real_model)               # only defines get_policy / get_config (imports are inside the functions)
    old = sys.modules.get("internnav.model")
    for k, v in vars(old).items():                    # keep what ref_loader already attached to the namespace stub
        if not k.startswith("__") and not hasattr(real_model, k):
            setattr(real_model, k, v)
    sys.modules["internnav.model"] = real_model
    sys.modules["internnav"].model = real_model
    if "internnav.agent" not in sys.modules:
        m = types.ModuleType("internnav.agent")
        m.__path__ = [str(r / "agent")]
        sys.modules["internnav.agent"] = m
    base = importlib.import_module("internnav.agent.base")
    cfg = importlib.import_module("internnav.configs.agent")
    return SimpleNamespace(Agent=base.Agent, AgentCfg=cfg.AgentCfg, model=real_model)


SETTINGS = {   # scripts/eval/configs/h1_internvla_n1_async_cfg.py + what vln_default_config.py:314-317 merges in (internvla_n1_cfg.model_dump())
    "policy_name": "InternVLAN1_Policy", "state_encoder": None, "env_num": 2, "sim_num": 1, "model_path": "checkpoints/InternVLA-N1-DualVLN",
    "camera_intrinsic": [[585.0, 0.0, 320.0], [0.0, 585.0, 240.0], [0.0, 0.0, 1.0]], "width": 640, "height": 480, "hfov": 79,
    "resize_w": 384, "resize_h": 384, "max_new_tokens": 1024, "num_frames": 32, "num_history": 8, "num_future_steps": 4, "device": "cuda:0",
    "predict_step_nums": 32, "continuous_traj": True, "infer_mode": "partial_async", "vis_debug": False, "vis_debug_path": "./logs/x"}


class _ScriptedModel:
    """stands in for the HIP-backed InternVLAN1ForCausalLM: answers are scripted token strings, trajectories a straight line."""
    device = torch.device("cpu")
    config = SimpleNamespace(system1="nextdit_async", n_query=4)

    def __init__(self, answers):
        self.answers, self.calls = list(answers), []

    def eval(self):
        return self

    def generate(self, input_ids=None, pixel_values=None, image_grid_thw=None, **kw):
        assert kw["max_new_tokens"] == 128 and kw["do_sample"] is False and kw["return_dict_in_generate"] is True
        B = input_ids.shape[0]
        self.calls.append(("generate", B))
        out = []
        for _ in range(B):
            a = self.answers.pop(0)
            if isinstance(a, Exception):
                raise a
            out.append(torch.tensor([ord(c) for c in a]))
        n = max(len(o) for o in out)
        out = torch.stack([torch.cat([o, torch.zeros(n - len(o), dtype=torch.long)]) for o in out])
        return SimpleNamespace(sequences=torch.cat([input_ids, out], 1))

    def generate_latents(self, ids, pv, grid, rows=None):
        self.calls.append(("latents", ids.shape[0]))
        return torch.zeros(ids.shape[0] if rows is None else len(rows), 4, 8)

    def generate_traj(self, traj_latents=None, images_dp=None, depths_dp=None):
        self.calls.append(("traj", traj_latents.shape[0], tuple(images_dp.shape)))
        t = torch.zeros(32 * traj_latents.shape[0], 32, 3)
        t[:, :, 0] = 0.4          # x4-scaled 0.1 m forward increments
        return t


class _Tok:
    def __call__(self, texts, return_tensors="pt"):
        return {"input_ids": torch.tensor([[ord(c) % 251 for c in texts[0]][:64]])}

    def decode(self, ids, skip_special_tokens=True):
        return "".join(chr(int(i)) for i in ids if int(i) > 0)


class _Proc:
    tokenizer = _Tok()
    image_token = "<|image_pad|>"

    def apply_chat_template(self, conv, tokenize=False, add_generation_prompt=True):
        return "".join("<|image_pad|>" if c["type"] == "image" else c["text"] for m in conv for c in m["content"])

    def __call__(self, text, images, return_tensors="pt"):
        n = len(images)
        return {"input_ids": self.tokenizer(text)["input_ids"], "pixel_values": torch.zeros(4 * n, 1176),
                "image_grid_thw": torch.tensor([[1, 2, 2]] * n)}


def _obs(n=1):
    return [{"rgb": np.zeros((480, 640, 3), np.uint8), "depth": np.zeros((480, 640, 1), np.float32), "instruction": "walk to the door"} for _ in range(n)]


def test_agent_init_from_config_alone_through_the_reference_registry(ref, monkeypatch):
    import internnav_amd
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1ModelConfig, InternVLAN1Net

    agent_cls = internnav_amd.register_all()
    assert ref.Agent.agents["internvla_n1"] is InternVLAN1Agent is agent_cls
    # the reference's factories now resolve our policy; other names still reach the reference's own branches
    assert ref.model.get_policy("InternVLAN1_Policy") is InternVLAN1Net and ref.model.get_config("InternVLAN1_Policy") is InternVLAN1ModelConfig
    with pytest.raises(ValueError, match="not found"):
        ref.model.get_policy("no_such_policy")                       # internnav/model/__init__.py:30
    internnav_amd.register_all()                                     # idempotent
    loaded = []
    model = _ScriptedModel(["215 206", "↑↑", "12 34", "7"] + [RuntimeError("boom")] * 2)

    def fake_load(ms):
        loaded.append(dict(ms))
        return model, _Proc()

    monkeypatch.setattr(InternVLAN1Net, "_load", classmethod(lambda cls, ms: fake_load(ms)))
    cfg = ref.AgentCfg(server_port=8023, model_name="internvla_n1", ckpt_path="", model_settings=dict(SETTINGS))
    agent = ref.Agent.init(cfg)                                      # internnav/agent/base.py:40-45 -> cls(config)
    assert isinstance(agent, InternVLAN1Agent) and agent.mode == "partial_async"
    assert loaded and loaded[0]["model_path"] == SETTINGS["model_path"] and loaded[0]["device"] == "cuda:0" and loaded[0]["num_history"] == 8
    agent.reset()
    out = agent.step(_obs(2))                                        # env 0: pixel goal -> S1; env 1: two forward arrows
    assert [o["action"] for o in out] == [[1], [1]] and all(o["ideal_flag"] is True for o in out)
    json.dumps(out)                                                  # AgentServer serialises the result (comm_utils/server.py:62)
    assert ("generate", 2) in model.calls or model.calls.count(("generate", 1)) == 2
    assert ("traj", 1, (1, 2, 224, 224, 3)) in model.calls           # look-down pair resized to 224 x 224 (internvla_n1_agent.py:309-333)
    agent.reset([1])                                                 # VLNDistributedEvaluator.terminate_ops -> agent.reset(ids) (:196)
    out = agent.step(_obs(2))                                        # env 1 restarts: S2 "12 34" -> pixel goal -> S1; env 0 continues its plan
    assert [o["action"] for o in out] == [[1], [1]]
    # one-number pixel goal: IndexError inside s2_step (internvla_n1_policy.py:187) -> reset + retry without look-down -> the retry's
    # generate raises -> STOP, never an exception out of step() (:168-189)
    agent.reset([0])
    out = agent.step(_obs(1))
    assert out[0]["action"] == [0] and agent.s2_failures == 1


def test_capacity_and_engine_errors_are_not_swallowed(ref, monkeypatch):
    from internnav_amd._lib import EngineError
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1Net
    from internnav_amd.runtime import CapacityError

    for exc in (CapacityError("2155 tokens > max_seq_len"), EngineError("hip launch failed")):
        model = _ScriptedModel([exc])
        monkeypatch.setattr(InternVLAN1Net, "_load", classmethod(lambda cls, ms, model=model: (model, _Proc())))
        agent = InternVLAN1Agent(ref.AgentCfg(model_name="internvla_n1", model_settings=dict(SETTINGS)))
        agent.reset()
        with pytest.raises(type(exc)):
            agent.step(_obs(1))


def test_sync_mode_with_an_async_checkpoint_is_rejected_at_construction(ref, monkeypatch):
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1Net

    monkeypatch.setattr(InternVLAN1Net, "_load", classmethod(lambda cls, ms: (_ScriptedModel([]), _Proc())))
    with pytest.raises(ValueError, match="partial_async"):
        InternVLAN1Agent(ref.AgentCfg(model_name="internvla_n1", model_settings=dict(SETTINGS, infer_mode="sync")))


def test_s2_and_s1_batches_are_chunked_to_engine_capacity(ref, monkeypatch):
    from internnav_amd.agent import InternVLAN1Agent
    from internnav_amd.policy import InternVLAN1Net

    model = _ScriptedModel(["215 206"] * 5)
    model.qwen = SimpleNamespace(B_max=2)
    model.s1 = SimpleNamespace(b_max=3)
    monkeypatch.setattr(InternVLAN1Net, "_load", classmethod(lambda cls, ms: (model, _Proc())))
    agent = InternVLAN1Agent(ref.AgentCfg(model_name="internvla_n1", model_settings=dict(SETTINGS, env_num=5)))
    agent.reset()
    out = agent.step(_obs(5))
    assert [o["action"] for o in out] == [[1]] * 5
    assert [c[1] for c in model.calls if c[0] == "generate"] == [2, 2, 1]
    assert [c[1] for c in model.calls if c[0] == "traj"] == [3, 2]


# ---------------------------------------------------------------------------------------------- BASELINE config #1: CMA-style plumbing
class _CMAStyleNet(torch.nn.Module):
    """ResNet-ish RGB / depth encoders + 1-layer GRU + action head with the reference's CMANet call contract (cma_policy.py:331-341):
    forward(batch) with batch = {mode, observations, rnn_states, prev_actions, masks} -> (actions, rnn_states, progress)."""

    def __init__(self, config):
        super().__init__()
        self.model_config = config.model_cfg["model"]
        h = self.model_config["state_encoder"]["hidden_size"]

        def enc(c):
            return torch.nn.Sequential(torch.nn.Conv2d(c, 8, 7, 4, 3), torch.nn.ReLU(), torch.nn.Conv2d(8, 16, 3, 2, 1), torch.nn.ReLU(),
                                       torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten())

        self.rgb, self.depth = enc(3), enc(1)
        self.instr = torch.nn.Embedding(64, 16)
        self.gru = torch.nn.GRU(16 + 16 + 16 + 4, h, batch_first=True)
        self.head = torch.nn.Linear(h, 4)
        self.progress = torch.nn.Linear(h, 1)

    def forward(self, batch):
        assert batch["mode"] == "act"
        o = batch["observations"]
        x = torch.cat([self.rgb(o["rgb"].permute(0, 3, 1, 2).float() / 255.0), self.depth(o["depth"].permute(0, 3, 1, 2)),
                       self.instr(o["instruction"]).mean(1), torch.nn.functional.one_hot(batch["prev_actions"][:, 0], 4).float()], -1)
        h0 = (batch["rnn_states"] * batch["masks"][:, None]).permute(1, 0, 2).contiguous()
        y, h1 = self.gru(x[:, None], h0)
        return self.head(y[:, 0]).argmax(-1, keepdim=True), h1.permute(1, 0, 2), self.progress(y[:, 0])


class _CMAStyleConfig:
    def __init__(self, model_cfg):
        self.model_cfg = model_cfg


def test_config1_cma_style_policy_through_the_registry(ref):
    """BASELINE configs[0]: batch 1, CPU PyTorch, no GPU - the registry / forward(batch) / step / act plumbing only."""
    import internnav_amd

    internnav_amd.register_all()
    internnav_amd._table()["CMA_Policy_stub"] = (_CMAStyleNet, _CMAStyleConfig)
    get_policy, get_config = ref.model.get_policy, ref.model.get_config
    name = "cma_plumbing_stub"
    ref.Agent.agents.pop(name, None)

    @ref.Agent.register(name)                                         # the reference's decorator (internnav/agent/base.py:26-38)
    class CmaStyleAgent(ref.Agent):
        """the reference's CmaAgent wiring (cma_agent.py:25-40, 100-131) reduced to its plugin calls."""

        def __init__(self, config):
            super().__init__(config)
            ms = config.model_settings
            torch.manual_seed(0)
            self.policy = get_policy(ms["policy_name"])(config=get_config(ms["policy_name"])(model_cfg={"model": ms}))
            self.policy.eval()
            self.reset()

        def reset(self, reset_index=None):
            self.rnn = torch.zeros(1, 1, self.config.model_settings["state_encoder"]["hidden_size"])
            self.prev = torch.zeros(1, 1, dtype=torch.long)
            self.mask = torch.zeros(1)

        def _batch(self, obs):
            o = {"rgb": torch.from_numpy(obs["rgb"])[None], "depth": torch.from_numpy(obs["depth"])[None],
                 "instruction": torch.tensor([[ord(c) % 64 for c in obs["instruction"]]])}
            return {"mode": "act", "observations": o, "rnn_states": self.rnn, "prev_actions": self.prev, "masks": self.mask}

        def step(self, obs):
            with torch.no_grad():
                actions, self.rnn, _ = self.policy.forward(self._batch(obs[0]))
            self.prev, self.mask = actions, torch.ones(1)
            return [{"action": [int(actions[0, 0])], "ideal_flag": True}]

        def act(self, obs, env=None, info=None):                       # habitat_default_evaluator.py:117
            return self.step([obs])[0]["action"][0]

    with pytest.raises(ValueError, match="already registered"):
        ref.Agent.register(name)(CmaStyleAgent)
    cfg = ref.AgentCfg(model_name=name, model_settings={"policy_name": "CMA_Policy_stub", "state_encoder": {"hidden_size": 32, "rnn_type": "GRU"}})
    agent = ref.Agent.init(cfg)
    obs = _obs(1)[0]
    obs["rgb"] = np.random.default_rng(0).integers(0, 255, (64, 64, 3), dtype=np.uint8)
    obs["depth"] = np.random.default_rng(1).random((64, 64, 1), dtype=np.float32)
    a1 = [agent.step([obs])[0]["action"][0] for _ in range(3)]
    agent.reset()
    a2 = [agent.act(obs, env=None, info={}) for _ in range(3)]
    assert a1 == a2 and all(a in (0, 1, 2, 3) for a in a1)             # deterministic, recurrent state threaded and reset
    internnav_amd._table().pop("CMA_Policy_stub")


# ---------------------------------------------------------------------------------------------- f2: the real-world async driver, GPU-less
def test_async_agent_driver_follows_the_reference_cadence():
    """internvla_n1_agent_realworld.py:125-164 on a scripted model: S2 on the first frame / after PLAN_STEP_GAP frames / on a look-down
    frame, discrete answers returned once, a pixel goal followed by one continuous System-1 trajectory per frame."""
    from internnav_amd.async_agent import InternVLAN1AsyncAgent

    model = _ScriptedModel(["↓", "215 206", "↑↑→"])
    ag = InternVLAN1AsyncAgent(SimpleNamespace(device="cpu", model_path="unused", resize_w=64, resize_h=64, num_history=4, plan_step_gap=3),
                               model=model, processor=_Proc())
    rgb, depth = np.zeros((48, 64, 3), np.uint8), np.zeros((48, 64), np.float32)
    out = ag.step(rgb, depth, None, "walk to the door", None)
    assert out.output_action == [5] and out.output_trajectory is None          # look-down request, returned once
    out = ag.step(rgb, depth, None, "walk to the door", None, look_down=True)   # the caller looks down and says so
    assert out.output_pixel == [206, 215] and out.output_trajectory is not None and out.output_action is None
    assert out.output_trajectory.shape == (33, 2) and abs(out.output_trajectory[-1, 0] - 3.2) < 1e-5   # 32 x 0.1 m straight ahead
    n_gen = sum(1 for c in model.calls if c[0] == "generate")
    for _ in range(4):                                                          # episode_idx - last_s2_idx <= PLAN_STEP_GAP: System-1 only, history grows
        out = ag.step(rgb, depth, None, "walk to the door", None)
        assert out.output_trajectory is not None and out.output_pixel is None
    assert sum(1 for c in model.calls if c[0] == "generate") == n_gen and len(ag.rgb_list) == 5
    out = ag.step(rgb, depth, None, "walk to the door", None)                   # older than PLAN_STEP_GAP -> System-2 again
    assert out.output_action == [1, 1, 3] and sum(1 for c in model.calls if c[0] == "generate") == n_gen + 1
    v, w = ag.trajectory_tovw(np.array([[0.0, 0.0, 0.0], [3.0, 4.0, 0.9]]))
    assert v == 0.5 and w == 0.5
    ag.reset()
    assert ag.episode_idx == 0 and ag.output_latent is None

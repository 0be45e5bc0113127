"""TEST INFRASTRUCTURE ONLY - tests/golden/agent_trace.json: the reference's OWN agent + policy host logic executed on a script.

Runs in the build container only (needs /root/reference). Imports, unmodified:
  * internnav/agent/internvla_n1_agent.py   (InternVLAN1Agent: step :243-407, should_infer_s2 :210-241, the S2 thread :133-208),
  * internnav/model/basemodel/internvla_n1/internvla_n1_policy.py (InternVLAN1Net: s2_step :110-200, step_no_infer, s1_step_latent),
  * internnav/model/utils/vln_utils.py (traj_to_actions, split_and_clean), internnav/agent/base.py, internnav/configs/*, internnav/model/__init__.py
under namespace stubs for what this image lacks (cv2, imageio, gym, the logger), with `InternVLAN1ForCausalLM.from_pretrained`,
`AutoProcessor.from_pretrained` and `AutoTokenizer.from_pretrained` answering with the scripted objects of oracle/agent_script.py.
The agent's `time.sleep` polling (0.5 s / 0.2 s) is shortened 100x; nothing else is touched.

Per scenario the fixture holds: the action dict of every step, and everything the scripted model was handed, in order (`tape`): the
exact chat text of every System-2 call, the sha1 of every image (bytes after the policy's PIL handling), the generate kwargs, the
text + images of every generate_latents call, and sha1s of the float32 System-1 inputs (224x224 look-down pairs: PIL resize / 255,
depth x10 clipped at 5).  `python -m oracle.make_golden_agent` rewrites it.
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import sys
import time as _time
import types
from pathlib import Path
from types import SimpleNamespace

import torch

from . import agent_script as A
from . import ref_loader as R

GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden"

SETTINGS = {   # scripts/eval/configs/h1_internvla_n1_async_cfg.py + internvla_n1_cfg.model_dump() (vln_default_config.py:314-317)
    "policy_name": "InternVLAN1_Policy", "state_encoder": None, "env_num": 1, "sim_num": 1, "model_path": "checkpoints/InternVLA-N1-DualVLN",
    "camera_intrinsic": [[585.0, 0.0, 320.0], [0.0, 585.0, 240.0], [0.0, 0.0, 1.0]], "width": 640, "height": 480, "hfov": 79,
    "resize_w": 384, "resize_h": 384, "max_new_tokens": 1024, "num_frames": 32, "num_history": 8, "num_future_steps": 4, "device": "cpu",
    "predict_step_nums": 32, "continuous_traj": True, "infer_mode": "partial_async", "vis_debug": False, "vis_debug_path": "./logs/x"}


def load_reference_agent():
    """the reference's agent + policy modules, real code, with the three from_pretrained loaders answering from `holder`."""
    R.setup()
    r = R.REF / "internnav"
    for name, attrs in (("imageio", {"get_writer": lambda *a, **k: None}),
                        ("gym", {}), ("gym.spaces", {"Box": lambda **kw: SimpleNamespace(**kw)})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    for name in ("internnav.utils", "internnav.utils.common_log_util"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    import logging

    sys.modules["internnav.utils.common_log_util"].common_logger = logging.getLogger("ref")
    # the real internnav/model/__init__.py (get_policy / get_config) in place of ref_loader's namespace stub
    spec = importlib.util.spec_from_file_location("internnav.model", str(r / "model" / "__init__.py"), submodule_search_locations=[str(r / "model")])
    real_model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(real_model)
    old = sys.modules.get("internnav.model")
    for k, v in vars(old).items():
        if not k.startswith("__") and not hasattr(real_model, k):
            setattr(real_model, k, v)
    sys.modules["internnav.model"] = real_model
    sys.modules["internnav"].model = real_model
    if "internnav.agent" not in sys.modules:
        m = types.ModuleType("internnav.agent")
        m.__path__ = [str(r / "agent")]
        sys.modules["internnav.agent"] = m
    # ref_loader may have replaced ModelCfg by a lambda for other fixtures: the agent needs the real pydantic class (model_dump)
    for name in ("internnav.configs.model.base_encoders", "internnav.configs.model"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", None):
            del sys.modules[name]
    pol = importlib.import_module("internnav.model.basemodel.internvla_n1.internvla_n1_policy")
    holder = SimpleNamespace(model=None, processor=None)
    pol.InternVLAN1ForCausalLM = SimpleNamespace(from_pretrained=lambda *a, **k: holder.model)
    pol.AutoProcessor = SimpleNamespace(from_pretrained=lambda *a, **k: holder.processor)
    pol.AutoTokenizer = SimpleNamespace(from_pretrained=lambda *a, **k: holder.processor.tokenizer)
    ag = importlib.import_module("internnav.agent.internvla_n1_agent")
    ag.time = SimpleNamespace(sleep=lambda s: _time.sleep(s / 100.0))
    base = importlib.import_module("internnav.agent.base")
    cfg = importlib.import_module("internnav.configs.agent")
    return SimpleNamespace(Agent=base.Agent, AgentCfg=cfg.AgentCfg, agent_mod=ag, policy_mod=pol, holder=holder)


def run_reference(ref, name: str, scn: dict) -> dict:
    model = A.ScriptedModel({name: scn}, system1="nextdit" if scn["mode"] == "sync" else "nextdit_async")
    ref.holder.model, ref.holder.processor = model, A.ScriptedProcessor()
    ms = dict(SETTINGS, infer_mode=scn["mode"])
    agent = ref.Agent.init(ref.AgentCfg(model_name="internvla_n1", ckpt_path="", model_settings=ms))   # internnav/agent/base.py:40-45
    assert type(agent).__module__ == "internnav.agent.internvla_n1_agent" and type(agent.policy).__module__.endswith("internvla_n1_policy")
    agent.reset()
    steps, raised = [], None
    for t in range(scn["steps"]):
        obs = A.make_obs(name, t, scn["instruction"])
        try:
            out = agent.step([obs])
        except Exception as ex:  # noqa: BLE001 - the "garbage" scenario: IndexError out of step() (internvla_n1_agent.py:282)
            raised = {"step": t, "type": type(ex).__name__}
            break
        assert len(out) == 1 and out[0]["ideal_flag"] is True
        steps.append([int(a) for a in out[0]["action"]])
    assert not model.exhausted, f"scenario {name}: the script ran out"
    return {"mode": scn["mode"], "instruction": scn["instruction"], "actions": steps, "raises": raised, "tape": model.tape[name],
            "answers_left": len(model.scripts[name]["answers"]), "trajs_left": len(model.scripts[name]["trajs"])}


def main():
    ref = load_reference_agent()
    import contextlib
    import io

    out = {}
    for name, scn in A.SCENARIOS.items():
        t0 = _time.time()
        with contextlib.redirect_stdout(io.StringIO()):       # the reference prints every step
            out[name] = run_reference(ref, name, scn)
        r = out[name]
        kinds = [e["kind"] for e in r["tape"]]
        print(f"[agent_trace] {name}: {len(r['actions'])} steps, {kinds.count('s2')} S2 calls, {kinds.count('latents')} latent calls, "
              f"{kinds.count('s1')} S1 calls, raises={r['raises']}, actions={''.join(str(a[0]) if a[0] >= 0 else 'D' for a in r['actions'])} "
              f"({_time.time() - t0:.1f} s)", flush=True)
        assert len(r["actions"]) >= 40 or r["raises"], name
    GOLD.mkdir(parents=True, exist_ok=True)
    (GOLD / "agent_trace.json").write_text(json.dumps({"settings": SETTINGS, "scenarios": out}, ensure_ascii=False, indent=0))
    print(f"wrote {GOLD / 'agent_trace.json'} ({(GOLD / 'agent_trace.json').stat().st_size} bytes)")


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()

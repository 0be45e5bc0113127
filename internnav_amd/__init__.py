"""internnav_amd: MI355X-native InternVLA-N1 policy inference behind InternNav's agent / model plugin surface.

`register_all()` is the one call a harness adds: it installs the batched HIP-backed agent under 'internvla_n1' in
the reference's `Agent` registry and routes `get_policy` / `get_config` of `internnav.model` for 'InternVLAN1_Policy' and
'NavDP_Policy' to this package's classes; every other policy name keeps going to the reference's own factory
(internnav/model/__init__.py:1-56), so CMA / RDP / Seq2Seq agents are untouched.
"""
from __future__ import annotations

import sys

_POLICIES = {}


def _table():
    if not _POLICIES:
        from .navdp import NavDPModelConfig, NavDPNet
        from .policy import InternVLAN1ModelConfig, InternVLAN1Net

        _POLICIES["InternVLAN1_Policy"] = (InternVLAN1Net, InternVLAN1ModelConfig)
        _POLICIES["NavDP_Policy"] = (NavDPNet, NavDPModelConfig)        # constructed through NavDPNet.from_pretrained(path, config=...)
    return _POLICIES


def get_policy(policy_name: str):
    """this package's counterpart of internnav.model.get_policy (internnav/model/__init__.py:1-30) for the policies it implements."""
    try:
        return _table()[policy_name][0]
    except KeyError:
        raise ValueError(f"Policy {policy_name} not found") from None


def get_config(policy_name: str):
    """counterpart of internnav.model.get_config (internnav/model/__init__.py:33-56)."""
    try:
        return _table()[policy_name][1]
    except KeyError:
        raise ValueError(f"Policy {policy_name} not found") from None


def register_all(agent_name: str = "internvla_n1"):
    """Install into an importable `internnav`: the agent registry entry and the model factories. Idempotent.
    Returns the agent class. Raises ImportError when `internnav` is not importable (nothing to install into)."""
    from internnav.agent.base import Agent

    from .agent import InternVLAN1Agent

    Agent.agents[agent_name] = InternVLAN1Agent            # replaces the PyTorch agent (Agent.register raises on duplicates, base.py:33-34)
    import internnav.model as ref_model

    if not getattr(ref_model.get_policy, "_internnav_amd", False):
        ref_get_policy, ref_get_config = ref_model.get_policy, ref_model.get_config

        def patched_get_policy(policy_name):
            return _table()[policy_name][0] if policy_name in _table() else ref_get_policy(policy_name)

        def patched_get_config(policy_name):
            return _table()[policy_name][1] if policy_name in _table() else ref_get_config(policy_name)

        patched_get_policy._internnav_amd = patched_get_config._internnav_amd = True
        # modules that did `from internnav.model import get_policy, get_config` before this call hold the old functions: rebind them
        for mod in list(sys.modules.values()):
            d = getattr(mod, "__dict__", None)
            if not d:
                continue
            if d.get("get_policy") is ref_get_policy:
                d["get_policy"] = patched_get_policy
            if d.get("get_config") is ref_get_config:
                d["get_config"] = patched_get_config
        ref_model.get_policy, ref_model.get_config = patched_get_policy, patched_get_config
    return InternVLAN1Agent

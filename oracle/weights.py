"""Seeded weights / inputs shared by the oracle, the tests and the engines: defined in internnav_amd/synthetic.py
(host-side tables only, no arithmetic of the policy) and re-exported here for the oracle-side scripts."""
from internnav_amd.synthetic import *  # noqa: F401,F403
from internnav_amd.synthetic import N1_NAVDP_CFG, N1_NEXTDIT_CFG, N1_NEXTDIT_CFG_FFN1024, N1_NEXTDIT_VARIANTS, NAVDPNET_CFG, QWEN_N1_CFG, QWEN_TEST_CFG  # noqa: F401

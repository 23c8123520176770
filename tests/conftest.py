"""Shared pytest configuration.

* registers the ``gpu`` marker (tests that need a real MI355X; the driver runs ``-m gpu`` on the GPU
  box and ``-m "not gpu"`` in the build container),
* puts the repo root (for ``oracle``) and ``pika-zoo_amd/`` (for ``pikazoo_amd``) on ``sys.path``.

Nothing here reads /root/reference: that path does not exist on the GPU box.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "pika-zoo_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the in-tree HIP library exists and is current (hipcc cross-compiles without a GPU).
    This only *builds* the product; nothing falls back to the CPU if the build is impossible."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pz_build", REPO / "pika-zoo_amd" / "build.py")
    pz_build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pz_build)
    try:
        pz_build.build()
    except Exception as exc:  # noqa: BLE001 - the tests that need the library will say so loudly
        print(f"[conftest] could not build libpikazoo_hip.so: {exc}")


def load_golden(name):
    d = dict(np.load(GOLDEN / f"{name}.npz"))
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


FULL_FIXTURES = [
    "cfg2_human_human", "cfg3_p2_computer", "p1_computer", "both_computer", "serve_alternate",
    "serve_random", "winning_score_1", "winning_score_3", "cfg5_wrappers_float",
    "wrappers_int_table", "simplify_only",
    "normal_state_only", "normal_state_inside_ballpos", "normal_state_outside_ballpos",
    "normalize_observation", "record_stats_raw", "full_wrapper_stack", "cfg1_one_env_10k",
]
# wrapper stacks the step kernel cannot fuse: part of the stack runs on the step's outputs (oracle/wrappers_oracle.py,
# pikazoo_amd/wrappers/*.py)
UNFUSED_FIXTURES = ["unfused_ballpos_above_normalize", "unfused_stats_between_reward_wrappers", "unfused_doubled_wrappers"]
DIGEST_FIXTURES = ["digest_human_human", "digest_p2_computer", "digest_both_computer_random_serve"]


def golden_state(d, t):
    """Full int32 [W, lanes] state after step t of a full fixture (rng counter kept as int32)."""
    st = d["states"][t].astype(np.int32)
    st[43] = d["rng_counter"][t]
    return st


def fused_options_from_meta(meta):
    """Fused-kernel options equivalent to the fixture's reference wrapper stack."""
    from oracle.ref_capture import fused_options

    return fused_options(meta["wrappers"])


def apply_product_wrappers(env, wrappers):
    """Wrap a pikazoo_amd env with the same classes, in the same order, as the reference stack."""
    import pikazoo_amd.wrappers as W
    from oracle.ref_capture import wrapper_stack

    for name, kw in wrapper_stack(wrappers):
        kw = dict(kw)
        if "additional_reward" in kw:
            kw["additional_reward"] = tuple(kw["additional_reward"])
        env = getattr(W, name)(env, **kw)
    return env


def oracle_config_from_meta(meta, **over):
    from oracle import pz_oracle as po

    kw = dict(meta["env_kwargs"])
    opt = fused_options_from_meta(meta)
    args = dict(
        winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
        is_player1_computer=kw.get("is_player1_computer", False),
        is_player2_computer=kw.get("is_player2_computer", False),
        seed=meta["seed"], env_id_base=meta["env_id_base"], auto_reset=True, **opt)
    args.update(over)
    return po.make_config(**args)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pz_oracle as po

    po.build()
    return po

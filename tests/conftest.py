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


def load_golden(name):
    d = dict(np.load(GOLDEN / f"{name}.npz"))
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


FULL_FIXTURES = [
    "cfg2_human_human", "cfg3_p2_computer", "p1_computer", "both_computer", "serve_alternate",
    "serve_random", "winning_score_1", "winning_score_3", "cfg5_wrappers_float",
    "wrappers_int_table", "simplify_only",
]
DIGEST_FIXTURES = ["digest_human_human", "digest_p2_computer", "digest_both_computer_random_serve"]


def golden_state(d, t):
    """Full int32 [W, lanes] state after step t of a full fixture (rng counter kept as int32)."""
    st = d["states"][t].astype(np.int32)
    st[43] = d["rng_counter"][t]
    return st


def oracle_config_from_meta(meta, **over):
    from oracle import pz_oracle as po

    kw = dict(meta["env_kwargs"])
    wr = meta["wrappers"] or {}
    args = dict(
        winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
        is_player1_computer=kw.get("is_player1_computer", False),
        is_player2_computer=kw.get("is_player2_computer", False),
        simplify_action=bool(wr.get("simplify_action")),
        additional_reward=wr.get("additional_reward"), x_line=wr.get("x_line", 216),
        y_line=wr.get("y_line", 176), seed=meta["seed"], env_id_base=meta["env_id_base"],
        auto_reset=True)
    args.update(over)
    return po.make_config(**args)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pz_oracle as po

    po.build()
    return po
